// fc_instrument.hpp -- every ablation / A-B / timeline switch of the kernels, in ONE place.
//
// The product build (csrc/Makefile) never defines FC_INSTRUMENT: all switches below are then fixed at their
// product values and any attempt to set one from the command line is a compile error.  Diagnostic builds
// (tools/build_variant.sh, tools/microbench/*) pass -DFC_INSTRUMENT plus the switches they want; several of them
// produce WRONG RESULTS by design (they remove loads, stores, barriers or arithmetic to time what is left).
#pragma once

#if !defined(FC_INSTRUMENT)
#if defined(FC_COLS_DBG) || defined(FC_ROWSM_DBG) || defined(FC_COLS_TIMELINE) || defined(FC_ROWS_TIMELINE) || \
    defined(FC_ROWS_NO_FOLD) || defined(FC_COLS_SPLIT_GATHER) || defined(FC_COLS_NO_PREWAIT) || defined(FC_COLS_PAIR_TRANSPOSE) || defined(FC_NT_SLOADS) ||              \
    defined(FC_NT_STORES) || defined(FC_NT_LOADS) || defined(FC_NO_PACKED) || defined(FC_MULTIF_S_EARLY) || defined(FC_ROWS_NO_BLOCK_PAD) || \
    defined(FC_COLS_NO_BLOCK_PAD) || defined(FC_ROWS_STAGGER_TICKS) || defined(FC_ROWS_STAGGER_RAMP)
#error "kernel instrumentation switches need -DFC_INSTRUMENT (diagnostic builds only; the product never sets them)"
#endif
#endif

// ---- output-column kernel (fast_cols.hpp)
#ifndef FC_COLS_DBG
#define FC_COLS_DBG 0            // wrong results: 1 skip pair pass, 2 no barriers between stages, 4 no gather loads, 8 no stores,
#endif                           //                16 contiguous gather addresses (tiled mode), 32 extra packed arithmetic while landing
#ifndef FC_COLS_TIMELINE
#define FC_COLS_TIMELINE 0       // 1: one workgroup stamps the 100 MHz wall clock at every phase boundary (tools/cols_timeline.py)
#endif
#ifndef FC_COLS_TIMELINE_BASE
#define FC_COLS_TIMELINE_BASE 0  // first stamped tile of that workgroup (16 tiles are stamped)
#endif
#ifndef FC_COLS_TIMELINE_WG
#define FC_COLS_TIMELINE_WG 0
#endif
#ifndef FC_COLS_SPLIT_GATHER
#define FC_COLS_SPLIT_GATHER 2   // next tile's gather issued in: 0 one go at the start of the tile, 1 halves (start, after stage 3),
#endif                           // 2 thirds (start, after stage 3, between the two rounds of stage 2): 28.3 / 27.4 / 27.2 us per map
#ifndef FC_COLS_PAIR_TRANSPOSE
#define FC_COLS_PAIR_TRANSPOSE 2 // landing order of the bin pairs (fast_cols.hpp: pair_of_unit): 0 consecutive, 1 permuted for 4-column tiles,
#endif                           // 2 also for the 8-column configurations whose consecutive pairs share a bank
#ifndef FC_COLS_NO_BLOCK_PAD
#define FC_COLS_NO_BLOCK_PAD 0   // 1: the dense LDS image of a column (no pad between the stage-1 blocks, no run rotation; fast_cols.hpp: col_layout; A/B)
#endif
#ifndef FC_COLS_NO_PREWAIT
#define FC_COLS_NO_PREWAIT 0     // 1: without the vmcnt(0) ahead of the store burst
#endif

// ---- spectral-row kernels (fast_rows.hpp, fast_rows_multi.hpp)
#ifndef FC_NT_SLOADS
#define FC_NT_SLOADS 0           // 1: streaming loads for the image-spectrum rows
#endif
#ifndef FC_ROWS_NO_BLOCK_PAD
#define FC_ROWS_NO_BLOCK_PAD 0   // 1: the stage-1 blocks of a row m1 cells apart in LDS (no padding against the stage-2 bank conflicts; A/B)
#endif
#ifndef FC_ROWS_STAGGER_TICKS
#define FC_ROWS_STAGGER_TICKS 0  // > 0 (experiment, needs plan option timeline_ptr = a zeroed device buffer of 4096 ints): the k-th workgroup of the launch's first round
#endif                           //   to arrive on its CU (a counter per CU, keyed by XCC_ID and HW_ID) waits k x this many 10-ns ticks before it starts
#ifndef FC_ROWS_STAGGER_RAMP
#define FC_ROWS_STAGGER_RAMP 0   // with FC_ROWS_STAGGER_TICKS: 1 = workgroup b of the first round waits b / 1024 x TICKS (a chip-wide ramp) instead of its rank on its CU x TICKS
#endif
#ifndef FC_ROWS_NO_FOLD
#define FC_ROWS_NO_FOLD 0        // 1: forward stage 1 as a phase of its own for every map
#endif
#ifndef FC_ROWSM_DBG
#define FC_ROWSM_DBG 0           // multi-map kernel, wrong results: 1 P5 without its LDS reads and stage-1 arithmetic, 2 no stores
#endif
// F > 1 walk: register pairs of the image-spectrum row requested BEFORE the forward butterfly (0 .. R3 / 2; the rest
// right after it).  All 11 early (rounds 1-2): 27-50 spilled registers at L = 4224; 0 / 4 / 8 early: none, and
// 56.7 / 53.2 / 54.6 us per map at F = 4 on one box (profiles/r03i_f4_image_row_load_placement.txt)
#ifndef FC_MULTIF_S_EARLY
#define FC_MULTIF_S_EARLY 4
#endif
#ifndef FC_ROWS_TIMELINE
#define FC_ROWS_TIMELINE 0       // 1: one workgroup stamps the wall clock at every phase boundary (tools/rows_timeline.py)
#endif
#ifndef FC_ROWS_TIMELINE_WG
#define FC_ROWS_TIMELINE_WG 1000
#endif

// ---- memory-operation flavours (fc_common.hpp)
#ifndef FC_NT_STORES
#define FC_NT_STORES 1           // 0: plain instead of streaming stores for the intermediate and the maps
#endif
#ifndef FC_NT_LOADS
#define FC_NT_LOADS 0            // 1: streaming loads for the output kernel's gather
#endif
// FC_NO_PACKED (defined / not): scalar instead of packed FP32 complex arithmetic

// ---- stamps (expand to nothing in the product)
#if FC_COLS_TIMELINE && defined(__HIP_DEVICE_COMPILE__)
#define FC_COLS_STAMP(slot) do { if (wg == FC_COLS_TIMELINE_WG && threadIdx.x == 0 && g.timeline && it >= FC_COLS_TIMELINE_BASE && it < FC_COLS_TIMELINE_BASE + 16) g.timeline[(it - FC_COLS_TIMELINE_BASE) * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define FC_COLS_STAMP(slot) ((void)0)
#endif
#if FC_ROWS_TIMELINE && defined(__HIP_DEVICE_COMPILE__)
// (slot 0 of a map also leaves the shader-cycle counter in slot 6: delta s_memtime / delta wall clock x 100 MHz = the clock the workgroup ran at)
#define FC_ROWS_STAMP(slot) do { if (group == FC_ROWS_TIMELINE_WG && kernel0 == 0 && threadIdx.x == 0 && g.timeline && m < 16) { g.timeline[m * 8 + (slot)] = wall_clock64(); if ((slot) == 0) g.timeline[m * 8 + 6] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define FC_ROWS_STAMP(slot) ((void)0)
#endif

// store of the multi-map row kernel: the "no stores" ablation keeps the address arithmetic and a never-true guard
#if (FC_ROWSM_DBG & 2)
#define FC_ROWSM_STORE(ptr, val) do { const ::fc::c32 fc_w_ = (val); if (fc_w_.x == 1.2345e-30f) *(ptr) = fc_w_; } while (0)
#else
#define FC_ROWSM_STORE(ptr, val) FC_STREAM_STORE(ptr, val)
#endif
