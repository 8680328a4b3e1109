// emu_runners.hpp -- TEST-ONLY: the host contexts and runner objects that execute the specialised workgroup bodies
// sequentially (see emu.cpp).  Shared by emu.cpp and the per-group translation units emu_rows_g*.cpp / emu_cols_g*.cpp,
// which instantiate one group of configurations of fast_paths.hpp each (build time only).
#pragma once
#include <vector>

#include "pipeline.hpp"

namespace emu {
using namespace fc;

struct HostCtx {
    int tid = 0, nthreads = 1;
    void sync() const {}
};

// Phase-structured bodies (fast_rows.hpp): every phase is run for all NT threads before the
// next one starts, with one State per emulated thread.
template <class State>
struct HostPhaseCtx {
    int NT;
    std::vector<State> st;
    explicit HostPhaseCtx(int nt) : NT(nt), st(nt) {}
    template <class F>
    void phase(F&& f) {
        for (int t = 0; t < NT; t++) f(t, st[t]);
    }
    template <class F>
    void phase_nosync(F&& f) {
        for (int t = 0; t < NT; t++) f(t, st[t]);
    }
    template <bool NOSYNC, class F>
    void phase_dbg(F&& f) {
        for (int t = 0; t < NT; t++) f(t, st[t]);
    }
    // the GPU's lane exchange (DPP row_ror:8): the peer's value was produced in an earlier phase
    template <class Acc>
    c32 peer8(int t, Acc&& acc) {
        return acc(st[t ^ 8]);
    }
};

struct EmuFastRows {
    const FastRowsArgs& a;
    c32* lds;
    int rows;
    int group = 0;   // > 1: multi-map body; the emulator holds one kernel at a time, so the walk
                     // over `group` maps is emulated with the same kernel (strides 0): the loop,
                     // the prefetch slot and the LDS reuse are exercised, the indexing is not
    template <class Cfg, int NZ2>
    void go() {
        {   // (always the walk over maps / (map, feature) pairs, as the product's launchers: a walk of one map where group <= 1)
            const int group = this->group > 1 ? this->group : 1;
            FastRowsArgs b = a;
            b.a_kernel_stride = 0;
            b.y_kernel_stride = 0;
            for (int grp = 0; grp < (rows + Cfg::RPW - 1) / Cfg::RPW; grp++) {
                for (int i = 0; i < Cfg::LDS_ELEMS; i++) lds[i] = mk(1e30f, -1e30f);
                // (as the product's launcher: LINEAR is the only variant of a configuration whose m1 is a whole number of tiles)
                constexpr bool ALWAYS_LINEAR = (2 * Cfg::m1) % FC_Y_TILE_W == 0;
                const bool linear = ALWAYS_LINEAR || fast_rows_multi_linear(b, Cfg::L, Cfg::m1);
                if (a.F > 1) {   // the walk over (map, feature) pairs
                    HostPhaseCtx<RowMultiState<Cfg, true>> ctx(Cfg::NT);
                    if (linear) fast_rows_multi_body<Cfg, NZ2, true, true>(ctx, lds, b, grp, 0, group, rows);
                    else if constexpr (!ALWAYS_LINEAR) fast_rows_multi_body<Cfg, NZ2, false, true>(ctx, lds, b, grp, 0, group, rows);
                } else {
                    HostPhaseCtx<RowMultiState<Cfg>> ctx(Cfg::NT);
                    if (linear) fast_rows_multi_body<Cfg, NZ2, true>(ctx, lds, b, grp, 0, group, rows);
                    else if constexpr (!ALWAYS_LINEAR) fast_rows_multi_body<Cfg, NZ2, false>(ctx, lds, b, grp, 0, group, rows);
                }
            }
        }
    }
};

struct EmuFastRowsFwd {
    const FastRowsFwdArgs& a;
    c32* lds;
    int rows;
    template <class Cfg>
    void go() {
        for (int grp = 0; grp < (rows + Cfg::RPW - 1) / Cfg::RPW; grp++) {
            for (int i = 0; i < Cfg::LDS_ELEMS; i++) lds[i] = mk(1e30f, -1e30f);
            HostPhaseCtx<RowFwdState> ctx(Cfg::NT);
            fast_rows_fwd_body<Cfg>(ctx, lds, a, grp, rows);
        }
    }
};

struct EmuFastColsFwd {
    const FastColsFwdArgs& a;
    c32* lds;
    int nwg;
    template <class Cfg, int NZ2>
    void go() {
        for (int wg = 0; wg < nwg; wg++) {
            for (int i = 0; i < Cfg::LDS_ELEMS; i++) lds[i] = mk(1e30f, -1e30f);
            HostPhaseCtx<ColFwdState> ctx(Cfg::NT);
            fast_cols_fwd_body<Cfg, NZ2>(ctx, lds, a, wg, nwg);
        }
    }
};

struct EmuFastCols {
    const FastColsArgs& a;
    c32* lds;
    int nwg;
    template <class Cfg>
    void go() {
        FastColsArgs b = a;
        int sgrid = 0;
        // the emulator uses 8 "persistent workgroups" where the product's launcher would slice the tail round, so that the
        // sliced body runs on the CPU tier too (e.g. 18 tiles = 2 full rounds of 8 + 2 tiles in 4 slices each)
        const bool sliced = (Cfg::M <= FC_SLICE_MAX_M) && fast_cols_slice_plan(Cfg::M, Cfg::T, 8, b, sgrid);
        const int loops = sliced ? 8 : nwg;
        FastColsArgs q = a;            // dynamic tile queue (unsliced launches): counters zeroed, chunks of two tiles
        if (a.queue) q.queue_shift = 1;      // (the counters are zero between launches: the last workgroup out zeroes them)
        b.queue = nullptr;
        for (int wg = 0; wg < loops; wg++) {
            for (int i = 0; i < Cfg::LDS_ELEMS; i++) lds[i] = mk(1e30f, -1e30f);
            if (a.y_tiled) {
                HostPhaseCtx<ColPairState<Cfg>> pctx(Cfg::NT);
                if constexpr (Cfg::M <= FC_SLICE_MAX_M) {
                    if (sliced) {     // the tail round in column slices (fast_cols_slice_plan filled `b`)
                        if (wg < sgrid) fast_cols_body<Cfg, true, true>(pctx, lds, b, wg, sgrid);
                        continue;
                    }
                }
                if (q.queue) fast_cols_body<Cfg, true, false, true>(pctx, lds, q, wg, nwg);
                else fast_cols_body<Cfg, true>(pctx, lds, q, wg, nwg);
                continue;
            }
            HostPhaseCtx<ColState<Cfg>> ctx(Cfg::NT);
            q.queue = nullptr;         // (row-major intermediate: static deal only, as the product's launcher)
            fast_cols_body<Cfg, false>(ctx, lds, q, wg, nwg);
        }
    }
};


// per-group entry points (defined in emu_rows_g<G>.cpp / emu_cols_g<G>.cpp); false: no configuration of that group matches
#define EMU_DECL_ROWS(G)                                                   \
    bool fast_rows_g##G(int L, int nz2, EmuFastRows& run);                 \
    bool fast_rows_fwd_g##G(int L, EmuFastRowsFwd& run);
EMU_DECL_ROWS(0) EMU_DECL_ROWS(1) EMU_DECL_ROWS(2)
#undef EMU_DECL_ROWS
#define EMU_DECL_COLS(G)                                                   \
    bool fast_cols_g##G(int M, int T, EmuFastCols& run);                   \
    bool fast_cols_fwd_g##G(int M, int T, bool pruned, EmuFastColsFwd& run);
EMU_DECL_COLS(0) EMU_DECL_COLS(1)
#undef EMU_DECL_COLS

inline bool run_fast_rows(int L, int nz2, EmuFastRows& run) { return fast_rows_g0(L, nz2, run) || fast_rows_g1(L, nz2, run) || fast_rows_g2(L, nz2, run); }
inline bool run_fast_rows_fwd(int L, EmuFastRowsFwd& run) { return fast_rows_fwd_g0(L, run) || fast_rows_fwd_g1(L, run) || fast_rows_fwd_g2(L, run); }
inline bool run_fast_cols(int M, int T, EmuFastCols& run) { return fast_cols_g0(M, T, run) || fast_cols_g1(M, T, run); }
inline bool run_fast_cols_fwd(int M, int T, bool pruned, EmuFastColsFwd& run) {
    return fast_cols_fwd_g0(M, T, pruned, run) || fast_cols_fwd_g1(M, T, pruned, run);
}
static_assert(FC_ROW_GROUPS == 3 && FC_COL_GROUPS == 2, "one translation unit per group");

}  // namespace emu
