// fast_paths.hpp -- registry of the specialised (compile-time) kernel configurations.
//
// The generic kernels (kernels_body.hpp) handle every supported length; the configurations
// listed here replace the two hot kernels for the transform lengths that matter (the BASELINE
// configs), chosen so that every stage keeps most lanes busy and the LDS budget allows
// 3 waves per SIMD.  The planner prefers these lengths (planner.hpp: choose_length).
#pragma once
#include <vector>

#include "fast_cols.hpp"
#include "fast_cols_fwd.hpp"
#include "fast_rows.hpp"
#include "fast_rows_fwd.hpp"
#include "fast_rows_multi.hpp"
#include "planner.hpp"

namespace fc {

// planner factors of the two awkward BASELINE windows (see fast_rows_factor): above ~0.5 the planner prefers the next
// convenient length (1152 / 4224) and crops; exact_window plans use the window's own kernels regardless
constexpr double FC_FACTOR_1088 = 0.60, FC_FACTOR_4160 = 0.60;

// X(L, R1, R2, R3, NT, RPW, NZ2): see fast_rows.hpp.  RPW rows of L = R1 x R2 x R3 points per workgroup of NT threads, chosen so that
// RPW * R1 * R2 (the stage-3 butterflies, one per thread) fill the lanes and stage 2 takes one round where possible; NZ2 = non-zero
// stage-2 inputs the variant reads (kernel width <= NZ2 * R3), listed ascending per length (the dispatcher takes the first that
// covers the kernel).  What shaped the list (history/; A/B files under profiles/):
//   * BASELINE: 4224 = 8.24.22 (cfg3; cfg4's 4160 window runs on it and crops), 2112 = 8.12.22 x2 (cfg5), 1152 = 8.12.12 x2 (cfg2's
//     1088 window), 288 = 4.6.12 x8 (cfg1).  4160 = 8.20.26 and 1088 = 8.17.8 are those two windows' OWN kernels (round 5's search over
//     their radix splits, r05d_native_window_search.txt: 10.16.26 -> 8.20.26 -4 %, 17.4.16 x2 -> 8.17.8 -2 %; 16.10.26, 5.32.26, 8.26.20,
//     20.8.26, 16.13.20, 13.16.20 and 4.17.16, 17.4.16 x3 measured slower): still slower than the convenient length + crop
//     (r05e_native_window_ab.txt), used by exact_window plans only (planner factors above).
//   * Round 4 filled the ladder (r04z_size_sweep.txt): a transform is at most ~1.13 x the padded size per dimension above 1000 pixels.
//   * Long rows: 160 KB of LDS hold two rows of >= 7040 points (one 640- / 768-thread workgroup per CU); 5120 ... 6144 run one row per
//     256-thread workgroup, three workgroups per CU (r04g: -11 ... -25 % against the R1 = 16 forms on 320 / 384 threads).  More than
//     128 VGPRs forbid a second 6-wave workgroup on a CU (round 3), hence no 384-thread configurations.
//   * Stage 3 reads and writes its run of R3 values 16 bytes at a time, lane q's run R3 * 8 bytes after lane q - 1's: 128 bytes at
//     R3 = 16 -- every lane of an LDS access group on the same banks (8-way conflict), 4-way at R3 = 24, 2-way at 12 / 20, none at 10,
//     14, 18, 22, 26.  The R3 = 16 forms went (r04h_row_and_column_configs_ab.txt: rows -11 ... -25 %); 4608 as 8.32.18 and 1152 as
//     8.8.18 x4 measured SLOWER than the R3 = 24 / 12 forms -- bank conflicts are one term, not the whole cost.
//   * Late round 4: 384, 480, 672, 864, 960 (images of 300-900 pixels had only 576 and 768 between 288 and 1088) and 1280, 3360 into the
//     widest gaps above (r04t / r04u size sweeps).  1440 = 8.10.18 x2 (M = 720 = 6.10.12) and 1680 = 10.12.14 x2 (M = 840 = 6.10.14) were
//     built too and dropped: a 1440 x 1440 transform took 402 us per 64 maps against 375 for 1536 x 1536, 1680 x 1680 as long as 1760 x 1760.
//   * A search over every admissible (R1, R2, R3, threads) of each length (tools/config_variant.py, r04i_config_search_*.txt) moved
//     4608 to 12.16.24 (-5 %), 6144 to 16.16.24 (-10 %), 3072 to 16.16.12 (-9 %) and 2560 to 16.16.10 (-14 %): a power-of-two stage 2
//     of 16 points against 24 / 32.  No rule came out of it -- 5120 as 16.16.20 is 6 % SLOWER than 8.32.20, 4608 as 16.16.18 14 % slower
//     than 12.16.24 -- and everything else it tried was within noise (+-3 %) or slower (up to +90 %); the list is what survived.
//   * Radix orders at 4224 (round 2): 8.24.22 25.0 us per map, 12.16.22 25.6, 11.16.24 27.3, 16.12.22 28.0: small R1, fat stage 3.
// Three groups, one translation unit each per kernel family (kernels_rows*_g?.hip): build time only.
#define FC_FAST_ROW_CONFIGS_G0(X)   \
    X(8448, 16, 24, 22, 768, 2, 3)  \
    X(8448, 16, 24, 22, 768, 2, 6)  \
    X(8448, 16, 24, 22, 768, 2, 24) \
    X(7680, 16, 24, 20, 768, 2, 4)  \
    X(7680, 16, 24, 20, 768, 2, 24) \
    X(7040, 10, 32, 22, 640, 2, 3)  \
    X(7040, 10, 32, 22, 640, 2, 6)  \
    X(7040, 10, 32, 22, 640, 2, 32) \
    X(6144, 16, 16, 24, 256, 1, 3)  \
    X(6144, 16, 16, 24, 256, 1, 6)  \
    X(6144, 16, 16, 24, 256, 1, 16) \
    X(5632, 16, 16, 22, 256, 1, 3)  \
    X(5632, 16, 16, 22, 256, 1, 16) \
    X(5120, 8, 32, 20, 256, 1, 4)   \
    X(5120, 8, 32, 20, 256, 1, 7)   \
    X(5120, 8, 32, 20, 256, 1, 32)
#define FC_FAST_ROW_CONFIGS_G1(X)   \
    X(4608, 12, 16, 24, 192, 1, 3)  \
    X(4608, 12, 16, 24, 192, 1, 6)  \
    X(4608, 12, 16, 24, 192, 1, 16) \
    X(4224, 8, 24, 22, 192, 1, 3)   \
    X(4224, 8, 24, 22, 192, 1, 6)   \
    X(4224, 8, 24, 22, 192, 1, 24)  \
    X(4160, 8, 20, 26, 192, 1, 3)   \
    X(4160, 8, 20, 26, 192, 1, 5)   \
    X(4160, 8, 20, 26, 192, 1, 20)  \
    X(3840, 8, 24, 20, 192, 1, 4)   \
    X(3840, 8, 24, 20, 192, 1, 24)  \
    X(3520, 10, 16, 22, 192, 1, 3)  \
    X(3520, 10, 16, 22, 192, 1, 16) \
    X(3360, 10, 24, 14, 256, 1, 5)  \
    X(3360, 10, 24, 14, 256, 1, 10) \
    X(3360, 10, 24, 14, 256, 1, 24) \
    X(3072, 16, 16, 12, 256, 1, 6)  \
    X(3072, 16, 16, 12, 256, 1, 16) \
    X(2816, 8, 16, 22, 192, 1, 3)   \
    X(2816, 8, 16, 22, 192, 1, 16)  \
    X(2560, 16, 16, 10, 256, 1, 7)  \
    X(2560, 16, 16, 10, 256, 1, 16)
#define FC_FAST_ROW_CONFIGS_G2(X)   \
    X(2304, 8, 16, 18, 256, 2, 4)   \
    X(2304, 8, 16, 18, 256, 2, 16)  \
    X(2112, 8, 12, 22, 192, 2, 3)   \
    X(2112, 8, 12, 22, 192, 2, 12)  \
    X(1920, 8, 12, 20, 192, 2, 3)   \
    X(1920, 8, 12, 20, 192, 2, 12)  \
    X(1760, 10, 8, 22, 192, 2, 3)   \
    X(1760, 10, 8, 22, 192, 2, 8)   \
    X(1536, 8, 16, 12, 256, 2, 3)   \
    X(1536, 8, 16, 12, 256, 2, 6)   \
    X(1536, 8, 16, 12, 256, 2, 16)  \
    X(1344, 6, 16, 14, 192, 2, 3)   \
    X(1344, 6, 16, 14, 192, 2, 16)  \
    X(1280, 8, 16, 10, 256, 2, 4)   \
    X(1280, 8, 16, 10, 256, 2, 7)   \
    X(1280, 8, 16, 10, 256, 2, 16)  \
    X(1152, 8, 12, 12, 192, 2, 3)   \
    X(1152, 8, 12, 12, 192, 2, 6)   \
    X(1152, 8, 12, 12, 192, 2, 12)  \
    X(1088, 8, 17, 8, 192, 1, 4)    \
    X(1088, 8, 17, 8, 192, 1, 8)    \
    X(1088, 8, 17, 8, 192, 1, 17)   \
    X(960, 6, 16, 10, 192, 2, 4)    \
    X(960, 6, 16, 10, 192, 2, 7)    \
    X(960, 6, 16, 10, 192, 2, 16)   \
    X(864, 4, 12, 18, 192, 4, 4)    \
    X(864, 4, 12, 18, 192, 4, 12)   \
    X(768, 4, 16, 12, 256, 4, 3)    \
    X(768, 4, 16, 12, 256, 4, 6)    \
    X(768, 4, 16, 12, 256, 4, 16)   \
    X(672, 4, 12, 14, 192, 4, 3)    \
    X(672, 4, 12, 14, 192, 4, 12)   \
    X(576, 4, 12, 12, 192, 4, 3)    \
    X(576, 4, 12, 12, 192, 4, 12)   \
    X(480, 4, 12, 10, 192, 4, 4)    \
    X(480, 4, 12, 10, 192, 4, 12)   \
    X(384, 4, 8, 12, 192, 6, 3)     \
    X(384, 4, 8, 12, 192, 6, 8)     \
    X(288, 4, 6, 12, 192, 8, 3)     \
    X(288, 4, 6, 12, 192, 8, 6)
#define FC_FAST_ROW_CONFIGS(X) FC_FAST_ROW_CONFIGS_G0(X) FC_FAST_ROW_CONFIGS_G1(X) FC_FAST_ROW_CONFIGS_G2(X)
constexpr int FC_ROW_GROUPS = 3;
// the configurations of group G only
#define FC_ROW_CONFIGS_OF_GROUP(G, X)                                   \
    do {                                                                \
        if constexpr ((G) == 0) { FC_FAST_ROW_CONFIGS_G0(X) }           \
        else if constexpr ((G) == 1) { FC_FAST_ROW_CONFIGS_G1(X) }      \
        else { FC_FAST_ROW_CONFIGS_G2(X) }                              \
    } while (0)

struct FastRowsInfo {
    bool ok = false;
    int L = 0, R1 = 0, R2 = 0, R3 = 0, NT = 0, RPW = 1;
    int max_kw = 0;      // largest kernel width the fast kernel accepts
    size_t lds_bytes = 0;
};

// Is there a fast row kernel for transform length L able to take kernels up to max_kw wide?
inline FastRowsInfo fast_rows_lookup(int L, int max_kw) {
    FastRowsInfo r;
#define FC_X(LL, A, B, C, NTT, RP, NZ)                                                    \
    if (!r.ok && L == LL) {                                                               \
        using Cfg = RowCfg<LL, A, B, C, NTT, RP>;                                         \
        if (max_kw <= Cfg::m1) {                                                          \
            r.ok = true; r.L = LL; r.R1 = A; r.R2 = B; r.R3 = C; r.NT = NTT; r.RPW = RP;  \
            r.max_kw = Cfg::m1; r.lds_bytes = (size_t)Cfg::LDS_ELEMS * sizeof(c32);       \
        }                                                                                 \
    }
    FC_FAST_ROW_CONFIGS(FC_X)
#undef FC_X
    return r;
}

inline bool fast_rows_length(int L, int max_kw) { return fast_rows_lookup(L, max_kw < 1 ? 1 : max_kw).ok; }

// Cost per point of a specialised length relative to the planner's generic estimate (planner.hpp: LengthFactor; 0 = no
// specialised kernel).  0.45 is what the specialised kernels measure against the generic ones; the two BASELINE windows
// that are awkward to factor carry what their own kernels measure against the next convenient length + crop
// (profiles/r04*_native_window_ab.txt): 1088 = 2^6 x 17 against 1152, 4160 = 2^6 x 5 x 13 against 4224.
inline double fast_rows_factor(int L, int max_kw) {
    if (!fast_rows_length(L, max_kw)) return 0.0;
    if (L == 1088) return FC_FACTOR_1088;
    if (L == 4160) return FC_FACTOR_4160;
    if (L >= 5120) return 0.60;      // the long-row configurations run at ~270 Gpx/s-equivalent against ~360 up to 4608 points
    return 0.45;
}

// Calls run.template go<Cfg, NZ2>() for the first listed configuration of length L (in group G: the translation units
// of a kernel family instantiate one group each) whose NZ2 covers `nz2_needed` (configurations are listed with
// ascending NZ2).  false if there is none.
template <int G, class Runner>
inline bool fast_rows_dispatch_group(int L, int nz2_needed, Runner&& run) {
#define FC_X(LL, A, B, C, NTT, RP, NZ)                          \
    if (L == LL && nz2_needed <= NZ) {                          \
        run.template go<RowCfg<LL, A, B, C, NTT, RP>, NZ>();    \
        return true;                                            \
    }
    FC_ROW_CONFIGS_OF_GROUP(G, FC_X);
#undef FC_X
    return false;
}
template <class Runner>
inline bool fast_rows_dispatch(int L, int nz2_needed, Runner&& run) {
    return fast_rows_dispatch_group<0>(L, nz2_needed, run) || fast_rows_dispatch_group<1>(L, nz2_needed, run) ||
           fast_rows_dispatch_group<2>(L, nz2_needed, run);
}

// Forward image-row kernel (fast_rows_fwd.hpp): one instantiation per length (the first listed
// configuration of that length; NZ2 does not matter).  run.template go<Cfg>().
template <int G, class Runner>
inline bool fast_rows_fwd_dispatch_group(int L, Runner&& run) {
    bool done = false;
#define FC_X(LL, A, B, C, NTT, RP, NZ)                          \
    if (!done && L == LL) {                                     \
        run.template go<RowCfg<LL, A, B, C, NTT, RP>>();        \
        done = true;                                            \
    }
    FC_ROW_CONFIGS_OF_GROUP(G, FC_X);
#undef FC_X
    return done;
}
template <class Runner>
inline bool fast_rows_fwd_dispatch(int L, Runner&& run) {
    return fast_rows_fwd_dispatch_group<0>(L, run) || fast_rows_fwd_dispatch_group<1>(L, run) || fast_rows_fwd_dispatch_group<2>(L, run);
}

// Host tables of a fast row configuration.
struct FastRowsTables {
    Plan1D plan;               // radices (R1, R2, R3)
    std::vector<c32> tw1;      // w_L^j, j < m1
    std::vector<c32> tw2;      // stage-2 table
    std::vector<int> relayout; // register-order index -> generic position (RowsFwdArgs::out_map)
};

inline FastRowsTables make_fast_rows_tables(const FastRowsInfo& fi, const Plan1D& generic) {
    FastRowsTables t;
    t.plan = make_plan1d_seq(fi.L, {fi.R1, fi.R2, fi.R3});
    const int m1 = fi.L / fi.R1;
    const StageDesc& s1 = t.plan.desc.st[0];
    const StageDesc& s2 = t.plan.desc.st[1];
    t.tw1.assign(t.plan.tw.begin() + s1.tw_off, t.plan.tw.begin() + s1.tw_off + m1);
    t.tw2.assign(t.plan.tw.begin() + s2.tw_off, t.plan.tw.begin() + s2.tw_off + (fi.R2 - 1) * fi.R3);
    const int NB3 = fi.R1 * fi.R2;
    t.relayout.assign(fi.L, 0);
    for (int k = 0; k < fi.L; k++) {
        int p = t.plan.pos[k];
        int q = p / fi.R3, a = p % fi.R3;
        int ro = ((a >> 1) * NB3 + q) * 2 + (a & 1);
        t.relayout[ro] = generic.pos[k];
    }
    return t;
}

// ---------------------------------------------------------------------------------------
// Output (column) kernel configurations: X(M, R1, R2, R3, T, NT), see fast_cols.hpp: M = R1 x R2 x R3 complex points (transform
// length 2M along h), T columns per tile, NT = R1 * R2 * T threads (one stage-3 butterfly per thread).  T = 16 (full 128-byte lines of
// the tiled intermediate) while 16 columns fit the LDS (M <= 1056), 8 up to M = 2304, 4 above.  The forward column kernel
// (fast_cols_fwd.hpp) runs the same configurations.
//   * cfg3: M = 2112 = 6.16.22, 8 columns, 768 threads (8.12.22 until the round-4 search: 6.16.22 is 4 % faster, roofline fraction
//     0.68 -> 0.71; 4 columns / 384 threads, two workgroups per CU, measured 47.8 against 36.1 us per map in round 1); cfg5: 1056 =
//     6.8.22; cfg2: 576 = 6.8.12; cfg1: 144 = 4.6.6.
//   * Round 4: the R3 = 16 / 24 forms were replaced (stage-3 bank conflicts, see the row list): output kernel -2 ... -15 % for eleven
//     lengths (r04h_row_and_column_configs_ab.txt); 1024-thread workgroups (16 waves, <= 128 VGPRs) where R3 = 10 / 12 / 18 needs
//     them.  M = 1152 as 8.8.18 / T = 16, 576 as 4.8.18 and 288 as 6.8.6 measured no better than the listed forms.
//   * The same search (r04i_config_search_*.txt): M = 3520 as 10.16.22 (-4 %), 1760 as 5.16.22 (-6 %); 1056 as 3.16.22, 1152 as 12.8.12,
//     576 as 3.16.12 / 4.12.12, 1408 as 4.16.22 within noise over three repetitions and left alone.
//   * M = 2080 = 8.13.20 (832 threads) and 544 = 17.4.8 (8 columns): the cfg4 / cfg2 windows' own kernels (exact_window plans).  Round 5
//     (r05d_native_window_search.txt): 8.10.26 (radix 26 beside seven rounds of prefetch: 12 spilled registers) -> 8.13.20 on 832 threads
//     (128 registers: 4 spilled), -12 % (13.8.20 -11 %, 4.20.26 -6 %, 5.16.26 / 10.8.26 +-1 %, 4.26.20 +23 %); 544 as 8-column tiles 17.4.8: the output
//     kernel itself +6 % at 64 maps, but the forward column passes (image, kernels) get twice the tiles: the step -1 % at 64 maps,
//     -11 % at cfg2's 16 (2.17.16 / 17.2.16 at 16 columns: no difference).
#define FC_FAST_COL_CONFIGS_G0(X) \
    X(4224, 8, 24, 22, 4, 768)    \
    X(3840, 8, 24, 20, 4, 768)    \
    X(3520, 10, 16, 22, 4, 640)    \
    X(3072, 8, 32, 12, 4, 1024)    \
    X(2816, 8, 16, 22, 4, 512)    \
    X(2560, 8, 32, 10, 4, 1024)    \
    X(2304, 8, 16, 18, 8, 1024)    \
    X(2112, 6, 16, 22, 8, 768)    \
    X(2080, 8, 13, 20, 8, 832)    \
    X(1920, 8, 12, 20, 8, 768)    \
    X(1760, 5, 16, 22, 8, 640)    \
    X(1680, 6, 20, 14, 8, 960)
#define FC_FAST_COL_CONFIGS_G1(X) \
    X(1536, 8, 16, 12, 8, 1024)    \
    X(1408, 8, 8, 22, 8, 512)    \
    X(1280, 8, 16, 10, 8, 1024)    \
    X(1152, 8, 12, 12, 8, 768)    \
    X(1056, 6, 8, 22, 16, 768)    \
    X(960, 6, 8, 20, 16, 768)     \
    X(880, 5, 8, 22, 16, 640)     \
    X(768, 8, 8, 12, 16, 1024)     \
    X(672, 6, 8, 14, 16, 768)     \
    X(640, 8, 8, 10, 16, 1024)    \
    X(576, 6, 8, 12, 16, 768)     \
    X(544, 17, 4, 8, 8, 544)      \
    X(480, 6, 8, 10, 16, 768)     \
    X(432, 6, 6, 12, 16, 576)     \
    X(384, 4, 8, 12, 16, 512)     \
    X(336, 4, 6, 14, 16, 384)     \
    X(288, 4, 6, 12, 16, 384)     \
    X(240, 4, 6, 10, 16, 384)     \
    X(192, 4, 8, 6, 16, 512)      \
    X(144, 4, 6, 6, 16, 384)
#define FC_FAST_COL_CONFIGS(X) FC_FAST_COL_CONFIGS_G0(X) FC_FAST_COL_CONFIGS_G1(X)
constexpr int FC_COL_GROUPS = 2;
#define FC_COL_CONFIGS_OF_GROUP(G, X)                                   \
    do {                                                                \
        if constexpr ((G) == 0) { FC_FAST_COL_CONFIGS_G0(X) }           \
        else { FC_FAST_COL_CONFIGS_G1(X) }                              \
    } while (0)

struct FastColsInfo {
    bool ok = false;
    int M = 0, R1 = 0, R2 = 0, R3 = 0, T = 0, NT = 0;
    size_t lds_bytes = 0;
};

inline FastColsInfo fast_cols_lookup(int M) {
    FastColsInfo r;
#define FC_X(MM, A, B, C, TT, NTT)                                                   \
    if (!r.ok && M == MM) {                                                          \
        using Cfg = ColCfg<MM, A, B, C, TT, NTT>;                                    \
        r.ok = true; r.M = MM; r.R1 = A; r.R2 = B; r.R3 = C; r.T = TT; r.NT = NTT;   \
        r.lds_bytes = (size_t)Cfg::LDS_ELEMS * sizeof(c32);                          \
    }
    FC_FAST_COL_CONFIGS(FC_X)
#undef FC_X
    return r;
}

inline bool fast_cols_length(int M, int) { return fast_cols_lookup(M).ok; }
inline double fast_cols_factor(int M, int) {
    if (!fast_cols_lookup(M).ok) return 0.0;
    if (M == 544) return FC_FACTOR_1088;
    if (M == 2080) return FC_FACTOR_4160;
    if (M >= 2560) return 0.60;      // 4-column tiles
    return 0.45;
}

// Measured cost of the two hot kernels per point of their transforms, picoseconds (profiles/r04k_cost_per_point_by_length.txt:
// the multi-map row kernel per point of a spectrum row, the output kernel per complex point of a stored column; one MI355X,
// 63 x 63 kernels, 16-64 maps per launch).  The block-wise planner (fftconv_api.cpp: choose_save_tiling) weighs block
// transforms against each other and against one pass with them; only the ratios matter.
inline double fast_rows_ps(int L) {
    switch (L) {
        case 1152: return 2.52; case 1344: return 2.71; case 1536: return 2.21; case 1760: return 2.53; case 1920: return 2.39;
        case 2112: return 2.41; case 2304: return 2.23; case 2560: return 2.38; case 2816: return 2.43; case 3072: return 2.12;
        case 3520: return 2.55; case 3840: return 2.42; case 4224: return 2.39; case 4608: return 2.42; case 5120: return 2.38;
        case 5632: return 2.40; case 6144: return 2.38; case 7040: return 2.86; case 7680: return 2.62; case 8448: return 2.78;
        case 1088: case 4160: return 2.9;
        default: return 2.7;             // the small lengths (launch-bound at their own sizes)
    }
}
inline double fast_cols_ps(int M) {
    switch (M) {
        case 576: return 2.94; case 672: return 2.84; case 768: return 2.97; case 880: return 3.06; case 960: return 2.84;
        case 1056: return 2.94; case 1152: return 2.89; case 1280: return 2.78; case 1408: return 3.01; case 1536: return 2.83;
        case 1760: return 2.95; case 1920: return 2.92; case 2112: return 2.80; case 2304: return 2.80;
        case 2560: return 3.76; case 2816: return 3.66; case 3072: return 3.62; case 3520: return 3.70; case 3840: return 3.69;
        case 4224: return 3.67;
        case 544: case 2080: return 3.3;
        default: return 3.1;
    }
}

// Forward column kernel (fast_cols_fwd.hpp): same configurations; NZ2 = 3 (pruned, short kernels)
// or R2 (any input length).  run.template go<Cfg, NZ2>().
template <int G, class Runner>
inline bool fast_cols_fwd_dispatch_group(int M, int T, bool pruned, Runner&& run) {
#define FC_X(MM, A, B, C, TT, NTT)                                       \
    if (M == MM && T == TT) {                                            \
        if constexpr (B > 3) {                                           \
            if (pruned) { run.template go<ColCfg<MM, A, B, C, TT, NTT>, 3>(); return true; }  \
        }                                                                \
        run.template go<ColCfg<MM, A, B, C, TT, NTT>, B>();              \
        return true;                                                     \
    }
    FC_COL_CONFIGS_OF_GROUP(G, FC_X);
#undef FC_X
    return false;
}
template <class Runner>
inline bool fast_cols_fwd_dispatch(int M, int T, bool pruned, Runner&& run) {
    return fast_cols_fwd_dispatch_group<0>(M, T, pruned, run) || fast_cols_fwd_dispatch_group<1>(M, T, pruned, run);
}

// may the pruned variant take columns of h_in samples?  (one non-zero input per stage-1 butterfly,
// at most 3 non-zero inputs per stage-2 butterfly)
inline bool fast_cols_fwd_pruned_ok(const FastColsInfo& fi, int h_in) {
    const int nz = (h_in + 1) / 2, m1 = fi.M / fi.R1;
    return fi.R2 > 3 && nz <= m1 && nz <= 3 * fi.R3;
}

template <int G, class Runner>
inline bool fast_cols_dispatch_group(int M, int T, Runner&& run) {
#define FC_X(MM, A, B, C, TT, NTT)                          \
    if (M == MM && T == TT) {                               \
        run.template go<ColCfg<MM, A, B, C, TT, NTT>>();    \
        return true;                                        \
    }
    FC_COL_CONFIGS_OF_GROUP(G, FC_X);
#undef FC_X
    return false;
}
template <class Runner>
inline bool fast_cols_dispatch(int M, int T, Runner&& run) {
    return fast_cols_dispatch_group<0>(M, T, run) || fast_cols_dispatch_group<1>(M, T, run);
}

// Sliced tail round of the output kernel (fast_cols.hpp, SLICED): for the small transforms (M <= 1056: the launches of
// small problems are a handful of rounds of tiles) a last, partial round of `rem` tiles on `want` persistent workgroups is
// cut into S = 2^k column slices per tile, S <= want / rem and at least 2 columns per slice.  Fills in the tail fields
// of `a` (and shrinks a.ntiles to the full rounds) and returns the grid to launch; false: launch unsliced.
constexpr int FC_SLICE_MAX_M = 1056;
inline bool fast_cols_slice_plan(int M, int T, int want, FastColsArgs& a, int& grid) {
    a.tail_first = a.tail_tiles = a.slice_shift = 0;
    if (!a.y_tiled || M > FC_SLICE_MAX_M || want < 2 || a.ntiles <= 0) return false;
    if (T & (T - 1)) return false;        // slices are power-of-two fractions of a tile ...
    const int full_rounds = a.ntiles / want, rem = a.ntiles - full_rounds * want;
    // (launches of less than one round stay whole: cfg1's 18 tiles as 144 two-column slices took 24.6 instead of 22.3 us
    //  per step -- every slice pays the tile's whole phase latency)
    if (rem == 0 || full_rounds == 0) return false;
    int shift = 0;
    // ... of an EVEN number of columns: the gather and the landing work on column pairs (fast_cols.hpp)
    while ((2 << shift) * rem <= want && (T >> (shift + 1)) >= 2 && ((T >> (shift + 1)) & 1) == 0) shift++;
    if (shift == 0) return false;
    a.tail_first = full_rounds * want;
    a.tail_tiles = rem;
    a.slice_shift = shift;
    a.ntiles = full_rounds * want;
    grid = want;
    return true;
}

struct FastColsTables {
    Plan1D plan;                   // radices (R1, R2, R3)
    std::vector<c32> tw1, tw2;
    std::vector<PairEntry> pairs;  // positions in the fast plan's order
    std::vector<int> rowoff;       // M+1: row-major Y: row offset feeding LDS position p
    std::vector<int> pair_row_of;  // M+1: spectrum row i (producer order) -> row of the pair-adjacent tiled intermediate
};

// `producer`: the plan whose digit-reversed order the spectrum rows are produced in.
inline FastColsTables make_fast_cols_tables(const FastColsInfo& fi, const Plan1D& producer, int y_pitch) {
    FastColsTables t;
    t.plan = make_plan1d_seq(fi.M, {fi.R1, fi.R2, fi.R3});
    const int m1 = fi.M / fi.R1;
    const StageDesc& s1 = t.plan.desc.st[0];
    const StageDesc& s2 = t.plan.desc.st[1];
    t.tw1.assign(t.plan.tw.begin() + s1.tw_off, t.plan.tw.begin() + s1.tw_off + m1);
    t.tw2.assign(t.plan.tw.begin() + s2.tw_off, t.plan.tw.begin() + s2.tw_off + (fi.R2 - 1) * fi.R3);
    t.pairs = make_pair_table(t.plan);
    t.rowoff.assign(fi.M + 1, 0);
    for (int k = 0; k < fi.M; k++) t.rowoff[t.plan.pos[k]] = producer.pos[k] * y_pitch;
    t.rowoff[fi.M] = fi.M * y_pitch;
    // pair-adjacent tile rows: pair p = (bin p, bin M-p) at rows 2p, 2p+1;
    // p = 0: (DC, Nyquist); p = M/2: (middle bin, padding)
    t.pair_row_of.assign(fi.M + 1, 0);
    for (int k = 0; k < fi.M; k++) {
        int r = (k <= fi.M / 2) ? 2 * k : 2 * (fi.M - k) + 1;
        t.pair_row_of[producer.pos[k]] = r;
    }
    t.pair_row_of[fi.M] = 1;
    return t;
}

}  // namespace fc
