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
#include "fast_cols_wide.hpp"
#include "fast_rows.hpp"
#include "fast_rows_multi.hpp"
#include "fast_rows_pair.hpp"
#include "planner.hpp"

namespace fc {

// X(L, R1, R2, R3, NT, RPW, NZ2): see fast_rows.hpp.  NT = 192 threads (3 waves) everywhere;
// RPW rows per workgroup chosen so that RPW * R1 * R2 = 192 stage-3 butterflies fill the lanes.
//   4224 = 8 x 24 x 22 (cfg3; cfg4's 4160 window also runs on it): butterflies 528 / 176 / 192
//   2112 = 8 x 12 x 22, 2 rows (cfg5):                            528 / 352 / 192
//   1152 = 6 x  8 x 24, 4 rows (cfg2's 1088 window):              768 / 576 / 192
//   8448 = 16 x 24 x 22, 384 threads, two workgroups per CU (8192-sized images): 528 / 352 / 384
//   6144 = 16 x 24 x 16, 384 threads, two workgroups per CU (images between 4224 and 6144): 384 / 256 / 384
//   3072 = 8 x 24 x 16 (images around 2500 - 3000):               384 / 128 / 192
//   1536 = 8 x 12 x 16, 2 rows (1280-wide images):                384 / 256 / 192
//    768 = 4 x 12 x 16, 4 rows (640 / 720-sized images):          768 / 256 / 192
//    576 = 4 x 12 x 12, 4 rows (512-sized images):                576 / 192 / 192
//    288 = 4 x  6 x 12, 8 rows (cfg1):                            576 / 384 / 192
// Listed with ascending NZ2 per length (the dispatcher takes the first that covers the kernel).
#define FC_FAST_ROW_CONFIGS(X)      \
    X(8448, 16, 24, 22, 384, 1, 3)  \
    X(8448, 16, 24, 22, 384, 1, 6)  \
    X(8448, 16, 24, 22, 384, 1, 24) \
    X(6144, 16, 24, 16, 384, 1, 3)  \
    X(6144, 16, 24, 16, 384, 1, 6)  \
    X(6144, 16, 24, 16, 384, 1, 24) \
    X(4224, 8, 24, 22, 192, 1, 3)   \
    X(4224, 8, 24, 22, 192, 1, 6)   \
    X(4224, 8, 24, 22, 192, 1, 24)  \
    X(3072, 8, 24, 16, 192, 1, 3)   \
    X(3072, 8, 24, 16, 192, 1, 6)   \
    X(3072, 8, 24, 16, 192, 1, 24)  \
    X(2112, 8, 12, 22, 192, 2, 3)   \
    X(2112, 8, 12, 22, 192, 2, 12)  \
    X(1536, 8, 12, 16, 192, 2, 3)   \
    X(1536, 8, 12, 16, 192, 2, 12)  \
    X(1152, 6, 8, 24, 192, 4, 3)    \
    X(1152, 6, 8, 24, 192, 4, 8)    \
    X(768, 4, 12, 16, 192, 4, 3)    \
    X(768, 4, 12, 16, 192, 4, 12)   \
    X(576, 4, 12, 12, 192, 4, 3)    \
    X(576, 4, 12, 12, 192, 4, 12)   \
    X(288, 4, 6, 12, 192, 8, 3)     \
    X(288, 4, 6, 12, 192, 8, 6)

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

inline bool fast_rows_length(int L) { return fast_rows_lookup(L, 1).ok; }

// Calls run.template go<Cfg, NZ2>() for the first listed configuration of length L whose NZ2
// covers `nz2_needed` (configurations are listed with ascending NZ2).  false if there is none.
template <class Runner>
inline bool fast_rows_dispatch(int L, int nz2_needed, Runner&& run) {
#define FC_X(LL, A, B, C, NTT, RP, NZ)                          \
    if (L == LL && nz2_needed <= NZ) {                          \
        run.template go<RowCfg<LL, A, B, C, NTT, RP>, NZ>();    \
        return true;                                            \
    }
    FC_FAST_ROW_CONFIGS(FC_X)
#undef FC_X
    return false;
}

// Variants built on one-row-per-workgroup configurations only (paired rows, persistent).
template <class Runner>
inline bool fast_rows_rpw1_dispatch(int L, int nz2_needed, Runner&& run) {
#define FC_X(LL, A, B, C, NTT, RP, NZ)                                      \
    if constexpr (RP == 1) {                                                \
        if (L == LL && nz2_needed <= NZ) {                                  \
            run.template go<RowCfg<LL, A, B, C, NTT, RP>, NZ>();            \
            return true;                                                    \
        }                                                                   \
    }
    FC_FAST_ROW_CONFIGS(FC_X)
#undef FC_X
    return false;
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
// Output (column) kernel configurations: X(M, R1, R2, R3, T, NT), see fast_cols.hpp.
// M = 2112 (FFT_H = 4224, cfg3): 8 x 12 x 22, 8 columns per tile, 768 threads:
//   stage butterflies per tile 2112 / 1408 / 768 -> 2.75 / 1.83 / 1.0 rounds of 768 lanes.
// ---------------------------------------------------------------------------------------
//   M = 1056 (FFT_H 2112, cfg5): 6 x 8 x 22, 16 columns per tile (full 128-byte lines), 768 threads
//   M =  576 (transform 1152, cfg2): 6 x 8 x 12, 16 columns, 768 threads
//   M = 4224 (transform 8448, 8192-sized images): 8 x 24 x 22, 4 columns per tile (LDS), 768 threads
//   M = 3072 (transform 6144): 8 x 24 x 16, 4 columns per tile, 768 threads
//   M = 1536 / 768 / 384 (transforms 3072 / 1536 / 768): 8 x 12 x 16 (8 columns), 6 x 8 x 16, 4 x 6 x 16
//   M =  288 (transform 576, 512-sized images): 4 x 6 x 12, 16 columns, 384 threads
//   M =  144 (FFT_H 288, cfg1): 4 x 6 x 6, 16 columns, 384 threads
//   M = 2112 also as 4 columns / 384 threads (two workgroups per CU): measured much slower
//   (32-byte gather pieces: 47.8 vs 36.1 us per map); kept for A/B via FFTCONV_COLS_T=4.
#define FC_FAST_COL_CONFIGS(X)   \
    X(4224, 8, 24, 22, 4, 768)   \
    X(3072, 8, 24, 16, 4, 768)   \
    X(2112, 8, 12, 22, 8, 768)   \
    X(2112, 8, 12, 22, 4, 384)   \
    X(1536, 8, 12, 16, 8, 768)   \
    X(1056, 6, 8, 22, 16, 768)   \
    X(768, 6, 8, 16, 16, 768)    \
    X(576, 6, 8, 12, 16, 768)    \
    X(384, 4, 6, 16, 16, 384)    \
    X(288, 4, 6, 12, 16, 384)    \
    X(144, 4, 6, 6, 16, 384)

struct FastColsInfo {
    bool ok = false;
    int M = 0, R1 = 0, R2 = 0, R3 = 0, T = 0, NT = 0;
    size_t lds_bytes = 0;
};

// prefer_T > 0: only the configuration with that tile width
inline FastColsInfo fast_cols_lookup(int M, int prefer_T = 0) {
    FastColsInfo r;
#define FC_X(MM, A, B, C, TT, NTT)                                                   \
    if (!r.ok && M == MM && (prefer_T <= 0 || prefer_T == TT)) {                     \
        using Cfg = ColCfg<MM, A, B, C, TT, NTT>;                                    \
        r.ok = true; r.M = MM; r.R1 = A; r.R2 = B; r.R3 = C; r.T = TT; r.NT = NTT;   \
        r.lds_bytes = (size_t)Cfg::LDS_ELEMS * sizeof(c32);                          \
    }
    FC_FAST_COL_CONFIGS(FC_X)
#undef FC_X
    return r;
}

// Paired-row kernel: the one-row-per-workgroup configurations.
template <class Runner>
inline bool fast_rows_pair_dispatch(int L, int nz2_needed, Runner&& run) { return fast_rows_rpw1_dispatch(L, nz2_needed, run); }

inline bool fast_cols_length(int M) { return fast_cols_lookup(M).ok; }

// Forward column kernel (fast_cols_fwd.hpp): same configurations; NZ2 = 3 (pruned, short kernels)
// or R2 (any input length).  run.template go<Cfg, NZ2>().
template <class Runner>
inline bool fast_cols_fwd_dispatch(int M, int T, bool pruned, Runner&& run) {
#define FC_X(MM, A, B, C, TT, NTT)                                       \
    if (M == MM && T == TT) {                                            \
        if (pruned) run.template go<ColCfg<MM, A, B, C, TT, NTT>, 3>();  \
        else run.template go<ColCfg<MM, A, B, C, TT, NTT>, B>();         \
        return true;                                                     \
    }
    FC_FAST_COL_CONFIGS(FC_X)
#undef FC_X
    return false;
}

// may the pruned variant take columns of h_in samples?  (one non-zero input per stage-1 butterfly,
// at most 3 non-zero inputs per stage-2 butterfly)
inline bool fast_cols_fwd_pruned_ok(const FastColsInfo& fi, int h_in) {
    const int nz = (h_in + 1) / 2, m1 = fi.M / fi.R1;
    return fi.R2 > 3 && nz <= m1 && nz <= 3 * fi.R3;
}

template <class Runner>
inline bool fast_cols_dispatch(int M, int T, Runner&& run) {
#define FC_X(MM, A, B, C, TT, NTT)                          \
    if (M == MM && T == TT) {                               \
        run.template go<ColCfg<MM, A, B, C, TT, NTT>>();    \
        return true;                                        \
    }
    FC_FAST_COL_CONFIGS(FC_X)
#undef FC_X
    return false;
}

struct FastColsTables {
    Plan1D plan;                   // radices (R1, R2, R3)
    std::vector<c32> tw1, tw2;
    std::vector<PairEntry> pairs;  // positions in the fast plan's order
    std::vector<int> rowoff;       // M+1: Y row offset feeding LDS position p
    std::vector<int> tile_row_of;  // M+1: generic spectrum row i -> row of the tiled intermediate
    std::vector<int> tile_lpos;    // M+1: tile row -> LDS landing position in the output kernel
    std::vector<int> pair_row_of;  // M+1: generic spectrum row i -> row of the pair-adjacent tile (mode 3)
    std::vector<int> pair_row_seq; // M+1: spectrum rows sorted by pair_row_of (processing order of the persistent row kernel)
    // precombined intermediate (fast_rows_pair.hpp): one RowPair per paired-row workgroup, and the
    // LDS landing position of every tile row
    std::vector<RowPair> row_pairs;  // M/2 + 1
    std::vector<int> lpos;           // M
};

// row_order (tiled intermediate): 0 = tile rows in the output kernel's LDS order (sequential,
// conflict-free landing); 1 = rows of workgroups i and i+8 adjacent (half-lines meet in one L2);
// 2 = spectrum row order (rows i, i+1 adjacent: what the persistent row kernel writes back to back).
inline FastColsTables make_fast_cols_tables(const FastColsInfo& fi, const Plan1D& generic, int y_pitch, int row_order = 0) {
    FastColsTables t;
    t.plan = make_plan1d_seq(fi.M, {fi.R1, fi.R2, fi.R3});
    const int m1 = fi.M / fi.R1;
    const StageDesc& s1 = t.plan.desc.st[0];
    const StageDesc& s2 = t.plan.desc.st[1];
    t.tw1.assign(t.plan.tw.begin() + s1.tw_off, t.plan.tw.begin() + s1.tw_off + m1);
    t.tw2.assign(t.plan.tw.begin() + s2.tw_off, t.plan.tw.begin() + s2.tw_off + (fi.R2 - 1) * fi.R3);
    t.pairs = make_pair_table(t.plan);
    t.rowoff.assign(fi.M + 1, 0);
    for (int k = 0; k < fi.M; k++) t.rowoff[t.plan.pos[k]] = generic.pos[k] * y_pitch;
    t.rowoff[fi.M] = fi.M * y_pitch;
    // Tiled (not precombined) intermediate: spectrum row i (= workgroup i of the single-row
    // kernel) goes to tile row r(i) chosen so that workgroups i and i+8 -- same XCD under
    // round-robin dispatch, running at about the same time -- write ADJACENT 64-byte tile rows,
    // i.e. the two halves of one 128-byte line meet in that XCD's L2 before going to HBM.
    t.tile_row_of.assign(fi.M + 1, 0);
    t.tile_lpos.assign(fi.M + 1, 0);
    std::vector<int> bin_of_row(fi.M, 0);
    for (int k = 0; k < fi.M; k++) bin_of_row[generic.pos[k]] = k;
    const int full = (fi.M / 16) * 16;
    for (int i = 0; i < fi.M; i++) {
        int r = t.plan.pos[bin_of_row[i]];
        if (row_order == 1) {
            r = i;
            if (i < full) r = 16 * (i / 16) + 2 * (i % 8) + ((i / 8) % 2);
        } else if (row_order == 2) {
            r = i;
        }
        t.tile_row_of[i] = r;
        t.tile_lpos[r] = t.plan.pos[bin_of_row[i]];
    }
    t.tile_row_of[fi.M] = fi.M;
    t.tile_lpos[fi.M] = fi.M;
    // pair-adjacent tile rows (mode 3): pair p = (bin p, bin M-p) at rows 2p, 2p+1;
    // p = 0: (DC, Nyquist); p = M/2: (middle bin, padding)
    t.pair_row_of.assign(fi.M + 1, 0);
    for (int k = 0; k < fi.M; k++) {
        int r = (k <= fi.M / 2) ? 2 * k : 2 * (fi.M - k) + 1;
        t.pair_row_of[generic.pos[k]] = r;
    }
    t.pair_row_of[fi.M] = 1;
    {
        std::vector<int> inv(fi.M + 2, -1);
        for (int i = 0; i <= fi.M; i++) inv[t.pair_row_of[i]] = i;
        for (int r = 0; r < fi.M + 2; r++)
            if (inv[r] >= 0) t.pair_row_seq.push_back(inv[r]);
    }
    const int M = fi.M;
    t.lpos.assign(M, 0);
    for (int u = 0; u + 1 < M / 2; u++) {   // regular pairs k = u + 1
        const int k = u + 1;
        RowPair rp{};
        rp.rowA = generic.pos[k]; rp.rowB = generic.pos[M - k];
        rp.outA = 2 * u; rp.outB = 2 * u + 1; rp.kind = 0;
        double ang = -2.0 * M_PI * (double)k / (double)(2 * M);
        rp.w = mk((float)std::cos(ang), (float)std::sin(ang));
        t.row_pairs.push_back(rp);
        t.lpos[rp.outA] = t.plan.pos[k];
        t.lpos[rp.outB] = t.plan.pos[M - k];
    }
    {   // DC + Nyquist -> Z_0
        RowPair rp{};
        rp.rowA = generic.pos[0]; rp.rowB = M; rp.outA = M - 2; rp.outB = -1; rp.kind = 1; rp.w = mk(1.f, 0.f);
        t.row_pairs.push_back(rp);
        t.lpos[rp.outA] = t.plan.pos[0];
    }
    {   // self-paired middle bin
        RowPair rp{};
        rp.rowA = generic.pos[M / 2]; rp.rowB = rp.rowA; rp.outA = M - 1; rp.outB = -1; rp.kind = 2; rp.w = mk(1.f, 0.f);
        t.row_pairs.push_back(rp);
        t.lpos[rp.outA] = t.plan.pos[M / 2];
    }
    return t;
}

// ---------------------------------------------------------------------------------------
// 16-column output kernel (fast_cols_wide.hpp): X(H, R2, R3, R4, NT), transform M = 2H.
//   M = 2112 = 2 x (6 x 8 x 22): the cfg3 / cfg4 output pass.
// ---------------------------------------------------------------------------------------
//   (8 x 12 x 11: the in-register radix is kept small -- the kernel holds a whole half tile plus a
//    prefetched one in registers and must stay under the 168 VGPRs of 3 waves per SIMD)
#define FC_FAST_COLW_CONFIGS(X) \
    X(1056, 8, 12, 11, 768)

struct FastColsWideInfo {
    bool ok = false;
    int H = 0, R2 = 0, R3 = 0, R4 = 0, NT = 0;
    size_t lds_bytes = 0;
};

inline FastColsWideInfo fast_cols_wide_lookup(int M) {
    FastColsWideInfo r;
#define FC_X(HH, A, B, C, NTT)                                              \
    if (!r.ok && M == 2 * HH) {                                             \
        using Cfg = ColWideCfg<HH, A, B, C, NTT>;                           \
        r.ok = true; r.H = HH; r.R2 = A; r.R3 = B; r.R4 = C; r.NT = NTT;    \
        r.lds_bytes = (size_t)Cfg::LDS_ELEMS * sizeof(c32);                 \
    }
    FC_FAST_COLW_CONFIGS(FC_X)
#undef FC_X
    return r;
}

template <class Runner>
inline bool fast_cols_wide_dispatch(int M, Runner&& run) {
#define FC_X(HH, A, B, C, NTT)                               \
    if (M == 2 * HH) {                                       \
        run.template go<ColWideCfg<HH, A, B, C, NTT>>();     \
        return true;                                         \
    }
    FC_FAST_COLW_CONFIGS(FC_X)
#undef FC_X
    return false;
}

struct FastColsWideTables {
    Plan1D plan;                        // radices (2, R2, R3, R4)
    std::vector<c32> tw3, twA, twF, wh, wl;
    std::vector<unsigned> ppA, ppB;
    std::vector<int> tile_row_of;       // M+1: generic spectrum row i -> row of the 16-column tile
};

inline FastColsWideTables make_fast_cols_wide_tables(const FastColsWideInfo& fi, const Plan1D& generic) {
    FastColsWideTables t;
    const int H = fi.H, M = 2 * H, m2 = H / fi.R2;
    t.plan = make_plan1d_seq(M, {2, fi.R2, fi.R3, fi.R4});
    const StageDesc& s1 = t.plan.desc.st[0];   // radix 2, m = H:   w_M^b
    const StageDesc& s2 = t.plan.desc.st[1];   // radix R2, m = m2: w_H^(b c)
    const StageDesc& s3 = t.plan.desc.st[2];   // radix R3, m = R4
    t.twF.assign(t.plan.tw.begin() + s1.tw_off, t.plan.tw.begin() + s1.tw_off + m2);
    t.twA.assign(t.plan.tw.begin() + s2.tw_off, t.plan.tw.begin() + s2.tw_off + m2);
    t.tw3.assign(t.plan.tw.begin() + s3.tw_off, t.plan.tw.begin() + s3.tw_off + (fi.R3 - 1) * fi.R4);
    const int N = 2 * M;
    for (int i = 0; i < M / 32 + 2; i++) {
        double a = -2.0 * M_PI * (double)(32 * i) / (double)N;
        t.wh.push_back(mk((float)std::cos(a), (float)std::sin(a)));
    }
    for (int i = 0; i < 32; i++) {
        double a = -2.0 * M_PI * (double)i / (double)N;
        t.wl.push_back(mk((float)std::cos(a), (float)std::sin(a)));
    }
    const std::vector<int>& pos = t.plan.pos;   // even bins in [0, H), odd bins in [H, 2H)
    for (int i = 0; i <= H / 2; i++) {          // half A: k = 2i
        const int k = 2 * i;
        unsigned a = (unsigned)pos[k], b = (i == 0) ? (unsigned)H : (unsigned)pos[M - k];
        t.ppA.push_back(a | (b << 16));
    }
    for (int i = 0; i < H / 2; i++) {           // half B: k = 2i + 1
        const int k = 2 * i + 1;
        unsigned a = (unsigned)(pos[k] - H), b = (unsigned)(pos[M - k] - H);
        t.ppB.push_back(a | (b << 16));
    }
    t.tile_row_of.assign(M + 1, 0);
    for (int k = 0; k < M; k++) t.tile_row_of[generic.pos[k]] = (k & 1) ? pos[k] + 1 : pos[k];
    t.tile_row_of[M] = H;
    return t;
}

}  // namespace fc
