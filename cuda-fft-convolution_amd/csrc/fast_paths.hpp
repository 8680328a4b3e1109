// fast_paths.hpp -- registry of the specialised (compile-time) kernel configurations.
//
// The generic kernels (kernels_body.hpp) handle every supported length; the configurations
// listed here replace the two hot kernels for the transform lengths that matter (the BASELINE
// configs), chosen so that every stage keeps >= 90 % of the lanes busy and the LDS budget allows
// 3 waves per SIMD.  X(L, R1, R2, R3, NT, NZ2): see fast_rows.hpp.
#pragma once
#include <vector>

#include "fast_cols.hpp"
#include "fast_rows.hpp"
#include "fast_rows_pair.hpp"
#include "planner.hpp"

namespace fc {

// L = 4224 (cfg3, cfg5's 2112 handled separately): 8 x 24 x 22 with 192 threads per row:
//   stage butterflies 528 / 176 / 192 -> 2.75 / 0.92 / 1.0 rounds of 192 lanes.
#define FC_FAST_ROW_CONFIGS(X) \
    X(4224, 8, 24, 22, 192, 6) \
    X(4224, 8, 24, 22, 192, 24)

struct FastRowsInfo {
    bool ok = false;
    int L = 0, R1 = 0, R2 = 0, R3 = 0, NT = 0;
    int max_kw = 0;      // largest kernel width the fast kernel accepts
    size_t lds_bytes = 0;
};

// Is there a fast row kernel for transform length L able to take kernels up to max_kw wide?
inline FastRowsInfo fast_rows_lookup(int L, int max_kw) {
    FastRowsInfo r;
#define FC_X(LL, A, B, C, NTT, NZ)                                                        \
    if (!r.ok && L == LL) {                                                               \
        using Cfg = RowCfg<LL, A, B, C, NTT>;                                             \
        constexpr int XR = row_x_rounds<Cfg>();                        \
        int lim = Cfg::m1 < XR * NTT ? Cfg::m1 : XR * NTT;                                \
        if (max_kw <= lim) {                                                              \
            r.ok = true; r.L = LL; r.R1 = A; r.R2 = B; r.R3 = C; r.NT = NTT;              \
            r.max_kw = lim; r.lds_bytes = (size_t)Cfg::LDS_ELEMS * sizeof(c32);           \
        }                                                                                 \
    }
    FC_FAST_ROW_CONFIGS(FC_X)
#undef FC_X
    return r;
}

// Calls run.template go<Cfg, NZ2>() for the first listed configuration of length L whose NZ2
// covers `nz2_needed` (configurations are listed with ascending NZ2).  false if there is none.
template <class Runner>
inline bool fast_rows_dispatch(int L, int nz2_needed, Runner&& run) {
#define FC_X(LL, A, B, C, NTT, NZ)                              \
    if (L == LL && nz2_needed <= NZ) {                          \
        run.template go<RowCfg<LL, A, B, C, NTT>, NZ>();        \
        return true;                                            \
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
    std::vector<int> relayout; // register-order index -> generic position (see relayout_rows_body)
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
#define FC_FAST_COL_CONFIGS(X) \
    X(2112, 8, 12, 22, 8, 768)

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

// Paired-row kernel: same configurations as the single-row kernel.
template <class Runner>
inline bool fast_rows_pair_dispatch(int L, int nz2_needed, Runner&& run) { return fast_rows_dispatch(L, nz2_needed, run); }

template <class Runner>
inline bool fast_cols_dispatch(int M, Runner&& run) {
#define FC_X(MM, A, B, C, TT, NTT)                          \
    if (M == MM) {                                          \
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

}  // namespace fc
