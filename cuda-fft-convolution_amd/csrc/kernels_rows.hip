// kernels_rows.hip -- the one-map spectral-row kernels (fast_rows.hpp: plain and persistent variants)
// (one of the kernels_*.hip translation units; see kernels_common.hpp).
#include "kernels_common.hpp"

namespace fc {
namespace {

// Workgroup -> (row group, kernel).  order 0: blockIdx = (group, kernel).  order 1: 1-D grid with
// the kernel index fastest: the workgroups that multiply by the same image-spectrum rows run back
// to back.  order 2: additionally XCD-aware -- blocks b and b+8 share an XCD (round-robin
// dispatch; a speed assumption only), so XCD x = b % 8 walks groups x, x+8, ... with the kernel
// index fastest and every spectrum row is fetched once per XCD L2 instead of once per kernel.
template <class Cfg, int NZ2, bool MULTIF>
__global__ void __launch_bounds__(Cfg::NT, 3) k_fast_rows(FastRowsArgs a, int rows, int order, int groups, int nk) {
    int group, kernel;
    if (order == 0) {
        group = (int)blockIdx.x;
        kernel = (int)blockIdx.y;
    } else if (order == 1) {
        const int b = (int)blockIdx.x;
        group = b / nk;
        kernel = b - group * nk;
    } else {
        const int b = (int)blockIdx.x;
        const int xcd = b & 7, sq = b >> 3;
        const int gl = sq / nk;
        kernel = sq - gl * nk;
        group = gl * 8 + xcd;
        if (group >= groups) return;
    }
    DevPhaseCtx<RowState<Cfg, MULTIF>> ctx;
    fast_rows_body<Cfg, NZ2, MULTIF>(ctx, reinterpret_cast<c32*>(fc_smem), a, group, kernel, rows);
}

template <class Cfg, int NZ2, bool MULTIF>
__global__ void __launch_bounds__(Cfg::NT, 3) k_fast_rows_persist(FastRowsArgs a, int rows, int total_items) {
    // contiguous run of items per workgroup; the first (total % nwg) workgroups take one more
    const int nwg = (int)gridDim.x, wg = (int)blockIdx.x;
    const int base = total_items / nwg, rem = total_items - base * nwg;
    const int item0 = wg * base + (wg < rem ? wg : rem);
    const int item1 = item0 + base + (wg < rem ? 1 : 0);
    DevPhaseCtx<RowState<Cfg, MULTIF>> ctx;
    fast_rows_persist_body<Cfg, NZ2, MULTIF>(ctx, reinterpret_cast<c32*>(fc_smem), a, rows, item0, item1);
}

struct FastRowsLauncher {
    const FastRowsArgs& a;
    int rows, kernels;
    hipStream_t s;
    int persist_wgs = 0;   // > 0: persistent variant with that many workgroups
    int order = 0;         // workgroup order of the plain variant (see k_fast_rows)
    hipError_t err = hipSuccess;
    template <class Cfg, int NZ2>
    void go() {
        if constexpr (Cfg::RPW == 1) {
            if (persist_wgs > 0) {
                if (a.F > 1) launch_persist<Cfg, NZ2, true>();
                else launch_persist<Cfg, NZ2, false>();
                return;
            }
        }
        if (a.F > 1) launch<Cfg, NZ2, true>();
        else launch<Cfg, NZ2, false>();
    }
    template <class Cfg, int NZ2, bool MULTIF>
    void launch_persist() {
        static unsigned long long attr_mask = 0;
        const size_t lds = (size_t)Cfg::LDS_ELEMS * sizeof(c32);
        err = ensure_lds_attr(k_fast_rows_persist<Cfg, NZ2, MULTIF>, attr_mask);
        if (err != hipSuccess) return;
        const int total = rows * kernels;
        const int grid = total < persist_wgs ? total : persist_wgs;
        hipLaunchKernelGGL((k_fast_rows_persist<Cfg, NZ2, MULTIF>), dim3(grid), dim3(Cfg::NT), lds, s, a, rows, total);
        err = hipGetLastError();
    }
    template <class Cfg, int NZ2, bool MULTIF>
    void launch() {
        static unsigned long long attr_mask = 0;
        const size_t lds = (size_t)Cfg::LDS_ELEMS * sizeof(c32);
        err = ensure_lds_attr(k_fast_rows<Cfg, NZ2, MULTIF>, attr_mask);
        if (err != hipSuccess) return;
        const int groups = (rows + Cfg::RPW - 1) / Cfg::RPW;
        dim3 grid(groups, kernels);
        if (order == 1) grid = dim3(groups * kernels);
        if (order == 2) grid = dim3(8 * ((groups + 7) / 8) * kernels);
        hipLaunchKernelGGL((k_fast_rows<Cfg, NZ2, MULTIF>), grid, dim3(Cfg::NT), lds, s, a, rows, order, groups, kernels);
        err = hipGetLastError();
    }
};

}  // namespace

hipError_t launch_fast_rows(int L, int nz2, const FastRowsArgs& a, int rows, int kernels, int persist_wgs, int order, hipStream_t s) {
    if (rows <= 0 || kernels <= 0) return hipSuccess;
    FastRowsLauncher l{a, rows, kernels, s, persist_wgs, order};
    if (!fast_rows_dispatch(L, nz2, l)) return hipErrorInvalidValue;
    return l.err;
}

}  // namespace fc
