// kernels_rows.hip -- the one-map spectral-row kernel (fast_rows.hpp; F > 1 and single-map launches)
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

struct FastRowsLauncher {
    const FastRowsArgs& a;
    int rows, kernels;
    hipStream_t s;
    int order = 0;         // workgroup order of the plain variant (see k_fast_rows)
    hipError_t err = hipSuccess;
    template <class Cfg, int NZ2>
    void go() {
        if (a.F > 1) launch<Cfg, NZ2, true>();
        else launch<Cfg, NZ2, false>();
    }
    template <class Cfg, int NZ2, bool MULTIF>
    void launch() {
        static LdsAttrMask attr_mask{0};
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

template <class Cfg>
__global__ void __launch_bounds__(Cfg::NT, 3) k_fast_rows_fwd(FastRowsFwdArgs a, int rows) {
    DevPhaseCtx<RowFwdState> ctx;
    fast_rows_fwd_body<Cfg>(ctx, reinterpret_cast<c32*>(fc_smem), a, (int)blockIdx.x, rows);
}

struct FastRowsFwdLauncher {
    const FastRowsFwdArgs& a;
    int rows;
    hipStream_t s;
    hipError_t err = hipSuccess;
    template <class Cfg>
    void go() {
        static LdsAttrMask attr_mask{0};
        const size_t lds = (size_t)Cfg::LDS_ELEMS * sizeof(c32);
        err = ensure_lds_attr(k_fast_rows_fwd<Cfg>, attr_mask);
        if (err != hipSuccess) return;
        const int groups = (rows + Cfg::RPW - 1) / Cfg::RPW;
        hipLaunchKernelGGL((k_fast_rows_fwd<Cfg>), dim3(groups), dim3(Cfg::NT), lds, s, a, rows);
        err = hipGetLastError();
    }
};

}  // namespace

hipError_t launch_fast_rows_fwd(int L, const FastRowsFwdArgs& a, int rows, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    FastRowsFwdLauncher l{a, rows, s};
    if (!fast_rows_fwd_dispatch(L, l)) return hipErrorInvalidValue;
    return l.err;
}

hipError_t launch_fast_rows(int L, int nz2, const FastRowsArgs& a, int rows, int kernels, int order, hipStream_t s) {
    if (rows <= 0 || kernels <= 0) return hipSuccess;
    FastRowsLauncher l{a, rows, kernels, s, order};
    if (!fast_rows_dispatch(L, nz2, l)) return hipErrorInvalidValue;
    return l.err;
}

}  // namespace fc
