// kernels_rows_multi_f.hip -- the multi-map spectral-row kernel for F > 1: the walk over (map, feature) pairs
// (fast_rows_multi.hpp, MULTIF).  A translation unit of its own so that it compiles beside kernels_rows_multi.hip
// (one of the kernels_*.hip translation units; see kernels_common.hpp).
#include "kernels_common.hpp"

namespace fc {
namespace {

// XCD-aware 1-D grid as k_fast_rows' order 2: blocks b and b + 8 share an XCD (round-robin dispatch; a speed
// assumption only), XCD x walks row groups x, x + 8, ... with the walk index fastest, so the workgroups that
// need the same F image-spectrum rows run side by side on one L2.
template <class Cfg, int NZ2, bool LINEAR>
__global__ void __launch_bounds__(Cfg::NT, 3) k_fast_rows_multi_f(FastRowsArgs a, int rows, int kernels, int per_wg, int groups, int walks) {
    const int b = (int)blockIdx.x;
    const int xcd = b & 7, sq = b >> 3;
    const int gl = sq / walks;
    const int walk = sq - gl * walks;
    const int group = gl * 8 + xcd;
    if (group >= groups) return;
    const int kernel0 = walk * per_wg;
    const int nk = kernels - kernel0 < per_wg ? kernels - kernel0 : per_wg;
    DevPhaseCtx<RowMultiState<Cfg, true>> ctx;
    fast_rows_multi_body<Cfg, NZ2, LINEAR, true>(ctx, reinterpret_cast<c32*>(fc_smem), a, group, kernel0, nk, rows);
}

struct FastRowsMultiFLauncher {
    const FastRowsArgs& a;
    int rows, kernels, per_wg;
    hipStream_t s;
    hipError_t err = hipSuccess;
    template <class Cfg, int NZ2>
    void go() {
        if (fast_rows_multi_linear(a, Cfg::L, Cfg::m1)) launch<Cfg, NZ2, true>();
        else launch<Cfg, NZ2, false>();
    }
    template <class Cfg, int NZ2, bool LINEAR>
    void launch() {
        static unsigned long long attr_mask = 0;
        const size_t lds = (size_t)Cfg::LDS_ELEMS * sizeof(c32);
        const int groups = (rows + Cfg::RPW - 1) / Cfg::RPW;
        const int walks = (kernels + per_wg - 1) / per_wg;
        err = ensure_lds_attr(k_fast_rows_multi_f<Cfg, NZ2, LINEAR>, attr_mask);
        if (err != hipSuccess) return;
        const dim3 grid(8 * ((groups + 7) / 8) * walks);
        hipLaunchKernelGGL((k_fast_rows_multi_f<Cfg, NZ2, LINEAR>), grid, dim3(Cfg::NT), lds, s, a, rows, kernels, per_wg, groups, walks);
        err = hipGetLastError();
    }
};

}  // namespace

hipError_t launch_fast_rows_multi_f(int L, int nz2, const FastRowsArgs& a, int rows, int kernels, int kernels_per_wg, hipStream_t s) {
    FastRowsMultiFLauncher l{a, rows, kernels, kernels_per_wg, s};
    if (!fast_rows_dispatch(L, nz2, l)) return hipErrorInvalidValue;
    return l.err;
}

}  // namespace fc
