// kernels_rows_multi.hip -- the multi-map spectral-row kernel (fast_rows_multi.hpp), the default
// (one of the kernels_*.hip translation units; see kernels_common.hpp).
#include "kernels_common.hpp"

namespace fc {
namespace {

template <class Cfg, int NZ2, bool LINEAR>
__global__ void __launch_bounds__(Cfg::NT, 3) k_fast_rows_multi(FastRowsArgs a, int rows, int kernels, int per_wg) {
    const int group = (int)blockIdx.x;
    const int kernel0 = (int)blockIdx.y * per_wg;
    const int nk = kernels - kernel0 < per_wg ? kernels - kernel0 : per_wg;
    DevPhaseCtx<RowMultiState<Cfg>> ctx;
    fast_rows_multi_body<Cfg, NZ2, LINEAR>(ctx, reinterpret_cast<c32*>(fc_smem), a, group, kernel0, nk, rows);
}

// (Keep the F > 1 kernels in THIS translation unit: compiled in one of their own -- tried for the build time --
// the same source comes out with 49 instead of 27 spilled registers at L = 4224 and the row kernel 20 % slower at
// F = 2 ... 8; hipcc -Rpass-analysis=kernel-resource-usage shows it, tools/f_scaling.py measures it.)
// F > 1: the walk over (map, feature) pairs.  XCD-aware 1-D grid as k_fast_rows' order 2: blocks b and b + 8
// share an XCD (round-robin dispatch; a speed assumption only), XCD x walks row groups x, x + 8, ... with the
// walk index fastest, so the workgroups that need the same F image-spectrum rows run side by side on one L2.
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

// resident workgroups per CU of the F = 1 multi-map kernel a launch with these arguments would use (the runtime's
// occupancy calculator: registers, LDS, waves)
struct FastRowsMultiOccupancy {
    const FastRowsArgs& a;
    int result = 0;
    hipError_t err = hipSuccess;
    template <class Cfg, int NZ2>
    void go() {
        if (fast_rows_multi_linear(a, Cfg::L, Cfg::m1)) query<Cfg, NZ2, true>();
        else query<Cfg, NZ2, false>();
    }
    template <class Cfg, int NZ2, bool LINEAR>
    void query() {
        static LdsAttrMask attr_mask{0};
        err = ensure_lds_attr(k_fast_rows_multi<Cfg, NZ2, LINEAR>, attr_mask);
        if (err != hipSuccess) return;
        err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&result, reinterpret_cast<const void*>(k_fast_rows_multi<Cfg, NZ2, LINEAR>), Cfg::NT,
                                                           (size_t)Cfg::LDS_ELEMS * sizeof(c32));
    }
};

struct FastRowsMultiLauncher {
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
        const size_t lds = (size_t)Cfg::LDS_ELEMS * sizeof(c32);
        const int groups = (rows + Cfg::RPW - 1) / Cfg::RPW;
        const int walks = (kernels + per_wg - 1) / per_wg;
        if (a.F > 1) {
            static LdsAttrMask attr_mask_f{0};
            err = ensure_lds_attr(k_fast_rows_multi_f<Cfg, NZ2, LINEAR>, attr_mask_f);
            if (err != hipSuccess) return;
            const dim3 grid(8 * ((groups + 7) / 8) * walks);
            hipLaunchKernelGGL((k_fast_rows_multi_f<Cfg, NZ2, LINEAR>), grid, dim3(Cfg::NT), lds, s, a, rows, kernels, per_wg, groups, walks);
        } else {
            static LdsAttrMask attr_mask{0};
            err = ensure_lds_attr(k_fast_rows_multi<Cfg, NZ2, LINEAR>, attr_mask);
            if (err != hipSuccess) return;
            const dim3 grid(groups, walks);
            hipLaunchKernelGGL((k_fast_rows_multi<Cfg, NZ2, LINEAR>), grid, dim3(Cfg::NT), lds, s, a, rows, kernels, per_wg);
        }
        err = hipGetLastError();
    }
};

}  // namespace

hipError_t fast_rows_multi_wgs_per_cu(int L, int nz2, const FastRowsArgs& a, int* wgs_per_cu) {
    FastRowsMultiOccupancy q{a};
    if (!fast_rows_dispatch(L, nz2, q)) return hipErrorInvalidValue;
    if (q.err == hipSuccess && wgs_per_cu) *wgs_per_cu = q.result;
    return q.err;
}

hipError_t launch_fast_rows_multi(int L, int nz2, const FastRowsArgs& a, int rows, int kernels, int kernels_per_wg, hipStream_t s) {
    if (rows <= 0 || kernels <= 0) return hipSuccess;
    if (a.F < 1 || kernels_per_wg < 1) return hipErrorInvalidValue;
    FastRowsMultiLauncher l{a, rows, kernels, kernels_per_wg, s};
    if (!fast_rows_dispatch(L, nz2, l)) return hipErrorInvalidValue;
    return l.err;
}

}  // namespace fc
