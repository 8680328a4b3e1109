// kernels_cols_fwd.hip -- the forward-column kernel (fast_cols_fwd.hpp)
// (one of the kernels_*.hip translation units; see kernels_common.hpp).
#include "kernels_common.hpp"

namespace fc {
namespace {

template <class Cfg, int NZ2>
__global__ void __launch_bounds__(Cfg::NT, 3) k_fast_cols_fwd(FastColsFwdArgs a) {
    DevPhaseCtx<ColFwdState> ctx;
    fast_cols_fwd_body<Cfg, NZ2>(ctx, reinterpret_cast<c32*>(fc_smem), a, (int)blockIdx.x, (int)gridDim.x);
}

// Image columns and kernel columns in ONE launch: the two passes are independent (the kernels' h-transform is the
// only image-independent part of a convolve) and each is a small launch of its own at small problem sizes -- cfg1: 16 + 2
// tiles, cfg2: 64 + 64 -- where a kernel boundary (~1.5 us + ramp) is a visible part of the step.  A workgroup first takes
// its share of the image tiles, then of the kernel tiles, the second share rotated by the number of image tiles so that
// workgroups without an image tile start on the kernels at once.
template <class Cfg, int NZ2B>
__global__ void __launch_bounds__(Cfg::NT, 3) k_fast_cols_fwd_pair(FastColsFwdArgs a, FastColsFwdArgs b) {
    DevPhaseCtx<ColFwdState> ctx;
    const int wg = (int)blockIdx.x, nwg = (int)gridDim.x;
    fast_cols_fwd_body<Cfg, Cfg::R2>(ctx, reinterpret_cast<c32*>(fc_smem), a, wg, nwg);
    const int wgb = (int)(((unsigned)wg + (unsigned)nwg - (unsigned)(a.ntiles % nwg)) % (unsigned)nwg);
    fast_cols_fwd_body<Cfg, NZ2B>(ctx, reinterpret_cast<c32*>(fc_smem), b, wgb, nwg);
}

struct FastColsFwdPairLauncher {
    const FastColsFwdArgs& a;
    const FastColsFwdArgs& b;
    int num_cus;
    hipStream_t s;
    hipError_t err = hipSuccess;
    template <class Cfg, int NZ2B>
    void go() {
        static LdsAttrMask attr_mask{0};
        const size_t lds = (size_t)Cfg::LDS_ELEMS * sizeof(c32);
        err = ensure_lds_attr(k_fast_cols_fwd_pair<Cfg, NZ2B>, attr_mask);
        if (err != hipSuccess) return;
        const int per_cu = (int)((size_t)(160 * 1024) / lds) < 768 / Cfg::NT ? (int)((size_t)(160 * 1024) / lds) : 768 / Cfg::NT;
        const int want = num_cus * (per_cu < 1 ? 1 : per_cu);
        const int total = a.ntiles + b.ntiles;
        const int grid = total < want ? total : want;
        hipLaunchKernelGGL((k_fast_cols_fwd_pair<Cfg, NZ2B>), dim3(grid), dim3(Cfg::NT), lds, s, a, b);
        err = hipGetLastError();
    }
};

struct FastColsFwdLauncher {
    const FastColsFwdArgs& a;
    int num_cus;
    hipStream_t s;
    hipError_t err = hipSuccess;
    template <class Cfg, int NZ2>
    void go() {
        static LdsAttrMask attr_mask{0};
        const size_t lds = (size_t)Cfg::LDS_ELEMS * sizeof(c32);
        err = ensure_lds_attr(k_fast_cols_fwd<Cfg, NZ2>, attr_mask);
        if (err != hipSuccess) return;
        const int per_cu = (int)((size_t)(160 * 1024) / lds) < 768 / Cfg::NT ? (int)((size_t)(160 * 1024) / lds) : 768 / Cfg::NT;
        const int want = num_cus * (per_cu < 1 ? 1 : per_cu);
        const int grid = a.ntiles < want ? a.ntiles : want;
        hipLaunchKernelGGL((k_fast_cols_fwd<Cfg, NZ2>), dim3(grid), dim3(Cfg::NT), lds, s, a);
        err = hipGetLastError();
    }
};

}  // namespace

hipError_t launch_fast_cols_fwd(int M, int T, bool pruned, const FastColsFwdArgs& a, int num_cus, hipStream_t s) {
    if (a.ntiles <= 0) return hipSuccess;
    FastColsFwdLauncher l{a, num_cus, s};
    if (!fast_cols_fwd_dispatch(M, T, pruned, l)) return hipErrorInvalidValue;
    return l.err;
}

hipError_t launch_fast_cols_fwd_pair(int M, int T, const FastColsFwdArgs& image, const FastColsFwdArgs& kernels, bool kernels_pruned,
                                     int num_cus, hipStream_t s) {
    if (image.ntiles <= 0 || kernels.ntiles <= 0) return hipErrorInvalidValue;
    FastColsFwdPairLauncher l{image, kernels, num_cus, s};
    if (!fast_cols_fwd_dispatch(M, T, kernels_pruned, l)) return hipErrorInvalidValue;
    return l.err;
}

}  // namespace fc
