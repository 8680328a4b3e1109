// kernels_cols.hip -- the output-column kernel (fast_cols.hpp)
// (one of the kernels_*.hip translation units; see kernels_common.hpp).
#include "kernels_common.hpp"

namespace fc {
namespace {

template <class Cfg, bool TILED, bool SLICED = false>
__global__ void __launch_bounds__(Cfg::NT, 3) k_fast_cols(FastColsArgs a) {
    DevPhaseCtx<std::conditional_t<TILED, ColPairState<Cfg>, ColState<Cfg>>> ctx;
    fast_cols_body<Cfg, TILED, SLICED>(ctx, reinterpret_cast<c32*>(fc_smem), a, (int)blockIdx.x, (int)gridDim.x);
}

struct FastColsLauncher {
    const FastColsArgs& a;
    int max_wg;
    hipStream_t s;
    hipError_t err = hipSuccess;
    template <class Cfg>
    void go() {
        if (a.y_tiled) launch<Cfg, true>();
        else launch<Cfg, false>();
    }
    template <class Cfg, bool TILED>
    void launch() {
        static LdsAttrMask attr_mask{0};
        const size_t lds = (size_t)Cfg::LDS_ELEMS * sizeof(c32);
        err = ensure_lds_attr(k_fast_cols<Cfg, TILED>, attr_mask);
        if (err != hipSuccess) return;
        // persistent: as many workgroups as fit at once (LDS-limited), one or two per CU
        const int per_cu = (int)((size_t)(160 * 1024) / lds) < 768 / Cfg::NT ? (int)((size_t)(160 * 1024) / lds) : 768 / Cfg::NT;
        const int want = max_wg * (per_cu < 1 ? 1 : per_cu);
        if constexpr (TILED && Cfg::M <= FC_SLICE_MAX_M) {    // small transforms: a partial last round of tiles is dealt in column slices
            FastColsArgs b = a;
            int sgrid = 0;
            if (fast_cols_slice_plan(Cfg::M, Cfg::T, want, b, sgrid)) {
                static LdsAttrMask attr_mask_s{0};
                err = ensure_lds_attr(k_fast_cols<Cfg, TILED, true>, attr_mask_s);
                if (err != hipSuccess) return;
                hipLaunchKernelGGL((k_fast_cols<Cfg, TILED, true>), dim3(sgrid), dim3(Cfg::NT), lds, s, b);
                err = hipGetLastError();
                return;
            }
        }
        const int grid = a.ntiles < want ? a.ntiles : want;
        hipLaunchKernelGGL((k_fast_cols<Cfg, TILED>), dim3(grid), dim3(Cfg::NT), lds, s, a);
        err = hipGetLastError();
    }
};

}  // namespace

hipError_t launch_fast_cols(int M, int T, const FastColsArgs& a, int num_cus, hipStream_t s) {
    if (a.ntiles <= 0) return hipSuccess;
    FastColsLauncher l{a, num_cus, s};
    if (!fast_cols_dispatch(M, T, l)) return hipErrorInvalidValue;
    return l.err;
}

}  // namespace fc
