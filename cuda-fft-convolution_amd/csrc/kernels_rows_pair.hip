// kernels_rows_pair.hip -- the paired-row kernel (fast_rows_pair.hpp, path mode 3: A/B and tests)
// (one of the kernels_*.hip translation units; see kernels_common.hpp).
#include "kernels_common.hpp"

#ifndef FC_PAIR_XCD_REMAP
#define FC_PAIR_XCD_REMAP 0   // 1: XCD-aware (pair, kernel) order for the paired-row kernel (measured slower)
#endif

namespace fc {
namespace {

template <class Cfg, int NZ2, bool MULTIF>
__global__ void __launch_bounds__(2 * Cfg::NT, 3) k_fast_rows_pair(FastRowsPairArgs a) {
    // Workgroup -> (row pair, kernel).  All kernels of a batch multiply by the SAME image-spectrum
    // rows, so the workgroups that share a row pair should run back to back on one XCD and take
    // the rows from its L2: blocks b and b+8 share an XCD (round-robin dispatch; speed only), so
    // XCD x = b % 8 walks pairs x, x+8, ... with the kernel index running fastest.
#if FC_PAIR_XCD_REMAP
    const int b = (int)blockIdx.x;
    const int xcd = b & 7, sq = b >> 3;
    const int pl = sq / a.nk;
    const int kernel = sq - pl * a.nk;
    const int pair = pl * 8 + xcd;
    if (pair >= a.npairs) return;
#else
    const int pair = (int)blockIdx.x, kernel = (int)blockIdx.y;
#endif
    DevPhaseCtx<RowPairState<Cfg, MULTIF>> ctx;
    fast_rows_pair_body<Cfg, NZ2, MULTIF>(ctx, reinterpret_cast<c32*>(fc_smem), a, pair, kernel);
}

struct FastRowsPairLauncher {
    const FastRowsPairArgs& a;
    int pairs, kernels;
    hipStream_t s;
    hipError_t err = hipSuccess;
    template <class Cfg, int NZ2>
    void go() {
        if (a.r.F > 1) launch<Cfg, NZ2, true>();
        else launch<Cfg, NZ2, false>();
    }
    template <class Cfg, int NZ2, bool MULTIF>
    void launch() {
        static unsigned long long attr_mask = 0;
        const size_t lds = (size_t)(2 * (Cfg::L + 16) + Cfg::T2N + Cfg::m1) * sizeof(c32);
        err = ensure_lds_attr(k_fast_rows_pair<Cfg, NZ2, MULTIF>, attr_mask);
        if (err != hipSuccess) return;
        FastRowsPairArgs aa = a;
        aa.nk = kernels;
        aa.npairs = pairs;
#if FC_PAIR_XCD_REMAP
        const dim3 grid(8 * ((pairs + 7) / 8) * kernels);
#else
        const dim3 grid(pairs, kernels);
#endif
        hipLaunchKernelGGL((k_fast_rows_pair<Cfg, NZ2, MULTIF>), grid, dim3(2 * Cfg::NT), lds, s, aa);
        err = hipGetLastError();
    }
};

}  // namespace

hipError_t launch_fast_rows_pair(int L, int nz2, const FastRowsPairArgs& a, int pairs, int kernels, hipStream_t s) {
    if (pairs <= 0 || kernels <= 0) return hipSuccess;
    FastRowsPairLauncher l{a, pairs, kernels, s};
    if (!fast_rows_pair_dispatch(L, nz2, l)) return hipErrorInvalidValue;
    return l.err;
}

}  // namespace fc
