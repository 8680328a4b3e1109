// kernels.hip -- HIP kernels for gfx950 (MI355X): thin wrappers that bind the HIP thread index
// and workgroup barrier to the workgroup bodies of kernels_body.hpp.
//
// Roofline: every kernel here is HBM/LDS-bound streaming work (no dense contraction), so MFMA
// is deliberately unused; see DESIGN.md for the algorithmic bytes per launch.
#include "kernels.hpp"

#ifndef FC_PAIR_XCD_REMAP
#define FC_PAIR_XCD_REMAP 0   // 1: XCD-aware (pair, kernel) order for the paired-row kernel (measured slower)
#endif

namespace fc {
namespace {

struct DevCtx {
    int tid, nthreads;
    __device__ __forceinline__ void sync() const { __syncthreads(); }
};

extern __shared__ __attribute__((aligned(16))) unsigned char fc_smem[];

__global__ void __launch_bounds__(512) k_cols_r2c(ColsR2CArgs a) {
    DevCtx ctx{(int)threadIdx.x, (int)blockDim.x};
    cols_r2c_body(ctx, reinterpret_cast<c32*>(fc_smem), a, (int)blockIdx.x, (int)blockIdx.y);
}

__global__ void __launch_bounds__(512) k_rows_fwd(RowsFwdArgs a) {
    DevCtx ctx{(int)threadIdx.x, (int)blockDim.x};
    rows_fwd_body(ctx, reinterpret_cast<c32*>(fc_smem), a, (int)blockIdx.x);
}

__global__ void __launch_bounds__(512) k_spectral_rows(SpectralRowsArgs a) {
    DevCtx ctx{(int)threadIdx.x, (int)blockDim.x};
    spectral_rows_body(ctx, reinterpret_cast<c32*>(fc_smem), a, (int)blockIdx.x, (int)blockIdx.y);
}

__global__ void __launch_bounds__(512) k_cols_c2r(ColsC2RArgs a) {
    DevCtx ctx{(int)threadIdx.x, (int)blockDim.x};
    cols_c2r_body(ctx, reinterpret_cast<c32*>(fc_smem), a, (int)blockIdx.x, (int)blockIdx.y);
}

// Reverses every plane of `pe` floats: for a column-major kh x kw plane that is the flip along both
// axes, kernel(end:-1:1, end:-1:1, f) of demoCudaConvolutionFFT.m:67-69.
__global__ void __launch_bounds__(256) k_flip_planes(const float* __restrict__ src, float* __restrict__ dst, int pe, long total) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long plane = i / pe;
        const int e = (int)(i - plane * pe);
        dst[i] = src[plane * pe + (pe - 1 - e)];
    }
}

// dst window += src window (overlap-add of block results into the full map): maps are column-major,
// h contiguous; the source is clipped to the destination.
__global__ void __launch_bounds__(256) k_add_window(float* __restrict__ dst, int dst_h, int dst_w, size_t dst_map_stride, int y0, int x0,
                                                    const float* __restrict__ src, int src_h, int src_w, size_t src_map_stride) {
    const int x = (int)blockIdx.x, map = (int)blockIdx.y;
    if (x >= src_w || x0 + x >= dst_w) return;
    const float* s = src + (size_t)map * src_map_stride + (size_t)x * src_h;
    float* d = dst + (size_t)map * dst_map_stride + (size_t)(x0 + x) * dst_h + y0;
    const int n = src_h < dst_h - y0 ? src_h : dst_h - y0;
    for (int y = (int)threadIdx.x; y < n; y += (int)blockDim.x) d[y] += s[y];
}

// Raises the dynamic-LDS limit of a kernel once per device (the attribute is per device; a
// process may drive several GPUs through different plans).
template <class K>
hipError_t ensure_lds_attr(K kernel, unsigned long long& done_mask) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done_mask & bit) return hipSuccess;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) done_mask |= bit;
    return e;
}

// ---- fast path -------------------------------------------------------------------------
template <class State>
struct DevPhaseCtx {
    State st;
    template <class F>
    __device__ __forceinline__ void phase(F&& f) {
        f((int)threadIdx.x, st);
        __syncthreads();
    }
    template <class F>
    __device__ __forceinline__ void phase_nosync(F&& f) {
        f((int)threadIdx.x, st);
    }
    template <bool NOSYNC, class F>
    __device__ __forceinline__ void phase_dbg(F&& f) {
        f((int)threadIdx.x, st);
        if (!NOSYNC) __syncthreads();
    }
    // value the accessor designates in lane (this ^ 8): DPP row_ror:8 (rotate by 8 within each
    // row of 16 lanes), no LDS involved
    template <class Acc>
    __device__ __forceinline__ c32 peer8(int, Acc&& acc) {
        const c32 v = acc(st);
        c32 r;
        r.x = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v.x), 0x128, 0xf, 0xf, false));
        r.y = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v.y), 0x128, 0xf, 0xf, false));
        return r;
    }
};

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

template <class Cfg, int NZ2, bool LINEAR>
__global__ void __launch_bounds__(Cfg::NT, 3) k_fast_rows_multi(FastRowsArgs a, int rows, int kernels, int per_wg) {
    const int group = (int)blockIdx.x;
    const int kernel0 = (int)blockIdx.y * per_wg;
    const int nk = kernels - kernel0 < per_wg ? kernels - kernel0 : per_wg;
    DevPhaseCtx<RowMultiState<Cfg>> ctx;
    fast_rows_multi_body<Cfg, NZ2, LINEAR>(ctx, reinterpret_cast<c32*>(fc_smem), a, group, kernel0, nk, rows);
}

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
        static unsigned long long attr_mask = 0;
        const size_t lds = (size_t)Cfg::LDS_ELEMS * sizeof(c32);
        err = ensure_lds_attr(k_fast_rows_multi<Cfg, NZ2, LINEAR>, attr_mask);
        if (err != hipSuccess) return;
        const int groups = (rows + Cfg::RPW - 1) / Cfg::RPW;
        const dim3 grid(groups, (kernels + per_wg - 1) / per_wg);
        hipLaunchKernelGGL((k_fast_rows_multi<Cfg, NZ2, LINEAR>), grid, dim3(Cfg::NT), lds, s, a, rows, kernels, per_wg);
        err = hipGetLastError();
    }
};

template <class Cfg, int MODE>
__global__ void __launch_bounds__(Cfg::NT, 3) k_fast_cols(FastColsArgs a) {
    DevPhaseCtx<std::conditional_t<MODE == 3, ColPairState<Cfg>, ColState<Cfg>>> ctx;
    fast_cols_body<Cfg, MODE>(ctx, reinterpret_cast<c32*>(fc_smem), a, (int)blockIdx.x, (int)gridDim.x);
}

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

#ifndef FC_PAIR_XCD_REMAP
#define FC_PAIR_XCD_REMAP 0
#endif

template <class Cfg>
__global__ void __launch_bounds__(Cfg::NT, Cfg::NT / 256) k_fast_cols_wide(FastColsWideArgs a) {
    DevPhaseCtx<ColWideState<Cfg>> ctx;
    fast_cols_wide_body<Cfg>(ctx, reinterpret_cast<c32*>(fc_smem), a, (int)blockIdx.x, (int)gridDim.x);
}

struct FastColsWideLauncher {
    const FastColsWideArgs& a;
    int num_cus;
    hipStream_t s;
    hipError_t err = hipSuccess;
    template <class Cfg>
    void go() {
        static unsigned long long attr_mask = 0;
        const size_t lds = (size_t)Cfg::LDS_ELEMS * sizeof(c32);
        err = ensure_lds_attr(k_fast_cols_wide<Cfg>, attr_mask);
        if (err != hipSuccess) return;
        const int grid = a.ntiles < num_cus ? a.ntiles : num_cus;   // persistent, one workgroup per CU
        hipLaunchKernelGGL((k_fast_cols_wide<Cfg>), dim3(grid), dim3(Cfg::NT), lds, s, a);
        err = hipGetLastError();
    }
};

template <class Cfg, int NZ2>
__global__ void __launch_bounds__(Cfg::NT, 3) k_fast_cols_fwd(FastColsFwdArgs a) {
    DevPhaseCtx<ColFwdState> ctx;
    fast_cols_fwd_body<Cfg, NZ2>(ctx, reinterpret_cast<c32*>(fc_smem), a, (int)blockIdx.x, (int)gridDim.x);
}

struct FastColsFwdLauncher {
    const FastColsFwdArgs& a;
    int num_cus;
    hipStream_t s;
    hipError_t err = hipSuccess;
    template <class Cfg, int NZ2>
    void go() {
        static unsigned long long attr_mask = 0;
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

struct FastColsLauncher {
    const FastColsArgs& a;
    int max_wg;
    hipStream_t s;
    hipError_t err = hipSuccess;
    template <class Cfg>
    void go() {
        if (a.y_precombined) {
            if constexpr (Cfg::T == 8) launch<Cfg, 2>();   // precombined tiles are 8 columns wide
            else err = hipErrorInvalidValue;
        } else if (a.y_tiled && a.y_pair_rows) launch<Cfg, 3>();
        else if (a.y_tiled) launch<Cfg, 1>();
        else launch<Cfg, 0>();
    }
    template <class Cfg, int PRE>
    void launch() {
        static unsigned long long attr_mask = 0;
        const size_t lds = (size_t)Cfg::LDS_ELEMS * sizeof(c32);
        err = ensure_lds_attr(k_fast_cols<Cfg, PRE>, attr_mask);
        if (err != hipSuccess) return;
        // persistent: as many workgroups as fit at once (LDS-limited), one or two per CU
        const int per_cu = (int)((size_t)(160 * 1024) / lds) < 768 / Cfg::NT ? (int)((size_t)(160 * 1024) / lds) : 768 / Cfg::NT;
        const int want = max_wg * (per_cu < 1 ? 1 : per_cu);
        const int grid = a.ntiles < want ? a.ntiles : want;
        hipLaunchKernelGGL((k_fast_cols<Cfg, PRE>), dim3(grid), dim3(Cfg::NT), lds, s, a);
        err = hipGetLastError();
    }
};

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

hipError_t launch_fast_rows_multi(int L, int nz2, const FastRowsArgs& a, int rows, int kernels, int kernels_per_wg, hipStream_t s) {
    if (rows <= 0 || kernels <= 0) return hipSuccess;
    if (a.F != 1 || kernels_per_wg < 1) return hipErrorInvalidValue;
    FastRowsMultiLauncher l{a, rows, kernels, kernels_per_wg, s};
    if (!fast_rows_dispatch(L, nz2, l)) return hipErrorInvalidValue;
    return l.err;
}

hipError_t launch_fast_rows_pair(int L, int nz2, const FastRowsPairArgs& a, int pairs, int kernels, hipStream_t s) {
    if (pairs <= 0 || kernels <= 0) return hipSuccess;
    FastRowsPairLauncher l{a, pairs, kernels, s};
    if (!fast_rows_pair_dispatch(L, nz2, l)) return hipErrorInvalidValue;
    return l.err;
}

hipError_t launch_fast_cols(int M, int T, const FastColsArgs& a, int num_cus, hipStream_t s) {
    if (a.ntiles <= 0) return hipSuccess;
    FastColsLauncher l{a, num_cus, s};
    if (!fast_cols_dispatch(M, T, l)) return hipErrorInvalidValue;
    return l.err;
}

hipError_t launch_fast_cols_fwd(int M, int T, bool pruned, const FastColsFwdArgs& a, int num_cus, hipStream_t s) {
    if (a.ntiles <= 0) return hipSuccess;
    FastColsFwdLauncher l{a, num_cus, s};
    if (!fast_cols_fwd_dispatch(M, T, pruned, l)) return hipErrorInvalidValue;
    return l.err;
}

hipError_t launch_fast_cols_wide(int M, const FastColsWideArgs& a, int num_cus, hipStream_t s) {
    if (a.ntiles <= 0) return hipSuccess;
    FastColsWideLauncher l{a, num_cus, s};
    if (!fast_cols_wide_dispatch(M, l)) return hipErrorInvalidValue;
    return l.err;
}

hipError_t launch_flip_planes(const float* src, float* dst, int plane_elems, long nplanes, hipStream_t s) {
    const long total = (long)plane_elems * nplanes;
    if (total <= 0) return hipSuccess;
    const long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(k_flip_planes, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, s, src, dst, plane_elems, total);
    return hipGetLastError();
}

hipError_t launch_add_window(float* dst, int dst_h, int dst_w, size_t dst_map_stride, int y0, int x0, const float* src, int src_h,
                             int src_w, size_t src_map_stride, int nmaps, hipStream_t s) {
    if (nmaps <= 0 || src_w <= 0 || src_h <= 0 || y0 >= dst_h || x0 >= dst_w) return hipSuccess;
    hipLaunchKernelGGL(k_add_window, dim3((unsigned)src_w, (unsigned)nmaps), dim3(256), 0, s, dst, dst_h, dst_w, dst_map_stride, y0, x0,
                       src, src_h, src_w, src_map_stride);
    return hipGetLastError();
}

hipError_t kernels_init() {
    const int lim = 160 * 1024;
    hipError_t e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_cols_r2c), hipFuncAttributeMaxDynamicSharedMemorySize, lim)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_rows_fwd), hipFuncAttributeMaxDynamicSharedMemorySize, lim)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_spectral_rows), hipFuncAttributeMaxDynamicSharedMemorySize, lim)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_cols_c2r), hipFuncAttributeMaxDynamicSharedMemorySize, lim)) != hipSuccess) return e;
    return hipSuccess;
}

hipError_t launch_cols_r2c(const ColsR2CArgs& a, int tiles, int planes, int threads, size_t lds_bytes, hipStream_t s) {
    if (tiles <= 0 || planes <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_cols_r2c, dim3(tiles, planes), dim3(threads), lds_bytes, s, a);
    return hipGetLastError();
}
hipError_t launch_rows_fwd(const RowsFwdArgs& a, int rows, int threads, size_t lds_bytes, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_rows_fwd, dim3(rows), dim3(threads), lds_bytes, s, a);
    return hipGetLastError();
}
hipError_t launch_spectral_rows(const SpectralRowsArgs& a, int rows, int kernels, int threads, size_t lds_bytes, hipStream_t s) {
    if (rows <= 0 || kernels <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_spectral_rows, dim3(rows, kernels), dim3(threads), lds_bytes, s, a);
    return hipGetLastError();
}
hipError_t launch_cols_c2r(const ColsC2RArgs& a, int tiles, int kernels, int threads, size_t lds_bytes, hipStream_t s) {
    if (tiles <= 0 || kernels <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_cols_c2r, dim3(tiles, kernels), dim3(threads), lds_bytes, s, a);
    return hipGetLastError();
}

}  // namespace fc
