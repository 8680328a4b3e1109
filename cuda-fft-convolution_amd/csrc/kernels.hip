// kernels.hip -- HIP kernels for gfx950 (MI355X): thin wrappers that bind the HIP thread index
// and workgroup barrier to the workgroup bodies of kernels_body.hpp (generic kernels, small helper
// kernels); the specialised kernels live in kernels_rows*.hip / kernels_cols*.hip.
//
// Roofline: every kernel here is HBM/LDS-bound streaming work (no dense contraction), so MFMA
// is deliberately unused; see DESIGN.md for the algorithmic bytes per launch.
#include "kernels.hpp"

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

// dst map (dst_h x dst_w, contiguous) = the window [off_h, off_h + dst_h) x [off_w, off_w + dst_w) of the
// src map (src_h rows per column): the "full" / "same" / "valid" regions of the padded window
__global__ void __launch_bounds__(256) k_crop_maps(const float* __restrict__ src, int src_h, size_t src_map_stride, float* __restrict__ dst,
                                                   int dst_h, int dst_w, size_t dst_map_stride, int off_h, int off_w) {
    const int x = (int)blockIdx.x, map = (int)blockIdx.y;
    const float* s = src + (size_t)map * src_map_stride + (size_t)(off_w + x) * src_h + off_h;
    float* d = dst + (size_t)map * dst_map_stride + (size_t)x * dst_h;
    for (int y = (int)threadIdx.x; y < dst_h; y += (int)blockDim.x) d[y] = s[y];
}

// Image spectrum between the engine's internal order and the reference's natural order
// [f][FFT_W][FFT_H/2+1] (cufftExecR2C output, src/cudaFFTData.cu:90-103): natural element (x, y) of
// plane f is internal element S[f][row_of[y]][col_of[x]]; `scale` undoes / applies the folded
// normalisation.  A one-off utility: no attempt at coalescing the permuted side.
template <bool EXPORT>
__global__ void __launch_bounds__(256) k_spectrum_reorder(c32* __restrict__ S, size_t s_plane, int s_pitch, c32* __restrict__ nat, int fw, int ch,
                                                          const int* __restrict__ row_of, const int* __restrict__ col_of, float scale) {
    const int x = (int)blockIdx.x, f = (int)blockIdx.y;
    const size_t base = (size_t)f * s_plane + col_of[x];
    c32* n = nat + ((size_t)f * fw + x) * ch;
    for (int y = (int)threadIdx.x; y < ch; y += (int)blockDim.x) {
        const size_t i = base + (size_t)row_of[y] * s_pitch;
        if (EXPORT) n[y] = scale * S[i];
        else S[i] = scale * n[y];
    }
}

// dst map (dst_h x dst_w, contiguous, LARGER than the source) = the src map in its top-left corner,
// zeros elsewhere: the reference's alternative next-power-of-two window (computeFFTsize,
// src/cudaConvFFTData.h:67-94) around the ceil16 window the engine computes
__global__ void __launch_bounds__(256) k_pad_maps(const float* __restrict__ src, int src_h, int src_w, size_t src_map_stride,
                                                  float* __restrict__ dst, int dst_h, size_t dst_map_stride) {
    const int x = (int)blockIdx.x, map = (int)blockIdx.y;
    const float* s = src + (size_t)map * src_map_stride + (size_t)x * src_h;
    float* d = dst + (size_t)map * dst_map_stride + (size_t)x * dst_h;
    const int n = x < src_w ? src_h : 0;
    for (int y = (int)threadIdx.x; y < dst_h; y += (int)blockDim.x) d[y] = y < n ? s[y] : 0.f;
}

}  // namespace

hipError_t launch_pad_maps(const float* src, int src_h, int src_w, size_t src_map_stride, float* dst, int dst_h, int dst_w,
                           size_t dst_map_stride, int nmaps, hipStream_t s) {
    if (nmaps <= 0 || dst_h <= 0 || dst_w <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_pad_maps, dim3((unsigned)dst_w, (unsigned)nmaps), dim3(256), 0, s, src, src_h, src_w, src_map_stride, dst, dst_h,
                       dst_map_stride);
    return hipGetLastError();
}

hipError_t launch_spectrum_reorder(bool to_natural, c32* S, size_t s_plane, int s_pitch, c32* nat, int fw, int ch, int planes,
                                   const int* row_of, const int* col_of, float scale, hipStream_t s) {
    if (fw <= 0 || ch <= 0 || planes <= 0) return hipSuccess;
    if (to_natural) hipLaunchKernelGGL(k_spectrum_reorder<true>, dim3((unsigned)fw, (unsigned)planes), dim3(256), 0, s, S, s_plane, s_pitch, nat, fw, ch, row_of, col_of, scale);
    else hipLaunchKernelGGL(k_spectrum_reorder<false>, dim3((unsigned)fw, (unsigned)planes), dim3(256), 0, s, S, s_plane, s_pitch, nat, fw, ch, row_of, col_of, scale);
    return hipGetLastError();
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

hipError_t launch_crop_maps(const float* src, int src_h, size_t src_map_stride, float* dst, int dst_h, int dst_w, size_t dst_map_stride,
                            int off_h, int off_w, int nmaps, hipStream_t s) {
    if (nmaps <= 0 || dst_h <= 0 || dst_w <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_crop_maps, dim3((unsigned)dst_w, (unsigned)nmaps), dim3(256), 0, s, src, src_h, src_map_stride, dst, dst_h, dst_w,
                       dst_map_stride, off_h, off_w);
    return hipGetLastError();
}

// ---- the specialised kernels: their translation units (kernels_*_g<G>.hip) hold one group of configurations each; the
// ---- first group that has the length takes the launch
#define FC_DECL_GROUP(G)                                                                                                              \
    hipError_t launch_fast_rows_fwd_g##G(int L, const FastRowsFwdArgs& a, int rows, hipStream_t s, bool* matched);                   \
    hipError_t launch_fast_rows_multi_g##G(int L, int nz2, const FastRowsArgs& a, int rows, int kernels, int kernels_per_wg, hipStream_t s, bool* matched); \
    hipError_t fast_rows_multi_wgs_per_cu_g##G(int L, int nz2, const FastRowsArgs& a, int* wgs_per_cu, bool* matched);
FC_DECL_GROUP(0) FC_DECL_GROUP(1) FC_DECL_GROUP(2)
#undef FC_DECL_GROUP
#define FC_DECL_GROUP(G)                                                                                                              \
    hipError_t launch_fast_cols_g##G(int M, int T, const FastColsArgs& a, int num_cus, hipStream_t s, bool* matched);                 \
    hipError_t launch_fast_cols_fwd_g##G(int M, int T, bool pruned, const FastColsFwdArgs& a, int num_cus, hipStream_t s, bool* matched); \
    hipError_t launch_fast_cols_fwd_pair_g##G(int M, int T, const FastColsFwdArgs& image, const FastColsFwdArgs& kernels, bool kernels_pruned, \
                                              int num_cus, hipStream_t s, bool* matched);
FC_DECL_GROUP(0) FC_DECL_GROUP(1)
#undef FC_DECL_GROUP
static_assert(FC_ROW_GROUPS == 3 && FC_COL_GROUPS == 2, "one translation unit per group: keep kernels.hip and the Makefile in step");

hipError_t launch_fast_rows_fwd(int L, const FastRowsFwdArgs& a, int rows, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    bool m = false;
    hipError_t e = launch_fast_rows_fwd_g0(L, a, rows, s, &m);
    if (!m) e = launch_fast_rows_fwd_g1(L, a, rows, s, &m);
    if (!m) e = launch_fast_rows_fwd_g2(L, a, rows, s, &m);
    return e;
}

hipError_t launch_fast_rows_multi(int L, int nz2, const FastRowsArgs& a, int rows, int kernels, int kernels_per_wg, hipStream_t s);

// one map per workgroup: the multi-map walk with a walk length of 1 (round 4: the separate one-map kernel is gone)
hipError_t launch_fast_rows(int L, int nz2, const FastRowsArgs& a, int rows, int kernels, hipStream_t s) {
    return launch_fast_rows_multi(L, nz2, a, rows, kernels, 1, s);
}

hipError_t launch_fast_rows_multi(int L, int nz2, const FastRowsArgs& a, int rows, int kernels, int kernels_per_wg, hipStream_t s) {
    if (rows <= 0 || kernels <= 0) return hipSuccess;
    if (a.F < 1 || kernels_per_wg < 1) return hipErrorInvalidValue;
    bool m = false;
    hipError_t e = launch_fast_rows_multi_g0(L, nz2, a, rows, kernels, kernels_per_wg, s, &m);
    if (!m) e = launch_fast_rows_multi_g1(L, nz2, a, rows, kernels, kernels_per_wg, s, &m);
    if (!m) e = launch_fast_rows_multi_g2(L, nz2, a, rows, kernels, kernels_per_wg, s, &m);
    return e;
}

hipError_t fast_rows_multi_wgs_per_cu(int L, int nz2, const FastRowsArgs& a, int* wgs_per_cu) {
    bool m = false;
    hipError_t e = fast_rows_multi_wgs_per_cu_g0(L, nz2, a, wgs_per_cu, &m);
    if (!m) e = fast_rows_multi_wgs_per_cu_g1(L, nz2, a, wgs_per_cu, &m);
    if (!m) e = fast_rows_multi_wgs_per_cu_g2(L, nz2, a, wgs_per_cu, &m);
    return e;
}

hipError_t launch_fast_cols(int M, int T, const FastColsArgs& a, int num_cus, hipStream_t s) {
    if (a.ntiles <= 0) return hipSuccess;
    bool m = false;
    hipError_t e = launch_fast_cols_g0(M, T, a, num_cus, s, &m);
    if (!m) e = launch_fast_cols_g1(M, T, a, num_cus, s, &m);
    return e;
}

hipError_t launch_fast_cols_fwd(int M, int T, bool pruned, const FastColsFwdArgs& a, int num_cus, hipStream_t s) {
    if (a.ntiles <= 0) return hipSuccess;
    bool m = false;
    hipError_t e = launch_fast_cols_fwd_g0(M, T, pruned, a, num_cus, s, &m);
    if (!m) e = launch_fast_cols_fwd_g1(M, T, pruned, a, num_cus, s, &m);
    return e;
}

hipError_t launch_fast_cols_fwd_pair(int M, int T, const FastColsFwdArgs& image, const FastColsFwdArgs& kernels, bool kernels_pruned,
                                     int num_cus, hipStream_t s) {
    if (image.ntiles <= 0 || kernels.ntiles <= 0) return hipErrorInvalidValue;
    bool m = false;
    hipError_t e = launch_fast_cols_fwd_pair_g0(M, T, image, kernels, kernels_pruned, num_cus, s, &m);
    if (!m) e = launch_fast_cols_fwd_pair_g1(M, T, image, kernels, kernels_pruned, num_cus, s, &m);
    return e;
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
