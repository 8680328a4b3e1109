// kernels.hip -- HIP kernels for gfx950 (MI355X): thin wrappers that bind the HIP thread index
// and workgroup barrier to the workgroup bodies of kernels_body.hpp.
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

}  // namespace

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
