// kernels_common.hpp -- device-side glue shared by the kernels_*.hip translation units: the phase
// context that binds the workgroup bodies to HIP threads and barriers, the dynamic-LDS symbol and
// the per-device LDS attribute helper.  (The kernels are split over several translation units
// only so that they compile in parallel.)
#pragma once
#include <atomic>
#include <type_traits>

#include "kernels.hpp"

namespace fc {
namespace {

extern __shared__ __attribute__((aligned(16))) unsigned char fc_smem[];

// Raises the dynamic-LDS limit of a kernel once per device (the attribute is per device; a
// process may drive several GPUs through different plans).
// (the per-device threads of fftconv_multi_convolve come through here at the same time: the mask is atomic)
using LdsAttrMask = std::atomic<unsigned long long>;
template <class K>
hipError_t ensure_lds_attr(K kernel, LdsAttrMask& done_mask) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done_mask.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) done_mask.fetch_or(bit, std::memory_order_release);
    return e;
}

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

}  // namespace
}  // namespace fc
