// fc_common.hpp -- shared host/device definitions for the FFT-convolution engine.
//
// Everything in the *.hpp files of this directory is written once and compiled twice:
//   - by hipcc for gfx950 (the product: kernels.hip -> libfftconv.so), and
//   - by g++ for the host inside tests/emu (a test-only executor that runs the very same
//     workgroup bodies sequentially so the algorithm can be checked without a GPU).
// The host build is test infrastructure; the product library contains no CPU compute path.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define FC_HD __host__ __device__ __forceinline__
#define FC_D __device__ __forceinline__
#else
#define FC_HD inline
#define FC_D inline
#endif

#if defined(__clang__)
#define FC_NOUNROLL _Pragma("clang loop unroll(disable)")
#elif defined(__GNUC__)
#define FC_NOUNROLL _Pragma("GCC unroll 1")
#else
#define FC_NOUNROLL
#endif

// A value that is the same in every lane of the wave but that the compiler cannot prove uniform
// (e.g. loaded from a table at a uniform index): readfirstlane moves it to an SGPR, which keeps
// the addresses built from it scalar (no VGPR pairs, no waterfall loops).
#if defined(__HIP_DEVICE_COMPILE__)
#define FC_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#else
#define FC_UNIFORM(x) (x)
#endif

// Streaming store of one complex value: the intermediate and the maps are written once and not
// re-read by this kernel (FC_NT_STORES=0 restores plain stores for A/B runs).
#ifndef FC_NT_STORES
#define FC_NT_STORES 1
#endif
#if defined(__HIP_DEVICE_COMPILE__) && FC_NT_STORES
#define FC_STREAM_STORE(ptr, val)                                                          \
    do {                                                                                   \
        const ::fc::c32 fc_v_ = (val);                                                     \
        typedef float fc_f2_ __attribute__((ext_vector_type(2)));                          \
        fc_f2_ fc_t_ = {fc_v_.x, fc_v_.y};                                                 \
        __builtin_nontemporal_store(fc_t_, reinterpret_cast<fc_f2_*>(ptr));                \
    } while (0)
#else
#define FC_STREAM_STORE(ptr, val) (*(ptr) = (val))
#endif

// Streaming 16-byte load (data read exactly once): FC_NT_LOADS=0 restores plain loads for A/B runs.
#ifndef FC_NT_LOADS
#define FC_NT_LOADS 1
#endif
#if defined(__HIP_DEVICE_COMPILE__) && FC_NT_LOADS
#define FC_STREAM_LOAD16(dst, ptr)                                                         \
    do {                                                                                   \
        typedef float fc_f4_ __attribute__((ext_vector_type(4)));                          \
        const fc_f4_ fc_t_ = __builtin_nontemporal_load(reinterpret_cast<const fc_f4_*>(ptr)); \
        (dst).a.x = fc_t_.x; (dst).a.y = fc_t_.y; (dst).b.x = fc_t_.z; (dst).b.y = fc_t_.w;  \
    } while (0)
#else
#define FC_STREAM_LOAD16(dst, ptr) ((dst) = *reinterpret_cast<const ::fc::c32x2*>(ptr))
#endif

// Scheduling fence (device only): keeps the compiler from interleaving the unrolled rounds of a
// phase, which multiplies their register footprint.
#if defined(__HIP_DEVICE_COMPILE__)
#define FC_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define FC_SCHED_FENCE() ((void)0)
#endif

// Makes a lane-varying integer opaque to the optimiser at this point (device only): what is computed
// from it inside a loop is not loop-invariant any more, so it is recomputed per iteration
// instead of being hoisted and kept in registers across the whole loop.
#if defined(__HIP_DEVICE_COMPILE__)
#define FC_OPAQUE(x) asm volatile("" : "+v"(x))
#else
#define FC_OPAQUE(x) ((void)0)
#endif

namespace fc {

struct alignas(8) c32 {
    float x, y;
};

FC_HD c32 mk(float x, float y) { c32 r; r.x = x; r.y = y; return r; }
FC_HD c32 operator+(c32 a, c32 b) { return mk(a.x + b.x, a.y + b.y); }
FC_HD c32 operator-(c32 a, c32 b) { return mk(a.x - b.x, a.y - b.y); }
FC_HD c32 cmul(c32 a, c32 b) { return mk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
// a * conj(b)
FC_HD c32 cmulc(c32 a, c32 b) { return mk(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }
FC_HD c32 conj(c32 a) { return mk(a.x, -a.y); }
FC_HD c32 scale(c32 a, float s) { return mk(a.x * s, a.y * s); }

// Compile-time loop: f(std::integral_constant<int, I>) for I in [0, N).
template <int I>
struct IC {
    static constexpr int value = I;
    constexpr operator int() const { return I; }
};
template <int B, int E, class F>
FC_HD void static_for(F&& f) {
    if constexpr (B < E) {
        f(IC<B>{});
        static_for<B + 1, E>(f);
    }
}

struct alignas(16) c32x2 {
    c32 a, b;
};

constexpr int FC_MAX_STAGES = 12;

// One stage of the in-place mixed-radix transform (see fft_lds.hpp).
struct StageDesc {
    int R;       // radix
    int m;       // sub-length: the stage works on blocks of n = R*m elements
    int tw_off;  // offset (in c32) of this stage's twiddles in the plan's table; -1 if m == 1
};

struct FftDesc {
    int L;   // transform length = product of radices
    int ns;  // number of stages
    StageDesc st[FC_MAX_STAGES];
};

// Entry of the real<->half-complex pair table (see kernels_body.hpp).
struct alignas(16) PairEntry {
    int a;  // LDS position of bin k
    int b;  // LDS position of bin M-k
    c32 w;  // exp(-2*pi*i*k/N), N = 2M
};

}  // namespace fc
