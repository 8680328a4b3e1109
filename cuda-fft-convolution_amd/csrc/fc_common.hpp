// fc_common.hpp -- shared host/device definitions for the FFT-convolution engine.
//
// Everything in the *.hpp files of this directory is written once and compiled twice:
//   - by hipcc for gfx950 (the product: kernels*.hip -> libfftconv.so), and
//   - by g++ for the host inside tests/emu (a test-only executor that runs the very same
//     workgroup bodies sequentially so the algorithm can be checked without a GPU).
// The host build is test infrastructure; the product library contains no CPU compute path.
#pragma once
#include <cstdint>

#include "fc_instrument.hpp"

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
// re-read by this kernel (fc_instrument.hpp: FC_NT_STORES = 0 restores plain stores in diagnostic builds).
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

// 16-byte load of data this kernel reads once.  FC_NT_LOADS=1 makes it a streaming (nontemporal)
// load; plain loads are the default: the output kernel gathers 64-byte halves of 128-byte lines
// whose other halves are gathered by a neighbouring CU of the same XCD, and a plain load keeps
// the line in that XCD's L2 for it (27.4 vs 28.5 us per map, FETCH_SIZE back to the algorithmic
// bytes once the gather is issued in two halves).
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

// Wait for every outstanding vector-memory operation of this wave (s_waitcnt vmcnt(0); gfx9
// encoding, the other counters left alone).  vmcnt counts loads and stores together and in
// order: a prefetch issued BEFORE a burst of stores can otherwise only be waited for together
// with (most of) those stores.  Placing this right before the burst -- when the prefetch has had
// several phases to arrive -- keeps the store latency off the critical path.
#if defined(__HIP_DEVICE_COMPILE__)
#define FC_WAIT_VMEM() __builtin_amdgcn_s_waitcnt(0x0F70)
#else
#define FC_WAIT_VMEM() ((void)0)
#endif

// Dynamic tile queue of the persistent column kernels (fast_cols.hpp: TileQueue): one returning agent-scope add on a
// counter word, a relaxed agent-scope load of one, and the XCD a workgroup runs on (HW_REG_XCC_ID bits 3:0 -- read, not
// inferred from the block index: with another kernel co-resident the round-robin placement is not what it is alone).  On the
// host (tests/emu: workgroups run one after the other) they are a plain increment / load and the block index modulo 8.
#if defined(__HIP_DEVICE_COMPILE__)
#define FC_QUEUE_TAKE(p) __hip_atomic_fetch_add((p), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define FC_QUEUE_PEEK(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define FC_QUEUE_PUT(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define FC_XCC_ID(wg) ((int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u))
#else
#define FC_QUEUE_PUT(p, v) (*(p) = (v))
#define FC_QUEUE_TAKE(p) ((*(p))++)
#define FC_QUEUE_PEEK(p) (*(p))
#define FC_XCC_ID(wg) ((wg) & 7)
#endif

namespace fc {

// A complex value.  Under clang (hipcc: device code and the host side of the library) it is a
// 2-lane float vector, so that a value always lives in one aligned 64-bit register pair and sums
// and products map to the packed-FP32 instructions of CDNA3/4 (v_pk_add_f32, v_pk_mul_f32,
// v_pk_fma_f32: two lanes per issue slot -- the hot kernels are VALU-issue bound).  The swizzled
// forms (multiplication by +-i, complex products) are single packed instructions with op_sel /
// neg modifiers, which the compiler does not fold by itself: see the FC_PK_* helpers below.
// Under g++ (tests/emu, test infrastructure) it is a plain struct with the same layout.
#if defined(__clang__)
typedef float c32 __attribute__((ext_vector_type(2)));
#define FC_C32_VECTOR 1
#else
struct alignas(8) c32 {
    float x, y;
};
#define FC_C32_VECTOR 0
#endif

FC_HD c32 mk(float x, float y) { c32 r; r.x = x; r.y = y; return r; }
#if !FC_C32_VECTOR
FC_HD c32 operator+(c32 a, c32 b) { return mk(a.x + b.x, a.y + b.y); }
FC_HD c32 operator-(c32 a, c32 b) { return mk(a.x - b.x, a.y - b.y); }
#endif

#if defined(__HIP_DEVICE_COMPILE__) && !defined(FC_NO_PACKED)
#define FC_PACKED 1
#else
#define FC_PACKED 0
#endif

#if FC_PACKED
// One packed instruction with source modifiers.  op_sel picks the half of each source that feeds
// the LOW lane, op_sel_hi the half that feeds the HIGH lane (0 = .x, 1 = .y); neg_lo / neg_hi
// negate a source in the low / high lane.
#define FC_PK2(op, dst, a, b, mods) asm(op " %0, %1, %2 " mods : "=v"(dst) : "v"(a), "v"(b))
#define FC_PK3(op, dst, a, b, c, mods) asm(op " %0, %1, %2, %3 " mods : "=v"(dst) : "v"(a), "v"(b), "v"(c))
#define FC_PK3S(op, dst, a, sb, c, mods) asm(op " %0, %1, %2, %3 " mods : "=v"(dst) : "v"(a), "s"(sb), "v"(c))   // src1 in SGPRs (constants)
#endif

// a + i*b and a - i*b
FC_HD c32 add_jb(c32 a, c32 b) {
#if FC_PACKED
    c32 r;   // (a.x - b.y, a.y + b.x)
    FC_PK2("v_pk_add_f32", r, a, b, "op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]");
    return r;
#else
    return mk(a.x - b.y, a.y + b.x);
#endif
}
FC_HD c32 sub_jb(c32 a, c32 b) {
#if FC_PACKED
    c32 r;   // (a.x + b.y, a.y - b.x)
    FC_PK2("v_pk_add_f32", r, a, b, "op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]");
    return r;
#else
    return mk(a.x + b.y, a.y - b.x);
#endif
}
// a * b
FC_HD c32 cmul(c32 a, c32 b) {
#if FC_PACKED
    c32 t, r;   // t = a.x * b;  r = a.y * (-b.y, b.x) + t
    FC_PK2("v_pk_mul_f32", t, a, b, "op_sel:[0,0] op_sel_hi:[0,1]");
    FC_PK3("v_pk_fma_f32", r, a, b, t, "op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]");
    return r;
#else
    return mk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
#endif
}
// a * conj(b)
FC_HD c32 cmulc(c32 a, c32 b) {
#if FC_PACKED
    c32 t, r;   // t = a.x * (b.x, -b.y);  r = a.y * (b.y, b.x) + t
    FC_PK2("v_pk_mul_f32", t, a, b, "op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[0,1]");
    FC_PK3("v_pk_fma_f32", r, a, b, t, "op_sel:[1,1,0] op_sel_hi:[1,0,1]");
    return r;
#else
    return mk(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
#endif
}
FC_HD c32 conj(c32 a) { return mk(a.x, -a.y); }
FC_HD c32 scale(c32 a, float s) {
#if FC_C32_VECTOR
    return a * s;
#else
    return mk(a.x * s, a.y * s);
#endif
}
// t + c * a (c real)
FC_HD c32 fma_real(float c, c32 a, c32 t) {
#if FC_C32_VECTOR
    c32 cc = {c, c};
    return __builtin_elementwise_fma(cc, a, t);
#else
    return mk(t.x + c * a.x, t.y + c * a.y);
#endif
}
// t + i*s*a and t - i*s*a (s real): (t.x -+ s*a.y, t.y +- s*a.x)
FC_HD c32 fma_js(float s, c32 a, c32 t) {
#if FC_PACKED
    c32 ss = {s, s}, r;
    FC_PK3S("v_pk_fma_f32", r, a, ss, t, "op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_lo:[1,0,0]");
    return r;
#else
    return mk(t.x - s * a.y, t.y + s * a.x);
#endif
}

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

// Stage-2 twiddles of the ROW kernels in LDS (round 5): row b holds the twiddles of that b for c = 1 .. R2 - 1 side by side, rows
// `pitch` entries apart with pitch = 2 (mod 4) -- a thread fetches its R2 - 1 twiddles TWO per 16-byte read (half the LDS
// instructions of one 8-byte read per twiddle: the stage-2 phases take as long as the arithmetic-heavy stage 3 although they issue a
// third of its vector instructions -- they are bound by LDS instruction issue, ~5.5 cycles per wave-instruction and CU whatever the
// width), and the rows of consecutive lanes start 8 (mod 32) banks apart, so the 16-byte reads of a group of lanes do not collide.
// The global table keeps its [(c - 1) * R3 + b] order.  Row kernels -2 ... -5 % (profiles/r05o_paired_twiddle_reads_ab.txt); the
// column kernels keep one 8-byte read per twiddle: the same change left the 8- / 16-column ones where they were and cost the
// 4-column ones 2 %.
constexpr int fc_tw2_pitch(int R2) {
    int p = R2 - 1;
    while (p % 4 != 2) p++;
    return p;
}
template <int R2, int R3, int NT>
FC_HD void fc_tw2_fill(c32* lds_tw2, const c32* table, int t) {
    constexpr int P = fc_tw2_pitch(R2);
    for (int i = t; i < (R2 - 1) * R3; i += NT) {
        const int c1 = i / R3, b = i - c1 * R3;
        lds_tw2[b * P + c1] = table[i];
    }
}
// f(c, twiddle) for c = 1 .. R2 - 1, the twiddles of row b read in pairs
template <int R2, class F>
FC_HD void fc_tw2_each(const c32* lds_tw2, int b, F&& f) {
    constexpr int P = fc_tw2_pitch(R2);
    const c32* row = lds_tw2 + b * P;
    static_for<0, (R2 - 1) / 2>([&](auto h_) {
        constexpr int h = decltype(h_)::value;
        const c32x2 w = *reinterpret_cast<const c32x2*>(row + 2 * h);
        f(IC<1 + 2 * h>{}, w.a);
        f(IC<2 + 2 * h>{}, w.b);
    });
    if constexpr ((R2 - 1) % 2) f(IC<R2 - 1>{}, row[R2 - 2]);
}

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
static_assert(sizeof(c32) == 8 && alignof(c32) == 8 && sizeof(PairEntry) == 16, "complex layout");

}  // namespace fc
