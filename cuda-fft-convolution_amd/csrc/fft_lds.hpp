// fft_lds.hpp -- in-place mixed-radix FFT of sequences resident in LDS.
//
// Forward = decimation in frequency, natural order in, DIGIT-REVERSED order out.
// Inverse = the exact stage-by-stage adjoint (decimation in time), digit-reversed in,
// natural out, unnormalised (inverse(forward(x)) = L*x).  Because both directions are in
// place and every butterfly reads and writes only its own R elements, any number of threads
// may share the work with one barrier per stage, and a convolution never needs a reorder
// pass: spectra of both operands are produced (and multiplied) in the same permuted order.
//
// Stage t (radix R, sub-length m, block length n = R*m) on a block starting at B:
//   forward: u_a = x[B + a*m + b];  v = DFT_R(u);  x[B + c*m + b] = v_c * w_n^{b c}
//   inverse: v_c = x[B + c*m + b] * conj(w_n^{b c});  u = IDFT_R(v);  x[B + a*m + b] = u_a
// with w_n = exp(-2 pi i / n).  Bin k = c_0 + R_0 (c_1 + R_1 (...)) of the forward transform
// ends at position c_0 L/R_0 + c_1 L/(R_0 R_1) + ...  (planner.hpp: digit_reverse_position).
//
// The per-stage twiddles are tabulated in access order, tw[(c-1)*m + b], so consecutive
// lanes read consecutive entries.
#pragma once
#include "butterflies.hpp"
#include "fc_common.hpp"

namespace fc {

// The radices the engine is built for.  Even radices go first (large m: unit-stride LDS
// access), odd radices last (lane stride R*8 B is conflict-free when R is odd).
#define FC_FOR_EACH_RADIX(X) X(2) X(3) X(4) X(5) X(7) X(8) X(11) X(13) X(16) X(17)

constexpr bool radix_supported(int r) {
    return r == 2 || r == 3 || r == 4 || r == 5 || r == 7 || r == 8 || r == 11 || r == 13 || r == 16 || r == 17;
}

// One stage over `nseq` sequences of length L stored at buf + seq*pitch.
template <int R, int SGN, class Ctx>
FC_HD void stage_run(const Ctx& ctx, c32* buf, int pitch, int nseq, int L, int m, const c32* tw) {
    const int nb = L / R;  // butterflies per sequence
    const int total = nb * nseq;
    const int n = R * m;
    for (int g = ctx.tid; g < total; g += ctx.nthreads) {
        int seq = 0, j = g;
        if (nseq > 1) {
            seq = g / nb;
            j = g - seq * nb;
        }
        int blk = 0, b = j;
        if (m != nb) {  // more than one block
            blk = j / m;
            b = j - blk * m;
        }
        c32* p = buf + seq * pitch + blk * n + b;
        c32 v[R];
        static_for<0, R>([&](auto a_) {
            constexpr int a = decltype(a_)::value;
            v[a] = p[a * m];
        });
        if constexpr (SGN < 0) {
            Dft<R, SGN>::run(v);
            if (m > 1) {
                static_for<1, R>([&](auto c_) {
                    constexpr int c = decltype(c_)::value;
                    v[c] = cmul(v[c], tw[(c - 1) * m + b]);
                });
            }
        } else {
            if (m > 1) {
                static_for<1, R>([&](auto c_) {
                    constexpr int c = decltype(c_)::value;
                    v[c] = cmulc(v[c], tw[(c - 1) * m + b]);
                });
            }
            Dft<R, SGN>::run(v);
        }
        static_for<0, R>([&](auto a_) {
            constexpr int a = decltype(a_)::value;
            p[a * m] = v[a];
        });
    }
}

template <int SGN, class Ctx>
FC_HD void stage_dispatch(const Ctx& ctx, c32* buf, int pitch, int nseq, int L, const StageDesc& s, const c32* tw_base) {
    const c32* tw = tw_base + (s.tw_off < 0 ? 0 : s.tw_off);
    switch (s.R) {
#define FC_CASE(RR)                                                   \
    case RR:                                                          \
        stage_run<RR, SGN>(ctx, buf, pitch, nseq, L, s.m, tw);        \
        break;
        FC_FOR_EACH_RADIX(FC_CASE)
#undef FC_CASE
        default:
            break;
    }
}

// All stages, forward.  Ends with a barrier.
template <class Ctx>
FC_HD void fft_forward(const Ctx& ctx, c32* buf, int pitch, int nseq, const FftDesc& d, const c32* tw) {
    for (int t = 0; t < d.ns; t++) {
        stage_dispatch<-1>(ctx, buf, pitch, nseq, d.L, d.st[t], tw);
        ctx.sync();
    }
}

// All stages, inverse (unnormalised).  Ends with a barrier.
template <class Ctx>
FC_HD void fft_inverse(const Ctx& ctx, c32* buf, int pitch, int nseq, const FftDesc& d, const c32* tw) {
    for (int t = d.ns - 1; t >= 0; t--) {
        stage_dispatch<+1>(ctx, buf, pitch, nseq, d.L, d.st[t], tw);
        ctx.sync();
    }
}

}  // namespace fc
