// fast_rows_fwd.hpp -- specialised forward w-transform of the image spectrum rows (the W half of
// cufftExecR2C on the image, src/cudaConvolutionFFT.cu:167), once per image.
//
// Same job as rows_fwd_body (kernels_body.hpp) for the transform lengths that have a RowCfg
// (fast_rows.hpp): three in-place LDS stages with compile-time radices, stage-1 twiddles from one
// table entry + a power chain, stage 3 in registers, and the result stored straight in the
// "register order" the spectral-row kernels read (16 bytes per lane, coalesced), scaled by
// 1 / (Lh * Lw) (src/cudaConvolutionFFT.cu:270 folded in).  In place: a workgroup reads its rows
// completely (stage 1) before it writes them (stage 3).
#pragma once
#include "fast_rows.hpp"

namespace fc {

struct FastRowsFwdArgs {
    c32* S;          // row r at S + r * pitch (all feature planes back to back)
    int pitch;
    int nvalid;      // columns < nvalid hold data, the rest of the row is zero padding
    float scale;
    const c32* tw1;  // w_L^j, j in [0, m1)
    const c32* tw2;  // stage-2 table [(c-1)*R3 + b]
};

struct RowFwdState {};

template <class C, class Ctx>
FC_HD void fast_rows_fwd_body(Ctx& ctx, c32* lds, const FastRowsFwdArgs& g, int group, int rows) {
    constexpr int R1 = C::R1, R2 = C::R2, R3 = C::R3, NT = C::NT, m1 = C::m1, RPW = C::RPW, S1 = C::S1, LR = C::LR;
    using State = RowFwdState;
    c32* tw2 = lds + RPW * LR;
    const int row0 = group * RPW;

    // stage 1 (radix R1 over inputs m1 apart, decimation in frequency): block c of the row gets
    // (sum_a x[j + a*m1] w_R1^{a c}) * w_L^{j c}; the loads zero-fill the padding
    ctx.phase([&](int t, State&) {
        fc_tw2_fill<C::R2, C::R3, C::NT>(tw2, g.tw2, t);
        static_for<0, C::RND1>([&](auto r_) {
            constexpr int r = decltype(r_)::value;
            const int u = t + NT * r;
            const int rr = u / C::NB1, j = u - rr * C::NB1;
            const int row = row0 + rr;
            if (rr < RPW && row < rows) {
                const c32* src = g.S + (size_t)row * g.pitch;
                c32 v[R1];
                static_for<0, R1>([&](auto a_) {
                    constexpr int a = decltype(a_)::value;
                    const int x = j + a * m1;
                    v[a] = (x < g.nvalid) ? src[x] : mk(0.f, 0.f);
                });
                Dft<R1, -1>::run(v);
                c32 p[R1];
                power_chain<R1>(g.tw1[j], p);
                c32* buf = lds + rr * LR;
                buf[j] = v[0];
                static_for<1, R1>([&](auto c_) {
                    constexpr int c = decltype(c_)::value;
                    buf[c * S1 + j] = cmul(v[c], p[c]);
                });
            }
        });
    });

    // stage 2 (radix R2, sub-length R3), all inputs
    ctx.phase([&](int t, State&) {
        static_for<0, C::RND2>([&](auto r_) {
            constexpr int r = decltype(r_)::value;
            const int u = t + NT * r;
            const int rr = u / C::NB2, w = u - rr * C::NB2;
            if (rr < RPW && row0 + rr < rows) {
                const int c1 = w / R3, b = w - c1 * R3;
                c32* p = lds + rr * LR + c1 * S1 + b;
                c32 v[R2];
                static_for<0, R2>([&](auto a_) {
                    constexpr int a = decltype(a_)::value;
                    v[a] = p[a * R3];
                });
                Dft<R2, -1>::run(v);
                p[0] = v[0];
                fc_tw2_each<R2>(tw2, b, [&](auto c_, c32 w) {
                    constexpr int c = decltype(c_)::value;
                    p[c * R3] = cmul(v[c], w);
                });
            }
        });
    });

    // stage 3 in registers, scaled, stored in register order: element a of butterfly q at
    // ((a >> 1) * NB3 + q) * 2 + (a & 1)
    ctx.phase_nosync([&](int t, State&) {
        const int rr = t / C::NB3, q = t - rr * C::NB3;
        const int row = row0 + rr;
        if (rr < RPW && row < rows) {
            const c32* p = lds + rr * LR + (q / R2) * S1 + (q % R2) * R3;
            c32 v[R3];
            static_for<0, R3 / 2>([&](auto h_) {
                constexpr int h = decltype(h_)::value;
                c32x2 w = *reinterpret_cast<const c32x2*>(p + 2 * h);
                v[2 * h] = w.a;
                v[2 * h + 1] = w.b;
            });
            Dft<R3, -1>::run(v);
            c32* dst = g.S + (size_t)row * g.pitch;
            static_for<0, R3 / 2>([&](auto h_) {
                constexpr int h = decltype(h_)::value;
                c32x2 w;
                w.a = scale(v[2 * h], g.scale);
                w.b = scale(v[2 * h + 1], g.scale);
                *reinterpret_cast<c32x2*>(dst + (size_t)(h * C::NB3 + q) * 2) = w;
            });
        }
    });
}

}  // namespace fc
