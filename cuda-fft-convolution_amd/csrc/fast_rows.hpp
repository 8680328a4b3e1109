// fast_rows.hpp -- specialised spectral-row kernel (the hot kernel #1 of the path).
//
// Same job as spectral_rows_body (kernels_body.hpp) -- forward w-transform of the kernel's column
// spectrum row, product with the image spectrum row, sum over features, inverse w-transform --
// but with everything the generic kernel decides at run time fixed at compile time:
//   * the transform length L = R1*R2*R3 and its three radices (three in-place LDS stages);
//   * NT threads per row with exactly one stage-3 butterfly per thread, so the last forward
//     stage, the pointwise product and the first inverse stage run back to back IN REGISTERS
//     (one LDS round trip and one barrier fewer, and the image spectrum is consumed straight
//     from the VGPRs it was prefetched into at the start of the row);
//   * the zero padding of the kernel row is never materialised: stage 1 sees one non-zero input
//     per butterfly (pure twiddle scaling) and stage 2 reads only its first NZ2 inputs;
//   * stage-1 twiddles are one table entry + an in-register power chain, stage-2 twiddles a
//     small LDS table; no integer division by run-time values anywhere.
// The image spectrum is read in the "register order" layout (rows_fwd_body stores it that way):
// element a of stage-3 butterfly q at ((a>>1)*NB3 + q)*2 + (a&1), i.e. 16 B per lane, fully
// coalesced.
//
// The body is written as a sequence of phases over a per-thread State so the host emulator can
// run it (tests/emu) before it ever touches a GPU.
#pragma once
#include "butterflies.hpp"
#include "fc_common.hpp"
#include "fc_instrument.hpp"

namespace fc {

// RPW rows of length L are transformed side by side by one workgroup of NT threads (short rows:
// several per workgroup so that every stage still fills the lanes).
template <int L_, int R1_, int R2_, int R3_, int NT_, int RPW_ = 1>
struct RowCfg {
    static constexpr int L = L_, R1 = R1_, R2 = R2_, R3 = R3_, NT = NT_, RPW = RPW_;
    static constexpr int m1 = L / R1;        // stage-1 sub-length (= R2*R3)
    static constexpr int NB1 = m1;           // butterflies per stage and row
    static constexpr int NB2 = R1 * R3;
    static constexpr int NB3 = R1 * R2;
    static constexpr int RND1 = (RPW * NB1 + NT - 1) / NT;
    static constexpr int RND2 = (RPW * NB2 + NT - 1) / NT;
    static constexpr int T2N = (R2 - 1) * R3;  // stage-2 twiddle entries
    static constexpr int LDS_ELEMS = RPW * L + T2N;  // c32
    static_assert(R1 * R2 * R3 == L, "radices must multiply to L");
    static_assert(RPW * NB3 <= NT, "one stage-3 butterfly per thread");
    static_assert(R3 % 2 == 0, "register-order layout pairs stage-3 elements");
    static_assert((R3 * 8) % 16 == 0, "stage-3 runs must be 16-byte aligned");
};

// position of element (q, a) of the register-order layout
template <class C>
FC_HD int reg_order_index(int q, int a) {
    return ((a >> 1) * C::NB3 + q) * 2 + (a & 1);
}

struct FastRowsArgs {
    const c32* A;            // kernel column spectra [n][f][i][a_pitch]
    size_t a_kernel_stride;
    size_t a_feat_stride;
    int a_pitch;
    int kw;
    const c32* S;            // image spectrum, register-order layout, [f][i][s_pitch]
    size_t s_feat_stride;
    int s_pitch;
    c32* Y;                  // [n][i][y_pitch]
    size_t y_kernel_stride;
    int y_pitch;
    int wout;
    int F;
    const c32* tw1;          // w_L^j, j in [0, m1)
    const c32* tw2;          // stage-2 table [(c-1)*R3 + b]
    // tiled intermediate (used when the fast output kernel consumes it): element (row i, w) at
    //   (w / TL) * y_tile_elems + y_row_of[i] * TL + (w % TL),   TL = 1 << y_tile_shift columns (8 or 16)
    // so that a column tile of the output kernel is (part of) one contiguous block whose rows
    // are already in that kernel's LDS order.  y_row_of == nullptr: plain [i][y_pitch] rows.
    const int* y_row_of;
    int y_tile_elems;        // (M+1) * TL
    int y_tile_shift;        // log2(TL)
    unsigned long long* timeline;  // FC_ROWS_TIMELINE builds only: per-phase wall-clock stamps of one workgroup (else unused)
};

template <class C, bool MULTIF>
struct RowState {
    c32 s[C::R3];                    // image spectrum of this thread's stage-3 butterfly
    c32 acc[MULTIF ? C::R3 : 1];     // feature accumulator (F > 1 only)
    c32 x[C::RND1];                  // kernel row prefetch: one value per stage-1 butterfly of this thread
};

// p[c] = w^c, c in [1, R)
template <int R>
FC_HD void power_chain(c32 w, c32 (&p)[R]) {
    p[0] = mk(1.f, 0.f);
    if constexpr (R > 1) p[1] = w;
    static_for<2, R>([&](auto c_) {
        constexpr int c = decltype(c_)::value;
        if constexpr (c % 2 == 0) p[c] = cmul(p[c / 2], p[c / 2]);
        else p[c] = cmul(p[c - 1], w);
    });
}

// kw must satisfy kw <= min(m1, NZ2*R3) (checked at plan time / by the launcher).
template <class C>
constexpr int row_x_rounds() { return C::RND1; }

template <class C, int NZ2, bool MULTIF, class Ctx>
FC_HD void fast_rows_body(Ctx& ctx, c32* lds, const FastRowsArgs& g, int group, int kernel, int rows) {
    constexpr int L = C::L, R1 = C::R1, R2 = C::R2, R3 = C::R3, NT = C::NT, m1 = C::m1, RPW = C::RPW;
    using State = RowState<C, MULTIF>;
    const int nF = MULTIF ? g.F : 1;
    c32* tw2 = lds + RPW * L;
    const int kw = g.kw;
    const int row0 = group * RPW;   // this workgroup owns spectrum rows [row0, row0 + RPW) (those < rows)

    // once per workgroup: stage-2 twiddles into LDS (first used in P2, after P1's barrier)
    ctx.phase_nosync([&](int t, State&) {
        for (int i = t; i < C::T2N; i += NT) tw2[i] = g.tw2[i];
    });

    for (int f = 0; f < nF; f++) {
        const c32* abase = g.A + (size_t)kernel * g.a_kernel_stride + (size_t)f * g.a_feat_stride;
        const c32* sbase = g.S + (size_t)f * g.s_feat_stride;

        // P0: issue the global loads of this (row group, feature): kernel rows and image spectrum
        ctx.phase_nosync([&](int t, State& st) {
            static_for<0, C::RND1>([&](auto r_) {
                constexpr int r = decltype(r_)::value;
                const int u = t + NT * r;
                const int rr = u / C::NB1, j = u - rr * C::NB1;
                const int row = row0 + rr;
                st.x[r] = (rr < RPW && row < rows && j < kw) ? abase[(size_t)row * g.a_pitch + j] : mk(0.f, 0.f);
            });
            const int rr = t / C::NB3, q = t - rr * C::NB3;
            if ((FC_ROWS1_DBG & 2) == 0 && rr < RPW && row0 + rr < rows) {
                const c32* srow = sbase + (size_t)(row0 + rr) * g.s_pitch;
                static_for<0, R3 / 2>([&](auto h_) {
                    constexpr int h = decltype(h_)::value;
                    c32x2 v;
#if FC_NT_SLOADS
                    FC_STREAM_LOAD16(v, srow + (size_t)(h * C::NB3 + q) * 2);
#else
                    v = *reinterpret_cast<const c32x2*>(srow + (size_t)(h * C::NB3 + q) * 2);
#endif
                    st.s[2 * h] = v.a;
                    st.s[2 * h + 1] = v.b;
                });
            }
        });

        // P1: forward stage 1, pruned: only input a = 0 of each butterfly is non-zero, so the
        // outputs are x[j] * w_L^{j c}
        ctx.phase([&](int t, State& st) {
            static_for<0, C::RND1>([&](auto r_) {
                constexpr int r = decltype(r_)::value;
                const int u = t + NT * r;
                const int rr = u / C::NB1, j = u - rr * C::NB1;
                if (rr < RPW && j < kw) {
                    c32* buf = lds + rr * L;
                    c32 p[R1];
                    power_chain<R1>(g.tw1[j], p);
                    buf[j] = st.x[r];
                    static_for<1, R1>([&](auto c_) {
                        constexpr int c = decltype(c_)::value;
                        buf[c * m1 + j] = cmul(st.x[r], p[c]);
                    });
                }
            });
        });

        // P2: forward stage 2 (radix R2, sub-length R3), inputs a >= NZ2 are structural zeros
        ctx.phase([&](int t, State&) {
            static_for<0, C::RND2>([&](auto r_) {
                constexpr int r = decltype(r_)::value;
                const int u = t + NT * r;
                const int rr = u / C::NB2, w = u - rr * C::NB2;
                if (rr < RPW) {
                    const int c1 = w / R3, b = w - c1 * R3;
                    c32* p = lds + rr * L + c1 * m1 + b;
                    c32 v[R2];
                    static_for<0, R2>([&](auto a_) {
                        constexpr int a = decltype(a_)::value;
                        if constexpr (a < NZ2) v[a] = (a * R3 + b < kw) ? p[a * R3] : mk(0.f, 0.f);
                        else v[a] = mk(0.f, 0.f);
                    });
                    Dft<R2, -1>::template run_nz<NZ2>(v);   // inputs a >= NZ2 are structural zeros
                    p[0] = v[0];
                    static_for<1, R2>([&](auto c_) {
                        constexpr int c = decltype(c_)::value;
                        p[c * R3] = cmul(v[c], tw2[(c - 1) * R3 + b]);
                    });
                }
            });
        });

        // P3: forward stage 3, product with the image spectrum, (feature sum,) inverse stage 3
        const bool last = (f == nF - 1);
        ctx.phase([&](int t, State& st) {
            const int rr = t / C::NB3, q = t - rr * C::NB3;
            if (rr < RPW) {
                c32* p = lds + rr * L + q * R3;
                c32 v[R3];
                static_for<0, R3 / 2>([&](auto h_) {
                    constexpr int h = decltype(h_)::value;
                    c32x2 w = *reinterpret_cast<const c32x2*>(p + 2 * h);
                    v[2 * h] = w.a;
                    v[2 * h + 1] = w.b;
                });
                Dft<R3, -1>::run(v);
                if constexpr (!MULTIF) {
                    static_for<0, R3>([&](auto a_) {
                        constexpr int a = decltype(a_)::value;
                        v[a] = cmul(v[a], st.s[a]);
                    });
                } else {
                    static_for<0, R3>([&](auto a_) {
                        constexpr int a = decltype(a_)::value;
                        c32 pr = cmul(v[a], st.s[a]);
                        st.acc[a] = (f == 0) ? pr : st.acc[a] + pr;
                        v[a] = st.acc[a];
                    });
                }
                if (last) {
                    Dft<R3, +1>::run(v);
                    static_for<0, R3 / 2>([&](auto h_) {
                        constexpr int h = decltype(h_)::value;
                        c32x2 w;
                        w.a = v[2 * h];
                        w.b = v[2 * h + 1];
                        *reinterpret_cast<c32x2*>(p + 2 * h) = w;
                    });
                }
            }
        });
    }

    // P4: inverse stage 2
    ctx.phase([&](int t, State&) {
        static_for<0, C::RND2>([&](auto r_) {
            constexpr int r = decltype(r_)::value;
            const int u = t + NT * r;
            const int rr = u / C::NB2, w = u - rr * C::NB2;
            if (rr < RPW) {
                const int c1 = w / R3, b = w - c1 * R3;
                c32* p = lds + rr * L + c1 * m1 + b;
                c32 v[R2];
                v[0] = p[0];
                static_for<1, R2>([&](auto c_) {
                    constexpr int c = decltype(c_)::value;
                    v[c] = cmulc(p[c * R3], tw2[(c - 1) * R3 + b]);
                });
                Dft<R2, +1>::run(v);
                static_for<0, R2>([&](auto a_) {
                    constexpr int a = decltype(a_)::value;
                    p[a * R3] = v[a];
                });
            }
        });
    });

    // P5: inverse stage 1 straight to global memory (natural w order, coalesced per a)
    if constexpr (FC_ROWS1_DBG & 4) return;
    const bool tiled = g.y_row_of != nullptr;
    c32* ybase = g.Y + (size_t)kernel * g.y_kernel_stride;
    ctx.phase_nosync([&](int t, State&) {
        static_for<0, C::RND1>([&](auto r_) {
            constexpr int r = decltype(r_)::value;
            const int u = t + NT * r;
            const int rr = u / C::NB1, j = u - rr * C::NB1;
            const int row = row0 + rr;
            if (rr < RPW && row < rows) {
                const c32* buf = lds + rr * L;
                c32 p[R1];
                power_chain<R1>(g.tw1[j], p);
                c32 v[R1];
                v[0] = buf[j];
                static_for<1, R1>([&](auto c_) {
                    constexpr int c = decltype(c_)::value;
                    v[c] = cmulc(buf[c * m1 + j], p[c]);
                });
                Dft<R1, +1>::run(v);
                c32* yrow = ybase + (tiled ? ((size_t)g.y_row_of[row] << g.y_tile_shift) : (size_t)row * g.y_pitch);
                static_for<0, R1>([&](auto a_) {
                    constexpr int a = decltype(a_)::value;
                    int w = j + a * m1;
                    if ((FC_ROWS1_DBG & 1) ? (v[a].x == 1.2345e-30f) : (w < g.wout)) {
                        if (tiled) FC_STREAM_STORE(&yrow[(size_t)(w >> g.y_tile_shift) * g.y_tile_elems + (w & ((1 << g.y_tile_shift) - 1))], v[a]);
                        else FC_STREAM_STORE(&yrow[w], v[a]);
                    }
                });
            }
        });
    });
}

}  // namespace fc
