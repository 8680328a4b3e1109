// fast_rows_pair.hpp -- paired-row variant of the specialised spectral-row kernel.
//
// One workgroup of 2*NT threads transforms the two spectrum rows of h-frequencies k and M-k side
// by side (each half runs exactly the phases of fast_rows.hpp on its own LDS row), and in the
// last phase one thread finishes the inverse stage 1 of BOTH rows for the same w, so that it
// holds X_k[w] and X_{M-k}[w] in registers and can apply, elementwise in w, the half-spectrum
// merge the output kernel would otherwise do in a separate table-driven LDS pass:
//     Z_k     = (X_k + conj(X_{M-k})) + i e^{+2 pi i k/N} (X_k - conj(X_{M-k}))
//     Z_{M-k} = conj(S) + i conj(G)            (same S, G; see kernels_body.hpp cols_c2r_body)
// The two result rows are written as ADJACENT rows of the tiled intermediate
//     Y[w / 8][row][w % 8],  rows (2u, 2u+1) for pair u,
// i.e. every (pair, 8-column tile) is one full 128-byte line, and an 8-column tile of the output
// kernel is one contiguous block of M rows (the Nyquist row has been folded into row Z_0).
// Special workgroups: (k = 0, M) produces the single row Z_0 = (X_0 + X_M) + i (X_0 - X_M);
// the self-paired k = M/2 produces Z = 2 conj(X).
#pragma once
#include "fast_rows.hpp"

#ifndef FC_PAIR_P5
#define FC_PAIR_P5 1   // final phase: 0 = both rows of a butterfly in one thread, 1 = lane-exchange variant
#endif
#ifndef FC_ROWS_DBG
#define FC_ROWS_DBG 0   // timing experiments only (wrong results): 1 no stores, 2 no final phase at all, 4 no S loads
#endif

namespace fc {

struct alignas(16) RowPair {
    int rowA, rowB;   // spectrum rows (generic order) of bins k and M-k
    int outA, outB;   // tile rows of Z_k and Z_{M-k}; outB < 0: single output
    int kind;         // 0 regular pair, 1 DC/Nyquist, 2 self-paired middle bin
    int pad;
    c32 w;            // exp(-2 pi i k / N)
};

struct FastRowsPairArgs {
    FastRowsArgs r;          // Y, y_tile_elems (= M*8) as for the tiled layout; y_row_of unused
    const RowPair* pairs;    // one entry per row pair
    int npairs, nk;          // row pairs per kernel, kernels in this launch (set by the launcher)
};

template <class C, bool MULTIF>
struct RowPairState : RowState<C, MULTIF> {
    // inverse stage-1 results of this thread's items, exchanged with lane ^ 8 in the last phase
    c32 xo[(2 * C::NB1 + 2 * C::NT - 1) / (2 * C::NT)][C::R1];
};

template <class C, int NZ2, bool MULTIF, class Ctx>
FC_HD void fast_rows_pair_body(Ctx& ctx, c32* lds, const FastRowsPairArgs& ga, int pair_index, int kernel) {
    static_assert(C::RPW == 1, "the paired variant is built on one-row-per-half configurations");
    constexpr int L = C::L, R1 = C::R1, R2 = C::R2, R3 = C::R3, NT = C::NT, m1 = C::m1;
    constexpr int NTW = 2 * NT;
    constexpr int LB = L + 16;   // row B starts 32 banks away from row A: the A/B interleaved reads of P5 do not collide
    constexpr int XR = row_x_rounds<C>();
    using State = RowPairState<C, MULTIF>;
    using PState = State;
    const FastRowsArgs& g = ga.r;
    const RowPair pr = ga.pairs[pair_index];
    const int nF = MULTIF ? g.F : 1;
    c32* tw2 = lds + 2 * LB;
    c32* tw1 = tw2 + C::T2N;     // stage-1 base twiddles in LDS too: no global-load latency in P1/P5
    const int kw = g.kw;


    for (int f = 0; f < nF; f++) {
        // P0: global loads of this (row pair, feature)
        ctx.phase_nosync([&](int t, State& st) {
            const int grp = t / NT, tt = t - grp * NT;
            const int row = grp ? pr.rowB : pr.rowA;
            const c32* arow = g.A + (size_t)kernel * g.a_kernel_stride + (size_t)f * g.a_feat_stride + (size_t)row * g.a_pitch;
            const c32* srow = g.S + (size_t)f * g.s_feat_stride + (size_t)row * g.s_pitch;
            static_for<0, XR>([&](auto r_) {
                constexpr int r = decltype(r_)::value;
                int j = tt + NT * r;
                st.x[r] = (j < kw) ? arow[j] : mk(0.f, 0.f);
            });
            if ((FC_ROWS_DBG & 4) == 0 && tt < C::NB3) {
                static_for<0, R3 / 2>([&](auto h_) {
                    constexpr int h = decltype(h_)::value;
                    c32x2 v = *reinterpret_cast<const c32x2*>(srow + (size_t)(h * C::NB3 + tt) * 2);
                    st.s[2 * h] = v.a;
                    st.s[2 * h + 1] = v.b;
                });
            }
        });

        // once per workgroup, AFTER the row loads are in flight: twiddle tables into LDS
        // (barrier: P1 reads tw1 entries written by other threads)
        if (f == 0) ctx.phase([&](int t, State&) {
            for (int i = t; i < C::T2N; i += NTW) tw2[i] = g.tw2[i];
            for (int i = t; i < m1; i += NTW) tw1[i] = g.tw1[i];
        });

        // P1: forward stage 1, pruned
        ctx.phase([&](int t, State& st) {
            const int grp = t / NT, tt = t - grp * NT;
            c32* buf = lds + grp * LB;
            static_for<0, XR>([&](auto r_) {
                constexpr int r = decltype(r_)::value;
                int j = tt + NT * r;
                if (j < kw) {
                    c32 p[R1];
                    power_chain<R1>(tw1[j], p);
                    buf[j] = st.x[r];
                    static_for<1, R1>([&](auto c_) {
                        constexpr int c = decltype(c_)::value;
                        buf[c * m1 + j] = cmul(st.x[r], p[c]);
                    });
                }
            });
        });

        // P2: forward stage 2, inputs a >= NZ2 are structural zeros
        ctx.phase([&](int t, State&) {
            const int grp = t / NT, tt = t - grp * NT;
            c32* buf = lds + grp * LB;
            static_for<0, C::RND2>([&](auto r_) {
                constexpr int r = decltype(r_)::value;
                int u = tt + NT * r;
                if (u < C::NB2) {
                    int c1 = u / R3, b = u - c1 * R3;
                    c32* p = buf + c1 * m1 + b;
                    c32 v[R2];
                    static_for<0, R2>([&](auto a_) {
                        constexpr int a = decltype(a_)::value;
                        if constexpr (a < NZ2) v[a] = (a * R3 + b < kw) ? p[a * R3] : mk(0.f, 0.f);
                        else v[a] = mk(0.f, 0.f);
                    });
                    Dft<R2, -1>::run(v);
                    p[0] = v[0];
                    static_for<1, R2>([&](auto c_) {
                        constexpr int c = decltype(c_)::value;
                        p[c * R3] = cmul(v[c], tw2[(c - 1) * R3 + b]);
                    });
                }
            });
        });

        // P3: forward stage 3, product, (feature sum,) inverse stage 3 -- in registers
        const bool last = (f == nF - 1);
        ctx.phase([&](int t, State& st) {
            const int grp = t / NT, tt = t - grp * NT;
            if (tt < C::NB3) {
                c32* p = lds + grp * LB + tt * R3;
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
                        c32 q = cmul(v[a], st.s[a]);
                        st.acc[a] = (f == 0) ? q : st.acc[a] + q;
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
        const int grp = t / NT, tt = t - grp * NT;
        c32* buf = lds + grp * LB;
        static_for<0, C::RND2>([&](auto r_) {
            constexpr int r = decltype(r_)::value;
            int u = tt + NT * r;
            if (u < C::NB2) {
                int c1 = u / R3, b = u - c1 * R3;
                c32* p = buf + c1 * m1 + b;
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

    // P5 (variant 0): inverse stage 1 of both rows for the same w in one thread, merge, tiled store
    if constexpr (FC_PAIR_P5 == 0) {
    constexpr int RND1P = (C::NB1 + NTW - 1) / NTW;
    c32* ybase0 = g.Y + (size_t)kernel * g.y_kernel_stride;
    ctx.phase_nosync([&](int t, State&) {
        static_for<0, RND1P>([&](auto r_) {
            constexpr int r = decltype(r_)::value;
            int j = t + NTW * r;
            if (j < C::NB1) {
                c32 p[R1];
                power_chain<R1>(tw1[j], p);
                c32 va[R1], vb[R1];
                va[0] = lds[j];
                vb[0] = lds[LB + j];
                static_for<1, R1>([&](auto c_) {
                    constexpr int c = decltype(c_)::value;
                    va[c] = cmulc(lds[c * m1 + j], p[c]);
                    vb[c] = cmulc(lds[LB + c * m1 + j], p[c]);
                });
                Dft<R1, +1>::run(va);
                Dft<R1, +1>::run(vb);
                static_for<0, R1>([&](auto a_) {
                    constexpr int a = decltype(a_)::value;
                    const int w = j + a * m1;
                    if (w < g.wout) {
                        c32* yt = ybase0 + (size_t)(w >> 3) * g.y_tile_elems + (w & 7);
                        const c32 xk = va[a], xm = vb[a];
                        if (pr.kind == 0) {
                            c32 S = mk(xk.x + xm.x, xk.y - xm.y);
                            c32 D = mk(xk.x - xm.x, xk.y + xm.y);
                            c32 G = cmulc(D, pr.w);
                            yt[pr.outA * 8] = mk(S.x - G.y, S.y + G.x);
                            yt[pr.outB * 8] = mk(S.x + G.y, -S.y + G.x);
                        } else if (pr.kind == 1) {
                            yt[pr.outA * 8] = mk(xk.x + xm.x, xk.x - xm.x);
                        } else {
                            yt[pr.outA * 8] = mk(2.f * xk.x, -2.f * xk.y);
                        }
                    }
                });
            }
        });
    });
    return;
    }
    // P5: inverse stage 1.  Item q = thread + NTW*r covers butterfly j = 8*(q/16) + q%8 of row
    // A (q & 8 == 0) or row B: within every 16 lanes, lanes i and i+8 hold X_k[w] and X_{M-k}[w]
    // for the same w, swap them with one lane exchange (DPP row_ror:8 on the GPU), each computes
    // its own merged row, and the 16 lanes store one full 128-byte line of the tiled intermediate.
    if constexpr (FC_ROWS_DBG & 2) return;
    constexpr int NIT = 2 * C::NB1;                 // single-row butterflies of the pair
    constexpr int RND5 = (NIT + NTW - 1) / NTW;
    static_assert(NTW % 16 == 0 && C::NB1 % 8 == 0, "item decode needs whole 16-lane groups");
    ctx.phase_nosync([&](int t, PState& st) {
        static_for<0, RND5>([&](auto r_) {
            constexpr int r = decltype(r_)::value;
            const int q = t + NTW * r;
            if (q < NIT) {
                const int j = ((q >> 4) << 3) + (q & 7);
                const c32* buf = lds + ((q >> 3) & 1) * LB;
                c32 p[R1];
                power_chain<R1>(tw1[j], p);
                c32 v[R1];
                v[0] = buf[j];
                static_for<1, R1>([&](auto c_) {
                    constexpr int c = decltype(c_)::value;
                    v[c] = cmulc(buf[c * m1 + j], p[c]);
                });
                Dft<R1, +1>::run(v);
                static_for<0, R1>([&](auto a_) {
                    constexpr int a = decltype(a_)::value;
                    st.xo[r][a] = v[a];
                });
            }
        });
    });
    c32* ybase = g.Y + (size_t)kernel * g.y_kernel_stride;
    if (pr.kind == 0) {
        // regular pair: one store per value, 16 lanes = (2 rows x 8 columns) = one 128-byte line
        ctx.phase_nosync([&](int t, PState& st) {
            static_for<0, RND5>([&](auto r_) {
                constexpr int r = decltype(r_)::value;
                const int q = t + NTW * r;
                // whole 16-lane groups are in or out together (NIT % 16 == 0): the exchange is safe
                if (q < NIT) {
                    const int j = ((q >> 4) << 3) + (q & 7);
                    const bool sel = (q >> 3) & 1;   // false: holds X_k, writes Z_k; true: X_{M-k}, Z_{M-k}
                    c32* yt = ybase + (size_t)((sel ? pr.outB : pr.outA) * 8) + (j & 7);
                    static_for<0, R1>([&](auto a_) {
                        constexpr int a = decltype(a_)::value;
                        const c32 own = st.xo[r][a];
                        c32 peer;
                        if constexpr (FC_ROWS_DBG & 8) peer = mk(own.y, own.x);
                        else peer = ctx.peer8(t, [](PState& s) -> c32& { return s.xo[decltype(r_)::value][decltype(a_)::value]; });
                        const int w = j + a * m1;
                        const c32 xk = sel ? peer : own, xm = sel ? own : peer;
                        c32 S = mk(xk.x + xm.x, xk.y - xm.y);
                        c32 D = mk(xk.x - xm.x, xk.y + xm.y);
                        c32 G = cmulc(D, pr.w);
                        const c32 z = sel ? mk(S.x + G.y, G.x - S.y) : mk(S.x - G.y, S.y + G.x);
                        if constexpr (FC_ROWS_DBG & 1) { if (z.x == 1.2345e-30f) yt[(size_t)(w >> 3) * g.y_tile_elems] = z; }
                        else if (w < g.wout) yt[(size_t)(w >> 3) * g.y_tile_elems] = z;
                    });
                }
            });
        });
    } else {
        // DC/Nyquist pair and self-paired middle bin: a single merged row, written by the A lanes
        ctx.phase_nosync([&](int t, PState& st) {
            static_for<0, RND5>([&](auto r_) {
                constexpr int r = decltype(r_)::value;
                const int q = t + NTW * r;
                if (q < NIT) {
                    const int j = ((q >> 4) << 3) + (q & 7);
                    const bool sel = (q >> 3) & 1;
                    c32* yt = ybase + (size_t)(pr.outA * 8) + (j & 7);
                    static_for<0, R1>([&](auto a_) {
                        constexpr int a = decltype(a_)::value;
                        const c32 own = st.xo[r][a];
                        const c32 peer = ctx.peer8(t, [](PState& s) -> c32& { return s.xo[decltype(r_)::value][decltype(a_)::value]; });
                        const int w = j + a * m1;
                        const c32 z = (pr.kind == 1) ? mk(own.x + peer.x, own.x - peer.x) : mk(2.f * own.x, -2.f * own.y);
                        if (!sel && w < g.wout) yt[(size_t)(w >> 3) * g.y_tile_elems] = z;
                    });
                }
            });
        });
    }
}

}  // namespace fc
