// fast_rows_pair.hpp -- the multi-map spectral-row kernel (fast_rows_multi.hpp, F = 1) with TWO rows per workgroup that run ONE PHASE APART.
//
// Why.  A row map is four phases between barriers: forward stage 2 and inverse stage 2 are bound by LDS instruction issue, stage 3 (two
// radix-R3 butterflies and the product in registers) by vector issue, inverse stage 1 by its stores.  Four independent one-row workgroups
// share a CU; whenever their phases coincide the resource of that phase is contended and the others idle.  Measured at 4224 points
// (profiles/r05x_*): two rows per workgroup in the SAME phase (RowCfg RPW = 2, same waves, same LDS per CU) take 13.2 us per four row maps
// and CU -- exactly what twelve waves in one phase predict -- the free-running one-row workgroups 10.1, the vector pipe alone 8.2.
// Offsetting the workgroups' starts does not help (they drift); here the offset is part of the program: the two halves of a workgroup
// (C::NT threads each, their own LDS row, one shared stage-2 twiddle image) execute the same phase sequence one barrier interval apart and
// meet at every barrier, so a vector-bound phase of one row always runs beside an LDS-bound phase of the other.
//
// Same phases, same arithmetic, same memory traffic as fast_rows_multi_body<C, NZ2, LINEAR, false> with C::RPW == 1 (the code of a phase
// is that body's, with the thread index and the LDS row of the half); F = 1 only.
#pragma once
#include "fast_rows_multi.hpp"

namespace fc {

template <class C, int NZ2, bool LINEAR, class Ctx>
FC_HD void fast_rows_pair_body(Ctx& ctx, c32* lds_all, const FastRowsArgs& g, int pair, int kernel0, int nk, int rows) {
    static_assert(C::RPW == 1, "the pair kernel is built from one-row configurations");
    constexpr int L = C::L, R1 = C::R1, R2 = C::R2, R3 = C::R3, NT = C::NT, m1 = C::m1, S1 = C::S1, LR = C::LR;
    using State = RowMultiState<C, false>;
    constexpr int LAG = 1;                       // barrier intervals between the two halves
    c32* tw2 = lds_all + 2 * LR;
    const int kw = g.kw;
    const bool tiled = g.y_row_of != nullptr;

    auto load_x = [&](int t, State& st, int row, bool live, int kernel) {
        const c32* abase = g.A + (size_t)kernel * g.a_kernel_stride;
        static_for<0, C::RND1>([&](auto r_) {
            constexpr int r = decltype(r_)::value;
            const int j = t + NT * r;
            st.x[r] = (live && j < C::NB1 && j < kw) ? abase[(size_t)row * g.a_pitch + j] : mk(0.f, 0.f);
        });
    };

    // once per workgroup: stage-2 twiddles into LDS (the first half fills the shared image), first kernel row, image-spectrum row
    ctx.phase_nosync([&](int tt, State& st) {
        const int half = tt / NT, t = tt - half * NT;
        const int row = 2 * pair + half;
        const bool live = row < rows;
        if (half == 0) fc_tw2_fill<R2, R3, NT>(tw2, g.tw2, t);
        load_x(t, st, row, live, kernel0);
        static_for<0, C::RND1>([&](auto r_) {
            constexpr int r = decltype(r_)::value;
            const int j = t + NT * r;
            const bool lv = live && j < C::NB1;
            st.w1[r] = lv ? g.tw1[j] : mk(1.f, 0.f);
            st.yoff[r] = lv ? (tiled ? (g.y_row_of[row] << g.y_tile_shift) : row * g.y_pitch) : 0;
        });
        if (live && t < C::NB3) {
            const c32* srow = g.S + (size_t)row * g.s_pitch;
            static_for<0, R3 / 2>([&](auto h_) {
                constexpr int h = decltype(h_)::value;
                c32x2 v = *reinterpret_cast<const c32x2*>(srow + (size_t)(h * C::NB3 + t) * 2);
                st.s[2 * h] = v.a;
                st.s[2 * h + 1] = v.b;
            });
        } else {
            static_for<0, R3>([&](auto a_) { st.s[decltype(a_)::value] = mk(0.f, 0.f); });
        }
    });
    // (the twiddle image is read from the first P2 on: two barriers later at the earliest)

    const int nsteps = 1 + 4 * nk;               // local steps of a half: P1 of the first map, then P2 P3 P4 P5 per map
    for (int s = 0; s < nsteps + LAG; s++) {
        ctx.phase([&](int tt, State& st) {
            const int half = tt / NT, t = tt - half * NT;
            const int ls = s - half * LAG;
            if (ls < 0 || ls >= nsteps) return;
            const int row = 2 * pair + half;
            const bool live = row < rows;
            c32* lds = lds_all + half * LR;
            if (ls == 0) {
                // P1: forward stage 1 of the first map, pruned (one non-zero input per butterfly)
                static_for<0, C::RND1>([&](auto r_) {
                    constexpr int r = decltype(r_)::value;
                    const int j = t + NT * r;
                    if (j < C::NB1 && j < kw) {
                        c32 p[R1];
                        power_chain<R1>(st.w1[r], p);
                        lds[j] = st.x[r];
                        static_for<1, R1>([&](auto c_) {
                            constexpr int c = decltype(c_)::value;
                            lds[c * S1 + j] = cmul(st.x[r], p[c]);
                        });
                    }
                });
                if (nk > 1) load_x(t, st, row, live, kernel0 + 1);      // the next kernel row flies during P2..P5
                return;
            }
            const int m = (ls - 1) >> 2, ph = (ls - 1) & 3;
            const int kernel = kernel0 + m;
            if (ph == 0) {
                // P2: forward stage 2 (ahead of it, from the second map on: the kernel row after this one -- this map's was consumed by the
                // fold at the end of the previous map's P5)
                if (m > 0 && m + 1 < nk) load_x(t, st, row, live, kernel + 1);
                static_for<0, C::RND2>([&](auto r_) {
                    constexpr int r = decltype(r_)::value;
                    const int w = t + NT * r;
                    if (w < C::NB2) {
                        const int c1 = w / R3, b = w - c1 * R3;
                        c32* p = lds + c1 * S1 + b;
                        c32 v[R2];
                        static_for<0, R2>([&](auto a_) {
                            constexpr int a = decltype(a_)::value;
                            if constexpr (a < NZ2) v[a] = (a * R3 + b < kw) ? p[a * R3] : mk(0.f, 0.f);
                            else v[a] = mk(0.f, 0.f);
                        });
                        Dft<R2, -1>::template run_nz<NZ2>(v);
                        p[0] = v[0];
                        fc_tw2_each<R2>(tw2, b, [&](auto c_, c32 wv) {
                            constexpr int c = decltype(c_)::value;
                            p[c * R3] = cmul(v[c], wv);
                        });
                    }
                });
            } else if (ph == 1) {
                // P3: forward stage 3, product with the image spectrum (registers), inverse stage 3
                if (t >= C::NB3) return;         // (configurations whose thread count is rounded up past one butterfly per thread)
                const int q = t;
                c32* p = lds + (q / R2) * S1 + (q % R2) * R3;
                c32 v[R3];
                static_for<0, R3 / 2>([&](auto h_) {
                    constexpr int h = decltype(h_)::value;
                    c32x2 w = *reinterpret_cast<const c32x2*>(p + 2 * h);
                    v[2 * h] = w.a;
                    v[2 * h + 1] = w.b;
                });
                Dft<R3, -1>::run(v);
                static_for<0, R3>([&](auto a_) {
                    constexpr int a = decltype(a_)::value;
                    v[a] = cmul(v[a], st.s[a]);
                });
                Dft<R3, +1>::run(v);
                static_for<0, R3 / 2>([&](auto h_) {
                    constexpr int h = decltype(h_)::value;
                    c32x2 w;
                    w.a = v[2 * h];
                    w.b = v[2 * h + 1];
                    *reinterpret_cast<c32x2*>(p + 2 * h) = w;
                });
            } else if (ph == 2) {
                // P4: inverse stage 2
                static_for<0, C::RND2>([&](auto r_) {
                    constexpr int r = decltype(r_)::value;
                    const int w = t + NT * r;
                    if (w < C::NB2) {
                        const int c1 = w / R3, b = w - c1 * R3;
                        c32* p = lds + c1 * S1 + b;
                        c32 v[R2];
                        v[0] = p[0];
                        fc_tw2_each<R2>(tw2, b, [&](auto c_, c32 wv) {
                            constexpr int c = decltype(c_)::value;
                            v[c] = cmulc(p[c * R3], wv);
                        });
                        Dft<R2, +1>::run(v);
                        static_for<0, R2>([&](auto a_) {
                            constexpr int a = decltype(a_)::value;
                            p[a * R3] = v[a];
                        });
                    }
                });
            } else {
                // P5: inverse stage 1 straight to global memory, forward stage 1 of the next map folded in (fast_rows_multi.hpp)
                c32* ybase = g.Y + (size_t)kernel * g.y_kernel_stride;
                FC_WAIT_VMEM();
                const bool fold = m + 1 < nk;
                if constexpr (LINEAR) {
                    char* yb = reinterpret_cast<char*>(ybase);
                    constexpr bool TWO_CHAINS = (m1 % FC_Y_TILE_W) != 0;
                    constexpr int SA = TWO_CHAINS ? 2 : 1;
                    const unsigned stride_b = (unsigned)((tiled ? ((SA * m1) >> g.y_tile_shift) * g.y_tile_elems : SA * m1) * (int)sizeof(c32));
                    static_for<0, C::RND1>([&](auto r_) {
                        constexpr int r = decltype(r_)::value;
                        int j = t + NT * r;
                        FC_OPAQUE(j);
                        if (live && j < C::NB1) {
                            c32 p[R1];
                            c32 v[R1];
                            power_chain<R1>(st.w1[r], p);
                            v[0] = lds[j];
                            static_for<1, R1>([&](auto c_) {
                                constexpr int c = decltype(c_)::value;
                                v[c] = cmulc(lds[c * S1 + j], p[c]);
                            });
                            Dft<R1, +1>::run(v);
                            const int jo = tiled ? (j >> g.y_tile_shift) * g.y_tile_elems + (j & ((1 << g.y_tile_shift) - 1)) : j;
                            const unsigned off0 = (unsigned)(st.yoff[r] + jo) * (unsigned)sizeof(c32);
                            unsigned off1 = off0;
                            if constexpr (TWO_CHAINS) {
                                const int j1 = j + m1;
                                const int jo1 = tiled ? (j1 >> g.y_tile_shift) * g.y_tile_elems + (j1 & ((1 << g.y_tile_shift) - 1)) : j1;
                                off1 = (unsigned)(st.yoff[r] + jo1) * (unsigned)sizeof(c32);
                            }
                            if (g.wout >= L) {
                                static_for<0, R1>([&](auto a_) {
                                    constexpr int a = decltype(a_)::value;
                                    const unsigned base = (TWO_CHAINS && (a & 1)) ? off1 : off0;
                                    FC_ROWSM_STORE(reinterpret_cast<c32*>(yb + (size_t)(base + (unsigned)(a / SA) * stride_b)), v[a]);
                                });
                            } else {
                                static_for<0, R1>([&](auto a_) {
                                    constexpr int a = decltype(a_)::value;
                                    const unsigned base = (TWO_CHAINS && (a & 1)) ? off1 : off0;
                                    if (j + a * m1 < g.wout) FC_ROWSM_STORE(reinterpret_cast<c32*>(yb + (size_t)(base + (unsigned)(a / SA) * stride_b)), v[a]);
                                });
                            }
                            if (fold && j < kw) {
                                lds[j] = st.x[r];
                                static_for<1, R1>([&](auto c_) {
                                    constexpr int c = decltype(c_)::value;
                                    lds[c * S1 + j] = cmul(st.x[r], p[c]);
                                });
                            }
                        }
                        FC_SCHED_FENCE();
                    });
                } else {
                    static_for<0, C::RND1>([&](auto r_) {
                        constexpr int r = decltype(r_)::value;
                        int j = t + NT * r;
                        FC_OPAQUE(j);
                        if (live && j < C::NB1) {
                            c32 p[R1];
                            power_chain<R1>(st.w1[r], p);
                            c32 v[R1];
                            v[0] = lds[j];
                            static_for<1, R1>([&](auto c_) {
                                constexpr int c = decltype(c_)::value;
                                v[c] = cmulc(lds[c * S1 + j], p[c]);
                            });
                            Dft<R1, +1>::run(v);
                            c32* yrow = ybase + st.yoff[r];
                            static_for<0, R1>([&](auto a_) {
                                constexpr int a = decltype(a_)::value;
                                const int w = j + a * m1;
                                if (w < g.wout) {
                                    if (tiled) FC_ROWSM_STORE(&yrow[(size_t)(w >> g.y_tile_shift) * g.y_tile_elems + (w & ((1 << g.y_tile_shift) - 1))], v[a]);
                                    else FC_ROWSM_STORE(&yrow[w], v[a]);
                                }
                            });
                            if (fold && j < kw) {
                                lds[j] = st.x[r];
                                static_for<1, R1>([&](auto c_) {
                                    constexpr int c = decltype(c_)::value;
                                    lds[c * S1 + j] = cmul(st.x[r], p[c]);
                                });
                            }
                        }
                        FC_SCHED_FENCE();
                    });
                }
            }
        });
    }
}

}  // namespace fc
