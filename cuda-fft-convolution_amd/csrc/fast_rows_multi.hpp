// fast_rows_multi.hpp -- spectral-row kernel, several maps per workgroup.
//
// A workgroup per (row group, kernel) pair pays the launch, the stage-2 twiddle fill, the exposed latency of its
// kernel-row load and a fresh fetch of the image-spectrum row, and it retires only after its stores have
// drained.  Here a workgroup keeps its image-spectrum row IN REGISTERS and walks G consecutive
// kernels with it (G = 1: the one-map launch of small problems):
//   * the image-spectrum row is fetched once per G maps instead of once per map (the largest
//     read of this kernel: 8*C bytes per map -> 8*C/G);
//   * the next kernel's row (kw complex values: one or two registers per thread) is prefetched
//     right after stage 1 has consumed the current one, so its latency hides behind four phases;
//   * the stores of map m drain while map m + 1 is transformed.
// P5 ends with a barrier because the next
// map's P1 overwrites the LDS row.
// MULTIF (F > 1, the reference's sumAlongFeatures case): the walk runs over (map, feature) pairs.  The
// image-spectrum row of feature f cannot stay in registers (F rows), so it is fetched at the start of
// P3 -- from the XCD's L2: the workgroups walking the same row for other maps run beside this one --
// and lands while the stage-3 butterfly is computed; the feature sum is kept in registers and only
// the last feature runs the inverse phases.  What the walk still saves per map: the launch, the
// twiddle fill, the store drain and the exposed kernel-row load of every (map, feature) step.
#pragma once
#include "fast_rows.hpp"

namespace fc {

template <class C, bool MULTIF = false>
struct RowMultiState {
    c32 s[C::R3];        // image spectrum of this thread's stage-3 butterfly (F = 1: whole walk; MULTIF: one step)
    c32 acc[MULTIF ? C::R3 : 1];   // feature sum of the current map (MULTIF)
    c32 x[C::RND1];      // kernel row of the current / next map
    c32 w1[C::RND1];     // stage-1 base twiddle w_L^j of this thread's butterflies (same for every map)
    int yoff[C::RND1];   // tiled intermediate: element offset of this thread's rows (same for every map)
};

// columns per tile of the tiled intermediate (pipeline.hpp: Geometry::y_tile_w)
constexpr int FC_Y_TILE_W = 16, FC_Y_TILE_SHIFT = 4;

// LINEAR (chosen by the launcher with fast_rows_multi_linear): see P5.
inline bool fast_rows_multi_linear(const FastRowsArgs& g, int L, int m1) {
    const bool tiled = g.y_row_of != nullptr;
    (void)L;
    return !tiled || ((2 * m1) & ((1 << g.y_tile_shift) - 1)) == 0;     // whole tiles, or half tiles (two chains: even / odd outputs)
}

template <class C, int NZ2, bool LINEAR, bool MULTIF = false, class Ctx>
FC_HD void fast_rows_multi_body(Ctx& ctx, c32* lds, const FastRowsArgs& g, int group, int kernel0, int nk, int rows) {
    constexpr int L = C::L, R1 = C::R1, R2 = C::R2, R3 = C::R3, NT = C::NT, m1 = C::m1, RPW = C::RPW, S1 = C::S1, LR = C::LR;
    using State = RowMultiState<C, MULTIF>;
    const int nF = MULTIF ? g.F : 1;
    c32* tw2 = lds + RPW * LR;
    const int kw = g.kw;
    const int row0 = group * RPW;
    const bool tiled = g.y_row_of != nullptr;
    constexpr bool FOLD = !(FC_ROWS_NO_FOLD);

    auto load_x = [&](int t, State& st, int kernel, int f) {
        const c32* abase = g.A + (size_t)kernel * g.a_kernel_stride + (MULTIF ? (size_t)f * g.a_feat_stride : 0);
        static_for<0, C::RND1>([&](auto r_) {
            constexpr int r = decltype(r_)::value;
            int u = t + NT * r;
            if constexpr (MULTIF) FC_OPAQUE(u);   // F > 1: index arithmetic recomputed per step, not hoisted out of the walk and spilled
            const int rr = u / C::NB1, j = u - rr * C::NB1;
            const int row = row0 + rr;
            st.x[r] = (rr < RPW && row < rows && j < kw) ? abase[(size_t)row * g.a_pitch + j] : mk(0.f, 0.f);
        });
    };

    // once per workgroup: stage-2 twiddles into LDS, first kernel row, image-spectrum row
    ctx.phase_nosync([&](int t, State& st) {
        fc_tw2_fill<R2, R3, NT>(tw2, g.tw2, t);
        load_x(t, st, kernel0, 0);
        // loaded once: inside the walk a global load in P5 would have to be waited for together
        // with the stores issued just before it (one in-order memory counter)
        static_for<0, C::RND1>([&](auto r_) {
            constexpr int r = decltype(r_)::value;
            const int u = t + NT * r;
            const int rr = u / C::NB1, j = u - rr * C::NB1;
            const int row = row0 + rr;
            const bool live = rr < RPW && row < rows;
            st.w1[r] = live ? g.tw1[j] : mk(1.f, 0.f);
            st.yoff[r] = live ? (tiled ? (g.y_row_of[row] << g.y_tile_shift) : row * g.y_pitch) : 0;
        });
        if constexpr (!MULTIF) {
            const int rr = t / C::NB3, q = t - rr * C::NB3;
            if (rr < RPW && row0 + rr < rows) {
                const c32* srow = g.S + (size_t)(row0 + rr) * g.s_pitch;
                static_for<0, R3 / 2>([&](auto h_) {
                    constexpr int h = decltype(h_)::value;
                    c32x2 v = *reinterpret_cast<const c32x2*>(srow + (size_t)(h * C::NB3 + q) * 2);
                    st.s[2 * h] = v.a;
                    st.s[2 * h + 1] = v.b;
                });
            } else {
                static_for<0, R3>([&](auto a_) { st.s[decltype(a_)::value] = mk(0.f, 0.f); });
            }
        }
    });

    for (int m = 0; m < nk; m++) {
        const int kernel = kernel0 + m;
        FC_ROWS_STAMP(0);

        // P1: forward stage 1, pruned (one non-zero input per butterfly) -- as a phase of its own
        // only for the first map of the walk; for the others it is folded into the previous map's P5
        // (FOLD): the thread that has just read the R1 LDS cells of butterfly j for the inverse
        // stage 1 is the only one that ever touches them, so it writes the next map's stage-1
        // outputs into them right away, with the twiddle chain it has at hand -- one phase and one
        // barrier fewer per map
        for (int f = 0; f < nF; f++) {
        if (!FOLD || m == 0 || f > 0) ctx.phase([&](int t, State& st) {
            static_for<0, C::RND1>([&](auto r_) {
                constexpr int r = decltype(r_)::value;
                int u = t + NT * r;
                if constexpr (MULTIF) FC_OPAQUE(u);
                const int rr = u / C::NB1, j = u - rr * C::NB1;
                if (rr < RPW && j < kw) {
                    c32* buf = lds + rr * LR;
                    c32 p[R1];
                    power_chain<R1>(st.w1[r], p);
                    buf[j] = st.x[r];
                    static_for<1, R1>([&](auto c_) {
                        constexpr int c = decltype(c_)::value;
                        buf[c * S1 + j] = cmul(st.x[r], p[c]);
                    });
                }
            });
        });

        // the next kernel row (next feature of this map, or the next map's first) flies during P2..P5
        if (f + 1 < nF) ctx.phase_nosync([&](int t, State& st) { load_x(t, st, kernel, f + 1); });
        else if (m + 1 < nk) ctx.phase_nosync([&](int t, State& st) { load_x(t, st, kernel + 1, 0); });

        FC_ROWS_STAMP(1);
        // P2: forward stage 2
        ctx.phase([&](int t, State&) {
            static_for<0, C::RND2>([&](auto r_) {
                constexpr int r = decltype(r_)::value;
                int u = t + NT * r;
                if constexpr (MULTIF) FC_OPAQUE(u);
                const int rr = u / C::NB2, w = u - rr * C::NB2;
                if (rr < RPW) {
                    const int c1 = w / R3, b = w - c1 * R3;
                    c32* p = lds + rr * LR + c1 * S1 + b;
                    c32 v[R2];
                    static_for<0, R2>([&](auto a_) {
                        constexpr int a = decltype(a_)::value;
                        if constexpr (a < NZ2) v[a] = (a * R3 + b < kw) ? p[a * R3] : mk(0.f, 0.f);
                        else v[a] = mk(0.f, 0.f);
                    });
                    Dft<R2, -1>::template run_nz<NZ2>(v);   // inputs a >= NZ2 are structural zeros
                    p[0] = v[0];
                    fc_tw2_each<R2>(tw2, b, [&](auto c_, c32 w) {
                        constexpr int c = decltype(c_)::value;
                        p[c * R3] = cmul(v[c], w);
                    });
                }
            });
        });

        FC_ROWS_STAMP(2);
        // P3: forward stage 3, product with the image spectrum (registers), inverse stage 3
        const bool last_f = (f == nF - 1);
        ctx.phase([&](int t_, State& st) {
            int t = t_;
            if constexpr (MULTIF) FC_OPAQUE(t);
            const int rr = t / C::NB3, q = t - rr * C::NB3;
            if (rr < RPW) {
                // MULTIF: this feature's image-spectrum row.  Only its first FC_MULTIF_S_EARLY register pairs are requested
                // ahead of the forward butterfly, the rest right after it: with the whole row in flight beside the
                // radix-22 butterfly (its in-register composite form needs ~70 registers of its own) and the feature
                // sum, the kernel spilled 27-50 registers at L = 4224, differently in every translation unit, and a
                // scratch reload shares the in-order memory counter with these very loads (59.7 -> 56.6 us per map at
                // F = 4 with none early and no spills, profiles/r03i_f4_image_row_load_placement.txt)
                // (configurations with several rows per workgroup or a stage 3 above radix 22 keep more per-thread state:
                //  nothing early there)
                constexpr int S_EARLY_CFG = (RPW > 1 || R3 > 22) ? 0 : FC_MULTIF_S_EARLY;
                constexpr int S_EARLY = (S_EARLY_CFG < R3 / 2) ? S_EARLY_CFG : R3 / 2;
                if constexpr (MULTIF && (0) < (S_EARLY)) {
                    if (row0 + rr < rows) {
                        const c32* srow = g.S + (size_t)f * g.s_feat_stride + (size_t)(row0 + rr) * g.s_pitch;
                        static_for<(0), (S_EARLY)>([&](auto h_) {
                            constexpr int h = decltype(h_)::value;
                            c32x2 w = *reinterpret_cast<const c32x2*>(srow + (size_t)(h * C::NB3 + q) * 2);
                            st.s[2 * h] = w.a;
                            st.s[2 * h + 1] = w.b;
                        });
                    } else {
                        static_for<2 * (0), 2 * (S_EARLY)>([&](auto a_) { st.s[decltype(a_)::value] = mk(0.f, 0.f); });
                    }
                }
                c32* p = lds + rr * LR + (q / R2) * S1 + (q % R2) * R3;     // run c of stage-1 block c1: q = c1 * R2 + c
                c32 v[R3];
                static_for<0, R3 / 2>([&](auto h_) {
                    constexpr int h = decltype(h_)::value;
                    c32x2 w = *reinterpret_cast<const c32x2*>(p + 2 * h);
                    v[2 * h] = w.a;
                    v[2 * h + 1] = w.b;
                });
                if constexpr (MULTIF) {
                    // The R3 LDS cells this thread has just read are its own until the next barrier: the
                    // feature sum is parked there while the butterfly runs beside the in-flight image row
                    // (sum + image row + butterfly do not fit the register file together: 58-100 spilled
                    // registers otherwise) and comes back, pair by pair, into the accumulation.
                    if (f > 0) {
                        static_for<0, R3 / 2>([&](auto h_) {
                            constexpr int h = decltype(h_)::value;
                            c32x2 w;
                            w.a = st.acc[2 * h];
                            w.b = st.acc[2 * h + 1];
                            *reinterpret_cast<c32x2*>(p + 2 * h) = w;
                        });
                    }
                    FC_SCHED_FENCE();
                }
                Dft<R3, -1>::run(v);
                if constexpr (MULTIF && (S_EARLY) < (R3 / 2)) {
                    if (row0 + rr < rows) {
                        const c32* srow = g.S + (size_t)f * g.s_feat_stride + (size_t)(row0 + rr) * g.s_pitch;
                        static_for<(S_EARLY), (R3 / 2)>([&](auto h_) {
                            constexpr int h = decltype(h_)::value;
                            c32x2 w = *reinterpret_cast<const c32x2*>(srow + (size_t)(h * C::NB3 + q) * 2);
                            st.s[2 * h] = w.a;
                            st.s[2 * h + 1] = w.b;
                        });
                    } else {
                        static_for<2 * (S_EARLY), 2 * (R3 / 2)>([&](auto a_) { st.s[decltype(a_)::value] = mk(0.f, 0.f); });
                    }
                }
                if constexpr (!MULTIF) {
                    static_for<0, R3>([&](auto a_) {
                        constexpr int a = decltype(a_)::value;
                        v[a] = cmul(v[a], st.s[a]);
                    });
                } else {
                    FC_SCHED_FENCE();
                    static_for<0, R3 / 2>([&](auto h_) {
                        constexpr int h = decltype(h_)::value;
                        c32 pa = cmul(v[2 * h], st.s[2 * h]);
                        c32 pb = cmul(v[2 * h + 1], st.s[2 * h + 1]);
                        if (f > 0) {
                            const c32x2 w = *reinterpret_cast<const c32x2*>(p + 2 * h);
                            pa = pa + w.a;
                            pb = pb + w.b;
                        }
                        st.acc[2 * h] = pa;
                        st.acc[2 * h + 1] = pb;
                        v[2 * h] = pa;
                        v[2 * h + 1] = pb;
                    });
                }
                if (!MULTIF || last_f) {
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
        }   // features

        FC_ROWS_STAMP(3);
        // P4: inverse stage 2
        ctx.phase([&](int t, State&) {
            static_for<0, C::RND2>([&](auto r_) {
                constexpr int r = decltype(r_)::value;
                int u = t + NT * r;
                if constexpr (MULTIF) FC_OPAQUE(u);
                const int rr = u / C::NB2, w = u - rr * C::NB2;
                if (rr < RPW) {
                    const int c1 = w / R3, b = w - c1 * R3;
                    c32* p = lds + rr * LR + c1 * S1 + b;
                    c32 v[R2];
                    v[0] = p[0];
                    fc_tw2_each<R2>(tw2, b, [&](auto c_, c32 w) {
                        constexpr int c = decltype(c_)::value;
                        v[c] = cmulc(p[c * R3], w);
                    });
                    Dft<R2, +1>::run(v);
                    static_for<0, R2>([&](auto a_) {
                        constexpr int a = decltype(a_)::value;
                        p[a * R3] = v[a];
                    });
                }
            });
        });

        FC_ROWS_STAMP(4);
        // P5: inverse stage 1 straight to global memory; the barrier protects the LDS row
        // against the next map's P1
        c32* ybase = g.Y + (size_t)kernel * g.y_kernel_stride;
        ctx.phase([&](int t, State& st) {
            // the next kernel row (prefetched after P1) has had three phases to arrive: take it off
            // the memory counter before the store burst, or the next P1 would wait for these stores
            FC_WAIT_VMEM();
            // LINEAR: the R1 outputs of a butterfly are m1 columns apart; when that is a whole number
            // of layout tiles (or the intermediate is row-major) and nothing is cropped, output a
            // sits at base + a * stride: a scalar base, one 32-bit add per store, no 64-bit tile
            // arithmetic (a fifth of this kernel's VALU instructions; a cropped window adds a compare)
            // Where m1 is an odd number of HALF tiles (2112 = 8.12.22: 264 columns = 16.5 tiles; 288 = 4.6.12: 72) the even and the
            // odd outputs form two such chains, 2 * m1 columns apart each: two bases instead of one (round 4: cfg5 and cfg1 had
            // run the general form below until then).
            if constexpr (LINEAR) {
                char* yb = reinterpret_cast<char*>(ybase);
                constexpr bool TWO_CHAINS = (m1 % FC_Y_TILE_W) != 0;
                constexpr int SA = TWO_CHAINS ? 2 : 1;            // outputs a and a + SA are SA * m1 columns = whole tiles apart
                const unsigned stride_b = (unsigned)((tiled ? ((SA * m1) >> g.y_tile_shift) * g.y_tile_elems : SA * m1) * (int)sizeof(c32));
                static_for<0, C::RND1>([&](auto r_) {
                    constexpr int r = decltype(r_)::value;
                    int u = t + NT * r;
                    FC_OPAQUE(u);
                    const int rr = u / C::NB1, j = u - rr * C::NB1;
                    if (rr < RPW && row0 + rr < rows) {
                        const c32* buf = lds + rr * LR;
                        c32 p[R1];
                        c32 v[R1];
                        if constexpr (FC_ROWSM_DBG & 1) {
                            static_for<0, R1>([&](auto c_) { v[decltype(c_)::value] = st.s[decltype(c_)::value]; });
                            if (FOLD && m + 1 < nk && j < kw) power_chain<R1>(st.w1[r], p);
                        } else {
                        power_chain<R1>(st.w1[r], p);
                        v[0] = buf[j];
                        static_for<1, R1>([&](auto c_) {
                            constexpr int c = decltype(c_)::value;
                            v[c] = cmulc(buf[c * S1 + j], p[c]);
                        });
                        Dft<R1, +1>::run(v);
                        }
                        const int jo = tiled ? (j >> g.y_tile_shift) * g.y_tile_elems + (j & ((1 << g.y_tile_shift) - 1)) : j;
                        const unsigned off0 = (unsigned)(st.yoff[r] + jo) * (unsigned)sizeof(c32);
                        unsigned off1 = off0;                      // base of the odd outputs (TWO_CHAINS)
                        if constexpr (TWO_CHAINS) {
                            const int j1 = j + m1;
                            const int jo1 = tiled ? (j1 >> g.y_tile_shift) * g.y_tile_elems + (j1 & ((1 << g.y_tile_shift) - 1)) : j1;
                            off1 = (unsigned)(st.yoff[r] + jo1) * (unsigned)sizeof(c32);
                        }
                        if (g.wout >= L) {   // nothing cropped (uniform)
                            static_for<0, R1>([&](auto a_) {
                                constexpr int a = decltype(a_)::value;
                                const unsigned base = (TWO_CHAINS && (a & 1)) ? off1 : off0;
                                FC_ROWSM_STORE(reinterpret_cast<c32*>(yb + (size_t)(base + (unsigned)(a / SA) * stride_b)), v[a]);
                            });
                        } else {             // cropped window (cfg4: 4160 columns of the 4224 transform)
                            static_for<0, R1>([&](auto a_) {
                                constexpr int a = decltype(a_)::value;
                                const unsigned base = (TWO_CHAINS && (a & 1)) ? off1 : off0;
                                if (j + a * m1 < g.wout) FC_ROWSM_STORE(reinterpret_cast<c32*>(yb + (size_t)(base + (unsigned)(a / SA) * stride_b)), v[a]);
                            });
                        }
                        if (FOLD && m + 1 < nk && j < kw) {   // forward stage 1 of the next map into the cells just read
                            c32* wbuf = lds + rr * LR;
                            wbuf[j] = st.x[r];
                            static_for<1, R1>([&](auto c_) {
                                constexpr int c = decltype(c_)::value;
                                wbuf[c * S1 + j] = cmul(st.x[r], p[c]);
                            });
                        }
                    }
                    FC_SCHED_FENCE();
                });
            } else
            static_for<0, C::RND1>([&](auto r_) {
                constexpr int r = decltype(r_)::value;
                int u = t + NT * r;
                FC_OPAQUE(u);   // twiddle chains and store offsets are recomputed per map, not kept (spilled) across the loop
                const int rr = u / C::NB1, j = u - rr * C::NB1;
                const int row = row0 + rr;
                if (rr < RPW && row < rows) {
                    const c32* buf = lds + rr * LR;
                    c32 p[R1];
                    power_chain<R1>(st.w1[r], p);
                    c32 v[R1];
                    v[0] = buf[j];
                    static_for<1, R1>([&](auto c_) {
                        constexpr int c = decltype(c_)::value;
                        v[c] = cmulc(buf[c * S1 + j], p[c]);
                    });
                    Dft<R1, +1>::run(v);
                    c32* yrow = ybase + st.yoff[r];
                    static_for<0, R1>([&](auto a_) {
                        constexpr int a = decltype(a_)::value;
                        int w = j + a * m1;
                        if (w < g.wout) {
                            if (tiled) FC_ROWSM_STORE(&yrow[(size_t)(w >> g.y_tile_shift) * g.y_tile_elems + (w & ((1 << g.y_tile_shift) - 1))], v[a]);
                            else FC_ROWSM_STORE(&yrow[w], v[a]);
                        }
                    });
                    if (FOLD && m + 1 < nk && j < kw) {   // forward stage 1 of the next map into the cells just read
                        c32* wbuf = lds + rr * LR;
                        wbuf[j] = st.x[r];
                        static_for<1, R1>([&](auto c_) {
                            constexpr int c = decltype(c_)::value;
                            wbuf[c * S1 + j] = cmul(st.x[r], p[c]);
                        });
                    }
                }
                FC_SCHED_FENCE();
            });
        });
        FC_ROWS_STAMP(5);
    }
}

}  // namespace fc
