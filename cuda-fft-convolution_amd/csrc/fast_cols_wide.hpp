// fast_cols_wide.hpp -- 16-column variant of the specialised output kernel.
//
// Why: with 8-column tiles the gather reads 64-byte pieces (half of every 128-byte line of the
// 16-column tiled intermediate, the layout the row kernel writes fastest); measured, a contiguous
// gather is worth ~6 us of the 35 us per map.  Sixteen columns of M complex bins do not fit the
// 160 KB of LDS -- so the transform is split at its first (radix-2) stage:
//
//   M = 2*H.  In the decimation-in-time inverse, stage 1 (radix 2, sub-length H) is the LAST one:
//       z[j] = uA[j] + uB[j] conj(w_M^j),   z[j+H] = uA[j] - uB[j] conj(w_M^j),   j < H,
//   where uA / uB are the H-point inverse transforms of the even / odd bins.  The pair merge of the
//   half spectrum (bins k and M-k) never crosses the two halves (k and M-k have the same parity).
//
// So one persistent workgroup per CU does, per tile of 16 columns:
//   half A (even bins; rows 0..H-1 + the Nyquist row): land -> merge pairs -> radix R4 -> radix R3
//          -> radix R2 with the results kept IN REGISTERS (24 complex per thread);
//   half B (odd bins):  land -> merge pairs -> radix R4 -> radix R3 -> radix R2, combined with the
//          registers of half A through the final radix-2 butterfly and stored straight to the map.
// LDS holds one half of the tile at a time (16 x (H+2) bins = 135 KB); each half is one fully
// contiguous 135 KB block of the intermediate (128 bytes per row), gathered one half ahead into
// registers while the other half is transformed.
//
// Tile layout written by the row kernel (tile = 16 columns, 128 bytes per row):
//   rows [0, H)      even bins in LDS order of the H-point plan, row H: Nyquist bin,
//   rows [H+1, 2H+1) odd bins in LDS order.
#pragma once
#include "butterflies.hpp"
#include "fast_rows.hpp"  // power_chain, c32x2
#include "fc_common.hpp"

namespace fc {

template <int H_, int R2_, int R3_, int R4_, int NT_>
struct ColWideCfg {
    static constexpr int H = H_, M = 2 * H_, R2 = R2_, R3 = R3_, R4 = R4_, NT = NT_, T = 16;
    static constexpr int m2 = H / R2;                 // radix-R2 stage sub-length (= R3*R4)
    static constexpr int NB4 = R2 * R3;               // radix-R4 butterflies per column and half
    static constexpr int NB3 = R2 * R4;
    static constexpr int NB2 = m2;
    static constexpr int LP = ((H + 1 + 13) / 16) * 16 + 2;   // column pitch: >= H+1 (Nyquist slot), == 2 mod 16
    static constexpr int NUNIT = (H * T) / 2;                 // 16-byte gather units per half
    static constexpr int UPT = (NUNIT + NT - 1) / NT;         // ... per thread
    static constexpr int RND4 = (NB4 * T + NT - 1) / NT;
    static constexpr int RND3 = (NB3 * T + NT - 1) / NT;
    static constexpr int RND2 = (NB2 * T + NT - 1) / NT;
    static constexpr int NPA = H / 2 + 1;             // pair items of half A: DC, pairs k = 2i, middle (i = H/2)
    static constexpr int NPB = H / 2;                 // half B: pairs k = 2i + 1
    static constexpr int RNDP = (NPA * T + NT - 1) / NT;
    static constexpr int T3N = (R3 - 1) * R4;
    static constexpr int NWH = M / 32 + 2;
    // LDS image (c32 units)
    static constexpr int OFF_T3 = T * LP;
    static constexpr int OFF_TA = OFF_T3 + T3N;       // w_H^j',  j' < m2 (radix-R2 stage base twiddles)
    static constexpr int OFF_TF = OFF_TA + m2;        // w_M^j',  j' < m2 (final radix-2 stage base twiddles)
    static constexpr int OFF_WH = OFF_TF + m2;
    static constexpr int OFF_WL = OFF_WH + NWH;
    static constexpr int OFF_PA = OFF_WL + 32;        // NPA dwords
    static constexpr int OFF_PB = OFF_PA + (NPA + 1) / 2;
    static constexpr int LDS_ELEMS = OFF_PB + (NPB + 1) / 2;
    static constexpr int TILE_ROWS = 2 * H + 1;
    static_assert(R2 * R3 * R4 == H, "radices must multiply to H");

    static_assert(H % 2 == 0, "pairs stay inside a half");
    static_assert(LDS_ELEMS * 8 <= 160 * 1024, "LDS budget");
};

struct FastColsWideArgs {
    const c32* Y;            // tiled: [n][w / 16][TILE_ROWS][16]
    size_t y_kernel_stride;
    float* out;              // kernel n at out + n*out_kernel_stride; (h, w) at w*fft_h + h
    size_t out_kernel_stride;
    int fft_h, fft_w;        // output window: fft_h <= 2M (cropped), fft_w % 16 == 0
    int tiles_per_kernel;    // fft_w / 16
    int ntiles;
    const c32* tw3;          // radix-R3 stage table [(c-1)*R4 + b]
    const c32* twA;          // w_H^j', j' < m2
    const c32* twF;          // w_M^j', j' < m2
    const c32* wh;           // w_N^(32 i), N = 2M
    const c32* wl;           // w_N^i, i < 32
    const unsigned* ppA;     // NPA entries a | b << 16 (local LDS positions), item i <-> bin k = 2i
    const unsigned* ppB;     // NPB entries, item i <-> bin k = 2i + 1
};

template <class C>
struct ColWideState {
    c32x2 pre[C::UPT];
    c32x2 pre_ny;
    c32 ua[C::RND2][C::R2];   // half A after its radix-R2 stage, kept across half B
};

template <class C, class Ctx>
FC_HD void fast_cols_wide_body(Ctx& ctx, c32* lds, const FastColsWideArgs& g, int wg, int nwg) {
    constexpr int H = C::H, M = C::M, R2 = C::R2, R3 = C::R3, R4 = C::R4, T = C::T, NT = C::NT, LP = C::LP, m2 = C::m2;
    constexpr int T2 = T / 2;   // 16-byte units per 128-byte row
    using State = ColWideState<C>;
    c32* tw3 = lds + C::OFF_T3;
    c32* twA = lds + C::OFF_TA;
    c32* twF = lds + C::OFF_TF;
    c32* wh = lds + C::OFF_WH;
    c32* wl = lds + C::OFF_WL;
    unsigned* ppA = reinterpret_cast<unsigned*>(lds + C::OFF_PA);
    unsigned* ppB = reinterpret_cast<unsigned*>(lds + C::OFF_PB);
    const size_t tile_elems = (size_t)C::TILE_ROWS * T;

    // XCD-aware tile order (see fast_cols.hpp): each XCD walks a contiguous run of tiles
    const int per_xcd = nwg / 8;
    const int wg_x = (nwg % 8 == 0) ? (wg % 8) * per_xcd + wg / 8 : wg;

    auto tile_base = [&](int tile) -> const c32* {
        const int kernel = tile / g.tiles_per_kernel;
        return g.Y + (size_t)kernel * g.y_kernel_stride + (size_t)(tile - kernel * g.tiles_per_kernel) * tile_elems;
    };
    // half 0: rows [0, H] (H = Nyquist), half 1: rows [H+1, 2H+1)
    auto issue_gather = [&](int t, State& st, int tile, int half) {
        const c32* Yh = tile_base(tile) + (half ? (size_t)(H + 1) * T : 0);
        static_for<0, C::UPT>([&](auto r_) {
            constexpr int r = decltype(r_)::value;
            if (t + NT * r < C::NUNIT) st.pre[r] = *reinterpret_cast<const c32x2*>(Yh + 2 * (t + NT * r));
        });
        if (!half && t < T2) st.pre_ny = *reinterpret_cast<const c32x2*>(Yh + (size_t)H * T + 2 * t);
    };
    auto land_gather = [&](int t, State& st, int half) {
        static_for<0, C::UPT>([&](auto r_) {
            constexpr int r = decltype(r_)::value;
            const int e = t + NT * r;
            const int u = e / T2, t2 = e % T2;
            if (e < C::NUNIT) {
                lds[(2 * t2) * LP + u] = st.pre[r].a;
                lds[(2 * t2 + 1) * LP + u] = st.pre[r].b;
            }
        });
        if (!half && t < T2) {
            lds[(2 * t) * LP + H] = st.pre_ny.a;
            lds[(2 * t + 1) * LP + H] = st.pre_ny.b;
        }
    };
    // merge the half spectrum inside one half: item i <-> bin k = 2i + half
    auto pair_pass = [&](int t, int half) {
        const unsigned* pp = half ? ppB : ppA;
        const int nitems = half ? C::NPB : C::NPA;
        FC_NOUNROLL
        for (int r = 0; r < C::RNDP; r++) {
            const int idx = t + NT * r;
            if (idx < nitems * T) {
                const int i = idx / T, col = idx % T;
                c32* z = lds + col * LP;
                const unsigned p = pp[i];
                const int pa = (int)(p & 0xffffu), pb = (int)(p >> 16);
                if (!half && i == 0) {               // DC + Nyquist
                    float x0 = z[pa].x, xm = z[pb].x;
                    z[pa] = mk(x0 + xm, x0 - xm);
                } else if (pa == pb) {               // self-paired middle bin
                    c32 x = z[pa];
                    z[pa] = mk(2.f * x.x, -2.f * x.y);
                } else {
                    const int k = 2 * i + half;
                    const c32 w = cmul(wh[k >> 5], wl[k & 31]);
                    c32 xk = z[pa], xm = z[pb];
                    c32 Ssum = mk(xk.x + xm.x, xk.y - xm.y);
                    c32 D = mk(xk.x - xm.x, xk.y + xm.y);
                    c32 G = cmulc(D, w);
                    z[pa] = mk(Ssum.x - G.y, Ssum.y + G.x);
                    z[pb] = mk(Ssum.x + G.y, -Ssum.y + G.x);
                }
            }
        }
    };
    auto stage_r4 = [&](int t) {   // radix R4 on contiguous runs
        FC_NOUNROLL
        for (int r4 = 0; r4 < C::RND4; r4++) {
        const int bi = t + NT * r4;
        if (bi >= C::NB4 * T) break;
        const int col = bi / C::NB4, q = bi % C::NB4;
        c32* p = lds + col * LP + q * R4;
        c32 v[R4];
        if constexpr (R4 % 2 == 0) {   // runs are 16-byte aligned: wide LDS accesses
            static_for<0, R4 / 2>([&](auto h_) {
                constexpr int h = decltype(h_)::value;
                c32x2 w = *reinterpret_cast<const c32x2*>(p + 2 * h);
                v[2 * h] = w.a;
                v[2 * h + 1] = w.b;
            });
            Dft<R4, +1>::run(v);
            static_for<0, R4 / 2>([&](auto h_) {
                constexpr int h = decltype(h_)::value;
                c32x2 w;
                w.a = v[2 * h];
                w.b = v[2 * h + 1];
                *reinterpret_cast<c32x2*>(p + 2 * h) = w;
            });
        } else {                       // odd radix: 8-byte accesses, lane stride R4*8 B is conflict-free
            static_for<0, R4>([&](auto a_) { v[decltype(a_)::value] = p[decltype(a_)::value]; });
            Dft<R4, +1>::run(v);
            static_for<0, R4>([&](auto a_) { p[decltype(a_)::value] = v[decltype(a_)::value]; });
        }
        }
    };
    auto stage_r3 = [&](int t) {   // radix R3, sub-length R4, blocks of m2
        FC_NOUNROLL
        for (int r = 0; r < C::RND3; r++) {   // one butterfly at a time: registers belong to ua / the gather
            const int idx = t + NT * r;
            if (idx < C::NB3 * T) {
                const int col = idx / C::NB3, u = idx % C::NB3;
                const int c2 = u / R4, b = u % R4;
                c32* p = lds + col * LP + c2 * m2 + b;
                c32 v[R3];
                v[0] = p[0];
                static_for<1, R3>([&](auto c_) {
                    constexpr int c = decltype(c_)::value;
                    v[c] = cmulc(p[c * R4], tw3[(c - 1) * R4 + b]);
                });
                Dft<R3, +1>::run(v);
                static_for<0, R3>([&](auto a_) {
                    constexpr int a = decltype(a_)::value;
                    p[a * R4] = v[a];
                });
            }
        }
    };
    // radix R2 stage (sub-length m2) of butterfly (col, j'): result u[a] is element j' + a*m2 of the half
    auto stage_r2 = [&](int col, int jp, c32 (&v)[R2]) {
        const c32* p = lds + col * LP + jp;
        c32 pw[R2];
        power_chain<R2>(twA[jp], pw);
        v[0] = p[0];
        static_for<1, R2>([&](auto c_) {
            constexpr int c = decltype(c_)::value;
            v[c] = cmulc(p[c * m2], pw[c]);
        });
        Dft<R2, +1>::run(v);
    };

    // prologue: tables into LDS, first half in flight
    const int first_tile = wg_x;
    ctx.phase([&](int t, State& st) {
        for (int i = t; i < C::T3N; i += NT) tw3[i] = g.tw3[i];
        for (int i = t; i < m2; i += NT) { twA[i] = g.twA[i]; twF[i] = g.twF[i]; }
        for (int i = t; i < C::NWH; i += NT) wh[i] = g.wh[i];
        for (int i = t; i < 32; i += NT) wl[i] = g.wl[i];
        for (int i = t; i < C::NPA; i += NT) ppA[i] = g.ppA[i];
        for (int i = t; i < C::NPB; i += NT) ppB[i] = g.ppB[i];
        if (first_tile < g.ntiles) issue_gather(t, st, first_tile, 0);
    });

    for (int tile = first_tile; tile < g.ntiles; tile += nwg) {
        const int kernel = tile / g.tiles_per_kernel;
        const int w0 = (tile - kernel * g.tiles_per_kernel) * T;
        const int next = tile + nwg;

        // ---- half A (even bins)
        ctx.phase([&](int t, State& st) { land_gather(t, st, 0); });
        ctx.phase([&](int t, State& st) {
            issue_gather(t, st, tile, 1);    // half B of this tile flies during the whole of half A
            pair_pass(t, 0);
        });
        ctx.phase([&](int t, State&) { stage_r4(t); });
        ctx.phase([&](int t, State&) { stage_r3(t); });
        ctx.phase([&](int t, State& st) {    // results stay in registers; barrier = LDS free for half B
            static_for<0, C::RND2>([&](auto r_) {
                constexpr int r = decltype(r_)::value;
                const int idx = t + NT * r;
                if (idx < C::NB2 * T) {
                    c32 v[R2];
                    stage_r2(idx / C::NB2, idx % C::NB2, v);
                    static_for<0, R2>([&](auto a_) { st.ua[r][decltype(a_)::value] = v[decltype(a_)::value]; });
                }
                FC_SCHED_FENCE();
            });
        });

        // ---- half B (odd bins)
        ctx.phase([&](int t, State& st) { land_gather(t, st, 1); });
        ctx.phase([&](int t, State&) { pair_pass(t, 1); });
        ctx.phase([&](int t, State&) { stage_r4(t); });
        ctx.phase([&](int t, State& st) {
            if (next < g.ntiles) issue_gather(t, st, next, 0);   // after the register-hungry radix-R4 stage
            stage_r3(t);
        });
        float* out = g.out + (size_t)kernel * g.out_kernel_stride;
        const int nout = g.fft_h >> 1;
        ctx.phase([&](int t, State& st) {    // last stages + final radix-2 butterfly + store
            static_for<0, C::RND2>([&](auto r_) {
                constexpr int r = decltype(r_)::value;
                const int idx = t + NT * r;
                if (idx < C::NB2 * T) {
                    const int col = idx / C::NB2, jp = idx % C::NB2;
                    c32 v[R2];
                    stage_r2(col, jp, v);
                    const c32 wf = twF[jp];   // w_M^j'
                    // 32-bit element offsets from the (uniform) map base: one VGPR per address
                    // instead of a 64-bit pair for each of the 2*R2*RND2 stores
                    c32* omap = reinterpret_cast<c32*>(out);
                    const unsigned off0 = (unsigned)(w0 + col) * (unsigned)nout + (unsigned)jp;
                    static_for<0, R2>([&](auto a_) {
                        constexpr int a = decltype(a_)::value;
                        // conj(w_M^(j' + a*m2)) = conj(w_M^j') * exp(+2 pi i a / (2*R2))
                        const c32 vb = mul_root<2 * R2, a, +1>(cmulc(v[a], wf));
                        const c32 va = st.ua[r][a];
                        const int j = jp + a * m2;
                        if (j < nout) FC_STREAM_STORE(&omap[off0 + (unsigned)(a * m2)], va + vb);
                        if (j + H < nout) FC_STREAM_STORE(&omap[off0 + (unsigned)(a * m2 + H)], va - vb);
                    });
                }
                FC_SCHED_FENCE();
            });
        });
    }
}

}  // namespace fc
