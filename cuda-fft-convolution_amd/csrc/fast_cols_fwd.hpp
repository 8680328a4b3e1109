// fast_cols_fwd.hpp -- specialised forward column kernel: real -> half-complex transform along h
// of T columns per tile, for the image (once per image) and for the kernels (the only part of
// the per-kernel work that does not depend on the image).  Replaces padData + the H half of
// cufftExecR2C (src/cudaConvFFTData.cuh:11-31, src/cudaConvolutionFFT.cu:155-167, :245-255).
//
// Same configurations, LDS image and tables as the output kernel (fast_cols.hpp: ColCfg, M =
// R1*R2*R3 complex points for real length 2M), run forwards: decimation in frequency, natural
// order in, the plan's digit-reversed order out -- so the spectrum rows it emits are directly in
// the LDS order of the output kernel (row p = LDS position p, row M = Nyquist).
//   * zero padding is never read: samples beyond h_in are zeros by construction;
//   * PRUNED variant (kernels, h_in <= 2*min(m1, NZ2*R3)): stage 1 has a single non-zero input per
//     butterfly (a twiddle scaling) and stage 2 reads only its first NZ2 inputs;
//   * persistent workgroups (tables loaded once), no run-time integer division.
#pragma once
#include "butterflies.hpp"
#include "fast_cols.hpp"
#include "fc_common.hpp"

namespace fc {

struct FastColsFwdArgs {
    const float* in;         // plane q at in + q*in_plane_stride; column c at + c*in_col_pitch, h contiguous
    size_t in_plane_stride;
    int in_col_pitch;
    int h_in;                // valid samples per column
    int ncols;               // columns per plane
    c32* out;                // plane q at out + q*out_plane_stride; row p in [0, M]: [p*out_pitch + c]
    size_t out_plane_stride;
    int out_pitch;
    int tiles_per_plane;     // ceil(ncols / T)
    int ntiles;              // tiles_per_plane * planes
    const c32* tw1;          // w_M^j, j < m1
    const c32* tw2;          // stage-2 table [(c-1)*R3 + b]
    const PairEntry* pairs;  // NPE entries (w = w_N^k), positions in this plan's order
    int* queue;              // dynamic tile queue (fast_cols.hpp): ONE counter, zero at launch (the last workgroup out zeroes it again)
                             // -- the tiles are taken in order, one ahead; nullptr: workgroup b owns tiles b, b + grid, ...
};

struct ColFwdState {
    int tk;                  // thread 0, dynamic tile queue: the ticket requested at the start of the tile
};

// NZ2 < R2: pruned variant (see header); NZ2 == R2: any h_in <= 2M
template <class C, int NZ2, class Ctx>
FC_HD void fast_cols_fwd_body(Ctx& ctx, c32* lds, const FastColsFwdArgs& g, int wg, int nwg) {
    constexpr int M = C::M, R1 = C::R1, R2 = C::R2, R3 = C::R3, T = C::T, NT = C::NT, LP = C::LP, m1 = C::m1;
    constexpr bool PRUNED = NZ2 < R2;
    using State = ColFwdState;
    c32* tw2 = lds + C::OFF_T2;
    c32* tw1 = lds + C::OFF_T1;
    c32* wh = lds + C::OFF_WH;
    c32* wl = lds + C::OFF_WL;
    unsigned* ppos = reinterpret_cast<unsigned*>(lds + C::OFF_PAIR);
    const int nz = (g.h_in + 1) / 2;   // non-zero packed complex samples per column
    // dynamic tile queue (fast_cols.hpp): qs[0] = the tile after this one (requested at the start of a tile by thread 0, written in
    // F4, read by everybody behind F5's barrier), qs[1] = the first tile
    const bool dyn = g.queue != nullptr;
    int* qs = reinterpret_cast<int*>(lds + C::OFF_QUEUE);

    ctx.phase([&](int t, State&) {
        for (int i = t; i < C::T2N; i += NT) tw2[i] = g.tw2[i];
        for (int i = t; i < m1; i += NT) tw1[i] = g.tw1[i];
        for (int i = t; i < C::NPE; i += NT) {
            const PairEntry e = g.pairs[i];
            ppos[i] = (unsigned)C::cell(e.a) | ((unsigned)C::cell(e.b) << 16);   // LDS cells (fast_cols.hpp: ColCfg::cell)
            if ((i & 31) == 0) wh[i >> 5] = e.w;
            if (i < 32) wl[i] = e.w;
        }
        if (dyn && t == 0) {       // the first two tiles of this workgroup
            const int k0 = FC_QUEUE_TAKE(g.queue), k1 = FC_QUEUE_TAKE(g.queue);
            qs[1] = k0 < k1 ? k0 : k1;
            qs[0] = k0 < k1 ? k1 : k0;
        }
    });

    int dyn_next = dyn ? FC_UNIFORM(qs[0]) : 0;
    const int first_tile = dyn ? FC_UNIFORM(qs[1]) : wg;
    // (everybody has read the two slots before thread 0 may write them again: a workgroup without a tile runs straight into
    //  the second body of the image + kernels launch, whose prologue fills them anew)
    if (dyn) ctx.phase([&](int, State&) {});
    for (int tile = first_tile; tile < g.ntiles;) {
        const int plane = tile / g.tiles_per_plane;
        const int c0 = (tile - plane * g.tiles_per_plane) * T;
        const float* in = g.in + (size_t)plane * g.in_plane_stride;
        c32* out = g.out + (size_t)plane * g.out_plane_stride;

        // F1: packed load z[n] = x[2n] + i x[2n+1] fused with stage 1 (radix R1, sub-length m1)
        ctx.phase([&](int t, [[maybe_unused]] State& st) {
            if (dyn && t == 0 && dyn_next < g.ntiles) st.tk = FC_QUEUE_TAKE(g.queue);
            auto sample = [&](const float* col, int n) -> c32 {
                float x0 = (2 * n < g.h_in) ? col[2 * n] : 0.f;
                float x1 = (2 * n + 1 < g.h_in) ? col[2 * n + 1] : 0.f;
                return mk(x0, x1);
            };
            if constexpr (PRUNED) {   // one non-zero input per butterfly: outputs z[j] * w_M^(j c)
                for (int idx = t; idx < T * nz; idx += NT) {
                    const int col = idx / nz, j = idx - col * nz;
                    if (c0 + col < g.ncols) {
                        const c32 z = sample(in + (size_t)(c0 + col) * g.in_col_pitch, j);
                        c32 p[R1];
                        power_chain<R1>(tw1[j], p);
                        c32* q = lds + col * LP + j;
                        q[0] = z;
                        static_for<1, R1>([&](auto c_) {
                            constexpr int c = decltype(c_)::value;
                            q[c * C::S1] = cmul(z, p[c]);
                        });
                    }
                }
            } else {
                FC_NOUNROLL
                for (int r = 0; r < C::RND1; r++) {
                    const int idx = t + NT * r;
                    if (idx < C::NB1 * T) {
                        const int col = idx / C::NB1, j = idx % C::NB1;
                        c32* q = lds + col * LP + j;
                        c32 v[R1];
                        if (c0 + col < g.ncols) {
                            const float* colp = in + (size_t)(c0 + col) * g.in_col_pitch;
                            static_for<0, R1>([&](auto a_) {
                                constexpr int a = decltype(a_)::value;
                                v[a] = (j + a * m1 < nz) ? sample(colp, j + a * m1) : mk(0.f, 0.f);
                            });
                        } else {
                            static_for<0, R1>([&](auto a_) { v[decltype(a_)::value] = mk(0.f, 0.f); });
                        }
                        Dft<R1, -1>::run(v);
                        c32 p[R1];
                        power_chain<R1>(tw1[j], p);
                        q[0] = v[0];
                        static_for<1, R1>([&](auto c_) {
                            constexpr int c = decltype(c_)::value;
                            q[c * C::S1] = cmul(v[c], p[c]);
                        });
                    }
                }
            }
        });

        // F2: stage 2 (radix R2, sub-length R3); pruned: inputs a >= NZ2 (and beyond nz) are zeros
        ctx.phase([&](int t, State&) {
            static_for<0, C::RND2>([&](auto r_) {
                constexpr int r = decltype(r_)::value;
                const int idx = t + NT * r;
                if (idx < C::NB2 * T) {
                    const int col = idx / C::NB2, u = idx % C::NB2;
                    const int c1 = u / R3, b = u % R3;
                    c32* p = lds + col * LP + c1 * C::S1 + b;
                    c32 v[R2];
                    static_for<0, R2>([&](auto a_) {
                        constexpr int a = decltype(a_)::value;
                        if constexpr (PRUNED) {
                            if constexpr (a < NZ2) v[a] = (a * R3 + b < nz && c0 + col < g.ncols) ? p[a * R3] : mk(0.f, 0.f);
                            else v[a] = mk(0.f, 0.f);
                        } else {
                            v[a] = p[a * R3];
                        }
                    });
                    if constexpr (PRUNED) Dft<R2, -1>::template run_nz<NZ2>(v);
                    else Dft<R2, -1>::run(v);
                    p[0] = v[0];
                    static_for<1, R2>([&](auto c_) {
                        constexpr int c = decltype(c_)::value;
                        p[c * R3] = cmul(v[c], tw2[(c - 1) * R3 + b]);
                    });
                }
            });
        });

        // F3: stage 3 (radix R3 on contiguous runs), one butterfly per thread
        ctx.phase([&](int t, State&) {
            const int col = t / C::NB3, q = t % C::NB3;
            c32* p = lds + col * LP + C::run_of_thread(q);
            c32 v[R3];
            static_for<0, R3 / 2>([&](auto h_) {
                constexpr int h = decltype(h_)::value;
                c32x2 w = *reinterpret_cast<const c32x2*>(p + 2 * h);
                v[2 * h] = w.a;
                v[2 * h + 1] = w.b;
            });
            Dft<R3, -1>::run(v);
            static_for<0, R3 / 2>([&](auto h_) {
                constexpr int h = decltype(h_)::value;
                c32x2 w;
                w.a = v[2 * h];
                w.b = v[2 * h + 1];
                *reinterpret_cast<c32x2*>(p + 2 * h) = w;
            });
        });

        // F4: split the packed transform into the spectrum of the real columns, in place
        ctx.phase([&](int t, [[maybe_unused]] State& st) {
            if (dyn && t == 0 && dyn_next < g.ntiles) qs[0] = st.tk;
            FC_NOUNROLL
            for (int r = 0; r < C::RNDP; r++) {
                const int idx = t + NT * r;
                if (idx < C::NPAIR * T) {
                    const int k = idx / T + 1, col = idx % T;
                    c32* z = lds + col * LP;
                    const unsigned pp = ppos[k];
                    const int pa = (int)(pp & 0xffffu), pb = (int)(pp >> 16);
                    const c32 w = cmul(wh[k >> 5], wl[k & 31]);
                    const c32 zk = z[pa], zm = z[pb];
                    const c32 E = mk(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
                    const c32 D = mk(0.5f * (zk.x - zm.x), 0.5f * (zk.y + zm.y));
                    const c32 G = cmul(w, D);
                    z[pa] = mk(E.x + G.y, E.y - G.x);
                    z[pb] = mk(E.x - G.y, -E.y - G.x);
                }
            }
            if (t < T) {                    // DC / Nyquist
                c32* z = lds + t * LP;
                const unsigned pp = ppos[0];
                const int pa = (int)(pp & 0xffffu), pb = (int)(pp >> 16);
                const c32 z0 = z[pa];
                z[pa] = mk(z0.x + z0.y, 0.f);
                z[pb] = mk(z0.x - z0.y, 0.f);
            } else if (t < 2 * T) {         // middle bin
                c32* z = lds + (t - T) * LP;
                const int pa = (int)(ppos[M / 2] & 0xffffu);
                z[pa] = conj(z[pa]);
            }
        });

        // F5: store rows 0..M (row = LDS position), columns fastest (T*8-byte pieces); the closing
        // barrier protects the LDS image against the next tile's stage 1
        ctx.phase([&](int t, State&) {
            for (int idx = t; idx < (M + 1) * T; idx += NT) {
                const int p = idx / T, col = idx % T;
                if (c0 + col < g.ncols) out[(size_t)p * g.out_pitch + c0 + col] = lds[col * LP + C::cell(p)];
            }
        });
        if (dyn) {
            tile = dyn_next;
            dyn_next = dyn_next < g.ntiles ? FC_UNIFORM(qs[0]) : g.ntiles;
        } else {
            tile += nwg;
        }
    }
    if (dyn) ctx.phase_nosync([&](int t, State&) { if (t == 0) queue_leave(g.queue, 1, nwg); });
}

}  // namespace fc
