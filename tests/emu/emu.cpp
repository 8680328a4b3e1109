// emu.cpp -- TEST-ONLY sequential executor of the engine's workgroup bodies on the host.
//
// Compiles the same headers the HIP kernels are built from (kernels_body.hpp, fft_lds.hpp,
// butterflies.hpp, planner.hpp, pipeline.hpp) with g++ and runs every workgroup one after the
// other with a single-thread context, so the algorithm (radix butterflies, digit-reversed
// in-place transforms, pair tables, layouts, argument blocks) can be checked against the oracle
// in the CPU-only test tier.  It is NOT part of the product: libfftconv.so has no CPU path and
// never links or loads this file.
#include <cstdio>
#include <cstring>
#include <vector>

#include "emu_runners.hpp"

using namespace fc;
using namespace emu;

namespace {
PlanTuning g_tune;   // path mode / rows group of the emulated plans (pipeline.hpp PlanTuning)
// the output window of a block of an overlap-save block-wise plan (fftconv_api.cpp: OutWindow), applied to the output kernel
struct { bool on = false; int h_lo = 0, h_hi = 0, w_first = 0, ncols = 0, pitch = 0; } g_win;
// dynamic tile queue of the persistent column kernels (plan option "dynamic_tiles"): the emulated "workgroups" run one after
// the other, so the first takes every tile of its home counter and then those of the others -- the ticket -> tile map, the
// hand-over through the LDS slot and the termination are what this exercises
bool g_dynamic_tiles = false;
int g_queue[FC_QUEUE_WORDS];
}  // namespace

extern "C" {

// Size (in complex elements) of the image spectrum buffer of a plan.
long emu_spectrum_elems(int H, int W, int F, int max_kh, int max_kw) {
    Geometry g;
    Tables t;
    if (!make_geometry(g, t, H, W, F, max_kh, max_kw, g_tune)) return -1;
    return (long)g.spectrum_elems();
}

// Image spectrum (what fftconv_plan_set_image leaves in the plan's spectrum buffer).
int emu_image_spectrum(const float* data, int H, int W, int F, int max_kh, int max_kw, float* spec_out) {
    Geometry g;
    Tables t;
    if (!make_geometry(g, t, H, W, F, max_kh, max_kw, g_tune)) return -1;
    DeviceTables d;
    d.queue = g_dynamic_tiles ? g_queue : nullptr;
    d.queue_fwd = g_dynamic_tiles;      // (the emulator exercises the forward kernels' queue too: plan option dynamic_tiles = 2)
    d.tw_m = t.pm.tw.data();
    d.tw_w = t.pw.tw.data();
    d.pairs = t.pairs.data();
    HostCtx ctx;
    std::vector<c32> lds(FC_LDS_BUDGET / sizeof(c32));
    c32* S = reinterpret_cast<c32*>(spec_out);
    // garbage-fill to catch reads of never-written cells
    for (size_t i = 0; i < g.spectrum_elems(); i++) S[i] = mk(1e30f, -1e30f);
    if (g.fast_fwd) {
        d.fc_tw1 = t.fcl.tw1.data(); d.fc_tw2 = t.fcl.tw2.data(); d.fc_pairs = t.fcl.pairs.data();
        FastColsFwdArgs fa = fast_cols_fwd_args(g, d, data, (size_t)H * W, H, H, W, F, S, (size_t)g.rows * g.s_pitch, g.s_pitch, false);
        EmuFastColsFwd run{fa, lds.data(), 3};
        if (!run_fast_cols_fwd(g.M, g.fast_cols.T, false, run)) return -8;
    } else {
        ColsR2CArgs ia = image_cols_args(g, t, d, data, S);
        for (int plane = 0; plane < F; plane++)
            for (int tile = 0; tile < tiles_for(W, g.T_cols); tile++) cols_r2c_body(ctx, lds.data(), ia, tile, plane);
    }
    if (g.fast_rows.ok) {   // specialised forward rows: stores in the fast row kernel's register order
        d.fr_tw1 = t.fr.tw1.data();
        d.fr_tw2 = t.fr.tw2.data();
        FastRowsFwdArgs fa = fast_rows_fwd_args(g, d, S);
        EmuFastRowsFwd run{fa, lds.data(), F * g.rows};
        if (!run_fast_rows_fwd(g.Lw, run)) return -9;
        return 0;
    }
    RowsFwdArgs ra = image_rows_args(g, t, d, S);
    for (int r = 0; r < F * g.rows; r++) rows_fwd_body(ctx, lds.data(), ra, r);
    return 0;
}

// Per-kernel loop from a given spectrum (what fftconv_plan_convolve does).
int emu_convolve_spectrum(const float* spec, int H, int W, int F, int max_kh, int max_kw, int n_kernel,
                          const float* const* kernels, const int* kh, const int* kw, float* const* out) {
    Geometry g;
    Tables t;
    if (!make_geometry(g, t, H, W, F, max_kh, max_kw, g_tune)) return -1;
    DeviceTables d;
    d.queue = g_dynamic_tiles ? g_queue : nullptr;
    d.queue_fwd = g_dynamic_tiles;      // (the emulator exercises the forward kernels' queue too: plan option dynamic_tiles = 2)
    d.tw_m = t.pm.tw.data();
    d.tw_w = t.pw.tw.data();
    d.pairs = t.pairs.data();
    HostCtx ctx;
    std::vector<c32> lds(FC_LDS_BUDGET / sizeof(c32));
    const c32* S = reinterpret_cast<const c32*>(spec);
    std::vector<c32> Y(g.y_elems_per_kernel());
    for (int k = 0; k < n_kernel; k++) {
        if (kh[k] < 1 || kw[k] < 1 || kh[k] > g.fft_h || kw[k] > g.fft_w) return -2;
        if ((kh[k] > max_kh || kw[k] > max_kw) && !(g.exact_window && kh[k] <= g.Lh && kw[k] <= g.Lw)) return -3;
        std::vector<c32> A((size_t)F * g.rows * a_pitch_for(kw[k]));
        for (auto& v : A) v = mk(1e30f, -1e30f);
        for (auto& v : Y) v = mk(1e30f, -1e30f);
        if (g.fast_fwd) {
            d.fc_tw1 = t.fcl.tw1.data(); d.fc_tw2 = t.fcl.tw2.data(); d.fc_pairs = t.fcl.pairs.data();
            FastColsFwdArgs fa = fast_cols_fwd_args(g, d, kernels[k], (size_t)kh[k] * kw[k], kh[k], kh[k], kw[k], F, A.data(),
                                                    (size_t)g.rows * a_pitch_for(kw[k]), a_pitch_for(kw[k]), true);
            EmuFastColsFwd run{fa, lds.data(), 2};
            if (!run_fast_cols_fwd(g.M, g.fast_cols.T, fast_cols_fwd_pruned_ok(g.fast_cols, kh[k]), run)) return -8;
        } else {
            ColsR2CArgs ka = kernel_cols_args(g, t, d, kernels[k], kh[k], kw[k], A.data());
            for (int plane = 0; plane < F; plane++)
                for (int tile = 0; tile < tiles_for(kw[k], g.T_cols); tile++) cols_r2c_body(ctx, lds.data(), ka, tile, plane);
        }
        if (g.fast_cols.ok) d.fc_pair_row_of = t.fcl.pair_row_of.data();
        if (g.fast_rows.ok) {
            if (kw[k] > g.fast_rows.max_kw) return -4;
            d.fr_tw1 = t.fr.tw1.data();
            d.fr_tw2 = t.fr.tw2.data();
            FastRowsArgs fa = fast_rows_args(g, d, A.data(), kw[k], S, Y.data());
            EmuFastRows run{fa, lds.data(), g.rows, (g.rows_multi_ok() && g.rows_group > 1) ? g.rows_group : 0};
            if (!run_fast_rows(g.Lw, fast_rows_nz2(g, kw[k]), run)) return -5;
        } else {
            SpectralRowsArgs sa = spectral_rows_args(g, t, d, A.data(), kw[k], S, Y.data());
            for (int r = 0; r < g.rows; r++) spectral_rows_body(ctx, lds.data(), sa, r, 0);
        }
        if (g.fast_cols.ok) {
            d.fc_tw1 = t.fcl.tw1.data();
            d.fc_tw2 = t.fcl.tw2.data();
            d.fc_pairs = t.fcl.pairs.data();
            d.fc_rowoff = t.fcl.rowoff.data();
            FastColsArgs fa = fast_cols_args(g, d, Y.data(), out[k], 0, 1);
            if (g_win.on) {
                fa.h_lo = g_win.h_lo; fa.fft_h = g_win.h_hi; fa.w_first = g_win.w_first; fa.out_pitch = g_win.pitch;
                fa.tiles_per_kernel = g_win.ncols / g.fast_cols.T; fa.ntiles = fa.tiles_per_kernel;
            }
            EmuFastCols run{fa, lds.data(), 3};   // 3 persistent workgroups share the tiles
            if (!run_fast_cols(g.M, g.fast_cols.T, run)) return -6;
        } else {
            if (g_win.on) return -7;
            ColsC2RArgs ca = cols_c2r_args(g, t, d, Y.data(), out[k], 0);
            for (int tile = 0; tile < tiles_for(g.fft_w, g.T_cols); tile++) cols_c2r_body(ctx, lds.data(), ca, tile, 0);
        }
    }
    return 0;
}

// Same contract as oracle_conv_fft (and as fftconv_convolution_fft): sizes may differ per
// kernel.  Returns 0 on success.
int emu_conv_fft(const float* data, int H, int W, int F, int max_kh, int max_kw, int n_kernel,
                 const float* const* kernels, const int* kh, const int* kw, float* const* out,
                 int* lh_out, int* lw_out) {
    Geometry g;
    Tables t;
    if (!make_geometry(g, t, H, W, F, max_kh, max_kw, g_tune)) return -1;
    if (lh_out) *lh_out = g.Lh;
    if (lw_out) *lw_out = g.Lw;
    std::vector<float> spec(2 * g.spectrum_elems());
    if (int rc = emu_image_spectrum(data, H, W, F, max_kh, max_kw, spec.data())) return rc;
    return emu_convolve_spectrum(spec.data(), H, W, F, max_kh, max_kw, n_kernel, kernels, kh, kw, out);
}

// 1-D self checks used by tests: forward transform of x (length L) -> natural-order spectrum.
int emu_fft1d(int L, const float* xin /* 2L floats */, float* xout /* 2L floats */, int inverse) {
    if (!length_supported(L)) return -1;
    Plan1D p = make_plan1d(L);
    HostCtx ctx;
    std::vector<c32> buf(L);
    if (!inverse) {
        for (int i = 0; i < L; i++) buf[i] = mk(xin[2 * i], xin[2 * i + 1]);
        fft_forward(ctx, buf.data(), L, 1, p.desc, p.tw.data());
        for (int k = 0; k < L; k++) { xout[2 * k] = buf[p.pos[k]].x; xout[2 * k + 1] = buf[p.pos[k]].y; }
    } else {
        for (int k = 0; k < L; k++) buf[p.pos[k]] = mk(xin[2 * k], xin[2 * k + 1]);
        fft_inverse(ctx, buf.data(), L, 1, p.desc, p.tw.data());
        for (int i = 0; i < L; i++) { xout[2 * i] = buf[i].x; xout[2 * i + 1] = buf[i].y; }
    }
    return 0;
}

// path mode (0 generic kernels only, 1 specialised + row-major intermediate, 2 default) and maps per
// workgroup of the multi-map row kernel (-1 auto) of the plans emulated from here on
void emu_set_tuning(int path_mode, int rows_group) { g_tune.path_mode = path_mode; g_tune.rows_group = rows_group; }
// 1: the emulated plans transform the ceil16 window itself (fftconv_plan_options.exact_window)
void emu_set_exact_window(int on) { g_tune.exact_window = on != 0; }
void emu_set_dynamic_tiles(int on) { g_dynamic_tiles = on != 0; }
void emu_allow_fast(int mode) { g_tune.path_mode = mode; }
// 1: the emulated plans are block plans of an overlap-save block-wise plan: H x W is the transform, the result circular
void emu_set_cyclic(int on) { g_tune.cyclic = on != 0; }
// the window the output kernel stores (on = 0: the whole window at out[k]): rows [h_lo, h_hi) of columns [w_first, w_first + ncols)
// of the result, row h of column w at out[k] + w * pitch + h
void emu_set_out_window(int on, int h_lo, int h_hi, int w_first, int ncols, int pitch) {
    g_win.on = on != 0; g_win.h_lo = h_lo; g_win.h_hi = h_hi; g_win.w_first = w_first; g_win.ncols = ncols; g_win.pitch = pitch;
}
// 1 if a plan of these sizes would use the fast spectral-row kernel
int emu_uses_fast_rows(int H, int W, int F, int max_kh, int max_kw) {
    Geometry g;
    Tables t;
    if (!make_geometry(g, t, H, W, F, max_kh, max_kw, g_tune)) return -1;
    return (g.fast_rows.ok ? 1 : 0) | (g.fast_cols.ok ? 2 : 0);
}

// the planner alone (no preference for lengths with specialised kernels)
int emu_choose_length(int need, int real_half, int exact) {
    return choose_length(need, real_half != 0, exact);
}
// transform lengths a plan of these sizes would use (path mode as set by emu_allow_fast)
int emu_plan_lengths(int H, int W, int F, int max_kh, int max_kw, int* lh, int* lw) {
    Geometry g;
    Tables t;
    if (!make_geometry(g, t, H, W, F, max_kh, max_kw, g_tune)) return -1;
    *lh = g.Lh;
    *lw = g.Lw;
    return 0;
}
int emu_length_supported(int L) { return length_supported(L) ? 1 : 0; }

}  // extern "C"
