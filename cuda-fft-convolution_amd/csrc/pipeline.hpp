// pipeline.hpp -- geometry of one plan and the argument blocks of its kernels.
// Host-only, shared by the product library (fftconv_api.cpp) and the test-only emulator so
// that both drive the workgroup bodies with identical arguments.
#pragma once
#include <algorithm>
#include <cstddef>

#include "fast_paths.hpp"
#include "kernels_body.hpp"
#include "planner.hpp"

namespace fc {

constexpr size_t FC_LDS_BUDGET = 160 * 1024;  // bytes of LDS one workgroup may claim (gfx950 CU)

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// Choices fixed at plan creation (include/fftconv.h: fftconv_plan_options).  The product reads no
// environment variables; tests and A/B runs pass these through fftconv_plan_create_ex.
struct PlanTuning {
    // 0 generic kernels only; 1 specialised kernels with a row-major intermediate; 2 (default)
    // specialised kernels with the tiled, pair-adjacent intermediate
    int path_mode = 2;
    // maps per workgroup of the multi-map row kernel (fast_rows_multi.hpp): -1 = chosen per
    // launch (rows_group_for), 0 / 1 = plain one-map kernel, > 1 = fixed
    int rows_group = -1;
    // largest transform length a plan may use (0 = whatever fits the LDS); larger problems are left
    // to the block-wise path of the one-shot entry
    int max_transform = 0;
    // transform lengths must equal the ceil16 window (the reference's circular modulus; needed to
    // exchange spectra in the reference's order); unsupported windows then fail
    bool exact_window = false;
    // the block plan of an overlap-save block-wise plan (fftconv_api.cpp): H x W IS the transform (a block of the image with
    // its history rows / columns), the result is the circular convolution modulo H x W, and the caller stores the part of
    // it that is not wrapped.  Needs both specialised hot kernels and the tiled intermediate.
    bool cyclic = false;
};

struct Geometry {
    // problem (src/cudaConvolutionFFT.cu:92-112)
    int H = 0, W = 0, F = 0;
    int max_kh = 0, max_kw = 0;
    int fft_h = 0, fft_w = 0;  // output window: computeFFTsize16(DATA + MAXK - 1)
    // internal transform
    int Lh = 0, Lw = 0;        // transform lengths (Lh even)
    int M = 0;                 // Lh / 2
    int rows = 0;              // M + 1 spectrum rows
    int s_pitch = 0;           // c32 per image-spectrum row
    int y_pitch = 0;           // c32 per intermediate row
    int wout = 0;              // columns of Y consumed by the output pass: min(Lw, fft_w)
    int lds_pitch = 0;         // c32 per LDS-resident column
    int T_cols = 1;            // columns per workgroup in the h-passes
    bool exact_window = false; // Lh == fft_h && Lw == fft_w: circular modulus equals the reference's
    FastRowsInfo fast_rows;    // specialised spectral-row kernel, if one exists for (Lw, max_kw)
    FastColsInfo fast_cols;    // specialised output kernel, if one exists for M
    bool fast_fwd = false;     // forward column transforms (image, kernels) by fast_cols_fwd.hpp: the
                               // spectrum rows are then in the fast plan's order, not the generic plan's
    int path_mode = 2;         // PlanTuning::path_mode
    // tiled intermediate [w / 16][row][16] (one 128-byte line per row and tile), rows of bins
    // (k, M-k) adjacent so that the output kernel merges them while landing: both hot kernels
    // specialised and the window a whole number of layout tiles
    static constexpr int y_tile_w = FC_Y_TILE_W;
    static constexpr int y_tile_shift = FC_Y_TILE_SHIFT;
    bool y_tiled() const { return path_mode == 2 && fast_rows.ok && fast_cols.ok && y_tile_w % fast_cols.T == 0 && fft_w % y_tile_w == 0; }
    int tile_rows() const { return y_tiled() ? M + 2 : rows; }
    int rows_group = -1;       // PlanTuning::rows_group
    bool rows_multi_ok() const { return rows_group != 0 && rows_group != 1 && fast_rows.ok; }
    // As many maps per workgroup as leaves >= 4 workgroups per resident slot (4 per CU), at most 16:
    // the walk amortises the image-spectrum row, the launch and the store drain, but a grid
    // that no longer fills the chip loses more than that.
    int rows_group_for(int nmaps, int num_cus) const {
        if (!rows_multi_ok()) return 1;
        if (rows_group > 1) return std::min(rows_group, nmaps);
        const int g1 = rows_group_auto(nmaps, num_cus);
        if (F == 1) return g1;
        // F > 1: the F image-spectrum rows of a workgroup are re-read for every map from the XCD's L2, and they
        // stay there only while few row groups are in flight per XCD, i.e. while many workgroups share a row
        // group: short walks (measured optimum 4-6 maps at F <= 8, 2 at F = 32; tools/f_group.py,
        // profiles/r02x_f_walk_length.txt: 16-map walks fetch 205 MB per map at F = 8, 4-map walks 62)
        // (round 3, spill-free kernel: 3 / 4 / 6 / 8 / 12 maps at F = 4: 56.2 / 55.1 / 56.9 / 56.7 / 60.0 us per map)
        return std::min(g1, std::max(2, std::min(4, 32 / F)));
    }
    int rows_slots_per_cu = 4; // resident workgroups of the multi-map row kernel per CU (the plan asks the runtime: light
                               // configurations hold 5-6, the 4224-point one 4)
    // Maps per workgroup of the multi-map row kernel for a launch of nmaps.  Large launches (eight or more rounds of
    // one-map workgroups): as many as leaves >= 4 rounds, at most 16 -- the walk amortises the image-spectrum row, the
    // launch and the store drain, but a grid that no longer fills the chip loses more than that.  Small launches: the
    // shortest walk whose FULL walks are all resident at once (the remainder walk of every row group, launched last,
    // fills the slots they leave and the first ones to free up) -- cfg2, 289 row groups x 16 maps on 1536 slots: walks of
    // 1 / 2 / 3 / 4 / 5 maps take 84.6 / 81.4 / 79.2 / 82.6 / 83.6 us per step; 3 = 1445 full walks + 289 single maps.
    int rows_group_auto(int nmaps, int num_cus) const {
        const long groups = (rows + fast_rows.RPW - 1) / fast_rows.RPW;
        const long slots = (long)num_cus * std::max(1, rows_slots_per_cu), total = groups * nmaps;
        if (total >= 8 * slots) return (int)std::max<long>(1, std::min<long>(16, std::min<long>(total / (4 * slots), nmaps)));
        if (total <= slots) return 1;        // everything resident at once
        for (int w = 2; w <= std::min(16, nmaps); w++)
            if (groups * (nmaps / w) <= slots) return w;
        return std::min(16, nmaps);
    }
    size_t spectrum_elems() const { return (size_t)F * rows * s_pitch; }
    size_t y_elems_per_kernel() const {
        return y_tiled() ? (size_t)(fft_w / y_tile_w) * tile_rows() * y_tile_w : (size_t)rows * y_pitch;
    }
    size_t map_elems() const { return (size_t)fft_h * fft_w; }
};

struct Tables {
    Plan1D pm;  // M-point complex transform (h direction)
    Plan1D pw;  // Lw-point complex transform (w direction)
    std::vector<PairEntry> pairs;
    FastRowsTables fr;  // only if Geometry::fast_rows.ok
    FastColsTables fcl; // only if Geometry::fast_cols.ok
    // natural h-frequency y -> spectrum row, natural w-frequency x -> element of a stored image-spectrum row
    std::vector<int> nat_row_of, nat_col_of;
};

// returns false if the sizes are invalid / unsupported
inline bool make_geometry(Geometry& g, Tables& t, int H, int W, int F, int max_kh, int max_kw, const PlanTuning& tune = PlanTuning()) {
    const int path_mode = (tune.path_mode < 0 || tune.path_mode > 2) ? 2 : tune.path_mode;
    const bool allow_fast = path_mode > 0;
    g.path_mode = path_mode;
    g.rows_group = tune.rows_group;
    if (H < 1 || W < 1 || F < 1 || max_kh < 1 || max_kw < 1) return false;
    g.H = H; g.W = W; g.F = F; g.max_kh = max_kh; g.max_kw = max_kw;
    // (F > 1: the XCD-aware workgroup order -- the workgroups that read the same F image-spectrum rows side by side on one L2 --
    //  lives in the walk kernel itself: kernels_rows_multi.inc: k_fast_rows_multi_f)
    g.fft_h = tune.cyclic ? H : fft_size16(H + max_kh - 1);
    g.fft_w = tune.cyclic ? W : fft_size16(W + max_kw - 1);
    if (tune.cyclic && (!allow_fast || (H & 1) || max_kh > H || max_kw > W)) return false;
    LengthPrefs prefs;   // the planner prefers lengths with specialised kernels (able to take max_kw)
    if (allow_fast) { prefs.fast_rows = &fast_rows_factor; prefs.fast_cols = &fast_cols_factor; prefs.max_kw = max_kw; }
    // (max_transform caps the search: a cheaper length above it must not turn a size that fits into one that does not)
    g.Lh = choose_length(H + max_kh - 1, true, g.fft_h, prefs, tune.max_transform);
    g.Lw = choose_length(W + max_kw - 1, false, g.fft_w, prefs, tune.max_transform);
    if (tune.exact_window || tune.cyclic) {
        if (!length_supported(g.fft_h / 2) || !length_supported(g.fft_w)) return false;
        g.Lh = g.fft_h;
        g.Lw = g.fft_w;
    }
    if (g.Lh < 2 || g.Lw < 1) return false;
    if (tune.max_transform > 0 && (g.Lh > tune.max_transform || g.Lw > tune.max_transform)) return false;
    g.M = g.Lh / 2;
    g.rows = g.M + 1;
    g.s_pitch = round_up(g.Lw, 8);
    g.wout = std::min(g.Lw, g.fft_w);
    g.y_pitch = round_up(g.wout, 8);
    g.lds_pitch = lds_col_pitch(g.M);
    g.T_cols = 8;
    while (g.T_cols > 1 && (size_t)g.T_cols * g.lds_pitch * sizeof(c32) > FC_LDS_BUDGET) g.T_cols /= 2;
    if ((size_t)g.T_cols * g.lds_pitch * sizeof(c32) > FC_LDS_BUDGET) return false;
    size_t row_lds = (size_t)g.Lw * sizeof(c32) * (F > 1 ? 2 : 1);
    if (row_lds > FC_LDS_BUDGET) return false;
    g.exact_window = (g.Lh == g.fft_h && g.Lw == g.fft_w);
    t.pm = make_plan1d(g.M);
    t.pw = make_plan1d(g.Lw);
    t.pairs = make_pair_table(t.pm);
    g.fast_rows = allow_fast ? fast_rows_lookup(g.Lw, max_kw) : FastRowsInfo();
    if (g.fast_rows.ok) t.fr = make_fast_rows_tables(g.fast_rows, t.pw);
    // the fast output kernel crops (window <= transform) but does not zero-fill (window > transform)
    g.fast_cols = (allow_fast && g.Lh >= g.fft_h && g.Lw >= g.fft_w) ? fast_cols_lookup(g.M) : FastColsInfo();
    if (g.fast_cols.ok && (g.fft_w % g.fast_cols.T != 0)) g.fast_cols = FastColsInfo();
    g.fast_fwd = g.fast_cols.ok;
    // the plan whose digit-reversed order the spectrum rows are produced in
    Plan1D producer = g.fast_fwd ? make_plan1d_seq(g.M, {g.fast_cols.R1, g.fast_cols.R2, g.fast_cols.R3}) : t.pm;
    if (g.fast_cols.ok) t.fcl = make_fast_cols_tables(g.fast_cols, producer, g.y_pitch);
    if (tune.cyclic && !g.y_tiled()) return false;
    t.nat_row_of.assign(g.rows, g.M);                       // bin M (Nyquist) lives in the extra row M
    for (int k = 0; k < g.M; k++) t.nat_row_of[k] = producer.pos[k];
    t.nat_col_of.assign(g.Lw, 0);
    if (g.fast_rows.ok) {                                   // stored element x holds transform position relayout[x]
        std::vector<int> inv(g.Lw, 0);
        for (int x = 0; x < g.Lw; x++) inv[t.fr.relayout[x]] = x;
        for (int k = 0; k < g.Lw; k++) t.nat_col_of[k] = inv[t.pw.pos[k]];
    } else {
        for (int k = 0; k < g.Lw; k++) t.nat_col_of[k] = t.pw.pos[k];
    }
    return true;
}

// Device-resident copies of the tables (pointers valid on whichever side runs the bodies).
struct DeviceTables {
    const c32* tw_m = nullptr;
    const c32* tw_w = nullptr;
    const PairEntry* pairs = nullptr;
    const c32* fr_tw1 = nullptr;
    const c32* fr_tw2 = nullptr;
    const int* fr_relayout = nullptr;
    const c32* fc_tw1 = nullptr;
    const c32* fc_tw2 = nullptr;
    const PairEntry* fc_pairs = nullptr;
    const int* fc_rowoff = nullptr;
    const int* fc_pair_row_of = nullptr;
    int* queue = nullptr;                     // dynamic tile queue of the persistent column kernels (FC_QUEUE_WORDS ints; plan option "dynamic_tiles")
    bool queue_fwd = false;                   // ... also for the forward column kernels ("dynamic_tiles" = 2; off by default, see fast_cols_fwd_args)
    unsigned long long* timeline = nullptr;   // FC_ROWS_TIMELINE / FC_COLS_TIMELINE builds only (plan option "timeline_ptr")
};

// image columns: planes = F, columns = W, valid samples = H
inline ColsR2CArgs image_cols_args(const Geometry& g, const Tables& t, const DeviceTables& d,
                                   const float* image, c32* S) {
    ColsR2CArgs a{};
    a.in = image; a.in_plane_stride = (size_t)g.H * g.W; a.in_col_pitch = g.H; a.h_in = g.H; a.ncols = g.W;
    a.out = S; a.out_plane_stride = (size_t)g.rows * g.s_pitch; a.out_pitch = g.s_pitch;
    a.M = g.M; a.T = g.T_cols; a.lds_pitch = g.lds_pitch;
    a.fd = t.pm.desc; a.tw = d.tw_m; a.pairs = d.pairs; a.npairs = (int)t.pairs.size();
    return a;
}

inline RowsFwdArgs image_rows_args(const Geometry& g, const Tables& t, const DeviceTables& d, c32* S) {
    RowsFwdArgs a{};
    a.S = S; a.pitch = g.s_pitch; a.nvalid = g.W;
    a.scale = (float)(1.0 / ((double)g.Lh * (double)g.Lw));
    a.fd = t.pw.desc; a.tw = d.tw_w;
    a.out_map = g.fast_rows.ok ? d.fr_relayout : nullptr;   // store straight in the fast row kernel's order
    return a;
}

// the same pass by the specialised kernel (fast_rows_fwd.hpp), when the row length has one
inline FastRowsFwdArgs fast_rows_fwd_args(const Geometry& g, const DeviceTables& d, c32* S) {
    FastRowsFwdArgs a{};
    a.S = S; a.pitch = g.s_pitch; a.nvalid = g.W;
    a.scale = (float)(1.0 / ((double)g.Lh * (double)g.Lw));
    a.tw1 = d.fr_tw1; a.tw2 = d.fr_tw2;
    return a;
}

// kernel columns of a group of same-sized kernels, packed [n][f][kw][kh]: planes = nk*F
inline int a_pitch_for(int kw) { return round_up(kw, 8); }

inline ColsR2CArgs kernel_cols_args(const Geometry& g, const Tables& t, const DeviceTables& d,
                                    const float* kernels, int kh, int kw, c32* A) {
    ColsR2CArgs a{};
    a.in = kernels; a.in_plane_stride = (size_t)kh * kw; a.in_col_pitch = kh; a.h_in = kh; a.ncols = kw;
    a.out = A; a.out_plane_stride = (size_t)g.rows * a_pitch_for(kw); a.out_pitch = a_pitch_for(kw);
    a.M = g.M; a.T = g.T_cols; a.lds_pitch = g.lds_pitch;
    a.fd = t.pm.desc; a.tw = d.tw_m; a.pairs = d.pairs; a.npairs = (int)t.pairs.size();
    return a;
}

inline SpectralRowsArgs spectral_rows_args(const Geometry& g, const Tables& t, const DeviceTables& d,
                                           const c32* A, int kw, const c32* S, c32* Y) {
    SpectralRowsArgs a{};
    a.A = A; a.a_pitch = a_pitch_for(kw); a.a_feat_stride = (size_t)g.rows * a.a_pitch;
    a.a_kernel_stride = (size_t)g.F * a.a_feat_stride; a.kw = kw;
    a.S = S; a.s_feat_stride = (size_t)g.rows * g.s_pitch; a.s_pitch = g.s_pitch;
    a.Y = Y; a.y_kernel_stride = g.y_elems_per_kernel(); a.y_pitch = g.y_pitch; a.wout = g.wout;
    a.F = g.F; a.fd = t.pw.desc; a.tw = d.tw_w;
    return a;
}

// fast spectral rows (S in register order)
inline FastRowsArgs fast_rows_args(const Geometry& g, const DeviceTables& d, const c32* A, int kw, const c32* S, c32* Y) {
    FastRowsArgs a{};
    a.A = A; a.a_pitch = a_pitch_for(kw); a.a_feat_stride = (size_t)g.rows * a.a_pitch;
    a.a_kernel_stride = (size_t)g.F * a.a_feat_stride; a.kw = kw;
    a.S = S; a.s_feat_stride = (size_t)g.rows * g.s_pitch; a.s_pitch = g.s_pitch;
    a.Y = Y; a.y_kernel_stride = g.y_elems_per_kernel(); a.y_pitch = g.y_pitch; a.wout = g.wout;
    a.F = g.F; a.tw1 = d.fr_tw1; a.tw2 = d.fr_tw2;
    a.y_row_of = g.y_tiled() ? d.fc_pair_row_of : nullptr;
    a.y_tile_elems = g.tile_rows() * g.y_tile_w; a.y_tile_shift = g.y_tile_shift;
    a.timeline = d.timeline;
    return a;
}

// fast output columns: nk kernels of the current batch
inline FastColsArgs fast_cols_args(const Geometry& g, const DeviceTables& d, const c32* Y, float* out,
                                   size_t out_kernel_stride, int nk) {
    FastColsArgs a{};
    a.Y = Y; a.y_kernel_stride = g.y_elems_per_kernel(); a.y_pitch = g.y_pitch;
    a.out = out; a.out_kernel_stride = out_kernel_stride; a.fft_h = g.fft_h; a.fft_w = g.fft_w;
    a.h_lo = 0; a.w_first = 0; a.out_pitch = g.fft_h;
    a.tiles_per_kernel = g.fft_w / g.fast_cols.T; a.ntiles = a.tiles_per_kernel * nk;
    a.rowoff = d.fc_rowoff; a.tw1 = d.fc_tw1; a.tw2 = d.fc_tw2; a.pairs = d.fc_pairs;
    a.y_tiled = g.y_tiled() ? 1 : 0; a.y_tile_elems = g.tile_rows() * g.y_tile_w; a.y_tile_shift = g.y_tile_shift;
    a.timeline = d.timeline;
    a.queue = d.queue;          // counters 0..7 (one per XCD)
    return a;
}

// forward column pass of `planes` planes of `ncols` columns with h_in valid samples each; of_kernels: the kernels' pass
// (counter 9 of the dynamic tile queue; the image's pass has counter 8 -- the two may run at the same time, on two
// streams or in one launch)
inline FastColsFwdArgs fast_cols_fwd_args(const Geometry& g, const DeviceTables& d, const float* in, size_t in_plane_stride,
                                          int in_col_pitch, int h_in, int ncols, int planes, c32* out,
                                          size_t out_plane_stride, int out_pitch, bool of_kernels) {
    FastColsFwdArgs a{};
    a.in = in; a.in_plane_stride = in_plane_stride; a.in_col_pitch = in_col_pitch; a.h_in = h_in; a.ncols = ncols;
    a.out = out; a.out_plane_stride = out_plane_stride; a.out_pitch = out_pitch;
    a.tiles_per_plane = (ncols + g.fast_cols.T - 1) / g.fast_cols.T; a.ntiles = a.tiles_per_plane * planes;
    a.tw1 = d.fc_tw1; a.tw2 = d.fc_tw2; a.pairs = d.fc_pairs;
    // The forward kernels take their tiles from ONE counter each, and their tiles are short (a pruned kernel column pass is ~2 us a
    // tile): the 1 024 tickets of a 64-kernel pass queue up on that one cache line -- +12 us per pass at every size, +6-10 us per
    // image pass (profiles/r05k_dynamic_tiles_by_size.txt).  They are 1-3 % of a step and lose next to nothing to a neighbour
    // that holds CUs: static deal unless "dynamic_tiles" = 2 asks for the queue (A/B, tests).
    a.queue = (d.queue && d.queue_fwd) ? d.queue + (of_kernels ? 9 : 8) * FC_QUEUE_STRIDE : nullptr;
    return a;
}

inline int fast_rows_nz2(const Geometry& g, int kw) { return (kw + g.fast_rows.R3 - 1) / g.fast_rows.R3; }

inline ColsC2RArgs cols_c2r_args(const Geometry& g, const Tables& t, const DeviceTables& d,
                                 const c32* Y, float* out, size_t out_kernel_stride) {
    ColsC2RArgs a{};
    a.Y = Y; a.y_kernel_stride = g.y_elems_per_kernel(); a.y_pitch = g.y_pitch; a.wvalid = g.wout;
    a.out = out; a.out_kernel_stride = out_kernel_stride; a.fft_h = g.fft_h; a.fft_w = g.fft_w;
    a.M = g.M; a.T = g.T_cols; a.lds_pitch = g.lds_pitch;
    a.fd = t.pm.desc; a.tw = d.tw_m; a.pairs = d.pairs; a.npairs = (int)t.pairs.size();
    return a;
}

inline int tiles_for(int ncols, int T) { return (ncols + T - 1) / T; }

}  // namespace fc
