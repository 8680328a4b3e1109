// blockwise.cpp -- block-wise plans of libfftconv.so and their planner.
#include <new>

#include "plan_internal.hpp"

// Block-wise plans: sizes whose padded window does not fit a single LDS-resident pass (about 20 000 samples along
// w), any size when fftconv_plan_options.max_transform forces it, and large one-pass sizes that run faster in blocks
// (blocks_preferred below).  Two forms:
//   overlap-save (the default path's specialised kernels exist for the block transform): the block plan is CYCLIC over
//     Lh x Lw samples (PlanTuning::cyclic).  Block (by, bx) is the image's rows [by * Bh - Sh, by * Bh + Bh) -- Bh = Lh - Sh
//     new rows behind Sh >= MAX_KERNEL_H - 1 rows of history, zeros outside the image -- and columns likewise; of its
//     circular result the first Sh rows / Sw columns are wrapped and belong to nobody, the rest IS rows [by * Bh, by * Bh + Bh)
//     of the maps, and the output kernel stores it there (OutWindow): no block maps, no summing pass, every element of the
//     maps written once.  A dimension one block covers has no history (Sh = 0, Lh >= FFT_H: plain zero padding).
//   overlap-add (otherwise): blocks of Bh x Bw samples, zero-padded by an ordinary plan, the block results summed on the
//     device into the full maps at their offsets (convolution is linear and the blocks partition the image).
// The block spectra are computed once per image and kept (the plan's "spectrum" is their concatenation, so the
// multi-device copy / broadcast works unchanged); kernels are processed in chunks that fit a few GiB of device maps.
// The reference has no such limit (cuFFT plans any size: src/cudaFFTData.cu:72-103, src/cudaConvFFTData.cu:92-98);
// kernels larger than MAX_KERNEL cannot be folded block-wise and are rejected.

namespace fc {

int tiled_unsupported(const char* what) {
    return api_fail(FFTCONV_ERR_UNSUPPORTED_SIZE, "%s is not available on a block-wise plan (the padded size does not fit one transform pass)", what);
}

// ---- overlap-save blocks: which block transform, how many blocks ----
// One dimension: transform length L (a length with specialised kernels), n blocks, S history samples in front of each
// block (S = 0 and L >= window when one block covers the dimension; else S = MAX_KERNEL - 1 rounded up to the layout tile).
struct DimChoice { int L = 0, n = 0, S = 0; };
struct SaveTiling {
    bool ok = false;
    DimChoice h, w;
    double ps = 0;    // estimated time per map, picoseconds
};

// estimated time per map of (h, w): the row kernel transforms every spectrum row of every block, the output kernel the
// stored columns only (fast_paths.hpp: measured cost per point of each specialised length; rows_ps / cols_ps < 0: those).
// Every block adds launches, its kernel-column pass and the gaps between them: ~25 us per block and launch, taken over
// 16 maps (profiles/r04k_blocks_vs_one_pass.txt: 18 blocks of 1344 x 3072 lose to one pass of 7680 x 7680 by that alone).
constexpr double kBlockOverheadPs = 1.6e6;
double tiling_ps(const DimChoice& h, const DimChoice& w, int FW, double rows_ps = -1.0, double cols_ps = -1.0) {
    const double rows = (double)h.n * w.n * (h.L / 2 + 1) * w.L * (rows_ps < 0 ? fast_rows_ps(w.L) : rows_ps);
    const double cols = (double)h.n * FW * (h.L / 2) * (cols_ps < 0 ? fast_cols_ps(h.L / 2) : cols_ps);
    return rows + cols + (h.n * w.n > 1 ? kBlockOverheadPs * h.n * w.n : 0.0);
}

std::vector<DimChoice> dim_choices(int window, int mk, bool w_dim, int mkw, int limit) {
    std::vector<DimChoice> v;
    const int S = round_up(std::max(0, mk - 1), Geometry::y_tile_w);
    for (int L = 32; L <= limit; L += 32) {
        const bool have = w_dim ? fast_rows_lookup(L, mkw).ok : fast_cols_lookup(L / 2).ok;
        if (!have || L < mk) continue;
        DimChoice c;
        c.L = L;
        if (L >= window) { c.n = 1; c.S = 0; }
        else if (L - S >= Geometry::y_tile_w) { c.S = S; c.n = (window + (L - S) - 1) / (L - S); }
        else continue;
        v.push_back(c);
    }
    return v;
}

// the cheapest overlap-save tiling of the window FH x FW within transforms of at most `limit` samples
SaveTiling choose_save_tiling(int FH, int FW, int mkh, int mkw, int limit) {
    SaveTiling best;
    const std::vector<DimChoice> hs = dim_choices(FH, mkh, false, mkw, limit), ws = dim_choices(FW, mkw, true, mkw, limit);
    for (const DimChoice& h : hs)
        for (const DimChoice& w : ws) {
            if (h.n > 1 && (h.L - h.S) % Geometry::y_tile_w) continue;
            const double ps = tiling_ps(h, w, FW);
            if (!best.ok || ps < best.ps) { best.ok = true; best.h = h; best.w = w; best.ps = ps; }
        }
    return best;
}

// a plan that fits one pass: do blocks of a shorter transform beat it?  Only the long transforms can lose: the 4-column
// output kernels (M >= 2560), the two-rows-per-CU row kernels (>= 7040 points), and lengths beyond the specialised ones
// (generic kernels, ~2.5 x the cost per point).  The model is good to ~5 %: blocks need a predicted 3 %.
bool blocks_preferred(const Geometry& g, const fftconv_plan_options* options) {
    if (g.path_mode != 2) return false;
    const bool fast = g.fast_rows.ok && g.fast_cols.ok && g.y_tiled();
    if (fast && g.Lh < 5120 && g.Lw < 7040) return false;
    if (!fast && g.Lh <= 8448 && g.Lw <= 8448) return false;      // small or oddly sized: not what blocks are for
    int limit = 4608;
    if (options && options->struct_size >= kOptionsMinSize && options->max_transform > 0) limit = std::min(limit, options->max_transform);
    const SaveTiling t = choose_save_tiling(g.fft_h, g.fft_w, g.max_kh, g.max_kw, limit);
    if (!t.ok || t.h.n * t.w.n < 2) return false;
    DimChoice h1, w1;
    h1.L = g.Lh; h1.n = 1; w1.L = g.Lw; w1.n = 1;
    const double one_pass = tiling_ps(h1, w1, g.fft_w, g.fast_rows.ok ? -1.0 : 6.0, g.fast_cols.ok ? -1.0 : 7.5);
    return t.ps < 0.97 * one_pass;
}

// creates the block plan of a tiled plan; FFTCONV_ERR_UNSUPPORTED_SIZE if no block shape works
int tiled_create(fftconv_plan* p, int H, int W, int F, int mkh, int mkw, void* hip_stream, const fftconv_plan_options* options) {
    int limit = 4224;
    const bool limited = options && options->struct_size >= kOptionsMinSize && options->max_transform > 0;
    if (limited) limit = std::min(limit, options->max_transform);
    TiledState* ts = new (std::nothrow) TiledState();
    if (!ts) return api_fail(FFTCONV_ERR_ALLOC, "out of host memory");
    fftconv_plan_options sub_opts = {};
    if (options && options->struct_size >= kOptionsMinSize) memcpy(&sub_opts, options, std::min(sizeof(sub_opts), options->struct_size));
    sub_opts.struct_size = sizeof(sub_opts);
    sub_opts.blockwise = 1;          // the block plan itself is a single pass
    ts->H = H; ts->W = W; ts->F = F; ts->mkh = mkh; ts->mkw = mkw;
    ts->FH = fft_size16(H + mkh - 1); ts->FW = fft_size16(W + mkw - 1);
    int rc = FFTCONV_ERR_UNSUPPORTED_SIZE;
    // overlap-save first: needs the specialised kernels of the default path for the block transform
    if (tuning_from(options).path_mode == 2) {
        const SaveTiling t = choose_save_tiling(ts->FH, ts->FW, mkh, mkw, limited ? limit : 4608);
        if (t.ok) {
            rc = plan_create_internal(&ts->sub, t.h.L, t.w.L, F, mkh, mkw, p->gpu_id, hip_stream, &sub_opts, true);
            if (rc && rc != FFTCONV_ERR_UNSUPPORTED_SIZE) { delete ts; return rc; }
            if (ts->sub) {
                ts->save = true;
                ts->Lh = t.h.L; ts->Lw = t.w.L; ts->Sh = t.h.S; ts->Sw = t.w.S;
                ts->Bh = t.h.L - t.h.S; ts->Bw = t.w.L - t.w.S; ts->nbh = t.h.n; ts->nbw = t.w.n;
            }
        }
    }
    if (!ts->sub) {                  // overlap-add over ordinary (zero-padded) block plans
        const int full_h = limit - mkh + 1, full_w = limit - mkw + 1;
        if (full_h < 1 || full_w < 1) {
            delete ts;
            return api_fail(FFTCONV_ERR_UNSUPPORTED_SIZE, "kernels up to %dx%d are too large for the block-wise path", mkh, mkw);
        }
        // fewest blocks first: tile only the dimension(s) that need it
        const int cand[3][2] = {{H, std::min(W, full_w)}, {std::min(H, full_h), W}, {std::min(H, full_h), std::min(W, full_w)}};
        for (int c = 0; c < 3 && !ts->sub; c++) {
            ts->Bh = cand[c][0]; ts->Bw = cand[c][1];
            rc = fftconv_plan_create_ex(&ts->sub, ts->Bh, ts->Bw, F, mkh, mkw, p->gpu_id, hip_stream, &sub_opts);
            if (rc && rc != FFTCONV_ERR_UNSUPPORTED_SIZE) { delete ts; return rc; }
        }
        if (!ts->sub) { delete ts; return rc; }
        ts->nbh = (H + ts->Bh - 1) / ts->Bh; ts->nbw = (W + ts->Bw - 1) / ts->Bw;
    }
    ts->nblk = ts->nbh * ts->nbw;
    ts->spec_elems = ts->sub->g.spectrum_elems();
    p->tiled = ts;
    Geometry& g = p->g;             // what fftconv_plan_get_info reports
    g = ts->sub->g;
    g.H = H; g.W = W; g.max_kh = mkh; g.max_kw = mkw; g.fft_h = ts->FH; g.fft_w = ts->FW; g.exact_window = false;
    p->num_cus = ts->sub->num_cus;
    return 0;
}

int tiled_set_image(fftconv_plan* p, const float* data, int location) {
    TiledState* ts = p->tiled;
    fftconv_plan* sub = ts->sub;
    ts->have_image = false;
    if (!ts->specs_x)
        if (int rc = ts->specs.ensure(ts->spec_total())) return rc;
    const int H = ts->H, W = ts->W, F = ts->F, Bh = ts->Bh, Bw = ts->Bw;
    FC_VERBOSE(p, "Data size: h=%d, w=%d, f=%d", H, W, F);
    FC_VERBOSE(p, "FFT size: h=%d, w=%d (block-wise, %s: %d x %d blocks of %d x %d samples, block transforms %d x %d)", ts->FH, ts->FW,
               ts->save ? "overlap-save" : "overlap-add", ts->nbh, ts->nbw, Bh, Bw, sub->g.Lh, sub->g.Lw);
    if (ts->save) {
        // block (by, bx) of the block plan's Lh x Lw samples: image rows [by * Bh - Sh, by * Bh + Bh) (zeros outside the image)
        const int Lh = ts->Lh, Lw = ts->Lw;
        if (location == FFTCONV_HOST) ts->hblk.resize((size_t)Lh * Lw * F);
        else if (int rc = ts->blk.ensure((size_t)Lh * Lw * F)) return rc;
        for (int b = 0; b < ts->nblk; b++) {
            const int y0 = (b % ts->nbh) * Bh - ts->Sh, x0 = (b / ts->nbh) * Bw - ts->Sw;       // image coordinates of the block's sample (0, 0)
            const int ys = std::max(0, y0), ye = std::min(H, y0 + Lh), xs = std::max(0, x0), xe = std::min(W, x0 + Lw);
            const bool any = ye > ys && xe > xs;
            if (int rc = fftconv_plan_use_spectrum_buffer(sub, ts->spec_base() + (size_t)b * ts->spec_elems, ts->spec_elems * sizeof(c32))) return rc;
            if (location == FFTCONV_HOST) {
                std::fill(ts->hblk.begin(), ts->hblk.end(), 0.f);
                for (int f = 0; f < F && any; f++)
                    for (int x = xs; x < xe; x++)
                        memcpy(&ts->hblk[((size_t)f * Lw + (x - x0)) * Lh + (ys - y0)], &data[((size_t)f * W + x) * H + ys], (size_t)(ye - ys) * sizeof(float));
                if (int rc = fftconv_plan_set_image(sub, ts->hblk.data(), FFTCONV_HOST)) return rc;   // (hblk is consumed on return: copied or staged)
            } else {
                HIP_TRY(hipMemsetAsync(ts->blk.p, 0, (size_t)Lh * Lw * F * sizeof(float), sub->stream));
                for (int f = 0; f < F && any; f++)
                    HIP_TRY(hipMemcpy2DAsync(ts->blk.p + ((size_t)f * Lw + (xs - x0)) * Lh + (ys - y0), (size_t)Lh * sizeof(float),
                                             data + ((size_t)f * W + xs) * H + ys, (size_t)H * sizeof(float), (size_t)(ye - ys) * sizeof(float),
                                             (size_t)(xe - xs), hipMemcpyDeviceToDevice, sub->stream));
                if (int rc = fftconv_plan_set_image(sub, ts->blk.p, FFTCONV_DEVICE)) return rc;
            }
        }
        ts->have_image = true;
        return 0;
    }
    if (location == FFTCONV_HOST) ts->hblk.assign((size_t)Bh * Bw * F, 0.f);
    else if (int rc = ts->blk.ensure((size_t)Bh * Bw * F)) return rc;
    for (int b = 0; b < ts->nblk; b++) {
        const int y0 = (b % ts->nbh) * Bh, x0 = (b / ts->nbh) * Bw;
        const int hv = std::min(Bh, H - y0), wv = std::min(Bw, W - x0);
        if (int rc = fftconv_plan_use_spectrum_buffer(sub, ts->spec_base() + (size_t)b * ts->spec_elems, ts->spec_elems * sizeof(c32))) return rc;
        if (location == FFTCONV_HOST) {
            std::fill(ts->hblk.begin(), ts->hblk.end(), 0.f);
            for (int f = 0; f < F; f++)
                for (int x = 0; x < wv; x++)
                    memcpy(&ts->hblk[((size_t)f * Bw + x) * Bh], &data[((size_t)f * W + (x0 + x)) * H + y0], (size_t)hv * sizeof(float));
            if (int rc = fftconv_plan_set_image(sub, ts->hblk.data(), FFTCONV_HOST)) return rc;   // (hblk is consumed on return: copied or staged)
        } else {
            // the block, zero-padded, on the device: one strided copy per feature plane (h is contiguous)
            if (hv < Bh || wv < Bw) HIP_TRY(hipMemsetAsync(ts->blk.p, 0, (size_t)Bh * Bw * F * sizeof(float), sub->stream));
            for (int f = 0; f < F; f++)
                HIP_TRY(hipMemcpy2DAsync(ts->blk.p + (size_t)f * Bw * Bh, (size_t)Bh * sizeof(float),
                                         data + ((size_t)f * W + x0) * H + y0, (size_t)H * sizeof(float), (size_t)hv * sizeof(float), (size_t)wv,
                                         hipMemcpyDeviceToDevice, sub->stream));
            if (int rc = fftconv_plan_set_image(sub, ts->blk.p, FFTCONV_DEVICE)) return rc;
        }
    }
    ts->have_image = true;
    return 0;
}

// The finished full-window maps of a chunk of kernels (`big`, nk maps from kernel k0 on) on their way to the caller: cropped to
// the plan's "output_region" where one is set, then copied to the caller's pointers -- or nothing at all where the blocks
// stored straight into the caller's packed maps.  more: another chunk follows and reuses `big`.
static int tiled_deliver(fftconv_plan* p, float* big, int nk, int k0, bool more, float* const* out, int out_location, float* out_packed) {
    TiledState* ts = p->tiled;
    fftconv_plan* sub = ts->sub;
    const size_t big_map = ts->big_map(), oe = p->out_elems();
    const float* src = big;
    if (p->opt_region != 0) {
        float* dst = out_packed ? out_packed + (size_t)k0 * oe : ts->crop.p;
        if (p->opt_region == 4) HIP_TRY(launch_pad_maps(big, ts->FH, ts->FW, big_map, dst, p->out_h, p->out_w, oe, nk, sub->stream));
        else HIP_TRY(launch_crop_maps(big, ts->FH, big_map, dst, p->out_h, p->out_w, oe, p->off_h, p->off_w, nk, sub->stream));
        src = dst;
    } else if (out_packed) {
        return 0;
    }
    if (!out_packed)
        for (int j = 0; j < nk; j++)
            HIP_TRY(hipMemcpyAsync(out[k0 + j], src + (size_t)j * oe, oe * sizeof(float),
                                   out_location == FFTCONV_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, sub->stream));
    if ((!out_packed && out_location == FFTCONV_HOST) || more) HIP_TRY(hipStreamSynchronize(sub->stream));   // `big` (and the crop staging) are reused by the next chunk
    return 0;
}

// Overlap-save: every block's run stores its rectangle of the maps from the output kernel (OutWindow) -- no block maps, no
// summing pass, every element of the maps written once.  Kernels of equal size go through the block plan group by group
// (run_group); host kernels, and device kernels that are not consecutive in memory, are packed on the device once per call.
int tiled_convolve_save(fftconv_plan* p, int n, const float* const* kernels, const int* kh, const int* kw, int kernel_location,
                        float* const* out, int out_location, float* out_packed) {
    TiledState* ts = p->tiled;
    fftconv_plan* sub = ts->sub;
    const size_t big_map = ts->big_map();
    const size_t budget = (size_t)6 << 30;
    // "output_region" other than the window: the blocks still store into full-window maps (the output kernel's stores are
    // pairs of rows at even offsets: an arbitrary rectangle is not), a crop kernel compacts the region out of them
    const bool cropped = p->opt_region != 0;
    const size_t oe = p->out_elems();
    const bool direct = out_packed && !cropped;       // the blocks store straight into the caller's packed maps
    const int nc = direct ? n : (int)std::max<size_t>(1, std::min<size_t>((size_t)n, budget / (big_map * sizeof(float))));
    if (!direct)
        if (int rc = ts->big.ensure(big_map * nc)) return rc;
    if (cropped && !out_packed)
        if (int rc = ts->crop.ensure(oe * nc)) return rc;
    FC_VERBOSE(p, "N Kernel: %d (block-wise, overlap-save: %d blocks, %d kernels per chunk)", n, ts->nblk, nc);
    struct Group { int first, count; const float* dk; };
    for (int k0 = 0; k0 < n; k0 += nc) {
        const int nk = std::min(nc, n - k0);
        float* big = direct ? out_packed + (size_t)k0 * big_map : ts->big.p;
        // groups of consecutive kernels of equal size, each packed on the device
        std::vector<Group> groups;
        size_t stage_total = 0;
        for (int j = 0; j < nk;) {
            int e = j + 1;
            while (e < nk && kh[k0 + e] == kh[k0 + j] && kw[k0 + e] == kw[k0 + j]) e++;
            const size_t per = (size_t)ts->F * kh[k0 + j] * kw[k0 + j];
            bool packed = kernel_location == FFTCONV_DEVICE;
            for (int i = j + 1; i < e && packed; i++) packed = kernels[k0 + i] == kernels[k0 + i - 1] + per;
            groups.push_back(Group{j, e - j, packed ? kernels[k0 + j] : nullptr});
            if (!packed) stage_total += per * (size_t)(e - j);
            j = e;
        }
        if (stage_total) {
            if (int rc = ts->kstage.ensure(stage_total)) return rc;
            size_t off = 0;
            for (Group& gr : groups) {
                if (gr.dk) continue;
                const size_t per = (size_t)ts->F * kh[k0 + gr.first] * kw[k0 + gr.first];
                gr.dk = ts->kstage.p + off;
                for (int i = 0; i < gr.count; i++, off += per)
                    HIP_TRY(hipMemcpyAsync(ts->kstage.p + off, kernels[k0 + gr.first + i], per * sizeof(float),
                                           kernel_location == FFTCONV_HOST ? hipMemcpyHostToDevice
                                           : kernel_location == FFTCONV_AUTO ? hipMemcpyDefault : hipMemcpyDeviceToDevice, sub->stream));
            }
        }
        for (int b = 0; b < ts->nblk; b++) {
            const int y0 = (b % ts->nbh) * ts->Bh, x0 = (b / ts->nbh) * ts->Bw;       // the block's rectangle of the maps starts here
            if (int rc = fftconv_plan_use_spectrum_buffer(sub, ts->spec_base() + (size_t)b * ts->spec_elems, ts->spec_elems * sizeof(c32))) return rc;
            if (int rc = fftconv_plan_mark_spectrum_valid(sub)) return rc;
            OutWindow win;
            win.map_stride = big_map; win.pitch = ts->FH;
            win.h_lo = ts->Sh; win.h_hi = ts->Sh + std::min(ts->Bh, ts->FH - y0);
            win.w_first = ts->Sw; win.ncols = std::min(ts->Bw, ts->FW - x0);
            int rc = 0;
            for (const Group& gr : groups) {
                // (row h of column w of the block's result belongs at row y0 + h - Sh of column x0 + w - Sw of the map)
                win.base = big + (size_t)gr.first * big_map + ((ptrdiff_t)(x0 - ts->Sw) * ts->FH + (y0 - ts->Sh));
                sub->win = &win;
                Sink sink;
                sink.packed = win.base;     // unused: the window decides where the maps go
                // one group that fits one chunk of column spectra: block 0 left the kernels' column spectra in the block plan
                // (every block runs the same transform), the other blocks reuse them
                if (b > 0 && groups.size() == 1 && gr.count <= batch_sizes(sub, gr.count, kw[k0 + gr.first]).nbA && !sub->deferred.on) {
                    sub->prepared.dk = gr.dk; sub->prepared.n = gr.count; sub->prepared.kh = kh[k0 + gr.first]; sub->prepared.kw = kw[k0 + gr.first];
                    sub->prepared.stream = sub->stream;
                }
                rc = run_group(sub, gr.count, gr.dk, kh[k0 + gr.first], kw[k0 + gr.first], sink);
                sub->win = nullptr;
                if (rc) return rc;
            }
        }
        if (int rc = tiled_deliver(p, big, nk, k0, k0 + nc < n, out, out_location, out_packed)) return rc;
        if (stage_total && k0 + nc < n) HIP_TRY(hipStreamSynchronize(sub->stream));                       // ... and so is the kernel staging
    }
    FC_VERBOSE(p, "FFT done");
    return 0;
}

// n kernels (pointers, any location) -> n full maps.  out_packed != nullptr: device memory, maps consecutive (the block
// results are summed straight into it); else one pointer per map in `out` (host or device memory).
int tiled_convolve(fftconv_plan* p, int n, const float* const* kernels, const int* kh, const int* kw, int kernel_location,
                   float* const* out, int out_location, float* out_packed) {
    TiledState* ts = p->tiled;
    fftconv_plan* sub = ts->sub;
    if (!ts->have_image) return api_fail(FFTCONV_ERR_NO_IMAGE, "no image spectrum: call fftconv_plan_set_image first");
    for (int k = 0; k < n; k++) {
        if (!kernels[k]) return api_fail(FFTCONV_ERR_INVALID_ARG, "kernel %d is NULL", k);
        if (!out_packed && !out[k]) return api_fail(FFTCONV_ERR_INVALID_ARG, "output %d is NULL", k);   // everything checked before anything is queued
        if (kh[k] < 1 || kw[k] < 1 || kh[k] > ts->FH || kw[k] > ts->FW)      // src/cudaConvolutionFFT.cu:242
            return api_fail(FFTCONV_ERR_KERNEL_SHAPE,
                        "Kernel and Data must have the same number of features and kernel size should be smaller than data size");
        if (kh[k] > ts->mkh || kw[k] > ts->mkw)
            return api_fail(FFTCONV_ERR_KERNEL_EXCEEDS_MAX, "kernel %dx%d exceeds MAX_KERNEL %dx%d (block-wise path)", kh[k], kw[k], ts->mkh, ts->mkw);
    }
    if (ts->save) return tiled_convolve_save(p, n, kernels, kh, kw, kernel_location, out, out_location, out_packed);
    const size_t big_map = ts->big_map(), blk_map = sub->g.map_elems();
    const size_t budget = (size_t)6 << 30;
    const int nc = (int)std::max<size_t>(1, std::min<size_t>((size_t)n, budget / ((big_map + blk_map) * sizeof(float))));
    const bool cropped = p->opt_region != 0;          // as in tiled_convolve_save: full-window maps first, the region cropped out of them
    const bool direct = out_packed && !cropped;
    if (!direct)
        if (int rc = ts->big.ensure(big_map * nc)) return rc;
    if (cropped && !out_packed)
        if (int rc = ts->crop.ensure(p->out_elems() * nc)) return rc;
    if (int rc = ts->tmp.ensure(blk_map * nc)) return rc;
    std::vector<float*> tptr(nc);
    for (int j = 0; j < nc; j++) tptr[j] = ts->tmp.p + (size_t)j * blk_map;
    FC_VERBOSE(p, "N Kernel: %d (block-wise: %d blocks, %d kernels per chunk)", n, ts->nblk, nc);
    for (int k0 = 0; k0 < n; k0 += nc) {
        const int nk = std::min(nc, n - k0);
        float* big = direct ? out_packed + (size_t)k0 * big_map : ts->big.p;
        // host (or mixed) kernels: on the device once per chunk, not once per block (every block convolves the same kernels)
        const float* const* kptr = kernels + k0;
        int kloc = kernel_location;
        std::vector<const float*> staged;
        if (kernel_location != FFTCONV_DEVICE && ts->nblk > 1) {
            size_t total = 0;
            for (int j = 0; j < nk; j++) total += (size_t)ts->F * kh[k0 + j] * kw[k0 + j];
            if (int rc = ts->kstage.ensure(total)) return rc;
            staged.resize(nk);
            size_t off = 0;
            for (int j = 0; j < nk; j++) {
                const size_t per = (size_t)ts->F * kh[k0 + j] * kw[k0 + j];
                HIP_TRY(hipMemcpyAsync(ts->kstage.p + off, kernels[k0 + j], per * sizeof(float),
                                       kernel_location == FFTCONV_HOST ? hipMemcpyHostToDevice : hipMemcpyDefault, sub->stream));
                staged[j] = ts->kstage.p + off;
                off += per;
            }
            kptr = staged.data();
            kloc = FFTCONV_DEVICE;
        }
        HIP_TRY(hipMemsetAsync(big, 0, big_map * nk * sizeof(float), sub->stream));
        for (int b = 0; b < ts->nblk; b++) {
            const int y0 = (b % ts->nbh) * ts->Bh, x0 = (b / ts->nbh) * ts->Bw;
            if (int rc = fftconv_plan_use_spectrum_buffer(sub, ts->spec_base() + (size_t)b * ts->spec_elems, ts->spec_elems * sizeof(c32))) return rc;
            if (int rc = fftconv_plan_mark_spectrum_valid(sub)) return rc;
            if (int rc = fftconv_plan_convolve(sub, nk, kptr, kh + k0, kw + k0, kloc, tptr.data(), FFTCONV_DEVICE)) return rc;
            hipError_t e = launch_add_window(big, ts->FH, ts->FW, big_map, y0, x0, ts->tmp.p, sub->g.fft_h, sub->g.fft_w, blk_map, nk, sub->stream);
            if (e != hipSuccess) return api_fail(FFTCONV_ERR_HIP, "overlap-add failed: %s", hipGetErrorString(e));
        }
        if (int rc = tiled_deliver(p, big, nk, k0, k0 + nc < n, out, out_location, out_packed)) return rc;
    }
    FC_VERBOSE(p, "FFT done");
    return 0;
}

}  // namespace fc
