// fftconv_api.cpp -- host side of libfftconv.so: the C ABI of include/fftconv.h on top of the
// HIP kernels.  C++ host code in the role of the reference's MEX gateways
// (src/cudaConvolutionFFT.cu, src/cudaFFTData.cu, src/cudaConvFFTData.cu); no CPU compute path.
#include <cstdarg>
#include <cstdlib>
#include <mutex>
#include <new>
#include <set>

#include "plan_internal.hpp"

namespace {

thread_local std::string g_last_error;

}  // namespace

namespace fc {
int api_fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}
std::string api_last_error() { return g_last_error; }
void api_set_last_error(const std::string& msg) { g_last_error = msg; }
}  // namespace fc

namespace {

// live plans of this process (fftconv_plan_is_live: the MEX gateways validate handles with it)
std::mutex g_live_mutex;
std::set<const fftconv_plan*> g_live_plans;

int cols_threads(const Geometry& g) {
    long work = (long)g.T_cols * g.M;
    if (work >= 4096) return 512;
    if (work >= 1024) return 256;
    if (work >= 256) return 128;
    return 64;
}
int rows_threads(const Geometry& g) {
    if (g.Lw >= 2048) return 256;
    if (g.Lw >= 512) return 128;
    return 64;
}

}  // namespace

namespace fc {

int use_device(const fftconv_plan* p) {
    HIP_TRY(hipSetDevice(p->gpu_id));
    return 0;
}

BatchSizes batch_sizes(const fftconv_plan* p, int n, int kw) {
    const Geometry& g = p->g;
    BatchSizes b;
    b.per_a = (size_t)g.F * g.rows * a_pitch_for(kw);
    const size_t y_bytes = g.y_elems_per_kernel() * sizeof(c32);
    // maps per launch.  auto: enough to amortise the last partially filled wave of workgroups (the
    // two hot kernels run ~2 "rounds" of workgroups per map on 256 CUs; 32 maps make both round
    // counts nearly integral at cfg3, and 64 let the multi-map row kernel walk 16 maps per workgroup
    // on a grid that still fills the chip), capped at 5 GiB of intermediate
    b.nbY = (int)p->opt_batch_maps;
    if (b.nbY <= 0) b.nbY = (int)std::max<size_t>(1, std::min<size_t>(64, ((size_t)5120 << 20) / y_bytes));
    b.nbY = std::min(b.nbY, n);
    // kernels per column-spectrum chunk.  auto: one chunk per launch of the row kernel -- the chunk (138 MB for
    // 64 kernels at cfg3) is then still in the 256-MB Infinity Cache when the row kernel prefetches its rows
    // (row kernel 22.55 -> 22.08 us per map against 192-kernel chunks, profiles/r02w_kernel_chunk_ab.txt);
    // option kernel_chunk_mb > 0: as many launches' worth as fit that budget
    b.nbA = b.nbY;
    if (p->opt_kernel_chunk_mb > 0) {
        const size_t a_budget = (size_t)p->opt_kernel_chunk_mb << 20;
        b.nbA = (int)std::max<size_t>(1, a_budget / (b.per_a * sizeof(c32)));
        b.nbA = std::max(b.nbY, b.nbA / b.nbY * b.nbY);
        b.nbA = std::min(b.nbA, (n + b.nbY - 1) / b.nbY * b.nbY);
    }
    return b;
}

int check_kernel_size(const fftconv_plan* p, int kh, int kw) {
    const Geometry& g = p->g;
    if (kh < 1 || kw < 1 || kh > g.fft_h || kw > g.fft_w)  // src/cudaConvolutionFFT.cu:242
        return api_fail(FFTCONV_ERR_KERNEL_SHAPE,
                    "Kernel and Data must have the same number of features and kernel size should be smaller than data size");
    if ((kh > g.max_kh || kw > g.max_kw) && !(g.exact_window && kh <= g.Lh && kw <= g.Lw))
        return api_fail(FFTCONV_ERR_KERNEL_EXCEEDS_MAX,
                    "kernel %dx%d exceeds MAX_KERNEL %dx%d and the internal transform (%dx%d) is not the %dx%d window",
                    kh, kw, g.max_kh, g.max_kw, g.Lh, g.Lw, g.fft_h, g.fft_w);
    if (g.fast_rows.ok && kw > g.fast_rows.max_kw)
        return api_fail(FFTCONV_ERR_KERNEL_EXCEEDS_MAX, "kernel width %d exceeds what this plan's row kernel accepts (%d)", kw,
                    g.fast_rows.max_kw);
    return 0;
}

}  // namespace fc

namespace {

// h-transform of the kernels [a0, a0 + na) of a packed group into the column-spectrum buffer A
int launch_kernel_cols(fftconv_plan* p, const float* dk, int a0, int na, int kh, int kw) {
    const Geometry& g = p->g;
    if (p->opt_flip_kernels) {   // template matching: correlate instead of convolve (demoCudaConvolutionFFT.m:63-69)
        const size_t chunk = (size_t)na * g.F * kh * kw;
        if (int rc = p->KF.ensure(chunk)) return rc;
        HIP_TRY(launch_flip_planes(dk + (size_t)a0 * g.F * kh * kw, p->KF.p, kh * kw, (long)na * g.F, p->stream));
        dk = p->KF.p;
        a0 = 0;
    }
    if (int rc = p->prof_begin(PK_KERNEL_COLS, na)) return rc;
    if (g.fast_fwd) {
        FastColsFwdArgs fa = fast_cols_fwd_args(g, p->d, dk + (size_t)a0 * g.F * kh * kw, (size_t)kh * kw, kh, kh, kw, na * g.F,
                                                p->A.p, (size_t)g.rows * a_pitch_for(kw), a_pitch_for(kw), true);
        HIP_TRY(launch_fast_cols_fwd(g.M, g.fast_cols.T, fast_cols_fwd_pruned_ok(g.fast_cols, kh), fa, p->num_cus, p->stream));
    } else {
        ColsR2CArgs ka = kernel_cols_args(g, p->t, p->d, dk + (size_t)a0 * g.F * kh * kw, kh, kw, p->A.p);
        HIP_TRY(launch_cols_r2c(ka, tiles_for(kw, g.T_cols), na * g.F, cols_threads(g), p->cols_lds(), p->stream));
    }
    return p->prof_end();
}

// the deferred kernel-column pass of fftconv_plan_prepare_kernels_packed, on its own (no set_image came first)
int flush_pending_prepare(fftconv_plan* p) {
    if (!p->deferred.on) return 0;
    const auto pd = p->deferred;
    p->deferred.on = false;
    const hipStream_t cur = p->stream;
    p->stream = pd.stream;                      // where the caller ordered the kernels' readiness
    const int rc = launch_kernel_cols(p, pd.dk, 0, pd.na, pd.kh, pd.kw);
    p->stream = cur;
    if (rc) return rc;
    p->prepared.dk = pd.dk; p->prepared.n = pd.n; p->prepared.kh = pd.kh; p->prepared.kw = pd.kw; p->prepared.stream = pd.stream;
    return 0;
}

// Core of the per-kernel loop (src/cudaConvolutionFFT.cu:204-291) for n kernels of one size,
// packed on the device at dk ([n][F][kw][kh]).
int run_group_impl(fftconv_plan* p, int n, const float* dk, int kh, int kw, const Sink& sink) {
    const Geometry& g = p->g;
    if (!p->have_image) return api_fail(FFTCONV_ERR_NO_IMAGE, "no image spectrum: call fftconv_plan_set_image first");
    if (int rc = check_kernel_size(p, kh, kw)) return rc;
    if (int rc = flush_pending_prepare(p)) return rc;
    const BatchSizes bs = batch_sizes(p, n, kw);
    const size_t per_a = bs.per_a;
    const int nbY = bs.nbY, nbA = bs.nbA;
    FC_VERBOSE(p, "Kernel size: h=%d, w=%d", kh, kw);                 // src/cudaConvolutionFFT.cu:240
    FC_VERBOSE(p, "N Kernel: %d (maps per launch %d, kernels per column-spectrum chunk %d, %s)", n, nbY, nbA,
               sink.packed ? "packed device output" : sink.location == FFTCONV_HOST ? "host output" : "device output");   // :68
    if (int rc = p->A.ensure(per_a * nbA)) return rc;
    if (int rc = p->Y.ensure(g.y_elems_per_kernel() * nbY)) return rc;
    const OutWindow* win = p->win;
    if (win && !g.fast_cols.ok) return api_fail(FFTCONV_ERR_INVALID_ARG, "an output window needs the specialised output kernel");
    const bool staged = (sink.packed == nullptr) && !win;
    // host output: two staging buffers, the copy-out of batch b overlaps the compute of batch b + 1
    // ... for maps of at least host_min_kb (1 MiB).  Smaller ones leave by blocking copies on the plan's stream:
    // the copy threads buy them nothing (a one-shot call would start and join them for a few hundred KB), and
    // small destination buffers are heap neighbours that share pages, which the runtime pins in place from
    // several threads at once.  A one-shot call on 92-KB maps died (SIGABRT / SIGSEGV, no message) about once
    // in 50-100 runs of tests/test_gpu_parity.py::test_blockwise_one_shot_matches_oracle on some boxes of the
    // pool and never on others; the cause was not isolated, these threads are what that call had to itself.
    const bool streamed = staged && sink.location == FFTCONV_HOST && p->opt_host_stream != 0 &&
                          p->out_elems() * sizeof(float) >= ((size_t)p->opt_host_min_kb << 10);
    // a region other than the whole window: the output kernel writes the window into O, a crop
    // kernel compacts the region into the destination (the caller's packed buffer or the staging OC)
    const bool cropped = p->opt_region != 0;
    const size_t oe = p->out_elems();
    DevBuf<float>& stage = cropped ? p->OC : p->O;
    if (cropped)
        if (int rc = p->O.ensure(g.map_elems() * nbY)) return rc;
    if (staged)
        if (int rc = stage.ensure(oe * nbY * (streamed ? 2 : 1))) return rc;
    if (streamed)
        if (int rc = ring_ensure(p)) return rc;
    if (p->Y.fresh) {   // before anything is written into it: the tuner may keep another allocation
        int tune_k = (int)p->opt_tune_placement;
        if (tune_k < 0) tune_k = win ? 0 : placement_auto_candidates(p, (size_t)std::min(nbY, n) * g.map_elems() * sizeof(float));
        if (tune_k > 1 && !win) {
            const bool direct = !cropped && !staged;   // the output kernel writes straight into the caller's packed buffer
            float* first_obase = cropped ? p->O.p : (staged ? stage.p : sink.packed);
            if (int rc = tune_intermediate_placement(p, tune_k, direct ? n : std::min(nbY, n), nbY, first_obase, direct ? oe : 0)) return rc;
        }
        p->Y.fresh = false;
    }
    int batch = 0;
    struct { bool valid = false; int first = 0, count = 0, buf = 0; } prev;

    const int T = g.T_cols;
    const int cthreads = cols_threads(g), rthreads = rows_threads(g);
    for (int a0 = 0; a0 < n; a0 += nbA) {
        const int na = std::min(nbA, n - a0);
        const bool have_cols = (a0 == 0 && p->prepared.dk == dk && p->prepared.n == n && p->prepared.kh == kh && p->prepared.kw == kw &&
                                p->prepared.stream == p->stream);
        p->prepared.dk = nullptr;   // A is about to be consumed / overwritten
        if (!have_cols)
            if (int rc = launch_kernel_cols(p, dk, a0, na, kh, kw)) return rc;
        for (int y0 = 0; y0 < na; y0 += nbY) {
            const int ny = std::min(nbY, na - y0);
            FC_VERBOSE(p, "maps %d..%d: spectral rows (%s, %d rows x %d points, %d maps per workgroup), output columns (%s, %d-point, %d columns per tile)",
                       a0 + y0, a0 + y0 + ny - 1, g.fast_rows.ok ? "specialised" : "generic", g.rows, g.Lw, g.rows_group_for(ny, p->num_cus),
                       g.fast_cols.ok ? "specialised" : "generic", g.M, g.fast_cols.ok ? g.fast_cols.T : g.T_cols);
            if (int rc = p->prof_begin(PK_SPECTRAL, ny)) return rc;
            if (g.rows_group_for(ny, p->num_cus) > 1) {
                FastRowsArgs fa = fast_rows_args(g, p->d, p->A.p + (size_t)y0 * per_a, kw, p->spec(), p->Y.p);
                HIP_TRY(launch_fast_rows_multi(g.Lw, fast_rows_nz2(g, kw), fa, g.rows, ny, g.rows_group_for(ny, p->num_cus), p->stream));
            } else if (g.fast_rows.ok) {
                FastRowsArgs fa = fast_rows_args(g, p->d, p->A.p + (size_t)y0 * per_a, kw, p->spec(), p->Y.p);
                HIP_TRY(launch_fast_rows(g.Lw, fast_rows_nz2(g, kw), fa, g.rows, ny, p->stream));
            } else {
                SpectralRowsArgs sa = spectral_rows_args(g, p->t, p->d, p->A.p + (size_t)y0 * per_a, kw, p->spec(), p->Y.p);
                HIP_TRY(launch_spectral_rows(sa, g.rows, ny, rthreads, p->rows_lds(), p->stream));
            }
            if (int rc = p->prof_end()) return rc;
            const int buf = streamed ? (batch & 1) : 0;
            float* dest = win ? win->base + (size_t)(a0 + y0) * win->map_stride                               // where the maps of this batch go
                          : staged ? stage.p + (size_t)buf * nbY * oe : sink.packed + (size_t)(a0 + y0) * oe;
            float* obase = cropped ? p->O.p : dest;                                                            // where the output kernel writes
            if (streamed && batch >= 2) {   // staging buffer `buf` still holds batch - 2 until its copy-out is over
                if (p->ring->nslots == 0) p->ring->wait_staging_free(buf);
                else HIP_TRY(hipStreamWaitEvent(p->stream, p->ring->copy_done[buf], 0));
            }
            if (int rc = p->prof_begin(PK_OUT_COLS, ny)) return rc;
            if (g.fast_cols.ok) {
                FastColsArgs fa = fast_cols_args(g, p->d, p->Y.p, obase, win ? win->map_stride : g.map_elems(), ny);
                if (win) {
                    fa.h_lo = win->h_lo; fa.fft_h = win->h_hi; fa.w_first = win->w_first; fa.out_pitch = win->pitch;
                    fa.tiles_per_kernel = win->ncols / g.fast_cols.T; fa.ntiles = fa.tiles_per_kernel * ny;
                }
                HIP_TRY(launch_fast_cols(g.M, g.fast_cols.T, fa, p->num_cus, p->stream));
            } else {
                ColsC2RArgs ca = cols_c2r_args(g, p->t, p->d, p->Y.p, obase, g.map_elems());
                HIP_TRY(launch_cols_c2r(ca, tiles_for(g.fft_w, T), ny, cthreads, p->cols_lds(), p->stream));
            }
            if (int rc = p->prof_end()) return rc;
            if (cropped && p->opt_region == 4)
                HIP_TRY(launch_pad_maps(p->O.p, g.fft_h, g.fft_w, g.map_elems(), dest, p->out_h, p->out_w, oe, ny, p->stream));
            else if (cropped)
                HIP_TRY(launch_crop_maps(p->O.p, g.fft_h, g.map_elems(), dest, p->out_h, p->out_w, oe, p->off_h, p->off_w, ny, p->stream));
            if (streamed) {
                HIP_TRY(hipEventRecord(p->ring->compute_done[buf], p->stream));
                if (prev.valid)
                    if (int rc = ring_drain(p, sink, prev.first, prev.count, prev.buf, stage.p + (size_t)prev.buf * nbY * oe)) return rc;
                prev.valid = true; prev.first = a0 + y0; prev.count = ny; prev.buf = buf;
                batch++;
            } else if (staged && sink.location == FFTCONV_HOST && p->opt_host_pinned && oe * sizeof(float) <= FC_PIN_OUT_BYTES / 2 &&
                       (ny > 1 || oe * sizeof(float) <= FC_PIN_ONE_MAP_BYTES)) {
                // small maps to host arrays: as many as fit the plan's pinned buffer come back in ONE copy and are handed
                // out by the CPU (a copy into pageable memory is a blocking runtime call per map)
                // (two halves: the copy of the next chunk runs while the CPU hands out the current one; a single map above
                //  FC_PIN_ONE_MAP_BYTES takes the plain copy below: the CPU's pass over it costs more than the runtime's pinning --
                //  one 324-KiB map: 77 against 60 us per convolve, four of them: 131 against 166, profiles/r04s_small_call_latency.txt)
                const size_t mb = oe * sizeof(float);
                const int per_copy = (int)std::min<size_t>((size_t)ny, (FC_PIN_OUT_BYTES / 2) / mb);
                const int nchunks = (ny + per_copy - 1) / per_copy;
                if (int rc = p->pin_out.ensure((size_t)per_copy * mb * (nchunks > 1 ? 2 : 1))) return rc;
                for (int h = 0; h < 2; h++)
                    if (!p->pin_out_done[h]) HIP_TRY(hipEventCreateWithFlags(&p->pin_out_done[h], hipEventDisableTiming));
                auto copy_chunk = [&](int c) -> hipError_t {
                    const int j0 = c * per_copy, nj = std::min(per_copy, ny - j0);
                    hipError_t e = hipMemcpyAsync(p->pin_out.p + (size_t)(c & 1) * per_copy * mb, stage.p + (size_t)j0 * oe, (size_t)nj * mb,
                                                  hipMemcpyDeviceToHost, p->stream);
                    if (e == hipSuccess) e = hipEventRecord(p->pin_out_done[c & 1], p->stream);
                    return e;
                };
                HIP_TRY(copy_chunk(0));
                for (int c = 0; c < nchunks; c++) {
                    if (c + 1 < nchunks) HIP_TRY(copy_chunk(c + 1));
                    HIP_TRY(hipEventSynchronize(p->pin_out_done[c & 1]));
                    const int j0 = c * per_copy, nj = std::min(per_copy, ny - j0);
                    for (int j = 0; j < nj; j++)
                        memcpy(sink.ptrs[a0 + y0 + j0 + j], p->pin_out.p + ((size_t)(c & 1) * per_copy + j) * mb, mb);
                }
                // (the staging buffer is free again: its last copy has been waited for)
            } else if (staged) {
                for (int j = 0; j < ny; j++) {
                    float* dst = sink.ptrs[a0 + y0 + j];
                    HIP_TRY(hipMemcpyAsync(dst, stage.p + (size_t)j * oe, oe * sizeof(float),
                                           sink.location == FFTCONV_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice,
                                           p->stream));
                }
                if (sink.location == FFTCONV_HOST) HIP_TRY(hipStreamSynchronize(p->stream));
            }
        }
    }
    if (streamed) {
        if (prev.valid)
            if (int rc = ring_drain(p, sink, prev.first, prev.count, prev.buf, stage.p + (size_t)prev.buf * nbY * oe)) return rc;
        hipError_t e = p->ring->wait_idle();
        if (e != hipSuccess) return api_fail(FFTCONV_ERR_HIP, "host-output copy failed: %s", hipGetErrorString(e));
    }
    FC_VERBOSE(p, "FFT done");                                        // src/cudaConvolutionFFT.cu:258
    return 0;
}

}  // namespace

namespace fc {

// run_group_impl + on failure: nothing of the host-output ring may still be writing into the
// caller's buffers when the error is returned
int run_group(fftconv_plan* p, int n, const float* dk, int kh, int kw, const Sink& sink) {
    const int rc = run_group_impl(p, n, dk, kh, kw, sink);
    if (rc && p->ring) {
        const std::string keep = g_last_error;
        (void)p->ring->wait_idle();
        g_last_error = keep;
    }
    return rc;
}

int check_thread_size(const double* thread_size, int n_thread_size) {
    // src/cudaConvolutionFFT.cu:72-73 -- the optional argument must have 4 elements
    if (thread_size != nullptr || n_thread_size != 0)
        if (n_thread_size != 4 || thread_size == nullptr)
            return api_fail(FFTCONV_ERR_THREAD_SIZE,
                        "CUDA Thread Size must be 4 integers : THREAD_PER_BLOCK_H, THREAD_PER_BLOCK_W, "
                        "THREAD_PER_BLOCK_D, THREAD_PER_BLOCK_2D");
    return 0;
}

}  // namespace fc

extern "C" {

int fftconv_fft_size16(int data_size) { return fft_size16(data_size); }
int fftconv_fft_size_pow2(int data_size) { return fft_size_pow2(data_size); }

const char* fftconv_last_error(void) { return g_last_error.c_str(); }

const char* fftconv_version(void) { return "fftconv-mi355x 0.1 (gfx950)"; }

int fftconv_device_count(int* count) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (count) *count = (e == hipSuccess) ? n : 0;
    if (e != hipSuccess || n == 0) return api_fail(FFTCONV_ERR_NO_DEVICE, "no HIP device: %s", hipGetErrorString(e));
    return 0;
}

int fftconv_plan_create(fftconv_plan** plan, int data_h, int data_w, int feature_dim, int max_kernel_h,
                        int max_kernel_w, int gpu_id, void* hip_stream) {
    return fftconv_plan_create_ex(plan, data_h, data_w, feature_dim, max_kernel_h, max_kernel_w, gpu_id, hip_stream, nullptr);
}

int fftconv_plan_is_live(const fftconv_plan* plan) {
    std::lock_guard<std::mutex> lk(g_live_mutex);
    return g_live_plans.count(plan) ? 1 : 0;
}

int fftconv_plan_create_ex(fftconv_plan** plan, int data_h, int data_w, int feature_dim, int max_kernel_h,
                           int max_kernel_w, int gpu_id, void* hip_stream, const fftconv_plan_options* options) {
    return plan_create_internal(plan, data_h, data_w, feature_dim, max_kernel_h, max_kernel_w, gpu_id, hip_stream, options, false);
}

}  // extern "C"

// cyclic: the block plan of an overlap-save block-wise plan (PlanTuning::cyclic) -- never block-wise itself
int fc::plan_create_internal(fftconv_plan** plan, int data_h, int data_w, int feature_dim, int max_kernel_h, int max_kernel_w, int gpu_id,
                         void* hip_stream, const fftconv_plan_options* options, bool cyclic) {
    if (!plan) return api_fail(FFTCONV_ERR_INVALID_ARG, "plan is NULL");
    *plan = nullptr;
    if (data_h < 1 || data_w < 1 || feature_dim < 1) return api_fail(FFTCONV_ERR_INVALID_ARG, "Invalid data input");
    if (max_kernel_h < 1 || max_kernel_w < 1) return api_fail(FFTCONV_ERR_INVALID_ARG, "Invalid maximum kernel size");
    int ndev = 0;
    if (int rc = fftconv_device_count(&ndev)) return rc;
    if (gpu_id < 0) HIP_TRY(hipGetDevice(&gpu_id));
    if (gpu_id >= ndev) return api_fail(FFTCONV_ERR_NO_DEVICE, "gpu_id %d out of range (%d devices)", gpu_id, ndev);
    fftconv_plan* p = new (std::nothrow) fftconv_plan();
    if (!p) return api_fail(FFTCONV_ERR_ALLOC, "out of host memory");
    if (options && options->struct_size < kOptionsMinSize) {
        delete p;
        return api_fail(FFTCONV_ERR_INVALID_ARG, "fftconv_plan_options.struct_size is not set");
    }
    PlanTuning tune = tuning_from(options);
    tune.cyclic = cyclic;
    if (cyclic) { tune.exact_window = false; tune.max_transform = 0; }
    p->opt_verbose = options_verbose(options) ? 1 : 0;
    p->gpu_id = gpu_id;
    p->stream = reinterpret_cast<hipStream_t>(hip_stream);
    const bool may_block = !cyclic && !options_no_blockwise(options) && !tune.exact_window;
    bool single_pass = make_geometry(p->g, p->t, data_h, data_w, feature_dim, max_kernel_h, max_kernel_w, tune);
    // large single-pass sizes run on the slower long-transform kernels: blocks of a mid-sized transform (overlap-save, the
    // output kernel storing each block's rectangle of the maps directly) are faster where the cost model says so
    if (single_pass && may_block && blocks_preferred(p->g, options)) single_pass = false;
    if (!single_pass) {
        // too large for one LDS-resident pass (or beyond max_transform), or faster in blocks: a block-wise plan, unless the caller opted out
        int rc = FFTCONV_ERR_UNSUPPORTED_SIZE;
        if (cyclic)
            (void)api_fail(FFTCONV_ERR_UNSUPPORTED_SIZE, "no specialised kernels for a %dx%d block transform with kernels up to %dx%d", data_h, data_w,
                       max_kernel_h, max_kernel_w);
        else if (options_no_blockwise(options) || tune.exact_window)
            (void)api_fail(FFTCONV_ERR_UNSUPPORTED_SIZE, "sizes %dx%dx%d with kernels up to %dx%d do not fit the single-pass LDS transform%s", data_h,
                       data_w, feature_dim, max_kernel_h, max_kernel_w, tune.max_transform > 0 ? " within max_transform" : "");
        else
            rc = tiled_create(p, data_h, data_w, feature_dim, max_kernel_h, max_kernel_w, hip_stream, options);
        if (rc) { delete p; return rc; }
        { std::lock_guard<std::mutex> lk(g_live_mutex); g_live_plans.insert(p); }
        *plan = p;
        return 0;
    }
    int rc = 0;
    do {
        if ((rc = use_device(p))) break;
        hipError_t e = kernels_init();
        if (e != hipSuccess) { rc = api_fail(FFTCONV_ERR_HIP, "kernel setup failed: %s", hipGetErrorString(e)); break; }
        if ((rc = p->tw_m.ensure(p->t.pm.tw.size()))) break;
        if ((rc = p->tw_w.ensure(p->t.pw.tw.size()))) break;
        if ((rc = p->pairs.ensure(p->t.pairs.size()))) break;
        auto cp = [&](void* dst, const void* src, size_t bytes) -> int {
            HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
            return 0;
        };
        if ((rc = cp(p->tw_m.p, p->t.pm.tw.data(), p->t.pm.tw.size() * sizeof(c32)))) break;
        if ((rc = cp(p->tw_w.p, p->t.pw.tw.data(), p->t.pw.tw.size() * sizeof(c32)))) break;
        if ((rc = cp(p->pairs.p, p->t.pairs.data(), p->t.pairs.size() * sizeof(PairEntry)))) break;
        p->d.tw_m = p->tw_m.p;
        p->d.tw_w = p->tw_w.p;
        p->d.pairs = p->pairs.p;
        {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, gpu_id) == hipSuccess && prop.multiProcessorCount > 0)
                p->num_cus = prop.multiProcessorCount;
        }
        if (p->g.fast_cols.ok) {
            const FastColsTables& ft = p->t.fcl;
            if ((rc = p->fc_tw1.ensure(ft.tw1.size()))) break;
            if ((rc = p->fc_tw2.ensure(ft.tw2.size()))) break;
            if ((rc = p->fc_pairs.ensure(ft.pairs.size()))) break;
            if ((rc = p->fc_rowoff.ensure(ft.rowoff.size()))) break;
            if ((rc = cp(p->fc_tw1.p, ft.tw1.data(), ft.tw1.size() * sizeof(c32)))) break;
            if ((rc = cp(p->fc_tw2.p, ft.tw2.data(), ft.tw2.size() * sizeof(c32)))) break;
            if ((rc = cp(p->fc_pairs.p, ft.pairs.data(), ft.pairs.size() * sizeof(PairEntry)))) break;
            if ((rc = cp(p->fc_rowoff.p, ft.rowoff.data(), ft.rowoff.size() * sizeof(int)))) break;
            p->d.fc_tw1 = p->fc_tw1.p;
            p->d.fc_tw2 = p->fc_tw2.p;
            p->d.fc_pairs = p->fc_pairs.p;
            p->d.fc_rowoff = p->fc_rowoff.p;
            if ((rc = p->fc_pair_row_of.ensure(ft.pair_row_of.size()))) break;
            if ((rc = cp(p->fc_pair_row_of.p, ft.pair_row_of.data(), ft.pair_row_of.size() * sizeof(int)))) break;
            p->d.fc_pair_row_of = p->fc_pair_row_of.p;
            // Dynamic tile queue of the persistent OUTPUT kernel (option "dynamic_tiles"): on by default where a tile is long enough
            // for its one-ahead ticket to arrive in time -- M >= 432.  Alone on the GPU it then measures equal or up to 3 % faster
            // than the static deal, beside another kernel it loses half as much (profiles/r05a_contention_ab.txt); on the short
            // tiles of small transforms (M = 336: +18 %, cfg1's M = 144: +4 us a launch) it does not pay
            // (profiles/r05k_dynamic_tiles_by_size.txt).  Zeroed once; every launch leaves the counters at zero.
            if ((rc = p->queue.ensure(FC_QUEUE_WORDS))) break;
            if (hipMemset(p->queue.p, 0, FC_QUEUE_WORDS * sizeof(int)) != hipSuccess) { rc = api_fail(FFTCONV_ERR_HIP, "hipMemset of the tile queue failed"); break; }
            p->opt_dynamic_tiles = p->g.M >= 432 ? 1 : 0;
            p->d.queue = p->opt_dynamic_tiles ? p->queue.p : nullptr;
        }
        if (p->g.fast_rows.ok && p->g.F == 1) {   // resident workgroups per CU of the multi-map row kernel: what rows_group_auto deals over
            FastRowsArgs qa = fast_rows_args(p->g, p->d, nullptr, p->g.max_kw, nullptr, nullptr);
            int per_cu = 0;
            if (fast_rows_multi_wgs_per_cu(p->g.Lw, fast_rows_nz2(p->g, std::min(p->g.max_kw, p->g.fast_rows.max_kw)), qa, &per_cu) == hipSuccess && per_cu > 0)
                p->g.rows_slots_per_cu = per_cu;
            else
                (void)hipGetLastError();
        }
        if (p->g.fast_rows.ok) {
            const FastRowsTables& fr = p->t.fr;
            if ((rc = p->fr_tw1.ensure(fr.tw1.size()))) break;
            if ((rc = p->fr_tw2.ensure(fr.tw2.size()))) break;
            if ((rc = p->fr_map.ensure(fr.relayout.size()))) break;
            if ((rc = cp(p->fr_tw1.p, fr.tw1.data(), fr.tw1.size() * sizeof(c32)))) break;
            if ((rc = cp(p->fr_tw2.p, fr.tw2.data(), fr.tw2.size() * sizeof(c32)))) break;
            if ((rc = cp(p->fr_map.p, fr.relayout.data(), fr.relayout.size() * sizeof(int)))) break;
            p->d.fr_tw1 = p->fr_tw1.p;
            p->d.fr_tw2 = p->fr_tw2.p;
            p->d.fr_relayout = p->fr_map.p;
        }
    } while (0);
    if (rc) {
        p->release_all();
        delete p;
        return rc;
    }
    { std::lock_guard<std::mutex> lk(g_live_mutex); g_live_plans.insert(p); }
    *plan = p;
    return 0;
}

extern "C" {

int fftconv_plan_destroy(fftconv_plan* plan) {
    if (!plan) return 0;
    {
        std::lock_guard<std::mutex> lk(g_live_mutex);
        if (!g_live_plans.erase(plan)) return api_fail(FFTCONV_ERR_INVALID_ARG, "not a live plan");
    }
    (void)hipSetDevice(plan->gpu_id);
    (void)hipStreamSynchronize(plan->stream);
    if (plan->tiled) {
        plan->tiled->release();
        delete plan->tiled;
        plan->tiled = nullptr;
    }
    plan->release_all();
    delete plan;
    return 0;
}

int fftconv_plan_get_info(const fftconv_plan* plan, fftconv_plan_info* info) {
    if (!plan || !info) return api_fail(FFTCONV_ERR_INVALID_ARG, "NULL argument");
    const Geometry& g = plan->g;
    info->data_h = g.H; info->data_w = g.W; info->feature_dim = g.F;
    info->max_kernel_h = g.max_kh; info->max_kernel_w = g.max_kw;
    info->fft_h = g.fft_h; info->fft_w = g.fft_w;
    info->transform_h = g.Lh; info->transform_w = g.Lw;
    info->spectrum_rows = g.rows; info->spectrum_pitch = g.s_pitch;
    info->gpu_id = plan->gpu_id;
    info->exact_window = g.exact_window ? 1 : 0;
    info->spectrum_bytes = g.spectrum_elems() * sizeof(c32);
    info->map_bytes = g.map_elems() * sizeof(float);
    info->out_h = plan->opt_region ? plan->out_h : g.fft_h;
    info->out_w = plan->opt_region ? plan->out_w : g.fft_w;
    info->out_map_bytes = plan->out_elems() * sizeof(float);
    info->workspace_bytes = plan->A.bytes() + plan->Y.bytes() + plan->K.bytes() + plan->O.bytes() + plan->I.bytes();
    if (const TiledState* ts = plan->tiled) {     // block-wise: the window of the whole image; the spectrum is every block's
        info->spectrum_bytes = ts->spec_total() * sizeof(c32);
        info->map_bytes = ts->big_map() * sizeof(float);
        fftconv_plan_info si;
        if (fftconv_plan_get_info(ts->sub, &si) == 0)
            info->workspace_bytes = si.workspace_bytes + ts->big.bytes() + ts->tmp.bytes() + ts->blk.bytes() + ts->crop.bytes() + (ts->specs_x ? 0 : ts->specs.bytes());
    }
    return 0;
}

int fftconv_plan_set_image(fftconv_plan* plan, const float* data, int location) {
    if (!plan || !data) return api_fail(FFTCONV_ERR_INVALID_ARG, "Invalid data input");
    if (location != FFTCONV_HOST && location != FFTCONV_DEVICE) return api_fail(FFTCONV_ERR_INVALID_ARG, "bad location");
    fftconv_plan* p = plan;
    const Geometry& g = p->g;
    if (int rc = use_device(p)) return rc;
    if (p->tiled) return tiled_set_image(p, data, location);
    const float* dimg = data;
    bool image_pinned = false;      // the caller's array has been consumed by the CPU: no wait for the GPU at the end
    if (location == FFTCONV_HOST) {
        const size_t n = (size_t)g.H * g.W * g.F;
        if (p->opt_host_pinned && n * sizeof(float) <= FC_PIN_IMAGE_BYTES) {
            if (int rc = p->pin_img.ensure(n * sizeof(float))) return rc;
            if (int rc = p->pin_img.wait()) return rc;
            memcpy(p->pin_img.p, data, n * sizeof(float));
            image_pinned = true;
            if (n * sizeof(float) <= FC_PIN_INPLACE_BYTES) {
                dimg = reinterpret_cast<const float*>(p->pin_img.p);      // the column pass reads it in place
            } else {
                if (int rc = p->I.ensure(n)) return rc;
                HIP_TRY(hipMemcpyAsync(p->I.p, p->pin_img.p, n * sizeof(float), hipMemcpyHostToDevice, p->stream));
                dimg = p->I.p;
            }
        } else {
            if (int rc = p->I.ensure(n)) return rc;
            HIP_TRY(hipMemcpyAsync(p->I.p, data, n * sizeof(float), hipMemcpyHostToDevice, p->stream));
            dimg = p->I.p;
        }
    }
    FC_VERBOSE(p, "Using GPU : %d", p->gpu_id);                                                    // src/cudaConvolutionFFT.cu:87
    FC_VERBOSE(p, "Data size: h=%d, w=%d, f=%d", g.H, g.W, g.F);                                   // :100
    FC_VERBOSE(p, "FFT size: h=%d, w=%d (internal transform %d x %d, %s column pass, %s row pass, %s intermediate)", g.fft_h, g.fft_w, g.Lh, g.Lw,   // :114
               g.fast_fwd ? "specialised" : "generic", g.fast_rows.ok ? "specialised" : "generic", g.y_tiled() ? "tiled" : "row-major");
    // (with the fast row kernel the w-pass stores the spectrum directly in that kernel's register order)
    if (int rc = p->ensure_spectrum()) return rc;
    c32* sgen = p->spec();
    if (p->deferred.on && p->deferred.stream != p->stream)
        if (int rc = flush_pending_prepare(p)) return rc;
    if (int rc = p->prof_begin(PK_IMAGE_COLS, g.F)) return rc;
    if (g.fast_fwd && p->deferred.on) {
        // the deferred kernel-column pass of fftconv_plan_prepare_kernels_packed and the image's column pass: ONE launch
        const auto pd = p->deferred;
        p->deferred.on = false;
        FastColsFwdArgs fa = fast_cols_fwd_args(g, p->d, dimg, (size_t)g.H * g.W, g.H, g.H, g.W, g.F, sgen,
                                                (size_t)g.rows * g.s_pitch, g.s_pitch, false);
        FastColsFwdArgs fk = fast_cols_fwd_args(g, p->d, pd.dk, (size_t)pd.kh * pd.kw, pd.kh, pd.kh, pd.kw, pd.na * g.F, p->A.p,
                                                (size_t)g.rows * a_pitch_for(pd.kw), a_pitch_for(pd.kw), true);
        FC_VERBOSE(p, "image columns (%d tiles) and the columns of %d kernels (%d tiles) in one launch", fa.ntiles, pd.na, fk.ntiles);
        HIP_TRY(launch_fast_cols_fwd_pair(g.M, g.fast_cols.T, fa, fk, fast_cols_fwd_pruned_ok(g.fast_cols, pd.kh), p->num_cus, p->stream));
        p->prepared.dk = pd.dk; p->prepared.n = pd.n; p->prepared.kh = pd.kh; p->prepared.kw = pd.kw; p->prepared.stream = p->stream;
    } else if (g.fast_fwd) {
        FastColsFwdArgs fa = fast_cols_fwd_args(g, p->d, dimg, (size_t)g.H * g.W, g.H, g.H, g.W, g.F, sgen,
                                                (size_t)g.rows * g.s_pitch, g.s_pitch, false);
        HIP_TRY(launch_fast_cols_fwd(g.M, g.fast_cols.T, false, fa, p->num_cus, p->stream));
    } else {
        ColsR2CArgs ia = image_cols_args(g, p->t, p->d, dimg, sgen);
        HIP_TRY(launch_cols_r2c(ia, tiles_for(g.W, g.T_cols), g.F, cols_threads(g), p->cols_lds(), p->stream));
    }
    if (int rc = p->prof_end()) return rc;
    if (image_pinned)
        if (int rc = p->pin_img.mark(p->stream)) return rc;      // the column pass (or the copy) is the last reader
    if (int rc = p->prof_begin(PK_IMAGE_ROWS, g.F)) return rc;
    if (g.fast_rows.ok) {
        HIP_TRY(launch_fast_rows_fwd(g.Lw, fast_rows_fwd_args(g, p->d, sgen), g.F * g.rows, p->stream));
    } else {
        RowsFwdArgs ra = image_rows_args(g, p->t, p->d, sgen);
        HIP_TRY(launch_rows_fwd(ra, g.F * g.rows, rows_threads(g), (size_t)g.Lw * sizeof(c32), p->stream));
    }
    if (int rc = p->prof_end()) return rc;
    if (location == FFTCONV_HOST && !image_pinned) HIP_TRY(hipStreamSynchronize(p->stream));
    p->have_image = true;
    return 0;
}

int fftconv_plan_spectrum(fftconv_plan* plan, void** device_ptr, size_t* bytes) {
    if (!plan) return api_fail(FFTCONV_ERR_INVALID_ARG, "plan is NULL");
    if (int rc = use_device(plan)) return rc;
    if (TiledState* ts = plan->tiled) {          // every block's spectrum, one after the other
        if (!ts->specs_x)
            if (int rc = ts->specs.ensure(ts->spec_total())) return rc;
        if (device_ptr) *device_ptr = ts->spec_base();
        if (bytes) *bytes = ts->spec_total() * sizeof(c32);
        return 0;
    }
    if (int rc = plan->ensure_spectrum()) return rc;
    if (device_ptr) *device_ptr = plan->spec();
    if (bytes) *bytes = plan->g.spectrum_elems() * sizeof(c32);
    return 0;
}

static int spectrum_exchange(fftconv_plan* p, float* spectrum, int location, bool to_natural) {
    if (!p || !spectrum) return api_fail(FFTCONV_ERR_INVALID_ARG, "NULL argument");
    if (location != FFTCONV_HOST && location != FFTCONV_DEVICE) return api_fail(FFTCONV_ERR_INVALID_ARG, "bad location");
    if (p->tiled) return tiled_unsupported("the spectrum in the reference's order");
    const Geometry& g = p->g;
    if (!g.exact_window)
        return api_fail(FFTCONV_ERR_UNSUPPORTED_SIZE,
                    "this plan transforms %dx%d, not the %dx%d window: create it with fftconv_plan_options.exact_window = 1 to exchange "
                    "spectra in the reference's order", g.Lh, g.Lw, g.fft_h, g.fft_w);
    if (to_natural && !p->have_image) return api_fail(FFTCONV_ERR_NO_IMAGE, "no image spectrum: call fftconv_plan_set_image first");
    if (int rc = use_device(p)) return rc;
    if (int rc = p->ensure_spectrum()) return rc;
    if (!p->nat_row_of.p) {
        if (int rc = p->nat_row_of.ensure(p->t.nat_row_of.size())) return rc;
        if (int rc = p->nat_col_of.ensure(p->t.nat_col_of.size())) return rc;
        HIP_TRY(hipMemcpy(p->nat_row_of.p, p->t.nat_row_of.data(), p->t.nat_row_of.size() * sizeof(int), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(p->nat_col_of.p, p->t.nat_col_of.data(), p->t.nat_col_of.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    const size_t n = (size_t)g.F * g.fft_w * g.rows;
    c32* nat = reinterpret_cast<c32*>(spectrum);
    if (location == FFTCONV_HOST) {
        if (int rc = p->NS.ensure(n)) return rc;
        nat = p->NS.p;
        if (!to_natural) HIP_TRY(hipMemcpyAsync(nat, spectrum, n * sizeof(c32), hipMemcpyHostToDevice, p->stream));
    }
    // the folded 1/(Lh*Lw) of src/cudaConvolutionFFT.cu:270 comes off on the way out and goes on on the way in
    const double norm = (double)g.Lh * (double)g.Lw;
    HIP_TRY(launch_spectrum_reorder(to_natural, p->spec(), (size_t)g.rows * g.s_pitch, g.s_pitch, nat, g.fft_w, g.rows, g.F,
                                    p->nat_row_of.p, p->nat_col_of.p, (float)(to_natural ? norm : 1.0 / norm), p->stream));
    if (location == FFTCONV_HOST) {
        if (to_natural) HIP_TRY(hipMemcpyAsync(spectrum, nat, n * sizeof(c32), hipMemcpyDeviceToHost, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
    }
    if (!to_natural) p->have_image = true;
    return 0;
}

int fftconv_plan_export_spectrum(fftconv_plan* plan, float* spectrum, int location) { return spectrum_exchange(plan, spectrum, location, true); }
int fftconv_plan_import_spectrum(fftconv_plan* plan, const float* spectrum, int location) {
    return spectrum_exchange(plan, const_cast<float*>(spectrum), location, false);
}

int fftconv_plan_use_spectrum_buffer(fftconv_plan* plan, void* device_ptr, size_t bytes) {
    if (!plan) return api_fail(FFTCONV_ERR_INVALID_ARG, "plan is NULL");
    const size_t need = (plan->tiled ? plan->tiled->spec_total() : plan->g.spectrum_elems()) * sizeof(c32);
    if (device_ptr) {
        if (bytes < need || (reinterpret_cast<uintptr_t>(device_ptr) & 15))
            return api_fail(FFTCONV_ERR_INVALID_ARG, "spectrum buffer too small (%zu < %zu bytes) or not 16-byte aligned", bytes, need);
    }
    if (plan->tiled) {
        plan->tiled->specs_x = reinterpret_cast<c32*>(device_ptr);
        plan->tiled->have_image = false;
        return 0;
    }
    plan->Sx = reinterpret_cast<c32*>(device_ptr);
    plan->have_image = false;
    return 0;
}

int fftconv_plan_mark_spectrum_valid(fftconv_plan* plan) {
    if (!plan) return api_fail(FFTCONV_ERR_INVALID_ARG, "plan is NULL");
    if (plan->tiled) {
        if (!plan->tiled->spec_base()) return api_fail(FFTCONV_ERR_NO_IMAGE, "the plan has no spectrum buffer yet (fftconv_plan_spectrum / fftconv_plan_use_spectrum_buffer)");
        plan->tiled->have_image = true;
        return 0;
    }
    if (!plan->spec()) return api_fail(FFTCONV_ERR_NO_IMAGE, "the plan has no spectrum buffer yet (fftconv_plan_spectrum / fftconv_plan_use_spectrum_buffer)");
    plan->have_image = true;
    return 0;
}

int fftconv_plan_convolve_packed(fftconv_plan* plan, int n_kernel, const float* kernels_device, int kernel_h,
                                 int kernel_w, float* out_device) {
    if (!plan || n_kernel < 0) return api_fail(FFTCONV_ERR_INVALID_ARG, "Wrong number of inputs");
    if (n_kernel == 0) return 0;
    if (!kernels_device || !out_device) return api_fail(FFTCONV_ERR_INVALID_ARG, "NULL kernel or output pointer");
    if (int rc = use_device(plan)) return rc;
    if (plan->tiled) {
        const size_t per = (size_t)plan->tiled->F * kernel_h * kernel_w;
        std::vector<const float*> kp(n_kernel);
        std::vector<int> khs(n_kernel, kernel_h), kws(n_kernel, kernel_w);
        for (int j = 0; j < n_kernel; j++) kp[j] = kernels_device + per * j;
        const int rc = tiled_convolve(plan, n_kernel, kp.data(), khs.data(), kws.data(), FFTCONV_DEVICE, nullptr, FFTCONV_DEVICE, out_device);
        if (rc) {     // as fftconv_plan_convolve: nothing of a failed call may still be running when the error is returned
            const std::string keep = g_last_error;
            (void)hipStreamSynchronize(plan->tiled->sub->stream);
            g_last_error = keep;
        }
        return rc;
    }
    Sink sink;
    sink.packed = out_device;
    return run_group(plan, n_kernel, kernels_device, kernel_h, kernel_w, sink);
}

int fftconv_plan_prepare_kernels_packed(fftconv_plan* plan, int n_kernel, const float* kernels_device, int kernel_h,
                                        int kernel_w) {
    if (!plan || n_kernel < 0) return api_fail(FFTCONV_ERR_INVALID_ARG, "Wrong number of inputs");
    if (n_kernel == 0) return 0;
    if (!kernels_device) return api_fail(FFTCONV_ERR_INVALID_ARG, "NULL kernel pointer");
    fftconv_plan* p = plan;
    if (p->tiled) return 0;        // block-wise: the kernels are transformed per block inside convolve
    if (int rc = use_device(p)) return rc;
    if (int rc = check_kernel_size(p, kernel_h, kernel_w)) return rc;
    const BatchSizes bs = batch_sizes(p, n_kernel, kernel_w);
    if (int rc = flush_pending_prepare(p)) return rc;      // an earlier request nobody consumed
    if (int rc = p->A.ensure(bs.per_a * bs.nbA)) return rc;
    p->prepared.dk = nullptr;
    const bool timed_apart = p->profile && (p->profile_mask & ((1u << PK_KERNEL_COLS) | (1u << PK_IMAGE_COLS)));   // per-kind figures wanted
    if (p->opt_defer_prepare && p->g.fast_fwd && !p->opt_flip_kernels && !timed_apart) {   // deferred: rides in the launch of the next image's column pass
        p->deferred.on = true; p->deferred.dk = kernels_device; p->deferred.n = n_kernel; p->deferred.na = std::min(bs.nbA, n_kernel);
        p->deferred.kh = kernel_h; p->deferred.kw = kernel_w; p->deferred.stream = p->stream;
        return 0;
    }
    if (int rc = launch_kernel_cols(p, kernels_device, 0, std::min(bs.nbA, n_kernel), kernel_h, kernel_w)) return rc;
    p->prepared.dk = kernels_device; p->prepared.n = n_kernel; p->prepared.kh = kernel_h; p->prepared.kw = kernel_w;
    p->prepared.stream = p->stream;
    return 0;
}

int fftconv_plan_convolve(fftconv_plan* plan, int n_kernel, const float* const* kernels, const int* kernel_h,
                          const int* kernel_w, int kernel_location, float* const* out, int out_location) {
    if (!plan || n_kernel < 0) return api_fail(FFTCONV_ERR_INVALID_ARG, "Wrong number of inputs");
    if (n_kernel == 0) return 0;
    if (!kernels || !kernel_h || !kernel_w || !out) return api_fail(FFTCONV_ERR_INVALID_ARG, "Kernel must be a cell array");
    fftconv_plan* p = plan;
    const Geometry& g = p->g;
    if (int rc = use_device(p)) return rc;
    if (p->tiled) {
        if (int rc = tiled_convolve(p, n_kernel, kernels, kernel_h, kernel_w, kernel_location, out, out_location, nullptr)) {
            const std::string keep = g_last_error;
            (void)hipStreamSynchronize(p->tiled->sub->stream);   // nothing may still be writing into the caller's buffers
            g_last_error = keep;
            return rc;
        }
        return 0;
    }
    if (!p->have_image) return api_fail(FFTCONV_ERR_NO_IMAGE, "no image spectrum: call fftconv_plan_set_image first");
    // validate everything up front so nothing is launched on a bad cell (the reference fails
    // mid-loop and leaks: SURVEY D3)
    for (int k = 0; k < n_kernel; k++) {
        if (!kernels[k] || !out[k]) return api_fail(FFTCONV_ERR_INVALID_ARG, "kernel or output %d is NULL", k);
        if (int rc = check_kernel_size(p, kernel_h[k], kernel_w[k])) return rc;
    }
    // groups of consecutive kernels of equal size
    int k0 = 0;
    while (k0 < n_kernel) {
        int k1 = k0 + 1;
        while (k1 < n_kernel && kernel_h[k1] == kernel_h[k0] && kernel_w[k1] == kernel_w[k0]) k1++;
        const int n = k1 - k0, kh = kernel_h[k0], kw = kernel_w[k0];
        const size_t per = (size_t)g.F * kh * kw;
        const float* dk = nullptr;
        const bool kernels_pinned = kernel_location == FFTCONV_HOST && p->opt_host_pinned && per * n * sizeof(float) <= FC_PIN_INPLACE_BYTES;
        if (kernels_pinned) {
            // a small set of host kernels: gathered by the CPU into the plan's pinned buffer and read there, in place, by
            // the kernels' column pass (one pass over them) -- no copy command, against one blocking copy per kernel
            if (int rc = p->pin_k.ensure(per * n * sizeof(float))) return rc;
            if (int rc = p->pin_k.wait()) return rc;
            for (int j = 0; j < n; j++) memcpy(p->pin_k.p + per * j * sizeof(float), kernels[k0 + j], per * sizeof(float));
            dk = reinterpret_cast<const float*>(p->pin_k.p);
            if (p->prepared.dk == dk) p->prepared.dk = nullptr;      // column spectra of the buffer's previous contents
            if (p->deferred.on && p->deferred.dk == dk) p->deferred.on = false;
        } else {
            if (int rc = p->K.ensure(per * n)) return rc;
            for (int j = 0; j < n; j++)
                HIP_TRY(hipMemcpyAsync(p->K.p + per * j, kernels[k0 + j], per * sizeof(float),
                                       kernel_location == FFTCONV_HOST ? hipMemcpyHostToDevice
                                       : kernel_location == FFTCONV_AUTO ? hipMemcpyDefault : hipMemcpyDeviceToDevice,
                                       p->stream));
            dk = p->K.p;
        }
        Sink sink;
        sink.ptrs = out + k0;
        sink.location = out_location;
        const int rcg = run_group(p, n, dk, kh, kw, sink);
        if (kernels_pinned) {
            const std::string keep = g_last_error;
            const int rcm = p->pin_k.mark(p->stream);
            if (rcg) g_last_error = keep;
            else if (rcm) return rcm;
        }
        if (rcg) return rcg;
        k0 = k1;
    }
    if (out_location == FFTCONV_HOST) HIP_TRY(hipStreamSynchronize(p->stream));
    return 0;
}

int fftconv_plan_set_stream(fftconv_plan* plan, void* hip_stream) {
    if (!plan) return api_fail(FFTCONV_ERR_INVALID_ARG, "plan is NULL");
    // (profile events already recorded stay valid: they are read later by fftconv_plan_get_profile,
    //  whichever stream they were recorded on -- collecting them here would block the host)
    // (prepared column spectra stay: they count again once the plan is back on the stream they were produced on --
    //  the image transform of the multi-GPU step borrows the plan for a side stream and hands it back; a DEFERRED
    //  preparation is launched now, on the stream it was asked for)
    if (!plan->tiled && plan->deferred.on) {
        if (int rc = use_device(plan)) return rc;
        if (int rc = flush_pending_prepare(plan)) return rc;
    }
    plan->stream = reinterpret_cast<hipStream_t>(hip_stream);
    if (plan->tiled) return fftconv_plan_set_stream(plan->tiled->sub, hip_stream);
    return 0;
}

int fftconv_plan_synchronize(fftconv_plan* plan) {
    if (!plan) return api_fail(FFTCONV_ERR_INVALID_ARG, "plan is NULL");
    if (int rc = use_device(plan)) return rc;
    if (!plan->tiled)
        if (int rc = flush_pending_prepare(plan)) return rc;
    HIP_TRY(hipStreamSynchronize(plan->stream));
    return 0;
}

int fftconv_plan_set_option(fftconv_plan* plan, const char* name, long value) {
    if (!plan || !name) return api_fail(FFTCONV_ERR_INVALID_ARG, "NULL argument");
    if (plan->tiled && strcmp(name, "output_region")) {      // block-wise: options act on the block plan ("output_region": on this one, below)
        if (!strcmp(name, "verbose")) plan->opt_verbose = value != 0;
        return fftconv_plan_set_option(plan->tiled->sub, name, value);
    }
    if (!strcmp(name, "tune_placement")) { plan->opt_tune_placement = value < 0 ? -1 : (value > 8 ? 8 : value); return 0; }
    if (!strcmp(name, "kernel_chunk_mb")) { plan->opt_kernel_chunk_mb = value < 0 ? 0 : value; plan->prepared.dk = nullptr; plan->deferred.on = false; return 0; }
    if (!strcmp(name, "batch_maps")) { plan->opt_batch_maps = value < 0 ? 0 : value; plan->prepared.dk = nullptr; plan->deferred.on = false; return 0; }
    if (!strcmp(name, "profile")) {
        if (value != 0 && plan->deferred.on && (plan->profile_mask & ((1u << PK_KERNEL_COLS) | (1u << PK_IMAGE_COLS)))) {   // per-kind figures: the kernels' column pass is timed as a launch of its own
            if (int rc = use_device(plan)) return rc;
            if (int rc = flush_pending_prepare(plan)) return rc;
        }
        plan->profile = value != 0;
        return 0;
    }
    if (!strcmp(name, "verbose")) { plan->opt_verbose = value != 0; return 0; }
    if (!strcmp(name, "dynamic_tiles")) {
        // 1: the persistent output kernel takes its tiles from a queue in device memory (fast_cols.hpp: TileQueue; the default
        // from M = 432 on) instead of a fixed share per workgroup (0); 2: the forward column kernels (image, kernels) too.  A step
        // that shares the GPU with another kernel -- the broadcast of a multi-GPU run, another library's work -- loses half as much
        // with the queue.
        if (value < 0 || value > 2) return api_fail(FFTCONV_ERR_INVALID_ARG, "dynamic_tiles is 0, 1 or 2");
        if (value && !plan->g.fast_cols.ok) value = 0;      // (generic kernels: one workgroup per tile, dealt by the hardware)
        if (value && !plan->queue.p) {     // zeroed once: every launch leaves the counters at zero (fast_cols.hpp: queue_leave)
            if (int rc = use_device(plan)) return rc;
            if (int rc = plan->queue.ensure(FC_QUEUE_WORDS)) return rc;
            HIP_TRY(hipMemset(plan->queue.p, 0, FC_QUEUE_WORDS * sizeof(int)));
        }
        plan->opt_dynamic_tiles = value;
        plan->d.queue = value ? plan->queue.p : nullptr;
        plan->d.queue_fwd = value == 2;
        return 0;
    }
    if (!strcmp(name, "defer_prepare")) {
        if (!value && plan->deferred.on) {
            if (int rc = use_device(plan)) return rc;
            if (int rc = flush_pending_prepare(plan)) return rc;
        }
        plan->opt_defer_prepare = value != 0;
        return 0;
    }
    if (!strcmp(name, "profile_kinds")) { plan->profile_mask = value <= 0 ? ~0u : (unsigned)value; return 0; }
    if (!strcmp(name, "rows_group")) { plan->g.rows_group = value <= 0 ? -1 : (int)value; return 0; }
#if FC_ROWS_TIMELINE || FC_COLS_TIMELINE || FC_ROWS_STAGGER_TICKS
    // diagnostic builds only (tools/rows_timeline.py, tools/cols_timeline.py): device buffer one workgroup stamps
    if (!strcmp(name, "timeline_ptr")) { plan->d.timeline = reinterpret_cast<unsigned long long*>((uintptr_t)value); return 0; }
#endif
    if (!strcmp(name, "output_region")) {
        const Geometry& g = plan->g;
        int oh = g.fft_h, ow = g.fft_w, fh = 0, fw = 0;
        if (value == 1) { oh = g.H + g.max_kh - 1; ow = g.W + g.max_kw - 1; }
        else if (value == 2) { oh = g.H; ow = g.W; fh = (g.max_kh - 1) / 2; fw = (g.max_kw - 1) / 2; }
        else if (value == 3) { oh = g.H - g.max_kh + 1; ow = g.W - g.max_kw + 1; fh = g.max_kh - 1; fw = g.max_kw - 1; }
        else if (value == 4) { oh = fft_size_pow2(g.H + g.max_kh - 1); ow = fft_size_pow2(g.W + g.max_kw - 1); }
        else if (value != 0) return api_fail(FFTCONV_ERR_INVALID_ARG, "output_region is 0 (window), 1 (full), 2 (same), 3 (valid) or 4 (pow2 window)");
        if (oh < 1 || ow < 1) return api_fail(FFTCONV_ERR_INVALID_ARG, "output_region %ld is empty for %dx%d data and %dx%d kernels", value, g.H, g.W, g.max_kh, g.max_kw);
        if (int rc = use_device(plan)) return rc;
        HIP_TRY(hipStreamSynchronize(plan->stream));
        plan->release_ring();          // sized for the map bytes
        // (block-wise plans: g is the whole image's geometry; the blocks keep storing full-window maps and the region is cropped
        //  out of them on delivery, blockwise.cpp: tiled_deliver)
        plan->opt_region = value; plan->out_h = oh; plan->out_w = ow; plan->off_h = fh; plan->off_w = fw;
        return 0;
    }
    if (!strcmp(name, "flip_kernels")) { plan->opt_flip_kernels = value != 0; plan->prepared.dk = nullptr; plan->deferred.on = false; return 0; }
    if (!strcmp(name, "host_pinned")) { plan->opt_host_pinned = value != 0; return 0; }
    if (!strcmp(name, "host_min_kb")) {
        if (value < 0 || value > (1 << 30)) return api_fail(FFTCONV_ERR_INVALID_ARG, "option '%s' out of range", name);
        plan->opt_host_min_kb = value;
        return 0;
    }
    if (!strcmp(name, "host_stream") || !strcmp(name, "host_threads") || !strcmp(name, "host_chunk_kb") || !strcmp(name, "host_slots")) {
        if (value < 0 || value > (1 << 20)) return api_fail(FFTCONV_ERR_INVALID_ARG, "option '%s' out of range", name);
        if (int rc = use_device(plan)) return rc;
        HIP_TRY(hipStreamSynchronize(plan->stream));
        plan->release_ring();      // rebuilt with the new shape by the next host-output call
        if (!strcmp(name, "host_stream") && value > 2) return api_fail(FFTCONV_ERR_INVALID_ARG, "host_stream is 0, 1 or 2");
        (!strcmp(name, "host_stream") ? plan->opt_host_stream : name[5] == 't' ? plan->opt_host_threads
         : name[5] == 'c' ? plan->opt_host_chunk_kb : plan->opt_host_slots) = value;
        return 0;
    }
    return api_fail(FFTCONV_ERR_INVALID_ARG, "unknown option '%s'", name);
}

int fftconv_plan_get_option(fftconv_plan* plan, const char* name, long* value) {
    if (!plan || !name || !value) return api_fail(FFTCONV_ERR_INVALID_ARG, "null argument");
    if (!strcmp(name, "blockwise")) { *value = plan->tiled ? plan->tiled->nblk : 0; return 0; }   // read-only: number of blocks (0 = one pass)
    if (!strcmp(name, "overlap_save")) { *value = plan->tiled && plan->tiled->save ? 1 : 0; return 0; }   // read-only: blocks stored by the output kernel (1) or summed (0)
    if (plan->tiled && strcmp(name, "output_region")) return fftconv_plan_get_option(plan->tiled->sub, name, value);
    if (!strcmp(name, "batch_maps")) { *value = plan->opt_batch_maps; return 0; }
    if (!strcmp(name, "kernel_chunk_mb")) { *value = plan->opt_kernel_chunk_mb; return 0; }
    if (!strcmp(name, "tune_placement")) { *value = plan->opt_tune_placement; return 0; }
    if (!strcmp(name, "tuned_candidates")) { *value = plan->tuned_candidates; return 0; }
    if (!strcmp(name, "tuned_best")) { *value = plan->tuned_best; return 0; }
    if (!strcmp(name, "rows_group")) { *value = plan->g.rows_group; return 0; }
    // read-only: which passes of this plan run on specialised (compile-time) kernels: bit 0 the spectral rows (w), bit 1 the
    // column passes (h: image / kernel columns forwards, output columns); 3 = no generic kernel runs
    if (!strcmp(name, "specialised_kernels")) { *value = (plan->g.fast_rows.ok ? 1 : 0) | (plan->g.fast_cols.ok ? 2 : 0); return 0; }
    if (!strcmp(name, "rows_slots_per_cu")) { *value = plan->g.rows_slots_per_cu; return 0; }   // read-only: resident row workgroups per CU
    if (!strcmp(name, "host_stream")) { *value = plan->opt_host_stream; return 0; }
    if (!strcmp(name, "host_min_kb")) { *value = plan->opt_host_min_kb; return 0; }
    if (!strcmp(name, "host_pinned")) { *value = plan->opt_host_pinned; return 0; }
    if (!strcmp(name, "output_region")) { *value = plan->opt_region; return 0; }
    if (!strcmp(name, "flip_kernels")) { *value = plan->opt_flip_kernels; return 0; }
    if (!strcmp(name, "profile")) { *value = plan->profile ? 1 : 0; return 0; }
    if (!strcmp(name, "verbose")) { *value = plan->opt_verbose; return 0; }
    if (!strcmp(name, "dynamic_tiles")) { *value = plan->opt_dynamic_tiles; return 0; }
    if (!strcmp(name, "defer_prepare")) { *value = plan->opt_defer_prepare; return 0; }
    if (!strcmp(name, "prepare_pending")) { *value = plan->deferred.on ? 1 : 0; return 0; }   // read-only: a recorded, not yet launched preparation
    return api_fail(FFTCONV_ERR_INVALID_ARG, "unknown option '%s'", name);
}

int fftconv_plan_get_profile(fftconv_plan* plan, fftconv_profile* prof, int reset) {
    if (!plan || !prof) return api_fail(FFTCONV_ERR_INVALID_ARG, "NULL argument");
    if (plan->tiled) return fftconv_plan_get_profile(plan->tiled->sub, prof, reset);
    if (int rc = use_device(plan)) return rc;
    if (int rc = plan->prof_collect()) return rc;
    for (int i = 0; i < PK_COUNT; i++) {
        prof->ms[i] = plan->prof_ms[i];
        prof->launches[i] = plan->prof_launches[i];
        prof->units[i] = plan->prof_units[i];
        if (reset) { plan->prof_ms[i] = 0; plan->prof_launches[i] = 0; plan->prof_units[i] = 0; }
    }
    return 0;
}

int fftconv_fft_data(const float* data, int data_h, int data_w, int feature_dim, int kernel_h, int kernel_w, int gpu_id,
                     fftconv_plan** fft_data) {
    if (!fft_data) return api_fail(FFTCONV_ERR_INVALID_ARG, "Invalid input to MEX file.");
    *fft_data = nullptr;
    if (!data) return api_fail(FFTCONV_ERR_INVALID_ARG, "Invalid input to MEX file.");  // src/cudaFFTData.cu:49-54
    fftconv_plan* p = nullptr;
    if (int rc = fftconv_plan_create(&p, data_h, data_w, feature_dim, kernel_h, kernel_w, gpu_id, nullptr)) return rc;
    if (int rc = fftconv_plan_set_image(p, data, FFTCONV_HOST)) {
        std::string keep = g_last_error;
        fftconv_plan_destroy(p);
        g_last_error = keep;
        return rc;
    }
    *fft_data = p;
    return 0;
}

int fftconv_conv_fft_data(fftconv_plan* fft_data, int n_kernel, const float* const* kernels, const int* kernel_h,
                          const int* kernel_w, const int* kernel_f, const double* thread_size, int n_thread_size,
                          float* const* out) {
    if (!fft_data || !fftconv_plan_is_live(fft_data)) return api_fail(FFTCONV_ERR_INVALID_ARG, "Invalid input to MEX file.");  // src/cudaConvFFTData.cu:68
    if (int rc = check_thread_size(thread_size, n_thread_size)) return rc;
    if (kernel_f)
        for (int k = 0; k < n_kernel; k++)
            if (kernel_f[k] != fft_data->g.F)
                return api_fail(FFTCONV_ERR_KERNEL_SHAPE,
                            "Kernel and Data must have the same number of features and kernel size should be smaller than data size");
    return fftconv_plan_convolve(fft_data, n_kernel, kernels, kernel_h, kernel_w, FFTCONV_AUTO, out, FFTCONV_HOST);
}

}  // extern "C"
