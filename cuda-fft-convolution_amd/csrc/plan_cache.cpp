// plan_cache.cpp -- the one-shot entries of include/fftconv.h (the MEX body, src/cudaConvolutionFFT.cu:27-311) and the
// process-wide plan cache behind them.
#include <chrono>
#include <mutex>

#include "plan_internal.hpp"

namespace {

// ---------------------------------------------------------------------------------------------------------
// Plan cache of the one-shot entries (include/fftconv.h).  The reference pays plan creation, six cudaMallocs and
// the teardown in every MEX call (src/cudaConvolutionFFT.cu:127-142,144-185,302-310); a cached plan keeps its
// tables, its device scratch (sized by the first call) and its host copy threads.  The cache object is never
// destroyed (plans own joinable threads and HIP objects: nothing of that may run from a static destructor at
// process exit, after the HIP runtime has gone) -- fftconv_cache_clear() is the release.
// ---------------------------------------------------------------------------------------------------------
struct CacheKey {
    int H, W, F, mkh, mkw, gpu;
    int kernel_path, rows_group, max_transform, exact_window, blockwise;
    bool operator==(const CacheKey& o) const {
        return H == o.H && W == o.W && F == o.F && mkh == o.mkh && mkw == o.mkw && gpu == o.gpu && kernel_path == o.kernel_path &&
               rows_group == o.rows_group && max_transform == o.max_transform && exact_window == o.exact_window && blockwise == o.blockwise;
    }
};
struct CacheEntry {
    CacheKey key;
    fftconv_plan* plan;
    unsigned long stamp;
    size_t bytes;
};
struct PlanCache {
    std::mutex m;
    std::vector<CacheEntry> idle;     // plans no call is using (a plan in use is simply not in here)
    int max_plans = 4;
    // device scratch of the idle plans together: a quarter of the device's memory, at most 48 GiB (set at the first put: needs
    // the device); 0 = not yet known.  Whatever the limit, idle plans are released when an allocation fails (cache_release_idle).
    size_t max_bytes = 0;
    unsigned long clock = 0;
    long hits = 0, misses = 0;
};
PlanCache& plan_cache() {
    static PlanCache* c = new PlanCache();
    return *c;
}

CacheKey cache_key(int H, int W, int F, int mkh, int mkw, int gpu, const fftconv_plan_options* o) {
    CacheKey k{H, W, F, mkh, mkw, gpu, 0, 0, 0, 0, 0};
    if (o && o->struct_size >= kOptionsMinSize) {
        k.kernel_path = o->kernel_path; k.rows_group = o->rows_group <= 0 ? 0 : o->rows_group;
        k.max_transform = o->max_transform > 0 ? o->max_transform : 0; k.exact_window = o->exact_window != 0;
        k.blockwise = options_no_blockwise(o) ? 1 : 0;
    }
    return k;
}

size_t plan_device_bytes(const fftconv_plan* p) {
    size_t b = p->tw_m.bytes() + p->tw_w.bytes() + p->pairs.bytes() + p->S.bytes() + p->A.bytes() + p->Y.bytes() + p->K.bytes() + p->KF.bytes() +
               p->O.bytes() + p->OC.bytes() + p->I.bytes() + p->NS.bytes();
    if (const TiledState* ts = p->tiled) {
        b += ts->specs.bytes() + ts->big.bytes() + ts->tmp.bytes() + ts->blk.bytes() + ts->kstage.bytes() + ts->crop.bytes();
        if (ts->sub) b += plan_device_bytes(ts->sub);
    }
    return b;
}

// a cached idle plan for this key, or nullptr
fftconv_plan* cache_take(const CacheKey& key) {
    PlanCache& c = plan_cache();
    std::lock_guard<std::mutex> lk(c.m);
    for (size_t i = 0; i < c.idle.size(); i++)
        if (c.idle[i].key == key) {
            fftconv_plan* p = c.idle[i].plan;
            c.idle.erase(c.idle.begin() + (long)i);
            c.hits++;
            return p;
        }
    c.misses++;
    return nullptr;
}

// hand a plan (back) to the cache; plans pushed out by the limits are destroyed (outside the lock)
void cache_put(const CacheKey& key, fftconv_plan* p) {
    PlanCache& c = plan_cache();
    std::vector<fftconv_plan*> drop;
    {
        std::lock_guard<std::mutex> lk(c.m);
        if (c.max_plans <= 0) drop.push_back(p);
        else {
            if (c.max_bytes == 0) {
                size_t free_b = 0, total_b = 0;
                if (hipSetDevice(p->gpu_id) != hipSuccess || hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); total_b = 0; }
                c.max_bytes = total_b ? std::min<size_t>((size_t)48 << 30, total_b / 4) : (size_t)48 << 30;
            }
            c.idle.push_back(CacheEntry{key, p, ++c.clock, plan_device_bytes(p)});
            auto total = [&] { size_t t = 0; for (const CacheEntry& e : c.idle) t += e.bytes; return t; };
            while (!c.idle.empty() && ((int)c.idle.size() > c.max_plans || (c.idle.size() > 1 && total() > c.max_bytes))) {
                size_t lru = 0;
                for (size_t i = 1; i < c.idle.size(); i++)
                    if (c.idle[i].stamp < c.idle[lru].stamp) lru = i;
                drop.push_back(c.idle[lru].plan);
                c.idle.erase(c.idle.begin() + (long)lru);
            }
        }
    }
    for (fftconv_plan* d : drop) fftconv_plan_destroy(d);
}

}  // namespace

bool fc::cache_release_idle() {
    std::vector<CacheEntry> drop;
    {
        PlanCache& c = plan_cache();
        std::lock_guard<std::mutex> lk(c.m);
        drop.swap(c.idle);
    }
    if (drop.empty()) return false;
    const std::string keep = api_last_error();
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = -1; }
    for (CacheEntry& e : drop) (void)fftconv_plan_destroy(e.plan);     // (makes the plan's device current)
    if (dev >= 0) (void)hipSetDevice(dev);
    api_set_last_error(keep);
    return true;
}

namespace {

thread_local fftconv_call_timing g_call_timing = {0, 0, 0, 0, 0, 0};
double ms_since(const std::chrono::steady_clock::time_point& t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace

extern "C" {

int fftconv_cache_configure(int max_plans, size_t max_bytes) {
    if (max_plans < 0) return api_fail(FFTCONV_ERR_INVALID_ARG, "max_plans must not be negative");
    std::vector<fftconv_plan*> drop;
    {
        PlanCache& c = plan_cache();
        std::lock_guard<std::mutex> lk(c.m);
        c.max_plans = max_plans;
        if (max_bytes) c.max_bytes = max_bytes;
        while ((int)c.idle.size() > c.max_plans) {
            size_t lru = 0;
            for (size_t i = 1; i < c.idle.size(); i++)
                if (c.idle[i].stamp < c.idle[lru].stamp) lru = i;
            drop.push_back(c.idle[lru].plan);
            c.idle.erase(c.idle.begin() + (long)lru);
        }
    }
    for (fftconv_plan* d : drop) fftconv_plan_destroy(d);
    return 0;
}

int fftconv_cache_clear(void) {
    std::vector<CacheEntry> drop;
    {
        PlanCache& c = plan_cache();
        std::lock_guard<std::mutex> lk(c.m);
        drop.swap(c.idle);
    }
    int rc = 0;
    for (CacheEntry& e : drop)
        if (int r = fftconv_plan_destroy(e.plan)) rc = r;
    return rc;
}

int fftconv_cache_stats(long* plans, long* hits, long* misses, size_t* device_bytes) {
    PlanCache& c = plan_cache();
    std::lock_guard<std::mutex> lk(c.m);
    if (plans) *plans = (long)c.idle.size();
    if (hits) *hits = c.hits;
    if (misses) *misses = c.misses;
    if (device_bytes) {
        *device_bytes = 0;
        for (const CacheEntry& e : c.idle) *device_bytes += e.bytes;
    }
    return 0;
}

int fftconv_last_call_timing(fftconv_call_timing* timing) {
    if (!timing) return api_fail(FFTCONV_ERR_INVALID_ARG, "NULL argument");
    *timing = g_call_timing;
    return 0;
}

int fftconv_convolution_fft(const float* data, int data_h, int data_w, int feature_dim, int max_kernel_h,
                            int max_kernel_w, int n_kernel, const float* const* kernels, const int* kernel_h,
                            const int* kernel_w, const int* kernel_f, const double* thread_size, int n_thread_size,
                            int gpu_id, float* const* out, int* fft_h, int* fft_w) {
    return fftconv_convolution_fft_ex(data, data_h, data_w, feature_dim, max_kernel_h, max_kernel_w, n_kernel, kernels, kernel_h,
                                      kernel_w, kernel_f, FFTCONV_HOST, thread_size, n_thread_size, gpu_id, out, fft_h, fft_w, nullptr);
}

int fftconv_convolution_fft_ex(const float* data, int data_h, int data_w, int feature_dim, int max_kernel_h,
                               int max_kernel_w, int n_kernel, const float* const* kernels, const int* kernel_h,
                               const int* kernel_w, const int* kernel_f, int kernel_location, const double* thread_size,
                               int n_thread_size, int gpu_id, float* const* out, int* fft_h, int* fft_w,
                               const fftconv_plan_options* options) {
    if (kernel_location != FFTCONV_HOST && kernel_location != FFTCONV_DEVICE && kernel_location != FFTCONV_AUTO)
        return api_fail(FFTCONV_ERR_INVALID_ARG, "bad kernel location");
    // argument checks in the reference's order (src/cudaConvolutionFFT.cu:45-89)
    if (!data || data_h < 1 || data_w < 1 || feature_dim < 1) return api_fail(FFTCONV_ERR_INVALID_ARG, "Invalid data input");
    if (n_kernel < 0 || (n_kernel > 0 && (!kernels || !kernel_h || !kernel_w || !out)))
        return api_fail(FFTCONV_ERR_INVALID_ARG, "Kernel must be a cell array");
    if (int rc = check_thread_size(thread_size, n_thread_size)) return rc;
    if (kernel_f)
        for (int k = 0; k < n_kernel; k++)
            if (kernel_f[k] != feature_dim)  // src/cudaConvolutionFFT.cu:242
                return api_fail(FFTCONV_ERR_KERNEL_SHAPE,
                            "Kernel and Data must have the same number of features and kernel size should be smaller than data size");
    if (fft_h) *fft_h = fft_size16(data_h + max_kernel_h - 1);
    if (fft_w) *fft_w = fft_size16(data_w + max_kernel_w - 1);
    // the plan: from the cache (same problem, device and options as an earlier call), else built now.
    // (sizes beyond one single-pass plan: the plan is block-wise -- overlap-add over ordinary plans -- by itself)
    const auto t0 = std::chrono::steady_clock::now();
    fftconv_call_timing tm = {0, 0, 0, 0, 0, 0};
    if (gpu_id < 0) {
        int ndev = 0;
        if (int rc = fftconv_device_count(&ndev)) return rc;
        HIP_TRY(hipGetDevice(&gpu_id));
    }
    const CacheKey key = cache_key(data_h, data_w, feature_dim, max_kernel_h, max_kernel_w, gpu_id, options);
    fftconv_plan* p = cache_take(key);
    tm.cache_hit = p ? 1 : 0;
    if (!p) {
        if (int rc = fftconv_plan_create_ex(&p, data_h, data_w, feature_dim, max_kernel_h, max_kernel_w, gpu_id, nullptr, options)) return rc;
        // a plan that will not be kept (cache switched off: fftconv_cache_configure(0, ...)) has nothing to gain from timing candidate
        // placements of its intermediate (~100 ms against ~1 % of one call): the automatic default (placement.cpp) is for plans that live on
        bool kept;
        {
            PlanCache& c = plan_cache();
            std::lock_guard<std::mutex> lk(c.m);
            kept = c.max_plans > 0;
        }
        if (!kept) (void)fftconv_plan_set_option(p, "tune_placement", 0);
    }
    (void)fftconv_plan_set_option(p, "verbose", options_verbose(options) ? 1 : 0);
    tm.plan_ms = ms_since(t0);
    const auto t1 = std::chrono::steady_clock::now();
    // The usual small call -- one group of equally sized host kernels that fits the plan's pinned buffer: the kernels are
    // staged first, so that their column pass rides in the launch of the image's column pass (k_fast_cols_fwd_pair: one
    // launch fewer on a path that is a chain of five small dependent kernels).  Anything else, and anything that fails a
    // check, takes the ordinary order below and reports its errors from there.
    const float* staged_dk = nullptr;
    if (n_kernel > 0 && kernel_location == FFTCONV_HOST && !p->tiled && p->opt_host_pinned && p->g.fast_fwd && !p->opt_flip_kernels && !p->profile) {
        bool same = true;
        for (int k = 0; k < n_kernel && same; k++)
            same = kernels[k] && out[k] && kernel_h[k] == kernel_h[0] && kernel_w[k] == kernel_w[0];
        const size_t per = (size_t)feature_dim * (size_t)std::max(kernel_h[0], 0) * (size_t)std::max(kernel_w[0], 0);
        if (same && per > 0 && per * n_kernel * sizeof(float) <= FC_PIN_INPLACE_BYTES && kernel_h[0] <= p->g.max_kh && kernel_w[0] <= p->g.max_kw &&
            use_device(p) == 0 && p->pin_k.ensure(per * n_kernel * sizeof(float)) == 0 && p->pin_k.wait() == 0) {
            for (int j = 0; j < n_kernel; j++) memcpy(p->pin_k.p + per * j * sizeof(float), kernels[j], per * sizeof(float));
            const long keep_defer = p->opt_defer_prepare;
            p->opt_defer_prepare = 1;
            const int rcp = fftconv_plan_prepare_kernels_packed(p, n_kernel, reinterpret_cast<const float*>(p->pin_k.p), kernel_h[0], kernel_w[0]);
            p->opt_defer_prepare = keep_defer;
            if (rcp == 0 && p->deferred.on) staged_dk = reinterpret_cast<const float*>(p->pin_k.p);
            else p->deferred.on = false;
        }
    }
    int rc = fftconv_plan_set_image(p, data, FFTCONV_HOST);
    tm.image_ms = ms_since(t1);
    const auto t2 = std::chrono::steady_clock::now();
    if (!rc && staged_dk) {
        Sink sink;
        sink.ptrs = out;
        sink.location = FFTCONV_HOST;
        rc = run_group(p, n_kernel, staged_dk, kernel_h[0], kernel_w[0], sink);
        const std::string keep_err = api_last_error();
        const int rcm = p->pin_k.mark(p->stream);
        if (!rc) rc = rcm; else api_set_last_error(keep_err);
        if (!rc) {
            hipError_t e = hipStreamSynchronize(p->stream);
            if (e != hipSuccess) rc = api_fail(FFTCONV_ERR_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(e));
        }
    } else if (!rc) {
        rc = fftconv_plan_convolve(p, n_kernel, kernels, kernel_h, kernel_w, kernel_location, out, FFTCONV_HOST);
    }
    if (rc) p->deferred.on = false;       // (a failed image leaves no request behind in a plan that goes back into the cache)
    tm.convolve_ms = ms_since(t2);
    const auto t3 = std::chrono::steady_clock::now();
    std::string keep = api_last_error();
    // argument-class failures were found before anything was queued and leave the plan as it was; after a HIP or
    // allocation failure the plan is not trusted again
    const bool reusable = rc == 0 || rc == FFTCONV_ERR_INVALID_ARG || rc == FFTCONV_ERR_KERNEL_SHAPE || rc == FFTCONV_ERR_KERNEL_EXCEEDS_MAX ||
                          rc == FFTCONV_ERR_THREAD_SIZE;
    if (!p->tiled) { p->prepared.dk = nullptr; p->deferred.on = false; }   // (both may name pin_k, whose contents the next call replaces)
    if (reusable) cache_put(key, p);
    else fftconv_plan_destroy(p);
    if (rc) api_set_last_error(keep);
    tm.release_ms = ms_since(t3);
    tm.total_ms = ms_since(t0);
    g_call_timing = tm;
    if (options_verbose(options))
        fprintf(stderr, "fftconv: one-shot call %.3f ms = plan %.3f (%s) + image %.3f + %d kernels %.3f + release %.3f\n", tm.total_ms, tm.plan_ms,
                tm.cache_hit ? "cached" : "created", tm.image_ms, n_kernel, tm.convolve_ms, tm.release_ms);
    return rc;
}

}  // extern "C"
