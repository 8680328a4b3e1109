// plan_internal.hpp -- what the host translation units of libfftconv.so share (not installed): the plan object behind
// include/fftconv.h, its device / pinned buffers, and the internal entry points each unit offers the others.
//   fftconv_api.cpp    plan core: creation, image transform, the per-kernel loop (run_group), options, two-step pair
//   plan_cache.cpp     plan cache of the one-shot entries + fftconv_convolution_fft[_ex] (the MEX body)
//   host_ring.cpp      host-output streaming (copy threads, pinned ring)
//   blockwise.cpp      block-wise plans (overlap-save / overlap-add) and their planner
//   placement.cpp      opt-in placement tuning of the intermediate
//   fftconv_multi.cpp  several GPUs from one process (public API + api_internal.hpp only)
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/fftconv.h"
#include "api_internal.hpp"
#include "kernels.hpp"
#include "pipeline.hpp"

constexpr long FC_HOST_MIN_KB = 1024;   // default of plan option "host_min_kb" (0 reproduces the threaded path for small maps: tests)

#define HIP_TRY(call)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fc::api_fail(FFTCONV_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),    \
                                __FILE__, __LINE__);                                               \
    } while (0)

namespace fc {

// smallest fftconv_plan_options this library accepts: the struct as it was before `blockwise` was appended
constexpr size_t kOptionsMinSize = offsetof(fftconv_plan_options, exact_window) + sizeof(int);
inline bool options_no_blockwise(const fftconv_plan_options* o) {
    return o && o->struct_size >= offsetof(fftconv_plan_options, blockwise) + sizeof(int) && o->blockwise == 1;
}
inline bool options_verbose(const fftconv_plan_options* o) {
    return o && o->struct_size >= offsetof(fftconv_plan_options, verbose) + sizeof(int) && o->verbose != 0;
}
inline PlanTuning tuning_from(const fftconv_plan_options* o) {
    PlanTuning t;
    if (!o || o->struct_size < kOptionsMinSize) return t;
    t.path_mode = o->kernel_path == 1 ? 0 : o->kernel_path == 2 ? 1 : 2;
    t.rows_group = o->rows_group <= 0 ? -1 : o->rows_group;
    t.max_transform = o->max_transform > 0 ? o->max_transform : 0;
    t.exact_window = o->exact_window != 0;
    return t;
}

// plan_cache.cpp: destroys every idle plan of the one-shot entries' cache (device scratch, pinned staging, copy threads);
// true if there was one.  Called by DevBuf::ensure when the device is out of memory.
bool cache_release_idle();

enum { PK_KERNEL_COLS = 0, PK_SPECTRAL = 1, PK_OUT_COLS = 2, PK_IMAGE_COLS = 3, PK_IMAGE_ROWS = 4, PK_COUNT = 5 };

struct EventPair {
    hipEvent_t start, stop;
    int kind;
    long units;
};

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;  // elements
    bool fresh = false;   // (re)allocated since the flag was last cleared
    int ensure(size_t n) {
        if (n <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        fresh = true;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), n * sizeof(T));
        if (e != hipSuccess && cache_release_idle()) {
            // out of device memory while the one-shot entries' plan cache holds idle plans (up to 48 GiB of scratch): they go
            // first -- the reference's contract is "all device memory is released" between calls -- and the request is repeated
            (void)hipGetLastError();
            e = hipMalloc(reinterpret_cast<void**>(&p), n * sizeof(T));
        }
        if (e != hipSuccess) {
            p = nullptr;
            (void)hipGetLastError();
            return api_fail(FFTCONV_ERR_ALLOC, "hipMalloc of %zu bytes failed: %s", n * sizeof(T), hipGetErrorString(e));
        }
        cap = n;
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    size_t bytes() const { return cap * sizeof(T); }
};

// Pinned host staging of the small-call path (host arrays in / out of a few hundred KB: the sizes the reference's demo
// calls with).  A copy between pageable memory and the device is a blocking runtime call of 10-25 us whatever its size,
// and the reference's entry makes one per kernel and one per map (src/cudaConvolutionFFT.cu:148,231,286).  Here the
// CPU copies the caller's small arrays into / out of pinned buffers of the plan: an image or a kernel set of at most
// FC_PIN_INPLACE_BYTES is then read by the column kernels IN PLACE over PCIe (no copy command at all), a larger one
// (up to FC_PIN_IMAGE_BYTES) crosses in one asynchronous copy, and the maps of a launch come back in ONE copy.
// `busy` is recorded behind the last GPU work that reads the buffer; the next fill waits for it.
constexpr size_t FC_PIN_INPLACE_BYTES = (size_t)512 << 10;
constexpr size_t FC_PIN_IMAGE_BYTES = (size_t)1 << 20;
constexpr size_t FC_PIN_OUT_BYTES = (size_t)8 << 20;
constexpr size_t FC_PIN_ONE_MAP_BYTES = (size_t)64 << 10;
struct PinBuf {
    char* p = nullptr;
    size_t cap = 0;
    hipEvent_t busy = nullptr;
    bool in_use = false;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        if (int rc = wait()) return rc;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        const size_t want = (bytes + 65535) & ~(size_t)65535;
        hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&p), want, hipHostMallocDefault);
        if (e != hipSuccess) { p = nullptr; return api_fail(FFTCONV_ERR_ALLOC, "hipHostMalloc of %zu bytes failed: %s", want, hipGetErrorString(e)); }
        cap = want;
        return 0;
    }
    int wait() {                      // until the GPU work that reads the buffer is over
        if (!in_use) return 0;
        in_use = false;
        hipError_t e = hipEventSynchronize(busy);
        if (e != hipSuccess) return api_fail(FFTCONV_ERR_HIP, "hipEventSynchronize failed: %s", hipGetErrorString(e));
        return 0;
    }
    int mark(hipStream_t s) {         // everything queued on s so far may read the buffer
        if (!busy) {
            hipError_t e = hipEventCreateWithFlags(&busy, hipEventDisableTiming);
            if (e != hipSuccess) { busy = nullptr; return api_fail(FFTCONV_ERR_HIP, "hipEventCreate failed: %s", hipGetErrorString(e)); }
        }
        hipError_t e = hipEventRecord(busy, s);
        if (e != hipSuccess) return api_fail(FFTCONV_ERR_HIP, "hipEventRecord failed: %s", hipGetErrorString(e));
        in_use = true;
        return 0;
    }
    void release() {
        if (in_use && busy) (void)hipEventSynchronize(busy);
        in_use = false;
        if (busy) (void)hipEventDestroy(busy);
        busy = nullptr;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
    }
};

}  // namespace fc

#include "host_ring.hpp"

struct TiledState;

// Where the block plan of an overlap-save block-wise plan stores its maps (set around each run by tiled_convolve): rows
// [h_lo, h_hi) of columns [w_first, w_first + ncols) of the block's circular result, row h of column w of map j at
// base + j * map_stride + w * pitch + h -- the block's rectangle of the full maps (base is offset accordingly).
struct OutWindow {
    float* base;
    size_t map_stride;
    int pitch, h_lo, h_hi, w_first, ncols;
};

using namespace fc;   // (host units only: this header is not installed)

struct fftconv_plan {
    TiledState* tiled = nullptr;   // block-wise plan: sizes beyond one LDS-resident pass, or large sizes that run faster in blocks (see TiledState)
    const OutWindow* win = nullptr;   // block plan of an overlap-save plan: the output kernel writes this window, whatever the sink says
    Geometry g;
    Tables t;
    DeviceTables d;
    int gpu_id = 0;
    hipStream_t stream = nullptr;
    bool have_image = false;
    DevBuf<c32> tw_m, tw_w;
    DevBuf<PairEntry> pairs;
    DevBuf<c32> S;     // image spectrum (own buffer)
    c32* Sx = nullptr; // caller-owned spectrum buffer, if any
    c32* spec() const { return Sx ? Sx : S.p; }   // S is allocated by the first use that needs it (ensure_spectrum)
    int ensure_spectrum() { return Sx ? 0 : S.ensure(g.spectrum_elems()); }
    DevBuf<c32> A;     // kernel column spectra of the current chunk
    DevBuf<c32> Y;     // intermediate of the current map batch
    DevBuf<float> K;   // packed kernels staged on the device
    DevBuf<float> KF;  // flipped copy of the current chunk of kernels ("flip_kernels")
    long opt_flip_kernels = 0;
    // "output_region": which part of the padded window a map holds (MAX_KERNEL sizes K):
    // 0 window FFT_H x FFT_W (the reference), 1 full (DATA + K - 1), 2 same (DATA, centred), 3 valid (DATA - K + 1)
    long opt_region = 0;
    int out_h = 0, out_w = 0, off_h = 0, off_w = 0;
    DevBuf<float> OC;  // cropped maps staged for the copy-out
    size_t out_elems() const { return opt_region ? (size_t)out_h * out_w : g.map_elems(); }   // (block-wise plans: g holds the whole window)
    DevBuf<float> O;   // output staging (pointer-array / host output)
    DevBuf<float> I;   // image staging (host input)
    PinBuf pin_img, pin_k, pin_out;   // pinned host staging of small host arrays (PinBuf above)
    hipEvent_t pin_out_done[2] = {nullptr, nullptr};   // copy into each half of pin_out complete
    long opt_host_pinned = 1;         // 0: small host arrays take the plain copies (A/B, tests)
    DevBuf<c32> fr_tw1, fr_tw2;
    DevBuf<int> fr_map;
    DevBuf<c32> fc_tw1, fc_tw2;
    DevBuf<PairEntry> fc_pairs;
    DevBuf<int> fc_rowoff, fc_pair_row_of;
    DevBuf<int> queue;                    // counters of the dynamic tile queue (option "dynamic_tiles"; allocated when it is first set)
    long opt_dynamic_tiles = 0;           // 1: the persistent output kernel takes its tiles from a queue (fast_cols.hpp: TileQueue); 2: the forward column kernels too
    DevBuf<int> nat_row_of, nat_col_of;   // natural-order spectrum exchange (uploaded on first use)
    DevBuf<c32> NS;                       // its device staging for host callers
    int num_cus = 256;
    long opt_batch_maps = 0;
    long opt_kernel_chunk_mb = 0;
    int tuned_candidates = 0, tuned_best = 0;   // of the last placement tuning (fftconv_plan_get_option)
    long opt_tune_placement = -1;  // > 1: that many candidate allocations of the intermediate are tried (tune_intermediate_placement); 0 / 1: never;
                                   // -1 (default): automatic -- placement_auto_candidates (placement.cpp)
    long opt_host_stream = 1;      // copy-out of host maps: 0 blocking, 1 direct by host threads, 2 pinned ring
    long opt_host_min_kb = FC_HOST_MIN_KB;   // maps smaller than this leave by blocking copies whatever host_stream says
    long opt_host_threads = 0;     // host copy threads of the output ring (0 = auto)
    long opt_host_chunk_kb = 0;    // ring chunk size (0 = auto)
    long opt_host_slots = 0;       // ring chunks (0 = auto)
    long opt_defer_prepare = 0;    // 1: fftconv_plan_prepare_kernels_packed only records its request (see `deferred`)
    long opt_verbose = 0;          // 1: per-stage sizes and launch shapes to stderr (the reference's `debug`, src/cudaConvolutionFFT.cu:9)
    HostRing* ring = nullptr;      // created on the first host-output convolve
    bool profile = false;
    unsigned profile_mask = ~0u;   // which kinds (bit = PK_* index) are timed while `profile` is on
    bool prof_open = false;        // the last prof_begin recorded a start event
    // kernel column spectra of the first chunk already in A (fftconv_plan_prepare_kernels_packed)
    struct { const float* dk = nullptr; int n = 0, kh = 0, kw = 0; hipStream_t stream = nullptr; } prepared;   // (stream: the one A was produced on)
    // fftconv_plan_prepare_kernels_packed DEFERRED: the kernels' column pass is launched by whichever comes first, the
    // next set_image on the same stream (then in ONE launch with the image's column pass: launch_fast_cols_fwd_pair) or
    // the next convolve / any call that must see it done (flush_pending_prepare)
    struct { bool on = false; const float* dk = nullptr; int n = 0, na = 0, kh = 0, kw = 0; hipStream_t stream = nullptr; } deferred;
    std::vector<EventPair> pending;
    std::vector<EventPair> pool;
    double prof_ms[PK_COUNT] = {0, 0, 0, 0, 0};
    long prof_launches[PK_COUNT] = {0, 0, 0, 0, 0};
    long prof_units[PK_COUNT] = {0, 0, 0, 0, 0};

    size_t cols_lds() const { return (size_t)g.T_cols * g.lds_pitch * sizeof(c32); }
    size_t rows_lds() const { return (size_t)g.Lw * sizeof(c32) * (g.F > 1 ? 2 : 1); }

    int prof_begin(int kind, long units) {
        prof_open = profile && ((profile_mask >> kind) & 1u);
        if (!prof_open) return 0;
        EventPair ep;
        if (!pool.empty()) {
            ep = pool.back();
            pool.pop_back();
        } else {
            HIP_TRY(hipEventCreate(&ep.start));
            HIP_TRY(hipEventCreate(&ep.stop));
        }
        ep.kind = kind;
        ep.units = units;
        HIP_TRY(hipEventRecord(ep.start, stream));
        pending.push_back(ep);
        return 0;
    }
    int prof_end() {
        if (!prof_open) return 0;
        prof_open = false;
        HIP_TRY(hipEventRecord(pending.back().stop, stream));
        return 0;
    }
    int prof_collect() {
        for (EventPair& ep : pending) {
            HIP_TRY(hipEventSynchronize(ep.stop));
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, ep.start, ep.stop));
            prof_ms[ep.kind] += ms;
            prof_launches[ep.kind] += 1;
            prof_units[ep.kind] += ep.units;
            pool.push_back(ep);
        }
        pending.clear();
        return 0;
    }
    void release_ring() {
        if (ring) {
            ring->shutdown();
            delete ring;
            ring = nullptr;
        }
    }
    void release_all() {
        release_ring();
        for (EventPair& ep : pending) { (void)hipEventDestroy(ep.start); (void)hipEventDestroy(ep.stop); }
        for (EventPair& ep : pool) { (void)hipEventDestroy(ep.start); (void)hipEventDestroy(ep.stop); }
        pending.clear();
        pool.clear();
        tw_m.release(); tw_w.release(); pairs.release();
        S.release(); A.release(); Y.release(); K.release(); KF.release(); O.release(); OC.release(); I.release();
        fr_tw1.release(); fr_tw2.release(); fr_map.release();
        fc_tw1.release(); fc_tw2.release(); fc_pairs.release(); fc_rowoff.release(); fc_pair_row_of.release();
        nat_row_of.release(); nat_col_of.release(); NS.release(); queue.release();
        pin_img.release(); pin_k.release(); pin_out.release();
        for (int h = 0; h < 2; h++) { if (pin_out_done[h]) (void)hipEventDestroy(pin_out_done[h]); pin_out_done[h] = nullptr; }
    }
};

namespace fc {

// the reference's debug prints (src/cudaConvolutionFFT.cu:60,68,100,114,240,258), behind plan option "verbose"
#define FC_VERBOSE(p, ...) do { if ((p)->opt_verbose) { fprintf(stderr, "fftconv: " __VA_ARGS__); fputc('\n', stderr); } } while (0)

// where the maps of a group go
struct Sink {
    float* packed = nullptr;        // device base, maps consecutive
    float* const* ptrs = nullptr;   // or one pointer per map
    int location = FFTCONV_DEVICE;  // of ptrs
};

struct BatchSizes {
    size_t per_a;  // c32 of column spectrum per kernel
    int nbY;       // maps per spectral/output launch
    int nbA;       // kernels per column-spectrum chunk (a multiple of nbY)
};

// ---- plan core (fftconv_api.cpp) ----
int use_device(const fftconv_plan* p);
BatchSizes batch_sizes(const fftconv_plan* p, int n, int kw);
int check_kernel_size(const fftconv_plan* p, int kh, int kw);
int check_thread_size(const double* thread_size, int n_thread_size);
// Core of the per-kernel loop for n kernels of one size, packed on the device at dk ([n][F][kw][kh]); on failure nothing of the
// host-output ring is still writing into the caller's buffers when the error is returned
int run_group(fftconv_plan* p, int n, const float* dk, int kh, int kw, const Sink& sink);
// cyclic: the block plan of an overlap-save block-wise plan (PlanTuning::cyclic) -- never block-wise itself
int plan_create_internal(fftconv_plan** plan, int data_h, int data_w, int feature_dim, int max_kernel_h, int max_kernel_w, int gpu_id,
                         void* hip_stream, const fftconv_plan_options* options, bool cyclic);

// ---- host-output streaming (host_ring.cpp) ----
// pinned ring + copy stream + host copy threads of the host-output path, sized for this plan's maps
int ring_ensure(fftconv_plan* p);
// queue the copy-out of the maps [first, first + count) that sit in staging buffer `buf`
int ring_drain(fftconv_plan* p, const Sink& sink, int first, int count, int buf, const float* staging);

// ---- placement tuning (placement.cpp) ----
int tune_intermediate_placement(fftconv_plan* p, int k, int n, int nbY, float* out, size_t out_stride_per_map);
int placement_auto_candidates(const fftconv_plan* p, size_t launch_map_bytes);

// ---- block-wise plans (blockwise.cpp) ----
bool blocks_preferred(const Geometry& g, const fftconv_plan_options* options);
int tiled_create(fftconv_plan* p, int H, int W, int F, int mkh, int mkw, void* hip_stream, const fftconv_plan_options* options);
int tiled_set_image(fftconv_plan* p, const float* data, int location);
int tiled_convolve(fftconv_plan* p, int n, const float* const* kernels, const int* kh, const int* kw, int kernel_location,
                   float* const* out, int out_location, float* out_packed);
int tiled_unsupported(const char* what);

}  // namespace fc

struct TiledState {
    fftconv_plan* sub = nullptr;     // the block plan (an ordinary plan on the same stream)
    int H = 0, W = 0, F = 0, mkh = 0, mkw = 0;
    int Bh = 0, Bw = 0, nbh = 0, nbw = 0, nblk = 0, FH = 0, FW = 0;
    // overlap-save (see the comment above): the block plan is cyclic over Lh x Lw samples, block (by, bx) reads the image rows
    // [by * Bh - Sh, by * Bh + Bh) and stores the rows [by * Bh, by * Bh + Bh) of the maps straight from the output kernel
    bool save = false;
    int Lh = 0, Lw = 0, Sh = 0, Sw = 0;
    size_t spec_elems = 0;           // c32 per block spectrum
    DevBuf<c32> specs;               // block spectra, [block][spec_elems] (own buffer)
    c32* specs_x = nullptr;          // caller-owned instead (fftconv_plan_use_spectrum_buffer)
    DevBuf<float> big, tmp, blk;     // full maps of a kernel chunk, block maps of that chunk, one zero-padded image block
    DevBuf<float> crop;              // "output_region" maps of a kernel chunk, cropped out of `big` for the copy-out
    DevBuf<float> kstage;            // host kernels of a chunk, staged on the device once (every block convolves them)
    std::vector<float> hblk;         // host staging of one image block
    bool have_image = false;
    c32* spec_base() const { return specs_x ? specs_x : specs.p; }
    size_t spec_total() const { return spec_elems * (size_t)nblk; }
    size_t big_map() const { return (size_t)FH * FW; }
    void release() {
        if (sub) fftconv_plan_destroy(sub);
        sub = nullptr;
        specs.release(); big.release(); tmp.release(); blk.release(); kstage.release(); crop.release();
    }
};
