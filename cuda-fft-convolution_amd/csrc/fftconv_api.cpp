// fftconv_api.cpp -- host side of libfftconv.so: the C ABI of include/fftconv.h on top of the
// HIP kernels.  C++ host code in the role of the reference's MEX gateways
// (src/cudaConvolutionFFT.cu, src/cudaFFTData.cu, src/cudaConvFFTData.cu); no CPU compute path.
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <new>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fftconv.h"
#include "api_internal.hpp"
#include "kernels.hpp"
#include "pipeline.hpp"

using namespace fc;

constexpr long FC_HOST_MIN_KB = 1024;   // default of plan option "host_min_kb" (0 reproduces the threaded path for small maps: tests)

namespace {

thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

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

// smallest fftconv_plan_options this library accepts: the struct as it was before `blockwise` was appended
constexpr size_t kOptionsMinSize = offsetof(fftconv_plan_options, exact_window) + sizeof(int);
bool options_no_blockwise(const fftconv_plan_options* o) {
    return o && o->struct_size >= offsetof(fftconv_plan_options, blockwise) + sizeof(int) && o->blockwise == 1;
}

bool options_verbose(const fftconv_plan_options* o) {
    return o && o->struct_size >= offsetof(fftconv_plan_options, verbose) + sizeof(int) && o->verbose != 0;
}

PlanTuning tuning_from(const fftconv_plan_options* o) {
    PlanTuning t;
    if (!o || o->struct_size < kOptionsMinSize) return t;
    t.path_mode = o->kernel_path == 1 ? 0 : o->kernel_path == 2 ? 1 : 2;
    t.rows_group = o->rows_group <= 0 ? -1 : o->rows_group;
    t.max_transform = o->max_transform > 0 ? o->max_transform : 0;
    t.exact_window = o->exact_window != 0;
    return t;
}

#define HIP_TRY(call)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(FFTCONV_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),    \
                        __FILE__, __LINE__);                                                       \
    } while (0)

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
        if (e != hipSuccess) {
            p = nullptr;
            return fail(FFTCONV_ERR_ALLOC, "hipMalloc of %zu bytes failed: %s", n * sizeof(T), hipGetErrorString(e));
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
        if (e != hipSuccess) { p = nullptr; return fail(FFTCONV_ERR_ALLOC, "hipHostMalloc of %zu bytes failed: %s", want, hipGetErrorString(e)); }
        cap = want;
        return 0;
    }
    int wait() {                      // until the GPU work that reads the buffer is over
        if (!in_use) return 0;
        in_use = false;
        hipError_t e = hipEventSynchronize(busy);
        if (e != hipSuccess) return fail(FFTCONV_ERR_HIP, "hipEventSynchronize failed: %s", hipGetErrorString(e));
        return 0;
    }
    int mark(hipStream_t s) {         // everything queued on s so far may read the buffer
        if (!busy) {
            hipError_t e = hipEventCreateWithFlags(&busy, hipEventDisableTiming);
            if (e != hipSuccess) { busy = nullptr; return fail(FFTCONV_ERR_HIP, "hipEventCreate failed: %s", hipGetErrorString(e)); }
        }
        hipError_t e = hipEventRecord(busy, s);
        if (e != hipSuccess) return fail(FFTCONV_ERR_HIP, "hipEventRecord failed: %s", hipGetErrorString(e));
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

// Host-output streaming (the reference's blocking pageable cudaMemcpy of every
// map, src/cudaConvolutionFFT.cu:284-286, and the async intent of
// src/cudaConvFFTDataStreams.cu:368-369,429-430): finished maps leave the device while the next
// batch is computed.  Two ways out, both driven by a few host threads of the plan:
//   direct (default)  each thread copies whole maps from the device staging buffer straight into
//                     the caller's memory on a stream of its own (the HIP runtime pins pageable
//                     pages in place: measured 51 GB/s on MI355X, the PCIe rate);
//   ring              the maps travel through a ring of pinned chunks on one copy stream and the
//                     threads move the landed chunks into the caller's buffers (buffers the
//                     caller pinned itself are written directly by the DMA engine).
struct HostRing {
    int gpu_id = 0;
    size_t chunk_bytes = 0;
    int nslots = 0;
    char* base = nullptr;              // hipHostMalloc: nslots * chunk_bytes
    hipStream_t copy_stream = nullptr;
    std::vector<hipEvent_t> landed;    // per slot: its D2H copy has finished
    hipEvent_t compute_done[2] = {nullptr, nullptr};  // per device staging buffer
    hipEvent_t copy_done[2] = {nullptr, nullptr};
    struct Task { int slot; char* dst; size_t bytes; const char* src; int buf; bool pinned; };  // slot < 0: direct copy from src (pinned: the caller pinned dst itself)
    std::deque<Task> q;
    std::mutex m;
    std::condition_variable cv_task, cv_slot;
    std::vector<char> busy;            // slot claimed (from acquire until its host copy is done)
    int next_slot = 0;
    int open_tasks = 0;
    int open_direct[2] = {0, 0};       // direct copies still reading device staging buffer 0 / 1
    bool stop = false;
    hipError_t worker_error = hipSuccess;
    std::vector<std::thread> workers;

    // Fresh caller buffers (malloc'ed, never touched) would be faulted in page by page inside the
    // runtime's pinning of the destination; populating them here, in the copy threads and ahead
    // of the DMA, costs nothing for resident pages and is several times faster for new ones.
    static void prefault(char* dst, size_t bytes) {
#ifdef MADV_POPULATE_WRITE
        const uintptr_t a = (reinterpret_cast<uintptr_t>(dst) + 4095) & ~(uintptr_t)4095;
        const uintptr_t b = (reinterpret_cast<uintptr_t>(dst) + bytes) & ~(uintptr_t)4095;
        if (b > a) {
#ifdef MADV_HUGEPAGE
            // fresh destinations of many megabytes (what mxCreateNumericArray hands out): let the kernel back them with 2-MB
            // pages where it may -- one fault and one clear per 2 MB instead of 512 (cfg3, 64 fresh maps: the page clearing was
            // 4/5 of the call); a no-op for resident pages and where transparent huge pages are off
            if (b - a >= ((size_t)4 << 20)) (void)madvise(reinterpret_cast<void*>(a), b - a, MADV_HUGEPAGE);
#endif
            (void)madvise(reinterpret_cast<void*>(a), b - a, MADV_POPULATE_WRITE);
        }
#else
        (void)dst; (void)bytes;
#endif
    }
    // Streams and bounce buffers of the copy threads are created HERE, by the thread that owns the plan, before the
    // threads start, and destroyed by it after they were joined: several fresh threads calling
    // hipStreamCreateWithFlags at the same time corrupted the runtime's heap about once in 50 starts (glibc abort
    // in free() inside libhsa-runtime64 under hipStreamCreateWithFlags -- native backtrace in
    // profiles/r02y_host_thread_stream_create_abort.txt; that was the small-map incident of DESIGN.md 6).
    std::vector<hipStream_t> own_streams;
    char* bounce_base = nullptr;       // two pinned pages per copy thread (unaligned ends of a destination)
    hipError_t prepare_workers(int nthreads) {
        if (hipHostMalloc(reinterpret_cast<void**>(&bounce_base), (size_t)8192 * nthreads, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            bounce_base = nullptr;
        }
        for (int i = 0; i < nthreads; i++) {
            hipStream_t st = nullptr;
            hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
            if (e != hipSuccess) return e;
            own_streams.push_back(st);
        }
        return hipSuccess;
    }
    void work(int i) {
        (void)hipSetDevice(gpu_id);
        work_loop(own_streams[i], bounce_base ? bounce_base + (size_t)8192 * i : nullptr);
    }
    // Device -> pageable caller memory.  The runtime pins the destination pages in place; destinations of
    // different maps may be heap neighbours that share their first / last page, and those pages would be pinned by
    // two threads at once.  So only the whole pages INSIDE the destination take the direct road; the unaligned
    // head and tail (< 4 KB each) land in this thread's pinned bounce buffer and are copied by the CPU.
    static hipError_t copy_out(char* dst, const char* src, size_t n, hipStream_t own, char* bounce) {
        size_t head = (size_t)((4096 - (reinterpret_cast<uintptr_t>(dst) & 4095)) & 4095);
        if (!bounce) head = 0;
        if (head > n) head = n;
        const size_t tail = bounce ? ((n - head) & 4095) : 0;
        const size_t mid = n - head - tail;
        hipError_t e = hipSuccess;
        if (mid) {
            prefault(dst + head, mid);
            e = hipMemcpyAsync(dst + head, src + head, mid, hipMemcpyDeviceToHost, own);
        }
        if (e == hipSuccess && head) e = hipMemcpyAsync(bounce, src, head, hipMemcpyDeviceToHost, own);
        if (e == hipSuccess && tail) e = hipMemcpyAsync(bounce + 4096, src + head + mid, tail, hipMemcpyDeviceToHost, own);
        if (e == hipSuccess) e = hipStreamSynchronize(own);
        if (e == hipSuccess && head) memcpy(dst, bounce, head);
        if (e == hipSuccess && tail) memcpy(dst + head + mid, bounce + 4096, tail);
        return e;
    }
    void work_loop(hipStream_t own, char* bounce) {
        for (;;) {
            Task t;
            {
                std::unique_lock<std::mutex> lk(m);
                cv_task.wait(lk, [&] { return stop || !q.empty(); });
                if (q.empty()) return;
                t = q.front();
                q.pop_front();
            }
            hipError_t e;
            if (t.slot < 0) {
                e = hipStreamWaitEvent(own, compute_done[t.buf], 0);
                if (e == hipSuccess) e = copy_out(t.dst, t.src, t.bytes, own, t.pinned ? nullptr : bounce);
            } else {
                e = hipEventSynchronize(landed[t.slot]);
                if (e == hipSuccess) memcpy(t.dst, base + (size_t)t.slot * chunk_bytes, t.bytes);
            }
            {
                std::lock_guard<std::mutex> lk(m);
                if (e != hipSuccess && worker_error == hipSuccess) worker_error = e;
                if (t.slot < 0) open_direct[t.buf]--; else busy[t.slot] = 0;
                open_tasks--;
            }
            cv_slot.notify_all();
        }
    }
    int acquire() {  // next slot in ring order, once its previous contents have been copied out
        std::unique_lock<std::mutex> lk(m);
        const int s = next_slot;
        cv_slot.wait(lk, [&] { return !busy[s]; });
        busy[s] = 1;
        next_slot = (s + 1) % nslots;
        return s;
    }
    void unclaim(int s) {
        { std::lock_guard<std::mutex> lk(m); busy[s] = 0; }
        cv_slot.notify_all();
    }
    void submit(int slot, char* dst, size_t bytes) {
        { std::lock_guard<std::mutex> lk(m); q.push_back(Task{slot, dst, bytes, nullptr, 0, false}); open_tasks++; }
        cv_task.notify_one();
    }
    void submit_direct(const char* src, char* dst, size_t bytes, int buf, bool pinned) {
        { std::lock_guard<std::mutex> lk(m); q.push_back(Task{-1, dst, bytes, src, buf, pinned}); open_tasks++; open_direct[buf]++; }
        cv_task.notify_one();
    }
    // the direct copies out of staging buffer `buf` have finished: it may be overwritten
    void wait_staging_free(int buf) {
        std::unique_lock<std::mutex> lk(m);
        cv_slot.wait(lk, [&] { return open_direct[buf] == 0; });
    }
    // every queued chunk has reached the caller's memory (also drains the copy stream)
    hipError_t wait_idle() {
        hipError_t e = hipStreamSynchronize(copy_stream);
        std::unique_lock<std::mutex> lk(m);
        cv_slot.wait(lk, [&] { return open_tasks == 0; });
        if (e == hipSuccess) e = worker_error;
        worker_error = hipSuccess;
        return e;
    }
    void shutdown() {
        if (!workers.empty()) {
            { std::lock_guard<std::mutex> lk(m); stop = true; }
            cv_task.notify_all();
            for (std::thread& t : workers) t.join();
            workers.clear();
        }
        for (hipStream_t st : own_streams) (void)hipStreamDestroy(st);
        own_streams.clear();
        if (bounce_base) (void)hipHostFree(bounce_base);
        bounce_base = nullptr;
        for (hipEvent_t e : landed) (void)hipEventDestroy(e);
        landed.clear();
        for (int i = 0; i < 2; i++) {
            if (compute_done[i]) (void)hipEventDestroy(compute_done[i]);
            if (copy_done[i]) (void)hipEventDestroy(copy_done[i]);
            compute_done[i] = copy_done[i] = nullptr;
        }
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
        copy_stream = nullptr;
        if (base) (void)hipHostFree(base);
        base = nullptr;
    }
};

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

struct TiledState;

// Where the block plan of an overlap-save block-wise plan stores its maps (set around each run by tiled_convolve): rows
// [h_lo, h_hi) of columns [w_first, w_first + ncols) of the block's circular result, row h of column w of map j at
// base + j * map_stride + w * pitch + h -- the block's rectangle of the full maps (base is offset accordingly).
struct OutWindow {
    float* base;
    size_t map_stride;
    int pitch, h_lo, h_hi, w_first, ncols;
};

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
    size_t out_elems() const { return opt_region ? (size_t)out_h * out_w : g.map_elems(); }
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
    long opt_dynamic_tiles = 0;           // 1: the persistent column kernels take their tiles from a queue (fast_cols.hpp: TileQueue)
    DevBuf<int> nat_row_of, nat_col_of;   // natural-order spectrum exchange (uploaded on first use)
    DevBuf<c32> NS;                       // its device staging for host callers
    int num_cus = 256;
    long opt_batch_maps = 0;
    long opt_kernel_chunk_mb = 0;
    int tuned_candidates = 0, tuned_best = 0;   // of the last placement tuning (fftconv_plan_get_option)
    long opt_tune_placement = 0;   // > 1: that many candidate allocations of the intermediate are tried (tune_intermediate_placement)
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

namespace {

// the reference's debug prints (src/cudaConvolutionFFT.cu:60,68,100,114,240,258), behind plan option "verbose"
#define FC_VERBOSE(p, ...) do { if ((p)->opt_verbose) { fprintf(stderr, "fftconv: " __VA_ARGS__); fputc('\n', stderr); } } while (0)

int use_device(const fftconv_plan* p) {
    HIP_TRY(hipSetDevice(p->gpu_id));
    return 0;
}

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
        return fail(FFTCONV_ERR_KERNEL_SHAPE,
                    "Kernel and Data must have the same number of features and kernel size should be smaller than data size");
    if ((kh > g.max_kh || kw > g.max_kw) && !(g.exact_window && kh <= g.Lh && kw <= g.Lw))
        return fail(FFTCONV_ERR_KERNEL_EXCEEDS_MAX,
                    "kernel %dx%d exceeds MAX_KERNEL %dx%d and the internal transform (%dx%d) is not the %dx%d window",
                    kh, kw, g.max_kh, g.max_kw, g.Lh, g.Lw, g.fft_h, g.fft_w);
    if (g.fast_rows.ok && kw > g.fast_rows.max_kw)
        return fail(FFTCONV_ERR_KERNEL_EXCEEDS_MAX, "kernel width %d exceeds what this plan's row kernel accepts (%d)", kw,
                    g.fast_rows.max_kw);
    return 0;
}

// pinned ring + copy stream + host copy threads of the host-output path, sized for this plan's maps
int ring_ensure(fftconv_plan* p) {
    if (p->ring) return 0;
    const size_t map_bytes = p->out_elems() * sizeof(float);
    size_t chunk = p->opt_host_chunk_kb > 0 ? (size_t)p->opt_host_chunk_kb << 10 : (size_t)8 << 20;
    chunk = std::min(chunk, (map_bytes + 4095) / 4096 * 4096);
    chunk = std::max<size_t>(4096, chunk / 4096 * 4096);
    const bool use_ring = p->opt_host_stream == 2;
    // (direct copies: the threads also pre-fault fresh destination pages, which is CPU work -- up to 8 of them for big maps)
    const unsigned direct_threads = map_bytes >= ((size_t)8 << 20) ? 8u : 4u;
    int nthreads = p->opt_host_threads > 0 ? (int)p->opt_host_threads
                   : (int)std::max(1u, std::min(use_ring ? 6u : direct_threads, std::thread::hardware_concurrency() / 2));
    int nslots = !use_ring ? 0 : p->opt_host_slots > 0 ? (int)p->opt_host_slots : std::max(8, 2 * nthreads + 2);
    HostRing* r = new (std::nothrow) HostRing();
    if (!r) return fail(FFTCONV_ERR_ALLOC, "out of host memory");
    r->gpu_id = p->gpu_id;
    r->chunk_bytes = chunk;
    r->nslots = nslots;
    r->busy.assign(nslots, 0);
    // one plan at a time in the whole process sets its ring up (streams, events, pinned memory): the per-device
    // threads of fftconv_multi_convolve each do this on their plan's first host-output call
    static std::mutex setup_mutex;
    std::lock_guard<std::mutex> setup_lock(setup_mutex);
    hipError_t e = hipSuccess;
    if (nslots > 0) {
        e = hipHostMalloc(reinterpret_cast<void**>(&r->base), chunk * nslots, hipHostMallocDefault);
        if (e != hipSuccess) r->base = nullptr;
    }
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&r->copy_stream, hipStreamNonBlocking);
    for (int i = 0; i < 2 && e == hipSuccess; i++) {
        e = hipEventCreateWithFlags(&r->compute_done[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&r->copy_done[i], hipEventDisableTiming);
    }
    for (int i = 0; i < nslots && e == hipSuccess; i++) {
        hipEvent_t ev = nullptr;
        e = hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventBlockingSync);
        if (e == hipSuccess) r->landed.push_back(ev);
    }
    if (e == hipSuccess) e = r->prepare_workers(nthreads);
    if (e != hipSuccess) {
        r->shutdown();
        delete r;
        return fail(FFTCONV_ERR_HIP, "host-output ring setup failed: %s", hipGetErrorString(e));
    }
    for (int i = 0; i < nthreads; i++) r->workers.emplace_back([r, i] { r->work(i); });
    p->ring = r;
    return 0;
}

bool caller_pinned(const void* ptr) {
    hipPointerAttribute_t at;
    hipError_t e = hipPointerGetAttributes(&at, ptr);
    if (e != hipSuccess) {
        (void)hipGetLastError();   // pageable memory is reported as an error: not one
        return false;
    }
    return at.type == hipMemoryTypeHost;
}

// queue the copy-out of the maps [first, first + count) that sit in staging buffer `buf`
int ring_drain(fftconv_plan* p, const Sink& sink, int first, int count, int buf, const float* staging) {
    HostRing* r = p->ring;
    const size_t map_bytes = p->out_elems() * sizeof(float);
    if (r->nslots == 0) {   // direct: whole maps, one per host thread at a time
        for (int j = 0; j < count; j++)
            r->submit_direct(reinterpret_cast<const char*>(staging + (size_t)j * p->out_elems()),
                             reinterpret_cast<char*>(sink.ptrs[first + j]), map_bytes, buf, caller_pinned(sink.ptrs[first + j]));
        return 0;
    }
    HIP_TRY(hipStreamWaitEvent(r->copy_stream, r->compute_done[buf], 0));
    for (int j = 0; j < count; j++) {
        char* dst = reinterpret_cast<char*>(sink.ptrs[first + j]);
        const char* src = reinterpret_cast<const char*>(staging + (size_t)j * p->out_elems());
        if (caller_pinned(dst)) {
            HIP_TRY(hipMemcpyAsync(dst, src, map_bytes, hipMemcpyDeviceToHost, r->copy_stream));
            continue;
        }
        for (size_t off = 0; off < map_bytes; off += r->chunk_bytes) {
            const size_t n = std::min(r->chunk_bytes, map_bytes - off);
            const int s = r->acquire();
            hipError_t e = hipMemcpyAsync(r->base + (size_t)s * r->chunk_bytes, src + off, n, hipMemcpyDeviceToHost, r->copy_stream);
            if (e == hipSuccess) e = hipEventRecord(r->landed[s], r->copy_stream);
            if (e != hipSuccess) {
                r->unclaim(s);
                return fail(FFTCONV_ERR_HIP, "device-to-host copy failed: %s", hipGetErrorString(e));
            }
            r->submit(s, dst + off, n);
        }
    }
    HIP_TRY(hipEventRecord(r->copy_done[buf], r->copy_stream));
    return 0;
}

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

// Opt-in placement tuning of the intermediate (option tune_placement = k > 1).  On this memory system the
// output kernel runs in one of two states, 4 % apart, and WHICH physical allocations hold the intermediate
// and the maps decides it (profiles/r02x_placement_class_map.txt); nothing in user space can
// ask for the fast pairing, but it can be found: right after the intermediate was (re)allocated, up to k
// candidate allocations of it are timed with the real output kernel writing into the caller's map buffer
// (interleaved, after ~50 ms of load so that the clocks have settled), the fastest is kept, the others are
// freed.  The probes write into `out`, which the convolve that follows overwrites; they read the candidates as
// allocated (the driver hands out zeroed memory).  Blocking (~70 ms), once per allocation: what FFTW calls
// measuring at plan time.
int tune_intermediate_placement(fftconv_plan* p, int n, int nbY, float* out, size_t out_stride_per_map) {
    // (out_stride_per_map > 0: the call's batches write to out + first_map * stride, and every batch's
    // destination is probed -- an 18-GB map buffer spans several placement regions; 0: one staging buffer)
    const Geometry& g = p->g;
    const int k = (int)p->opt_tune_placement;
    p->Y.fresh = false;
    if (k < 2 || !g.fast_cols.ok || n < 1) return 0;
    {   // the tuner synchronises and frees: not inside a stream capture (the first convolve of a graph keeps its allocation)
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(p->stream, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
        if (cs != hipStreamCaptureStatusNone) return 0;
    }
    const int nbatch = out_stride_per_map ? std::min(16, (n + nbY - 1) / nbY) : 1;
    std::vector<DevBuf<c32>> cand((size_t)k);
    cand[0] = p->Y;
    p->Y = DevBuf<c32>();
    int nc = 1;
    // the states go with regions of physical memory some 10-100 GB wide (profiles/r02x_placement_class_map.txt), and
    // allocations made one after the other are neighbours: spacers (up to 12 GiB each, an eighth of what is free at
    // most; freed again below) put the candidates into different regions
    std::vector<void*> spacers;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
    const size_t spacer_bytes = std::min<size_t>((size_t)12 << 30, free_b / 8);
    for (; nc < k; nc++) {
        if (spacer_bytes >= ((size_t)1 << 30)) {
            void* sp = nullptr;
            if (hipMalloc(&sp, spacer_bytes) == hipSuccess) spacers.push_back(sp);
            else (void)hipGetLastError();
        }
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&cand[nc].p), cand[0].cap * sizeof(c32));
        if (e != hipSuccess) { (void)hipGetLastError(); cand[nc].p = nullptr; break; }   // as many as fit
        cand[nc].cap = cand[0].cap;
    }
    for (void* sp : spacers) (void)hipFree(sp);
    // every candidate reads the same contents (zeros): recycled allocations may hold anything, and what is timed
    // must be the placement, not NaNs or denormals in one of them (transient peak: k intermediates + the spacers)
    for (int c = 0; c < nc; c++)
        if (hipMemsetAsync(cand[c].p, 0, cand[c].cap * sizeof(c32), p->stream) != hipSuccess) (void)hipGetLastError();
    auto launch = [&](const DevBuf<c32>& y) -> hipError_t {   // the output launches of the whole call
        for (int b = 0; b < nbatch; b++) {
            const int ny = std::min(nbY, n - b * nbY);
            FastColsArgs fa = fast_cols_args(g, p->d, y.p, out + (size_t)b * nbY * out_stride_per_map, g.map_elems(), ny);
            hipError_t e = launch_fast_cols(g.M, g.fast_cols.T, fa, p->num_cus, p->stream);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    };
    int best = 0;
    hipError_t err = hipSuccess;
    hipEvent_t ev[2] = {nullptr, nullptr};
    std::vector<double> ms((size_t)nc, 0.0);
    do {
        if (nc < 2) break;
        if ((err = hipEventCreate(&ev[0])) != hipSuccess || (err = hipEventCreate(&ev[1])) != hipSuccess) break;
        // settle the clocks: ~50 ms of this kernel, measured with the first launch
        if ((err = hipEventRecord(ev[0], p->stream)) != hipSuccess || (err = launch(cand[0])) != hipSuccess ||
            (err = hipEventRecord(ev[1], p->stream)) != hipSuccess || (err = hipEventSynchronize(ev[1])) != hipSuccess) break;
        float one = 0.f;
        if ((err = hipEventElapsedTime(&one, ev[0], ev[1])) != hipSuccess) break;
        const int warm = std::min(200, std::max(2, (int)(50.0f / std::max(one, 0.05f))));
        for (int i = 0; i < warm && err == hipSuccess; i++) err = launch(cand[i % nc]);
        if (err != hipSuccess) break;
        for (int rep = 0; rep < 3 && err == hipSuccess; rep++)
            for (int c = 0; c < nc && err == hipSuccess; c++) {
                float t = 0.f;
                if ((err = hipEventRecord(ev[0], p->stream)) != hipSuccess || (err = launch(cand[c])) != hipSuccess ||
                    (err = hipEventRecord(ev[1], p->stream)) != hipSuccess || (err = hipEventSynchronize(ev[1])) != hipSuccess ||
                    (err = hipEventElapsedTime(&t, ev[0], ev[1])) != hipSuccess) break;
                ms[c] += t;
            }
        if (err != hipSuccess) break;
        for (int c = 1; c < nc; c++)
            if (ms[c] < ms[best]) best = c;
    } while (false);
    for (hipEvent_t e : ev)
        if (e) (void)hipEventDestroy(e);
    (void)hipStreamSynchronize(p->stream);
    for (int c = 0; c < nc; c++) {
        if (c == best) continue;
        if (cand[c].p) (void)hipFree(cand[c].p);
        cand[c].p = nullptr;
    }
    p->Y = cand[best];
    p->Y.fresh = false;
    p->tuned_candidates = nc;
    p->tuned_best = best;
    if (err != hipSuccess) return fail(FFTCONV_ERR_HIP, "placement tuning failed: %s", hipGetErrorString(err));
    return 0;
}

// Core of the per-kernel loop (src/cudaConvolutionFFT.cu:204-291) for n kernels of one size,
// packed on the device at dk ([n][F][kw][kh]).
int run_group_impl(fftconv_plan* p, int n, const float* dk, int kh, int kw, const Sink& sink) {
    const Geometry& g = p->g;
    if (!p->have_image) return fail(FFTCONV_ERR_NO_IMAGE, "no image spectrum: call fftconv_plan_set_image first");
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
    if (win && !g.fast_cols.ok) return fail(FFTCONV_ERR_INVALID_ARG, "an output window needs the specialised output kernel");
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
        if (p->opt_tune_placement > 1 && !win) {
            const bool direct = !cropped && !staged;   // the output kernel writes straight into the caller's packed buffer
            float* first_obase = cropped ? p->O.p : (staged ? stage.p : sink.packed);
            if (int rc = tune_intermediate_placement(p, direct ? n : std::min(nbY, n), nbY, first_obase, direct ? oe : 0)) return rc;
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
                HIP_TRY(launch_fast_rows(g.Lw, fast_rows_nz2(g, kw), fa, g.rows, ny, g.rows_wg_order, p->stream));
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
        if (e != hipSuccess) return fail(FFTCONV_ERR_HIP, "host-output copy failed: %s", hipGetErrorString(e));
    }
    FC_VERBOSE(p, "FFT done");                                        // src/cudaConvolutionFFT.cu:258
    return 0;
}

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
            return fail(FFTCONV_ERR_THREAD_SIZE,
                        "CUDA Thread Size must be 4 integers : THREAD_PER_BLOCK_H, THREAD_PER_BLOCK_W, "
                        "THREAD_PER_BLOCK_D, THREAD_PER_BLOCK_2D");
    return 0;
}


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
}  // namespace

int plan_create_internal(fftconv_plan** plan, int data_h, int data_w, int feature_dim, int max_kernel_h, int max_kernel_w, int gpu_id,
                         void* hip_stream, const fftconv_plan_options* options, bool cyclic);

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
    DevBuf<float> kstage;            // host kernels of a chunk, staged on the device once (every block convolves them)
    std::vector<float> hblk;         // host staging of one image block
    bool have_image = false;
    c32* spec_base() const { return specs_x ? specs_x : specs.p; }
    size_t spec_total() const { return spec_elems * (size_t)nblk; }
    size_t big_map() const { return (size_t)FH * FW; }
    void release() {
        if (sub) fftconv_plan_destroy(sub);
        sub = nullptr;
        specs.release(); big.release(); tmp.release(); blk.release(); kstage.release();
    }
};

namespace {

int tiled_unsupported(const char* what) {
    return fail(FFTCONV_ERR_UNSUPPORTED_SIZE, "%s is not available on a block-wise plan (the padded size does not fit one transform pass)", what);
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
    if (!ts) return fail(FFTCONV_ERR_ALLOC, "out of host memory");
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
            return fail(FFTCONV_ERR_UNSUPPORTED_SIZE, "kernels up to %dx%d are too large for the block-wise path", mkh, mkw);
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
                if (int rc = fftconv_plan_set_image(sub, ts->hblk.data(), FFTCONV_HOST)) return rc;   // synchronous for host input
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
            if (int rc = fftconv_plan_set_image(sub, ts->hblk.data(), FFTCONV_HOST)) return rc;   // synchronous for host input
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

// Overlap-save: every block's run stores its rectangle of the maps from the output kernel (OutWindow) -- no block maps, no
// summing pass, every element of the maps written once.  Kernels of equal size go through the block plan group by group
// (run_group); host kernels, and device kernels that are not consecutive in memory, are packed on the device once per call.
int tiled_convolve_save(fftconv_plan* p, int n, const float* const* kernels, const int* kh, const int* kw, int kernel_location,
                        float* const* out, int out_location, float* out_packed) {
    TiledState* ts = p->tiled;
    fftconv_plan* sub = ts->sub;
    const size_t big_map = ts->big_map();
    const size_t budget = (size_t)6 << 30;
    const int nc = out_packed ? n : (int)std::max<size_t>(1, std::min<size_t>((size_t)n, budget / (big_map * sizeof(float))));
    if (!out_packed)
        if (int rc = ts->big.ensure(big_map * nc)) return rc;
    FC_VERBOSE(p, "N Kernel: %d (block-wise, overlap-save: %d blocks, %d kernels per chunk)", n, ts->nblk, nc);
    struct Group { int first, count; const float* dk; };
    for (int k0 = 0; k0 < n; k0 += nc) {
        const int nk = std::min(nc, n - k0);
        float* big = out_packed ? out_packed + (size_t)k0 * big_map : ts->big.p;
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
        if (!out_packed) {
            for (int j = 0; j < nk; j++)
                HIP_TRY(hipMemcpyAsync(out[k0 + j], big + (size_t)j * big_map, big_map * sizeof(float),
                                       out_location == FFTCONV_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, sub->stream));
            if (out_location == FFTCONV_HOST || k0 + nc < n) HIP_TRY(hipStreamSynchronize(sub->stream));   // `big` is reused by the next chunk
        }
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
    if (!ts->have_image) return fail(FFTCONV_ERR_NO_IMAGE, "no image spectrum: call fftconv_plan_set_image first");
    for (int k = 0; k < n; k++) {
        if (!kernels[k]) return fail(FFTCONV_ERR_INVALID_ARG, "kernel %d is NULL", k);
        if (!out_packed && !out[k]) return fail(FFTCONV_ERR_INVALID_ARG, "output %d is NULL", k);   // everything checked before anything is queued
        if (kh[k] < 1 || kw[k] < 1 || kh[k] > ts->FH || kw[k] > ts->FW)      // src/cudaConvolutionFFT.cu:242
            return fail(FFTCONV_ERR_KERNEL_SHAPE,
                        "Kernel and Data must have the same number of features and kernel size should be smaller than data size");
        if (kh[k] > ts->mkh || kw[k] > ts->mkw)
            return fail(FFTCONV_ERR_KERNEL_EXCEEDS_MAX, "kernel %dx%d exceeds MAX_KERNEL %dx%d (block-wise path)", kh[k], kw[k], ts->mkh, ts->mkw);
    }
    if (ts->save) return tiled_convolve_save(p, n, kernels, kh, kw, kernel_location, out, out_location, out_packed);
    const size_t big_map = ts->big_map(), blk_map = sub->g.map_elems();
    const size_t budget = (size_t)6 << 30;
    const int nc = (int)std::max<size_t>(1, std::min<size_t>((size_t)n, budget / ((big_map + blk_map) * sizeof(float))));
    if (!out_packed)
        if (int rc = ts->big.ensure(big_map * nc)) return rc;
    if (int rc = ts->tmp.ensure(blk_map * nc)) return rc;
    std::vector<float*> tptr(nc);
    for (int j = 0; j < nc; j++) tptr[j] = ts->tmp.p + (size_t)j * blk_map;
    FC_VERBOSE(p, "N Kernel: %d (block-wise: %d blocks, %d kernels per chunk)", n, ts->nblk, nc);
    for (int k0 = 0; k0 < n; k0 += nc) {
        const int nk = std::min(nc, n - k0);
        float* big = out_packed ? out_packed + (size_t)k0 * big_map : ts->big.p;
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
            if (e != hipSuccess) return fail(FFTCONV_ERR_HIP, "overlap-add failed: %s", hipGetErrorString(e));
        }
        if (!out_packed) {
            for (int j = 0; j < nk; j++) {
                HIP_TRY(hipMemcpyAsync(out[k0 + j], big + (size_t)j * big_map, big_map * sizeof(float),
                                       out_location == FFTCONV_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, sub->stream));
            }
            if (out_location == FFTCONV_HOST || k0 + nc < n) HIP_TRY(hipStreamSynchronize(sub->stream));   // `big` is reused by the next chunk
        }
    }
    FC_VERBOSE(p, "FFT done");
    return 0;
}

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
    size_t max_bytes = (size_t)48 << 30;
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
        b += ts->specs.bytes() + ts->big.bytes() + ts->tmp.bytes() + ts->blk.bytes() + ts->kstage.bytes();
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

thread_local fftconv_call_timing g_call_timing = {0, 0, 0, 0, 0, 0};
double ms_since(const std::chrono::steady_clock::time_point& t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace

extern "C" {

int fftconv_cache_configure(int max_plans, size_t max_bytes) {
    if (max_plans < 0) return fail(FFTCONV_ERR_INVALID_ARG, "max_plans must not be negative");
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
    if (!timing) return fail(FFTCONV_ERR_INVALID_ARG, "NULL argument");
    *timing = g_call_timing;
    return 0;
}

int fftconv_fft_size16(int data_size) { return fft_size16(data_size); }
int fftconv_fft_size_pow2(int data_size) { return fft_size_pow2(data_size); }

const char* fftconv_last_error(void) { return g_last_error.c_str(); }

const char* fftconv_version(void) { return "fftconv-mi355x 0.1 (gfx950)"; }

int fftconv_device_count(int* count) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (count) *count = (e == hipSuccess) ? n : 0;
    if (e != hipSuccess || n == 0) return fail(FFTCONV_ERR_NO_DEVICE, "no HIP device: %s", hipGetErrorString(e));
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
int plan_create_internal(fftconv_plan** plan, int data_h, int data_w, int feature_dim, int max_kernel_h, int max_kernel_w, int gpu_id,
                         void* hip_stream, const fftconv_plan_options* options, bool cyclic) {
    if (!plan) return fail(FFTCONV_ERR_INVALID_ARG, "plan is NULL");
    *plan = nullptr;
    if (data_h < 1 || data_w < 1 || feature_dim < 1) return fail(FFTCONV_ERR_INVALID_ARG, "Invalid data input");
    if (max_kernel_h < 1 || max_kernel_w < 1) return fail(FFTCONV_ERR_INVALID_ARG, "Invalid maximum kernel size");
    int ndev = 0;
    if (int rc = fftconv_device_count(&ndev)) return rc;
    if (gpu_id < 0) HIP_TRY(hipGetDevice(&gpu_id));
    if (gpu_id >= ndev) return fail(FFTCONV_ERR_NO_DEVICE, "gpu_id %d out of range (%d devices)", gpu_id, ndev);
    fftconv_plan* p = new (std::nothrow) fftconv_plan();
    if (!p) return fail(FFTCONV_ERR_ALLOC, "out of host memory");
    if (options && options->struct_size < kOptionsMinSize) {
        delete p;
        return fail(FFTCONV_ERR_INVALID_ARG, "fftconv_plan_options.struct_size is not set");
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
            (void)fail(FFTCONV_ERR_UNSUPPORTED_SIZE, "no specialised kernels for a %dx%d block transform with kernels up to %dx%d", data_h, data_w,
                       max_kernel_h, max_kernel_w);
        else if (options_no_blockwise(options) || tune.exact_window)
            (void)fail(FFTCONV_ERR_UNSUPPORTED_SIZE, "sizes %dx%dx%d with kernels up to %dx%d do not fit the single-pass LDS transform%s", data_h,
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
        if (e != hipSuccess) { rc = fail(FFTCONV_ERR_HIP, "kernel setup failed: %s", hipGetErrorString(e)); break; }
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
            // dynamic tile queue of the persistent column kernels: on by default (option "dynamic_tiles"; alone on the GPU it
            // measures equal or up to 2 % faster than the static deal, beside another kernel it loses half as much:
            // profiles/r05a_contention_ab.txt).  Zeroed once; every launch leaves the counters at zero.
            if ((rc = p->queue.ensure(FC_QUEUE_WORDS))) break;
            if (hipMemset(p->queue.p, 0, FC_QUEUE_WORDS * sizeof(int)) != hipSuccess) { rc = fail(FFTCONV_ERR_HIP, "hipMemset of the tile queue failed"); break; }
            p->opt_dynamic_tiles = 1;
            p->d.queue = p->queue.p;
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
        if (!g_live_plans.erase(plan)) return fail(FFTCONV_ERR_INVALID_ARG, "not a live plan");
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
    if (!plan || !info) return fail(FFTCONV_ERR_INVALID_ARG, "NULL argument");
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
        info->map_bytes = info->out_map_bytes = ts->big_map() * sizeof(float);
        info->out_h = ts->FH; info->out_w = ts->FW;
        fftconv_plan_info si;
        if (fftconv_plan_get_info(ts->sub, &si) == 0)
            info->workspace_bytes = si.workspace_bytes + ts->big.bytes() + ts->tmp.bytes() + ts->blk.bytes() + (ts->specs_x ? 0 : ts->specs.bytes());
    }
    return 0;
}

int fftconv_plan_set_image(fftconv_plan* plan, const float* data, int location) {
    if (!plan || !data) return fail(FFTCONV_ERR_INVALID_ARG, "Invalid data input");
    if (location != FFTCONV_HOST && location != FFTCONV_DEVICE) return fail(FFTCONV_ERR_INVALID_ARG, "bad location");
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
    if (!plan) return fail(FFTCONV_ERR_INVALID_ARG, "plan is NULL");
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
    if (!p || !spectrum) return fail(FFTCONV_ERR_INVALID_ARG, "NULL argument");
    if (location != FFTCONV_HOST && location != FFTCONV_DEVICE) return fail(FFTCONV_ERR_INVALID_ARG, "bad location");
    if (p->tiled) return tiled_unsupported("the spectrum in the reference's order");
    const Geometry& g = p->g;
    if (!g.exact_window)
        return fail(FFTCONV_ERR_UNSUPPORTED_SIZE,
                    "this plan transforms %dx%d, not the %dx%d window: create it with fftconv_plan_options.exact_window = 1 to exchange "
                    "spectra in the reference's order", g.Lh, g.Lw, g.fft_h, g.fft_w);
    if (to_natural && !p->have_image) return fail(FFTCONV_ERR_NO_IMAGE, "no image spectrum: call fftconv_plan_set_image first");
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
    if (!plan) return fail(FFTCONV_ERR_INVALID_ARG, "plan is NULL");
    const size_t need = (plan->tiled ? plan->tiled->spec_total() : plan->g.spectrum_elems()) * sizeof(c32);
    if (device_ptr) {
        if (bytes < need || (reinterpret_cast<uintptr_t>(device_ptr) & 15))
            return fail(FFTCONV_ERR_INVALID_ARG, "spectrum buffer too small (%zu < %zu bytes) or not 16-byte aligned", bytes, need);
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
    if (!plan) return fail(FFTCONV_ERR_INVALID_ARG, "plan is NULL");
    if (plan->tiled) {
        if (!plan->tiled->spec_base()) return fail(FFTCONV_ERR_NO_IMAGE, "the plan has no spectrum buffer yet (fftconv_plan_spectrum / fftconv_plan_use_spectrum_buffer)");
        plan->tiled->have_image = true;
        return 0;
    }
    if (!plan->spec()) return fail(FFTCONV_ERR_NO_IMAGE, "the plan has no spectrum buffer yet (fftconv_plan_spectrum / fftconv_plan_use_spectrum_buffer)");
    plan->have_image = true;
    return 0;
}

int fftconv_plan_convolve_packed(fftconv_plan* plan, int n_kernel, const float* kernels_device, int kernel_h,
                                 int kernel_w, float* out_device) {
    if (!plan || n_kernel < 0) return fail(FFTCONV_ERR_INVALID_ARG, "Wrong number of inputs");
    if (n_kernel == 0) return 0;
    if (!kernels_device || !out_device) return fail(FFTCONV_ERR_INVALID_ARG, "NULL kernel or output pointer");
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
    if (!plan || n_kernel < 0) return fail(FFTCONV_ERR_INVALID_ARG, "Wrong number of inputs");
    if (n_kernel == 0) return 0;
    if (!kernels_device) return fail(FFTCONV_ERR_INVALID_ARG, "NULL kernel pointer");
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
    if (!plan || n_kernel < 0) return fail(FFTCONV_ERR_INVALID_ARG, "Wrong number of inputs");
    if (n_kernel == 0) return 0;
    if (!kernels || !kernel_h || !kernel_w || !out) return fail(FFTCONV_ERR_INVALID_ARG, "Kernel must be a cell array");
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
    if (!p->have_image) return fail(FFTCONV_ERR_NO_IMAGE, "no image spectrum: call fftconv_plan_set_image first");
    // validate everything up front so nothing is launched on a bad cell (the reference fails
    // mid-loop and leaks: SURVEY D3)
    for (int k = 0; k < n_kernel; k++) {
        if (!kernels[k] || !out[k]) return fail(FFTCONV_ERR_INVALID_ARG, "kernel or output %d is NULL", k);
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
    if (!plan) return fail(FFTCONV_ERR_INVALID_ARG, "plan is NULL");
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
    if (!plan) return fail(FFTCONV_ERR_INVALID_ARG, "plan is NULL");
    if (int rc = use_device(plan)) return rc;
    if (!plan->tiled)
        if (int rc = flush_pending_prepare(plan)) return rc;
    HIP_TRY(hipStreamSynchronize(plan->stream));
    return 0;
}

int fftconv_plan_set_option(fftconv_plan* plan, const char* name, long value) {
    if (!plan || !name) return fail(FFTCONV_ERR_INVALID_ARG, "NULL argument");
    if (plan->tiled) {      // block-wise: options act on the block plan; the window options have no block-wise form
        if (!strcmp(name, "output_region") && value != 0) return tiled_unsupported("output_region");
        if (!strcmp(name, "verbose")) plan->opt_verbose = value != 0;
        return fftconv_plan_set_option(plan->tiled->sub, name, value);
    }
    if (!strcmp(name, "tune_placement")) { plan->opt_tune_placement = value < 0 ? 0 : (value > 8 ? 8 : value); return 0; }
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
        // the persistent column kernels (output columns; image and kernel columns forwards) take their tiles from a queue in
        // device memory (1, the default where the plan has specialised column kernels) instead of a fixed share per workgroup
        // (0): see fast_cols.hpp: TileQueue.  A step that shares the GPU with another kernel -- the broadcast of a multi-GPU
        // run, another library's work -- loses half as much with the queue; 0 is kept for A/B runs.
        if (value && !plan->g.fast_cols.ok) value = 0;      // (generic kernels: one workgroup per tile, dealt by the hardware)
        if (value && !plan->queue.p) {     // zeroed once: every launch leaves the counters at zero (fast_cols.hpp: queue_leave)
            if (int rc = use_device(plan)) return rc;
            if (int rc = plan->queue.ensure(FC_QUEUE_WORDS)) return rc;
            HIP_TRY(hipMemset(plan->queue.p, 0, FC_QUEUE_WORDS * sizeof(int)));
        }
        plan->opt_dynamic_tiles = value != 0;
        plan->d.queue = value ? plan->queue.p : nullptr;
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
#if FC_ROWS_TIMELINE || FC_COLS_TIMELINE
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
        else if (value != 0) return fail(FFTCONV_ERR_INVALID_ARG, "output_region is 0 (window), 1 (full), 2 (same), 3 (valid) or 4 (pow2 window)");
        if (oh < 1 || ow < 1) return fail(FFTCONV_ERR_INVALID_ARG, "output_region %ld is empty for %dx%d data and %dx%d kernels", value, g.H, g.W, g.max_kh, g.max_kw);
        if (int rc = use_device(plan)) return rc;
        HIP_TRY(hipStreamSynchronize(plan->stream));
        plan->release_ring();          // sized for the map bytes
        plan->opt_region = value; plan->out_h = oh; plan->out_w = ow; plan->off_h = fh; plan->off_w = fw;
        return 0;
    }
    if (!strcmp(name, "flip_kernels")) { plan->opt_flip_kernels = value != 0; plan->prepared.dk = nullptr; plan->deferred.on = false; return 0; }
    if (!strcmp(name, "host_pinned")) { plan->opt_host_pinned = value != 0; return 0; }
    if (!strcmp(name, "host_min_kb")) {
        if (value < 0 || value > (1 << 30)) return fail(FFTCONV_ERR_INVALID_ARG, "option '%s' out of range", name);
        plan->opt_host_min_kb = value;
        return 0;
    }
    if (!strcmp(name, "host_stream") || !strcmp(name, "host_threads") || !strcmp(name, "host_chunk_kb") || !strcmp(name, "host_slots")) {
        if (value < 0 || value > (1 << 20)) return fail(FFTCONV_ERR_INVALID_ARG, "option '%s' out of range", name);
        if (int rc = use_device(plan)) return rc;
        HIP_TRY(hipStreamSynchronize(plan->stream));
        plan->release_ring();      // rebuilt with the new shape by the next host-output call
        if (!strcmp(name, "host_stream") && value > 2) return fail(FFTCONV_ERR_INVALID_ARG, "host_stream is 0, 1 or 2");
        (!strcmp(name, "host_stream") ? plan->opt_host_stream : name[5] == 't' ? plan->opt_host_threads
         : name[5] == 'c' ? plan->opt_host_chunk_kb : plan->opt_host_slots) = value;
        return 0;
    }
    return fail(FFTCONV_ERR_INVALID_ARG, "unknown option '%s'", name);
}

int fftconv_plan_get_option(fftconv_plan* plan, const char* name, long* value) {
    if (!plan || !name || !value) return fail(FFTCONV_ERR_INVALID_ARG, "null argument");
    if (!strcmp(name, "blockwise")) { *value = plan->tiled ? plan->tiled->nblk : 0; return 0; }   // read-only: number of blocks (0 = one pass)
    if (!strcmp(name, "overlap_save")) { *value = plan->tiled && plan->tiled->save ? 1 : 0; return 0; }   // read-only: blocks stored by the output kernel (1) or summed (0)
    if (plan->tiled) return fftconv_plan_get_option(plan->tiled->sub, name, value);
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
    return fail(FFTCONV_ERR_INVALID_ARG, "unknown option '%s'", name);
}

int fftconv_plan_get_profile(fftconv_plan* plan, fftconv_profile* prof, int reset) {
    if (!plan || !prof) return fail(FFTCONV_ERR_INVALID_ARG, "NULL argument");
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
        return fail(FFTCONV_ERR_INVALID_ARG, "bad kernel location");
    // argument checks in the reference's order (src/cudaConvolutionFFT.cu:45-89)
    if (!data || data_h < 1 || data_w < 1 || feature_dim < 1) return fail(FFTCONV_ERR_INVALID_ARG, "Invalid data input");
    if (n_kernel < 0 || (n_kernel > 0 && (!kernels || !kernel_h || !kernel_w || !out)))
        return fail(FFTCONV_ERR_INVALID_ARG, "Kernel must be a cell array");
    if (int rc = check_thread_size(thread_size, n_thread_size)) return rc;
    if (kernel_f)
        for (int k = 0; k < n_kernel; k++)
            if (kernel_f[k] != feature_dim)  // src/cudaConvolutionFFT.cu:242
                return fail(FFTCONV_ERR_KERNEL_SHAPE,
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
    if (!p)
        if (int rc = fftconv_plan_create_ex(&p, data_h, data_w, feature_dim, max_kernel_h, max_kernel_w, gpu_id, nullptr, options)) return rc;
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
        const std::string keep_err = g_last_error;
        const int rcm = p->pin_k.mark(p->stream);
        if (!rc) rc = rcm; else g_last_error = keep_err;
        if (!rc) {
            hipError_t e = hipStreamSynchronize(p->stream);
            if (e != hipSuccess) rc = fail(FFTCONV_ERR_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(e));
        }
    } else if (!rc) {
        rc = fftconv_plan_convolve(p, n_kernel, kernels, kernel_h, kernel_w, kernel_location, out, FFTCONV_HOST);
    }
    if (rc) p->deferred.on = false;       // (a failed image leaves no request behind in a plan that goes back into the cache)
    tm.convolve_ms = ms_since(t2);
    const auto t3 = std::chrono::steady_clock::now();
    std::string keep = g_last_error;
    // argument-class failures were found before anything was queued and leave the plan as it was; after a HIP or
    // allocation failure the plan is not trusted again
    const bool reusable = rc == 0 || rc == FFTCONV_ERR_INVALID_ARG || rc == FFTCONV_ERR_KERNEL_SHAPE || rc == FFTCONV_ERR_KERNEL_EXCEEDS_MAX ||
                          rc == FFTCONV_ERR_THREAD_SIZE;
    if (reusable) cache_put(key, p);
    else fftconv_plan_destroy(p);
    if (rc) g_last_error = keep;
    tm.release_ms = ms_since(t3);
    tm.total_ms = ms_since(t0);
    g_call_timing = tm;
    if (options_verbose(options))
        fprintf(stderr, "fftconv: one-shot call %.3f ms = plan %.3f (%s) + image %.3f + %d kernels %.3f + release %.3f\n", tm.total_ms, tm.plan_ms,
                tm.cache_hit ? "cached" : "created", tm.image_ms, n_kernel, tm.convolve_ms, tm.release_ms);
    return rc;
}

int fftconv_fft_data(const float* data, int data_h, int data_w, int feature_dim, int kernel_h, int kernel_w, int gpu_id,
                     fftconv_plan** fft_data) {
    if (!fft_data) return fail(FFTCONV_ERR_INVALID_ARG, "Invalid input to MEX file.");
    *fft_data = nullptr;
    if (!data) return fail(FFTCONV_ERR_INVALID_ARG, "Invalid input to MEX file.");  // src/cudaFFTData.cu:49-54
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
    if (!fft_data || !fftconv_plan_is_live(fft_data)) return fail(FFTCONV_ERR_INVALID_ARG, "Invalid input to MEX file.");  // src/cudaConvFFTData.cu:68
    if (int rc = check_thread_size(thread_size, n_thread_size)) return rc;
    if (kernel_f)
        for (int k = 0; k < n_kernel; k++)
            if (kernel_f[k] != fft_data->g.F)
                return fail(FFTCONV_ERR_KERNEL_SHAPE,
                            "Kernel and Data must have the same number of features and kernel size should be smaller than data size");
    return fftconv_plan_convolve(fft_data, n_kernel, kernels, kernel_h, kernel_w, FFTCONV_AUTO, out, FFTCONV_HOST);
}

}  // extern "C"
