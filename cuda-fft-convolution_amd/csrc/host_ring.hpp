// host_ring.hpp -- the host-output streaming object of a plan (see host_ring.cpp); included by plan_internal.hpp
#pragma once
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

namespace fc {

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

}  // namespace fc
