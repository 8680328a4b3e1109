// host_ring.cpp -- host-output streaming of libfftconv.so: set-up of a plan's copy threads / pinned
// ring and the queueing of finished maps (the object itself: host_ring.hpp).
#include <new>

#include "plan_internal.hpp"

namespace fc {

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
    if (!r) return api_fail(FFTCONV_ERR_ALLOC, "out of host memory");
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
        return api_fail(FFTCONV_ERR_HIP, "host-output ring setup failed: %s", hipGetErrorString(e));
    }
    for (int i = 0; i < nthreads; i++) r->workers.emplace_back([r, i] { r->work(i); });
    p->ring = r;
    return 0;
}

static bool caller_pinned(const void* ptr) {
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
                return api_fail(FFTCONV_ERR_HIP, "device-to-host copy failed: %s", hipGetErrorString(e));
            }
            r->submit(s, dst + off, n);
        }
    }
    HIP_TRY(hipEventRecord(r->copy_done[buf], r->copy_stream));
    return 0;
}

}  // namespace fc
