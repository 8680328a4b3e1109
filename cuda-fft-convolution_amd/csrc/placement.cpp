// placement.cpp -- placement tuning of a plan's intermediate (plan option tune_placement; automatic for large launches on a mostly free device).
#include "plan_internal.hpp"

namespace fc {

// Option tune_placement left alone (-1): how many candidates the first convolve of a fresh intermediate tries.  Tuning pays where a launch is long
// enough for 3 % of the output kernel to matter and costs ~100 ms plus transient allocations, so: launches that write at least 2 GiB of maps (cfg3 and
// cfg4's share: 4.6 GB; cfg5, cfg2, cfg1: never) on a device that is at least 60 % free at that moment (a process that has the GPU to itself; several
// processes sharing one device -- the rehearsals of the N > 1 tests -- stop tuning as soon as the others' buffers are there), five candidates.  Why a
// default: which state an untuned plan gets is a property of the box -- six of six fresh processes slow on one (369 against 373.5 Gpixel-filters/s at
// cfg3), five of six fast on another (profiles/r05q_default_vs_tuned_final_library*.txt).
int placement_auto_candidates(const fftconv_plan* p, size_t launch_map_bytes) {
    if (!p->g.fast_cols.ok || launch_map_bytes < ((size_t)2 << 30)) return 0;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return 0; }
    if (free_b < total_b / 10 * 6) return 0;
    return 5;
}

// Placement tuning of the intermediate (option tune_placement = k > 1, or the automatic choice above).  On this memory system the
// output kernel runs in one of two states, 4 % apart, and WHICH physical allocations hold the intermediate
// and the maps decides it (profiles/r02x_placement_class_map.txt); nothing in user space can
// ask for the fast pairing, but it can be found: right after the intermediate was (re)allocated, up to k
// candidate allocations of it are timed with the real output kernel writing into the caller's map buffer
// (interleaved, after ~50 ms of load so that the clocks have settled), the fastest is kept, the others are
// freed.  The probes write into `out`, which the convolve that follows overwrites; they read the candidates as
// allocated (the driver hands out zeroed memory).  Blocking (~70 ms), once per allocation: what FFTW calls
// measuring at plan time.
int tune_intermediate_placement(fftconv_plan* p, int k, int n, int nbY, float* out, size_t out_stride_per_map) {
    // (out_stride_per_map > 0: the call's batches write to out + first_map * stride, and every batch's
    // destination is probed -- an 18-GB map buffer spans several placement regions; 0: one staging buffer)
    const Geometry& g = p->g;
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
    if (err != hipSuccess) return api_fail(FFTCONV_ERR_HIP, "placement tuning failed: %s", hipGetErrorString(err));
    return 0;
}

}  // namespace fc
