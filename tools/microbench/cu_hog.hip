// cu_hog.hip -- a stand-in for "another kernel on the GPU" (bench.py --contend, tools/contention_ab.sh).
//
// On the ranks of a multi-GPU run the RCCL broadcast of the next step's image spectrum is co-resident with this step's
// maps (multi_gpu.py: depth-2 pipeline; src/cudaConvFFTDataStreams.cu:279-289,338-447 is the intent): a collective's
// channels are a few dozen workgroups, one per CU, that sit there for the length of the transfer.  No multi-GPU box is
// available to the builder, so this library puts K such workgroups on the one GPU: each holds `lds_bytes` of LDS (small:
// the hot kernels' workgroups still fit beside it and only share the CU's issue slots; large: the persistent output
// kernel's 148-KB workgroup does NOT fit, which is the case the static tile deal cannot survive) and spins on light
// vector work until the host says stop (a flag in mapped host memory) or `max_ms` have passed -- every wave reaches that
// exit.  Each workgroup records the XCC and CU it ran on, so the caller can report how many distinct CUs were held.
//
// Two forms: cu_hog_run(stream, ...) queues ONE launch that holds its CUs for duration_us on the caller's stream -- the
// per-step broadcast of the pipeline, queued where the broadcast would be (multi_gpu.FilterShardedConvolver side_work); and
// cu_hog_start / cu_hog_stop, workgroups that stay for a whole timed region (small LDS only: a workgroup that cannot be
// placed beside them would wait for the stop that comes after it).
// C ABI (ctypes): cu_hog_run(stream, blocks, threads, lds_bytes, duration_us) -> 0 / HIP error; cu_hog_last_places(&distinct_cus);
//                 cu_hog_start(blocks, threads, lds_bytes, max_ms) -> 0 / HIP error; cu_hog_stop(&distinct_cus, &ran_ms) -> 0.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC cu_hog.hip -o libcuhog.so   (test / measurement tooling, not product)
#include <hip/hip_runtime.h>

#include <set>

namespace {

extern __shared__ __attribute__((aligned(16))) unsigned char hog_smem[];

__global__ void __launch_bounds__(1024) k_hog(const int* stop, unsigned long long max_ticks, unsigned* where, float* sink, int lds_bytes) {
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;              // HW_REG_XCC_ID[3:0]
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);                      // HW_REG_HW_ID: CU in [11:8], SH [12], SE [15:13]
        where[blockIdx.x] = (xcc << 16) | ((hw >> 8) & 0xffu);
    }
    float* l = reinterpret_cast<float*>(hog_smem);
    for (int i = threadIdx.x; i < lds_bytes / 4; i += blockDim.x) l[i] = (float)i;       // the LDS is really in use
    __syncthreads();
    float a = (float)threadIdx.x, b = 1.0001f;
    for (;;) {
        // ~2 us of light work: a collective's channel mostly waits on the link and copies a little
        for (int i = 0; i < 64; i++) {
            a = __builtin_fmaf(a, b, 0.5f);
            __builtin_amdgcn_s_sleep(8);
        }
        int s = 0;
        if ((threadIdx.x & 63) == 0) s = __hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        s = __shfl(s, 0);
        if (s || wall_clock64() - t0 > max_ticks) break;
    }
    if (a == 1.2345e-30f) sink[0] = a + l[threadIdx.x % (lds_bytes / 4 > 0 ? lds_bytes / 4 : 1)];
}

struct Hog {
    int* stop_h = nullptr;          // mapped host memory
    int* stop_d = nullptr;
    unsigned* where = nullptr;
    float* sink = nullptr;
    hipStream_t s = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int blocks = 0;
    bool running = false;
} g;

int setup() {
    hipError_t e;
    if (g.stop_h) return 0;
    if ((e = hipHostMalloc(reinterpret_cast<void**>(&g.stop_h), 64, hipHostMallocMapped)) != hipSuccess) return (int)e;
    if ((e = hipHostGetDevicePointer(reinterpret_cast<void**>(&g.stop_d), g.stop_h, 0)) != hipSuccess) return (int)e;
    if ((e = hipMalloc(reinterpret_cast<void**>(&g.where), 256 * sizeof(unsigned))) != hipSuccess) return (int)e;
    if ((e = hipMalloc(reinterpret_cast<void**>(&g.sink), 64)) != hipSuccess) return (int)e;
    if ((e = hipStreamCreateWithFlags(&g.s, hipStreamNonBlocking)) != hipSuccess) return (int)e;
    if ((e = hipEventCreate(&g.e0)) != hipSuccess || (e = hipEventCreate(&g.e1)) != hipSuccess) return (int)e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_hog), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return (int)e;
    *g.stop_h = 0;
    return 0;
}

}  // namespace

extern "C" {

// one launch on the caller's stream: `blocks` workgroups hold a CU each for duration_us (the stop flag stays 0)
int cu_hog_run(void* stream, int blocks, int threads, int lds_bytes, int duration_us) {
    if (blocks < 1 || blocks > 256 || threads < 64 || threads > 1024 || lds_bytes < 0 || lds_bytes > 160 * 1024 || duration_us < 1 || duration_us > 10000000) return -1;
    if (int rc = setup()) return rc;
    if (g.running) return -2;
    g.blocks = blocks;
    hipLaunchKernelGGL(k_hog, dim3(blocks), dim3(threads), (size_t)lds_bytes, reinterpret_cast<hipStream_t>(stream), g.stop_d,
                       (unsigned long long)duration_us * 100ull, g.where, g.sink, lds_bytes);
    return (int)hipGetLastError();
}

// distinct places (XCC, SE / SH / CU) of the workgroups of the last launch (the device must be idle: call after a synchronise)
int cu_hog_last_places(int* distinct_cus) {
    if (!g.where || g.blocks < 1) return -1;
    unsigned w[256];
    hipError_t e = hipMemcpy(w, g.where, g.blocks * sizeof(unsigned), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return (int)e;
    std::set<unsigned> places(w, w + g.blocks);
    if (distinct_cus) *distinct_cus = (int)places.size();
    return 0;
}

int cu_hog_start(int blocks, int threads, int lds_bytes, int max_ms) {
    if (g.running || blocks < 1 || blocks > 256 || threads < 64 || threads > 1024 || lds_bytes < 0 || lds_bytes > 160 * 1024 || max_ms < 1) return -1;
    hipError_t e;
    if (int rc = setup()) return rc;
    *g.stop_h = 0;
    g.blocks = blocks;
    if ((e = hipEventRecord(g.e0, g.s)) != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_hog, dim3(blocks), dim3(threads), (size_t)lds_bytes, g.s, g.stop_d, (unsigned long long)max_ms * 100000ull, g.where, g.sink, lds_bytes);
    if ((e = hipGetLastError()) != hipSuccess) return (int)e;
    if ((e = hipEventRecord(g.e1, g.s)) != hipSuccess) return (int)e;
    g.running = true;
    return 0;
}

// stops the workgroups, waits for them; distinct_cus: how many different (XCC, SE/SH/CU) places they held
int cu_hog_stop(int* distinct_cus, float* ran_ms) {
    if (!g.running) return -1;
    __atomic_store_n(g.stop_h, 1, __ATOMIC_SEQ_CST);
    hipError_t e = hipStreamSynchronize(g.s);
    g.running = false;
    if (e != hipSuccess) return (int)e;
    unsigned w[256];
    if ((e = hipMemcpy(w, g.where, g.blocks * sizeof(unsigned), hipMemcpyDeviceToHost)) != hipSuccess) return (int)e;
    std::set<unsigned> places(w, w + g.blocks);
    if (distinct_cus) *distinct_cus = (int)places.size();
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, g.e0, g.e1) != hipSuccess) ms = -1.f;
    if (ran_ms) *ran_ms = ms;
    return 0;
}

}  // extern "C"
