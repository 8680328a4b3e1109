// stage3_exchange.hip -- what the hand-over between stage 2 and stage 3 of the row transform costs, two ways (DESIGN.md 2:
// north_star's "wavefront shuffle for the small-DFT stage" against what the kernels do).
//
// The row kernel (fast_rows_multi.hpp, 4224 = 8 x 24 x 22) leaves stage 2 with R2 = 24 values per thread (one per stage-2
// output c of its butterfly (c1, b)) and enters stage 3 with R3 = 22 values per thread (the run b = 0..21 of one (c1, c)):
// a 24 x 22 transpose among the lanes that share c1.
//   kind 0  through LDS, as built: 24 ds_write_b64 at stride R3 cells, barrier, 11 ds_read_b128 of the thread's own run;
//   kind 1  through the lanes: one ds_bpermute_b32 per dword a thread receives (2 x 22 = 44) -- a LOWER bound of the shuffle
//           form: it leaves out the selects that pick which of its 24 registers a lane sends in each round (registers cannot
//           be indexed by a lane-varying value without v_cndmask chains or scratch);
//   kind 2  kind 1 plus those selects in their cheapest form: a barrel of v_cndmask over the 24 x 2 source registers per
//           round (5 select levels per dword sent).
// Same launch shape as the row kernel: 192 threads per workgroup, 4 workgroups per CU (38 KB of LDS each), 3 waves per SIMD.
// Prints nanoseconds and shader cycles per hand-over (per thread-run of one row map) and per row of 4224 points.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int R2 = 24, R3 = 22, NT = 192, L = 8 * R2 * R3, ITER = 2000;

template <int KIND>
__global__ void __launch_bounds__(NT, 3) k_exchange(float* out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];      // 38 KB: the row + tables of the real kernel
    const int t = threadIdx.x;
    f2 v[R2];
    for (int c = 0; c < R2; c++) v[c] = f2{(float)(t + c), (float)(t - c)};
    f2* row = reinterpret_cast<f2*>(lds);
    // stage-2 butterfly (c1, b) of thread t < 176: cells c1 * 528 + c * 22 + b; stage-3 run of thread t: cells t * 22 ...
    const int c1 = t / R3, b = t - c1 * R3;
    f2 acc = {0.f, 0.f};
    for (int it = 0; it < ITER; it++) {
        if (KIND == 0) {
            if (t < 8 * R3) {
#pragma unroll
                for (int c = 0; c < R2; c++) row[c1 * (R2 * R3) + c * R3 + b] = v[c];
            }
            __syncthreads();
            f2 w[R3];
#pragma unroll
            for (int h = 0; h < R3 / 2; h++) {
                const f4 x = *reinterpret_cast<const f4*>(&row[t * R3 + 2 * h]);
                w[2 * h] = f2{x.x, x.y};
                w[2 * h + 1] = f2{x.z, x.w};
            }
            __syncthreads();
#pragma unroll
            for (int c = 0; c < R3; c++) { v[c].x += w[c].y * 1e-9f; v[c].y += w[c].x * 1e-9f; }
        } else {
            // lane t receives, in round r, one complex value from lane (group base + (b + r) % 22): 2 bpermutes per round
            const int lane = t & 63;
            f2 w[R3];
#pragma unroll
            for (int r = 0; r < R3; r++) {
                const int src = (lane & ~31) | ((lane + r) & 31);
                f2 send = v[r % R2];
                if (KIND == 2) {
                    // which register a lane sends depends on the lane: a select tree over the 24 candidates (5 levels)
                    const int want = (lane + r) % R2;
#pragma unroll
                    for (int c = 0; c < R2; c++) {
                        const bool take = (want == c);
                        send.x = take ? v[c].x : send.x;
                        send.y = take ? v[c].y : send.y;
                    }
                }
                w[r].x = __int_as_float(__builtin_amdgcn_ds_bpermute(src << 2, __float_as_int(send.x)));
                w[r].y = __int_as_float(__builtin_amdgcn_ds_bpermute(src << 2, __float_as_int(send.y)));
            }
#pragma unroll
            for (int c = 0; c < R3; c++) { v[c].x += w[c].y * 1e-9f; v[c].y += w[c].x * 1e-9f; }
        }
    }
    for (int c = 0; c < R2; c++) acc += v[c];
    out[blockIdx.x * NT + t] = acc.x + acc.y + lds[(t * 7) % 64] * 0.f;
}

int main() {
    int dev = 0;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    float* out;
    CHECK(hipMalloc(&out, (size_t)cus * 4 * NT * sizeof(float)));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const size_t lds = 38 * 1024;
    const char* names[] = {"LDS round trip (24 ds_write_b64, barrier, 11 ds_read_b128, barrier)", "44 ds_bpermute_b32 (lower bound: no selects)",
                           "44 ds_bpermute_b32 + per-round register selects"};
    auto run = [&](int kind, auto kern) {
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(kern, dim3(cus * 4), dim3(NT), lds, 0, out);
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(kern, dim3(cus * 4), dim3(NT), lds, 0, out);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, a, b));
        // one hand-over per iteration and workgroup = one row map's stage-2 -> stage-3 exchange; 4 workgroups per CU run side by side
        const double ns_per_row = ms * 1e6 / ITER / 4.0;       // CU time per row map
        printf("kind %d  %-72s %8.3f ms  %7.1f ns of CU time per row map (of ~2690 per row map in the kernel)\n", kind, names[kind], ms, ns_per_row);
    };
    run(0, k_exchange<0>);
    run(1, k_exchange<1>);
    run(2, k_exchange<2>);
    return 0;
}
