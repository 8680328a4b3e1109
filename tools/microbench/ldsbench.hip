// ldsbench.hip -- LDS instruction throughput in the shapes the row kernel uses: ds_read_b64 vs ds_read2_b64
// (same bytes), ds_write_b64 vs ds_write2_b64, ds_read_b128 / ds_write_b128; lanes read consecutive 8-byte
// elements (conflict-free).  16 independent accesses are issued, then one s_waitcnt; 1..4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int ITER = 2048;

template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t = threadIdx.x;
    for (int i = t; i < 16384; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    const unsigned a8 = (unsigned)t * 8u;      // byte address of this lane's 8-byte element (consecutive lanes)
    const unsigned a16 = (unsigned)t * 16u;
    f2 r[16];
    f4 q[8];
    for (int i = 0; i < 16; i++) r[i] = f2{0.f, 0.f};
    for (int i = 0; i < 8; i++) q[i] = f4{0.f, 0.f, 0.f, 0.f};
    f2 acc = {0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
    for (int it = 0; it < ITER; it++) {
        if (KIND == 0) {   // 16 x ds_read_b64, 2 KB apart
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r[i]) : "v"(a8), "n"(i * 2048));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 16; i++) acc += r[i];
        } else if (KIND == 1) {   // 8 x ds_read2_b64 (offsets in 8-byte units): the same 16 elements
#pragma unroll
            for (int i = 0; i < 8; i++) { f4 v; asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(a8), "n"((2 * i) * 256 % 256), "n"((2 * i + 1) * 22 % 256)); q[i] = v; }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 8; i++) acc += f2{q[i].x + q[i].z, q[i].y + q[i].w};
        } else if (KIND == 2) {   // 16 x ds_write_b64
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(a8), "v"(r[i]), "n"(i * 2048) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (KIND == 3) {   // 8 x ds_write2_b64
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("ds_write2_b64 %0, %1, %2 offset0:%3 offset1:%4" :: "v"(a8), "v"(r[2 * i]), "v"(r[2 * i + 1]), "n"((2 * i) % 256), "n"((2 * i + 1) * 22 % 256) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (KIND == 4) {   // 8 x ds_read_b128
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[i]) : "v"(a16), "n"(i * 4096));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 8; i++) acc += f2{q[i].x + q[i].z, q[i].y + q[i].w};
        } else {                  // 8 x ds_write_b128
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(a16), "v"(q[i]), "n"(i * 4096) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + t] = acc.x + acc.y;
    if (t == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    float* out; unsigned long long* clk;
    CHECK(hipMalloc(&out, 1024 * 256 * sizeof(float)));
    CHECK(hipMalloc(&clk, 16));
    const char* names[] = {"16 x ds_read_b64", "8 x ds_read2_b64", "16 x ds_write_b64", "8 x ds_write2_b64", "8 x ds_read_b128", "8 x ds_write_b128"};
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int wps = 1; wps <= 4; wps++) {
        printf("-- %d wave(s) per SIMD (128 bytes per lane and iteration in every variant)\n", wps);
        auto run = [&](int kind, auto kern) {
            const int blocks = 256 * wps;
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 65536, 0, out, clk);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(a));
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 65536, 0, out, clk);
            CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
            float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
            unsigned long long h[2]; CHECK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
            const double ghz = (double)h[0] / ((double)h[1] * 10.0);
            // CU-level: 4 * wps waves each moving 64 lanes * 128 B per iteration
            const double bytes_per_cu = 4.0 * wps * 64 * 128.0 * ITER;
            printf("%-20s: %.3f ms, clock %.2f GHz, %.1f B/clk/CU\n", names[kind], ms, ghz, bytes_per_cu / (ms * 1e6 * ghz));
        };
        run(0, k<0>); run(1, k<1>); run(2, k<2>); run(3, k<3>); run(4, k<4>); run(5, k<5>);
    }
    return 0;
}
