// valubench.hip -- issue rates of the instructions the row kernel is made of, VALU-only kernels:
// v_fma_f32, v_pk_fma_f32, v_pk_add_f32 (with op_sel), v_pk_mul_f32, v_mov_b32, and LDS b64 / b128 reads
// and writes, at 1..4 waves per SIMD.  Prints wave-instructions per nanosecond per CU and, with the
// measured shader clock (s_memtime / wall clock), cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int ITER = 4096, UNR = 16;

template <int KIND>
__global__ void __launch_bounds__(256) k_valu(float* out, unsigned long long* clk) {
    f2 a[UNR];
    const float s = (float)threadIdx.x * 1e-9f;
    for (int i = 0; i < UNR; i++) a[i] = f2{s + i, s - i};
    f2 b = {1.0000001f, 0.9999999f}, c = {1e-9f, -1e-9f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = wall_clock64();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < UNR; i++) {
            if (KIND == 0) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(c.x)); }
            if (KIND == 1) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); }
            if (KIND == 2) { asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "+v"(a[i]) : "v"(c)); }
            if (KIND == 3) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); }
            if (KIND == 4) { asm volatile("v_mov_b32 %0, %1" : "=v"(a[i].x) : "v"(a[(i + 1) % UNR].y)); }
            if (KIND == 5) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(c.x)); }
            if (KIND == 6) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i].x) : "v"(c.x)); }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = wall_clock64();
    f2 acc = {0, 0};
    for (int i = 0; i < UNR; i++) acc += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

// LDS: KIND 0 read b64, 1 write b64, 2 read b128, 3 write b128 (conflict-free: consecutive lanes consecutive addresses)
template <int KIND>
__global__ void __launch_bounds__(256) k_lds(float* out, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t = threadIdx.x;
    for (int i = t; i < 8192; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    f2 v2 = {1.f, 2.f};
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 v4 = {1.f, 2.f, 3.f, 4.f};
    f2 acc2 = {0, 0};
    f4 acc4 = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = wall_clock64();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < UNR; i++) {
            const int base = (i & 3) * 2048;
            if (KIND == 0) { acc2 += *reinterpret_cast<volatile f2*>(&lds[base + 2 * t]); }
            if (KIND == 1) { *reinterpret_cast<volatile f2*>(&lds[base + 2 * t]) = v2; }
            if (KIND == 2) { f4 x = *reinterpret_cast<volatile f4*>(&lds[(i & 1) * 4096 + 4 * t]); acc4 += x; }
            if (KIND == 3) { *reinterpret_cast<volatile f4*>(&lds[(i & 1) * 4096 + 4 * t]) = v4; }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = wall_clock64();
    out[blockIdx.x * blockDim.x + t] = acc2.x + acc2.y + acc4.x + acc4.w;
    if (t == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

int main() {
    float* out;
    unsigned long long* clk;
    CHECK(hipMalloc(&out, 256 * 8 * 256 * sizeof(float) * 4));
    CHECK(hipMalloc(&clk, 16));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_add_f32 op_sel", "v_pk_mul_f32", "v_mov_b32", "v_add_f32", "v_add_u32"};
    const char* lnames[] = {"ds_read_b64", "ds_write_b64", "ds_read_b128", "ds_write_b128"};
    for (int wps = 1; wps <= 4; wps++) {          // waves per SIMD = blocks of 256 threads (4 waves) per CU
        printf("-- %d wave(s) per SIMD\n", wps);
        auto run = [&](const char* name, auto kern, size_t lds) {
            const int blocks = 256 * wps;
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, out, clk);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(a));
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, out, clk);
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, a, b));
            unsigned long long h[2];
            CHECK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
            const double ghz = (double)h[0] / ((double)h[1] * 10.0);           // shader cycles per ns (wall clock = 100 MHz)
            const double winstr = (double)wps * ITER * UNR;                    // wave-instructions per SIMD
            printf("%-22s: %.3f ms kernel, clock %.2f GHz, %.2f cycles per wave-instruction per SIMD (in-kernel: %.2f)\n", name, ms, ghz,
                   ms * 1e6 * ghz / winstr, (double)h[0] / ((double)ITER * UNR * wps));
        };
        run(names[0], k_valu<0>, 0); run(names[1], k_valu<1>, 0); run(names[2], k_valu<2>, 0); run(names[3], k_valu<3>, 0);
        run(names[4], k_valu<4>, 0); run(names[5], k_valu<5>, 0); run(names[6], k_valu<6>, 0);
        run(lnames[0], k_lds<0>, 32768); run(lnames[1], k_lds<1>, 32768); run(lnames[2], k_lds<2>, 32768); run(lnames[3], k_lds<3>, 32768);
    }
    return 0;
}
