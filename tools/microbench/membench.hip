// membench.hip -- what the memory system gives for the output kernel's access pattern, without its
// compute: (a) float4 copy; (b) the kernel's pattern: gather of 64-byte pieces of the tiled
// intermediate (8 of the 16 columns of a 128-byte row, rows 128 bytes apart) + 8-byte-per-lane
// streaming stores of whole map columns; (c) the same with full 128-byte rows (16-column tiles);
// (d) (b) with 16-byte-per-lane stores.  Standalone: hipcc --offload-arch=gfx950 -O3 membench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int M = 2112, ROWS = M + 2, FW = 4224, FH = 4224, TL = 16;   // tiled Y: [FW/16][ROWS][16] c32 per map

__global__ void __launch_bounds__(256) k_copy(const f4* __restrict__ in, f4* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(in[i], &out[i]);
}

// One workgroup per (map, T-column tile) item, grid-stride persistent.  T = 8: 4 lanes per 64-byte piece;
// T = 16: 8 lanes per 128-byte row.  The loaded values are summed into the stored ones so nothing is elided;
// stores: each lane 8 bytes (ST16 = 0) or 16 bytes (ST16 = 1), consecutive lanes consecutive addresses in one column.
template <int T, int ST16, int NT>
__global__ void __launch_bounds__(NT) k_pattern(const f4* __restrict__ Y, float* __restrict__ out, int maps) {
    constexpr int LPR = T / 2;                    // lanes (16 B each) per row piece
    constexpr int RPI = NT / LPR;                 // rows per load instruction of the workgroup
    const int tiles_per_map = FW / T, ntiles = tiles_per_map * maps;
    const int nwg = gridDim.x, per_xcd = nwg / 8;
    const int wg_x = (nwg % 8 == 0) ? ((int)blockIdx.x % 8) * per_xcd + (int)blockIdx.x / 8 : (int)blockIdx.x;
    const int t = threadIdx.x;
    for (int tile = wg_x; tile < ntiles; tile += nwg) {
        const int map = tile / tiles_per_map, w0 = (tile - map * tiles_per_map) * T;
        const f4* Yt = Y + ((size_t)map * (FW / TL) * ROWS * TL + (size_t)(w0 / TL) * ROWS * TL + (w0 % TL)) / 2;   // f4 = 2 c32
        f4 acc = {0.f, 0.f, 0.f, 0.f};
        // gather: ROWS rows x T columns
        constexpr int NLD = (ROWS * LPR + NT - 1) / NT;
        f4 v[8];
        for (int r0 = 0; r0 < NLD; r0 += 8) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int e = t + NT * (r0 + u);
                const int row = e / LPR, p = e % LPR;
                v[u] = (row < ROWS) ? Yt[(size_t)row * (TL / 2) + p] : acc;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) acc += v[u];
        }
        // store: T columns x FH floats
        float* o = out + (size_t)map * FW * FH + (size_t)w0 * FH;
        if (ST16) {
            constexpr int PER = T * FH / 4;       // f4 per tile
            for (int e = t; e < PER; e += NT) {
                const int col = e / (FH / 4), q = e % (FH / 4);
                __builtin_nontemporal_store(acc, reinterpret_cast<f4*>(o + (size_t)col * FH) + q);
            }
        } else {
            constexpr int PER = T * FH / 2;
            f2 a2 = {acc.x + acc.z, acc.y + acc.w};
            for (int e = t; e < PER; e += NT) {
                const int col = e / (FH / 2), q = e % (FH / 2);
                __builtin_nontemporal_store(a2, reinterpret_cast<f2*>(o + (size_t)col * FH) + q);
            }
        }
    }
}

template <class F>
float time_ms(F&& f, int reps) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    f();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) f();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int maps = 32;
    const size_t ybytes = (size_t)maps * (FW / TL) * ROWS * TL * 8, obytes = (size_t)maps * FW * FH * 4;
    void *Y, *O;
    CHECK(hipMalloc(&Y, ybytes)); CHECK(hipMalloc(&O, obytes));
    CHECK(hipMemset(Y, 0, ybytes)); CHECK(hipMemset(O, 0, obytes));
    const double gb = (ybytes + obytes) / 1e9;
    printf("maps %d: Y %.2f GB + maps %.2f GB per pass\n", maps, ybytes / 1e9, obytes / 1e9);
    {
        const size_t n = (ybytes < obytes ? ybytes : obytes) / 16;   // both buffers hold at least n float4
        float ms = time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(256 * 16), dim3(256), 0, 0, (const f4*)Y, (f4*)O, n); }, 5);
        printf("float4 copy (nt stores)            : %.3f ms  %.0f GB/s\n", ms, 2.0 * n * 16 / 1e6 / ms);
    }
    auto run = [&](const char* name, auto kern, int nt, int wgs) {
        float ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(wgs), dim3(nt), 0, 0, (const f4*)Y, (float*)O, maps); }, 5);
        printf("%-35s: %.3f ms  %.0f GB/s  (%.2f us per map)\n", name, ms, gb * 1e3 / ms, ms * 1e3 / maps);
    };
    for (int wgs : {256, 512, 1024, 2048}) {
        printf("-- %d workgroups\n", wgs);
        run("T=8  64B pieces, 8B stores, 256thr", k_pattern<8, 0, 256>, 256, wgs);
        run("T=8  64B pieces, 16B stores, 256thr", k_pattern<8, 1, 256>, 256, wgs);
        run("T=16 128B rows, 8B stores, 256thr", k_pattern<16, 0, 256>, 256, wgs);
        run("T=16 128B rows, 16B stores, 256thr", k_pattern<16, 1, 256>, 256, wgs);
        run("T=4  32B pieces, 8B stores, 256thr", k_pattern<4, 0, 256>, 256, wgs);
    }
    // Infinity Cache (256 MB, memory side): does an intermediate that was JUST WRITTEN and is smaller than the
    // cache come back faster than one from HBM?  nm maps are written (streaming float4 stores, as the row
    // kernel does), then gathered + stored with the T = 8 pattern; only the second kernel is timed.
    printf("-- intermediate written right before it is read (Infinity Cache residency), T=8 pattern, 2048 x 256 threads\n");
    for (int nm : {1, 2, 3, 4, 8, 32}) {
        const size_t yb = (size_t)nm * (FW / TL) * ROWS * TL * 8;
        hipEvent_t a, b;
        CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
        float tot = 0;
        const int reps = 6;
        for (int r = 0; r < reps + 1; r++) {
            hipLaunchKernelGGL(k_copy, dim3(256 * 16), dim3(256), 0, 0, (const f4*)O, (f4*)Y, (yb < obytes ? yb : obytes) / 16);   // "row kernel": writes Y (never past either buffer)
            CHECK(hipEventRecord(a));
            hipLaunchKernelGGL((k_pattern<8, 0, 256>), dim3(2048), dim3(256), 0, 0, (const f4*)Y, (float*)O, nm);
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, a, b));
            if (r > 0) tot += ms;
        }
        const double gbm = (yb + (size_t)nm * FW * FH * 4) / 1e9;
        printf("%2d map(s) (%4.0f MB intermediate): %.3f ms  %.0f GB/s  (%.2f us per map)\n", nm, yb / 1e6, tot / reps, gbm * 1e3 / (tot / reps), tot / reps * 1e3 / nm);
    }
    printf("-- 256 workgroups of 768 threads (the kernel's shape)\n");
    run("T=8  64B pieces, 8B stores, 768thr", k_pattern<8, 0, 768>, 768, 256);
    run("T=8  64B pieces, 16B stores, 768thr", k_pattern<8, 1, 768>, 768, 256);
    return 0;
}
