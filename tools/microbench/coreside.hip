// coreside.hip -- can the two hot kernels of cfg3 share a CU?  (VERDICT r02 "next" item 3.)
//
// The row kernel (k_fast_rows_multi, 192 threads, 38 KB of LDS, vector-issue bound with idle HBM) and
// the output kernel (k_fast_cols, HBM bound with 68 % idle vector slots) run one after the other in the
// product.  Its 8-column / 768-thread output workgroup owns 148 KB of LDS, so nothing can sit beside it.
// This probe instantiates THE SAME kernel bodies (the library's headers) in a shape that can share a CU:
// output workgroups of 4 columns / 384 threads (77 KB, 6 waves; `colsN` = N of them, persistent) beside
// row workgroups (2 x 38 KB, 6 waves), on two streams, and times
//     rows alone | cols8 alone (the product's pair, run back to back)
//     cols4 alone (1 and 2 workgroups per CU)
//     cols4 (launched first, 1 per CU) || rows of the NEXT batch
// on buffers filled with finite random data (timing only; results are not checked here).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I../../cuda-fft-convolution_amd/csrc coreside.hip -o coreside
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "kernels_common.hpp"
#include "pipeline.hpp"

using namespace fc;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

using RowC = RowCfg<4224, 8, 24, 22, 192, 1>;
using Col8 = ColCfg<2112, 8, 12, 22, 8, 768>;
using Col4 = ColCfg<2112, 8, 12, 22, 4, 384>;

template <class Cfg, int NZ2, bool LINEAR>
__global__ void __launch_bounds__(Cfg::NT, 3) pk_rows(FastRowsArgs a, int rows, int kernels, int per_wg) {
    const int group = (int)blockIdx.x;
    const int kernel0 = (int)blockIdx.y * per_wg;
    const int nk = kernels - kernel0 < per_wg ? kernels - kernel0 : per_wg;
    DevPhaseCtx<RowMultiState<Cfg>> ctx;
    fast_rows_multi_body<Cfg, NZ2, LINEAR>(ctx, reinterpret_cast<c32*>(fc_smem), a, group, kernel0, nk, rows);
}

template <class Cfg>
__global__ void __launch_bounds__(Cfg::NT, 3) pk_cols(FastColsArgs a) {
    DevPhaseCtx<ColPairState<Cfg>> ctx;
    fast_cols_body<Cfg, true>(ctx, reinterpret_cast<c32*>(fc_smem), a, (int)blockIdx.x, (int)gridDim.x);
}

// which CU did each workgroup of a launch land on?  (census of the persistent cols4 grid)
__global__ void pk_census(int* cu_of_wg) {
    if (threadIdx.x == 0) {
        unsigned hwid = 0, xcc = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        cu_of_wg[blockIdx.x] = (int)(((xcc & 0xf) << 16) | (hwid & 0xffff));
    }
}

template <class T>
T* dev_upload(const std::vector<T>& v) {
    T* p = nullptr;
    CHECK(hipMalloc(reinterpret_cast<void**>(&p), v.size() * sizeof(T)));
    CHECK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return p;
}

__global__ void pk_fill(float* p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = (float)(x & 0xffff) * (1.0f / 65536.0f) - 0.5f;
    }
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int maps = argc > 1 ? atoi(argv[1]) : 64;
    const int reps = argc > 2 ? atoi(argv[2]) : 12;
    Geometry g;
    Tables t;
    if (!make_geometry(g, t, 4096, 4096, 1, 127, 127)) { printf("geometry failed\n"); return 1; }
    printf("transform %d x %d, M %d, rows %d, tiled %d, maps per launch %d\n", g.Lh, g.Lw, g.M, g.rows, (int)g.y_tiled(), maps);
    int num_cus = 256;
    {
        hipDeviceProp_t prop;
        CHECK(hipGetDeviceProperties(&prop, 0));
        num_cus = prop.multiProcessorCount;
    }
    DeviceTables d;
    d.fr_tw1 = dev_upload(t.fr.tw1);
    d.fr_tw2 = dev_upload(t.fr.tw2);
    d.fc_tw1 = dev_upload(t.fcl.tw1);
    d.fc_tw2 = dev_upload(t.fcl.tw2);
    d.fc_pairs = dev_upload(t.fcl.pairs);
    d.fc_rowoff = dev_upload(t.fcl.rowoff);
    d.fc_pair_row_of = dev_upload(t.fcl.pair_row_of);

    const int kw = 127;
    const size_t per_a = (size_t)g.rows * a_pitch_for(kw);
    const size_t ye = g.y_elems_per_kernel();
    c32 *A, *S, *Y0, *Y1;
    float* out;
    CHECK(hipMalloc(reinterpret_cast<void**>(&A), per_a * maps * sizeof(c32)));
    CHECK(hipMalloc(reinterpret_cast<void**>(&S), g.spectrum_elems() * sizeof(c32)));
    CHECK(hipMalloc(reinterpret_cast<void**>(&Y0), ye * maps * sizeof(c32)));
    CHECK(hipMalloc(reinterpret_cast<void**>(&Y1), ye * maps * sizeof(c32)));
    CHECK(hipMalloc(reinterpret_cast<void**>(&out), g.map_elems() * maps * sizeof(float)));
    hipLaunchKernelGGL(pk_fill, dim3(4096), dim3(256), 0, 0, (float*)A, per_a * maps * 2, 1u);
    hipLaunchKernelGGL(pk_fill, dim3(4096), dim3(256), 0, 0, (float*)S, g.spectrum_elems() * 2, 2u);
    hipLaunchKernelGGL(pk_fill, dim3(4096), dim3(256), 0, 0, (float*)Y0, ye * maps * 2, 3u);
    hipLaunchKernelGGL(pk_fill, dim3(4096), dim3(256), 0, 0, (float*)Y1, ye * maps * 2, 4u);
    CHECK(hipDeviceSynchronize());

    auto k_rows = pk_rows<RowC, 6, true>;
    auto k_c8 = pk_cols<Col8>;
    auto k_c4 = pk_cols<Col4>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_rows), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_c8), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_c4), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const size_t lds_rows = (size_t)RowC::LDS_ELEMS * 8, lds_c8 = (size_t)Col8::LDS_ELEMS * 8, lds_c4 = (size_t)Col4::LDS_ELEMS * 8;
    printf("LDS per workgroup: rows %zu, cols8 %zu, cols4 %zu bytes\n", lds_rows, lds_c8, lds_c4);

    hipStream_t sr, sc;
    CHECK(hipStreamCreateWithFlags(&sr, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));

    const int per_wg = g.rows_group_for(maps, num_cus);
    auto rows_launch = [&](c32* Y, hipStream_t s, size_t lds) {
        FastRowsArgs fa = fast_rows_args(g, d, A, kw, S, Y);
        const dim3 grid(g.rows, (maps + per_wg - 1) / per_wg);
        hipLaunchKernelGGL(k_rows, grid, dim3(RowC::NT), lds, s, fa, g.rows, maps, per_wg);
    };
    auto cols_launch = [&](auto kern, int T, int NT, size_t lds, const c32* Y, int wgs, hipStream_t s) {
        FastColsArgs fa = fast_cols_args(g, d, Y, out, g.map_elems(), maps);
        fa.tiles_per_kernel = g.fft_w / T;
        fa.ntiles = fa.tiles_per_kernel * maps;
        hipLaunchKernelGGL(kern, dim3(wgs), dim3(NT), lds, s, fa);
    };
    auto wall = [&](auto&& body) {   // mean host wall time per repetition, clocks warmed first
        for (int i = 0; i < 3; i++) body();
        CHECK(hipDeviceSynchronize());
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; i++) body();
        CHECK(hipDeviceSynchronize());
        const auto t1 = std::chrono::steady_clock::now();
        return std::chrono::duration<double, std::micro>(t1 - t0).count() / reps;
    };
    auto report = [&](const char* name, double us) { printf("%-62s %9.1f us  %6.2f us per map\n", name, us, us / maps); };

    // warm the clocks (80 ms+)
    for (int i = 0; i < 30; i++) rows_launch(Y1, sr, lds_rows);
    CHECK(hipDeviceSynchronize());

    if (argc > 3) {   // power mode: each kernel alone, back to back for ~5 s, while tools/power_by_kernel.sh samples rocm-smi
        auto soak = [&](const char* name, auto&& body, double seconds) {
            const auto t0 = std::chrono::steady_clock::now();
            long n = 0;
            while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
                for (int i = 0; i < 20; i++) body();
                CHECK(hipDeviceSynchronize());
                n += 20;
            }
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
            printf("%-40s %9.1f us per launch  %6.2f us per map  (%ld launches)\n", name, us, us / maps, n);
        };
        printf("PHASE rows\n");
        soak("rows alone, back to back", [&] { rows_launch(Y1, sr, lds_rows); }, 6.0);
        printf("PHASE cols8\n");
        soak("cols8 alone, back to back", [&] { cols_launch(k_c8, 8, 768, lds_c8, Y0, num_cus, sc); }, 6.0);
        printf("PHASE pair\n");
        soak("rows then cols8 (the product's step)", [&] { rows_launch(Y1, sr, lds_rows); CHECK(hipStreamSynchronize(sr)); cols_launch(k_c8, 8, 768, lds_c8, Y0, num_cus, sc); CHECK(hipStreamSynchronize(sc)); }, 6.0);
        return 0;
    }
    const double t_rows = wall([&] { rows_launch(Y1, sr, lds_rows); });
    report("rows alone (4 workgroups per CU)", t_rows);
    const double t_rows2 = wall([&] { rows_launch(Y1, sr, 78 * 1024); });
    report("rows alone, LDS padded to 78 KB (2 workgroups per CU)", t_rows2);
    const double t_c8 = wall([&] { cols_launch(k_c8, 8, 768, lds_c8, Y0, num_cus, sc); });
    report("cols8 alone (product: 768 threads, 1 per CU)", t_c8);
    const double t_serial = wall([&] { rows_launch(Y1, sr, lds_rows); CHECK(hipStreamSynchronize(sr)); cols_launch(k_c8, 8, 768, lds_c8, Y0, num_cus, sc); CHECK(hipStreamSynchronize(sc)); });
    report("rows then cols8, one after the other (status quo)", t_serial);
    const double t_c4_1 = wall([&] { cols_launch(k_c4, 4, 384, lds_c4, Y0, num_cus, sc); });
    report("cols4 alone, 1 workgroup per CU (256)", t_c4_1);
    const double t_c4_2 = wall([&] { cols_launch(k_c4, 4, 384, lds_c4, Y0, 2 * num_cus, sc); });
    report("cols4 alone, 2 workgroups per CU (512)", t_c4_2);

    // census: where do 256 cols4-shaped workgroups land on an idle chip?
    {
        int* cu_dev = nullptr;
        CHECK(hipMalloc(reinterpret_cast<void**>(&cu_dev), 4096 * sizeof(int)));
        auto k_cen = pk_census;
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_cen), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL(k_cen, dim3(num_cus), dim3(384), lds_c4, sc, cu_dev);
        CHECK(hipDeviceSynchronize());
        std::vector<int> cu(num_cus);
        CHECK(hipMemcpy(cu.data(), cu_dev, num_cus * sizeof(int), hipMemcpyDeviceToHost));
        std::vector<int> sorted = cu;
        std::sort(sorted.begin(), sorted.end());
        int distinct = 0;
        for (size_t i = 0; i < sorted.size(); i++)
            if (i == 0 || sorted[i] != sorted[i - 1]) distinct++;
        printf("census: %d cols4-shaped workgroups (trivial kernel) landed on %d distinct (xcc, hw_id[15:0]) slots\n", num_cus, distinct);
    }

    // the pair that could share CUs: cols4 first (resident: 1 per CU), rows of the next batch right after
    hipEvent_t ec, er;
    CHECK(hipEventCreateWithFlags(&ec, hipEventDisableTiming));
    CHECK(hipEventCreateWithFlags(&er, hipEventDisableTiming));
    auto pair = [&](int cols_wgs, bool cols_first, size_t rows_lds) {
        return wall([&] {
            if (cols_first) { cols_launch(k_c4, 4, 384, lds_c4, Y0, cols_wgs, sc); rows_launch(Y1, sr, rows_lds); }
            else            { rows_launch(Y1, sr, rows_lds); cols_launch(k_c4, 4, 384, lds_c4, Y0, cols_wgs, sc); }
            CHECK(hipEventRecord(ec, sc));
            CHECK(hipEventRecord(er, sr));
            CHECK(hipStreamWaitEvent(sc, er, 0));   // next repetition: both streams start together again
            CHECK(hipStreamWaitEvent(sr, ec, 0));
        });
    };
    const double t_p1 = pair(num_cus, true, lds_rows);
    report("cols4 (256, launched first) || rows", t_p1);
    const double t_p2 = pair(num_cus, false, lds_rows);
    report("rows (launched first) || cols4 (256)", t_p2);
    const double t_p3 = pair(2 * num_cus, true, lds_rows);
    report("cols4 (512, launched first) || rows", t_p3);
    const double t_p4 = pair(num_cus, true, 52 * 1024);
    report("cols4 (256, first) || rows padded to 52 KB (at most 1 beside cols4)", t_p4);
    // sanity: the product's 8-column kernel cannot share a CU
    const double t_p8 = wall([&] {
        cols_launch(k_c8, 8, 768, lds_c8, Y0, num_cus, sc);
        rows_launch(Y1, sr, lds_rows);
        CHECK(hipEventRecord(ec, sc));
        CHECK(hipEventRecord(er, sr));
        CHECK(hipStreamWaitEvent(sc, er, 0));
        CHECK(hipStreamWaitEvent(sr, ec, 0));
    });
    report("cols8 (launched first) || rows (cannot share a CU)", t_p8);
    printf("\nsummary per map: status quo %.2f us; best shared pair %.2f us; sum of the parts rows + cols4(256) %.2f us\n",
           t_serial / maps, std::min(std::min(t_p1, t_p2), std::min(t_p3, t_p4)) / maps, (t_rows + t_c4_1) / maps);
    return 0;
}
