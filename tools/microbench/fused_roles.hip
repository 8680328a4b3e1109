// fused_roles.hip -- second co-residency probe (VERDICT r02 "next" item 3), after coreside.hip showed that
// two kernels on two streams do not share CUs usefully (the dispatcher races them, fat workgroups starve
// behind thin ones and persistent grids with late starters end up one after the other).
//
// ONE launch, 2 persistent workgroups of 384 threads per CU (LDS admits exactly two), each taking a role at
// run time: the first workgroup to arrive on a CU (claim by physical CU id) transforms output columns
// (fast_cols_body, 4-column tiles, of batch b), the second walks spectral rows (fast_rows_multi_body, two rows
// per workgroup, of batch b + 1).  Same kernel bodies as the product (the library's headers), garbage-in /
// garbage-out data: timing only.  Modes isolate the parts:
//   both      1 C + 1 R per CU               C only / R only: the same launch with the other role exiting at once
//   all C / all R: both workgroups of every CU in one role
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I../../cuda-fft-convolution_amd/csrc fused_roles.hip -o fused_roles
// (ablations: add -DFC_INSTRUMENT -DFC_ROWSM_DBG=2 -DFC_COLS_DBG=12 for the compute-only variant "fused_roles_nomem")
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "kernels_common.hpp"
#include "pipeline.hpp"

using namespace fc;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

#ifndef ROLE_ATTR
#define ROLE_ATTR __forceinline__   // -DROLE_ATTR="__attribute__((noinline))": each role a function of its own
#endif
using Row1 = RowCfg<4224, 8, 24, 22, 192, 1>;
using Row2 = RowCfg<4224, 8, 24, 22, 384, 2>;
using Col8 = ColCfg<2112, 8, 12, 22, 8, 768>;
using Col4 = ColCfg<2112, 8, 12, 22, 4, 384>;

struct FusedArgs {
    FastRowsArgs r;
    int rows, kernels, per_wg;
    FastColsArgs c;
    unsigned* ctl;   // [4096 + 2 * key + arrival] waves per SIMD (4 bits each); [0] row work counter, [1] number of R workgroups, [8 + xcc] C workgroups per XCD, [64 + cu key] claims
    int mode;        // 0 both, 1 C only (R exits), 2 R only (C exits), 3 all C, 4 all R
    int n_c;         // C workgroups expected (tile stride)
};

// each role is a function of its own (not inlined): the register allocation of one body does not disturb the
// other's -- inlined into one kernel the pair came out with 16 spilled VGPRs, and a scratch access inside the
// output-column loop has to be waited for together with the prefetch gather issued before it (one in-order counter)
template <class CC>
__device__ ROLE_ATTR void role_cols(const FusedArgs* a, int idx) {
    DevPhaseCtx<ColPairState<CC>> ctx;
    fast_cols_body<CC, true>(ctx, reinterpret_cast<c32*>(fc_smem), a->c, idx, a->n_c);
}

template <class RC, int NZ2>
__device__ ROLE_ATTR void role_rows(const FusedArgs* a, int* sh) {
    c32* lds = reinterpret_cast<c32*>(fc_smem);
    const int groups = (a->rows + RC::RPW - 1) / RC::RPW;
    const int walks = (a->kernels + a->per_wg - 1) / a->per_wg;
    DevPhaseCtx<RowMultiState<RC>> ctx;
    for (;;) {
        if (threadIdx.x == 0) sh[2] = (int)atomicAdd(&a->ctl[0], 1u);
        __syncthreads();
        const int item = sh[2];
        __syncthreads();
        if (item >= groups * walks) break;
        const int walk = item / groups, group = item - walk * groups;
        const int kernel0 = walk * a->per_wg;
        const int nk = a->kernels - kernel0 < a->per_wg ? a->kernels - kernel0 : a->per_wg;
        fast_rows_multi_body<RC, NZ2, true>(ctx, lds, a->r, group, kernel0, nk, a->rows);
    }
}

template <class RC, int NZ2, class CC>
__global__ void __launch_bounds__(384, 3) pk_fused(const FusedArgs* __restrict__ ap) {
    c32* lds = reinterpret_cast<c32*>(fc_smem);
    constexpr int TOP = (RC::LDS_ELEMS > CC::LDS_ELEMS ? RC::LDS_ELEMS : CC::LDS_ELEMS);
    int* sh = reinterpret_cast<int*>(lds + TOP);   // [0] role, [1] index, [2] work item
    if (threadIdx.x == 0) {
        const int mode = ap->mode;
        unsigned* ctl = ap->ctl;
        unsigned hwid = 0, xcc = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7;
        const unsigned key = (xcc << 8) | ((hwid >> 8) & 0xff);   // XCD, shader engine / array, CU
        const unsigned slot = atomicAdd(&ctl[64 + key], 1u);
        int role = (mode == 3 || mode == 5) ? 0 : (mode == 4) ? 1 : (mode == 6 || mode == 7) ? (int)((slot & 1) ^ 1) : (int)(slot & 1);   // 0 = C, 1 = R
        int idx;
        if (role == 0) idx = (int)atomicAdd(&ctl[8 + xcc], 1u) * 8 + (int)xcc;   // fast_cols_body: wg % 8 = XCD, wg / 8 = slot in it
        else idx = (int)atomicAdd(&ctl[1], 1u);
        if (((mode == 1 || mode == 6) && role == 1) || (mode == 2 && role == 0)) role = 2;
        sh[0] = role;
        sh[1] = idx;
        sh[3] = (int)(key * 2 + (slot & 1));
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {   // which SIMD did each wave of this workgroup land on?
        unsigned hw = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        atomicAdd(&ap->ctl[4096 + sh[3]], 1u << (4 * ((hw >> 4) & 3)));
    }
    const int role = sh[0], idx = sh[1];
    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(ap->ctl + 8192);
    unsigned long long t0 = 0, r0 = 0;
    if (threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    if (role == 0) role_cols<CC>(ap, idx);
    else if (role == 1) role_rows<RC, NZ2>(ap, sh);
    if (threadIdx.x == 0 && role != 2) {   // sum over the workgroups of (shader cycles, 100 MHz ticks) spent in the role
        atomicAdd(&stamps[0], (unsigned long long)__builtin_amdgcn_s_memtime() - t0);
        atomicAdd(&stamps[1], (unsigned long long)__builtin_amdgcn_s_memrealtime() - r0);
    }
}


// ---------------------------------------------------------------------------------------------------------
// Wave-specialised form: ONE workgroup of 768 threads per CU (12 waves = 3 per SIMD: always resident together --
// two 384-thread workgroups at this register budget are NOT admitted side by side, the counters of this probe show
// 5.8 waves per CU for them).  Waves 0-5 run the output-column body (4-column tiles), waves 6-11 the spectral-row
// body (2 rows at a time); every phase boundary of either half is the whole workgroup's s_barrier, so the halves
// advance in lockstep, and whichever half runs out of work first keeps arriving at barriers until the other is done.
template <class State>
struct WsPhaseCtx {
    State st;
    int toff;
    template <class F>
    __device__ __forceinline__ void phase(F&& f) {
        f((int)threadIdx.x - toff, st);
        __syncthreads();
    }
    template <class F>
    __device__ __forceinline__ void phase_nosync(F&& f) {
        f((int)threadIdx.x - toff, st);
    }
    template <bool NOSYNC, class F>
    __device__ __forceinline__ void phase_dbg(F&& f) {
        f((int)threadIdx.x - toff, st);
        if (!NOSYNC) __syncthreads();
    }
};

template <class RC, int NZ2, class CC>
__global__ void __launch_bounds__(768, 3) pk_ws(const FusedArgs* __restrict__ ap) {
    c32* lds = reinterpret_cast<c32*>(fc_smem);
    volatile int* done = reinterpret_cast<volatile int*>(lds + CC::LDS_ELEMS + RC::LDS_ELEMS);
    const int role = __builtin_amdgcn_readfirstlane((int)threadIdx.x) >= 384 ? 1 : 0;   // wave-uniform: a scalar branch
    if (threadIdx.x < 2) done[threadIdx.x] = 0;
    __syncthreads();
    const int wg = (int)blockIdx.x, nwg = (int)gridDim.x;
    const int mode = ap->mode;   // 10 both halves work, 11 only the column half, 12 only the row half
    if (role == 0) {
        if (mode != 12) {
            WsPhaseCtx<ColPairState<CC>> ctx;
            ctx.toff = 0;
            fast_cols_body<CC, true>(ctx, lds, ap->c, wg, nwg);
        }
    } else {
        if (mode != 11) {
            const int groups = (ap->rows + RC::RPW - 1) / RC::RPW;
            const int walks = (ap->kernels + ap->per_wg - 1) / ap->per_wg;
            WsPhaseCtx<RowMultiState<RC>> ctx;
            ctx.toff = 384;
            for (int item = wg; item < groups * walks; item += nwg) {
                const int walk = item / groups, group = item - walk * groups;
                const int kernel0 = walk * ap->per_wg;
                const int nk = ap->kernels - kernel0 < ap->per_wg ? ap->kernels - kernel0 : ap->per_wg;
                fast_rows_multi_body<RC, NZ2, true>(ctx, lds + CC::LDS_ELEMS, ap->r, group, kernel0, nk, ap->rows);
            }
        }
    }
    // this half is done: keep pairing the other half's barriers until it is done too (both leave after the same barrier)
    if ((int)threadIdx.x == role * 384) done[role] = 1;
    for (;;) {
        __syncthreads();
        if (done[1 - role]) break;
    }
}

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
    const int walk = argc > 3 ? atoi(argv[3]) : 16;
    Geometry g;
    Tables t;
    if (!make_geometry(g, t, 4096, 4096, 1, 127, 127)) { printf("geometry failed\n"); return 1; }
    int num_cus = 256;
    {
        hipDeviceProp_t prop;
        CHECK(hipGetDeviceProperties(&prop, 0));
        num_cus = prop.multiProcessorCount;
    }
    printf("transform %d x %d, maps per launch %d, walk %d maps, %d CUs\n", g.Lh, g.Lw, maps, walk, num_cus);
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
    unsigned* ctl;
    CHECK(hipMalloc(reinterpret_cast<void**>(&A), per_a * maps * sizeof(c32)));
    CHECK(hipMalloc(reinterpret_cast<void**>(&S), g.spectrum_elems() * sizeof(c32)));
    CHECK(hipMalloc(reinterpret_cast<void**>(&Y0), ye * maps * sizeof(c32)));
    CHECK(hipMalloc(reinterpret_cast<void**>(&Y1), ye * maps * sizeof(c32)));
    CHECK(hipMalloc(reinterpret_cast<void**>(&out), g.map_elems() * maps * sizeof(float)));
    CHECK(hipMalloc(reinterpret_cast<void**>(&ctl), (8192 + 16) * sizeof(unsigned)));
    hipLaunchKernelGGL(pk_fill, dim3(4096), dim3(256), 0, 0, (float*)A, per_a * maps * 2, 1u);
    hipLaunchKernelGGL(pk_fill, dim3(4096), dim3(256), 0, 0, (float*)S, g.spectrum_elems() * 2, 2u);
    hipLaunchKernelGGL(pk_fill, dim3(4096), dim3(256), 0, 0, (float*)Y0, ye * maps * 2, 3u);
    hipLaunchKernelGGL(pk_fill, dim3(4096), dim3(256), 0, 0, (float*)Y1, ye * maps * 2, 4u);
    CHECK(hipDeviceSynchronize());

    auto k_rows1 = pk_rows<Row1, 6, true>;
    auto k_rows2 = pk_rows<Row2, 6, true>;
    auto k_c8 = pk_cols<Col8>;
    auto k_fused = pk_fused<Row2, 6, Col4>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_rows1), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_rows2), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_c8), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fused), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const size_t lds_r1 = (size_t)Row1::LDS_ELEMS * 8, lds_r2 = (size_t)Row2::LDS_ELEMS * 8, lds_c8 = (size_t)Col8::LDS_ELEMS * 8;
    const size_t lds_f = (size_t)std::max(Row2::LDS_ELEMS, Col4::LDS_ELEMS) * 8 + 16;
    printf("LDS per workgroup: rows1 %zu, rows2 %zu, cols8 %zu, cols4 %zu, fused %zu bytes\n", lds_r1, lds_r2, lds_c8, (size_t)Col4::LDS_ELEMS * 8, lds_f);

    hipStream_t s;
    CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    FusedArgs* fargs_dev = nullptr;
    CHECK(hipMalloc(reinterpret_cast<void**>(&fargs_dev), 16 * sizeof(FusedArgs)));
    auto rows_args = [&](c32* Y) { return fast_rows_args(g, d, A, kw, S, Y); };
    auto cols_args = [&](const c32* Y, int T) {
        FastColsArgs fa = fast_cols_args(g, d, Y, out, g.map_elems(), maps);
        fa.tiles_per_kernel = g.fft_w / T;
        fa.ntiles = fa.tiles_per_kernel * maps;
        return fa;
    };
    auto fused = [&](int mode) {
        FusedArgs fa;
        fa.r = rows_args(Y1);
        fa.rows = g.rows; fa.kernels = maps; fa.per_wg = walk;
        fa.c = cols_args(Y0, 4);
        fa.ctl = ctl;
        fa.mode = mode;
        fa.n_c = (mode == 3) ? 2 * num_cus : num_cus;
        const int grid = (mode == 5) ? num_cus : 2 * num_cus;
        CHECK(hipMemsetAsync(ctl, 0, (8192 + 16) * sizeof(unsigned), s));
        { const unsigned long long init[4] = {0ull, 0ull, 0ull, 0ull}; CHECK(hipMemcpyAsync(ctl + 8192, init, sizeof(init), hipMemcpyHostToDevice, s)); }
        CHECK(hipMemcpyAsync(fargs_dev + mode, &fa, sizeof(fa), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_fused, dim3(grid), dim3(384), lds_f, s, (const FusedArgs*)(fargs_dev + mode));
    };
    auto k_ws = pk_ws<Row2, 6, Col4>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ws), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const size_t lds_ws = (size_t)(Row2::LDS_ELEMS + Col4::LDS_ELEMS) * 8 + 16;
    auto ws = [&](int mode) {
        FusedArgs fa;
        fa.r = rows_args(Y1);
        fa.rows = g.rows; fa.kernels = maps; fa.per_wg = walk;
        fa.c = cols_args(Y0, 4);
        fa.ctl = ctl;
        fa.mode = mode;
        fa.n_c = num_cus;
        CHECK(hipMemcpyAsync(fargs_dev + (mode - 10) + 5, &fa, sizeof(fa), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_ws, dim3(num_cus), dim3(768), lds_ws, s, (const FusedArgs*)(fargs_dev + (mode - 10) + 5));
    };
    auto wall = [&](auto&& body) {
        for (int i = 0; i < 3; i++) body();
        CHECK(hipDeviceSynchronize());
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; i++) body();
        CHECK(hipDeviceSynchronize());
        const auto t1 = std::chrono::steady_clock::now();
        return std::chrono::duration<double, std::micro>(t1 - t0).count() / reps;
    };
    auto report = [&](const char* name, double us) {
        unsigned long long st[4];
        CHECK(hipMemcpy(st, ctl + 8192, sizeof(st), hipMemcpyDeviceToHost));
        const double mhz = st[1] ? (double)st[0] / (double)st[1] * 100.0 : 0.0;
        printf("%-66s %9.1f us  %6.2f us per map   [shader clock of the last fused launch %.0f MHz]\n", name, us, us / maps, mhz);
    };
    const int per_wg = g.rows_group_for(maps, num_cus);
    auto rows1 = [&] { hipLaunchKernelGGL(k_rows1, dim3(g.rows, (maps + per_wg - 1) / per_wg), dim3(192), lds_r1, s, rows_args(Y1), g.rows, maps, per_wg); };
    auto rows2 = [&] { hipLaunchKernelGGL(k_rows2, dim3((g.rows + 1) / 2, (maps + per_wg - 1) / per_wg), dim3(384), lds_r2, s, rows_args(Y1), g.rows, maps, per_wg); };
    auto cols8 = [&] { hipLaunchKernelGGL(k_c8, dim3(num_cus), dim3(768), lds_c8, s, cols_args(Y0, 8)); };

    if (argc > 4 && argv[4][0] == 'w') {   // wave-specialised kernel only (first GPU contact of a new barrier protocol: keep it short)
        printf("wave-specialised kernel, LDS %zu bytes\n", lds_ws);
        for (int mode : {11, 12, 10}) {
            ws(mode);
            CHECK(hipDeviceSynchronize());
            printf("mode %d ran once\n", mode);
        }
        const double a = wall([&] { ws(11); });
        const double b = wall([&] { ws(12); });
        const double c = wall([&] { ws(10); });
        printf("ws columns half only %.1f us (%.2f per map) | rows half only %.1f us (%.2f) | both halves %.1f us (%.2f per map)\n", a, a / maps, b, b / maps, c, c / maps);
        return 0;
    }
    if (argc > 4) {   // counter runs (rocprofv3 --pmc): two dispatches of each fused mode, in this order: 3, 4, 5, 2, 7
        for (int mode : {3, 4, 5, 2, 7})
            for (int i = 0; i < 2; i++) { fused(mode); CHECK(hipDeviceSynchronize()); }
        rows1(); rows2(); cols8();
        CHECK(hipDeviceSynchronize());
        return 0;
    }
    for (int i = 0; i < 30; i++) rows1();
    CHECK(hipDeviceSynchronize());
    const double t_r1 = wall(rows1);
    report("product rows (192 threads, 4 per CU)", t_r1);
    const double t_r2 = wall(rows2);
    report("rows, 2 rows per workgroup (384 threads, 2 per CU), plain launch", t_r2);
    const double t_c8 = wall(cols8);
    report("product cols8 (768 threads, 1 per CU)", t_c8);
    const double t_sq = wall([&] { rows1(); cols8(); });
    report("product pair back to back (status quo)", t_sq);
    const double t_f3 = wall([&] { fused(3); });
    report("fused launch, all C   (2 x cols4 per CU)", t_f3);
    const double t_f4 = wall([&] { fused(4); });
    report("fused launch, all R   (2 x rows2 per CU, work queue)", t_f4);
    const double t_f1 = wall([&] { fused(1); });
    report("fused launch, C only  (1 x cols4 per CU, other workgroup exits)", t_f1);
    const double t_f2 = wall([&] { fused(2); });
    report("fused launch, R only  (1 x rows2 per CU, other workgroup exits)", t_f2);
    auto census = [&](const char* what) {
        CHECK(hipDeviceSynchronize());
        std::vector<unsigned> h(8192);
        CHECK(hipMemcpy(h.data(), ctl, 8192 * sizeof(unsigned), hipMemcpyDeviceToHost));
        int hist[4] = {0, 0, 0, 0};
        for (int i = 64; i < 64 + 2048; i++) if (h[i]) hist[h[i] > 3 ? 3 : h[i]]++;
        printf("   census (%s): CUs with 1 / 2 / more workgroups: %d / %d / %d; C per XCD", what, hist[1], hist[2], hist[3]);
        for (int x = 0; x < 8; x++) printf(" %u", h[8 + x]);
        printf("\n");
        for (int sl = 0; sl < 2; sl++) {   // waves per SIMD of the first / second arrival, as sorted patterns
            int pat[8] = {0};   // 2211, 3111, 2220, 3210, 3300, other
            for (int k = 0; k < 2048; k++) {
                unsigned v = h[4096 + 2 * k + sl];
                if (!v) continue;
                int c[4] = {(int)(v & 15), (int)((v >> 4) & 15), (int)((v >> 8) & 15), (int)((v >> 12) & 15)};
                std::sort(c, c + 4);
                const int code = c[3] * 1000 + c[2] * 100 + c[1] * 10 + c[0];
                pat[code == 2211 ? 0 : code == 3111 ? 1 : code == 2220 ? 2 : code == 3210 ? 3 : code == 3300 ? 4 : 5]++;
            }
            printf("   waves per SIMD, %s arrival: 2-2-1-1 x%d, 3-1-1-1 x%d, 2-2-2-0 x%d, 3-2-1-0 x%d, 3-3-0-0 x%d, other x%d\n", sl ? "second" : "first",
                   pat[0], pat[1], pat[2], pat[3], pat[4], pat[5]);
        }
    };
    const double t_f5 = wall([&] { fused(5); });
    report("fused kernel, grid of 256, all C (placement left to the dispatcher)", t_f5);
    census("grid 256, all C");
    const double t_f6 = wall([&] { fused(6); });
    report("fused launch, C only, C = SECOND arrival on the CU", t_f6);
    census("C = second arrival");
    const double t_f1b = wall([&] { fused(1); });
    report("fused launch, C only, C = FIRST arrival (again)", t_f1b);
    census("C = first arrival");
    const double t_f7 = wall([&] { fused(7); });
    report("fused launch, C + R, C = second arrival, R = first", t_f7);
    census("C second + R first");
    const double t_f0 = wall([&] { fused(0); });
    report("fused launch, C + R   (1 x cols4 beside 1 x rows2 on every CU)", t_f0);
    {
        std::vector<unsigned> h(8192);
        CHECK(hipMemcpy(h.data(), ctl, 8192 * sizeof(unsigned), hipMemcpyDeviceToHost));
        int cus = 0, two = 0;
        for (int i = 64; i < 4096; i++) { if (h[i]) cus++; if (h[i] == 2) two++; }
        printf("claims of the last launch: %d CUs seen, %d with exactly two workgroups; R workgroups %u; C per XCD", cus, two, h[1]);
        for (int x = 0; x < 8; x++) printf(" %u", h[8 + x]);
        printf("\n");
    }
    printf("\nsummary per map: status quo %.2f | C only %.2f, R only %.2f -> max %.2f, sum %.2f | C + R together %.2f\n", t_sq / maps,
           t_f1 / maps, t_f2 / maps, std::max(t_f1, t_f2) / maps, (t_f1 + t_f2) / maps, t_f0 / maps);
    return 0;
}
