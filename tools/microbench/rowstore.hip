// rowstore.hip -- what the memory system gives for the ROW kernel's store pattern without its compute:
// one workgroup of 192 threads per spectrum row walks 16 maps and writes its 4224 complex values per map into
// the tiled intermediate Y[map][w/16][row][16] (8 bytes per lane, 16 lanes = one 128-byte line, the lines of a
// row 270 KB apart), streaming stores, in the kernel's order (3 rounds x 8 stores, butterfly outputs 33 tiles
// apart).  Variants: (1) the same with 16 bytes per lane; (2) two adjacent rows per workgroup (256-byte
// pieces); (3) a row-major intermediate (33.8 KB contiguous per row).
// (4) / (5): an intermediate of tile width 4 (Y[map][w/4][row][4]) written by one row / two adjacent rows per workgroup.
// Standalone: hipcc --offload-arch=gfx950 -O3 rowstore.hip -o rowstore
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int M = 2112, ROWS = M + 2, NROW = M + 1, FW = 4224, TL = 16, NT = 192, WALK = 16;
constexpr size_t MAP_ELEMS = (size_t)(FW / TL) * ROWS * TL;   // c32 per map

// MODE 0: kernel pattern; 1: 16 B per lane; 2: two rows per workgroup; 3: row-major
template <int MODE>
__global__ void __launch_bounds__(NT) k_store(f2* __restrict__ Y, int maps) {
    const int row = blockIdx.x, map0 = blockIdx.y * WALK, t = threadIdx.x;
    f2 v = {(float)t, (float)row};
    for (int m = map0; m < map0 + WALK && m < maps; m++) {
        f2* yb = Y + (size_t)m * MAP_ELEMS;
        if (MODE == 0) {
            for (int r = 0; r < 3; r++) {
                const int j = t + NT * r;
                if (j < 528) {
                    f2* p = yb + ((size_t)(j >> 4) * ROWS + row) * TL + (j & 15);
#pragma unroll
                    for (int a = 0; a < 8; a++) __builtin_nontemporal_store(v, p + (size_t)a * 33 * ROWS * TL);
                }
            }
        } else if (MODE == 1) {   // lane writes columns 2q, 2q+1 (16 B): 8 lanes per line
            for (int r = 0; r < 2; r++) {
                const int q = t + NT * r;      // pair index within the 264-column butterfly block (264 pairs)
                if (q < 264) {
                    const int j = 2 * q;
                    f4* p = reinterpret_cast<f4*>(yb + ((size_t)(j >> 4) * ROWS + row) * TL + (j & 15));
                    f4 w = {v.x, v.y, v.x, v.y};
#pragma unroll
                    for (int a = 0; a < 8; a++) __builtin_nontemporal_store(w, p + (size_t)a * 33 * ROWS * TL / 2);
                }
            }
        } else if (MODE == 2) {   // workgroup owns rows 2*row, 2*row+1: 32 lanes = 256 contiguous bytes
            for (int r = 0; r < 6; r++) {
                const int u = t + NT * r;      // 0..1055: (j, which row)
                if (u < 1056) {
                    const int j = (u >> 5) * 16 + (u & 15), rr = (u >> 4) & 1;
                    const int rw = 2 * row + rr;
                    if (rw < NROW) {
                        f2* p = yb + ((size_t)(j >> 4) * ROWS + rw) * TL + (j & 15);
#pragma unroll
                        for (int a = 0; a < 8; a++) __builtin_nontemporal_store(v, p + (size_t)a * 33 * ROWS * TL);
                    }
                }
            }
        } else if (MODE == 4) {   // tile width 4 (Y[map][w/4][row][4]): lane writes columns 2q, 2q+1, two lanes = one 32-byte piece
            for (int r = 0; r < 2; r++) {
                const int q = t + NT * r;
                if (q < 264) {
                    const int j = 2 * q;
                    f4 w = {v.x, v.y, v.x, v.y};
#pragma unroll
                    for (int a = 0; a < 8; a++) {
                        const int jj = j + a * 528;
                        __builtin_nontemporal_store(w, reinterpret_cast<f4*>(yb + ((size_t)(jj >> 2) * ROWS + row) * 4 + (jj & 3)));
                    }
                }
            }
        } else if (MODE == 5) {   // tile width 4, workgroup owns rows 2*row, 2*row+1: four lanes = one 64-byte piece
            for (int r = 0; r < 3; r++) {
                const int u = t + NT * r;      // 0..527: (pair of columns q, which row)
                if (u < 528) {
                    const int q = (u >> 2) * 2 + (u & 1), rr = (u >> 1) & 1, rw = 2 * row + rr;
                    f4 w = {v.x, v.y, v.x, v.y};
                    if (rw < NROW) {
#pragma unroll
                        for (int a = 0; a < 8; a++) {
                            const int jj = 2 * q + a * 528;
                            __builtin_nontemporal_store(w, reinterpret_cast<f4*>(yb + ((size_t)(jj >> 2) * ROWS + rw) * 4 + (jj & 3)));
                        }
                    }
                }
            }
        } else {                  // row-major: row contiguous
            f2* p0 = yb + (size_t)row * FW;
            for (int r = 0; r < 3; r++) {
                const int j = t + NT * r;
                if (j < 528) {
#pragma unroll
                    for (int a = 0; a < 8; a++) __builtin_nontemporal_store(v, p0 + j + a * 528);
                }
            }
        }
    }
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int maps = 64;
    const size_t ybytes = (size_t)maps * MAP_ELEMS * 8;
    void* Y;
    CHECK(hipMalloc(&Y, ybytes));
    CHECK(hipMemset(Y, 0, ybytes));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    auto run = [&](const char* name, auto kern, dim3 grid) {
        for (int rep = 0; rep < 3; rep++) {
            for (int i = 0; i < 12; i++) hipLaunchKernelGGL(kern, grid, dim3(NT), 0, 0, (f2*)Y, maps);   // clocks settle
            CHECK(hipEventRecord(a));
            for (int i = 0; i < 6; i++) hipLaunchKernelGGL(kern, grid, dim3(NT), 0, 0, (f2*)Y, maps);
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, a, b));
            ms /= 6;
            printf("%-44s: %.3f ms  %.0f GB/s written  (%.2f us per map)\n", name, ms, (double)NROW * FW * 8 * maps / 1e6 / ms, ms * 1e3 / maps);
        }
    };
    run("kernel pattern (8 B/lane, 128-B lines)", k_store<0>, dim3(NROW, maps / WALK));
    run("16 B/lane", k_store<1>, dim3(NROW, maps / WALK));
    run("two adjacent rows per workgroup (256-B pieces)", k_store<2>, dim3((NROW + 1) / 2, maps / WALK));
    run("row-major intermediate (contiguous rows)", k_store<3>, dim3(NROW, maps / WALK));
    run("tile width 4, one row per workgroup (32-B pieces)", k_store<4>, dim3(NROW, maps / WALK));
    run("tile width 4, two adjacent rows per workgroup (64-B pieces)", k_store<5>, dim3((NROW + 1) / 2, maps / WALK));
    return 0;
}
