// kernels_body.hpp -- the workgroup bodies of the four generic kernels of the engine.
//
// Data layouts (all device-resident; h = MATLAB row index, contiguous in the reference's
// H x W x F column-major arrays, src/cudaConvFFTData.cuh:26-27):
//   real planes      [f][w][h]                 as the caller hands them over
//   column spectrum  [f][i][x]   i in [0,M]    M = Lh/2; row i holds h-frequency k with
//                                              i = pos_M(k) (digit-reversed), i = M: Nyquist
//   image spectrum S [f][i][p]   p in [0,Lw)   w-frequency in digit-reversed order pos_Lw(.)
//                                              pre-scaled by 1/(Lh*Lw)
//                                              (src/cudaConvolutionFFT.cu:270 folded in)
//   intermediate  Y  [n][i][w]                 after the inverse w-transform, natural w
//   maps        out  [n][w][h]   FFT_H x FFT_W column-major windows
//                                              (src/cudaConvolutionFFT.cu:198-200)
// Rows i of the spectra are never reordered: every row is processed independently by the
// w-passes and the h-passes consume exactly the order they produce.
//
// Each body is a template over a context {tid, nthreads, sync()}; kernels*.hip instantiate it
// with the HIP thread/barrier, tests/emu with a sequential host context.
#pragma once
#include "fc_common.hpp"
#include "fft_lds.hpp"

namespace fc {

// ---------------------------------------------------------------------------------------
// cols_r2c: real -> half-complex transform along h of T columns per workgroup.
// Replaces padData + the H half of cufftExecR2C (src/cudaConvFFTData.cuh:11-31,
// src/cudaConvolutionFFT.cu:155-167, :245-255) for the image and for the kernels.
// ---------------------------------------------------------------------------------------
struct ColsR2CArgs {
    const float* in;      // plane q at in + q*in_plane_stride; column c at + c*in_col_pitch, h contiguous
    size_t in_plane_stride;
    int in_col_pitch;     // floats between columns (= data height)
    int h_in;             // valid input samples per column (rest is zero padding)
    int ncols;            // columns in this plane
    c32* out;             // plane q at out + q*out_plane_stride; rows i in [0,M]: [i*out_pitch + c]
    size_t out_plane_stride;
    int out_pitch;        // c32 between rows
    int M;                // complex transform length (Lh/2)
    int T;                // columns per workgroup
    int lds_pitch;        // c32 per column in LDS (>= M+1)
    FftDesc fd;           // M-point transform
    const c32* tw;
    const PairEntry* pairs;
    int npairs;
};

template <class Ctx>
FC_HD void cols_r2c_body(const Ctx& ctx, c32* lds, const ColsR2CArgs& a, int tile, int plane) {
    const int M = a.M, T = a.T, LP = a.lds_pitch;
    const int c0 = tile * T;
    const float* in = a.in + (size_t)plane * a.in_plane_stride;
    c32* out = a.out + (size_t)plane * a.out_plane_stride;
    // load: z[n] = x[2n] + i x[2n+1], zero beyond h_in (this IS the zero padding)
    for (int idx = ctx.tid; idx < T * M; idx += ctx.nthreads) {
        int t = idx / M, n = idx - t * M;
        int c = c0 + t;
        float x0 = 0.f, x1 = 0.f;
        if (c < a.ncols) {
            const float* col = in + (size_t)c * a.in_col_pitch;
            if (2 * n < a.h_in) x0 = col[2 * n];
            if (2 * n + 1 < a.h_in) x1 = col[2 * n + 1];
        }
        lds[t * LP + n] = mk(x0, x1);
    }
    ctx.sync();
    fft_forward(ctx, lds, LP, T, a.fd, a.tw);
    // split the packed transform into the spectrum of the real sequence, in place
    for (int idx = ctx.tid; idx < T * a.npairs; idx += ctx.nthreads) {
        int t = idx / a.npairs, p = idx - t * a.npairs;
        c32* z = lds + t * LP;
        PairEntry e = a.pairs[p];
        if (p == 0) {
            c32 z0 = z[e.a];
            z[e.a] = mk(z0.x + z0.y, 0.f);  // DC
            z[e.b] = mk(z0.x - z0.y, 0.f);  // Nyquist -> extra slot M
        } else if (e.a == e.b) {
            z[e.a] = conj(z[e.a]);
        } else {
            c32 zk = z[e.a], zm = z[e.b];
            c32 E = mk(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
            c32 D = mk(0.5f * (zk.x - zm.x), 0.5f * (zk.y + zm.y));
            c32 G = cmul(e.w, D);
            z[e.a] = mk(E.x + G.y, E.y - G.x);
            z[e.b] = mk(E.x - G.y, -E.y - G.x);
        }
    }
    ctx.sync();
    // store rows, t fastest (T*8-byte segments)
    for (int idx = ctx.tid; idx < (M + 1) * T; idx += ctx.nthreads) {
        int i = idx / T, t = idx - i * T;
        int c = c0 + t;
        if (c < a.ncols) out[(size_t)i * a.out_pitch + c] = lds[t * LP + i];
    }
}

// ---------------------------------------------------------------------------------------
// rows_fwd: forward complex transform along w of one spectrum row, in place in global
// memory, with the 1/(Lh*Lw) normalisation folded in (image only, once per image).
// The W half of cufftExecR2C (src/cudaConvolutionFFT.cu:167).
// ---------------------------------------------------------------------------------------
struct RowsFwdArgs {
    c32* S;          // row r at S + r*pitch
    int pitch;
    int nvalid;      // columns < nvalid hold data, the rest of the row is treated as zero
    float scale;
    FftDesc fd;      // Lw-point transform
    const c32* tw;
    const int* out_map;  // optional: element x of the stored row is transform position out_map[x]
                         // (the fast row kernel's register order, fast_rows.hpp); nullptr = identity
};

template <class Ctx>
FC_HD void rows_fwd_body(const Ctx& ctx, c32* lds, const RowsFwdArgs& a, int row) {
    const int L = a.fd.L;
    c32* g = a.S + (size_t)row * a.pitch;
    for (int x = ctx.tid; x < L; x += ctx.nthreads) lds[x] = (x < a.nvalid) ? g[x] : mk(0.f, 0.f);
    ctx.sync();
    fft_forward(ctx, lds, L, 1, a.fd, a.tw);
    if (a.out_map) {
        for (int x = ctx.tid; x < L; x += ctx.nthreads) g[x] = scale(lds[a.out_map[x]], a.scale);
    } else {
        for (int x = ctx.tid; x < L; x += ctx.nthreads) g[x] = scale(lds[x], a.scale);
    }
}

// ---------------------------------------------------------------------------------------
// spectral_rows: per (kernel n, spectrum row i): forward w-transform of the kernel's column
// spectrum row (only kw non-zero inputs: the zero padding is never materialised), pointwise
// complex product with the image spectrum row, sum over features, inverse w-transform.
// Replaces the W half of cufftExecR2C on the padded kernel, elementwiseProductAndNormalize,
// the W half of cufftExecC2R and (by linearity) sumAlongFeatures
// (src/cudaConvolutionFFT.cu:255-282, src/cudaConvFFTData.cuh:47-92).
// ---------------------------------------------------------------------------------------
struct SpectralRowsArgs {
    const c32* A;        // kernel column spectra [n][f][i][a_pitch]
    size_t a_kernel_stride;  // c32 between kernels
    size_t a_feat_stride;    // c32 between features
    int a_pitch;
    int kw;              // non-zero entries per row
    const c32* S;        // image spectrum [f][i][s_pitch]
    size_t s_feat_stride;
    int s_pitch;
    c32* Y;              // [n][i][y_pitch]
    size_t y_kernel_stride;
    int y_pitch;
    int wout;            // columns of Y to write (<= Lw)
    int F;
    FftDesc fd;          // Lw-point transform
    const c32* tw;
};

// lds: L c32 (F == 1) or 2L c32 (F > 1: second half accumulates over features)
template <class Ctx>
FC_HD void spectral_rows_body(const Ctx& ctx, c32* lds, const SpectralRowsArgs& a, int row, int kernel) {
    const int L = a.fd.L;
    c32* buf = lds;
    c32* acc = lds + L;
    for (int f = 0; f < a.F; f++) {
        const c32* arow = a.A + (size_t)kernel * a.a_kernel_stride + (size_t)f * a.a_feat_stride + (size_t)row * a.a_pitch;
        const c32* srow = a.S + (size_t)f * a.s_feat_stride + (size_t)row * a.s_pitch;
        for (int x = ctx.tid; x < L; x += ctx.nthreads) buf[x] = (x < a.kw) ? arow[x] : mk(0.f, 0.f);
        ctx.sync();
        fft_forward(ctx, buf, L, 1, a.fd, a.tw);
        if (a.F == 1) {
            for (int x = ctx.tid; x < L; x += ctx.nthreads) buf[x] = cmul(buf[x], srow[x]);
        } else if (f == 0) {
            for (int x = ctx.tid; x < L; x += ctx.nthreads) acc[x] = cmul(buf[x], srow[x]);
        } else {
            for (int x = ctx.tid; x < L; x += ctx.nthreads) acc[x] = acc[x] + cmul(buf[x], srow[x]);
        }
        ctx.sync();
    }
    c32* res = (a.F == 1) ? buf : acc;
    fft_inverse(ctx, res, L, 1, a.fd, a.tw);
    c32* yrow = a.Y + (size_t)kernel * a.y_kernel_stride + (size_t)row * a.y_pitch;
    for (int x = ctx.tid; x < a.wout; x += ctx.nthreads) yrow[x] = res[x];
}

// ---------------------------------------------------------------------------------------
// cols_c2r: half-complex -> real inverse transform along h of T columns per workgroup,
// written straight into the caller's FFT_H x FFT_W window.  The H half of cufftExecC2R plus
// the final store of sumAlongFeatures (src/cudaConvolutionFFT.cu:273-282).
// ---------------------------------------------------------------------------------------
struct ColsC2RArgs {
    const c32* Y;        // [n][i][y_pitch]
    size_t y_kernel_stride;
    int y_pitch;
    int wvalid;          // columns < wvalid exist in Y; others are zero
    float* out;          // kernel n at out + n*out_kernel_stride; element (h, w) at w*fft_h + h
    size_t out_kernel_stride;
    int fft_h, fft_w;    // output window (reference's ceil16 sizes)
    int M;               // Lh/2
    int T;
    int lds_pitch;
    FftDesc fd;          // M-point transform
    const c32* tw;
    const PairEntry* pairs;
    int npairs;
};

template <class Ctx>
FC_HD void cols_c2r_body(const Ctx& ctx, c32* lds, const ColsC2RArgs& a, int tile, int kernel) {
    const int M = a.M, T = a.T, LP = a.lds_pitch;
    const int w0 = tile * T;
    const c32* Y = a.Y + (size_t)kernel * a.y_kernel_stride;
    // gather rows, t fastest
    for (int idx = ctx.tid; idx < (M + 1) * T; idx += ctx.nthreads) {
        int i = idx / T, t = idx - i * T;
        int w = w0 + t;
        lds[t * LP + i] = (w < a.wvalid) ? Y[(size_t)i * a.y_pitch + w] : mk(0.f, 0.f);
    }
    ctx.sync();
    // merge the half spectrum into the packed complex sequence, in place
    for (int idx = ctx.tid; idx < T * a.npairs; idx += ctx.nthreads) {
        int t = idx / a.npairs, p = idx - t * a.npairs;
        c32* z = lds + t * LP;
        PairEntry e = a.pairs[p];
        if (p == 0) {
            float x0 = z[e.a].x, xm = z[e.b].x;
            z[e.a] = mk(x0 + xm, x0 - xm);
        } else if (e.a == e.b) {
            c32 x = z[e.a];
            z[e.a] = mk(2.f * x.x, -2.f * x.y);
        } else {
            c32 xk = z[e.a], xm = z[e.b];
            c32 Ssum = mk(xk.x + xm.x, xk.y - xm.y);
            c32 D = mk(xk.x - xm.x, xk.y + xm.y);
            c32 G = cmulc(D, e.w);
            z[e.a] = mk(Ssum.x - G.y, Ssum.y + G.x);
            z[e.b] = mk(Ssum.x + G.y, -Ssum.y + G.x);
        }
    }
    ctx.sync();
    fft_inverse(ctx, lds, LP, T, a.fd, a.tw);
    // store: out[w][2n], out[w][2n+1] = re, im of z[n]; zero-fill up to fft_h
    float* out = a.out + (size_t)kernel * a.out_kernel_stride;
    const int half = a.fft_h / 2;  // fft_h is a multiple of 16
    for (int idx = ctx.tid; idx < T * half; idx += ctx.nthreads) {
        int t = idx / half, n = idx - t * half;
        int w = w0 + t;
        if (w < a.fft_w) {
            c32 v = (n < M) ? lds[t * LP + n] : mk(0.f, 0.f);
            *reinterpret_cast<c32*>(out + (size_t)w * a.fft_h + 2 * n) = v;
        }
    }
}

}  // namespace fc
