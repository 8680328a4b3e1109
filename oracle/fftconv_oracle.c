/*
 * fftconv_oracle.c -- CPU restatement (double precision) of the reference's
 * cudaConvolutionFFT hot path.  TEST INFRASTRUCTURE ONLY; see fftconv_oracle.h.
 * PARITY UNPINNED by the reference's own tests (it has none): pinned against NumPy
 * float64 fft2/ifft2, brute-force convolution and the demo script's invariants in
 * tests/test_oracle.py.
 *
 * Each step cites the reference lines it follows (paths relative to /root/reference):
 *   sizing            src/cudaConvFFTData.h:96-102, src/cudaConvolutionFFT.cu:103-112
 *   zero padding      src/cudaConvFFTData.cuh:24-30          (top-left corner, h contiguous)
 *   transform         src/cudaConvolutionFFT.cu:122-142,167,255 (rank-2, n={FFT_W,FFT_H}, batch F)
 *                     demoCudaConvolutionFFT.m:78-88          (fft2(x, fft_h, fft_w) per channel)
 *   product + scale   src/cudaConvFFTData.cuh:62-65, src/cudaConvolutionFFT.cu:270
 *                     (plain complex product, NOT conjugate; scale 1/(FFT_W*FFT_H))
 *   inverse           src/cudaConvolutionFFT.cu:273, demoCudaConvolutionFFT.m:98-102
 *   feature sum       src/cudaConvFFTData.cuh:84-90          (z ascending)
 *   output shape      src/cudaConvolutionFFT.cu:198-200,284  (full FFT_H x FFT_W window)
 *
 * The DFT itself is cuFFT's in the reference (closed source, CUDA 6.0 -- compile.m:2);
 * here it is a textbook mixed-radix decimation-in-time FFT with a naive DFT for any
 * remaining prime factor, so every length is supported.
 */
#include "fftconv_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { double re, im; } cplx;

/* ------------------------------------------------------------------ sizing */

/* src/cudaConvFFTData.h:96-102 */
int oracle_fft_size16(int data_size)
{
    int mod = data_size / 16;
    int rem = data_size % 16;
    return (mod * 16) + ((rem > 0) ? 16 : 0);
}

int oracle_num_threads(int threads)
{
#ifdef _OPENMP
    int mx = omp_get_max_threads();
    if (threads <= 0 || threads > mx) return mx;
    return threads;
#else
    (void)threads;
    return 1;
#endif
}

/* ------------------------------------------------------------------ 1-D FFT */

typedef struct {
    int n;
    int nfac;
    int fac[64];
    cplx *tw;       /* tw[k] = exp(-2*pi*i*k/n), k in [0,n) */
} plan1d;

static int plan1d_init(plan1d *p, int n)
{
    p->n = n;
    p->nfac = 0;
    int m = n;
    while (m % 4 == 0) { p->fac[p->nfac++] = 4; m /= 4; }
    while (m % 2 == 0) { p->fac[p->nfac++] = 2; m /= 2; }
    for (int f = 3; (long)f * f <= m; f += 2)
        while (m % f == 0) { p->fac[p->nfac++] = f; m /= f; }
    if (m > 1) p->fac[p->nfac++] = m;
    p->tw = (cplx *)malloc(sizeof(cplx) * (size_t)n);
    if (!p->tw) return -1;
    for (int k = 0; k < n; k++) {
        double a = -2.0 * M_PI * (double)k / (double)n;
        p->tw[k].re = cos(a);
        p->tw[k].im = sin(a);
    }
    return 0;
}

static void plan1d_free(plan1d *p) { free(p->tw); p->tw = NULL; }

static inline cplx cmul(cplx a, cplx b)
{
    cplx r = { a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re };
    return r;
}

/* twiddle exp(sign*2*pi*i*idx/N) from the forward table */
static inline cplx twid(const plan1d *p, long idx, int sign)
{
    cplx w = p->tw[idx % p->n];
    if (sign > 0) w.im = -w.im;
    return w;
}

/*
 * Decimation in time.  Computes out[k], k<n, of the length-n DFT of
 * in[0], in[is], in[2*is], ...;  `ts` = N/n is the stride into the twiddle table.
 * sign = -1 forward, +1 inverse (unnormalised).
 */
static void fft_rec(const plan1d *P, int n, const cplx *in, long is, cplx *out,
                    int ts, int fi, int sign, cplx *tmp)
{
    if (n == 1) { out[0] = in[0]; return; }
    const int p = P->fac[fi];
    const int m = n / p;
    for (int r = 0; r < p; r++)
        fft_rec(P, m, in + (long)r * is, is * p, out + (long)r * m, ts * p, fi + 1, sign, tmp);

    if (p == 2) {
        for (int k = 0; k < m; k++) {
            cplx a = out[k];
            cplx b = cmul(out[m + k], twid(P, (long)k * ts, sign));
            out[k].re = a.re + b.re;     out[k].im = a.im + b.im;
            out[m + k].re = a.re - b.re; out[m + k].im = a.im - b.im;
        }
    } else if (p == 4) {
        for (int k = 0; k < m; k++) {
            cplx a = out[k];
            cplx b = cmul(out[m + k],     twid(P, (long)k * ts, sign));
            cplx c = cmul(out[2 * m + k], twid(P, 2L * k * ts, sign));
            cplx d = cmul(out[3 * m + k], twid(P, 3L * k * ts, sign));
            cplx s0 = { a.re + c.re, a.im + c.im }, s1 = { a.re - c.re, a.im - c.im };
            cplx s2 = { b.re + d.re, b.im + d.im }, s3 = { b.re - d.re, b.im - d.im };
            /* forward: -i*s3 ; inverse: +i*s3 */
            cplx js3;
            if (sign < 0) { js3.re = s3.im;  js3.im = -s3.re; }
            else          { js3.re = -s3.im; js3.im = s3.re; }
            out[k].re         = s0.re + s2.re;  out[k].im         = s0.im + s2.im;
            out[m + k].re     = s1.re + js3.re; out[m + k].im     = s1.im + js3.im;
            out[2 * m + k].re = s0.re - s2.re;  out[2 * m + k].im = s0.im - s2.im;
            out[3 * m + k].re = s1.re - js3.re; out[3 * m + k].im = s1.im - js3.im;
        }
    } else {
        /* generic prime radix: X[k + q*m] = sum_r (t_r) * w_p^{r q},  t_r = out[r*m+k] * w_n^{r k} */
        const int wp = P->n / p;     /* table stride of exp(-2 pi i / p) */
        for (int k = 0; k < m; k++) {
            for (int r = 0; r < p; r++)
                tmp[r] = cmul(out[(long)r * m + k], twid(P, (long)r * k * ts, sign));
            for (int q = 0; q < p; q++) {
                cplx acc = tmp[0];
                for (int r = 1; r < p; r++) {
                    cplx w = twid(P, ((long)r * q % p) * wp, sign);
                    acc.re += tmp[r].re * w.re - tmp[r].im * w.im;
                    acc.im += tmp[r].re * w.im + tmp[r].im * w.re;
                }
                out[(long)q * m + k] = acc;
            }
        }
    }
}

/* in-place-looking helper: y = DFT(x), x and y distinct, contiguous */
static void fft1d(const plan1d *P, const cplx *x, cplx *y, int sign, cplx *tmp)
{
    fft_rec(P, P->n, x, 1, y, 1, 0, sign, tmp);
}

/* ------------------------------------------------------------------ 2-D FFT
 * Array a is FFT_W columns of FFT_H contiguous elements (MATLAB column-major,
 * matching the cuFFT geometry n={FFT_W,FFT_H} of src/cudaConvolutionFFT.cu:122-142).
 */
typedef struct {
    int fh, fw;
    plan1d ph, pw;
    int maxp;
} plan2d;

static int max_factor(const plan1d *p)
{
    int mx = 1;
    for (int i = 0; i < p->nfac; i++) if (p->fac[i] > mx) mx = p->fac[i];
    return mx;
}

static int plan2d_init(plan2d *P, int fh, int fw)
{
    P->fh = fh; P->fw = fw;
    if (plan1d_init(&P->ph, fh)) return -1;
    if (plan1d_init(&P->pw, fw)) { plan1d_free(&P->ph); return -1; }
    int a = max_factor(&P->ph), b = max_factor(&P->pw);
    P->maxp = a > b ? a : b;
    return 0;
}

static void plan2d_free(plan2d *P) { plan1d_free(&P->ph); plan1d_free(&P->pw); }

/* work: at least 2*max(fh,fw) + maxp cplx.  ncols_nz: columns >= ncols_nz are known zero
 * on input of a forward transform (skips their column FFTs; pure speed, same result). */
static void fft2d(const plan2d *P, cplx *a, int sign, cplx *work, int ncols_nz)
{
    const int fh = P->fh, fw = P->fw;
    const int mx = fh > fw ? fh : fw;
    cplx *bi = work, *bo = work + mx, *tmp = work + 2 * mx;
    /* along H (contiguous) */
    for (int x = 0; x < fw; x++) {
        if (x >= ncols_nz) continue;       /* all-zero column stays zero */
        cplx *col = a + (size_t)x * fh;
        memcpy(bi, col, sizeof(cplx) * (size_t)fh);
        fft1d(&P->ph, bi, bo, sign, tmp);
        memcpy(col, bo, sizeof(cplx) * (size_t)fh);
    }
    /* along W (stride fh) */
    for (int y = 0; y < fh; y++) {
        for (int x = 0; x < fw; x++) bi[x] = a[(size_t)x * fh + y];
        fft1d(&P->pw, bi, bo, sign, tmp);
        for (int x = 0; x < fw; x++) a[(size_t)x * fh + y] = bo[x];
    }
}

/* zero-pad one feature plane into the top-left corner -- src/cudaConvFFTData.cuh:24-30 */
static void pad_plane(cplx *dst, int fh, int fw, const float *src, int h, int w)
{
    memset(dst, 0, sizeof(cplx) * (size_t)fh * fw);
    for (int x = 0; x < w; x++)
        for (int y = 0; y < h; y++)
            dst[(size_t)x * fh + y].re = (double)src[(size_t)x * h + y];
}

/* ------------------------------------------------------------------ the path */

static int conv_fft_impl(const float *data, int H, int W, int F,
                         int max_kernel_h, int max_kernel_w,
                         int n_kernel, const float *const *kernels,
                         const int *kh, const int *kw,
                         float *const *out32, double *const *out64, int threads)
{
    if (H <= 0 || W <= 0 || F <= 0 || max_kernel_h <= 0 || max_kernel_w <= 0 || n_kernel < 0)
        return -1;
    /* src/cudaConvolutionFFT.cu:103-112 */
    const int FFT_H = oracle_fft_size16(H + max_kernel_h - 1);
    const int FFT_W = oracle_fft_size16(W + max_kernel_w - 1);
    const size_t P = (size_t)FFT_H * FFT_W;
    for (int k = 0; k < n_kernel; k++)       /* src/cudaConvolutionFFT.cu:242 */
        if (kh[k] > FFT_H || kw[k] > FFT_W || kh[k] <= 0 || kw[k] <= 0) return -1;

    plan2d plan;
    if (plan2d_init(&plan, FFT_H, FFT_W)) return -2;
    const int mx = FFT_H > FFT_W ? FFT_H : FFT_W;
    const size_t work_n = 2 * (size_t)mx + plan.maxp + 8;

    /* image spectrum, once for all kernels -- src/cudaConvolutionFFT.cu:144-169 */
    cplx *fdata = (cplx *)malloc(sizeof(cplx) * P * (size_t)F);
    if (!fdata) { plan2d_free(&plan); return -2; }
    const int nthr = oracle_num_threads(threads);
    int fail = 0;
#ifdef _OPENMP
#pragma omp parallel num_threads(nthr)
#endif
    {
        cplx *work = (cplx *)malloc(sizeof(cplx) * work_n);
        if (!work) {
#ifdef _OPENMP
#pragma omp atomic write
#endif
            fail = 1;
        }
#ifdef _OPENMP
#pragma omp for schedule(dynamic)
#endif
        for (int z = 0; z < F; z++) {
            if (!work) continue;
            pad_plane(fdata + P * z, FFT_H, FFT_W, data + (size_t)H * W * z, H, W);
            fft2d(&plan, fdata + P * z, -1, work, W);
        }
        free(work);
    }
    if (fail) { free(fdata); plan2d_free(&plan); return -2; }

    /* per-kernel loop -- src/cudaConvolutionFFT.cu:204-291 */
    const double scale = 1.0 / ((double)FFT_W * (double)FFT_H);   /* :270 */
#ifdef _OPENMP
#pragma omp parallel num_threads(nthr)
#endif
    {
        cplx *work = (cplx *)malloc(sizeof(cplx) * work_n);
        cplx *buf = (cplx *)malloc(sizeof(cplx) * P);
        double *acc = (double *)malloc(sizeof(double) * P);
        if (!work || !buf || !acc) {
#ifdef _OPENMP
#pragma omp atomic write
#endif
            fail = 1;
        }
#ifdef _OPENMP
#pragma omp for schedule(dynamic)
#endif
        for (int k = 0; k < n_kernel; k++) {
            if (!work || !buf || !acc) continue;
            for (int z = 0; z < F; z++) {
                /* pad + forward transform of the kernel plane (:245-255) */
                pad_plane(buf, FFT_H, FFT_W, kernels[k] + (size_t)kh[k] * kw[k] * z, kh[k], kw[k]);
                fft2d(&plan, buf, -1, work, kw[k]);
                /* elementwiseProductAndNormalize (cuh:62-65): scale * (data * kernel) */
                const cplx *fd = fdata + P * z;
                for (size_t i = 0; i < P; i++) {
                    double re = scale * (fd[i].re * buf[i].re - fd[i].im * buf[i].im);
                    double im = scale * (fd[i].im * buf[i].re + fd[i].re * buf[i].im);
                    buf[i].re = re; buf[i].im = im;
                }
                /* unnormalised inverse (:273) */
                fft2d(&plan, buf, +1, work, FFT_W);
                /* sumAlongFeatures (cuh:84-90): z ascending, real part (C2R output is real) */
                if (z == 0) for (size_t i = 0; i < P; i++) acc[i] = buf[i].re;
                else        for (size_t i = 0; i < P; i++) acc[i] += buf[i].re;
            }
            if (out32) for (size_t i = 0; i < P; i++) out32[k][i] = (float)acc[i];
            if (out64) memcpy(out64[k], acc, sizeof(double) * P);
        }
        free(work); free(buf); free(acc);
    }
    free(fdata);
    plan2d_free(&plan);
    return fail ? -2 : 0;
}

int oracle_conv_fft(const float *data, int H, int W, int F,
                    int max_kernel_h, int max_kernel_w,
                    int n_kernel, const float *const *kernels,
                    const int *kh, const int *kw,
                    float *const *out, int threads)
{
    return conv_fft_impl(data, H, W, F, max_kernel_h, max_kernel_w, n_kernel, kernels,
                         kh, kw, out, NULL, threads);
}

int oracle_conv_fft_f64(const float *data, int H, int W, int F,
                        int max_kernel_h, int max_kernel_w,
                        int n_kernel, const float *const *kernels,
                        const int *kh, const int *kw,
                        double *const *out, int threads)
{
    return conv_fft_impl(data, H, W, F, max_kernel_h, max_kernel_w, n_kernel, kernels,
                         kh, kw, NULL, out, threads);
}

/* ------------------------------------------------------------------ brute force */

int oracle_conv_direct(const float *data, int H, int W, int F,
                       int max_kernel_h, int max_kernel_w,
                       const float *kernel, int kh, int kw,
                       double *out)
{
    const int FFT_H = oracle_fft_size16(H + max_kernel_h - 1);
    const int FFT_W = oracle_fft_size16(W + max_kernel_w - 1);
    if (kh > FFT_H || kw > FFT_W) return -1;
    memset(out, 0, sizeof(double) * (size_t)FFT_H * FFT_W);
    for (int z = 0; z < F; z++)
        for (int kx = 0; kx < kw; kx++)
            for (int ky = 0; ky < kh; ky++) {
                const double kv = (double)kernel[(size_t)z * kh * kw + (size_t)kx * kh + ky];
                if (kv == 0.0) continue;
                for (int x = 0; x < W; x++) {
                    const int ox = (x + kx) % FFT_W;
                    const float *dcol = data + (size_t)z * H * W + (size_t)x * H;
                    double *ocol = out + (size_t)ox * FFT_H;
                    for (int y = 0; y < H; y++)
                        ocol[(y + ky) % FFT_H] += kv * (double)dcol[y];
                }
            }
    return 0;
}
