/*
 * fftconv_cpu_f32.cpp -- single-precision, real-transform CPU restatement of the reference's
 * CPU path (demoCudaConvolutionFFT.m:78-102: fft2(x, fft_h, fft_w) .* fft2(k, fft_h, fft_w),
 * ifft2 per channel, real(sum(., 3))), multithreaded over the kernels.
 *
 * TEST / MEASUREMENT INFRASTRUCTURE ONLY (see fftconv_oracle.h): the second CPU baseline that
 * SURVEY.md 8(d) asks for beside the complex128 oracle -- what a tuned host implementation of the
 * same maths costs (half spectra, fp32) -- timed by bench.py's cpu_baseline leg and pinned to
 * the float64 oracle by tests/test_oracle.py.  Never linked into or loaded by libfftconv.so.
 *
 * Layout as the reference's: planes are column-major, h contiguous (src/cudaConvFFTData.cuh:
 * 26-27); the halved dimension of the spectrum is h (src/cudaConvolutionFFT.cu:122-142).
 * Transforms are Stockham autosort FFTs (radices 2, 3, 4, 5 and any odd prime); the w-direction
 * transforms run over whole [w][CH] planes with the CH = FFT_H/2 + 1 spectrum rows as the
 * contiguous inner loop, so nothing is ever transposed.
 */
#include <cmath>
#include <complex>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

typedef std::complex<float> cf;

int ceil16(int n) { return (n + 15) / 16 * 16; }   // src/cudaConvFFTData.h:96-102

std::vector<int> radices_of(int n) {
    std::vector<int> r;
    while (n % 4 == 0) { r.push_back(4); n /= 4; }
    while (n % 2 == 0) { r.push_back(2); n /= 2; }
    for (int p = 3; p <= n; p += 2)
        while (n % p == 0) { r.push_back(p); n /= p; }
    return r;
}

// One Stockham plan: length n, twiddles w_n^j (double-computed).
struct Plan {
    int n = 0;
    std::vector<int> rad;
    std::vector<cf> tw;   // exp(-2 pi i j / n), j < n
    explicit Plan(int n_) : n(n_), rad(radices_of(n_)), tw(n_) {
        for (int j = 0; j < n; j++) {
            const double a = -2.0 * M_PI * (double)j / (double)n;
            tw[j] = cf((float)std::cos(a), (float)std::sin(a));
        }
    }
};

inline cf mul_i(cf a, bool inverse) { return inverse ? cf(-a.imag(), a.real()) : cf(a.imag(), -a.real()); }   // a * (+-i)

// One radix-R pass over a block of `len` points with stride s (q: the contiguous batch index):
//   y[q + s*(R*p + k)] = w_len^{p k} * sum_j x[q + s*(p + j*m)] * w_R^{j k},  m = len / R.
// `tstep` = n / len indexes the plan's twiddle table; inverse conjugates every root.
template <int R>
void pass_small(const Plan& P, int len, long s, bool inverse, const cf* x, cf* y) {
    const int m = len / R, tstep = P.n / len;
    for (int p = 0; p < m; p++) {
        cf w[R];
        for (int k = 1; k < R; k++) {
            const cf t = P.tw[(size_t)((long)p * k % len) * tstep];
            w[k] = inverse ? std::conj(t) : t;
        }
        const cf* xp = x + s * p;
        cf* yp = y + s * (long)R * p;
        if (R == 2) {
            for (long q = 0; q < s; q++) {
                const cf a = xp[q], b = xp[q + s * m];
                yp[q] = a + b;
                yp[q + s] = (a - b) * w[1];
            }
        } else if (R == 4) {
            for (long q = 0; q < s; q++) {
                const cf a = xp[q], b = xp[q + s * m], c = xp[q + 2 * s * m], d = xp[q + 3 * s * m];
                const cf s0 = a + c, s1 = a - c, s2 = b + d, s3 = mul_i(b - d, inverse);
                yp[q] = s0 + s2;
                yp[q + s] = (s1 + s3) * w[1];
                yp[q + 2 * s] = (s0 - s2) * w[2];
                yp[q + 3 * s] = (s1 - s3) * w[3];
            }
        }
    }
}

// any radix (3, 5, 7, 11, ...): O(R^2) butterfly with roots from the table
void pass_any(const Plan& P, int R, int len, long s, bool inverse, const cf* x, cf* y) {
    const int m = len / R, tstep = P.n / len, rstep = P.n / R;
    std::vector<cf> root(R), w(R), a(R);
    for (int k = 0; k < R; k++) {
        const cf t = P.tw[(size_t)k * rstep];
        root[k] = inverse ? std::conj(t) : t;
    }
    for (int p = 0; p < m; p++) {
        for (int k = 0; k < R; k++) {
            const cf t = P.tw[(size_t)((long)p * k % len) * tstep];
            w[k] = inverse ? std::conj(t) : t;
        }
        const cf* xp = x + s * p;
        cf* yp = y + s * (long)R * p;
        for (long q = 0; q < s; q++) {
            for (int j = 0; j < R; j++) a[j] = xp[q + s * (long)j * m];
            for (int k = 0; k < R; k++) {
                cf acc = a[0];
                for (int j = 1; j < R; j++) acc += a[j] * root[(j * k) % R];
                yp[q + s * k] = acc * w[k];
            }
        }
    }
}

// Transform of length P.n over `batch` interleaved sequences: element t of sequence q at
// buf[q + batch * t].  Ping-pongs between buf and tmp; returns the buffer holding the result.
cf* fft_batch(const Plan& P, long batch, bool inverse, cf* buf, cf* tmp) {
    cf *x = buf, *y = tmp;
    int len = P.n;
    long s = batch;
    for (int R : P.rad) {
        if (R == 2) pass_small<2>(P, len, s, inverse, x, y);
        else if (R == 4) pass_small<4>(P, len, s, inverse, x, y);
        else pass_any(P, R, len, s, inverse, x, y);
        len /= R;
        s *= R;
        std::swap(x, y);
    }
    return x;
}

// real column of n = 2M samples (zero beyond `valid`) -> M + 1 spectrum bins
void rfft_col(const Plan& PM, const float* col, int valid, cf* z, cf* tmp, cf* bins) {
    const int M = PM.n;
    for (int t = 0; t < M; t++) {
        const float re = (2 * t < valid) ? col[2 * t] : 0.f, im = (2 * t + 1 < valid) ? col[2 * t + 1] : 0.f;
        z[t] = cf(re, im);
    }
    cf* Z = fft_batch(PM, 1, false, z, tmp);
    bins[0] = cf(Z[0].real() + Z[0].imag(), 0.f);
    bins[M] = cf(Z[0].real() - Z[0].imag(), 0.f);
    for (int k = 1; k < M; k++) {
        const cf a = Z[k], b = std::conj(Z[M - k]);
        const double ang = -M_PI * (double)k / (double)M;
        const cf w((float)std::cos(ang), (float)std::sin(ang));
        const cf e = 0.5f * (a + b), o = cf(0.f, -0.5f) * (a - b);
        bins[k] = e + w * o;
    }
}

// M + 1 spectrum bins -> real column of n = 2M samples (unnormalised inverse: n * x)
void irfft_col(const Plan& PM, const cf* bins, cf* z, cf* tmp, float* col) {
    const int M = PM.n;
    for (int k = 0; k < M; k++) {
        const cf a = bins[k], b = std::conj(bins[M - k]);
        const double ang = M_PI * (double)k / (double)M;
        const cf w((float)std::cos(ang), (float)std::sin(ang));
        z[k] = (a + b) + cf(0.f, 1.f) * w * (a - b);
    }
    cf* Z = fft_batch(PM, 1, true, z, tmp);
    for (int t = 0; t < M; t++) { col[2 * t] = Z[t].real(); col[2 * t + 1] = Z[t].imag(); }
}

// forward 2-D real transform of an h x w plane into spec[x][CH] (unnormalised)
cf* rfft2_plane(const Plan& PM, const Plan& PW, const float* plane, int h, int w, int CH, cf* spec, cf* tmp, cf* z, cf* zt) {
    const int FW = PW.n;
    for (int x = 0; x < FW; x++) {
        if (x < w) rfft_col(PM, plane + (size_t)x * h, h, z, zt, spec + (size_t)x * CH);
        else std::memset(static_cast<void*>(spec + (size_t)x * CH), 0, (size_t)CH * sizeof(cf));
    }
    return fft_batch(PW, CH, false, spec, tmp);
}

}  // namespace

extern "C" {

/* Same contract as oracle_conv_fft (fftconv_oracle.h).  threads <= 0: all. */
int cpu_f32_conv_fft(const float* data, int H, int W, int F, int max_kernel_h, int max_kernel_w, int n_kernel,
                     const float* const* kernels, const int* kh, const int* kw, float* const* out, int threads) {
    if (H < 1 || W < 1 || F < 1 || max_kernel_h < 1 || max_kernel_w < 1 || n_kernel < 0) return -1;
    const int FH = ceil16(H + max_kernel_h - 1), FW = ceil16(W + max_kernel_w - 1);   // src/cudaConvolutionFFT.cu:103-110
    const int M = FH / 2, CH = M + 1;
    for (int k = 0; k < n_kernel; k++)
        if (kh[k] < 1 || kw[k] < 1 || kh[k] > FH || kw[k] > FW) return -1;             // :242
    const Plan PM(M), PW(FW);
    const size_t plane = (size_t)FW * CH;
    // image spectra, once (src/cudaConvolutionFFT.cu:144-169), pre-scaled by 1/(FFT_W*FFT_H) (:270)
    std::vector<cf> D((size_t)F * plane);
    {
        std::vector<cf> a(plane), b(plane), z(M), zt(M);
        const float scale = 1.0f / ((float)FW * (float)FH);
        for (int f = 0; f < F; f++) {
            const cf* r = rfft2_plane(PM, PW, data + (size_t)f * H * W, H, W, CH, a.data(), b.data(), z.data(), zt.data());
            for (size_t i = 0; i < plane; i++) D[(size_t)f * plane + i] = r[i] * scale;
        }
    }
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = threads > 0 ? threads : omp_get_max_threads();
    if (nthreads > n_kernel) nthreads = n_kernel > 0 ? n_kernel : 1;
#endif
    (void)threads;
#pragma omp parallel num_threads(nthreads)
    {
        std::vector<cf> a(plane), b(plane), z(M), zt(M);
        std::vector<float> col(FH);
#pragma omp for schedule(dynamic, 1)
        for (int k = 0; k < n_kernel; k++) {
            float* o = out[k];
            for (int f = 0; f < F; f++) {
                const float* kp = kernels[k] + (size_t)f * kh[k] * kw[k];
                cf* K = rfft2_plane(PM, PW, kp, kh[k], kw[k], CH, a.data(), b.data(), z.data(), zt.data());   // :245-255
                const cf* Df = D.data() + (size_t)f * plane;
                for (size_t i = 0; i < plane; i++) K[i] *= Df[i];                                               // :263-271
                cf* other = (K == a.data()) ? b.data() : a.data();
                const cf* R = fft_batch(PW, CH, true, K, other);                                                // :273 (w half)
                for (int x = 0; x < FW; x++) {                                                                  // :273 (h half) + :276-282
                    irfft_col(PM, R + (size_t)x * CH, z.data(), zt.data(), col.data());
                    float* oc = o + (size_t)x * FH;
                    if (f == 0) std::memcpy(oc, col.data(), (size_t)FH * sizeof(float));
                    else for (int y = 0; y < FH; y++) oc[y] += col[y];
                }
            }
        }
    }
    return 0;
}

}  // extern "C"
