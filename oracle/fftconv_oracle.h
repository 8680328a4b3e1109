/*
 * fftconv_oracle.h -- CPU restatement of the reference's cudaConvolutionFFT path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked into, imported by or
 * called from the product library (libfftconv.so).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load liboracle.so, and only as the checker /
 * the timed CPU baseline -- never as the thing shipped.
 *
 * PARITY UNPINNED BY THE REFERENCE: the reference ships no tests, golden vectors or
 * fixtures for this path, its arithmetic lives in closed-source
 * cuFFT (CUDA 6.0, compile.m:2; not in /root/reference), and it cannot be built here
 * (needs nvcc, cufft.h, MATLAB mex.h, gpu/mxGPUArray.h -- src/cudaConvolutionFFT.cu:1-4).
 * The oracle is therefore pinned by (i) the DFT definition, (ii) NumPy float64
 * fft2/ifft2 on the same inputs (tests/test_oracle.py), (iii) brute-force direct
 * convolution (oracle_conv_direct below), (iv) the invariants the reference's demo
 * script encodes (demoCudaConvolutionFFT.m:57-69,91-102,110-113) and (v), on the GPU box
 * only, the vendor's FFT library of this platform -- rocFFT behind hipFFT, the counterpart
 * of the cuFFT the reference calls, reached through torch.fft -- running the reference's
 * own sequence in float64 (tests/test_gpu_parity.py::
 * test_matches_the_vendor_fft_library_on_the_device: 3.6e-8 ... 5.9e-8 apart).
 */
#ifndef FFTCONV_ORACLE_H
#define FFTCONV_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* computeFFTsize16 -- src/cudaConvFFTData.h:96-102 */
int oracle_fft_size16(int data_size);

/*
 * The reference's one-shot path (src/cudaConvolutionFFT.cu:27-311) restated in double
 * precision with full complex transforms, i.e. the shape of the reference's own CPU
 * path (demoCudaConvolutionFFT.m:78-102: fft2(x, fft_h, fft_w) .* fft2(k, fft_h, fft_w),
 * ifft2, real(sum(.,3))).
 *
 *   data     : H x W x F, MATLAB column-major (index z*H*W + x*H + y, cudaConvFFTData.cuh:26-27)
 *   kernels  : n_kernel pointers, kernel k is kh[k] x kw[k] x F column-major
 *   out      : n_kernel caller buffers of FFT_H*FFT_W floats (column-major FFT_H x FFT_W,
 *              src/cudaConvolutionFFT.cu:198-200), FFT_X = fft_size16(DATA_X + MAXK_X - 1)
 *              (src/cudaConvolutionFFT.cu:103-110)
 *   threads  : number of OpenMP threads used over kernels (<=0: all)
 * returns 0, or -1 on invalid sizes (kernel larger than the FFT window,
 * src/cudaConvolutionFFT.cu:242).
 */
int oracle_conv_fft(const float *data, int H, int W, int F,
                    int max_kernel_h, int max_kernel_w,
                    int n_kernel, const float *const *kernels,
                    const int *kh, const int *kw,
                    float *const *out, int threads);

/* Same, results kept in double (for tolerance studies). */
int oracle_conv_fft_f64(const float *data, int H, int W, int F,
                        int max_kernel_h, int max_kernel_w,
                        int n_kernel, const float *const *kernels,
                        const int *kh, const int *kw,
                        double *const *out, int threads);

/*
 * Brute-force circular convolution modulo (FFT_H, FFT_W) summed over features, in
 * double: what the FFT path computes by the convolution theorem, including the
 * wrap-around the reference does not guard against.  O(H*W*kh*kw*F).
 */
int oracle_conv_direct(const float *data, int H, int W, int F,
                       int max_kernel_h, int max_kernel_w,
                       const float *kernel, int kh, int kw,
                       double *out);

/* number of threads oracle_conv_fft would use for `threads` */
int oracle_num_threads(int threads);

#ifdef __cplusplus
}
#endif
#endif
