/*
 * demo_planted_template.c -- a plain C caller of libfftconv.so doing what the reference's demo
 * script does (demoCudaConvolutionFFT.m:37-69,124-129): a 64 x 8 x 5 image with the template
 * kernel(:,:,1) planted in channel 1 at (5,2) (1-based), kernels flipped, one call of the one-shot
 * entry (the MEX body, src/cudaConvolutionFFT.cu:27-311), and -- what the script leaves to the eye --
 * a check that probing channel 1 with the flipped template answers with sum(template^2) = 22140 at
 * the planted offset shifted by (cn-1, cm-1).  With a second argument > 1 the same call goes through
 * the multi-device entry (src/cudaConvFFTDataStreams.cu's intent) with device 0 listed that many times.
 *
 *   gcc -std=c99 -Iinclude examples/demo_planted_template.c -Lcuda-fft-convolution_amd -lfftconv \
 *       -Wl,-rpath,$PWD/cuda-fft-convolution_amd -Wl,-rpath-link,/opt/rocm/lib -lm -o demo
 *   ./demo [n_plans_on_device_0]
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "fftconv.h"

enum { N = 64, M = 8, K = 5, CN = 10, CM = 4 };

int main(int argc, char **argv) {
    const int nplans = argc > 1 ? atoi(argv[1]) : 1;
    float *data = calloc((size_t)N * M * K, sizeof(float));   /* column-major n x m x k */
    float *probe = calloc((size_t)CN * CM * K, sizeof(float));
    unsigned s = 12345u;
    for (int i = 0; i < N * M * K; i++) { s = s * 1664525u + 1013904223u; data[i] = (float)(s >> 8) / 16777216.0f; }
    /* template: reshape(1:cn*cm, cn, cm); planted at data(5:(4+cn), 2:(1+cm), 1) */
    for (int x = 0; x < CM; x++)
        for (int y = 0; y < CN; y++) data[(size_t)(1 + x) * N + (4 + y)] = (float)(1 + x * CN + y);
    /* the probe: the template, flipped along both axes ("Flip Kernel (Required)"), in channel 1 only */
    for (int x = 0; x < CM; x++)
        for (int y = 0; y < CN; y++) probe[(size_t)x * CN + y] = (float)(1 + (CM - 1 - x) * CN + (CN - 1 - y));
    const int fh = fftconv_fft_size16(N + CN - 1), fw = fftconv_fft_size16(M + CM - 1);   /* 80 x 16 */
    float *map = malloc((size_t)fh * fw * sizeof(float));
    const float *kernels[1] = {probe};
    float *out[1] = {map};
    const int kh[1] = {CN}, kw[1] = {CM}, kf[1] = {K};
    const double threads[4] = {8, 8, 8, 16};   /* accepted and ignored (src/cudaConvolutionFFT.cu:72-82) */
    int rc, oh = 0, ow = 0;
    if (nplans <= 1) {
        rc = fftconv_convolution_fft(data, N, M, K, CN, CM, 1, kernels, kh, kw, kf, threads, 4, 0, out, &oh, &ow);
    } else {
        int *devs = calloc((size_t)nplans, sizeof(int));   /* device 0, nplans times */
        rc = fftconv_convolution_fft_multi(data, N, M, K, CN, CM, 1, kernels, kh, kw, kf, devs, nplans, out, &oh, &ow);
        free(devs);
    }
    if (rc != FFTCONV_OK) {
        fprintf(stderr, "fftconv status %d: %s\n", rc, fftconv_last_error());
        return 2;
    }
    int by = 0, bx = 0;
    for (int x = 0; x < fw; x++)
        for (int y = 0; y < fh; y++)
            if (map[(size_t)x * fh + y] > map[(size_t)bx * fh + by]) { by = y; bx = x; }
    const double peak = map[(size_t)bx * fh + by], want = 22140.0;   /* sum_{v=1..40} v^2 */
    printf("window %d x %d, peak %.3f at (%d, %d); expected %.0f at (%d, %d)\n", oh, ow, peak, by, bx, want, 4 + CN - 1, 1 + CM - 1);
    const int ok = oh == fh && ow == fw && by == 4 + CN - 1 && bx == 1 + CM - 1 && fabs(peak - want) <= 1e-4 * want;
    free(data); free(probe); free(map);
    puts(ok ? "OK" : "MISMATCH");
    return ok ? 0 : 1;
}
