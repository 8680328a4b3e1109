// cudaConvFFTDataStreams_mex.cpp -- MATLAB MEX gateway, the reference's multi-GPU / multi-stream entry
// (src/cudaConvFFTDataStreams.cu:121-522; it never built: SURVEY.md section 2, #5) in working form:
//
//   cvcell = cudaConvFFTDataStreams(fftData, kernelCell[, threadSize][, gpuIds])
//
// fftData is the complex single gpuArray cudaFFTData returns ((FFT_H/2+1) x FFT_W x F, :160-187);
// the kernels of the cell are dealt over one plan per GPU (contiguous blocks; the reference's sketch deals
// them round-robin over N_GPU x N_BATCH_PER_GPU plans, :273-328,338-447), the spectrum is copied from the
// first GPU to the others (:279-289), every map is the full FFT_H x FFT_W window in host memory, and the
// call returns when all GPUs are done (:452-468).  gpuIds (extension, 0-based): the devices to use, a
// device may be listed twice; default: every visible device.
// Needs the MathWorks GPU header (gpu/mxGPUArray.h) for its gpuArray argument.
#include <vector>

#include "fftconv.h"
#include "mex.h"
#if defined(__has_include)
#if __has_include("gpu/mxGPUArray.h")
#include "gpu/mxGPUArray.h"
#define FFTCONV_MEX_GPU 1
#endif
#endif
#ifndef FFTCONV_MEX_GPU
#define FFTCONV_MEX_GPU 0
#endif

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    const char* errId = "parallel:gpu:mexGPUExample:InvalidInput";   // src/cudaConvFFTDataStreams.cu:138
    (void)nlhs;
#if !FFTCONV_MEX_GPU
    (void)plhs; (void)nrhs; (void)prhs;
    mexErrMsgIdAndTxt(errId, "cudaConvFFTDataStreams was built without gpu/mxGPUArray.h: use cudaConvolutionFFT or the handle form of cudaFFTData");
#else
    mxInitGPU();                                                                             // :157
    if (nrhs < 2 || nrhs > 4 || !mxIsGPUArray(prhs[0]))
        mexErrMsgIdAndTxt(errId, "The data must be FFT-ed real array in GPU");               // :160-161
    if (nrhs > 2 && mxGetNumberOfElements(prhs[2]) != 0 && mxGetNumberOfElements(prhs[2]) != 4)
        mexErrMsgIdAndTxt(errId, "CUDA Thread Size must be 4 integers : THREAD_PER_BLOCK_H, THREAD_PER_BLOCK_W, "
                                 "THREAD_PER_BLOCK_D, THREAD_PER_BLOCK_2D");                  // :163-164 (values ignored)
    if (mxGetClassID(prhs[1]) != mxCELL_CLASS) mexErrMsgIdAndTxt(errId, "Kernel must be a cell array");   // :196-197
    std::vector<int> devs;
    if (nrhs > 3) {
        const double* ids = static_cast<const double*>(mxGetData(prhs[3]));
        for (size_t i = 0; i < mxGetNumberOfElements(prhs[3]); i++) devs.push_back((int)ids[i]);
    } else {
        int ndev = 0;
        if (fftconv_device_count(&ndev) != FFTCONV_OK) mexErrMsgIdAndTxt(errId, "%s", fftconv_last_error());
        for (int g = 0; g < ndev; g++) devs.push_back(g);                                    // N_GPU (:271 forces 1 in the reference)
    }
    const mxGPUArray* fd = mxGPUCreateFromMxArray(prhs[0]);
    const mwSize* fdim = mxGPUGetDimensions(fd);
    const mwSize fnd = mxGPUGetNumberOfDimensions(fd);
    bool good = mxGPUGetClassID(fd) == mxSINGLE_CLASS && mxGPUGetComplexity(fd) == mxCOMPLEX && fnd >= 2 && fnd <= 3 && fdim[0] >= 2;
    const int FFT_H = good ? ((int)fdim[0] - 1) * 2 : 0, FFT_W = good ? (int)fdim[1] : 0, F = good ? (fnd == 3 ? (int)fdim[2] : 1) : 0;   // :176-187
    // cudaFFTData returns ceil16-sized spectra only; a plan's window is the ceil16 of what it is given, so any other
    // size would make it read past the array
    if (good && (FFT_H % 16 != 0 || FFT_W % 16 != 0)) good = false;
    fftconv_multi* m = nullptr;
    auto fail = [&](const char* msg) {
        if (fd) mxGPUDestroyGPUArray(fd);
        if (m) fftconv_multi_destroy(m);
        mexErrMsgIdAndTxt(errId, "%s", msg);
    };
    if (!good) fail("The data must be FFT-ed real array in GPU");

    // the kernel cell before the plans: its largest kernel decides which row kernel they may use
    const int n = (int)mxGetNumberOfElements(prhs[1]);
    std::vector<const float*> kp(n);
    std::vector<int> kh(n), kw(n);
    int max_kh = 1, max_kw = 1;
    for (int k = 0; k < n; k++) {
        const mxArray* c = mxGetCell(prhs[1], k);
        const mwSize knd = (c && !mxIsGPUArray(c)) ? mxGetNumberOfDimensions(c) : 0;
        // host kernels only here: a device-resident kernel would have to live on the GPU that owns it
        if (!c || mxIsGPUArray(c) || mxGetClassID(c) != mxSINGLE_CLASS || knd < 2 || knd > 3 || (knd == 3 ? (int)mxGetDimensions(c)[2] : 1) != F)
            fail("Kernels must be host arrays of type float with the data's number of features");
        const mwSize* kd = mxGetDimensions(c);
        kp[k] = static_cast<const float*>(mxGetData(c));
        kh[k] = (int)kd[0]; kw[k] = (int)kd[1];
        if (kh[k] > max_kh) max_kh = kh[k];
        if (kw[k] > max_kw) max_kw = kw[k];
    }
    if (max_kh > FFT_H || max_kw > FFT_W)                                                    // :242-243 of the one-shot source
        fail("Kernel and Data must have the same number of features and kernel size should be smaller than data size");
    fftconv_plan_options opts = {};
    opts.struct_size = sizeof(opts);
    opts.exact_window = 1;      // data size + largest kernel - 1 == window: window and transform are FFT_H x FFT_W
    int rc = fftconv_multi_create(&m, FFT_H - max_kh + 1, FFT_W - max_kw + 1, F, max_kh, max_kw, devs.data(), (int)devs.size(), &opts);
    if (rc == FFTCONV_OK) {
        fftconv_plan* p0 = nullptr;
        fftconv_plan_info pi;
        rc = fftconv_multi_plan(m, 0, &p0, nullptr);
        if (rc == FFTCONV_OK) rc = fftconv_plan_get_info(p0, &pi);
        if (rc == FFTCONV_OK && (pi.fft_h != FFT_H || pi.fft_w != FFT_W || pi.spectrum_rows != (int)fdim[0]))
            fail("The data must be FFT-ed real array in GPU");    // the spectrum the plans would read is not the array given
    }
    if (rc == FFTCONV_OK) rc = fftconv_multi_import_spectrum(m, static_cast<const float*>(mxGPUGetDataReadOnly(fd)), FFTCONV_DEVICE);
    mxGPUDestroyGPUArray(fd);
    fd = nullptr;
    if (rc != FFTCONV_OK) fail(fftconv_last_error());

    plhs[0] = mxCreateCellMatrix(1, n);
    std::vector<float*> out(n);
    const mwSize cdims[2] = {(mwSize)FFT_H, (mwSize)FFT_W};
    for (int k = 0; k < n; k++) {
        mxArray* a = mxCreateNumericArray(2, cdims, mxSINGLE_CLASS, mxREAL);
        out[k] = static_cast<float*>(mxGetData(a));
        mxSetCell(plhs[0], k, a);
    }
    rc = fftconv_multi_convolve(m, n, kp.data(), kh.data(), kw.data(), FFTCONV_HOST, out.data(), FFTCONV_HOST);
    if (rc != FFTCONV_OK) fail(fftconv_last_error());
    fftconv_multi_destroy(m);
#endif
}
