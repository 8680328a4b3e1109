// cudaConvFFTData_mex.cpp -- MATLAB MEX gateway, step 2 of the reference's two-step API
// (src/cudaConvFFTData.cu:24-306):
//
//   cvcell = cudaConvFFTData(fftData, kernelCell[, threadSize])
//
// fftData is what cudaFFTData returned: the complex single gpuArray of the reference
// ((FFT_H/2+1) x FFT_W x F, src/cudaConvFFTData.cu:68-69,90-98: FFT_H = (dim0 - 1) * 2, FFT_W = dim1) --
// imported into a plan whose transform is that window (fftconv_plan_import_spectrum) -- or the
// opaque uint64 handle of the alternative form, whose plan is reused for every call.
#include <cstdint>
#include <vector>

#include "fftconv.h"
#include "mex.h"
#if defined(__has_include)
#if __has_include("gpu/mxGPUArray.h")
#include "gpu/mxGPUArray.h"   // MathWorks GPU MEX API: gpuArray kernels (src/cudaConvolutionFFT.cu:224-238)
#define FFTCONV_MEX_GPU 1
#endif
#endif
#ifndef FFTCONV_MEX_GPU
#define FFTCONV_MEX_GPU 0
#endif

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    const char* errId = FFTCONV_MEX_ERROR_ID;   // "cudaConvFFTData:InvalidInput" (src/cudaConvFFTData.cu:47)
    (void)nlhs;
    if (nrhs < 2 || nrhs > 3) mexErrMsgIdAndTxt(errId, "The data must be FFT-ed real array in GPU");   // :68-69
    const double* threads = nullptr;
    int nthreads = 0;
    if (nrhs > 2) {                                                                          // :71-72: checked by the library
        threads = static_cast<const double*>(mxGetData(prhs[2]));
        nthreads = (int)mxGetNumberOfElements(prhs[2]);
    }
    if (mxGetClassID(prhs[1]) != mxCELL_CLASS) mexErrMsgIdAndTxt(errId, "Kernel must be a cell array");   // :108-109
    // the kernel cell first: its largest kernel decides which row kernel an imported-spectrum plan may use
    const int n = (int)mxGetNumberOfElements(prhs[1]);
    std::vector<const float*> kp(n);
    std::vector<int> kh(n), kw(n), kf(n);
    bool any_gpu = false;
#if FFTCONV_MEX_GPU
    std::vector<const mxGPUArray*> views;   // released before every way out (mexErrMsg* does not run destructors in MATLAB)
    auto release_views = [&] { for (const mxGPUArray* g : views) mxGPUDestroyGPUArray(g); views.clear(); };
#else
    auto release_views = [] {};
#endif
    auto bad_kernel = [&] {
        release_views();
        mexErrMsgIdAndTxt(errId, "Kernels must be of type float and have features larger than 1");
    };
    int max_kh = 1, max_kw = 1;
    for (int k = 0; k < n; k++) {
        const mxArray* c = mxGetCell(prhs[1], k);
        if (!c) bad_kernel();
        const mwSize* kd = nullptr;
        mwSize knd = 0;
#if FFTCONV_MEX_GPU
        if (mxIsGPUArray(c)) {                                                              // gpuArray kernel (reference src/cudaConvFFTData.cu:153-167)
            const mxGPUArray* g = mxGPUCreateFromMxArray(c);
            views.push_back(g);
            knd = mxGPUGetNumberOfDimensions(g);
            if (mxGPUGetClassID(g) != mxSINGLE_CLASS || knd < 2 || knd > 3) bad_kernel();
            kd = mxGPUGetDimensions(g);
            kp[k] = static_cast<const float*>(mxGPUGetDataReadOnly(g));
            any_gpu = true;
        } else
#endif
        {
            knd = mxGetNumberOfDimensions(c);
            if (mxGetClassID(c) != mxSINGLE_CLASS || knd < 2 || knd > 3) bad_kernel();   // reference: "Kernels must be of type float ..."
            kd = mxGetDimensions(c);
            kp[k] = static_cast<const float*>(mxGetData(c));
        }
        kh[k] = (int)kd[0]; kw[k] = (int)kd[1]; kf[k] = knd == 3 ? (int)kd[2] : 1;
        if (kh[k] > max_kh) max_kh = kh[k];
        if (kw[k] > max_kw) max_kw = kw[k];
    }
    fftconv_plan* plan = nullptr;
    bool own_plan = false;     // a plan made here around an imported gpuArray spectrum: destroyed before returning
#if FFTCONV_MEX_GPU
    if (mxIsGPUArray(prhs[0])) {                                                             // the reference's form (:68, :90-98)
        mxInitGPU();
        const mxGPUArray* fd = mxGPUCreateFromMxArray(prhs[0]);
        const mwSize* fdim = mxGPUGetDimensions(fd);
        const mwSize fnd = mxGPUGetNumberOfDimensions(fd);
        bool good = mxGPUGetClassID(fd) == mxSINGLE_CLASS && mxGPUGetComplexity(fd) == mxCOMPLEX && fnd >= 2 && fnd <= 3 && fdim[0] >= 2;
        const int FFT_H = good ? ((int)fdim[0] - 1) * 2 : 0, FFT_W = good ? (int)fdim[1] : 0, F = good ? (fnd == 3 ? (int)fdim[2] : 1) : 0;   // :92-98
        // cudaFFTData only ever returns ceil16-sized spectra (src/cudaFFTData.cu:72-78): anything else is not its output,
        // and a plan's window (ceil16 of the sizes given) would be larger than the array handed over
        if (good && (FFT_H % 16 != 0 || FFT_W % 16 != 0)) good = false;
        const bool kernels_fit = max_kh <= FFT_H && max_kw <= FFT_W;
        int rc = FFTCONV_ERR_INVALID_ARG;
        if (good && kernels_fit) {
            fftconv_plan_options opts = {};
            opts.struct_size = sizeof(opts);
            opts.exact_window = 1;
            // data size + largest kernel - 1 == window: the window and the transform are FFT_H x FFT_W, the kernels convolve
            // circularly modulo it as the reference's do, and the plan picks a row kernel able to take the widest of them
            rc = fftconv_plan_create_ex(&plan, FFT_H - max_kh + 1, FFT_W - max_kw + 1, F, max_kh, max_kw, -1, nullptr, &opts);
            fftconv_plan_info pi;
            if (rc == FFTCONV_OK) rc = fftconv_plan_get_info(plan, &pi);
            if (rc == FFTCONV_OK && (pi.fft_h != FFT_H || pi.fft_w != FFT_W || pi.spectrum_rows != (int)fdim[0])) {
                good = false;                 // the spectrum the plan would read is not the array it was given
                rc = FFTCONV_ERR_INVALID_ARG;
            }
            if (rc == FFTCONV_OK) rc = fftconv_plan_import_spectrum(plan, static_cast<const float*>(mxGPUGetDataReadOnly(fd)), FFTCONV_DEVICE);
            if (rc == FFTCONV_OK) rc = fftconv_plan_synchronize(plan);
        }
        mxGPUDestroyGPUArray(fd);
        if (rc != FFTCONV_OK) {
            if (plan) fftconv_plan_destroy(plan);
            release_views();
            if (!good) mexErrMsgIdAndTxt(errId, "The data must be FFT-ed real array in GPU");
            if (!kernels_fit)                                                                // src/cudaConvFFTData.cu:181-182
                mexErrMsgIdAndTxt(errId, "Kernel and Data must have the same number of features and kernel size should be smaller than data size");
            mexErrMsgIdAndTxt(errId, "%s", fftconv_last_error());
        }
        own_plan = true;
    } else
#endif
    {
        if (mxGetClassID(prhs[0]) != mxUINT64_CLASS || mxGetNumberOfElements(prhs[0]) != 1) {
            release_views();
            mexErrMsgIdAndTxt(errId, "The data must be FFT-ed real array in GPU");           // :68-69
        }
        plan = reinterpret_cast<fftconv_plan*>((uintptr_t) * static_cast<const uint64_t*>(mxGetData(prhs[0])));
        // a handle cudaFFTData released, one from before `clear mex`, or any stray uint64: refused, not dereferenced
        if (!fftconv_plan_is_live(plan)) {
            release_views();
            mexErrMsgIdAndTxt(errId, "The data must be FFT-ed real array in GPU");
        }
    }
    fftconv_plan_info info;
    if (fftconv_plan_get_info(plan, &info) != FFTCONV_OK) {
        if (own_plan) fftconv_plan_destroy(plan);
        release_views();
        mexErrMsgIdAndTxt(errId, "%s", fftconv_last_error());
    }
    (void)any_gpu;   // fftconv_conv_fft_data tells host and device kernels apart itself (FFTCONV_AUTO)
    plhs[0] = mxCreateCellMatrix(1, n);                                                      // :112
    std::vector<float*> out(n);
    const mwSize cdims[2] = {(mwSize)info.fft_h, (mwSize)info.fft_w};
    for (int k = 0; k < n; k++) {
        mxArray* m = mxCreateNumericArray(2, cdims, mxSINGLE_CLASS, mxREAL);                 // :277-281
        out[k] = static_cast<float*>(mxGetData(m));
        mxSetCell(plhs[0], k, m);
    }
    const int rc = fftconv_conv_fft_data(plan, n, kp.data(), kh.data(), kw.data(), kf.data(), threads, nthreads, out.data());
    release_views();
    if (own_plan) fftconv_plan_destroy(plan);   // (the message of a failed call stays: destroy does not touch it on success)
    if (rc != FFTCONV_OK) mexErrMsgIdAndTxt(errId, "%s", fftconv_last_error());
}
