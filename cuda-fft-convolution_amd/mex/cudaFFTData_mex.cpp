// cudaFFTData_mex.cpp -- MATLAB MEX gateway, step 1 of the reference's two-step API
// (src/cudaFFTData.cu:18-160):
//
//   fftData = cudaFFTData(data, kernelH, kernelW[, gpuId])
//
// The reference returns the half spectrum as a complex single gpuArray, (FFT_H/2+1) x FFT_W x F
// (src/cudaFFTData.cu:90-103,150-158).  Where the MathWorks GPU header gpu/mxGPUArray.h is available
// this gateway does the same: the spectrum is exported in cuFFT's [f][FFT_W][FFT_H/2+1] order
// (fftconv_plan_export_spectrum) into a fresh gpuArray that MATLAB owns -- it can be inspected,
// edited, saved, and handed to cudaConvFFTData, exactly like the reference's.
// Alternative form (always available, and the fallback where the window has no transform of its own
// size): the spectrum stays inside an engine plan and `fftData` is an opaque uint64 handle to it,
// also accepted by cudaConvFFTData -- cheaper when the same image meets many kernel cells (no
// reordering per call): cudaFFTData(data, kH, kW, gpuId, 1).
// Extension: cudaFFTData(fftData) with a single uint64 argument releases a handle early; handles
// still alive when the MEX file is cleared are released by the mexAtExit hook.
#include <algorithm>
#include <cstdint>
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

namespace {
std::vector<fftconv_plan*> g_live;
void release_all() {
    for (fftconv_plan* p : g_live) fftconv_plan_destroy(p);
    g_live.clear();
}
}  // namespace

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    const char* errId = "parallel:gpu:mexGPUExample:InvalidInput";   // src/cudaFFTData.cu:28
    const char* errMsg = "Invalid input to MEX file.";               // src/cudaFFTData.cu:29
    (void)nlhs;
    mexAtExit(release_all);
    if (nrhs == 1 && mxGetClassID(prhs[0]) == mxUINT64_CLASS && mxGetNumberOfElements(prhs[0]) == 1) {
        fftconv_plan* p = reinterpret_cast<fftconv_plan*>((uintptr_t) * static_cast<const uint64_t*>(mxGetData(prhs[0])));
        auto it = std::find(g_live.begin(), g_live.end(), p);
        if (it == g_live.end() || !fftconv_plan_is_live(p)) mexErrMsgIdAndTxt(errId, errMsg);
        g_live.erase(it);
        fftconv_plan_destroy(p);
        return;
    }
    if (nrhs < 3 || nrhs > 5) mexErrMsgIdAndTxt(errId, errMsg);                              // :49-54 (nrhs != 3)
    const mxArray* mxDATA = prhs[0];
    const mwSize nd = mxGetNumberOfDimensions(mxDATA);
    if (mxGetClassID(mxDATA) != mxSINGLE_CLASS || nd < 2 || nd > 3) mexErrMsgIdAndTxt(errId, errMsg);   // F = 1 accepted
    const mwSize* dd = mxGetDimensions(mxDATA);
    const int H = (int)dd[0], W = (int)dd[1], F = nd == 3 ? (int)dd[2] : 1;
    const int kh = (int)mxGetScalar(prhs[1]), kw = (int)mxGetScalar(prhs[2]);                // :58-59
    const int gpu = nrhs > 3 ? (int)mxGetScalar(prhs[3]) : 0;
    const bool as_handle = nrhs > 4 && mxGetScalar(prhs[4]) != 0;
#if FFTCONV_MEX_GPU
    if (!as_handle) {   // the reference's form: a complex single gpuArray (src/cudaFFTData.cu:90-103,150)
        mxInitGPU();
        fftconv_plan_options opts = {};
        opts.struct_size = sizeof(opts);
        opts.exact_window = 1;                 // transform == the ceil16 window: the reference's own spectrum
        fftconv_plan* q = nullptr;
        int rc = fftconv_plan_create_ex(&q, H, W, F, kh, kw, gpu, nullptr, &opts);
        if (rc == FFTCONV_OK) {
            rc = fftconv_plan_set_image(q, static_cast<const float*>(mxGetData(mxDATA)), FFTCONV_HOST);
            fftconv_plan_info info;
            if (rc == FFTCONV_OK) rc = fftconv_plan_get_info(q, &info);
            if (rc == FFTCONV_OK) {
                const mwSize cdims[3] = {(mwSize)(info.fft_h / 2 + 1), (mwSize)info.fft_w, (mwSize)F};
                mxGPUArray* g = mxGPUCreateGPUArray(3, cdims, mxSINGLE_CLASS, mxCOMPLEX, MX_GPU_DO_NOT_INITIALIZE);
                rc = fftconv_plan_export_spectrum(q, static_cast<float*>(mxGPUGetData(g)), FFTCONV_DEVICE);
                if (rc == FFTCONV_OK) rc = fftconv_plan_synchronize(q);
                if (rc == FFTCONV_OK) plhs[0] = mxGPUCreateMxArrayOnGPU(g);
                mxGPUDestroyGPUArray(g);
            }
            fftconv_plan_destroy(q);
            if (rc != FFTCONV_OK) mexErrMsgIdAndTxt(FFTCONV_MEX_ERROR_ID, "%s", fftconv_last_error());
            return;
        }
        if (rc != FFTCONV_ERR_UNSUPPORTED_SIZE) mexErrMsgIdAndTxt(FFTCONV_MEX_ERROR_ID, "%s", fftconv_last_error());
        // the window itself has no supported transform: fall through to the handle form
    }
#else
    (void)as_handle;
#endif
    fftconv_plan* p = nullptr;
    if (fftconv_fft_data(static_cast<const float*>(mxGetData(mxDATA)), H, W, F, kh, kw, gpu, &p) != FFTCONV_OK)
        mexErrMsgIdAndTxt(FFTCONV_MEX_ERROR_ID, "%s", fftconv_last_error());
    g_live.push_back(p);
    plhs[0] = mxCreateNumericMatrix(1, 1, mxUINT64_CLASS, mxREAL);
    *static_cast<uint64_t*>(mxGetData(plhs[0])) = (uint64_t)(uintptr_t)p;
}
