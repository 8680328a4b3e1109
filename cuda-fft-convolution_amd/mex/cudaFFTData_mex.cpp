// cudaFFTData_mex.cpp -- MATLAB MEX gateway, step 1 of the reference's two-step API
// (src/cudaFFTData.cu:18-160):
//
//   fftData = cudaFFTData(data, kernelH, kernelW[, gpuId])
//
// The reference returns the half spectrum as a complex gpuArray (CFFT_H x FFT_W x F,
// src/cudaFFTData.cu:150-158).  Here the spectrum stays inside an engine plan (its layout is the
// engine's business) and `fftData` is an opaque uint64 handle to it, accepted by cudaConvFFTData.
// Extension: cudaFFTData(fftData) with a single uint64 argument releases the handle early; handles
// still alive when the MEX file is cleared are released by the mexAtExit hook.
#include <algorithm>
#include <cstdint>
#include <vector>

#include "fftconv.h"
#include "mex.h"

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
    if (nrhs < 3 || nrhs > 4) mexErrMsgIdAndTxt(errId, errMsg);                              // :49-54 (nrhs != 3)
    const mxArray* mxDATA = prhs[0];
    const mwSize nd = mxGetNumberOfDimensions(mxDATA);
    if (mxGetClassID(mxDATA) != mxSINGLE_CLASS || nd < 2 || nd > 3) mexErrMsgIdAndTxt(errId, errMsg);   // F = 1 accepted
    const mwSize* dd = mxGetDimensions(mxDATA);
    const int H = (int)dd[0], W = (int)dd[1], F = nd == 3 ? (int)dd[2] : 1;
    const int kh = (int)mxGetScalar(prhs[1]), kw = (int)mxGetScalar(prhs[2]);                // :58-59
    const int gpu = nrhs > 3 ? (int)mxGetScalar(prhs[3]) : 0;
    fftconv_plan* p = nullptr;
    if (fftconv_fft_data(static_cast<const float*>(mxGetData(mxDATA)), H, W, F, kh, kw, gpu, &p) != FFTCONV_OK)
        mexErrMsgIdAndTxt(FFTCONV_MEX_ERROR_ID, "%s", fftconv_last_error());
    g_live.push_back(p);
    plhs[0] = mxCreateNumericMatrix(1, 1, mxUINT64_CLASS, mxREAL);
    *static_cast<uint64_t*>(mxGetData(plhs[0])) = (uint64_t)(uintptr_t)p;
}
