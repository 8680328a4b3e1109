// cudaConvolutionFFT_mex.cpp -- MATLAB MEX gateway over libfftconv.so with the reference's
// positional signature (src/cudaConvolutionFFT.cu:15-22, demoCudaConvolutionFFT.m:124-129):
//
//   cvcell = cudaConvolutionFFT(data, maxKernelH, maxKernelW, kernelCell[, threadSize][, gpuId])
//
// Build on a host with MATLAB (mex.h is not present in the development image, so this file is
// not part of the default build):
//   mex -largeArrayDims -I<repo>/include cudaConvolutionFFT_mex.cpp -L<repo>/cuda-fft-convolution_amd -lfftconv
// gpuArray kernels (src/cudaConvolutionFFT.cu:224-238) would additionally need mxGPUArray.h and
// fftconv_plan_convolve(..., FFTCONV_DEVICE, ...); host arrays are handled here.
#include <vector>

#include "fftconv.h"
#include "mex.h"

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    const char* errId = FFTCONV_MEX_ERROR_ID;  // "cudaConvFFTData:InvalidInput" (reference :30)
    if (nrhs < 4 || nrhs > 6) mexErrMsgIdAndTxt(errId, "Wrong number of inputs");          // :45-46
    const mxArray* mxDATA = prhs[0];
    const mwSize nd = mxGetNumberOfDimensions(mxDATA);
    if (mxGetClassID(mxDATA) != mxSINGLE_CLASS || nd < 2 || nd > 3) mexErrMsgTxt("Invalid data input");  // :51-54 (F = 1 accepted)
    const mwSize* dd = mxGetDimensions(mxDATA);
    const int H = (int)dd[0], W = (int)dd[1], F = nd == 3 ? (int)dd[2] : 1;
    const int maxKH = (int)mxGetScalar(prhs[1]), maxKW = (int)mxGetScalar(prhs[2]);         // :58-59
    if (mxGetClassID(prhs[3]) != mxCELL_CLASS) mexErrMsgIdAndTxt(errId, "Kernel must be a cell array");  // :64-65
    const int n = (int)mxGetNumberOfElements(prhs[3]);
    const double* threads = nullptr;
    int nthreads = 0;
    if (nrhs > 4) { threads = (const double*)mxGetData(prhs[4]); nthreads = (int)mxGetNumberOfElements(prhs[4]); }  // :72-82
    const int gpu = nrhs > 5 ? (int)mxGetScalar(prhs[5]) : 0;                                // :85-89, 0-based

    std::vector<const float*> kp(n);
    std::vector<int> kh(n), kw(n), kf(n);
    for (int k = 0; k < n; k++) {
        const mxArray* c = mxGetCell(prhs[3], k);
        const mwSize knd = mxGetNumberOfDimensions(c);
        if (mxGetClassID(c) != mxSINGLE_CLASS || knd < 2 || knd > 3)
            mexErrMsgIdAndTxt(errId, "Kernels must be of type float and have features larger than 1");  // :210-211
        const mwSize* kd = mxGetDimensions(c);
        kp[k] = (const float*)mxGetData(c);
        kh[k] = (int)kd[0]; kw[k] = (int)kd[1]; kf[k] = knd == 3 ? (int)kd[2] : 1;
    }
    const int FFT_H = fftconv_fft_size16(H + maxKH - 1), FFT_W = fftconv_fft_size16(W + maxKW - 1);
    plhs[0] = mxCreateCellMatrix(1, n);                                                      // :202
    std::vector<float*> out(n);
    mwSize cdims[2] = {(mwSize)FFT_H, (mwSize)FFT_W};                                        // :198-200
    for (int k = 0; k < n; k++) {
        mxArray* m = mxCreateNumericArray(2, cdims, mxSINGLE_CLASS, mxREAL);                 // :284
        out[k] = (float*)mxGetData(m);
        mxSetCell(plhs[0], k, m);                                                            // :288
    }
    int rc = fftconv_convolution_fft((const float*)mxGetData(mxDATA), H, W, F, maxKH, maxKW, n, kp.data(), kh.data(),
                                     kw.data(), kf.data(), threads, nthreads, gpu, out.data(), nullptr, nullptr);
    if (rc != FFTCONV_OK) mexErrMsgIdAndTxt(errId, "%s", fftconv_last_error());
}
