/* mex.h -- TEST-ONLY miniature of MATLAB's MEX C API (the subset our gateways under
 * cuda-fft-convolution_amd/mex/ use), so that the gateways can be compiled and driven from the
 * test-suite on machines without MATLAB.  It is NOT MATLAB's header and is never installed or
 * linked into the product; a real build uses the mex.h of the MATLAB installation
 * (cuda-fft-convolution_amd/mex/compile_hip.m). */
#ifndef FFTCONV_TEST_MEX_H
#define FFTCONV_TEST_MEX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef size_t mwSize;
typedef size_t mwIndex;
typedef struct mxArray_tag mxArray;
typedef enum { mxUNKNOWN_CLASS = 0, mxCELL_CLASS = 1, mxDOUBLE_CLASS = 6, mxSINGLE_CLASS = 7, mxUINT64_CLASS = 13 } mxClassID;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;

mwSize mxGetNumberOfDimensions(const mxArray *a);
const mwSize *mxGetDimensions(const mxArray *a);
mxClassID mxGetClassID(const mxArray *a);
size_t mxGetNumberOfElements(const mxArray *a);
void *mxGetData(const mxArray *a);
double mxGetScalar(const mxArray *a);
mxArray *mxGetCell(const mxArray *cell, mwIndex i);
void mxSetCell(mxArray *cell, mwIndex i, mxArray *value);
mxArray *mxCreateCellMatrix(mwSize m, mwSize n);
mxArray *mxCreateNumericArray(mwSize ndim, const mwSize *dims, mxClassID cls, mxComplexity cplx);
mxArray *mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID cls, mxComplexity cplx);
void mxDestroyArray(mxArray *a);
int mexAtExit(void (*fn)(void));
void mexErrMsgIdAndTxt(const char *id, const char *fmt, ...);
void mexErrMsgTxt(const char *msg);

/* every gateway defines this */
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]);

#ifdef __cplusplus
}
#endif
#endif
