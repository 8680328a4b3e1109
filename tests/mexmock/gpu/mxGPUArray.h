/* gpu/mxGPUArray.h -- TEST-ONLY miniature of the MathWorks GPU MEX API (the subset the gateways
 * use for gpuArray kernels, src/cudaConvolutionFFT.cu:224-238), beside the mex.h miniature of this
 * directory.  Not MATLAB's header; never installed or linked into the product. */
#ifndef FFTCONV_TEST_MXGPUARRAY_H
#define FFTCONV_TEST_MXGPUARRAY_H
#include "../mex.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mxGPUArray_tag mxGPUArray;
#define MX_GPU_SUCCESS 0
typedef enum { MX_GPU_DO_NOT_INITIALIZE = 0, MX_GPU_INITIALIZE_VALUES = 1 } mxGPUInitialize;
int mxInitGPU(void);
int mxIsGPUArray(const mxArray *a);   /* bool in MATLAB */
const mxGPUArray *mxGPUCreateFromMxArray(const mxArray *a);
mxClassID mxGPUGetClassID(const mxGPUArray *g);
mwSize mxGPUGetNumberOfDimensions(const mxGPUArray *g);
const mwSize *mxGPUGetDimensions(const mxGPUArray *g);
const void *mxGPUGetDataReadOnly(const mxGPUArray *g);
void mxGPUDestroyGPUArray(const mxGPUArray *g);
mxComplexity mxGPUGetComplexity(const mxGPUArray *g);
/* a new gpuArray in device memory; mxGPUGetData: its writable device pointer;
 * mxGPUCreateMxArrayOnGPU: the mxArray to return to MATLAB (the mxGPUArray is still to be destroyed) */
mxGPUArray *mxGPUCreateGPUArray(mwSize ndim, const mwSize *dims, mxClassID cls, mxComplexity cplx, mxGPUInitialize init);
void *mxGPUGetData(mxGPUArray *g);
mxArray *mxGPUCreateMxArrayOnGPU(const mxGPUArray *g);

#ifdef __cplusplus
}
#endif
#endif
