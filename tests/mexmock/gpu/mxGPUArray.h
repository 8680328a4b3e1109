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
int mxInitGPU(void);
int mxIsGPUArray(const mxArray *a);   /* bool in MATLAB */
const mxGPUArray *mxGPUCreateFromMxArray(const mxArray *a);
mxClassID mxGPUGetClassID(const mxGPUArray *g);
mwSize mxGPUGetNumberOfDimensions(const mxGPUArray *g);
const mwSize *mxGPUGetDimensions(const mxGPUArray *g);
const void *mxGPUGetDataReadOnly(const mxGPUArray *g);
void mxGPUDestroyGPUArray(const mxGPUArray *g);

#ifdef __cplusplus
}
#endif
#endif
