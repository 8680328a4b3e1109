// mexmock.cpp -- TEST-ONLY implementation of the mex.h miniature next to it, plus a tiny C driver
// API so that the Python tests can build argument lists, call a gateway's mexFunction and read the
// results or the raised error.  Test infrastructure: never linked into the product.
#include "mex.h"
#include "gpu/mxGPUArray.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <string>
#include <vector>

struct mxArray_tag {
    mxClassID cls = mxUNKNOWN_CLASS;
    std::vector<mwSize> dims;
    std::vector<unsigned char> data;
    std::vector<mxArray*> cells;
    // a gpuArray: the data lives in device memory (dev), MATLAB reports an opaque class for the mxArray itself
    bool on_gpu = false;
    mxClassID gpu_cls = mxUNKNOWN_CLASS;
    mxComplexity gpu_cplx = mxREAL;
    void* dev = nullptr;
    bool owns_dev = false;   // allocated by mxGPUCreateGPUArray (freed with the array)
};
struct mxGPUArray_tag {
    const mxArray* src;
    mxArray* fresh;          // mxGPUCreateGPUArray: the array being built (handed over by mxGPUCreateMxArrayOnGPU)
    bool handed_over;
};
// device memory for gpuArrays the gateways create: the process's own HIP runtime (already loaded by libfftconv.so)
static void* dev_alloc(size_t bytes) {
    typedef int (*malloc_t)(void**, size_t);
    malloc_t f = (malloc_t)dlsym(RTLD_DEFAULT, "hipMalloc");
    void* p = nullptr;
    return (f && f(&p, bytes) == 0) ? p : nullptr;
}
static void dev_free(void* p) {
    typedef int (*free_t)(void*);
    free_t f = (free_t)dlsym(RTLD_DEFAULT, "hipFree");
    if (f && p) f(p);
}
int g_live_gpu_views = 0;

namespace {
struct MexError {
    std::string id, msg;
};
std::string g_err_id, g_err_msg;
std::vector<void (*)(void)> g_at_exit;
size_t elem_size(mxClassID c) { return c == mxSINGLE_CLASS ? 4 : (c == mxDOUBLE_CLASS || c == mxUINT64_CLASS) ? 8 : 0; }
size_t count(const mxArray* a) {
    size_t n = 1;
    for (mwSize d : a->dims) n *= d;
    return n;
}
}  // namespace

extern "C" {

mwSize mxGetNumberOfDimensions(const mxArray* a) { return a->dims.size(); }
const mwSize* mxGetDimensions(const mxArray* a) { return a->dims.data(); }
mxClassID mxGetClassID(const mxArray* a) { return a->on_gpu ? mxUNKNOWN_CLASS : a->cls; }
int mxInitGPU(void) { return MX_GPU_SUCCESS; }
int mxIsGPUArray(const mxArray* a) { return a && a->on_gpu ? 1 : 0; }
const mxGPUArray* mxGPUCreateFromMxArray(const mxArray* a) { g_live_gpu_views++; return new mxGPUArray_tag{a, nullptr, false}; }
mxComplexity mxGPUGetComplexity(const mxGPUArray* g) { return g->src->gpu_cplx; }
mxGPUArray* mxGPUCreateGPUArray(mwSize ndim, const mwSize* dims, mxClassID cls, mxComplexity cplx, mxGPUInitialize) {
    mxArray* a = new mxArray_tag();
    a->dims.assign(dims, dims + ndim);
    while (a->dims.size() < 2) a->dims.push_back(1);
    a->on_gpu = true; a->gpu_cls = cls; a->gpu_cplx = cplx; a->owns_dev = true;
    size_t n = 1;
    for (mwSize d : a->dims) n *= d;
    a->dev = dev_alloc(n * (cls == mxSINGLE_CLASS ? 4 : 8) * (cplx == mxCOMPLEX ? 2 : 1));
    g_live_gpu_views++;
    return new mxGPUArray_tag{a, a, false};
}
void* mxGPUGetData(mxGPUArray* g) { return g->src->dev; }
mxArray* mxGPUCreateMxArrayOnGPU(const mxGPUArray* g) { const_cast<mxGPUArray*>(g)->handed_over = true; return g->fresh; }
mxClassID mxGPUGetClassID(const mxGPUArray* g) { return g->src->gpu_cls; }
mwSize mxGPUGetNumberOfDimensions(const mxGPUArray* g) { return g->src->dims.size(); }
const mwSize* mxGPUGetDimensions(const mxGPUArray* g) { return g->src->dims.data(); }
const void* mxGPUGetDataReadOnly(const mxGPUArray* g) { return g->src->dev; }
void mxGPUDestroyGPUArray(const mxGPUArray* g) {
    g_live_gpu_views--;
    if (g->fresh && !g->handed_over) mxDestroyArray(g->fresh);   // never returned to MATLAB: goes with its view
    delete g;
}
size_t mxGetNumberOfElements(const mxArray* a) { return count(a); }
void* mxGetData(const mxArray* a) { return const_cast<unsigned char*>(a->data.data()); }
double mxGetScalar(const mxArray* a) {
    if (a->cls == mxDOUBLE_CLASS) return *reinterpret_cast<const double*>(a->data.data());
    if (a->cls == mxSINGLE_CLASS) return *reinterpret_cast<const float*>(a->data.data());
    if (a->cls == mxUINT64_CLASS) return (double)*reinterpret_cast<const uint64_t*>(a->data.data());
    return 0.0;
}
mxArray* mxGetCell(const mxArray* c, mwIndex i) { return i < c->cells.size() ? c->cells[i] : nullptr; }
void mxSetCell(mxArray* c, mwIndex i, mxArray* v) {
    if (i < c->cells.size()) c->cells[i] = v;
}
mxArray* mxCreateCellMatrix(mwSize m, mwSize n) {
    mxArray* a = new mxArray_tag();
    a->cls = mxCELL_CLASS;
    a->dims = {m, n};
    a->cells.assign(m * n, nullptr);
    return a;
}
mxArray* mxCreateNumericArray(mwSize ndim, const mwSize* dims, mxClassID cls, mxComplexity) {
    mxArray* a = new mxArray_tag();
    a->cls = cls;
    a->dims.assign(dims, dims + ndim);
    while (a->dims.size() < 2) a->dims.push_back(1);
    a->data.assign(count(a) * elem_size(cls), 0);   // MATLAB zero-fills too
    return a;
}
mxArray* mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID cls, mxComplexity c) {
    const mwSize d[2] = {m, n};
    return mxCreateNumericArray(2, d, cls, c);
}
void mxDestroyArray(mxArray* a) {
    if (!a) return;
    if (a->owns_dev) dev_free(a->dev);
    for (mxArray* c : a->cells) mxDestroyArray(c);
    delete a;
}
int mexAtExit(void (*fn)(void)) {
    for (auto f : g_at_exit)
        if (f == fn) return 0;
    g_at_exit.push_back(fn);
    return 0;
}
void mexErrMsgIdAndTxt(const char* id, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    throw MexError{id ? id : "", buf};   // MATLAB long-jumps out of the MEX function; so does this
}
void mexErrMsgTxt(const char* msg) { throw MexError{"", msg ? msg : ""}; }

// ---- driver API for the tests -------------------------------------------------------------------
mxArray* mock_new_numeric(int cls, int ndim, const uint64_t* dims, const void* data) {
    std::vector<mwSize> d(dims, dims + ndim);
    mxArray* a = mxCreateNumericArray(ndim, d.data(), (mxClassID)cls, mxREAL);
    if (data) memcpy(a->data.data(), data, a->data.size());
    return a;
}
// a gpuArray of class `cls` whose elements live at device pointer `dev` (owned by the test)
mxArray* mock_new_gpu_ex(int cls, int ndim, const uint64_t* dims, void* dev, int is_complex);
mxArray* mock_new_gpu(int cls, int ndim, const uint64_t* dims, void* dev) { return mock_new_gpu_ex(cls, ndim, dims, dev, 0); }
mxArray* mock_new_gpu_ex(int cls, int ndim, const uint64_t* dims, void* dev, int is_complex) {
    mxArray* a = new mxArray_tag();
    a->gpu_cplx = is_complex ? mxCOMPLEX : mxREAL;
    a->dims.assign(dims, dims + ndim);
    while (a->dims.size() < 2) a->dims.push_back(1);
    a->on_gpu = true;
    a->gpu_cls = (mxClassID)cls;
    a->dev = dev;
    return a;
}
int mock_live_gpu_views(void) { return g_live_gpu_views; }
int mock_is_gpu(const mxArray* a) { return a->on_gpu ? 1 : 0; }
int mock_gpu_is_complex(const mxArray* a) { return a->gpu_cplx == mxCOMPLEX ? 1 : 0; }
void* mock_gpu_ptr(const mxArray* a) { return a->dev; }   // mxGPUCreateFromMxArray not yet destroyed
mxArray* mock_new_cell(int n) { return mxCreateCellMatrix(1, n); }
void mock_set_cell(mxArray* c, int i, mxArray* v) { mxSetCell(c, i, v); }
mxArray* mock_get_cell(mxArray* c, int i) { return mxGetCell(c, i); }
int mock_class(const mxArray* a) { return (int)a->cls; }
int mock_ndim(const mxArray* a) { return (int)a->dims.size(); }
uint64_t mock_dim(const mxArray* a, int i) { return a->dims[i]; }
void* mock_data(mxArray* a) { return a->data.data(); }
void mock_free(mxArray* a) { mxDestroyArray(a); }
const char* mock_error_id(void) { return g_err_id.c_str(); }
const char* mock_error_msg(void) { return g_err_msg.c_str(); }
// calls a gateway; 0 = returned normally, 1 = raised an error (id/message kept)
int mock_call(void (*fn)(int, mxArray**, int, const mxArray**), int nlhs, mxArray** plhs, int nrhs, const mxArray** prhs) {
    g_err_id.clear();
    g_err_msg.clear();
    try {
        fn(nlhs, plhs, nrhs, prhs);
    } catch (const MexError& e) {
        g_err_id = e.id;
        g_err_msg = e.msg;
        return 1;
    }
    return 0;
}
// what MATLAB does at `clear mex` / exit
void mock_run_at_exit(void) {
    for (auto f : g_at_exit) f();
    g_at_exit.clear();
}

}  // extern "C"
