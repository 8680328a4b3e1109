// fftconv_multi.cpp -- several GPUs from one process, on top of the plan API of this library:
// what the reference's src/cudaConvFFTDataStreams.cu set out to do (one ConvPlan per GPU and
// stream :273-328, image spectrum copied GPU 0 -> GPU g :279-289, kernels dealt over the plans
// :338-447, barrier :452-468).  Host code only; the per-device work is the ordinary plan path.
#include <hip/hip_runtime.h>

#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fftconv.h"
#include "api_internal.hpp"

using fc::api_fail;

struct fftconv_multi {
    int H = 0, W = 0, F = 0;
    std::vector<int> dev;
    std::vector<hipStream_t> stream;
    std::vector<fftconv_plan*> plan;
    hipEvent_t spectrum_ready = nullptr;   // on dev[0]: the image spectrum is complete
    bool have_image = false;
};

namespace {

void shard(int n, int index, int parts, int* first, int* count) {
    const int base = n / parts, extra = n % parts;
    *first = index * base + (index < extra ? index : extra);
    *count = base + (index < extra ? 1 : 0);
}

void release(fftconv_multi* m) {
    for (size_t g = 0; g < m->plan.size(); g++)
        if (m->plan[g]) fftconv_plan_destroy(m->plan[g]);
    for (size_t g = 0; g < m->stream.size(); g++)
        if (m->stream[g]) { (void)hipSetDevice(m->dev[g]); (void)hipStreamDestroy(m->stream[g]); }
    if (m->spectrum_ready) { (void)hipSetDevice(m->dev[0]); (void)hipEventDestroy(m->spectrum_ready); }
    delete m;
}

}  // namespace

extern "C" {

int fftconv_multi_create(fftconv_multi** multi, int data_h, int data_w, int feature_dim, int max_kernel_h, int max_kernel_w,
                         const int* devices, int n_devices, const fftconv_plan_options* options) {
    if (!multi) return api_fail(FFTCONV_ERR_INVALID_ARG, "multi is NULL");
    *multi = nullptr;
    if (!devices || n_devices < 1) return api_fail(FFTCONV_ERR_INVALID_ARG, "no devices listed");
    int ndev = 0;
    if (int rc = fftconv_device_count(&ndev)) return rc;
    for (int g = 0; g < n_devices; g++)
        if (devices[g] < 0 || devices[g] >= ndev)
            return api_fail(FFTCONV_ERR_NO_DEVICE, "device %d out of range (%d devices)", devices[g], ndev);
    fftconv_multi* m = new (std::nothrow) fftconv_multi();
    if (!m) return api_fail(FFTCONV_ERR_ALLOC, "out of host memory");
    m->H = data_h; m->W = data_w; m->F = feature_dim;
    m->dev.assign(devices, devices + n_devices);
    m->stream.assign(n_devices, nullptr);
    m->plan.assign(n_devices, nullptr);
    int rc = 0;
    for (int g = 0; g < n_devices && !rc; g++) {
        hipError_t e = hipSetDevice(m->dev[g]);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&m->stream[g], hipStreamNonBlocking);
        if (e != hipSuccess) { rc = api_fail(FFTCONV_ERR_HIP, "stream setup on device %d failed: %s", m->dev[g], hipGetErrorString(e)); break; }
        rc = fftconv_plan_create_ex(&m->plan[g], data_h, data_w, feature_dim, max_kernel_h, max_kernel_w, m->dev[g], m->stream[g], options);
        if (!rc && g > 0 && m->dev[g] != m->dev[0]) {   // xGMI peer access in both directions (already enabled is fine)
            int can = 0;
            (void)hipDeviceCanAccessPeer(&can, m->dev[g], m->dev[0]);
            if (can) {
                e = hipDeviceEnablePeerAccess(m->dev[0], 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
                (void)hipGetLastError();
            }
        }
    }
    if (!rc) {
        hipError_t e = hipSetDevice(m->dev[0]);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&m->spectrum_ready, hipEventDisableTiming);
        if (e != hipSuccess) rc = api_fail(FFTCONV_ERR_HIP, "event setup failed: %s", hipGetErrorString(e));
    }
    if (rc) {
        const std::string keep = fc::api_last_error();
        release(m);
        fc::api_set_last_error(keep);
        return rc;
    }
    *multi = m;
    return 0;
}

int fftconv_multi_destroy(fftconv_multi* multi) {
    if (multi) release(multi);
    return 0;
}

int fftconv_multi_size(const fftconv_multi* multi) { return multi ? (int)multi->plan.size() : 0; }

int fftconv_multi_plan(fftconv_multi* multi, int index, fftconv_plan** plan, int* device) {
    if (!multi || index < 0 || index >= (int)multi->plan.size()) return api_fail(FFTCONV_ERR_INVALID_ARG, "bad multi handle or index");
    if (plan) *plan = multi->plan[index];
    if (device) *device = multi->dev[index];
    return 0;
}

int fftconv_multi_shard(const fftconv_multi* multi, int n_kernel, int index, int* first, int* count) {
    if (!multi || index < 0 || index >= (int)multi->plan.size() || n_kernel < 0 || !first || !count)
        return api_fail(FFTCONV_ERR_INVALID_ARG, "bad shard request");
    shard(n_kernel, index, (int)multi->plan.size(), first, count);
    return 0;
}

namespace {
// the spectrum of plan[0] (complete on stream[0]) to every other device: GPU 0 -> GPU g (the reference's
// cudaMemcpyPeerAsync, src/cudaConvFFTDataStreams.cu:282-287), each copy on the destination's stream
// behind an event, so the next convolve on that stream is ordered behind its copy
int distribute_spectrum(fftconv_multi* m) {
    void* src = nullptr;
    size_t bytes = 0;
    if (int rc = fftconv_plan_spectrum(m->plan[0], &src, &bytes)) return rc;
    hipError_t e = hipSetDevice(m->dev[0]);
    if (e == hipSuccess) e = hipEventRecord(m->spectrum_ready, m->stream[0]);
    if (e != hipSuccess) return api_fail(FFTCONV_ERR_HIP, "event record failed: %s", hipGetErrorString(e));
    for (size_t g = 1; g < m->plan.size(); g++) {
        void* dst = nullptr;
        if (int rc = fftconv_plan_spectrum(m->plan[g], &dst, nullptr)) return rc;
        e = hipSetDevice(m->dev[g]);
        if (e == hipSuccess) e = hipStreamWaitEvent(m->stream[g], m->spectrum_ready, 0);
        if (e == hipSuccess)
            e = (m->dev[g] == m->dev[0]) ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, m->stream[g])
                                         : hipMemcpyPeerAsync(dst, m->dev[g], src, m->dev[0], bytes, m->stream[g]);
        if (e != hipSuccess) return api_fail(FFTCONV_ERR_HIP, "spectrum copy to device %d failed: %s", m->dev[g], hipGetErrorString(e));
        if (int rc = fftconv_plan_mark_spectrum_valid(m->plan[g])) return rc;
    }
    m->have_image = true;
    return 0;
}
}  // namespace

int fftconv_multi_set_image(fftconv_multi* multi, const float* data, int location) {
    if (!multi || !data) return api_fail(FFTCONV_ERR_INVALID_ARG, "Invalid data input");
    multi->have_image = false;
    if (int rc = fftconv_plan_set_image(multi->plan[0], data, location)) return rc;
    return distribute_spectrum(multi);
}

int fftconv_multi_import_spectrum(fftconv_multi* multi, const float* spectrum, int location) {
    if (!multi || !spectrum) return api_fail(FFTCONV_ERR_INVALID_ARG, "Invalid data input");
    multi->have_image = false;
    if (int rc = fftconv_plan_import_spectrum(multi->plan[0], spectrum, location)) return rc;
    return distribute_spectrum(multi);
}

int fftconv_multi_convolve(fftconv_multi* multi, int n_kernel, const float* const* kernels, const int* kernel_h,
                           const int* kernel_w, int kernel_location, float* const* out, int out_location) {
    if (!multi || n_kernel < 0) return api_fail(FFTCONV_ERR_INVALID_ARG, "Wrong number of inputs");
    if (n_kernel == 0) return 0;
    if (!kernels || !kernel_h || !kernel_w || !out) return api_fail(FFTCONV_ERR_INVALID_ARG, "Kernel must be a cell array");
    fftconv_multi* m = multi;
    if (!m->have_image) return api_fail(FFTCONV_ERR_NO_IMAGE, "no image spectrum: call fftconv_multi_set_image first");
    const int parts = (int)m->plan.size();
    std::vector<int> rcs(parts, 0);
    std::vector<std::string> msgs(parts);
    auto work = [&](int g) {
        int first = 0, count = 0;
        shard(n_kernel, g, parts, &first, &count);
        if (count == 0) return;
        int rc = fftconv_plan_convolve(m->plan[g], count, kernels + first, kernel_h + first, kernel_w + first, kernel_location,
                                       out + first, out_location);
        if (!rc) rc = fftconv_plan_synchronize(m->plan[g]);   // the barrier of src/cudaConvFFTDataStreams.cu:452-468
        rcs[g] = rc;
        if (rc) msgs[g] = fc::api_last_error();
    };
    std::vector<std::thread> th;
    for (int g = 1; g < parts; g++) th.emplace_back(work, g);
    work(0);
    for (std::thread& t : th) t.join();
    for (int g = 0; g < parts; g++)
        if (rcs[g]) { fc::api_set_last_error(msgs[g]); return rcs[g]; }
    return 0;
}

int fftconv_convolution_fft_multi(const float* data, int data_h, int data_w, int feature_dim, int max_kernel_h, int max_kernel_w,
                                  int n_kernel, const float* const* kernels, const int* kernel_h, const int* kernel_w,
                                  const int* kernel_f, const int* devices, int n_devices, float* const* out, int* fft_h, int* fft_w) {
    if (!data || data_h < 1 || data_w < 1 || feature_dim < 1) return api_fail(FFTCONV_ERR_INVALID_ARG, "Invalid data input");
    if (n_kernel < 0 || (n_kernel > 0 && (!kernels || !kernel_h || !kernel_w || !out)))
        return api_fail(FFTCONV_ERR_INVALID_ARG, "Kernel must be a cell array");
    if (kernel_f)
        for (int k = 0; k < n_kernel; k++)
            if (kernel_f[k] != feature_dim)   // src/cudaConvolutionFFT.cu:242
                return api_fail(FFTCONV_ERR_KERNEL_SHAPE,
                                "Kernel and Data must have the same number of features and kernel size should be smaller than data size");
    if (fft_h) *fft_h = fftconv_fft_size16(data_h + max_kernel_h - 1);
    if (fft_w) *fft_w = fftconv_fft_size16(data_w + max_kernel_w - 1);
    fftconv_multi* m = nullptr;
    if (int rc = fftconv_multi_create(&m, data_h, data_w, feature_dim, max_kernel_h, max_kernel_w, devices, n_devices, nullptr)) return rc;
    int rc = fftconv_multi_set_image(m, data, FFTCONV_HOST);
    if (!rc) rc = fftconv_multi_convolve(m, n_kernel, kernels, kernel_h, kernel_w, FFTCONV_HOST, out, FFTCONV_HOST);
    const std::string keep = fc::api_last_error();
    fftconv_multi_destroy(m);
    if (rc) fc::api_set_last_error(keep);
    return rc;
}

}  // extern "C"
