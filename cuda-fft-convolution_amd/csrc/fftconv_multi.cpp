// fftconv_multi.cpp -- several GPUs from one process, on top of the plan API of this library:
// what the reference's src/cudaConvFFTDataStreams.cu set out to do (one ConvPlan per GPU and
// stream :273-328, image spectrum copied GPU 0 -> GPU g :279-289, kernels dealt over the plans
// :338-447, barrier :452-468).  Host code only; the per-device work is the ordinary plan path.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <mutex>
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
    std::vector<char> peer_direct;         // per position: 1 = same device as devices[0] or peer access enabled
    long opt_transport = 0;                // "spectrum_transport": 0 peer copies from devices[0], 1 one RCCL broadcast
    long opt_verbose = 0;
    std::vector<void*> comms;              // RCCL communicators, one per position (created on first use)
    bool rccl_failed = false;              // set-up was tried and refused: peer copies from then on
    int last_transport = 0;                // what the last distribution really used
};

namespace {

// RCCL is loaded on first use (dlopen): libfftconv.so does not depend on it, and a process that never asks for the
// broadcast transport never maps it.  Only the five entry points the single-process broadcast needs.
struct RcclApi {
    void* handle = nullptr;
    int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
    int (*CommDestroy)(void* comm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Broadcast)(const void* send, void* recv, size_t count, int datatype, int root, void* comm, hipStream_t stream) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
};

RcclApi& rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // An RCCL the process has ALREADY mapped comes first (a PyTorch process has its own bundled librccl in: loading the
        // system's copy beside it would put a second, different RCCL runtime into the process); only then the search path.
        const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* name : names) {
            api.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
            if (api.handle) break;
        }
        if (!api.handle && dlsym(RTLD_DEFAULT, "ncclCommInitAll")) api.handle = dlopen(nullptr, RTLD_NOW);   // mapped under another name
        for (const char* name : names) {
            if (api.handle) break;
            api.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        }
        if (!api.handle) return;
        auto sym = [&](const char* n) { return dlsym(api.handle, n); };
        api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(sym("ncclCommInitAll"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
        api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
        api.Broadcast = reinterpret_cast<decltype(api.Broadcast)>(sym("ncclBroadcast"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
        api.ok = api.CommInitAll && api.CommDestroy && api.GroupStart && api.GroupEnd && api.Broadcast;
        // the prototypes and the ncclInt8 = 0 used here are those of NCCL / RCCL 2.x: refuse anything else
        auto get_version = reinterpret_cast<int (*)(int*)>(sym("ncclGetVersion"));
        int version = 0;
        if (!get_version || get_version(&version) != 0 || version < 20000 || version >= 30000) api.ok = false;
    });
    return api;
}
constexpr int kNcclChar = 0;   // ncclInt8 / ncclChar of rccl.h

void shard(int n, int index, int parts, int* first, int* count) {
    const int base = n / parts, extra = n % parts;
    *first = index * base + (index < extra ? index : extra);
    *count = base + (index < extra ? 1 : 0);
}

void release(fftconv_multi* m) {
    for (size_t g = 0; g < m->comms.size(); g++)
        if (m->comms[g]) { (void)hipSetDevice(m->dev[g]); (void)rccl().CommDestroy(m->comms[g]); }
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
    m->peer_direct.assign(n_devices, 1);
    std::string peer_warning;
    int rc = 0;
    for (int g = 0; g < n_devices && !rc; g++) {
        hipError_t e = hipSetDevice(m->dev[g]);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&m->stream[g], hipStreamNonBlocking);
        if (e != hipSuccess) { rc = api_fail(FFTCONV_ERR_HIP, "stream setup on device %d failed: %s", m->dev[g], hipGetErrorString(e)); break; }
        rc = fftconv_plan_create_ex(&m->plan[g], data_h, data_w, feature_dim, max_kernel_h, max_kernel_w, m->dev[g], m->stream[g], options);
        if (!rc && g > 0 && m->dev[g] != m->dev[0]) {
            // the spectrum copy runs on the destination's stream and reads devices[0]'s memory: device g needs access to
            // devices[0] (this direction only; already enabled is fine).  Without it hipMemcpyPeerAsync still works but
            // stages through the host: not an error, but the caller can see it (fftconv_multi_get_option "peer_direct",
            // and the message fftconv_last_error() holds after a successful create).
            int can = 0;
            e = hipDeviceCanAccessPeer(&can, m->dev[g], m->dev[0]);
            if (e != hipSuccess) { (void)hipGetLastError(); can = 0; }
            if (can) {
                e = hipDeviceEnablePeerAccess(m->dev[0], 0);
                if (e == hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); e = hipSuccess; }
            }
            if (!can || e != hipSuccess) {
                m->peer_direct[g] = 0;
                char buf[200];
                snprintf(buf, sizeof(buf), "%sdevice %d has no direct access to device %d (%s): spectrum copies to it are staged",
                         peer_warning.empty() ? "warning: " : "; ", m->dev[g], m->dev[0], can ? hipGetErrorString(e) : "hipDeviceCanAccessPeer says no");
                peer_warning += buf;
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
    fc::api_set_last_error(peer_warning);   // "" when every destination has direct access
    return 0;
}

int fftconv_multi_set_option(fftconv_multi* multi, const char* name, long value) {
    if (!multi || !name) return api_fail(FFTCONV_ERR_INVALID_ARG, "NULL argument");
    if (!strcmp(name, "spectrum_transport")) {
        if (value != 0 && value != 1) return api_fail(FFTCONV_ERR_INVALID_ARG, "spectrum_transport is 0 (peer copies) or 1 (RCCL broadcast)");
        multi->opt_transport = value;
        return 0;
    }
    if (!strcmp(name, "verbose")) {
        multi->opt_verbose = value != 0;
        for (fftconv_plan* p : multi->plan)
            if (int rc = fftconv_plan_set_option(p, "verbose", value)) return rc;
        return 0;
    }
    return api_fail(FFTCONV_ERR_INVALID_ARG, "unknown option '%s'", name);
}

int fftconv_multi_get_option(const fftconv_multi* multi, const char* name, long* value) {
    if (!multi || !name || !value) return api_fail(FFTCONV_ERR_INVALID_ARG, "NULL argument");
    if (!strcmp(name, "spectrum_transport")) { *value = multi->opt_transport; return 0; }
    if (!strcmp(name, "transport_used")) { *value = multi->last_transport; return 0; }
    if (!strcmp(name, "verbose")) { *value = multi->opt_verbose; return 0; }
    if (!strcmp(name, "peer_direct")) {      // destinations (positions > 0) that devices[0]'s memory reaches without staging
        long n = 0;
        for (size_t g = 1; g < multi->peer_direct.size(); g++) n += multi->peer_direct[g] ? 1 : 0;
        *value = n;
        return 0;
    }
    return api_fail(FFTCONV_ERR_INVALID_ARG, "unknown option '%s'", name);
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
// north_star's form of the same step: ONE collective, ncclBroadcast of the spectrum buffer from devices[0] over the
// communicator of the listed devices (RCCL over xGMI), each rank's part on that device's stream.  Returns 1 when
// the broadcast was queued, 0 when RCCL is not usable here (the caller falls back to peer copies; the reason is
// kept as the thread's last-error text), < 0 on a hard error.
int broadcast_spectrum_rccl(fftconv_multi* m, void* src, size_t bytes) {
    if (m->rccl_failed) return 0;
    RcclApi& api = rccl();
    const int n = (int)m->plan.size();
    auto refuse = [&](const std::string& why) {
        m->rccl_failed = true;
        fc::api_set_last_error("warning: RCCL broadcast not used (" + why + "): peer copies instead");
        if (m->opt_verbose) fprintf(stderr, "fftconv_multi: RCCL broadcast not used (%s): peer copies instead\n", why.c_str());
        return 0;
    };
    if (!api.ok) return refuse("librccl.so could not be loaded");
    for (int a = 0; a < n; a++)
        for (int b = a + 1; b < n; b++)
            if (m->dev[a] == m->dev[b]) return refuse("a device is listed twice");
    if (m->comms.empty()) {
        m->comms.assign(n, nullptr);
        const int r = api.CommInitAll(m->comms.data(), n, m->dev.data());
        if (r != 0) {
            m->comms.clear();
            return refuse(std::string("ncclCommInitAll: ") + (api.GetErrorString ? api.GetErrorString(r) : "error"));
        }
    }
    std::vector<void*> dst(n, nullptr);
    dst[0] = src;
    for (int g = 1; g < n; g++)
        if (int rc = fftconv_plan_spectrum(m->plan[g], &dst[g], nullptr)) return rc;
    int r = api.GroupStart();
    int bad_device = -1;
    for (int g = 0; g < n && r == 0 && bad_device < 0; g++) {
        if (hipSetDevice(m->dev[g]) != hipSuccess) { bad_device = m->dev[g]; break; }
        r = api.Broadcast(src, dst[g], bytes, kNcclChar, 0, m->comms[g], m->stream[g]);
    }
    const int r2 = api.GroupEnd();      // always closed, whatever happened inside
    if (bad_device >= 0) return api_fail(FFTCONV_ERR_HIP, "hipSetDevice(%d) failed", bad_device);
    if (r == 0) r = r2;
    if (r != 0) return api_fail(FFTCONV_ERR_HIP, "ncclBroadcast of the image spectrum failed: %s", api.GetErrorString ? api.GetErrorString(r) : "error");
    for (int g = 1; g < n; g++)
        if (int rc = fftconv_plan_mark_spectrum_valid(m->plan[g])) return rc;
    return 1;
}

int distribute_spectrum(fftconv_multi* m) {
    void* src = nullptr;
    size_t bytes = 0;
    if (int rc = fftconv_plan_spectrum(m->plan[0], &src, &bytes)) return rc;
    if (m->opt_transport == 1 && m->plan.size() > 1) {
        const int r = broadcast_spectrum_rccl(m, src, bytes);
        if (r < 0) return r;
        if (r == 1) {
            m->last_transport = 1;
            m->have_image = true;
            if (m->opt_verbose) fprintf(stderr, "fftconv_multi: image spectrum (%zu bytes) broadcast over RCCL to %zu devices\n", bytes, m->plan.size() - 1);
            return 0;
        }
    }
    m->last_transport = 0;
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
        if (m->opt_verbose)
            fprintf(stderr, "fftconv_multi: image spectrum (%zu bytes) copied device %d -> device %d (%s)\n", bytes, m->dev[0], m->dev[g],
                    m->dev[g] == m->dev[0] ? "same device" : m->peer_direct[g] ? "peer access" : "staged");
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
