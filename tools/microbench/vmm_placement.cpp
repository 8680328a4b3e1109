// vmm_placement.cpp -- is the output kernel's fast / slow state (26.2 / 27.5 us per map at cfg3,
// profiles/r02w_placement_two_states.txt) a property of the map buffer's PHYSICAL memory or of its VIRTUAL
// address?  The same physical allocation (hipMemCreate) is mapped at several virtual addresses, and several
// physical allocations at the same virtual address; the plan (and its intermediate) stays fixed.
// Build: hipcc -O2 vmm_placement.cpp -I../../include -L../../cuda-fft-convolution_amd -lfftconv -Wl,-rpath,'$ORIGIN/../../cuda-fft-convolution_amd' -o vmm_placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "fftconv.h"
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define FC(x) do { int r_ = (x); if (r_ != 0) { printf("%s: %d %s\n", #x, r_, fftconv_last_error()); exit(1); } } while (0)

static fftconv_plan* plan;
static float* kern;
static const int N = 64, KH = 127, KW = 127;

static double trial(float* out) {
    for (int i = 0; i < 14; i++) FC(fftconv_plan_convolve_packed(plan, N, kern, KH, KW, out));
    FC(fftconv_plan_synchronize(plan));
    fftconv_profile pr;
    FC(fftconv_plan_set_option(plan, "profile", 1));
    FC(fftconv_plan_get_profile(plan, &pr, 1));
    for (int i = 0; i < 8; i++) FC(fftconv_plan_convolve_packed(plan, N, kern, KH, KW, out));
    FC(fftconv_plan_synchronize(plan));
    FC(fftconv_plan_get_profile(plan, &pr, 1));
    FC(fftconv_plan_set_option(plan, "profile", 0));
    printf("    rows %.2f  cols %.2f us per map\n", pr.ms[1] / pr.units[1] * 1e3, pr.ms[2] / pr.units[2] * 1e3);
    return pr.ms[2] / pr.units[2] * 1e3;
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    CHECK(hipSetDevice(0));
    float* img;
    CHECK(hipMalloc(&img, (size_t)4096 * 4096 * 4));
    CHECK(hipMemset(img, 0, (size_t)4096 * 4096 * 4));
    CHECK(hipMalloc(&kern, (size_t)N * KH * KW * 4));
    CHECK(hipMemset(kern, 0, (size_t)N * KH * KW * 4));
    FC(fftconv_plan_create(&plan, 4096, 4096, 1, KH, KW, 0, nullptr));
    FC(fftconv_plan_set_image(plan, img, FFTCONV_DEVICE));

    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CHECK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    size_t bytes = (size_t)N * 4224 * 4224 * 4;
    bytes = (bytes + gran - 1) / gran * gran;
    printf("granularity %zu, map buffer %zu bytes\n", gran, bytes);
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;

    // virtual ranges: 2-MB aligned, 1-GB aligned, 8-GB aligned
    const size_t aligns[3] = {(size_t)2 << 20, (size_t)1 << 30, (size_t)8 << 30};
    void* va[3];
    for (int i = 0; i < 3; i++) {
        CHECK(hipMemAddressReserve(&va[i], bytes, aligns[i], nullptr, 0));
        printf("virtual range %d (alignment %zu MiB): %p\n", i, aligns[i] >> 20, va[i]);
    }
    std::vector<hipMemGenericAllocationHandle_t> keep;
    for (int h = 0; h < 6; h++) {
        hipMemGenericAllocationHandle_t handle;
        CHECK(hipMemCreate(&handle, bytes, &prop, 0));
        keep.push_back(handle);   // never released before the end: every handle is distinct physical memory
        printf("physical allocation %d\n", h);
        for (int i = 0; i < 3; i++) {
            CHECK(hipMemMap(va[i], bytes, 0, handle, 0));
            CHECK(hipMemSetAccess(va[i], bytes, &acc, 1));
            printf("  at virtual range %d:", i);
            trial(static_cast<float*>(va[i]));
            CHECK(hipMemUnmap(va[i], bytes));
        }
    }
    // plain hipMalloc buffers for comparison
    for (int h = 0; h < 4; h++) {
        float* o;
        CHECK(hipMalloc(&o, bytes));
        printf("hipMalloc buffer %d at %p:", h, (void*)o);
        trial(o);
    }
    return 0;
}
