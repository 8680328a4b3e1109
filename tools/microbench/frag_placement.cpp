// frag_placement.cpp -- follow-up of vmm_placement.cpp: does churn in the device allocator (many allocations of
// odd sizes, half of them freed) decide whether fresh buffers put the output kernel in its fast or slow state?
// Build: hipcc -O2 frag_placement.cpp -I../../include -L../../cuda-fft-convolution_amd -lfftconv -Wl,-rpath,'$ORIGIN/../../cuda-fft-convolution_amd' -o frag_placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "fftconv.h"
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define FC(x) do { int r_ = (x); if (r_ != 0) { printf("%s: %d %s\n", #x, r_, fftconv_last_error()); exit(1); } } while (0)
static float* kern; static float* img;
static const int N = 64, KH = 127, KW = 127;
static const size_t OUT_BYTES = (size_t)N * 4224 * 4224 * 4;

static fftconv_plan* new_plan() {
    fftconv_plan* p;
    FC(fftconv_plan_create(&p, 4096, 4096, 1, KH, KW, 0, nullptr));
    FC(fftconv_plan_set_image(p, img, FFTCONV_DEVICE));
    return p;
}
static void trial(const char* tag, fftconv_plan* plan, float* out) {
    for (int i = 0; i < 14; i++) FC(fftconv_plan_convolve_packed(plan, N, kern, KH, KW, out));
    FC(fftconv_plan_synchronize(plan));
    fftconv_profile pr;
    FC(fftconv_plan_set_option(plan, "profile", 1));
    FC(fftconv_plan_get_profile(plan, &pr, 1));
    for (int i = 0; i < 8; i++) FC(fftconv_plan_convolve_packed(plan, N, kern, KH, KW, out));
    FC(fftconv_plan_synchronize(plan));
    FC(fftconv_plan_get_profile(plan, &pr, 1));
    FC(fftconv_plan_set_option(plan, "profile", 0));
    printf("%-58s rows %.2f  cols %.2f us per map\n", tag, pr.ms[1] / pr.units[1] * 1e3, pr.ms[2] / pr.units[2] * 1e3);
}
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    CHECK(hipSetDevice(0));
    CHECK(hipMalloc(&img, (size_t)4096 * 4096 * 4));
    CHECK(hipMalloc(&kern, (size_t)N * KH * KW * 4));
    // pseudo-random inputs (the kernels' speed depends on the data through the power limit)
    {
        std::vector<float> h((size_t)4096 * 4096);
        unsigned s = 12345u;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (s >> 8) * (1.0f / 16777216.0f); }
        CHECK(hipMemcpy(img, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(kern, h.data(), (size_t)N * KH * KW * 4, hipMemcpyHostToDevice));
    }
    fftconv_plan* p1 = new_plan();
    float *o1, *o2, *o3;
    CHECK(hipMalloc(&o1, OUT_BYTES));
    trial("fresh process: plan 1, map buffer 1", p1, o1);
    CHECK(hipMalloc(&o2, OUT_BYTES));
    trial("plan 1, map buffer 2 (allocated next)", p1, o2);
    // churn: 96 allocations of odd sizes (3 MB .. 1.5 GB), every other one freed
    std::vector<void*> keep;
    unsigned s = 777u;
    for (int i = 0; i < 96; i++) {
        s = s * 1664525u + 1013904223u;
        size_t bytes = ((size_t)(s >> 12) % 1500 + 3) * ((size_t)1 << 20) + ((s >> 4) & 0xff) * 4096;
        void* q;
        CHECK(hipMalloc(&q, bytes));
        keep.push_back(q);
    }
    for (size_t i = 0; i < keep.size(); i += 2) { CHECK(hipFree(keep[i])); keep[i] = nullptr; }
    CHECK(hipMalloc(&o3, OUT_BYTES));
    trial("after allocator churn: plan 1, map buffer 3", p1, o3);
    trial("plan 1, map buffer 1 again", p1, o1);
    fftconv_plan* p2 = new_plan();
    trial("plan 2 (intermediate allocated after the churn), buffer 1", p2, o1);
    trial("plan 2, buffer 3", p2, o3);
    for (int i = 0; i < 4; i++) {
        float* o;
        CHECK(hipMalloc(&o, OUT_BYTES));
        char tag[64];
        snprintf(tag, sizeof tag, "plan 1, further map buffer %d", i);
        trial(tag, p1, o);
    }
    for (int i = 0; i < 3; i++) {
        fftconv_plan* p = new_plan();
        char tag[64];
        snprintf(tag, sizeof tag, "further plan %d, buffer 1", i);
        trial(tag, p, o1);
    }
    return 0;
}
