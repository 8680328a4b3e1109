// class_map.cpp -- the output kernel's slow / fast state against WHICH allocation the map buffer is: 44 map
// buffers of 4.5 GB allocated one after the other (200 GB), each tried with the same plan (same intermediate),
// then the same buffers with a second plan.  Looks for the size of the regions that share a state.
// Build: as frag_placement.cpp
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
static double trial(fftconv_plan* plan, float* out, int warm) {
    for (int i = 0; i < warm; i++) FC(fftconv_plan_convolve_packed(plan, N, kern, KH, KW, out));
    FC(fftconv_plan_synchronize(plan));
    fftconv_profile pr;
    FC(fftconv_plan_set_option(plan, "profile", 1));
    FC(fftconv_plan_get_profile(plan, &pr, 1));
    for (int i = 0; i < 5; i++) FC(fftconv_plan_convolve_packed(plan, N, kern, KH, KW, out));
    FC(fftconv_plan_synchronize(plan));
    FC(fftconv_plan_get_profile(plan, &pr, 1));
    FC(fftconv_plan_set_option(plan, "profile", 0));
    return pr.ms[2] / pr.units[2] * 1e3;
}
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    CHECK(hipSetDevice(0));
    CHECK(hipMalloc(&img, (size_t)4096 * 4096 * 4));
    CHECK(hipMalloc(&kern, (size_t)N * KH * KW * 4));
    {
        std::vector<float> h((size_t)4096 * 4096);
        unsigned s = 12345u;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (s >> 8) * (1.0f / 16777216.0f); }
        CHECK(hipMemcpy(img, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(kern, h.data(), (size_t)N * KH * KW * 4, hipMemcpyHostToDevice));
    }
    fftconv_plan* p1 = new_plan();
    const int NB = 44;
    std::vector<float*> o(NB);
    for (int i = 0; i < NB; i++) CHECK(hipMalloc(&o[i], OUT_BYTES));
    fftconv_plan* p2 = new_plan();   // its intermediate is allocated at its first convolve, i.e. after all the buffers
    trial(p1, o[0], 14);
    printf("buffer: plan 1 / plan 2 (output kernel, us per map)\n");
    for (int i = 0; i < NB; i++) {
        const double a = trial(p1, o[i], 2);
        const double b = trial(p2, o[i], 2);
        printf("%2d (%p): %.2f %s / %.2f %s\n", i, (void*)o[i], a, a < 26.9 ? "fast" : "slow", b, b < 26.9 ? "fast" : "slow");
    }
    return 0;
}
