// pmc_states.cpp -- for hardware counters of the output kernel in its fast and in its slow placement state:
// classifies 12 map buffers by timing, then runs 4 convolves into the fastest and 4 into the slowest; under
// rocprofv3 --pmc the LAST 8 dispatches of k_fast_cols are those (first 4 fast, last 4 slow).
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
static double trial(fftconv_plan* plan, float* out, int warm, int reps) {
    for (int i = 0; i < warm; i++) FC(fftconv_plan_convolve_packed(plan, N, kern, KH, KW, out));
    FC(fftconv_plan_synchronize(plan));
    fftconv_profile pr;
    FC(fftconv_plan_set_option(plan, "profile", 1));
    FC(fftconv_plan_get_profile(plan, &pr, 1));
    for (int i = 0; i < reps; i++) FC(fftconv_plan_convolve_packed(plan, N, kern, KH, KW, out));
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
    fftconv_plan* p;
    FC(fftconv_plan_create(&p, 4096, 4096, 1, KH, KW, 0, nullptr));
    FC(fftconv_plan_set_image(p, img, FFTCONV_DEVICE));
    const int NB = 12;
    std::vector<float*> o(NB);
    for (int i = 0; i < NB; i++) CHECK(hipMalloc(&o[i], OUT_BYTES));
    trial(p, o[0], 6, 1);
    int fast = 0, slow = 0;
    double tf = 1e9, ts = 0;
    for (int i = 0; i < NB; i++) {
        const double t = trial(p, o[i], 1, 3);
        if (t < tf) { tf = t; fast = i; }
        if (t > ts) { ts = t; slow = i; }
    }
    printf("fastest buffer %d: %.2f us per map; slowest buffer %d: %.2f us per map (under the profiler)\n", fast, tf, slow, ts);
    const double a = trial(p, o[fast], 0, 4);
    const double b = trial(p, o[slow], 0, 4);
    printf("last 8 launches: 4 into the fast buffer (%.2f us per map), 4 into the slow one (%.2f)\n", a, b);
    return 0;
}
