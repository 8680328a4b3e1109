// planner.hpp -- host-side planning: transform lengths, radix sequences, twiddle / digit-reversal
// / real<->half-complex pair tables.  Header-only, host code only (used by the product library
// and by the test-only emulator).
//
// Sizing contract (reference: src/cudaConvFFTData.h:96-102 computeFFTsize16,
// src/cudaConvolutionFFT.cu:103-112): the OUTPUT window is FFT_X = ceil16(DATA_X + MAXK_X - 1).
// The INTERNAL transform length L_X only has to satisfy L_X >= DATA_X + MAXK_X - 1 (linear
// convolution support); it is chosen as the cheapest length that factors into the supported
// radices, preferring lengths that have specialised kernels.  When L_X == FFT_X (cfg1 / cfg3 / cfg5:
// 288, 4224, 2112) the engine computes exactly the reference's circular convolution modulo FFT_X;
// cfg2 (window 1088) and cfg4 (window 4160) run on 1152 / 4224 and crop.
#pragma once
#include <cmath>
#include <vector>

#include "fc_common.hpp"
#include "fft_lds.hpp"

namespace fc {

inline int fft_size16(int n) {  // src/cudaConvFFTData.h:96-102
    int mod = n / 16, rem = n % 16;
    return mod * 16 + (rem > 0 ? 16 : 0);
}

// The reference's alternative sizing (computeFFTsize, src/cudaConvFFTData.h:67-94; unused by its
// shipped path): align up to 16, then up to the next power of two.
inline int fft_size_pow2(int n) {
    int v = fft_size16(n), p = 1;
    while (p < v) p <<= 1;
    return p;
}

// Relative cost of one stage of radix R per element (butterfly + twiddle + LDS round trip).
inline double radix_cost(int R) {
    switch (R) {
        case 2: return 13.0;
        case 3: return 14.0;
        case 4: return 14.0;
        case 5: return 16.5;
        case 7: return 18.5;
        case 8: return 16.0;
        case 11: return 22.5;
        case 13: return 25.0;
        case 16: return 21.0;
        case 17: return 29.0;
        default: return 1e9;
    }
}

// Radix sequence (stage order) for length L, or empty if L has a prime factor > 17.
// Even radices first, largest first; odd radices after, ascending.
inline std::vector<int> factorize(int L) {
    std::vector<int> r;
    if (L < 1) return r;
    if (L == 1) return r;
    int e = 0, m = L;
    while (m % 2 == 0) { m /= 2; e++; }
    int a = e / 4, rem = e % 4;
    if (rem == 1 && a >= 1) {  // 16^(a-1) * 8 * 4
        for (int i = 0; i < a - 1; i++) r.push_back(16);
        r.push_back(8);
        r.push_back(4);
    } else {
        for (int i = 0; i < a; i++) r.push_back(16);
        if (rem == 3) r.push_back(8);
        if (rem == 2) r.push_back(4);
        if (rem == 1) r.push_back(2);
    }
    const int odd[] = {3, 5, 7, 11, 13, 17};
    for (int p : odd)
        while (m % p == 0) { r.push_back(p); m /= p; }
    if (m != 1) r.clear();
    if (m != 1 || (int)r.size() > FC_MAX_STAGES) return std::vector<int>();
    return r;
}

inline bool length_supported(int L) { return L == 1 || !factorize(L).empty(); }

// Lengths for which a specialised kernel exists (fast_paths.hpp) run ~2.5x faster than the generic
// kernels.  The cost factors come from fast_paths.hpp through this struct so that this header stays
// independent of it (nullptr: no preference); `ctx` is the widest kernel the row kernel must take.
// A factor is the measured cost per point of that length's specialised kernels relative to the generic
// estimate below (0: the length has none).
using LengthFactor = double (*)(int L, int ctx);
struct LengthPrefs {
    LengthFactor fast_rows = nullptr;   // w direction (complex transform of length L)
    LengthFactor fast_cols = nullptr;   // h direction (complex transform of length L / 2)
    int max_kw = 1;
};

inline double length_cost(int L, bool real_half, const LengthPrefs& prefs) {
    // real_half: the transform actually run is complex of length L/2 (+ pair pass)
    int Lc = real_half ? L / 2 : L;
    std::vector<int> r = factorize(Lc);
    if (Lc != 1 && r.empty()) return 1e30;
    double c = 40.0 + (real_half ? 10.0 : 0.0);
    for (int x : r) c += radix_cost(x) * (real_half ? 0.5 : 1.0);
    const double f = real_half ? (prefs.fast_cols ? prefs.fast_cols(Lc, 0) : 0.0) : (prefs.fast_rows ? prefs.fast_rows(L, prefs.max_kw) : 0.0);
    if (f > 0.0) c *= f;
    return c * (double)L;
}

// Cheapest supported length >= need (even if real_half).  `exact` (>= need, e.g. the ceil16
// window) wins ties and is preferred when within 10 % of the optimum, so that the common case
// reproduces the reference's circular-convolution modulus exactly.
// `cap` > 0: no length above it is considered (PlanTuning::max_transform); -1 if none fits.
inline int choose_length(int need, bool real_half, int exact, const LengthPrefs& prefs = LengthPrefs(), int cap = 0) {
    if (need < 1) need = 1;
    int best = -1;
    double bc = 1e30;
    int hi = 2 * need + 32;
    if (cap > 0) { hi = hi < cap ? hi : cap; if (exact > cap) exact = 0; }
    for (int L = need; L <= hi; L++) {
        if (real_half && (L & 1)) continue;
        double c = length_cost(L, real_half, prefs);
        if (c < bc) { bc = c; best = L; }
    }
    if (exact >= need && (!real_half || !(exact & 1))) {
        double c = length_cost(exact, real_half, prefs);
        if (c <= bc * 1.10) best = exact;
    }
    return best;
}

struct Plan1D {
    int L = 0;
    std::vector<int> radices;
    FftDesc desc{};
    std::vector<c32> tw;   // all stages, access order
    std::vector<int> pos;  // pos[k] = LDS position of forward-transform bin k
};

// Plan for an explicit radix sequence (product must be L).
inline Plan1D make_plan1d_seq(int L, const std::vector<int>& radices) {
    Plan1D p;
    p.L = L;
    p.radices = radices;
    p.desc.L = L;
    p.desc.ns = (int)p.radices.size();
    int n = L;
    for (int t = 0; t < p.desc.ns; t++) {
        int R = p.radices[t];
        int m = n / R;
        StageDesc& s = p.desc.st[t];
        s.R = R;
        s.m = m;
        if (m > 1) {
            s.tw_off = (int)p.tw.size();
            for (int c = 1; c < R; c++)
                for (int b = 0; b < m; b++) {
                    double ang = -2.0 * M_PI * (double)((long long)b * c % n) / (double)n;
                    p.tw.push_back(mk((float)std::cos(ang), (float)std::sin(ang)));
                }
        } else {
            s.tw_off = -1;
        }
        n = m;
    }
    if (p.tw.empty()) p.tw.push_back(mk(1.f, 0.f));
    p.pos.resize(L);
    for (int k = 0; k < L; k++) {
        int kk = k, pos = 0, len = L;
        for (int t = 0; t < p.desc.ns; t++) {
            int R = p.radices[t];
            len /= R;
            pos += (kk % R) * len;
            kk /= R;
        }
        p.pos[k] = pos;
    }
    return p;
}

inline Plan1D make_plan1d(int L) { return make_plan1d_seq(L, factorize(L)); }

// Pair table for the real <-> half-complex conversion around a complex transform of length M
// (real length N = 2M).  Entry 0 is the DC/Nyquist item (a = pos(0), b = M: the extra slot);
// entry p >= 1 pairs bin k = p with bin M-k; if 2k == M the entry is its own partner (a == b).
inline std::vector<PairEntry> make_pair_table(const Plan1D& pm) {
    const int M = pm.L;
    std::vector<PairEntry> t;
    PairEntry e0;
    e0.a = pm.pos[0];
    e0.b = M;
    e0.w = mk(1.f, 0.f);
    t.push_back(e0);
    for (int k = 1; 2 * k <= M; k++) {
        PairEntry e;
        e.a = pm.pos[k];
        e.b = pm.pos[M - k];
        double ang = -2.0 * M_PI * (double)k / (double)(2 * M);
        e.w = mk((float)std::cos(ang), (float)std::sin(ang));
        t.push_back(e);
    }
    return t;
}

// Row pitch (in c32) of one LDS-resident column sequence of M+1 bins: >= M+1 and == 2 mod 16
// so that the 8 columns of a gather tile land on distinct banks.
inline int lds_col_pitch(int M) {
    int p = M + 1;
    int r = p % 16;
    p += (r <= 2) ? (2 - r) : (18 - r);
    return p;
}

}  // namespace fc
