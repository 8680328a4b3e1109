// butterflies.hpp -- in-register DFTs of small length R ("radix-R butterflies").
//
// Dft<R, SGN>::run(v) replaces v[0..R) by its DFT with kernel exp(SGN*2*pi*i*j*k/R), natural
// order in and out (SGN = -1 forward, +1 unnormalised inverse).  All indices are compile-time
// so the arrays live in VGPRs.  Odd primes use the symmetric-pair form (O(R^2/2) FMAs, fine up
// to 17 for an HBM-bound transform); composites are built by Cooley-Tukey in registers.
#pragma once
#include "dft_consts.hpp"
#include "fc_common.hpp"

namespace fc {

// a * (SGN*i)
template <int SGN>
FC_HD c32 mul_j(c32 a) {
    if constexpr (SGN > 0) return mk(-a.y, a.x);
    else return mk(a.y, -a.x);
}
// a + (SGN*i)*b, a - (SGN*i)*b: one packed add each
template <int SGN>
FC_HD c32 add_j(c32 a, c32 b) {
    if constexpr (SGN > 0) return add_jb(a, b);
    else return sub_jb(a, b);
}
template <int SGN>
FC_HD c32 sub_j(c32 a, c32 b) {
    if constexpr (SGN > 0) return sub_jb(a, b);
    else return add_jb(a, b);
}

// a * exp(SGN*2*pi*i*K/R), K and R compile-time.
template <int R, int K, int SGN>
FC_HD c32 mul_root(c32 a) {
    constexpr int k = ((K % R) + R) % R;
    if constexpr (k == 0) {
        return a;
    } else if constexpr (2 * k == R) {
        return mk(-a.x, -a.y);
    } else if constexpr (4 * k == R) {
        return mul_j<SGN>(a);
    } else if constexpr (4 * k == 3 * R) {
        return mul_j<-SGN>(a);
    } else if constexpr ((8 * k) % R == 0) {
        // odd multiples of pi/4: (sc + i*SGN*ss)/sqrt(2), sc, ss = +-1
        constexpr int q = (8 * k) / R;  // 1,3,5,7
        constexpr float h = 0.70710678118654752f;
        constexpr float sc = (q == 1 || q == 7) ? 1.f : -1.f;
        constexpr float ss = ((q == 1 || q == 3) ? 1.f : -1.f) * (SGN > 0 ? 1.f : -1.f);
        // (sc*a + i*ss*a) * h
        c32 t = (ss * sc > 0.f) ? add_jb(a, a) : sub_jb(a, a);
        return scale(t, sc * h);
    } else {
        constexpr float c = Roots<R>::c[k];
        constexpr float s = (SGN > 0 ? 1.f : -1.f) * Roots<R>::s[k];
        // a*c + i*s*a
        return fma_js(s, a, scale(a, c));
    }
}

constexpr int smallest_factor(int r) {
    if (r % 4 == 0 && r > 4) return 4;
    if (r % 2 == 0) return 2;
    for (int f = 3; f * f <= r; f += 2)
        if (r % f == 0) return f;
    return r;
}
constexpr bool is_prime(int r) { return r >= 2 && smallest_factor(r) == r && r != 4; }

template <int R, int SGN, class Enable = void>
struct Dft;

// run_nz<NZ>(v): the same transform when v[NZ..R) are structural zeros (the zero padding of a short kernel row seen by
// the middle stage of a pruned transform): the composite form skips the butterflies whose inputs are a single value.
template <int SGN>
struct Dft<1, SGN, void> {
    static FC_HD void run(c32 (&)[1]) {}
    template <int NZ>
    static FC_HD void run_nz(c32 (&)[1]) {}
};

template <int SGN>
struct Dft<2, SGN, void> {
    static FC_HD void run(c32 (&v)[2]) {
        c32 a = v[0], b = v[1];
        v[0] = a + b;
        v[1] = a - b;
    }
    template <int NZ>
    static FC_HD void run_nz(c32 (&v)[2]) {
        if constexpr (NZ <= 1) v[1] = v[0];
        else run(v);
    }
};

template <int SGN>
struct Dft<4, SGN, void> {
    static FC_HD void run(c32 (&v)[4]) {
        c32 s0 = v[0] + v[2], s1 = v[0] - v[2];
        c32 s2 = v[1] + v[3], d = v[1] - v[3];
        v[0] = s0 + s2;
        v[1] = add_j<SGN>(s1, d);
        v[2] = s0 - s2;
        v[3] = sub_j<SGN>(s1, d);
    }
    template <int NZ>
    static FC_HD void run_nz(c32 (&v)[4]) {
        if constexpr (NZ <= 1) {
            v[1] = v[0]; v[2] = v[0]; v[3] = v[0];
        } else if constexpr (NZ == 2) {     // v[2] = v[3] = 0
            const c32 a = v[0], b = v[1];
            v[0] = a + b;
            v[1] = add_j<SGN>(a, b);
            v[2] = a - b;
            v[3] = sub_j<SGN>(a, b);
        } else {
            run(v);
        }
    }
};

// Odd prime P: X[m] = A_m + (SGN*i) B_m, X[P-m] = A_m - (SGN*i) B_m with
//   A_m = v0 + sum_k cos(2 pi m k / P) (v[k] + v[P-k]),  B_m = sum_k sin(2 pi m k / P) (v[k] - v[P-k]).
template <int P, int SGN>
struct DftOddPrime {
    static FC_HD void run(c32 (&v)[P]) {
        constexpr int H = (P - 1) / 2;
        c32 s[H], d[H];
        static_for<0, H>([&](auto k_) {
            constexpr int k = decltype(k_)::value;
            s[k] = v[k + 1] + v[P - 1 - k];
            d[k] = v[k + 1] - v[P - 1 - k];
        });
        c32 v0 = v[0];
        c32 x0 = v0;
        static_for<0, H>([&](auto k_) { x0 = x0 + s[decltype(k_)::value]; });
        v[0] = x0;
        static_for<1, H + 1>([&](auto m_) {
            constexpr int m = decltype(m_)::value;
            c32 A = v0, B = mk(0.f, 0.f);
            static_for<0, H>([&](auto k_) {
                constexpr int k = decltype(k_)::value;
                constexpr int idx = (m * (k + 1)) % P;
                constexpr float c = Roots<P>::c[idx];
                constexpr float sn = Roots<P>::s[idx];
                A = fma_real(c, s[k], A);
                if constexpr (k == 0) B = scale(d[k], sn);
                else B = fma_real(sn, d[k], B);
            });
            v[m] = add_j<SGN>(A, B);
            v[P - m] = sub_j<SGN>(A, B);
        });
    }
};

template <int R, int SGN>
struct Dft<R, SGN, std::enable_if_t<(R >= 3) && (R % 2 == 1) && is_prime(R)>> {
    static FC_HD void run(c32 (&v)[R]) { DftOddPrime<R, SGN>::run(v); }
    template <int NZ>
    static FC_HD void run_nz(c32 (&v)[R]) {
        if constexpr (NZ <= 1) static_for<1, R>([&](auto k_) { v[decltype(k_)::value] = v[0]; });
        else DftOddPrime<R, SGN>::run(v);
    }
};

// Composite R = R1*R2, decimation in frequency in registers:
//   y_c[b] = (sum_a v[a*R2 + b] w_R1^{a c}) * w_R^{b c};   X[c + R1*k] = sum_b y_c[b] w_R2^{b k}.
template <int R1, int R2, int SGN>
struct DftCT {
    static constexpr int R = R1 * R2;
    static FC_HD void run(c32 (&v)[R]) {
        c32 y[R];
        static_for<0, R2>([&](auto b_) {
            constexpr int b = decltype(b_)::value;
            c32 t[R1];
            static_for<0, R1>([&](auto a_) {
                constexpr int a = decltype(a_)::value;
                t[a] = v[a * R2 + b];
            });
            Dft<R1, SGN>::run(t);
            static_for<0, R1>([&](auto c_) {
                constexpr int c = decltype(c_)::value;
                y[c * R2 + b] = mul_root<R, b * c, SGN>(t[c]);
            });
        });
        static_for<0, R1>([&](auto c_) {
            constexpr int c = decltype(c_)::value;
            c32 t[R2];
            static_for<0, R2>([&](auto b_) {
                constexpr int b = decltype(b_)::value;
                t[b] = y[c * R2 + b];
            });
            Dft<R2, SGN>::run(t);
            static_for<0, R2>([&](auto k_) {
                constexpr int k = decltype(k_)::value;
                v[c + R1 * k] = t[k];
            });
        });
    }
};

// The same with v[NZ..R) zero, NZ <= R2: every first-stage butterfly has the single input v[b] (its R1 outputs are
// copies), and only the first NZ inputs of every second-stage transform are non-zero.
template <int R1, int R2, int SGN, int NZ>
struct DftCTnz {
    static constexpr int R = R1 * R2;
    static_assert(NZ >= 1 && NZ <= R2, "pruned composite: the non-zero inputs must lie in the first sub-block");
    static FC_HD void run(c32 (&v)[R]) {
        c32 y[R];
        static_for<0, NZ>([&](auto b_) {
            constexpr int b = decltype(b_)::value;
            const c32 x = v[b];
            static_for<0, R1>([&](auto c_) {
                constexpr int c = decltype(c_)::value;
                y[c * R2 + b] = mul_root<R, b * c, SGN>(x);
            });
        });
        static_for<0, R1>([&](auto c_) {
            constexpr int c = decltype(c_)::value;
            c32 t[R2];
            static_for<0, R2>([&](auto b_) {
                constexpr int b = decltype(b_)::value;
                if constexpr (b < NZ) t[b] = y[c * R2 + b];
                else t[b] = mk(0.f, 0.f);
            });
            Dft<R2, SGN>::template run_nz<NZ>(t);
            static_for<0, R2>([&](auto k_) {
                constexpr int k = decltype(k_)::value;
                v[c + R1 * k] = t[k];
            });
        });
    }
};

template <int R, int SGN>
struct Dft<R, SGN, std::enable_if_t<(R > 4) && !is_prime(R)>> {
    static constexpr int F1 = smallest_factor(R), F2 = R / smallest_factor(R);
    static FC_HD void run(c32 (&v)[R]) { DftCT<F1, F2, SGN>::run(v); }
    template <int NZ>
    static FC_HD void run_nz(c32 (&v)[R]) {
        if constexpr (NZ >= 1 && NZ <= F2) DftCTnz<F1, F2, SGN, NZ>::run(v);
        else DftCT<F1, F2, SGN>::run(v);
    }
};

}  // namespace fc
