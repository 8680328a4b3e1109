#!/usr/bin/env python3
"""Generates the golden fixtures of this directory (run once; fixtures are committed).

The reference ships NO golden vectors and has no Python to import, so
these fixtures are derived from the mathematical definition of its path with an independent
implementation: NumPy float64 `fft2/ifft2` in the shape of demoCudaConvolutionFFT.m:78-102, and
brute-force `sum_f conv2(data_f, kernel_f)` (demoCudaConvolutionFFT.m:91-96) where stated.
Inputs are float32 (what the reference's MEX accepts), expectations float64.

Each case_*.npz holds: data (H x W x F), maxk (2,), kernel_0..kernel_{n-1}, expect_0..expect_{n-1}
(FFT_H x FFT_W, FFT_X = ceil16(DATA_X + MAXK_X - 1)), and `how` (the generator used).
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def ceil16(n):
    return (n + 15) // 16 * 16


def fft_expect(data, mkh, mkw, kernels):
    H, W, F = data.shape
    fh, fw = ceil16(H + mkh - 1), ceil16(W + mkw - 1)
    D = np.fft.fft2(data.astype(np.float64), s=(fh, fw), axes=(0, 1))
    out = []
    for k in kernels:
        K = np.fft.fft2(k.astype(np.float64), s=(fh, fw), axes=(0, 1))
        out.append(np.real(np.fft.ifft2(D * K, axes=(0, 1))).sum(axis=2))
    return out


def direct_expect(data, mkh, mkw, kernels):
    """full linear convolution summed over features, embedded top-left in the window
    (valid because every kernel here is <= MAXK, so nothing wraps)"""
    H, W, F = data.shape
    fh, fw = ceil16(H + mkh - 1), ceil16(W + mkw - 1)
    out = []
    for k in kernels:
        kh, kw, _ = k.shape
        acc = np.zeros((fh, fw), dtype=np.float64)
        for f in range(F):
            for ky in range(kh):
                for kx in range(kw):
                    acc[ky:ky + H, kx:kx + W] += float(k[ky, kx, f]) * data[:, :, f].astype(np.float64)
        out.append(acc)
    return out


def save(name, data, mkh, mkw, kernels, expect, how):
    d = {"data": data.astype(np.float32), "maxk": np.array([mkh, mkw], dtype=np.int32), "how": np.array(how)}
    for i, (k, e) in enumerate(zip(kernels, expect)):
        d["kernel_%d" % i] = k.astype(np.float32)
        d["expect_%d" % i] = e
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
    print(name, data.shape, [k.shape for k in kernels], expect[0].shape)


def demo_case():
    """The reference's demo problem (demoCudaConvolutionFFT.m:37-69,110-113) with a fixed seed:
    64x8x5 data, 10x4x5 kernels, kernel copies planted in the data, kernels flipped,
    kernelCell = {kernel, kernel2 (kernel2(1) = 100), kernel}."""
    rng = np.random.default_rng(20131013)
    n, m, k, cn, cm = 64, 8, 5, 10, 4
    data = rng.random((n, m, k)).astype(np.float32)
    kernel = np.zeros((cn, cm, k), dtype=np.float32)
    kernel[:, :, 0] = np.arange(1, cn * cm + 1, dtype=np.float32).reshape((cn, cm), order="F")
    for i in range(1, k):
        kernel[:, :, i] = rng.random((cn, cm)).astype(np.float32)
    data[4:4 + cn, 1:1 + cm, 0] = kernel[:, :, 0]
    data[20:20 + cn, 0:cm, 1] = kernel[:, :, 0]
    data[0:cn, m - cm:m, k - 1] = kernel[:, :, 0]
    kernel[:, :, k - 1] = kernel[:, :, 0]
    kernel = kernel[::-1, ::-1, :].copy()          # "Flip Kernel (Required)"
    kernel2 = kernel.copy()
    kernel2[0, 0, 0] = 100.0                        # kernel2(1) = 100
    ks = [kernel, kernel2, kernel]
    return data, cn, cm, ks


def main():
    rng = np.random.default_rng(42)
    # 1. demo-shaped, brute force AND fft (they must agree; brute force is stored)
    data, cn, cm, ks = demo_case()
    e_dir = direct_expect(data, cn, cm, ks)
    e_fft = fft_expect(data, cn, cm, ks)
    assert max(np.abs(a - b).max() for a, b in zip(e_dir, e_fft)) < 1e-9
    save("case_demo", data, cn, cm, ks, e_dir, "direct (== numpy fft2 to 1e-9)")
    # 2. delta and all-ones kernels (closed forms): F = 1, 2-D-like
    data = rng.random((40, 24, 1)).astype(np.float32)
    delta = np.zeros((5, 3, 1), np.float32); delta[2, 1, 0] = 1.0
    ones = np.ones((5, 3, 1), np.float32)
    save("case_delta_ones", data, 5, 3, [delta, ones], direct_expect(data, 5, 3, [delta, ones]), "direct")
    # 3. mixed kernel sizes in one cell, non-square, even/odd sizes, F = 3
    data = rng.random((33, 47, 3)).astype(np.float32)
    ks = [rng.random((7, 5, 3)).astype(np.float32), rng.random((4, 5, 3)).astype(np.float32),
          rng.random((7, 2, 3)).astype(np.float32), rng.random((1, 1, 3)).astype(np.float32)]
    save("case_mixed_f3", data, 7, 5, ks, direct_expect(data, 7, 5, ks), "direct")
    # 4. F = 2, sizes whose window equals the transform (112 = 2^4*7)
    data = rng.standard_normal((100, 90, 2)).astype(np.float32)
    ks = [rng.standard_normal((13, 17, 2)).astype(np.float32) for _ in range(2)]
    save("case_f2_112", data, 13, 17, ks, fft_expect(data, 13, 17, ks), "numpy fft2")
    # 5. padded-size families of the BASELINE configs at reduced scale:
    #    144 = 2^4*3^2 (cfg1's 288), 272 = 2^4*17 (cfg2's 1088), 264 = 2^3*3*11 (cfg3/5), 208 = 2^4*13 (cfg4)
    for name, (H, W, kh, kw) in {"case_fam_144x272": (130, 242, 15, 31), "case_fam_264x208": (250, 190, 15, 19)}.items():
        data = rng.random((H, W, 1)).astype(np.float32)
        ks = [rng.random((kh, kw, 1)).astype(np.float32)]
        save(name, data, kh, kw, ks, fft_expect(data, kh, kw, ks), "numpy fft2")
    # 6. tiny / degenerate shapes
    data = rng.random((1, 1, 1)).astype(np.float32)
    ks = [np.array([[[0.5]]], np.float32)]
    save("case_1x1", data, 1, 1, ks, direct_expect(data, 1, 1, ks), "direct")
    data = rng.random((17, 1, 2)).astype(np.float32)
    ks = [rng.random((3, 1, 2)).astype(np.float32)]
    save("case_17x1", data, 3, 1, ks, direct_expect(data, 3, 1, ks), "direct")
    # 7. kernel larger than MAXK but within the window: circular aliasing modulo the ceil16 window
    #    (the reference does not guard this); window 48x32 is also the transform size
    data = rng.random((40, 20, 1)).astype(np.float32)
    ks = [rng.random((12, 16, 1)).astype(np.float32)]
    save("case_wrap", data, 9, 13, ks, fft_expect(data, 9, 13, ks), "numpy fft2 (circular, kernel > MAXK)")


if __name__ == "__main__":
    main()
