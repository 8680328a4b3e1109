#!/usr/bin/env python3
"""Prints, per shape, how far the product path and the CPU oracle are from the vendor FFT library of this device (rocFFT behind
hipFFT, through torch.fft) running the reference's own sequence in fp32 and fp64 -- the figures behind
tests/test_gpu_parity.py::test_matches_the_vendor_fft_library_on_the_device.  Test / measurement infrastructure only."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, util
import test_gpu_parity as T
fc = util.load_package()
orc = util.Oracle()
print("shape (H, W, F, kh, kw, maps)            product vs rocFFT f64   product vs rocFFT f32   oracle vs rocFFT f64   rocFFT f32 vs f64")
for shape in [(64, 8, 5, 10, 4, 3), (256, 256, 1, 31, 31, 1), (1024, 1024, 1, 63, 63, 2), (2048, 2048, 1, 63, 63, 1), (4096, 4096, 1, 127, 127, 1),
              (4096, 4096, 1, 63, 63, 1), (300, 260, 3, 31, 17, 2)]:
    H, W, F, kh, kw, n = shape
    rng = np.random.default_rng(sum(shape) + 11)
    data = rng.random((H, W, F), dtype=np.float32)
    ks = [rng.random((kh, kw, F), dtype=np.float32) for _ in range(n)]
    got = fc.cudaConvolutionFFT(data, kh, kw, ks)
    r32 = T._vendor_fft_conv(torch, data, kh, kw, ks, torch.float32)
    r64 = T._vendor_fft_conv(torch, data, kh, kw, ks, torch.float64)
    o = orc.conv_fft(data, kh, kw, ks)
    e = lambda a, b: max(util.rel_err(x, y) for x, y in zip(a, b))
    print("%-40s %-23.2e %-23.2e %-22.2e %.2e" % (shape, e(got, r64), e(got, r32), e(o, r64), e(r32, r64)))
