#!/usr/bin/env python3
"""Latency of the reference's own entry (host arrays in, host arrays out, synchronous) at the sizes a MATLAB user of the
reference calls it with: the demo's shape (demoCudaConvolutionFFT.m: 64 x 8 x 5 against three 10 x 4 x 5 kernels), BASELINE cfg1
and cfg2.  Cached one-shot calls (fftconv_convolution_fft through the ctypes mirror, caller buffers reused) and the same call on a
plan the caller keeps; where the time goes by fftconv_last_call_timing.  GPU box, repository root."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, util
fc = util.load_package()
REPS = int(os.environ.get("REPS", "300"))


def med(v):
    return float(np.median(np.asarray(v)))


for name, (H, W, F, kh, kw, n) in {"demo": (64, 8, 5, 10, 4, 3), "cfg1": (256, 256, 1, 31, 31, 1), "cfg1 x 4 kernels": (256, 256, 1, 31, 31, 4),
                                   "cfg2": (1024, 1024, 1, 63, 63, 16)}.items():
    img, ks = util.synth(11, H, W, F, kh, kw, n)
    out = fc.cudaConvolutionFFT(img, kh, kw, ks)                  # first call: plan creation
    bufs = [np.empty(o.shape, dtype=np.float32, order="F") for o in out]
    wall, parts = [], []
    for r in range(REPS):
        t0 = time.perf_counter()
        fc.cudaConvolutionFFT(img, kh, kw, ks, out=bufs)
        wall.append(time.perf_counter() - t0)
        t = fc.last_call_timing()
        parts.append((t["plan_ms"], t["image_ms"], t["convolve_ms"], t["release_ms"], t["total_ms"], t["cache_hit"]))
    assert all(np.array_equal(a, b) for a, b in zip(bufs, out))
    p = np.asarray(parts)
    print("%-18s cached one-shot: %7.1f us per call through ctypes (library %.1f us: plan %.1f  image %.1f  convolve %.1f  release %.1f; cache hits %d / %d)"
          % (name, med(wall) * 1e6, med(p[:, 4]) * 1e3, med(p[:, 0]) * 1e3, med(p[:, 1]) * 1e3, med(p[:, 2]) * 1e3, med(p[:, 3]) * 1e3, int(p[:, 5].sum()), REPS), flush=True)
    for pinned in (1, 0):
      with fc.Plan(H, W, F, kh, kw) as plan:
        plan.set_option("host_pinned", pinned)
        plan.set_image(img); plan.convolve(ks, out=bufs)
        w_img, w_conv = [], []
        for r in range(REPS):
            t0 = time.perf_counter(); plan.set_image(img); t1 = time.perf_counter(); plan.convolve(ks, out=bufs); t2 = time.perf_counter()
            w_img.append(t1 - t0); w_conv.append(t2 - t1)
        assert all(np.array_equal(a, b) for a, b in zip(bufs, out))
        print("%-18s kept plan, host_pinned %d: %7.1f us per call (set_image %.1f us, convolve %.1f us); maps %d x %d x %d = %.2f MB"
              % (name, pinned, (med(w_img) + med(w_conv)) * 1e6, med(w_img) * 1e6, med(w_conv) * 1e6, n, out[0].shape[0], out[0].shape[1], n * out[0].size * 4 / 1e6), flush=True)
