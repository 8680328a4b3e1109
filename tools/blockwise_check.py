#!/usr/bin/env python3
"""Overlap-save block-wise plans against the oracle (small, forced by max_transform) and against the one-pass plan (large, chosen
by the cost model), then per-kernel times of both on the large shapes.  GPU box, repository root."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, util
fc = util.load_package()
oracle = util.Oracle()
rng = np.random.default_rng(5)
for (H, W, F, kh, kw, n, mt) in [(700, 500, 2, 9, 13, 5, 288), (1000, 300, 1, 31, 17, 3, 576), (300, 1500, 3, 16, 16, 4, 576), (272, 272, 1, 17, 17, 2, 288),
                                 (560, 1130, 1, 1, 1, 2, 576)]:
    data = rng.standard_normal((H, W, F)).astype(np.float32)
    ks = [rng.standard_normal((kh, kw, F)).astype(np.float32) for _ in range(n)]
    if n > 2: ks[1] = rng.standard_normal((max(1, kh - 2), max(1, kw - 1), F)).astype(np.float32)
    ref = oracle.conv_fft(data, kh, kw, ks)
    with fc.Plan(H, W, F, kh, kw, options={"max_transform": mt}) as p:
        i = p.info
        p.set_image(data)
        got = p.convolve(ks)
        err = max(util.rel_err(g, r) for g, r in zip(got, ref))
        print("%dx%dx%d k %dx%d n %d max_transform %d: blocks %d transform %dx%d window %dx%d  err %.2e %s" % (H, W, F, kh, kw, n, mt, p.get_option("blockwise"),
              i.transform_h, i.transform_w, i.fft_h, i.fft_w, err, "ok" if err < 1e-5 else "FAILED"), flush=True)
dev = torch.device("cuda", 0)
for (H, W, K, n) in [(8192, 8192, 63, 4), (4900, 4900, 63, 4), (6000, 8000, 31, 3), (10000, 3000, 63, 3)]:
    img = torch.from_numpy(rng.random((1, W, H), dtype=np.float32)).to(dev)
    ker = torch.from_numpy(rng.random((n, 1, K, K), dtype=np.float32)).to(dev)
    outs = []
    for opts in (None, {"blockwise": 1}):
        with fc.Plan(H, W, 1, K, K, options=opts) as p:
            i = p.info
            out = torch.full((n, i.fft_w, i.fft_h), float("nan"), dtype=torch.float32, device=dev)
            p.set_image_device(img.data_ptr()); p.convolve_packed_device(n, ker.data_ptr(), K, K, out.data_ptr()); p.synchronize()
            outs.append((out, p.get_option("blockwise"), i.transform_h, i.transform_w))
    a, b = outs[0][0], outs[1][0]
    err = float((a - b).abs().max() / b.abs().max())
    print("%dx%d K=%d: default plan blocks %d transform %dx%d; one pass transform %dx%d; max difference %.2e of the maximum %s" % (H, W, K, outs[0][1], outs[0][2], outs[0][3],
          outs[1][2], outs[1][3], err, "ok" if err < 2e-6 and bool(torch.isfinite(a).all()) else "FAILED"), flush=True)
    del outs, a, b, out
