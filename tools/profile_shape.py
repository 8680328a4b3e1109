#!/usr/bin/env python3
"""Per-kernel HIP-event times of one device-resident step for arbitrary sizes: profile_shape.py H W K [filters] [F]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, util
fc = util.load_package()
H, W, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 64
F = int(sys.argv[5]) if len(sys.argv) > 5 else 1
opts = {"exact_window": 1} if os.environ.get("EXACT") else None
if os.environ.get("ONE_PASS"): opts = dict(opts or {}, blockwise=1)      # never block-wise (A/B of the overlap-save blocks)
dev = torch.device("cuda", 0)
rng = np.random.default_rng(1)
img = torch.from_numpy(rng.random((F, W, H), dtype=np.float32)).to(dev)
ker = torch.from_numpy(rng.random((n, F, K, K), dtype=np.float32)).to(dev)
with fc.Plan(H, W, F, K, K, options=opts) as p:
    i = p.info
    if os.environ.get("DYN"): p.set_option("dynamic_tiles", int(os.environ["DYN"]))      # tile queue of the column kernels on / off (A/B)
    stagger = None
    if os.environ.get("STAGGER"):      # -DFC_ROWS_STAGGER_TICKS builds: the per-CU arrival counters (+ what each first-round workgroup saw)
        stagger = torch.zeros(8192, dtype=torch.int32, device=dev)
        try: p.set_option("timeline_ptr", stagger.data_ptr())
        except Exception: stagger = None          # a library without the experiment
    out = torch.empty((n, i.fft_w, i.fft_h), dtype=torch.float32, device=dev)
    def step():
        p.set_image_device(img.data_ptr()); p.convolve_packed_device(n, ker.data_ptr(), K, K, out.data_ptr())
    for _ in range(30): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    p.set_option("profile", 1); p.profile(reset=True)
    for _ in range(5): step()
    torch.cuda.synchronize()
    pr = p.profile(reset=True)
    print("%dx%d K=%d F=%d n=%d window %dx%d transform %dx%d%s spec %d: %.1f us/step  %.1f Gpx/s | " % (H, W, K, F, n, i.fft_h, i.fft_w, i.transform_h, i.transform_w,
          (" x%d blocks" % p.get_option("blockwise")) if p.get_option("blockwise") else "", p.get_option("specialised_kernels"), dt * 1e6, n * i.fft_h * i.fft_w / dt / 1e9) +
          "  ".join("%s %.1f us x%d" % (k, v["ms"] / max(1, v["launches"]) * 1e3, v["launches"] // 5) for k, v in pr.items()))
    if stagger is not None:
        import collections
        st = stagger.cpu().numpy()
        keys, ks = st[2560:4608:2], st[2561:4608:2]
        per_cu = collections.Counter(int(x) for x in keys)
        same = sum(1 for b in range(256) if len({int(keys[b + 256 * j]) for j in range(4)}) == 1)
        print("   stagger: %d CU keys among the first 1024 workgroups, workgroups per key %s; blocks b, b+256, b+512, b+768 on one CU for %d of 256 b; first keys %s; ranks of blocks 0..15: %s"
              % (len(per_cu), dict(collections.Counter(per_cu.values())), same, [hex(int(x)) for x in keys[:12]], [int(x) for x in ks[:16]]))
