"""Row-kernel time per map against the number of feature planes F (the reference's sumAlongFeatures case):
cfg3 and cfg5 geometries, 64 maps.  usage (GPU box): python3 tools/f_scaling.py"""
import importlib, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fc = importlib.import_module("cuda-fft-convolution_amd")
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream(dev)
rng = np.random.default_rng(1)
for (H, k, n) in ((4096, 127, 64), (2048, 63, 64), (512, 31, 256)):
    for F in (1, 2, 3, 4, 8, 16, 32):
        if H == 4096 and F > 8:
            continue
        img = torch.from_numpy(rng.random((F, H, H), dtype=np.float32)).to(dev)
        ker = torch.from_numpy(rng.random((n, F, k, k), dtype=np.float32)).to(dev)
        plan = fc.Plan(H, H, F, k, k, gpuId=0, stream=stream.cuda_stream)
        plan.set_image_device(img.data_ptr())
        info = plan.info
        out = torch.empty((n, info.fft_w, info.fft_h), dtype=torch.float32, device=dev)
        for _ in range(max(3, int(0.08 / (n * F * 30e-6 * (info.fft_w / 4224.0) ** 2)))):
            plan.convolve_packed_device(n, ker.data_ptr(), k, k, out.data_ptr())
        torch.cuda.synchronize()
        plan.set_option("profile", 1); plan.profile(reset=True)
        for _ in range(4):
            plan.convolve_packed_device(n, ker.data_ptr(), k, k, out.data_ptr())
        torch.cuda.synchronize()
        p = plan.profile(reset=True)
        r = {kk: round(v["ms"] / max(1.0, v["units"]) * 1e3, 2) for kk, v in p.items() if v["launches"]}
        px = info.fft_w * info.fft_h
        tot = sum(r.get(x, 0) for x in ("kernel_cols", "spectral_rows", "cols_c2r"))
        print("%4d^2 (x) %3d^2  F %2d  window %d  per map: %s  -> %.1f Gpx/s, %.1f G(px*F)/s" % (H, k, F, info.fft_w, r, px / tot / 1e3, px * F / tot / 1e3), flush=True)
        plan.destroy(); del img, ker, out
