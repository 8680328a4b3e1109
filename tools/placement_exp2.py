"""Follow-up of placement_exp.py: what separates 21.4 us per map (row kernel, repeated 64-map calls) from
the 24-25 us of the bench?  Number of maps per call, idle gaps before a call, many back-to-back calls."""
import importlib, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fc = importlib.import_module("cuda-fft-convolution_amd")
dev = torch.device("cuda:0")
H = W = 4096; kh = kw = 127; F = 1; N = 256
rng = np.random.default_rng(1)
img = torch.from_numpy(rng.random((F, W, H), dtype=np.float32)).to(dev)
ker = torch.from_numpy(rng.random((N, F, kw, kh), dtype=np.float32)).to(dev)
stream = torch.cuda.current_stream(dev)
plan = fc.Plan(H, W, F, kh, kw, gpuId=0, stream=stream.cuda_stream)
plan.set_image_device(img.data_ptr())
out = torch.empty((N, 4224, 4224), dtype=torch.float32, device=dev)

def trial(tag, n, reps, warm=2, sleep=0.0, set_image=False):
    for _ in range(warm):
        plan.convolve_packed_device(n, ker.data_ptr(), kh, kw, out.data_ptr())
    torch.cuda.synchronize()
    if sleep:
        time.sleep(sleep)
    plan.set_option("profile", 1); plan.profile(reset=True)
    t0 = time.perf_counter()
    for _ in range(reps):
        if set_image:
            plan.set_image_device(img.data_ptr())
        plan.convolve_packed_device(n, ker.data_ptr(), kh, kw, out.data_ptr())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    p = plan.profile(reset=True); plan.set_option("profile", 0)
    r = {k: round(v["ms"] / max(1.0, v["units"]) * 1e3, 2) for k, v in p.items() if v["launches"]}
    print("%-34s n %3d reps %3d  %.1f Gpx/s  %s" % (tag, n, reps, n * reps * 4224 * 4224 / dt / 1e9, r), flush=True)

trial("first", 64, 6)
trial("64 again", 64, 6)
trial("64 x 24", 64, 24)
trial("128", 128, 6)
trial("128 again", 128, 6)
trial("256", 256, 3)
trial("256 again", 256, 6)
trial("256 + set_image each step", 256, 6, set_image=True)
trial("256 x 20", 256, 20)
trial("64 after 256", 64, 6)
trial("64 after 0.5 s idle", 64, 6, sleep=0.5)
trial("64 after 0.5 s idle, no warm", 64, 6, warm=0, sleep=0.5)
trial("256 after 0.5 s idle, no warm", 256, 3, warm=0, sleep=0.5)
trial("64 x 100", 64, 100)
