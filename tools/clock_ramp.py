"""How long does the GPU take to reach its sustained rate after an idle gap?  Per-step times (HIP events)
of 64-map cfg3 steps after 0.5 s of idle, and after a torch reduction like the bench's self-check."""
import importlib, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fc = importlib.import_module("cuda-fft-convolution_amd")
dev = torch.device("cuda:0")
H = W = 4096; kh = kw = 127; F = 1; N = 64
rng = np.random.default_rng(1)
img = torch.from_numpy(rng.random((F, W, H), dtype=np.float32)).to(dev)
ker = torch.from_numpy(rng.random((N, F, kw, kh), dtype=np.float32)).to(dev)
stream = torch.cuda.current_stream(dev)
plan = fc.Plan(H, W, F, kh, kw, gpuId=0, stream=stream.cuda_stream)
plan.set_image_device(img.data_ptr())
out = torch.empty((N, 4224, 4224), dtype=torch.float32, device=dev)

def series(tag, steps=60):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    ev[0].record()
    for i in range(steps):
        plan.convolve_packed_device(N, ker.data_ptr(), kh, kw, out.data_ptr())
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]
    cum = np.cumsum(ms)
    print(tag)
    print("  step ms:", " ".join("%.2f" % m for m in ms))
    print("  elapsed at step 5/10/20/40: %.0f %.0f %.0f %.0f ms" % (cum[4], cum[9], cum[19], cum[39]), flush=True)

series("cold (right after plan creation)")
series("hot (back to back)")
time.sleep(0.5)
series("after 0.5 s idle")
time.sleep(0.05)
series("after 50 ms idle")
time.sleep(0.005)
series("after 5 ms idle")
s = out.sum(dim=(1, 2), dtype=torch.float64); torch.cuda.synchronize()
series("after a torch reduction over the maps")
