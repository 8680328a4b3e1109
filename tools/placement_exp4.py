"""The output kernel's two states (26.2 / 27.5 us per map): slide the map buffer through one 24-GB allocation
in large steps with everything else fixed."""
import ctypes, importlib, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fc = importlib.import_module("cuda-fft-convolution_amd")
dev = torch.device("cuda:0")
H = W = 4096; kh = kw = 127; F = 1; n = 64
rng = np.random.default_rng(1)
img = torch.from_numpy(rng.random((F, W, H), dtype=np.float32)).to(dev)
ker = torch.from_numpy(rng.random((n, F, kw, kh), dtype=np.float32)).to(dev)
stream = torch.cuda.current_stream(dev)
map_bytes = 4224 * 4224 * 4
plan = fc.Plan(H, W, F, kh, kw, gpuId=0, stream=stream.cuda_stream)
plan.set_image_device(img.data_ptr())
big = torch.empty(n * map_bytes + (20 << 30), dtype=torch.uint8, device=dev)

def trial(tag, out_ptr, reps=8, warm=12):
    for _ in range(warm):
        plan.convolve_packed_device(n, ker.data_ptr(), kh, kw, out_ptr)
    torch.cuda.synchronize()
    plan.set_option("profile", 1); plan.profile(reset=True)
    for _ in range(reps):
        plan.convolve_packed_device(n, ker.data_ptr(), kh, kw, out_ptr)
    torch.cuda.synchronize()
    p = plan.profile(reset=True); plan.set_option("profile", 0)
    r = {k: round(v["ms"] / max(1.0, v["units"]) * 1e3, 2) for k, v in p.items() if k in ("spectral_rows", "cols_c2r")}
    print("%-24s rows %.2f cols %.2f" % (tag, r["spectral_rows"], r["cols_c2r"]), flush=True)

trial("warm", big.data_ptr(), warm=20)
for k in range(0, 41):
    trial("+%5d MiB" % (k * 512), big.data_ptr() + (k * 512 << 20), warm=2)
for k in range(0, 17):
    trial("+%5d MiB" % (k * 64), big.data_ptr() + (k * 64 << 20), warm=2)
