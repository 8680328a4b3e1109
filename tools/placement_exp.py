"""Does the placement of the map / intermediate buffers move the two hot kernels?  One process, cfg3
geometry, 64 maps per trial; prints the buffers' addresses beside the per-map kernel times.
usage (GPU box): python3 tools/placement_exp.py"""
import importlib, os, sys
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

def trial(tag, plan, out_ptr, reps=6):
    for _ in range(2):
        plan.convolve_packed_device(n, ker.data_ptr(), kh, kw, out_ptr)
    torch.cuda.synchronize()
    plan.set_option("profile", 1); plan.profile(reset=True)
    for _ in range(reps):
        plan.convolve_packed_device(n, ker.data_ptr(), kh, kw, out_ptr)
    torch.cuda.synchronize()
    p = plan.profile(reset=True); plan.set_option("profile", 0)
    r = {k: round(v["ms"] / max(1.0, v["units"]) * 1e3, 2) for k, v in p.items() if k in ("spectral_rows", "cols_c2r")}
    print("%-28s out %#x (mod 2M %#8x, mod 1G %#10x)  rows %.2f cols %.2f" % (tag, out_ptr, out_ptr % (2 << 20), out_ptr % (1 << 30), r["spectral_rows"], r["cols_c2r"]), flush=True)

def new_plan():
    plan = fc.Plan(H, W, F, kh, kw, gpuId=0, stream=stream.cuda_stream)
    plan.set_image_device(img.data_ptr())
    return plan

map_bytes = 4224 * 4224 * 4
plan = new_plan()
base = torch.empty(n * map_bytes + (64 << 20), dtype=torch.uint8, device=dev)
print("# A: same plan, same buffers, repeated")
for i in range(4):
    trial("A%d" % i, plan, base.data_ptr())
print("# B: same plan, map buffer shifted")
for off in (256, 4096, 65536, 1 << 20, 2 << 20, 3 << 20, 8 << 20, 32 << 20, (32 << 20) + 4096):
    trial("B +%d" % off, plan, base.data_ptr() + off)
print("# C: fresh plan (fresh intermediate) per trial, same map buffer; a spacer of odd size allocated in between")
spacers = []
for i in range(8):
    plan.destroy() if hasattr(plan, "destroy") else None
    spacers.append(torch.empty(((i * 37) % 11 + 1) * (3 << 20) + 4096 * i, dtype=torch.uint8, device=dev))
    hold = fc.Plan(512, 512, 1, 15, 15, gpuId=0, stream=stream.cuda_stream) if i % 2 else None
    plan = new_plan()
    trial("C%d" % i, plan, base.data_ptr())
print("# D: fresh map buffer per trial (torch allocator), same plan")
keep = []
for i in range(6):
    keep.append(torch.empty(((i * 53) % 7 + 1) * (5 << 20) + 512 * i, dtype=torch.uint8, device=dev))
    b = torch.empty(n * map_bytes, dtype=torch.uint8, device=dev); keep.append(b)
    trial("D%d" % i, plan, b.data_ptr())
