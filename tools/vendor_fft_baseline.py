#!/usr/bin/env python3
"""The reference's algorithm as a straight port onto the VENDOR's FFT library of this device (rocFFT behind hipFFT, through torch.fft):
what a hipify-style port of src/cudaConvolutionFFT.cu would run -- pad every kernel to the full FFT_H x FFT_W plane, R2C, multiply with the
image spectrum, C2R, everything device-resident -- timed per map beside the engine of this repository on the same GPU.  A measurement
for DESIGN.md (profiles/r05r_vendor_fft_baseline.txt); the library itself never links or calls hipFFT / rocFFT.
usage: vendor_fft_baseline.py [cfg2|cfg3|cfg4|cfg5] [maps]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, util
fc = util.load_package()
CFG = {"cfg2": (1024, 63, 16), "cfg3": (4096, 127, 64), "cfg4": (4096, 63, 64), "cfg5": (2048, 63, 64)}
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
N, K, n = CFG[name]
if len(sys.argv) > 2: n = int(sys.argv[2])
dev = torch.device("cuda", 0)
fh = fw = util.ceil16(N + K - 1)
g = torch.Generator(device="cpu").manual_seed(5)
img = torch.rand((1, N, N), generator=g, dtype=torch.float32).to(dev)          # [F][W][H]
ker = torch.rand((n, 1, K, K), generator=g, dtype=torch.float32).to(dev)       # [n][F][kw][kh]
P = fh * fw

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

# (1) one kernel at a time, as the reference's loop (src/cudaConvolutionFFT.cu:204-291)
out = torch.empty((n, fw, fh), dtype=torch.float32, device=dev)
pad_img = torch.zeros((1, fw, fh), dtype=torch.float32, device=dev); pad_img[:, :N, :N] = img
kp = torch.zeros((1, fw, fh), dtype=torch.float32, device=dev)
def loop():
    D = torch.fft.rfft2(pad_img)
    for j in range(n):
        kp.zero_(); kp[:, :K, :K] = ker[j]
        out[j] = torch.fft.irfft2(D * torch.fft.rfft2(kp), s=(fw, fh))[0]
t_loop = timed(loop, 3)
# (2) the same with the kernels batched 8 at a time through one batched plan (what a careful port would do)
B = min(8, n)
kpb = torch.zeros((B, fw, fh), dtype=torch.float32, device=dev)
def batched():
    D = torch.fft.rfft2(pad_img)
    for j in range(0, n, B):
        kpb.zero_(); kpb[:, :K, :K] = ker[j:j + B, 0]
        out[j:j + B] = torch.fft.irfft2(D * torch.fft.rfft2(kpb), s=(fw, fh))
t_b = timed(batched, 3)
ref = out[0].clone()
# (3) this repository's engine on the same problem
with fc.Plan(N, N, 1, K, K, stream=torch.cuda.current_stream().cuda_stream) as p:
    o2 = torch.empty((n, fw, fh), dtype=torch.float32, device=dev)
    def ours():
        p.set_image_device(img.data_ptr()); p.convolve_packed_device(n, ker.data_ptr(), K, K, o2.data_ptr())
    for _ in range(10): ours()
    t_o = timed(ours, 10)
    err = float((o2[0] - ref).abs().max() / ref.abs().max())
print("%s: %dx%d image, %d kernels of %dx%d, maps of %dx%d, device-resident, fp32" % (name, N, N, n, K, K, fh, fw))
print("  vendor FFT library, one kernel at a time (the reference's loop):  %8.1f us per map  %7.1f Gpixel-filters/s" % (t_loop / n * 1e6, n * P / t_loop / 1e9))
print("  vendor FFT library, kernels batched %d at a time:                 %8.1f us per map  %7.1f Gpixel-filters/s" % (B, t_b / n * 1e6, n * P / t_b / 1e9))
print("  this engine (hand-written kernels, padding never materialised):  %8.1f us per map  %7.1f Gpixel-filters/s   (%.1f x the batched port; maps agree to %.1e)" % (t_o / n * 1e6, n * P / t_o / 1e9, t_b / t_o, err))
