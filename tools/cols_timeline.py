"""Per-phase wall-clock timeline of the output kernel's workgroup 0 (needs a library built with
-DFC_COLS_TIMELINE=1: FFTCONV_LIB=... python tools/cols_timeline.py).  Stamps (100 MHz clock):
0 tile start, 1 next gather issued, 2 after stage 3, 3 after stage 2, 4 after the pre-store wait,
5 after stage 1 + stores, 6 after landing."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
dbg = torch.zeros(16 * 8, dtype=torch.int64, device="cuda")
import util
fc = util.load_package()
# usage: cols_timeline.py [H K maps]  (default: cfg3; large sizes are forced to one pass)
H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
kh = kw = int(sys.argv[2]) if len(sys.argv) > 2 else 127
n = int(sys.argv[3]) if len(sys.argv) > 3 else 64
img = torch.rand((1, W, H), dtype=torch.float32, device="cuda")
ker = torch.rand((n, 1, kw, kh), dtype=torch.float32, device="cuda")
plan = fc.Plan(H, W, 1, kh, kw, stream=torch.cuda.current_stream().cuda_stream, options={"blockwise": 1})
print("image %d x %d, kernels %d x %d, %d maps: transform %d x %d" % (H, W, kh, kw, n, plan.info.transform_h, plan.info.transform_w))
out = torch.empty((n, plan.info.fft_w, plan.info.fft_h), dtype=torch.float32, device="cuda")
plan.set_option("timeline_ptr", dbg.data_ptr())   # exists in FC_*_TIMELINE builds only
for rep in range(3):
    plan.set_image_device(img.data_ptr())
    plan.convolve_packed_device(n, ker.data_ptr(), kh, kw, out.data_ptr())
torch.cuda.synchronize()
t = dbg.cpu().numpy().reshape(16, 8).astype(np.float64) / 100.0   # microseconds
names = ["issue gather", "stage 3 (C2)", "stage 2 (C3)", "wait for gather", "stage 1 + stores (C4)", "landing (C5)"]
print("tile   " + "  ".join("%22s" % s for s in names) + "   total")
for it in range(2, 14):
    d = [t[it, k + 1] - t[it, k] for k in range(6)]
    print("%4d   " % it + "  ".join("%22.2f" % x for x in d) + "   %.2f" % (t[it + 1, 0] - t[it, 0]))
plan.destroy()
