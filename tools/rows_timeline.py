"""Per-phase wall-clock timeline of one workgroup of the multi-map row kernel (needs a library built
with -DFC_ROWS_TIMELINE=1: FFTCONV_LIB=... python tools/rows_timeline.py).  Stamps (100 MHz clock)
per map of the walk: 0 start, 1 after P1, 2 after P2, 3 after P3, 4 after P4, 5 after P5."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
dbg = torch.zeros(4096, dtype=torch.int64, device="cuda")    # stamps [16][8]; behind them the counters of -DFC_ROWS_STAGGER_TICKS builds
import util
fc = util.load_package()
H = W = 4096; kh = kw = 127; n = 64
img = torch.rand((1, W, H), dtype=torch.float32, device="cuda")
ker = torch.rand((n, 1, kw, kh), dtype=torch.float32, device="cuda")
plan = fc.Plan(H, W, 1, kh, kw, stream=torch.cuda.current_stream().cuda_stream)
out = torch.empty((n, plan.info.fft_w, plan.info.fft_h), dtype=torch.float32, device="cuda")
plan.set_option("timeline_ptr", dbg.data_ptr())   # exists in FC_*_TIMELINE builds only
for rep in range(int(os.environ.get("STEPS", "3"))):      # STEPS=80: the stamped launch is one of a warm process (clocks settled)
    plan.set_image_device(img.data_ptr())
    plan.convolve_packed_device(n, ker.data_ptr(), kh, kw, out.data_ptr())
torch.cuda.synchronize()
raw = dbg.cpu().numpy()[:128].reshape(16, 8)
t = raw.astype(np.float64) / 100.0
names = ["P1 fwd stage 1", "P2 fwd stage 2", "P3 stage 3 x S", "P4 inv stage 2", "P5 inv stage 1 + stores"]
print("map    " + "  ".join("%24s" % s for s in names) + "   total")
for m in range(1, 15):
    d = [t[m, k + 1] - t[m, k] for k in range(5)]
    mhz = (float(raw[m + 1, 6] - raw[m, 6]) / max(1.0, float(raw[m + 1, 0] - raw[m, 0])) * 100.0) if raw[m, 6] else 0.0
    print("%4d   " % m + "  ".join("%24.2f" % x for x in d) + "   %.2f" % (t[m + 1, 0] - t[m, 0]) + ("   shader clock %.0f MHz" % mhz if mhz else ""))
plan.destroy()
