"""PCIe-inclusive rate of the MEX-faithful host-in/host-out entry (not the bench's `value`)."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import util
fc = util.load_package()
for cfg, (H, W, kh, kw, n) in {"cfg2": (1024, 1024, 63, 63, 16), "cfg3 (16 of 256 kernels)": (4096, 4096, 127, 127, 16)}.items():
    img, ks = util.synth(3, H, W, 1, kh, kw, n)
    fc.cudaConvolutionFFT(img, kh, kw, ks[:1])          # warm-up (context, first-use allocations)
    t0 = time.perf_counter(); out = fc.cudaConvolutionFFT(img, kh, kw, ks); dt = time.perf_counter() - t0
    P = out[0].size
    print("%s: %d maps of %dx%d host->host in %.1f ms = %.2f Gpixel-filters/s (%.1f GB/s of maps over PCIe)"
          % (cfg, n, out[0].shape[0], out[0].shape[1], dt * 1e3, n * P / dt / 1e9, n * P * 4 / dt / 1e9))
