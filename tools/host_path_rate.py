"""PCIe-inclusive rate of the MEX-faithful host-in/host-out path (not the bench's `value`):
the one-shot entry (plan creation, pinned ring and fresh output arrays included) and the plan API
with caller buffers reused, streamed copy-out (default) against the blocking copy-out."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import util
fc = util.load_package()
for cfg, (H, W, kh, kw, n) in {"cfg2": (1024, 1024, 63, 63, 16), "cfg3 (64 of 256 kernels)": (4096, 4096, 127, 127, 64)}.items():
    img, ks = util.synth(3, H, W, 1, kh, kw, n)
    fc.cudaConvolutionFFT(img, kh, kw, ks[:1])          # warm-up (context, first-use allocations)
    t0 = time.perf_counter(); out = fc.cudaConvolutionFFT(img, kh, kw, ks); dt = time.perf_counter() - t0
    P = out[0].size
    print("%s one-shot: %d maps of %dx%d host->host in %.1f ms = %.2f Gpixel-filters/s (%.1f GB/s of maps over PCIe)"
          % (cfg, n, out[0].shape[0], out[0].shape[1], dt * 1e3, n * P / dt / 1e9, n * P * 4 / dt / 1e9), flush=True)
    ref = out
    with fc.Plan(H, W, 1, kh, kw) as plan:
        bufs = [np.ones(ref[0].shape, dtype=np.float32, order="F") for _ in range(n)]
        for label, opts in (("blocking", {"host_stream": 0}), ("direct, 2 threads", {"host_stream": 1}),
                            ("direct, 1 thread", {"host_threads": 1}), ("direct, 4 threads", {"host_threads": 4}),
                            ("pinned ring, 6 threads", {"host_stream": 2, "host_threads": 6}),
                            ("direct, 2 threads, batch 8", {"host_stream": 1, "host_threads": 2, "batch_maps": 8}),
                            ("blocking, batch 8", {"host_stream": 0, "batch_maps": 8})):
            for k, v in opts.items():
                plan.set_option(k, v)
            best = 1e9
            for rep in range(3):
                t0 = time.perf_counter()
                plan.set_image(img)
                plan.convolve(ks, out=bufs)
                best = min(best, time.perf_counter() - t0)
            ok = all(np.array_equal(a, b) for a, b in zip(bufs, ref))
            print("%s plan, %s: %.1f ms = %.2f Gpixel-filters/s (%.1f GB/s of maps), identical to one-shot: %s"
                  % (cfg, label, best * 1e3, n * P / best / 1e9, n * P * 4 / best / 1e9, ok), flush=True)
