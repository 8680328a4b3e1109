"""Where the one-shot entry's time goes (plan creation, image upload, convolve into fresh or resident buffers)."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import util
print(open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip(), os.cpu_count(), len(os.sched_getaffinity(0)))
fc = util.load_package()
def T(f, reps=3):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); best = min(best, time.perf_counter() - t0)
    return best * 1e3, r
for cfg, (H, W, kh, kw, n) in {"cfg1": (256, 256, 31, 31, 1), "cfg2": (1024, 1024, 63, 63, 16), "cfg3/32": (4096, 4096, 127, 127, 32)}.items():
    img, ks = util.synth(3, H, W, 1, kh, kw, n)
    fc.cudaConvolutionFFT(img, kh, kw, ks[:1])
    t_one, out = T(lambda: fc.cudaConvolutionFFT(img, kh, kw, ks))
    t_create, plan = T(lambda: fc.Plan(H, W, 1, kh, kw), reps=1)
    t_img, _ = T(lambda: plan.set_image(img))
    t_fresh, _ = T(lambda: plan.convolve(ks), reps=1)
    t_fresh2, _ = T(lambda: plan.convolve(ks), reps=2)
    bufs = [np.ones(out[0].shape, dtype=np.float32, order="F") for _ in range(n)]
    t_res, _ = T(lambda: plan.convolve(ks, out=bufs))
    for th in (1, 2, 4, 8, 12):
        plan.set_option("host_threads", th)
        plan.convolve(ks[:2])
        t_th, _ = T(lambda: plan.convolve(ks), reps=2)
        print("   fresh out, %d host threads: %.1f ms" % (th, t_th), flush=True)
    t_destroy, _ = T(lambda: plan.destroy(), reps=1)
    t_alloc, _ = T(lambda: [np.empty(out[0].shape, dtype=np.float32, order="F") for _ in range(n)])
    print("%s: one-shot %.1f ms | plan create %.1f, set_image %.1f, convolve first (fresh out) %.1f, again (fresh out) %.1f, "
          "resident out %.1f, destroy %.1f, np.empty %.2f" % (cfg, t_one, t_create, t_img, t_fresh, t_fresh2, t_res, t_destroy, t_alloc), flush=True)
