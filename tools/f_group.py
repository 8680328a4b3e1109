"""Row-kernel time per map for F > 1 against the walk length (maps per workgroup): fewer maps per walk =
more workgroups share an image-spectrum row on one XCD at the same time (L2 reuse)."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fc = importlib.import_module("cuda-fft-convolution_amd")
dev = torch.device("cuda:0"); stream = torch.cuda.current_stream(dev); rng = np.random.default_rng(1)
for (H, k, n, Fs) in ((4096, 127, 64, (2, 4, 8)), (2048, 63, 64, (4, 8, 32))):
    for F in Fs:
        img = torch.from_numpy(rng.random((F, H, H), dtype=np.float32)).to(dev)
        ker = torch.from_numpy(rng.random((n, F, k, k), dtype=np.float32)).to(dev)
        res = []
        for G in (1, 2, 3, 4, 6, 8, 16):
            plan = fc.Plan(H, H, F, k, k, gpuId=0, stream=stream.cuda_stream, options=fc.PlanOptions(rows_group=G))
            plan.set_image_device(img.data_ptr())
            out = torch.empty((n, plan.info.fft_w, plan.info.fft_h), dtype=torch.float32, device=dev)
            for _ in range(10): plan.convolve_packed_device(n, ker.data_ptr(), k, k, out.data_ptr())
            torch.cuda.synchronize(); plan.set_option("profile", 1); plan.profile(reset=True)
            for _ in range(4): plan.convolve_packed_device(n, ker.data_ptr(), k, k, out.data_ptr())
            torch.cuda.synchronize(); p = plan.profile(reset=True)
            res.append("G=%d: %.1f" % (G, p["spectral_rows"]["ms"] / p["spectral_rows"]["units"] * 1e3))
            plan.destroy(); del out
        print("%d^2 F %2d rows us per map  %s" % (H, F, "  ".join(res)), flush=True)
        del img, ker
