# row-kernel time per map at F = 4 / 8 (cfg3 geometry) and F = 8 / 32 (cfg5 geometry) for alternate builds
for lib in "$@"; do
  echo "== $lib"
  FFTCONV_LIB=$PWD/$lib python3 - <<'PY'
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
fc = importlib.import_module("cuda-fft-convolution_amd")
dev = torch.device("cuda:0"); stream = torch.cuda.current_stream(dev); rng = np.random.default_rng(1)
for (H, k, n, Fs) in ((4096, 127, 64, (4, 8)), (2048, 63, 64, (8, 32))):
    for F in Fs:
        img = torch.from_numpy(rng.random((F, H, H), dtype=np.float32)).to(dev)
        ker = torch.from_numpy(rng.random((n, F, k, k), dtype=np.float32)).to(dev)
        plan = fc.Plan(H, H, F, k, k, gpuId=0, stream=stream.cuda_stream)
        plan.set_image_device(img.data_ptr())
        out = torch.empty((n, plan.info.fft_w, plan.info.fft_h), dtype=torch.float32, device=dev)
        for _ in range(12): plan.convolve_packed_device(n, ker.data_ptr(), k, k, out.data_ptr())
        torch.cuda.synchronize(); plan.set_option("profile", 1); plan.profile(reset=True)
        for _ in range(4): plan.convolve_packed_device(n, ker.data_ptr(), k, k, out.data_ptr())
        torch.cuda.synchronize(); p = plan.profile(reset=True)
        print("  %d^2 F %2d rows %.2f us per map" % (H, F, p["spectral_rows"]["ms"] / p["spectral_rows"]["units"] * 1e3), flush=True)
        plan.destroy(); del img, ker, out
PY
done
