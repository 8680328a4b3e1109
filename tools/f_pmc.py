"""One multi-feature convolve for counter collection: python3 tools/f_pmc.py F G  (4096^2 image, 127^2 kernels, 64 maps)"""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fc = importlib.import_module("cuda-fft-convolution_amd")
F, G = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0"); stream = torch.cuda.current_stream(dev); rng = np.random.default_rng(1)
H, k, n = 4096, 127, 64
img = torch.from_numpy(rng.random((F, H, H), dtype=np.float32)).to(dev)
ker = torch.from_numpy(rng.random((n, F, k, k), dtype=np.float32)).to(dev)
plan = fc.Plan(H, H, F, k, k, gpuId=0, stream=stream.cuda_stream, options=fc.PlanOptions(rows_group=G))
plan.set_image_device(img.data_ptr())
out = torch.empty((n, plan.info.fft_w, plan.info.fft_h), dtype=torch.float32, device=dev)
for _ in range(3): plan.convolve_packed_device(n, ker.data_ptr(), k, k, out.data_ptr())
torch.cuda.synchronize()
