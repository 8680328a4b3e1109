#!/usr/bin/env python3
"""Throughput over image sizes the BASELINE configs do not name: >= 20 square and 4 rectangular images from 300 to
8192 pixels, kernels of 31 / 63 / 127, 64 filters, device-resident steps as bench.py times them.  Reports
Gpixel-filters/s on the REFERENCE's pixel count (the ceil16 window of src/cudaConvolutionFFT.cu:103-110, whatever
transform length the plan chose) and the chosen (L_h, L_w): what a hole in the ladder of specialised lengths costs.
usage (GPU box): python tools/size_sweep.py [--quick] [--filters 64] > profiles/rNN_size_sweep.txt"""
import argparse
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SQUARE = [300, 384, 500, 640, 800, 1000, 1200, 1400, 1600, 1800, 2048, 2300, 2600, 2900, 3300, 3700, 4096, 4300, 4800, 5400, 6000, 6600,
          7200, 8192]
RECT = [(480, 640), (1080, 1920), (2160, 3840), (3000, 4000)]


def ceil16(n):
    return (n + 15) // 16 * 16


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--filters", type=int, default=64)
    ap.add_argument("--quick", action="store_true", help="K = 63 only")
    ap.add_argument("--kernels", type=int, nargs="*", default=None)
    args = ap.parse_args()
    import numpy as np
    import torch
    import util
    fc = util.load_package()
    dev = torch.device("cuda", 0)
    ks = args.kernels or ([63] if args.quick else [31, 63, 127])
    print("# image HxW, K, window, transform (L_h x L_w), specialised (1 rows | 2 columns), us per step of %d maps, Gpixel-filters/s on the window" % args.filters)
    worst = None
    worst_mid = None
    rng = np.random.default_rng(1)
    for (H, W) in [(s, s) for s in SQUARE] + RECT:
        for k in ks:
            if k >= min(H, W):
                continue
            n = args.filters
            fh, fw = ceil16(H + k - 1), ceil16(W + k - 1)
            if n * fh * fw * 4 > 24e9:             # keep the maps of one step under 24 GB
                n = max(8, int(24e9 / (fh * fw * 4)) // 8 * 8)
            img = torch.from_numpy(rng.random((1, W, H), dtype=np.float32)).to(dev)
            ker = torch.from_numpy(rng.random((n, 1, k, k), dtype=np.float32)).to(dev)
            with fc.Plan(H, W, 1, k, k, stream=torch.cuda.current_stream(dev).cuda_stream) as p:
                i = p.info
                out = torch.empty((n, i.fft_w, i.fft_h), dtype=torch.float32, device=dev)

                def step():
                    p.set_image_device(img.data_ptr())
                    p.convolve_packed_device(n, ker.data_ptr(), k, k, out.data_ptr())

                step()
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                step()
                torch.cuda.synchronize(dev)
                est = time.perf_counter() - t0
                reps = max(3, min(200, int(0.25 / max(est, 1e-5))))
                for _ in range(max(2, min(100, int(0.1 / max(est, 1e-5))))):      # clocks
                    step()
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(reps):
                    step()
                torch.cuda.synchronize(dev)
                dt = (time.perf_counter() - t0) / reps
                s_img = float(img.sum(dtype=torch.float64).item())
                want = ker.sum(dim=(1, 2, 3), dtype=torch.float64) * s_img
                got = out.sum(dim=(1, 2), dtype=torch.float64)
                err = float(((got - want).abs() / want.abs()).max().item())
                gpx = n * fh * fw / dt / 1e9
                spec = p.get_option("specialised_kernels")
                print("%5dx%-5d K=%-3d window %5dx%-5d transform %5dx%-5d spec %d  maps %3d  %9.1f us  %6.1f Gpx/s  overhead %.2fx  %s"
                      % (H, W, k, i.fft_h, i.fft_w, i.transform_h, i.transform_w, spec, n, dt * 1e6, gpx,
                         i.transform_h * i.transform_w / float(fh * fw), "ok" if err < 1e-5 else "CHECKSUM %.2g" % err), flush=True)
                if min(H, W) >= 1000 and (worst is None or gpx < worst[0]):
                    worst = (gpx, H, W, k)
                if 500 <= min(H, W) < 1000 and (worst_mid is None or gpx < worst_mid[0]):
                    worst_mid = (gpx, H, W, k)
            del img, ker, out
    if worst:
        print("# minimum over images of 1000 pixels and more: %.1f Gpx/s at %dx%d K=%d" % worst)
    if worst_mid:
        print("# minimum over images of 500 ... 999 pixels (64 maps per step: 80-190 us, launch latency shows): %.1f Gpx/s at %dx%d K=%d" % worst_mid)


if __name__ == "__main__":
    main()
