#!/usr/bin/env python3
"""The reference's demo (demoCudaConvolutionFFT.m) as a Python caller of this library: the same
experiment set-up (:37-69), the same two CPU references -- conv2 per channel (:91-96) and
fft2 / .* / ifft2 (:78-102) -- the same call (:110-129: three kernels in a cell, thread sizes,
0-based GPU id), and, where the script draws figures (:137-155), the residuals as numbers.
Returns / prints the figures' content:  max|cvg(1:n+cn-1, 1:m+cm-1) - cvmatlab| (figure 4) etc.

    python examples/demo_cuda_convolution_fft.py            (needs an MI355X; exits 1 on a mismatch)
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def conv2_full(a, b):
    """MATLAB conv2(a, b): full 2-D linear convolution, float64"""
    n, m = a.shape
    cn, cm = b.shape
    out = np.zeros((n + cn - 1, m + cm - 1))
    for y in range(cn):
        for x in range(cm):
            out[y:y + n, x:x + m] += b[y, x] * a
    return out


def main(seed=0, gpu_id=0):
    fc = importlib.import_module("cuda-fft-convolution_amd")
    rng = np.random.default_rng(seed)
    # -- experiment set-up (:37-61)
    n, m, k, cn, cm = 64, 8, 5, 10, 4
    data = rng.random((n, m, k)).astype(np.float32)
    kernel = np.zeros((cn, cm, k), dtype=np.float32)
    kernel[:, :, 0] = np.arange(1, cn * cm + 1, dtype=np.float32).reshape((cn, cm), order="F")
    for i in range(1, k):
        kernel[:, :, i] = rng.random((cn, cm)).astype(np.float32)
    data[4:4 + cn, 1:1 + cm, 0] = kernel[:, :, 0]
    data[20:20 + cn, 0:cm, 1] = kernel[:, :, 0]
    data[0:cn, m - cm:m, k - 1] = kernel[:, :, 0]
    kernel[:, :, k - 1] = kernel[:, :, 0]
    # -- flip kernel (required) (:63-69)
    kernel = np.ascontiguousarray(kernel[::-1, ::-1, :])
    # -- MATLAB convolution, conv2 and FFT versions (:75-102)
    fft_h, fft_w = 80, 16
    F_data = np.fft.fft2(data.astype(np.float64), s=(fft_h, fft_w), axes=(0, 1))
    F_kernel = np.fft.fft2(kernel.astype(np.float64), s=(fft_h, fft_w), axes=(0, 1))
    cvmatlab = sum(conv2_full(data[:, :, i].astype(np.float64), kernel[:, :, i].astype(np.float64)) for i in range(k))
    mat_fft_conv = np.real(np.fft.ifft2(F_kernel * F_data, axes=(0, 1)).sum(axis=2))
    # -- convolution using the GPU (:106-131)
    kernel2 = kernel.copy()
    kernel2[0, 0, 0] = 100.0
    kernel_cell = [kernel, kernel2, kernel]
    threads_per_block_in = [8, 8, 8, 16]
    cvcell = fc.cudaConvolutionFFT(data, cn, cm, kernel_cell, threads_per_block_in, gpu_id)
    cvg, cvg2 = cvcell[0], cvcell[1]
    # -- comparison (:137-155), as numbers
    scale = np.abs(cvmatlab).max()
    res = {
        "window": cvg.shape,                                                              # figure 3, left: the padded window
        "fig4 max|cvg(1:n+cn-1,1:m+cm-1) - cvmatlab| / max|cvmatlab|": float(np.abs(cvg[:n + cn - 1, :m + cm - 1] - cvmatlab).max() / scale),
        "fig1 max|cvg - fft2/ifft2 path| / max": float(np.abs(cvg - mat_fft_conv).max() / scale),
        "cvcell{1} == cvcell{3}": bool(np.array_equal(cvcell[0], cvcell[2])),
        "max|(cvg2 - cvg) - (100 - kernel(1)) * data(:,:,1) at the top-left| / max":
            float(np.abs((cvg2.astype(np.float64) - cvg)[:n, :m] - (100.0 - float(kernel[0, 0, 0])) * data[:, :, 0]).max()
                  / (100.0 * np.abs(data[:, :, 0]).max())),
        "argmax(cvg)": tuple(int(v) for v in np.unravel_index(np.argmax(cvg), cvg.shape)),   # planted template of channel 1: (5,2)+(cn-1,cm-1) 1-based
    }
    ok = (res["window"] == (fft_h, fft_w) and res["fig4 max|cvg(1:n+cn-1,1:m+cm-1) - cvmatlab| / max|cvmatlab|"] < 1e-4
          and res["fig1 max|cvg - fft2/ifft2 path| / max"] < 1e-4 and res["cvcell{1} == cvcell{3}"]
          and res["max|(cvg2 - cvg) - (100 - kernel(1)) * data(:,:,1) at the top-left| / max"] < 1e-4
          and res["argmax(cvg)"] in ((13, 4), (9, 7)))
    return ok, res


if __name__ == "__main__":
    ok, res = main()
    for key, val in res.items():
        print("%-78s %s" % (key, val))
    print("OK" if ok else "MISMATCH")
    sys.exit(0 if ok else 1)
