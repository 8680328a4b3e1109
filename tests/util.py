"""Test helpers: package import, the CPU oracle (checker only), synthetic inputs."""
import ctypes
import importlib
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_NAME = "cuda-fft-convolution_amd"


def load_package():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    return importlib.import_module(PKG_NAME)


def ceil16(n):
    return (n + 15) // 16 * 16


def _build(path, target_dir, target=None):
    """make -C target_dir, one process at a time (pytest-xdist workers would otherwise run the same make side by side)"""
    import fcntl
    with open(os.path.join(target_dir, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            subprocess.run(["make", "-C", target_dir] + ([target] if target else []), check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    assert os.path.exists(path), path


def build_emu():
    d = os.path.join(ROOT, "tests", "emu")
    so = os.path.join(d, "libfftconv_emu.so")
    _build(so, d)
    return so


class Oracle:
    """ctypes view of oracle/liboracle.so -- the CPU restatement of the reference path.
    Used by tests / smoke / the bench's cpu_baseline leg only."""

    def __init__(self, native=False):
        """native=True: the same source built -march=native for THIS host (oracle/_native/, made on the spot; the bench's
        cpu_baseline leg only -- the tests check against the portable build)."""
        d = os.path.join(ROOT, "oracle")
        so = os.path.join(d, "_native", "liboracle.so") if native else os.path.join(d, "liboracle.so")
        if native or not os.path.exists(so):
            _build(so, d, "native" if native else None)
        self.lib = ctypes.CDLL(so)
        self.lib.oracle_num_threads.argtypes = [ctypes.c_int]

    def num_threads(self, threads=0):
        return self.lib.oracle_num_threads(threads)

    @staticmethod
    def _prep(data, kernels):
        d = np.asfortranarray(np.asarray(data, dtype=np.float32))
        if d.ndim == 2:
            d = np.asfortranarray(d[:, :, None])
        ks = []
        for k in kernels:
            k = np.asarray(k, dtype=np.float32)
            if k.ndim == 2:
                k = k[:, :, None]
            ks.append(np.asfortranarray(k))
        n = len(ks)
        kp = (ctypes.c_void_p * n)(*[k.ctypes.data for k in ks])
        kh = (ctypes.c_int * n)(*[k.shape[0] for k in ks])
        kw = (ctypes.c_int * n)(*[k.shape[1] for k in ks])
        return d, ks, n, kp, kh, kw

    def conv_fft(self, data, mkh, mkw, kernels, threads=0, f64=False):
        d, ks, n, kp, kh, kw = self._prep(data, kernels)
        H, W, F = d.shape
        fh, fw = ceil16(H + mkh - 1), ceil16(W + mkw - 1)
        dt = np.float64 if f64 else np.float32
        outs = [np.zeros((fh, fw), dtype=dt, order="F") for _ in range(n)]
        op = (ctypes.c_void_p * n)(*[o.ctypes.data for o in outs])
        fn = self.lib.oracle_conv_fft_f64 if f64 else self.lib.oracle_conv_fft
        rc = fn(ctypes.c_void_p(d.ctypes.data), H, W, F, mkh, mkw, n, kp, kh, kw, op, threads)
        if rc != 0:
            raise ValueError("oracle_conv_fft rc=%d" % rc)
        return outs

    def conv_direct(self, data, mkh, mkw, kernel):
        d, ks, n, kp, kh, kw = self._prep(data, [kernel])
        H, W, F = d.shape
        fh, fw = ceil16(H + mkh - 1), ceil16(W + mkw - 1)
        out = np.zeros((fh, fw), dtype=np.float64, order="F")
        rc = self.lib.oracle_conv_direct(ctypes.c_void_p(d.ctypes.data), H, W, F, mkh, mkw,
                                         ctypes.c_void_p(ks[0].ctypes.data), ks[0].shape[0], ks[0].shape[1],
                                         ctypes.c_void_p(out.ctypes.data))
        if rc != 0:
            raise ValueError("oracle_conv_direct rc=%d" % rc)
        return out


class CpuF32:
    """ctypes view of oracle/libcpu_f32.so: the fp32 half-spectrum CPU restatement (second CPU
    baseline of SURVEY.md 8(d)); same contract as Oracle.conv_fft."""

    def __init__(self, native=False):
        d = os.path.join(ROOT, "oracle")
        so = os.path.join(d, "_native", "libcpu_f32.so") if native else os.path.join(d, "libcpu_f32.so")
        if native or not os.path.exists(so):
            _build(so, d, "native" if native else None)
        self.lib = ctypes.CDLL(so)

    def conv_fft(self, data, mkh, mkw, kernels, threads=0):
        d, ks, n, kp, kh, kw = Oracle._prep(data, kernels)
        H, W, F = d.shape
        fh, fw = ceil16(H + mkh - 1), ceil16(W + mkw - 1)
        outs = [np.zeros((fh, fw), dtype=np.float32, order="F") for _ in range(n)]
        op = (ctypes.c_void_p * n)(*[o.ctypes.data for o in outs])
        rc = self.lib.cpu_f32_conv_fft(ctypes.c_void_p(d.ctypes.data), H, W, F, mkh, mkw, n, kp, kh, kw, op, threads)
        if rc != 0:
            raise ValueError("cpu_f32_conv_fft rc=%d" % rc)
        return outs


def numpy_fft_conv(data, mkh, mkw, kernels):
    """Independent float64 statement of demoCudaConvolutionFFT.m:78-102 with NumPy's pocketfft."""
    data = np.asarray(data, dtype=np.float64)
    if data.ndim == 2:
        data = data[:, :, None]
    H, W, F = data.shape
    fh, fw = ceil16(H + mkh - 1), ceil16(W + mkw - 1)
    D = np.fft.fft2(data, s=(fh, fw), axes=(0, 1))
    res = []
    for k in kernels:
        k = np.asarray(k, dtype=np.float64)
        if k.ndim == 2:
            k = k[:, :, None]
        K = np.fft.fft2(k, s=(fh, fw), axes=(0, 1))
        res.append(np.real(np.fft.ifft2(D * K, axes=(0, 1))).sum(axis=2))
    return res


def rel_err(a, b):
    """max|a-b| / max|b| -- the norm-relative parity metric of SURVEY.md 8(d)."""
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(np.asarray(a, dtype=np.float64) - b).max() / max(np.abs(b).max(), 1e-300))


def synth(cfg_seed, H, W, F, kh, kw, n):
    """Synthetic inputs of SURVEY.md 8(d): image U[0,1) seed 1234+cfg, kernel k U[0,1) seed 5678+cfg+k."""
    img = np.random.default_rng(1234 + cfg_seed).random((H, W, F), dtype=np.float32)
    ks = [np.random.default_rng(5678 + cfg_seed + k).random((kh, kw, F), dtype=np.float32) for k in range(n)]
    return np.asfortranarray(img), [np.asfortranarray(k) for k in ks]
