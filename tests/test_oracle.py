"""CPU tier: pins the oracle (oracle/fftconv_oracle.c) -- the reference ships no tests or golden
vectors, so the oracle is pinned against the committed NumPy-float64 fixtures, against NumPy's
fft2/ifft2 on fresh inputs, against brute-force convolution and against the invariants the
reference's demo script encodes (demoCudaConvolutionFFT.m)."""
import numpy as np
import pytest

import golden_util
import util


@pytest.mark.parametrize("n,want", [(1, 16), (15, 16), (16, 16), (17, 32), (73, 80), (286, 288), (1086, 1088),
                                    (4222, 4224), (4158, 4160), (2110, 2112), (0, 0)])
def test_fft_size16(oracle, n, want):
    # computeFFTsize16, src/cudaConvFFTData.h:96-102
    assert oracle.lib.oracle_fft_size16(n) == want


@pytest.mark.parametrize("case", golden_util.golden_cases())
def test_oracle_matches_golden(oracle, case):
    data, mkh, mkw, kernels, expect = golden_util.load_case(case)
    got = oracle.conv_fft(data, mkh, mkw, kernels, f64=True)
    for g, e in zip(got, expect):
        assert g.shape == e.shape
        assert util.rel_err(g, e) < 1e-12
    got32 = oracle.conv_fft(data, mkh, mkw, kernels)
    for g, e in zip(got32, expect):
        assert util.rel_err(g, e) < 5e-7   # float32 rounding of the stored result only


@pytest.mark.parametrize("shape", [(64, 8, 5, 10, 4), (33, 47, 3, 7, 5), (100, 90, 2, 13, 17), (256, 256, 1, 31, 31),
                                   (37, 53, 1, 37, 53), (20, 300, 2, 3, 40)])
def test_oracle_vs_numpy(oracle, shape):
    H, W, F, kh, kw = shape
    rng = np.random.default_rng(H * 1000 + W)
    data = rng.standard_normal((H, W, F)).astype(np.float32)
    ks = [rng.standard_normal((kh, kw, F)).astype(np.float32) for _ in range(2)]
    got = oracle.conv_fft(data, kh, kw, ks, f64=True)
    ref = util.numpy_fft_conv(data, kh, kw, ks)
    for g, r in zip(got, ref):
        assert util.rel_err(g, r) < 1e-12


def test_oracle_vs_direct_including_wraparound(oracle):
    rng = np.random.default_rng(5)
    data = rng.random((30, 22, 2)).astype(np.float32)
    k_small = rng.random((5, 4, 2)).astype(np.float32)
    k_big = rng.random((20, 30, 2)).astype(np.float32)   # > MAXK: wraps modulo the 48x32 window
    for k in (k_small, k_big):
        got = oracle.conv_fft(data, 5, 4, [k], f64=True)[0]
        ref = oracle.conv_direct(data, 5, 4, k)
        assert util.rel_err(got, ref) < 1e-12


def test_oracle_rejects_kernel_larger_than_window(oracle):
    data = np.zeros((10, 10, 1), np.float32)
    with pytest.raises(ValueError):
        oracle.conv_fft(data, 3, 3, [np.zeros((17, 3, 1), np.float32)])   # src/cudaConvolutionFFT.cu:242


def test_demo_invariants(oracle):
    """demoCudaConvolutionFFT.m: cvcell{1} == cvcell{3}; cvcell{2} - cvcell{1} is the
    (100 - kernel(1)) * data(:,:,1) image at the top-left (:110-113); the planted templates give
    the response peak of the first channel's auto-correlation at the planted offsets (:57-69)."""
    data, cn, cm, ks, expect = golden_util.load_case("case_demo")
    n, m, _ = data.shape
    got = oracle.conv_fft(data, cn, cm, ks, f64=True)
    assert np.array_equal(got[0], got[2])
    diff = got[1] - got[0]
    want = np.zeros_like(diff)
    want[:n, :m] = (100.0 - float(ks[0][0, 0, 0])) * data[:, :, 0].astype(np.float64)
    assert np.abs(diff - want).max() < 1e-9
    # outside the linear-convolution support the window is zero (up to round-off)
    assert np.abs(got[0][n + cn - 1:, :]).max() < 1e-9 and np.abs(got[0][:, m + cm - 1:]).max() < 1e-9
    # cropped part equals sum_i conv2(data_i, kernel_i)  (:91-96,149-155)
    assert util.rel_err(got[0][:n + cn - 1, :m + cm - 1], expect[0][:n + cn - 1, :m + cm - 1]) < 1e-12


def test_demo_planted_template_peaks(oracle):
    """demoCudaConvolutionFFT.m:57-69: response peaks at the planted offsets shifted by (cn-1, cm-1),
    peak value sum(template^2) = 22140 exactly (float64)"""
    data, cn, cm, ks, _ = golden_util.load_case("case_demo")
    golden_util.demo_planted_checks(lambda kernels: oracle.conv_fft(data, cn, cm, kernels, f64=True), data, cn, cm, ks, 1e-12)
    # and the fp32 half-spectrum CPU restatement (second CPU baseline) on the same structure
    c32 = util.CpuF32()
    golden_util.demo_planted_checks(lambda kernels: c32.conv_fft(data, cn, cm, kernels), data, cn, cm, ks, 1e-5)


@pytest.mark.parametrize("shape", [(64, 8, 5, 10, 4, 3), (100, 90, 2, 7, 9, 2), (272, 300, 3, 17, 33, 2), (1024, 1024, 1, 63, 63, 1)])
def test_cpu_f32_baseline_matches_oracle(oracle, shape):
    """oracle/fftconv_cpu_f32.cpp (fp32, half spectra: the second CPU baseline bench.py times) against
    the float64 oracle, ragged cell included"""
    H, W, F, kh, kw, n = shape
    img, ks = util.synth(5, H, W, F, kh, kw, n)
    ks[-1] = np.asfortranarray(ks[-1][:max(1, kh - 2), :max(1, kw - 3), :])
    got = util.CpuF32().conv_fft(img, kh, kw, ks)
    for g, r in zip(got, oracle.conv_fft(img, kh, kw, ks)):
        assert util.rel_err(g, r) < 1e-5
