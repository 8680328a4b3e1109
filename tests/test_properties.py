"""Property tests (hypothesis) of the host logic: filter / image sharding, the planner's transform
lengths, the window sizing of the reference (src/cudaConvFFTData.h:67-102), and linearity /
shift structure of the CPU oracle on random small problems."""
import ctypes
import importlib
import os
import subprocess

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

import util

mg = importlib.import_module("cuda-fft-convolution_amd.multi_gpu")


@pytest.fixture(scope="module")
def emu():
    d = os.path.join(util.ROOT, "tests", "emu")
    subprocess.run(["make", "-C", d], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return ctypes.CDLL(os.path.join(d, "libfftconv_emu.so"))


@given(n=st.integers(0, 5000), world=st.integers(1, 16))
def test_filter_shard_is_a_partition_into_near_equal_contiguous_blocks(n, world):
    blocks = [mg.filter_shard(n, r, world) for r in range(world)]
    pos = 0
    for first, count in blocks:
        assert first == pos and count >= 0
        pos += count
    assert pos == n
    counts = [c for _, c in blocks]
    assert max(counts) - min(counts) <= 1 and counts == sorted(counts, reverse=True)


@given(n=st.integers(1, 20000))
def test_window_sizes_of_the_reference(n):
    fc = util.load_package()
    s16, p2 = fc.fft_size16(n), fc.fft_size_pow2(n)
    assert s16 % 16 == 0 and n <= s16 < n + 16                       # computeFFTsize16
    assert p2 & (p2 - 1) == 0 and p2 >= s16 and (p2 == s16 or p2 < 2 * s16)   # computeFFTsize: align to 16, then to 2^k


@settings(max_examples=60, deadline=None)
@given(H=st.integers(1, 5000), W=st.integers(1, 5000), kh=st.integers(1, 200), kw=st.integers(1, 200), mode=st.sampled_from([0, 2]))
def test_planned_transform_lengths_cover_the_linear_convolution(emu, H, W, kh, kw, mode):
    emu.emu_set_tuning(mode, -1)
    lh, lw = ctypes.c_int(0), ctypes.c_int(0)
    rc = emu.emu_plan_lengths(H, W, 1, kh, kw, ctypes.byref(lh), ctypes.byref(lw))
    emu.emu_set_tuning(2, -1)
    assert rc == 0
    assert lh.value >= H + kh - 1 and lw.value >= W + kw - 1          # no wrap-around inside the support
    assert lh.value % 2 == 0                                         # real transform along h: even length
    assert emu.emu_length_supported(lh.value // 2) and emu.emu_length_supported(lw.value)
    assert lh.value <= 2 * (H + kh - 1) + 32 and lw.value <= 2 * (W + kw - 1) + 32


@settings(max_examples=25, deadline=None)
@given(H=st.integers(2, 24), W=st.integers(2, 24), F=st.integers(1, 3), kh=st.integers(1, 6), kw=st.integers(1, 6),
       dy=st.integers(0, 5), dx=st.integers(0, 5), seed=st.integers(0, 10 ** 6))
def test_oracle_linearity_and_delta_shift(oracle, H, W, F, kh, kw, dy, dx, seed):
    """conv is linear in the kernel, and a delta kernel at (dy, dx) in channel f shifts data(:,:,f)"""
    rng = np.random.default_rng(seed)
    data = rng.standard_normal((H, W, F)).astype(np.float32)
    k1 = rng.standard_normal((kh, kw, F)).astype(np.float32)
    k2 = rng.standard_normal((kh, kw, F)).astype(np.float32)
    a, b = 1.5, -0.25
    o1, o2, o3 = oracle.conv_fft(data, kh, kw, [k1, k2, (a * k1 + b * k2).astype(np.float32)], f64=True)
    assert np.abs(o3 - (a * o1 + b * o2)).max() <= 1e-5 * max(1.0, np.abs(o3).max())
    dy, dx = dy % kh, dx % kw
    delta = np.zeros((kh, kw, F), np.float32)
    delta[dy, dx, F - 1] = 1.0
    od = oracle.conv_fft(data, kh, kw, [delta], f64=True)[0]
    want = np.zeros_like(od)
    want[dy:dy + H, dx:dx + W] = data[:, :, F - 1]
    assert np.abs(od - want).max() < 1e-9
