"""Specialised ("fast") kernels: every path mode / layout variant must give the oracle's answer.
CPU tier runs the kernel bodies through the host emulator; GPU tier (-m gpu) runs the HIP kernels
through the C ABI with the same choices (fftconv_plan_options at plan creation)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import util

EMU_DIR = os.path.join(util.ROOT, "tests", "emu")

# shapes whose transform hits the specialised configurations (L = 4224 rows, M = 2112 columns)
ROW_SHAPES = [(20, 4096, 1, 5, 127, 2), (20, 4100, 2, 5, 120, 1), (12, 4000, 1, 3, 200, 1), (12, 3700, 3, 3, 500, 1)]
COL_SHAPES = [(4200, 10, 2, 25, 7, 1), (4096, 24, 1, 127, 9, 2), (3700, 12, 1, 520, 5, 1)]   # last: kernel too tall for the pruned forward pass
BOTH_SHAPE = (4096, 4096, 1, 127, 127, 1)
# the other specialised configurations: cfg1 (288), cfg2 (1088 window on a 1152 transform),
# cfg5 (2112), cfg4's 4160 window cropped from the 4224 transform, plus multi-feature / ragged
OTHER_SHAPES = [(256, 256, 1, 31, 31, 1), (1024, 1024, 1, 63, 63, 2), (2048, 2048, 1, 63, 63, 1),
                (4096, 300, 1, 63, 20, 1), (300, 4096, 2, 20, 63, 1), (1000, 1000, 3, 40, 50, 2),
                (2000, 260, 1, 100, 29, 1), (250, 280, 2, 9, 9, 2),
                (512, 512, 1, 31, 31, 2), (540, 500, 2, 37, 40, 1),    # 576 x 576 transforms
                (720, 640, 1, 21, 31, 2), (1280, 720, 1, 63, 47, 1), (1400, 1500, 2, 9, 11, 1),   # 768 / 1536
                (3000, 200, 1, 65, 31, 1), (150, 2900, 1, 7, 150, 1),   # 3072 in one dimension each
                (20, 8192, 1, 5, 127, 1), (8192, 40, 1, 127, 9, 1),     # 8448 in one dimension each (two workgroups per CU / 4-column tiles)
                (24, 6000, 1, 3, 100, 1), (6000, 40, 2, 90, 9, 1),      # 6144 in one dimension each
                # both kernels specialised with the 4-column output tiles (M = 3072 / 4224) and a short
                # row transform: the pair-adjacent intermediate + merge-while-landing path of those tiles
                (6000, 250, 1, 60, 31, 2), (8192, 260, 1, 127, 20, 1),
                # round 4: the lengths that close the gaps of the ladder, one dimension each (rows / columns)
                (24, 1300, 1, 5, 40, 1), (1300, 40, 2, 30, 9, 1),       # 1344
                (20, 1700, 2, 5, 50, 1), (1700, 40, 1, 50, 9, 2),       # 1760
                (16, 2500, 1, 3, 60, 1), (2500, 28, 1, 50, 5, 1),       # 2560
                (12, 3400, 1, 3, 100, 1), (3400, 28, 1, 100, 5, 1),     # 3520
                (10, 5000, 1, 3, 110, 1), (5000, 30, 1, 110, 3, 1),     # 5120
                (14, 2200, 1, 3, 70, 1), (2200, 28, 1, 70, 5, 1),       # 2304
                (10, 4400, 1, 3, 127, 1), (4400, 30, 1, 127, 3, 1),     # 4608
                (20, 1850, 1, 3, 63, 1), (1850, 28, 1, 63, 5, 1),       # 1920
                (14, 2700, 1, 3, 90, 1), (2700, 28, 1, 90, 5, 1),       # 2816
                (12, 3750, 1, 3, 60, 1), (3750, 30, 1, 60, 3, 1),       # 3840
                (10, 5500, 1, 3, 100, 1), (5500, 30, 1, 100, 3, 1),     # 5632
                (10, 6900, 1, 3, 127, 1), (6900, 30, 1, 127, 3, 1),     # 7040
                (10, 7500, 1, 3, 127, 1), (7500, 30, 1, 127, 3, 1),     # 7680
                # round 4, late: the small lengths between 288 and 1088 (images of 300 ... 900 pixels), rows / columns / both
                (20, 350, 1, 5, 30, 2), (350, 28, 2, 30, 5, 1), (340, 350, 1, 31, 31, 2),      # 384
                (20, 440, 1, 5, 36, 2), (440, 28, 1, 36, 5, 1), (400, 420, 2, 63, 50, 1),      # 480
                (20, 600, 1, 5, 63, 2), (600, 28, 1, 63, 5, 1), (480, 640, 1, 31, 31, 2),      # 672 (and 576 x 672 for VGA)
                (20, 800, 1, 5, 63, 2), (800, 28, 3, 63, 5, 1), (800, 800, 1, 63, 63, 1),      # 864
                (20, 830, 1, 5, 127, 2), (830, 28, 1, 127, 5, 1), (900, 880, 1, 31, 72, 2),    # 960
                # ... and two lengths into the widest gaps above 1000 (1152 -> 1344, 3072 -> 3520)
                (20, 1200, 1, 5, 63, 2), (1200, 28, 2, 63, 5, 1),       # 1280
                (12, 3300, 1, 3, 31, 2), (3300, 28, 1, 31, 5, 1), (12, 3200, 1, 3, 127, 1)]     # 3360
# (path_mode, rows_group): path mode 0 generic kernels, 1 specialised kernels + row-major
# intermediate, 2 (default) specialised kernels + tiled pair-adjacent intermediate; rows_group -1 auto
VARIANTS = [(0, -1), (1, -1), (2, -1), (2, 0), (2, 3)]


_REF_CACHE = {}


def reference(oracle, shape, seed):
    """oracle maps for (shape, seed), computed once per test session (the 4224 x 4224 case takes
    ~10 s of CPU and is shared by every variant)"""
    key = (shape, seed)
    if key not in _REF_CACHE:
        data, ks = make_inputs(shape, seed)
        H, W, F, kh, kw, n = shape
        _REF_CACHE[key] = oracle.conv_fft(data, kh, kw, ks)
    return _REF_CACHE[key]


def make_inputs(shape, seed):
    H, W, F, kh, kw, n = shape
    rng = np.random.default_rng(seed)
    data = rng.random((H, W, F), dtype=np.float32)
    ks = [rng.random((kh, kw, F), dtype=np.float32) for _ in range(n)]
    if n > 1:
        ks[1] = rng.random((max(1, kh - 1), max(1, kw // 4), F), dtype=np.float32)   # ragged cell
    return data, ks


@pytest.fixture(scope="module")
def emu():
    return ctypes.CDLL(util.build_emu())


def emu_conv(emu, data, mkh, mkw, kernels):
    d, ks, n, kp, kh, kw = util.Oracle._prep(data, kernels)
    H, W, F = d.shape
    outs = [np.full((util.ceil16(H + mkh - 1), util.ceil16(W + mkw - 1)), 7e7, dtype=np.float32, order="F") for _ in range(n)]
    op = (ctypes.c_void_p * n)(*[o.ctypes.data for o in outs])
    rc = emu.emu_conv_fft(ctypes.c_void_p(d.ctypes.data), H, W, F, mkh, mkw, n, kp, kh, kw, op, None, None)
    return rc, outs


def set_variant(emu, v):
    """emulator: path mode / rows group of the plans emulated from here on"""
    emu.emu_set_tuning(int(v[0]), int(v[1]) if len(v) > 1 else -1)


def plan_options(v):
    """the same choice as fftconv_plan_options fields for the C ABI"""
    mode, group = v[0], (v[1] if len(v) > 1 else -1)
    return {"kernel_path": {2: 0, 0: 1, 1: 2}[mode], "rows_group": 0 if group < 0 else max(1, group)}


@pytest.fixture
def tuned(emu):
    """sets the emulator's path mode / rows group for one test and restores the defaults after it"""
    yield lambda v: set_variant(emu, v)
    emu.emu_set_tuning(2, -1)


@pytest.mark.parametrize("shape", ROW_SHAPES + COL_SHAPES)
@pytest.mark.parametrize("mode", [1, 2])
def test_emulated_fast_kernels_one_dimension(emu, oracle, tuned, shape, mode):
    tuned((mode, -1))
    H, W, F, kh, kw, n = shape
    assert emu.emu_uses_fast_rows(H, W, F, kh, kw) in (1, 2)      # exactly one of the two fast kernels applies
    data, ks = make_inputs(shape, 17)
    rc, got = emu_conv(emu, data, kh, kw, ks)
    assert rc == 0
    for g, r in zip(got, oracle.conv_fft(data, kh, kw, ks)):
        assert util.rel_err(g, r) < 1e-5


@pytest.mark.parametrize("variant", VARIANTS)
def test_emulated_fast_kernels_all_variants(emu, oracle, tuned, variant):
    """both hot kernels specialised (4224 x 4224 window): every intermediate layout"""
    tuned(variant)
    H, W, F, kh, kw, n = BOTH_SHAPE
    assert emu.emu_uses_fast_rows(H, W, F, kh, kw) == (3 if variant[0] > 0 else 0)
    data, ks = make_inputs(BOTH_SHAPE, 3)
    rc, got = emu_conv(emu, data, kh, kw, ks)
    assert rc == 0
    assert util.rel_err(got[0], reference(oracle, BOTH_SHAPE, 3)[0]) < 1e-5


@pytest.mark.parametrize("shape", OTHER_SHAPES)
@pytest.mark.parametrize("mode", [1, 2])
def test_emulated_fast_kernels_other_configs(emu, oracle, tuned, shape, mode):
    tuned((mode, -1))
    H, W, F, kh, kw, n = shape
    assert emu.emu_uses_fast_rows(H, W, F, kh, kw) > 0
    data, ks = make_inputs(shape, 29)
    rc, got = emu_conv(emu, data, kh, kw, ks)
    assert rc == 0
    for g, r in zip(got, oracle.conv_fft(data, kh, kw, ks)):
        assert util.rel_err(g, r) < 1e-5


@pytest.mark.parametrize("shape", [ROW_SHAPES[0], (256, 256, 1, 31, 31, 1), (1024, 1024, 1, 63, 63, 1), (2048, 300, 1, 63, 20, 1),
                                   # F > 1: the walk over (map, feature) pairs (feature sum in registers)
                                   (300, 4096, 2, 20, 63, 1), (250, 280, 3, 9, 9, 2), (540, 500, 5, 37, 40, 1), (200, 2048, 2, 9, 63, 1)])
def test_emulated_multi_map_row_kernel(emu, oracle, tuned, shape):
    """fast_rows_multi.hpp (several maps per workgroup): the walk, its prefetch slot and the LDS reuse
    through the emulator (which repeats one kernel; distinct kernels per walk are a GPU test)"""
    tuned((2, 3))
    H, W, F, kh, kw, n = shape
    data, ks = make_inputs(shape, 43)
    rc, got = emu_conv(emu, data, kh, kw, ks)
    assert rc == 0
    for g, r in zip(got, oracle.conv_fft(data, kh, kw, ks)):
        assert util.rel_err(g, r) < 1e-5


# Dynamic tile queue of the persistent column kernels (fast_cols.hpp: TileQueue; plan option "dynamic_tiles"): the same bodies
# taking their tiles from the queue's counters instead of the static deal.  Emulated workgroups run one after the other, so
# the first drains its home counter and then the other seven (the stealing pass): the ticket -> tile map, the hand-over of
# the tile after next through the LDS slot and the termination are what is checked here; contention is a GPU matter.
DYN_SHAPES = [(256, 256, 1, 31, 31, 2), (1024, 1024, 1, 63, 63, 1), (2048, 300, 2, 63, 20, 1), (4096, 24, 1, 127, 9, 2),
              (4200, 10, 2, 25, 7, 1), (6000, 250, 1, 60, 31, 1), (340, 350, 1, 31, 31, 2), (20, 4096, 1, 5, 127, 2)]


@pytest.mark.parametrize("shape", DYN_SHAPES)
def test_emulated_dynamic_tile_queue(emu, oracle, shape):
    H, W, F, kh, kw, n = shape
    data, ks = make_inputs(shape, 71)
    emu.emu_set_dynamic_tiles(1)
    try:
        rc, got = emu_conv(emu, data, kh, kw, ks)
    finally:
        emu.emu_set_dynamic_tiles(0)
    assert rc == 0
    for g, r in zip(got, oracle.conv_fft(data, kh, kw, ks)):
        assert util.rel_err(g, r) < 1e-5


def test_dynamic_tile_queue_ticket_map_is_a_bijection():
    """ticket k of XCD v -> tile ((k >> s) * 8 + v) << s | (k & mask): every tile below n exactly once over the eight counters,
    ascending in k per counter (so a counter that has run out stays run out), chunks of 2^s consecutive tiles per XCD in turn"""
    def tile(v, k, s):
        return ((((k >> s) << 3) + v) << s) + (k & ((1 << s) - 1))
    for s in (0, 1, 3, 5):
        for n in (1, 7, 8, 33, 528, 1000):
            seen = []
            for v in range(8):
                prev = -1
                for k in range(n + 64):
                    t = tile(v, k, s)
                    assert t > prev
                    prev = t
                    if t < n:
                        seen.append(t)
                        assert (t >> s) % 8 == v
            assert sorted(seen) == list(range(n))


# the BASELINE windows that are awkward to factor have kernels of their own (1088 = 2^6 x 17: cfg2; 4160 = 2^6 x 5 x 13:
# cfg4); plans that must transform the window itself (exact_window: the spectrum exchange in the reference's order,
# src/cudaFFTData.cu:90-103, src/cudaConvFFTData.cu:92-98) run on them
NATIVE_SHAPES = [(1024, 40, 1, 63, 9, 2), (40, 1024, 2, 9, 63, 1), (1024, 1024, 1, 63, 63, 1), (1030, 1025, 1, 57, 64, 2),
                 (4096, 28, 1, 63, 5, 1), (24, 4096, 1, 5, 63, 2), (20, 4040, 2, 3, 120, 1)]


@pytest.mark.parametrize("shape", NATIVE_SHAPES)
@pytest.mark.parametrize("variant", [(2, -1), (1, -1), (2, 3)])
def test_emulated_native_window_kernels(emu, oracle, tuned, shape, variant):
    tuned(variant)
    emu.emu_set_exact_window(1)
    try:
        H, W, F, kh, kw, n = shape
        lh, lw = ctypes.c_int(0), ctypes.c_int(0)
        assert emu.emu_plan_lengths(H, W, F, kh, kw, ctypes.byref(lh), ctypes.byref(lw)) == 0
        assert (lh.value, lw.value) == (util.ceil16(H + kh - 1), util.ceil16(W + kw - 1))
        assert emu.emu_uses_fast_rows(H, W, F, kh, kw) > 0         # no generic kernel in the dimension(s) at 1088 / 4160
        if min(H, W) > 1000:
            assert emu.emu_uses_fast_rows(H, W, F, kh, kw) == 3
        data, ks = make_inputs(shape, 61)
        rc, got = emu_conv(emu, data, kh, kw, ks)
        assert rc == 0
        for g, r in zip(got, oracle.conv_fft(data, kh, kw, ks)):
            assert util.rel_err(g, r) < 1e-5
    finally:
        emu.emu_set_exact_window(0)


# Overlap-save blocks (block-wise plans whose block transform has specialised kernels, fftconv_api.cpp: tiled_convolve_save): the
# block plan is CYCLIC over Lh x Lw samples (a block of the image with S history rows / columns in front), and the output kernel
# stores the part of each block's circular result that is not wrapped straight into the block's rectangle of the full map.  The
# emulator runs the kernel bodies with the same arguments; the block loop below restates the host logic of the product.
def emu_overlap_save(emu, data, mkh, mkw, kernels, Lh, Lw):
    d, ks, n, kp, kh, kw = util.Oracle._prep(data, kernels)
    H, W, F = d.shape
    FH, FW = util.ceil16(H + mkh - 1), util.ceil16(W + mkw - 1)

    def dim(window, mk, L):      # blocks, history
        if L >= window:
            return 1, 0
        S = (max(0, mk - 1) + 15) // 16 * 16
        return -(-window // (L - S)), S
    (nbh, Sh), (nbw, Sw) = dim(FH, mkh, Lh), dim(FW, mkw, Lw)
    Bh, Bw = Lh - Sh, Lw - Sw
    outs = [np.full((FH, FW), 7e7, dtype=np.float32, order="F") for _ in range(n)]
    emu.emu_set_cyclic(1)
    try:
        for bx in range(nbw):
            for by in range(nbh):
                y0, x0 = by * Bh - Sh, bx * Bw - Sw
                blk = np.zeros((Lh, Lw, F), dtype=np.float32, order="F")
                ys, ye, xs, xe = max(0, y0), min(H, y0 + Lh), max(0, x0), min(W, x0 + Lw)
                if ye > ys and xe > xs:
                    blk[ys - y0:ye - y0, xs - x0:xe - x0, :] = d[ys:ye, xs:xe, :]
                emu.emu_set_out_window(1, Sh, Sh + min(Bh, FH - by * Bh), Sw, min(Bw, FW - bx * Bw), FH)
                op = (ctypes.c_void_p * n)(*[o.ctypes.data + 4 * (x0 * FH + y0) for o in outs])
                rc = emu.emu_conv_fft(ctypes.c_void_p(blk.ctypes.data), Lh, Lw, F, mkh, mkw, n, kp, kh, kw, op, None, None)
                if rc:
                    return rc, outs, (nbh, nbw)
    finally:
        emu.emu_set_cyclic(0)
        emu.emu_set_out_window(0, 0, 0, 0, 0, 0)
    return 0, outs, (nbh, nbw)


@pytest.mark.parametrize("case", [
    ((700, 500, 2, 9, 13, 3), 288, 288, (3, 2)),      # blocks in both dimensions, ragged edge blocks, ragged kernels
    ((600, 250, 1, 31, 17, 2), 288, 288, (3, 1)),     # one block covers w (no history there, zero padding instead)
    ((250, 1200, 1, 16, 33, 2), 288, 576, (1, 3)),    # ... covers h; 576-point rows
    ((1100, 270, 3, 1, 1, 1), 576, 288, (2, 1)),      # 1 x 1 kernels: no history at all
])
def test_emulated_overlap_save_blocks(emu, oracle, case):
    shape, Lh, Lw, blocks = case
    H, W, F, kh, kw, n = shape
    data, ks = make_inputs(shape, 61)
    rc, got, nb = emu_overlap_save(emu, data, kh, kw, ks, Lh, Lw)
    assert rc == 0 and nb == blocks
    for g, r in zip(got, oracle.conv_fft(data, kh, kw, ks)):
        assert util.rel_err(g, r) < 1e-5          # every element written (the 7e7 fill would show) and equal to the one-pass result


def test_cyclic_plans_need_both_specialised_kernels(emu):
    emu.emu_set_cyclic(1)
    try:
        assert emu.emu_uses_fast_rows(288, 576, 1, 9, 9) == 3
        assert emu.emu_uses_fast_rows(320, 576, 1, 9, 9) == -1        # no output kernel for 160 points: such a block plan is refused
        assert emu.emu_uses_fast_rows(288, 600, 1, 9, 9) == -1
    finally:
        emu.emu_set_cyclic(0)


def test_fast_row_kernel_rejects_too_wide_kernels(emu):
    # the fast row kernel takes kernels up to its stage-1 sub-length (528 for L = 4224, 704 for L = 7040, the widest); plans for
    # wider MAX_KERNEL_W fall back to the generic kernel at plan time
    assert emu.emu_uses_fast_rows(12, 3600, 1, 3, 800) == 0


def test_planner_discounts_only_lengths_whose_row_kernel_takes_max_kw(emu):
    """the length preference knows MAX_KERNEL_W: 4224 is preferred for 127-wide kernels, but a plan
    for 800-wide kernels (wider than any fast kernel takes) is not steered to a length it cannot use fast"""
    lh, lw = ctypes.c_int(0), ctypes.c_int(0)
    assert emu.emu_plan_lengths(12, 4096, 1, 3, 127, ctypes.byref(lh), ctypes.byref(lw)) == 0
    assert lw.value == 4224
    assert emu.emu_plan_lengths(12, 3600, 1, 3, 800, ctypes.byref(lh), ctypes.byref(lw)) == 0
    assert lw.value >= 4399 and emu.emu_uses_fast_rows(12, 3600, 1, 3, 800) == 0
    # 600-wide kernels: wider than the 4224-point kernel takes (528), so the plan moves to a length whose kernel does (7040: 704)
    assert emu.emu_plan_lengths(12, 3600, 1, 3, 600, ctypes.byref(lh), ctypes.byref(lw)) == 0
    assert emu.emu_uses_fast_rows(12, 3600, 1, 3, 600) in (0, 1)
    assert (lw.value in (5120, 6144, 7040)) == (emu.emu_uses_fast_rows(12, 3600, 1, 3, 600) == 1)     # (m1 = 640 / 768 / 704)


# ------------------------------------------------------------------------------------ GPU tier

@pytest.mark.gpu
@pytest.mark.parametrize("variant", VARIANTS)
def test_gpu_fast_kernels_all_variants(fftconv, oracle, variant):
    data, ks = make_inputs(BOTH_SHAPE, 5)
    H, W, F, kh, kw, n = BOTH_SHAPE
    got = fftconv.cudaConvolutionFFT(data, kh, kw, ks, options=plan_options(variant))
    assert util.rel_err(got[0], reference(oracle, BOTH_SHAPE, 5)[0]) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ROW_SHAPES + COL_SHAPES)
@pytest.mark.parametrize("mode", [1, 2])
def test_gpu_fast_kernels_one_dimension(fftconv, oracle, shape, mode):
    H, W, F, kh, kw, n = shape
    data, ks = make_inputs(shape, 23)
    got = fftconv.cudaConvolutionFFT(data, kh, kw, ks, options=plan_options((mode, -1)))
    for g, r in zip(got, oracle.conv_fft(data, kh, kw, ks)):
        assert util.rel_err(g, r) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("shape", OTHER_SHAPES)
@pytest.mark.parametrize("mode", [1, 2])
def test_gpu_fast_kernels_other_configs(fftconv, oracle, shape, mode):
    H, W, F, kh, kw, n = shape
    data, ks = make_inputs(shape, 31)
    got = fftconv.cudaConvolutionFFT(data, kh, kw, ks, options=plan_options((mode, -1)))
    for g, r in zip(got, oracle.conv_fft(data, kh, kw, ks)):
        assert util.rel_err(g, r) < 1e-5


@pytest.mark.gpu
def test_gpu_fast_kernels_many_maps_multi_feature(fftconv, oracle):
    """batched launches (several maps per launch, F = 2) through the plan API on the fast path"""
    H, W, F, kh, kw, n = 4100, 4090, 2, 120, 130, 5
    data, ks = util.synth(41, H, W, F, kh, kw, n)
    with fftconv.Plan(H, W, F, kh, kw) as p:
        assert (p.info.transform_h, p.info.transform_w) == (4224, 4224)
        p.set_image(data)
        for batch in (0, 2):
            p.set_option("batch_maps", batch)
            got = p.convolve(ks)
            ref = oracle.conv_fft(data, kh, kw, ks)
            for g, r in zip(got, ref):
                assert util.rel_err(g, r) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("group", [2, 5, -1])
@pytest.mark.parametrize("shape", [(40, 4096, 1, 7, 127, 7), (2048, 300, 1, 63, 20, 7), (1024, 1024, 1, 63, 63, 7),
                                   (256, 256, 1, 31, 31, 7), (300, 4096, 1, 20, 63, 11),
                                   (512, 512, 1, 31, 31, 7), (720, 640, 1, 21, 31, 7), (1280, 720, 1, 63, 47, 7),
                                   (200, 3000, 1, 31, 65, 7), (150, 6000, 1, 9, 70, 7), (120, 8192, 1, 7, 127, 7)])
def test_gpu_multi_map_row_kernel(fftconv, oracle, shape, group):
    """several maps per workgroup with DISTINCT kernels (walk indexing, prefetch of the next kernel
    row, partial last walk: 7 = 2+2+2+1 = 5+2) on every fast row configuration"""
    H, W, F, kh, kw, n = shape
    rng = np.random.default_rng(1000 + group)
    data = rng.random((H, W, F), dtype=np.float32)
    ks = [rng.random((kh, kw, F), dtype=np.float32) for _ in range(n)]
    with fftconv.Plan(H, W, F, kh, kw, options=plan_options((2, group))) as p:
        p.set_image(data)
        got = p.convolve(ks)
    for g, r in zip(got, oracle.conv_fft(data, kh, kw, ks)):
        assert util.rel_err(g, r) < 1e-5


@pytest.mark.gpu
def test_gpu_multi_map_row_kernel_auto_group_many_maps(fftconv, oracle):
    """enough maps that the per-launch choice walks several maps per workgroup (cfg1-sized maps)"""
    H, W, kh, kw, n = 256, 256, 31, 31, 1500
    rng = np.random.default_rng(99)
    data = rng.random((H, W, 1), dtype=np.float32)
    ks = [rng.random((kh, kw, 1), dtype=np.float32) for _ in range(n)]
    with fftconv.Plan(H, W, 1, kh, kw) as p:
        p.set_option("batch_maps", 1500)
        p.set_image(data)
        got = p.convolve(ks)
    idx = [0, 1, 15, 16, 17, 700, 1498, 1499]
    ref = oracle.conv_fft(data, kh, kw, [ks[i] for i in idx])
    for i, r in zip(idx, ref):
        assert util.rel_err(got[i], r) < 1e-5


def _specialised_vs_generic(fftconv, H, W, kh, kw, n, data, ks, expect_transform, sample, exact_window=0, variants=None):
    """device-resident maps of the specialised path (in one pass at the expected transform: option blockwise = 1 -- or each
    option set of `variants` with its own expectation about the plan) against the generic kernels (kernel_path 1) on the
    same inputs"""
    torch = pytest.importorskip("torch")
    kd = torch.from_numpy(np.ascontiguousarray(np.stack([np.transpose(x, (2, 1, 0)) for x in ks]))).cuda()
    runs = [(0, {}, None)] + [(2, o, e) for o, e in (variants or [({"blockwise": 1}, lambda p: expect_transform(p.info.transform_h, p.info.transform_w))])]
    generic = None
    for mode, extra, expect in runs:
        with fftconv.Plan(H, W, 1, kh, kw, options=dict(plan_options((mode, -1)), exact_window=exact_window, **extra)) as p:
            if expect is not None:
                assert expect(p), (extra, p.info.transform_h, p.info.transform_w, p.get_option("blockwise"))
            p.set_image(data)
            od = torch.empty((n, p.info.fft_w, p.info.fft_h), dtype=torch.float32, device="cuda")
            p.convolve_packed_device(n, kd.data_ptr(), kh, kw, od.data_ptr())
            p.synchronize()
            maps = [od[j].cpu().numpy() for j in sample]
            del od
        if mode == 0:
            generic = maps
        else:
            for a, b in zip(maps, generic):
                assert util.rel_err(a, b) < 1e-5, extra


@pytest.mark.gpu
@pytest.mark.parametrize("size,k", [(6000, 63), (8192, 127)])
def test_gpu_big_square_walk_with_remainder(fftconv, size, k):
    """both dimensions on the 6144 / 8448 configurations (two row workgroups per CU, 4-column output
    tiles, cropped window at 6000), 17 kernels = one full walk of 16 + a remainder of 1.  The float64
    oracle needs minutes at this size, so the specialised path is compared with the generic kernels
    (kernel_path 1, themselves pinned to the oracle at every smaller size) on the same inputs.
    Twice: in one pass (option blockwise = 1: the long-transform kernels) and as the default plan, which from about
    4900 x 4900 runs overlap-save blocks of a shorter transform where the planner's measured costs say so."""
    H = W = size
    n = 17
    rng = np.random.default_rng(size)
    data = rng.random((H, W, 1), dtype=np.float32)
    ks = [rng.random((k, k, 1), dtype=np.float32) for _ in range(n)]
    one_pass = lambda p: p.get_option("blockwise") == 0 and p.info.transform_h == p.info.transform_w and p.info.transform_h in (6144, 8448)
    default = lambda p: (p.get_option("blockwise") == 0 and p.info.transform_h in (6144, 8448)) or \
                        (p.get_option("blockwise") > 1 and p.get_option("overlap_save") == 1 and p.get_option("specialised_kernels") == 3)
    _specialised_vs_generic(fftconv, H, W, k, k, n, data, ks, None, (0, 7, n - 1), variants=[({"blockwise": 1}, one_pass), ({}, default)])


@pytest.mark.gpu
@pytest.mark.parametrize("shape", NATIVE_SHAPES)
@pytest.mark.parametrize("group", [-1, 3])
def test_gpu_native_window_kernels(fftconv, oracle, shape, group):
    """exact_window plans at the cfg2 / cfg4 windows (1088, 4160) run on those lengths' own kernels"""
    H, W, F, kh, kw, n = shape
    data, ks = make_inputs(shape, 67)
    with fftconv.Plan(H, W, F, kh, kw, options=dict(plan_options((2, group)), exact_window=1)) as p:
        assert p.info.exact_window == 1 and p.get_option("specialised_kernels") > 0
        if min(H, W) > 1000:
            assert p.get_option("specialised_kernels") == 3
        p.set_image(data)
        got = p.convolve(ks)
    for g, r in zip(got, oracle.conv_fft(data, kh, kw, ks)):
        assert util.rel_err(g, r) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("lh,lw", [(1088, 1088), (1088, 4160), (4160, 1088)])
def test_gpu_native_window_pairs_vs_generic(fftconv, lh, lw):
    kh, kw = 33, 47
    H, W = lh - kh + 1 - 3, lw - kw + 1 - 5
    n = 17
    rng = np.random.default_rng(lh * 10007 + lw)
    data = rng.standard_normal((H, W, 1)).astype(np.float32)
    ks = [rng.standard_normal((kh, kw, 1)).astype(np.float32) for _ in range(n)]
    _specialised_vs_generic(fftconv, H, W, kh, kw, n, data, ks, lambda a, b: (a, b) == (lh, lw), range(n), exact_window=1)


@pytest.mark.gpu
@pytest.mark.parametrize("lh,lw", [(288, 288), (576, 768), (768, 576), (1152, 1536), (1536, 1152), (2112, 3072), (3072, 2112),
                                   (4224, 576), (768, 4224), (6144, 288), (288, 6144), (8448, 768), (1536, 8448), (2112, 2112),
                                   # round 4's lengths
                                   (1344, 1760), (1760, 1344), (2560, 3520), (3520, 2560), (5120, 1344), (1344, 5120), (5120, 5120),
                                   (2304, 4608), (4608, 2304), (1920, 2816), (2816, 1920), (3840, 5632), (5632, 3840), (7040, 1920), (1920, 7680),
                                   (7680, 7040),
                                   # the small lengths of the late round-4 batch
                                   (384, 480), (480, 384), (672, 864), (864, 672), (960, 384), (384, 960), (480, 960), (864, 864), (672, 288),
                                   (288, 672), (960, 1152), (1152, 480),
                                   (1280, 3360), (3360, 1280), (1280, 1280), (3360, 3360), (1280, 576)])
def test_gpu_every_fast_length_pair_vs_generic(fftconv, lh, lw):
    """both kernels specialised, every transform length at least once along h and along w, 17
    kernels (multi-map walk + remainder), odd data sizes: specialised path against the generic
    kernels on the same inputs"""
    kh, kw = 9 + lh // 64, 7 + lw // 96
    H, W = lh - kh + 1 - 3, lw - kw + 1 - 5
    n = 17
    rng = np.random.default_rng(lh * 10007 + lw)
    data = rng.standard_normal((H, W, 1)).astype(np.float32)
    ks = [rng.standard_normal((kh, kw, 1)).astype(np.float32) for _ in range(n)]
    _specialised_vs_generic(fftconv, H, W, kh, kw, n, data, ks, lambda a, b: (a, b) == (lh, lw), range(n))
