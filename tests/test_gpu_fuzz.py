"""GPU tier: seeded random shapes against the oracle.  One dimension is drawn near a specialised
transform length (so every fast row / output configuration, every pruning variant NZ2, the cropped
and the tile-aligned store paths and the multi-map walk are hit with odd sizes), the other stays
small so that the float64 oracle finishes quickly.  Several kernels per case (distinct, ragged)."""
import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu

FAST_LENGTHS = [288, 384, 480, 576, 672, 768, 864, 960, 1152, 1280, 1344, 1536, 1760, 1920, 2112, 2304, 2560, 2816, 3072, 3360, 3520, 3840, 4224, 4608, 5120, 5632, 6144,
                7040, 7680, 8448]


def _cases():
    rng = np.random.default_rng(20261003)
    cases = []
    for L in FAST_LENGTHS:
        for rep in range(4):
            k_long = int(rng.integers(1, min(L // 8, 200) + 1))          # kernel extent along the long dimension
            slack = int(rng.integers(0, max(1, L // 12)))                  # how far below the transform length
            long_dim = max(1, L - k_long + 1 - slack)
            short_dim = int(rng.integers(3, 70))
            k_short = int(rng.integers(1, min(short_dim, 12) + 1))
            F = int(rng.choice([1, 1, 1, 2, 3]))
            n = int(rng.choice([1, 3, 6]))
            if rep % 2 == 0:
                cases.append((short_dim, long_dim, F, k_short, k_long, n))   # long along w: row kernels
            else:
                cases.append((long_dim, short_dim, F, k_long, k_short, n))   # long along h: column kernels
    return cases


@pytest.mark.parametrize("case", _cases())
def test_random_shapes_match_oracle(fftconv, oracle, case):
    H, W, F, kh, kw, n = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 32))
    data = rng.standard_normal((H, W, F)).astype(np.float32)
    ks = []
    for i in range(n):
        h = kh if i % 3 != 1 else max(1, kh - int(rng.integers(0, 3)))
        w = kw if i % 3 != 2 else max(1, kw - int(rng.integers(0, 3)))
        ks.append(rng.standard_normal((h, w, F)).astype(np.float32))
    got = fftconv.cudaConvolutionFFT(data, kh, kw, ks)
    ref = oracle.conv_fft(data, kh, kw, ks)
    for g, r in zip(got, ref):
        assert g.shape == r.shape
        assert util.rel_err(g, r) < 1e-5


def test_random_square_mid_sizes_many_maps(fftconv, oracle):
    """both dimensions specialised at once, enough maps for the multi-map walk (checked on a sample)"""
    rng = np.random.default_rng(7)
    for (H, W, kh, kw, n) in [(500, 530, 31, 17, 70), (700, 1400, 9, 40, 40), (250, 1000, 30, 60, 90)]:
        data = rng.random((H, W, 1), dtype=np.float32)
        ks = [rng.random((kh, kw, 1), dtype=np.float32) for _ in range(n)]
        with fftconv.Plan(H, W, 1, kh, kw) as plan:
            plan.set_image(data)
            got = plan.convolve(ks)
        idx = [0, 1, n // 2, n - 2, n - 1]
        ref = oracle.conv_fft(data, kh, kw, [ks[i] for i in idx])
        for i, r in zip(idx, ref):
            assert util.rel_err(got[i], r) < 1e-5
