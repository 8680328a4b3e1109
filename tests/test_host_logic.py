"""CPU tier: host logic of the product (planner, tables, layouts, argument blocks) exercised
through the test-only emulator, and the C-ABI library's load/export/no-GPU behaviour."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

import golden_util
import util

EMU_DIR = os.path.join(util.ROOT, "tests", "emu")


@pytest.fixture(scope="module")
def emu():
    lib = ctypes.CDLL(util.build_emu())
    lib.emu_spectrum_elems.restype = ctypes.c_long
    return lib


def emu_conv(emu, data, mkh, mkw, kernels):
    d, ks, n, kp, kh, kw = util.Oracle._prep(data, kernels)
    H, W, F = d.shape
    fh, fw = util.ceil16(H + mkh - 1), util.ceil16(W + mkw - 1)
    outs = [np.full((fh, fw), 7e7, dtype=np.float32, order="F") for _ in range(n)]
    op = (ctypes.c_void_p * n)(*[o.ctypes.data for o in outs])
    lh, lw = ctypes.c_int(), ctypes.c_int()
    rc = emu.emu_conv_fft(ctypes.c_void_p(d.ctypes.data), H, W, F, mkh, mkw, n, kp, kh, kw, op,
                          ctypes.byref(lh), ctypes.byref(lw))
    return rc, outs, (lh.value, lw.value)


LENGTHS = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 20, 22, 26, 32, 33, 34, 40, 64, 66, 80, 88,
           128, 136, 144, 256, 272, 528, 544, 1040, 1056, 1088, 2080, 2112, 4096, 4160, 4224, 2431, 245, 1001]


@pytest.mark.parametrize("L", LENGTHS)
def test_transform_1d(emu, L):
    """in-place DIF forward / DIT inverse of fft_lds.hpp against numpy, through the plan's
    digit-reversal table"""
    assert emu.emu_length_supported(L)
    rng = np.random.default_rng(L)
    x = (rng.standard_normal(L) + 1j * rng.standard_normal(L)).astype(np.complex64)
    xi = np.ascontiguousarray(x).view(np.float32).copy()
    xo = np.zeros(2 * L, np.float32)
    assert emu.emu_fft1d(L, xi.ctypes.data_as(ctypes.c_void_p), xo.ctypes.data_as(ctypes.c_void_p), 0) == 0
    ref = np.fft.fft(x.astype(np.complex128))
    assert np.abs(xo.view(np.complex64) - ref).max() / np.abs(ref).max() < 2e-6
    yo = np.zeros(2 * L, np.float32)
    assert emu.emu_fft1d(L, xo.ctypes.data_as(ctypes.c_void_p), yo.ctypes.data_as(ctypes.c_void_p), 1) == 0
    assert np.abs(yo.view(np.complex64) / L - x).max() / np.abs(x).max() < 2e-6


def test_unsupported_prime_lengths(emu):
    for L in (19, 23, 38, 4222, 1087):
        assert not emu.emu_length_supported(L)


@pytest.mark.parametrize("need,exact", [(286, 288), (1086, 1088), (4222, 4224), (4158, 4160), (2110, 2112), (73, 80),
                                        (11, 16), (1, 16), (500, 512), (1000, 1008), (8191, 8192)])
def test_choose_length(emu, need, exact):
    for real_half in (0, 1):
        L = emu.emu_choose_length(need, real_half, exact)
        assert L >= need
        assert emu.emu_length_supported(L // 2 if real_half else L)
        if real_half:
            assert L % 2 == 0
        assert L <= 2 * need + 32


def test_baseline_configs_use_the_reference_window(emu):
    # every BASELINE config's ceil16 window factors into the engine's radices, so with the generic
    # kernels the internal transform equals the reference's circular modulus
    for need, win in [(286, 288), (1086, 1088), (4222, 4224), (4158, 4160), (2110, 2112)]:
        assert emu.emu_choose_length(need, 1, win) == win
        assert emu.emu_choose_length(need, 0, win) == win


def test_plans_prefer_lengths_with_specialised_kernels(emu):
    """cfg1/3/5 run at their window; cfg2 (1088) and cfg4 (4160) move to 1152 / 4224 where
    specialised kernels exist; path mode 0 (generic kernels only) stays at the window"""
    lh, lw = ctypes.c_int(), ctypes.c_int()
    want = {(256, 31): 288, (1024, 63): 1152, (4096, 127): 4224, (4096, 63): 4224, (2048, 63): 2112}
    for (n, k), L in want.items():
        emu.emu_allow_fast(2)
        assert emu.emu_plan_lengths(n, n, 1, k, k, ctypes.byref(lh), ctypes.byref(lw)) == 0
        assert (lh.value, lw.value) == (L, L)
        emu.emu_allow_fast(0)
        assert emu.emu_plan_lengths(n, n, 1, k, k, ctypes.byref(lh), ctypes.byref(lw)) == 0
        assert (lh.value, lw.value) == (util.ceil16(n + k - 1),) * 2
    emu.emu_allow_fast(2)


@pytest.mark.parametrize("case", golden_util.golden_cases())
def test_emulated_pipeline_matches_golden(emu, case):
    data, mkh, mkw, kernels, expect = golden_util.load_case(case)
    rc, got, _ = emu_conv(emu, data, mkh, mkw, kernels)
    assert rc == 0
    for g, e in zip(got, expect):
        assert util.rel_err(g, e) < 1e-5


@pytest.mark.parametrize("shape", [(64, 8, 5, 10, 4, 3), (33, 47, 3, 7, 5, 2), (1, 1, 1, 1, 1, 1), (2, 3, 1, 1, 2, 1),
                                   (5, 5, 2, 5, 5, 1), (17, 1, 1, 3, 1, 1), (1, 40, 2, 1, 9, 2), (300, 20, 1, 21, 3, 1),
                                   (130, 260, 3, 12, 8, 2), (50, 60, 1, 19, 23, 1)])
def test_emulated_pipeline_matches_oracle(emu, oracle, shape):
    H, W, F, kh, kw, n = shape
    rng = np.random.default_rng(sum(shape))
    data = rng.random((H, W, F), dtype=np.float32)
    ks = [rng.random((kh, kw, F), dtype=np.float32) for _ in range(n)]
    if n > 1:
        ks[1] = rng.random((max(1, kh - 2), max(1, kw - 1), F), dtype=np.float32)   # ragged cell
    rc, got, L = emu_conv(emu, data, kh, kw, ks)
    assert rc == 0
    assert L[0] >= H + kh - 1 and L[1] >= W + kw - 1 and L[0] % 2 == 0
    ref = oracle.conv_fft(data, kh, kw, ks)
    for g, r in zip(got, ref):
        assert util.rel_err(g, r) < 1e-5


def _random_small_shapes(count, seed):
    rng = np.random.default_rng(seed)
    shapes = []
    for i in range(count):
        if i % 4 == 0:      # data + kernel - 1 on or just under the smallest specialised lengths (288 along either axis)
            kh, kw = int(rng.integers(1, 40)), int(rng.integers(1, 40))
            H = 288 - kh + 1 - int(rng.integers(0, 12)) if rng.random() < 0.7 else int(rng.integers(kh, 120))
            W = 288 - kw + 1 - int(rng.integers(0, 12)) if rng.random() < 0.7 else int(rng.integers(kw, 120))
        else:
            H, W = int(rng.integers(1, 150)), int(rng.integers(1, 150))
            kh, kw = int(rng.integers(1, min(H, 40) + 1)), int(rng.integers(1, min(W, 40) + 1))
        shapes.append((H, W, int(rng.choice([1, 1, 2, 3, 5])), kh, kw, int(rng.integers(1, 6))))
    return shapes


@pytest.mark.parametrize("shape", _random_small_shapes(24, 2024))
def test_emulated_pipeline_random_shapes(emu, oracle, shape):
    """the CPU-tier slice of the randomised parity run (tools/fuzz_gpu.py is its GPU form): seeded random shapes, ragged cells,
    the kernel bodies through the host emulator against the oracle"""
    H, W, F, kh, kw, n = shape
    rng = np.random.default_rng(sum(shape) * 31 + H)
    data = rng.standard_normal((H, W, F)).astype(np.float32)
    ks = [rng.standard_normal((int(rng.integers(1, kh + 1)) if rng.random() < 0.4 else kh,
                               int(rng.integers(1, kw + 1)) if rng.random() < 0.4 else kw, F)).astype(np.float32) for _ in range(n)]
    rc, got, L = emu_conv(emu, data, kh, kw, ks)
    assert rc == 0 and L[0] >= H + kh - 1 and L[1] >= W + kw - 1
    for g, r in zip(got, oracle.conv_fft(data, kh, kw, ks)):
        assert util.rel_err(g, r) < 1e-5


def test_emulated_oversize_kernel_policy(emu):
    """kernel > MAXK: reproduced (circular, like the reference) when the transform is the ceil16
    window, rejected otherwise"""
    rng = np.random.default_rng(3)
    data = rng.random((40, 20, 1), dtype=np.float32)          # window 48x32 == transform
    rc, got, L = emu_conv(emu, data, 9, 13, [rng.random((12, 16, 1), dtype=np.float32)])
    assert L == (48, 32) and rc == 0
    data = rng.random((64, 8, 1), dtype=np.float32)           # window 80x16, transform 80x11
    rc, _, L = emu_conv(emu, data, 10, 4, [rng.random((10, 6, 1), dtype=np.float32)])
    assert L[1] != 16 and rc == -3


# ---------------------------------------------------------------- the C-ABI library itself

def declared_symbols():
    hdr = open(os.path.join(util.ROOT, "include", "fftconv.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(fftconv_[a-z0-9_]+)\s*\(", hdr)))


def test_library_loads_and_exports_every_declared_symbol(fftconv):
    lib = fftconv.load_library()
    syms = declared_symbols()
    assert len(syms) >= 19
    for s in syms:
        assert hasattr(lib, s), s
    assert set(syms) == set(fftconv.EXPORTED_SYMBOLS)
    assert lib.fftconv_version().startswith(b"fftconv-mi355x")


def test_fft_size16_abi(fftconv):
    for n, want in [(1, 16), (16, 16), (17, 32), (4222, 4224), (73, 80)]:
        assert fftconv.fft_size16(n) == want


def test_argument_errors_need_no_gpu(fftconv):
    data = np.zeros((8, 8, 2), np.float32)
    k = np.zeros((3, 3, 2), np.float32)
    with pytest.raises(fftconv.FFTConvError) as e:      # src/cudaConvolutionFFT.cu:72-73
        fftconv.cudaConvolutionFFT(data, 3, 3, [k], [8, 8, 8])
    assert e.value.status == -2 and e.value.identifier == "cudaConvFFTData:InvalidInput"
    with pytest.raises(fftconv.FFTConvError) as e:      # :242 feature mismatch
        fftconv.cudaConvolutionFFT(data, 3, 3, [np.zeros((3, 3, 1), np.float32)])
    assert e.value.status == -3
    with pytest.raises(fftconv.FFTConvError):           # :64-65 kernel must be a cell
        fftconv.cudaConvolutionFFT(data, 3, 3, k)
    with pytest.raises(fftconv.FFTConvError):           # :51-54 single only
        fftconv.cudaConvolutionFFT(data.astype(np.float64), 3, 3, [k])


def test_compute_fails_loudly_without_gpu(fftconv):
    """the product has no CPU fallback: on a box without a GPU every compute entry raises"""
    if fftconv.device_count() > 0:
        pytest.skip("a GPU is present")
    data = np.zeros((8, 8, 1), np.float32)
    with pytest.raises(fftconv.FFTConvError) as e:
        fftconv.cudaConvolutionFFT(data, 3, 3, [np.zeros((3, 3, 1), np.float32)])
    assert e.value.status in (-6, -7)
    with pytest.raises(fftconv.FFTConvError):
        fftconv.Plan(8, 8, 1, 3, 3)


def test_round2_entry_points_argument_errors_need_no_gpu(fftconv):
    """pow2 sizing, the live-plan registry, plan options and the multi-device entry validate their
    arguments before anything touches a device"""
    import ctypes
    lib = fftconv.load_library()
    # computeFFTsize (src/cudaConvFFTData.h:67-94): align to 16, then to a power of two
    assert [fftconv.fft_size_pow2(n) for n in (1, 16, 17, 64, 65, 286, 4222)] == [16, 16, 32, 64, 128, 512, 8192]
    # no plan is live in a process that never created one: stray handles are not dereferenced
    for stray in (0, 8, 0xDEADBEEF0):
        assert lib.fftconv_plan_is_live(ctypes.c_void_p(stray)) == 0
    assert lib.fftconv_plan_destroy(ctypes.c_void_p(0xDEADBEEF0)) != 0          # refused, not freed
    # two-step convolve with a handle that is not a plan (src/cudaConvFFTData.cu:68)
    rc = lib.fftconv_conv_fft_data(ctypes.c_void_p(0xDEADBEEF0), 0, None, None, None, None, None, 0, None)
    assert rc == -1 and b"Invalid input to MEX file" in lib.fftconv_last_error()
    # multi-device entry: empty or NULL device list
    h = ctypes.c_void_p(None)
    assert lib.fftconv_multi_create(ctypes.byref(h), 8, 8, 1, 3, 3, None, 0, None) == -1 and not h.value
    assert lib.fftconv_multi_size(None) == 0 and lib.fftconv_multi_destroy(None) == 0
    # one-shot _ex: bad kernel location
    data = np.zeros((8, 8, 1), np.float32)
    rc = lib.fftconv_convolution_fft_ex(ctypes.c_void_p(data.ctypes.data), 8, 8, 1, 3, 3, 0, None, None, None, None, 7,
                                        None, 0, 0, None, None, None, None)
    assert rc == -1 and b"kernel location" in lib.fftconv_last_error()
    # filter blocks of the in-library sharding == the Python orchestration's (cfg4: 128 per GPU of 8)
    import importlib
    mg = importlib.import_module("cuda-fft-convolution_amd.multi_gpu")
    assert [mg.filter_shard(1024, r, 8) for r in range(8)] == [(128 * r, 128) for r in range(8)]
    assert [mg.filter_shard(10, r, 4) for r in range(4)] == [(0, 3), (3, 3), (6, 2), (8, 2)]


def test_plan_cache_entry_points_need_no_gpu(fftconv):
    """the plan cache of the one-shot entry (round 4): its control entries work -- and validate -- without a device"""
    import ctypes
    lib = fftconv.load_library()
    assert lib.fftconv_cache_configure(-1, 0) == -1
    assert lib.fftconv_cache_configure(4, 0) == 0
    st = fftconv.cache_stats()
    assert st["plans"] == 0 and st["device_bytes"] == 0
    assert lib.fftconv_cache_clear() == 0
    assert lib.fftconv_last_call_timing(None) == -1
    t = fftconv.last_call_timing()
    assert set(t) == {"plan_ms", "image_ms", "convolve_ms", "release_ms", "total_ms", "cache_hit"}
    # PlanOptions carries the field appended in 0.3 and its size is what the library checks
    o = fftconv.PlanOptions(verbose=1)
    assert o.struct_size == ctypes.sizeof(fftconv.PlanOptions) and o.verbose == 1


def test_multi_gpu_prepare_is_queued_ahead_of_the_wait_for_the_spectrum():
    """The image-independent part of a step (the kernels' column transforms) must be handed to the engine BEFORE the
    stream waits for the broadcast spectrum -- on ranks that do not transform the image that is the only thing that can
    overlap the broadcast (advisor, round 3: a deferred launch behind the wait loses it silently)."""
    import importlib
    mg = importlib.import_module("cuda-fft-convolution_amd.multi_gpu")
    log = []

    class Sync(mg.NullSync):
        def wait(self, ev, side=False):
            log.append("wait-side" if side else "wait-main")

    class Eng:
        sync = Sync()

        def new_spectrum(self):
            return object()

        def compute_spectrum(self, spec, image):
            log.append("transform")

        def prepare_kernels(self, first, count):
            log.append("prepare")

        def convolve(self, spec, first, count):
            log.append("convolve")

    class Dist:
        def broadcast(self, t, src=0, async_op=False):
            log.append("broadcast")

            class W:
                def wait(self):
                    pass
            return W()

    for rank in (0, 1):
        del log[:]
        conv = mg.FilterShardedConvolver(Eng(), Dist(), rank, 2, 8, src=0, depth=2, time_broadcast="wall")
        conv.run([None, None, None])
        # every convolve is preceded by its prepare, and that prepare comes before the wait on the main stream
        idx = [i for i, e in enumerate(log) if e == "convolve"]
        assert len(idx) == 3
        for i in idx:
            w = max(j for j in range(i) if log[j] == "wait-main")
            pr = max(j for j in range(i) if log[j] == "prepare")
            assert pr < w < i, (rank, log)
        assert log.count("transform") == (3 if rank == 0 else 0)
        assert len(conv.broadcast_ms()) == 3 and conv.broadcast_ms() == []


def test_public_header_is_plain_c(tmp_path):
    """include/fftconv.h is the drop-in boundary: plain C99 (pointers and sizes, no C++ or HIP types),
    and usable from C++ as well"""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    src = tmp_path / "use_header.c"
    src.write_text('#include "fftconv.h"\n'
                   'int use(void) {\n'
                   '    fftconv_plan_options o = {0};\n'
                   '    fftconv_plan *p = 0; fftconv_multi *m = 0; fftconv_plan_info info; fftconv_profile prof;\n'
                   '    o.struct_size = sizeof o; (void)info; (void)prof; (void)p; (void)m;\n'
                   '    return fftconv_fft_size16(17) + fftconv_fft_size_pow2(17) + (int)FFTCONV_AUTO + (int)FFTCONV_ERR_NO_IMAGE;\n'
                   '}\n')
    inc = os.path.join(util.ROOT, "include")
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", inc, "-fsyntax-only", str(src)], check=True)
    if shutil.which("g++"):
        subprocess.run(["g++", "-std=c++11", "-Wall", "-Werror", "-I", inc, "-x", "c++", "-fsyntax-only", str(src)], check=True)


def _resource_reports():
    """{kernel symbol: {vgpr, spill, scratch, occ}} from the build's -Rpass-analysis=kernel-resource-usage output
    (csrc/*.rpt, written by csrc/Makefile beside every kernels_*.o)"""
    import glob
    import re
    csrc = os.path.join(util.ROOT, "cuda-fft-convolution_amd", "csrc")
    subprocess.run(["make", "-C", csrc], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    out = {}
    for path in glob.glob(os.path.join(csrc, "*.rpt")):
        cur = None
        for line in open(path):
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                cur = out.setdefault(m.group(1), {})
                continue
            for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("spill", r"VGPRs Spill: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                             ("occ", r"Occupancy \[waves/SIMD\]: (\d+)")):
                m = re.search(pat, line)
                if m and cur is not None:
                    cur[key] = int(m.group(1))
    return out


def test_lds_bank_model_reproduces_the_documented_layout_of_cfg3():
    """tools/lds_bank_model.py chose the padded LDS images of the output kernel (csrc/fast_cols.hpp: FC_COL_LAYOUTS); the
    figures the kernel's comments and DESIGN.md 4 quote for cfg3's configuration (dense: 7 896 LDS cycles per tile, 2 160 of
    them conflicts = 27.4 %, the counters say 29.2 %; padded: 6 102) must keep coming out of it, and every listed layout
    must fit the LDS and be what the header lists"""
    import importlib.util
    import re
    spec = importlib.util.spec_from_file_location("lds_bank_model", os.path.join(util.ROOT, "tools", "lds_bank_model.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    assert m.model(2112, 6, 16, 22, 8, 768, 0, verbose=False)[:2] == (7896, 2160)
    assert m.model(2112, 6, 16, 22, 8, 768, 22, verbose=False, rot=15, by_unit=True, two_level=True)[:2] == (6102, 366)
    hdr = open(os.path.join(util.ROOT, "cuda-fft-convolution_amd", "csrc", "fast_cols.hpp")).read()
    listed = re.findall(r"^    X\((\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)\)", hdr, flags=re.M)
    assert len(listed) >= 5
    configs = {tuple(int(x) for x in c.split()[:5]): int(c.split()[5]) for c in m.CONFIGS.replace("\n", " ").split(";")}
    # the tool's configuration list is the header's (fast_paths.hpp: FC_FAST_COL_CONFIGS_G0 / _G1)
    fp = open(os.path.join(util.ROOT, "cuda-fft-convolution_amd", "csrc", "fast_paths.hpp")).read()
    cols = fp[fp.index("#define FC_FAST_COL_CONFIGS_G0(X)"):fp.index("#define FC_FAST_COL_CONFIGS(X)")]
    in_header = {tuple(int(x) for x in mm) for mm in re.findall(r"X\((\d+), (\d+), (\d+), (\d+), (\d+), (\d+)\)", cols)}
    assert in_header == {k + (v,) for k, v in configs.items()}, in_header ^ {k + (v,) for k, v in configs.items()}
    for ent in listed:
        M, R1, R2, R3, T, pad, rot = (int(x) for x in ent)
        assert (M, R1, R2, R3, T) in configs, ent                 # a configuration of fast_paths.hpp
        NT = configs[(M, R1, R2, R3, T)]
        dense = m.model(M, R1, R2, R3, T, NT, 0, verbose=False)
        padded = m.model(M, R1, R2, R3, T, NT, pad, verbose=False, rot=rot, by_unit=True, two_level=True)
        assert padded[2] <= 160 * 1024 and padded[0] < dense[0], (ent, dense, padded)


def test_hot_kernels_do_not_spill():
    """Register allocation is part of the product: a spilled register in a hot loop costs the output kernel 20 %
     and scratch traffic shares the in-order memory counter with the prefetches.  Asserted on the build's
    own report: no F = 1 kernel of the hot path spills or uses scratch, every one keeps 3 waves per SIMD, and the
    F > 1 walk (k_fast_rows_multi_f: feature sum + image row + butterfly do not fit 168 registers) stays within the
    documented bound."""
    rep = _resource_reports()
    if not rep:
        pytest.skip("no csrc/*.rpt resource reports beside the objects (a library built without csrc/Makefile)")
    hot = {k: v for k, v in rep.items() if any(s in k for s in ("k_fast_rows_multiI", "k_fast_colsI", "k_fast_cols_fwdI", "k_fast_rows_fwdI"))}
    assert len(hot) > 40, len(hot)
    bad = {k: v for k, v in hot.items() if v.get("spill", 0) != 0 or v.get("scratch", 0) != 0 or v.get("occ", 0) < 3}
    # the one exception: the output kernel of cfg4's own window (M = 2080, used by exact_window plans only -- default plans run that
    # window on the 4224-point kernels).  Round 5: 8.13.20 on 832 threads (13 waves: 128 registers): 4-5 spilled registers in the
    # tiled variants (six rounds of prefetch registers for 5.005 rounds of gather units), 12 in the row-major one -- against 12-21
    # for round 4's 8.10.26, and 12 % faster (profiles/r05d_native_window_search.txt; 13.8.20: 7-14, 4.20.26: 10-19)
    known = {k: v for k, v in bad.items() if "k_fast_colsINS_6ColCfgILi2080E" in k and v.get("occ", 0) >= 3 and
             v.get("spill", 0) <= (12 if "ELb0ELb0EEEv" in k else 5)}
    # ... and the ROW-MAJOR-intermediate variant (template argument TILED = false: generic row kernel beside a specialised column
    # kernel) of M = 3072 = 8.32.12 on 1024 threads (128 registers): 2 spilled registers; the tiled variant, the default, has none
    # (round 5, padded LDS image: its landing maps a dense position to a cell, p + p / m1 * pad: 6 spilled registers, and 2 in the same
    # variant of M = 2560 = 8.32.10)
    known.update({k: v for k, v in bad.items() if ("k_fast_colsINS_6ColCfgILi3072E" in k or "k_fast_colsINS_6ColCfgILi2560E" in k) and
                  "ELb0ELb0ELb0EEEv" in k and v.get("spill", 0) <= 6 and v.get("occ", 0) >= 3})
    bad = {k: v for k, v in bad.items() if k not in known}
    assert not bad, bad
    multi_f = {k: v for k, v in rep.items() if "k_fast_rows_multi_fI" in k}
    assert len(multi_f) >= 40 and all(v["occ"] >= 3 for v in multi_f.values())
    # Round 4: EVERY configuration of the F > 1 walk is spill-free (rounds 2-3: 19-48 spilled registers in the configurations with
    # two or four rows per workgroup -- what went to scratch were per-thread index computations hoisted out of the walk; they
    # are now recomputed per step, FC_OPAQUE in fast_rows_multi.hpp), and the one-map F > 1 kernel (22-68) is not built any more
    spilled = {k: v for k, v in multi_f.items() if v.get("spill", 0) or v.get("scratch", 0)}
    assert not spilled, spilled
    assert not [k for k in rep if "k_fast_rowsI" in k and k.endswith("ELb1EEEvNS_12FastRowsArgsEiiii")]


def test_bench_counts_distinct_devices_by_pci_id_then_by_index():
    """the N > 1 bench refuses to run on fewer distinct devices than ranks (bench.py: count_distinct_devices); a host whose
    devices all report one PCI address (or none) must not stop an honest run: the count falls back to the device indices"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(util.ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)

    def ident(host, pci, idx, uuid=None):
        return {"host": host, "pci_bus_id": pci, "uuid": uuid, "name": "MI355X", "device_index": idx, "pid": 1000 + idx}

    eight = [ident("n0", "0000:%02x:00.0" % (5 + 16 * i), i) for i in range(8)]
    assert bench.count_distinct_devices(eight) == (8, "pci_bus_id")
    same_pci = [ident("n0", "0000:00:00.0", i) for i in range(8)]
    n, by = bench.count_distinct_devices(same_pci)
    assert n == 8 and by.startswith("device_index")
    no_pci = [ident("n0", None, i) for i in range(4)]
    assert bench.count_distinct_devices(no_pci)[0] == 4
    shared = [ident("n0", "0000:05:00.0", 0) for _ in range(2)]          # two ranks on one GPU: the rehearsal, refused without --share-gpu
    assert bench.count_distinct_devices(shared)[0] == 1
    two_hosts = [ident("n0", "0000:05:00.0", 0), ident("n1", "0000:05:00.0", 0)]
    assert bench.count_distinct_devices(two_hosts) == (2, "pci_bus_id")


def test_committed_bench_line_keeps_the_contract():
    """the newest committed bench line (profiles/r<NN>z_cfg3_bench.json, written by tools/profile_round.sh on the GPU box) carries
    every key the driver's contract names, the roofline and the CPU baseline objects, and numbers that are consistent with
    each other"""
    import glob
    import json
    lines = sorted(glob.glob(os.path.join(util.ROOT, "profiles", "r[0-9][0-9]z_cfg3_bench.json")))
    if not lines:
        pytest.skip("no committed bench line")
    j = json.loads(open(lines[-1]).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["vs_baseline"] is None and j["dtype"] == "f32" and j["data"] == "synthetic" and j["higher_is_better"] is True
    assert "workload" in j["config"] and "4096x4096" in j["config"]["workload"]
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.5 < r["frac"] < 1.0
    assert r["traffic"] is None or 0.9 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.2      # no wasted re-reads
    c = j["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == j["unit"] and c["sample"]
    if "faithful_variants" in c:      # round 5: value = the fastest of the complex128 full-spectrum variants, each with its cores
        fv = {k: v for k, v in c["faithful_variants"].items() if "value" in v}
        assert c["value_from"] in fv and abs(c["value"] - max(v["value"] for v in fv.values())) < 1e-12
        assert "scipy_pocketfft_c128_full" in c["faithful_variants"] and all(v["cores"] >= 1 for v in fv.values())
    # value = maps x padded pixels / step time
    maps, P = 256, 4224 * 4224
    assert abs(j["value"] - maps * P / (j["ms_per_step"] * 1e-3) / 1e9) / j["value"] < 1e-3
