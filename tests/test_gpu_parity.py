"""GPU tier (-m gpu): parity of the HIP path, called through the C ABI (ctypes), against the CPU
oracle and the committed golden fixtures; edge cases and error behaviour of the reference's
interface; size-independent properties at BASELINE.json's full sizes.

Tolerance: max|out - ref| / max|ref| <= 1e-4 per map (north_star: "within 1e-4 relative fp32");
the fp32 engine is in practice ~1e-6, so the small cases assert 1e-5 to catch regressions."""
import ctypes
import os

import numpy as np
import pytest

import golden_util
import util

pytestmark = pytest.mark.gpu
TOL = 1e-4
TIGHT = 1e-5


@pytest.fixture(scope="module")
def fc(fftconv):
    assert fftconv.device_count() >= 1, "GPU tests need a GPU (no CPU fallback exists)"
    return fftconv


@pytest.mark.parametrize("case", golden_util.golden_cases())
def test_golden_fixtures(fc, case):
    data, mkh, mkw, kernels, expect = golden_util.load_case(case)
    got = fc.cudaConvolutionFFT(data, mkh, mkw, kernels)
    assert len(got) == len(expect)
    for g, e in zip(got, expect):
        assert g.shape == e.shape and g.dtype == np.float32 and g.flags.f_contiguous
        assert util.rel_err(g, e) < TIGHT


SHAPES = [
    (64, 8, 5, 10, 4, 3),      # the reference's demo problem (demoCudaConvolutionFFT.m:37-42)
    (33, 47, 3, 7, 5, 2), (100, 90, 2, 13, 17, 2), (256, 256, 1, 31, 31, 1),   # cfg1
    (1, 1, 1, 1, 1, 1), (2, 3, 1, 1, 2, 1), (5, 5, 2, 5, 5, 1), (17, 1, 1, 3, 1, 1), (1, 40, 2, 1, 9, 2),
    (300, 20, 1, 21, 3, 1), (130, 260, 3, 12, 8, 2), (50, 60, 1, 19, 23, 1), (31, 31, 7, 31, 31, 2),
    (500, 333, 2, 40, 27, 3), (1000, 77, 1, 88, 5, 2), (77, 1000, 1, 5, 88, 2), (513, 511, 1, 2, 2, 2),
]


@pytest.mark.parametrize("shape", SHAPES)
def test_parity_vs_oracle(fc, oracle, shape):
    H, W, F, kh, kw, n = shape
    rng = np.random.default_rng(sum(shape))
    data = rng.random((H, W, F), dtype=np.float32)
    ks = [rng.random((kh, kw, F), dtype=np.float32) for _ in range(n)]
    if n > 1:
        ks[1] = rng.random((max(1, kh - 2), max(1, kw - 1), F), dtype=np.float32)   # ragged cell
    got = fc.cudaConvolutionFFT(data, kh, kw, ks, [8, 8, 8, 16], 0)
    ref = oracle.conv_fft(data, kh, kw, ks)
    for g, r in zip(got, ref):
        assert g.shape == (util.ceil16(H + kh - 1), util.ceil16(W + kw - 1))
        assert util.rel_err(g, r) < TIGHT


def test_signed_data_and_large_dynamic_range(fc, oracle):
    rng = np.random.default_rng(9)
    data = (rng.standard_normal((200, 150, 2)) * 1e3).astype(np.float32)
    ks = [(rng.standard_normal((9, 11, 2)) * 1e-3).astype(np.float32)]
    got = fc.cudaConvolutionFFT(data, 9, 11, ks)
    assert util.rel_err(got[0], oracle.conv_fft(data, 9, 11, ks)[0]) < TIGHT


def test_demo_invariants(fc):
    """demoCudaConvolutionFFT.m:110-113: same kernel twice -> identical maps; kernel2(1) = 100
    -> difference is the scaled first data channel at the top-left."""
    data, cn, cm, ks, _ = golden_util.load_case("case_demo")
    n, m, _ = data.shape
    got = fc.cudaConvolutionFFT(data, cn, cm, ks, [8, 8, 8, 16], 0)
    assert np.array_equal(got[0], got[2])
    diff = got[1].astype(np.float64) - got[0]
    want = np.zeros_like(diff)
    want[:n, :m] = (100.0 - float(ks[0][0, 0, 0])) * data[:, :, 0]
    assert np.abs(diff - want).max() / np.abs(want).max() < TOL
    assert np.abs(got[0][n + cn - 1:, :]).max() < TOL * np.abs(got[0]).max()


def test_demo_planted_template_peaks(fc):
    """demoCudaConvolutionFFT.m:57-69 on the GPU path: the planted templates answer at their offsets
    shifted by (cn-1, cm-1), with sum(template^2) = 22140 at the peak (one-shot entry and flip_kernels plan)"""
    data, cn, cm, ks, _ = golden_util.load_case("case_demo")
    golden_util.demo_planted_checks(lambda kernels: fc.cudaConvolutionFFT(data, cn, cm, kernels), data, cn, cm, ks, 1e-5)
    # the same through a plan that does the demo's "Flip Kernel (Required)" step itself
    with fc.Plan(data.shape[0], data.shape[1], data.shape[2], cn, cm) as plan:
        plan.set_option("flip_kernels", 1)
        plan.set_image(data)
        golden_util.demo_planted_checks(lambda kernels: plan.convolve([np.ascontiguousarray(k[::-1, ::-1, :]) for k in kernels]),
                                        data, cn, cm, ks, 1e-5)


def test_empty_cell(fc):
    assert fc.cudaConvolutionFFT(np.zeros((8, 8, 1), np.float32), 3, 3, []) == []


def test_2d_inputs_are_single_channel(fc, oracle):
    # the reference rejects H x W x 1; the engine accepts it
    rng = np.random.default_rng(1)
    data = rng.random((40, 30), dtype=np.float32)
    k = rng.random((5, 7), dtype=np.float32)
    got = fc.cudaConvolutionFFT(data, 5, 7, [k])
    assert util.rel_err(got[0], oracle.conv_fft(data, 5, 7, [k])[0]) < TIGHT


def test_oversize_kernel_wraps_like_the_reference_when_window_is_exact(fc, oracle):
    data, mkh, mkw, kernels, expect = golden_util.load_case("case_wrap")
    got = fc.cudaConvolutionFFT(data, mkh, mkw, kernels)
    assert util.rel_err(got[0], expect[0]) < TIGHT
    assert util.rel_err(got[0], oracle.conv_direct(data, mkh, mkw, kernels[0])) < TIGHT


def test_error_behaviour(fc):
    data = np.zeros((64, 8, 2), np.float32)
    k = np.zeros((10, 4, 2), np.float32)
    with pytest.raises(fc.FFTConvError) as e:    # kernel larger than the FFT window, :242
        fc.cudaConvolutionFFT(data, 10, 4, [np.zeros((81, 4, 2), np.float32)])
    assert e.value.status == -3
    with pytest.raises(fc.FFTConvError) as e:    # > MAXK and transform (80x11) != window (80x16)
        fc.cudaConvolutionFFT(data, 10, 4, [np.zeros((10, 6, 2), np.float32)])
    assert e.value.status == -4
    with pytest.raises(fc.FFTConvError) as e:    # bad gpu id
        fc.cudaConvolutionFFT(data, 10, 4, [k], None, 99)
    assert e.value.status == -6
    # a failure leaves the library usable
    assert len(fc.cudaConvolutionFFT(data, 10, 4, [k])) == 1


def test_two_step_api_and_plan_reuse(fc, oracle):
    """cudaFFTData + cudaConvFFTData (src/cudaFFTData.cu, src/cudaConvFFTData.cu): one image
    spectrum reused across calls; a second image through the same plan"""
    rng = np.random.default_rng(4)
    data = rng.random((90, 70, 3), dtype=np.float32)
    ks = [rng.random((11, 9, 3), dtype=np.float32) for _ in range(5)]
    h = fc.cudaFFTData(data, 11, 9)
    a = fc.cudaConvFFTData(h, ks[:2])
    b = fc.cudaConvFFTData(h, ks[2:], [16, 8, 8, 32])
    ref = oracle.conv_fft(data, 11, 9, ks)
    for g, r in zip(a + b, ref):
        assert util.rel_err(g, r) < TIGHT
    with pytest.raises(fc.FFTConvError):
        fc.cudaConvFFTData(h, ks[:1], [1, 2, 3])
    data2 = rng.random((90, 70, 3), dtype=np.float32)
    h.set_image(data2)
    assert util.rel_err(h.convolve(ks[:1])[0], oracle.conv_fft(data2, 11, 9, ks[:1])[0]) < TIGHT
    h.destroy()


def test_convolve_before_image_fails(fc):
    with fc.Plan(16, 16, 1, 3, 3) as p:
        with pytest.raises(fc.FFTConvError) as e:
            p.convolve([np.zeros((3, 3, 1), np.float32)])
        assert e.value.status == -9


def test_device_resident_packed_path(fc, oracle):
    """the mode the benchmark times: image, kernels and maps all in HBM (torch tensors as plumbing)"""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    H, W, F, kh, kw, n = 300, 260, 2, 15, 13, 7
    img, ks = util.synth(21, H, W, F, kh, kw, n)
    img_d = torch.from_numpy(np.ascontiguousarray(np.transpose(img, (2, 1, 0)))).to(dev)
    kpk = np.stack([np.transpose(k, (2, 1, 0)) for k in ks])
    k_d = torch.from_numpy(np.ascontiguousarray(kpk)).to(dev)
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        with fc.Plan(H, W, F, kh, kw, 0, s.cuda_stream) as p:
            out = torch.empty((n, p.info.fft_w, p.info.fft_h), dtype=torch.float32, device=dev)
            # maps per launch x budget of the kernels' column-spectrum chunk (0 = one launch's worth per chunk;
            # 1 MiB and 512 MiB: chunks spanning one / several launches)
            for batch, chunk_mb in ((0, 0), (1, 0), (3, 0), (3, 1), (2, 512), (0, 1)):
                p.set_option("batch_maps", batch)
                p.set_option("kernel_chunk_mb", chunk_mb)
                out.fill_(float("nan"))
                p.set_image_device(img_d.data_ptr())
                p.convolve_packed_device(n, k_d.data_ptr(), kh, kw, out.data_ptr())
                p.synchronize()
                ref = oracle.conv_fft(img, kh, kw, ks)
                got = out.cpu().numpy()
                for j in range(n):
                    assert util.rel_err(got[j].T, ref[j]) < TIGHT


def test_prepare_kernels_split(fc, oracle):
    """the image-independent half of the packed convolution queued ahead (what overlaps the
    broadcast on the other ranks); a call with other arguments must not reuse it"""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    H, W, F, kh, kw, n = 200, 180, 2, 11, 9, 5
    img, ks = util.synth(51, H, W, F, kh, kw, n)
    k_d = torch.from_numpy(np.ascontiguousarray(np.stack([np.transpose(k, (2, 1, 0)) for k in ks]))).to(dev)
    k2_d = (k_d * 2.0).contiguous()
    ref = oracle.conv_fft(img, kh, kw, ks)
    with fc.Plan(H, W, F, kh, kw) as p:
        out = torch.empty((n, p.info.fft_w, p.info.fft_h), dtype=torch.float32, device=dev)
        p.prepare_kernels_packed_device(n, k_d.data_ptr(), kh, kw)     # before the image exists
        p.set_image(img)
        p.convolve_packed_device(n, k_d.data_ptr(), kh, kw, out.data_ptr())
        p.synchronize()
        for j in range(n):
            assert util.rel_err(out[j].cpu().numpy().T, ref[j]) < TIGHT
        # several column-spectrum chunks per call (2 maps per launch): only the first one is prepared ahead
        p.set_option("batch_maps", 2)
        out.fill_(float("nan"))
        p.prepare_kernels_packed_device(n, k_d.data_ptr(), kh, kw)
        p.convolve_packed_device(n, k_d.data_ptr(), kh, kw, out.data_ptr())
        p.synchronize()
        for j in range(n):
            assert util.rel_err(out[j].cpu().numpy().T, ref[j]) < TIGHT
        p.set_option("batch_maps", 0)
        p.prepare_kernels_packed_device(n, k_d.data_ptr(), kh, kw)
        p.convolve_packed_device(n, k2_d.data_ptr(), kh, kw, out.data_ptr())   # different kernels: recomputed
        p.synchronize()
        for j in range(n):
            assert util.rel_err(out[j].cpu().numpy().T, 2.0 * ref[j]) < TIGHT


def test_placement_tuning_keeps_results(fc, oracle):
    """option tune_placement: candidate allocations of the intermediate are timed against the map buffer
    before the first convolve writes anything; results are those of an untuned plan, the tuning happens once
    per allocation, and the getter reports it (device-resident and host-output destinations)"""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    H, W, F, kh, kw, n = 1024, 1024, 1, 63, 63, 5
    img, ks = util.synth(77, H, W, F, kh, kw, n)
    ref = oracle.conv_fft(img, kh, kw, ks)
    k_d = torch.from_numpy(np.ascontiguousarray(np.stack([np.transpose(k, (2, 1, 0)) for k in ks]))).to(dev)
    with fc.Plan(H, W, F, kh, kw) as p:
        assert p.get_option("tune_placement") == -1 and p.get_option("tuned_candidates") == 0      # -1: automatic (large launches only)
        p.set_option("tune_placement", 3)
        p.set_option("batch_maps", 2)            # three launches per call: every batch's destination is probed
        p.set_image(img)
        out = torch.full((n, p.info.fft_w, p.info.fft_h), float("nan"), dtype=torch.float32, device=dev)
        p.convolve_packed_device(n, k_d.data_ptr(), kh, kw, out.data_ptr())
        p.synchronize()
        assert p.get_option("tuned_candidates") == 3 and 0 <= p.get_option("tuned_best") < 3
        for j in range(n):
            assert util.rel_err(out[j].cpu().numpy().T, ref[j]) < TIGHT
        out.fill_(float("nan"))
        p.convolve_packed_device(n, k_d.data_ptr(), kh, kw, out.data_ptr())   # same allocation: not tuned again
        p.synchronize()
        for j in range(n):
            assert util.rel_err(out[j].cpu().numpy().T, ref[j]) < TIGHT
        with pytest.raises(fc.FFTConvError):
            p.get_option("no_such_option")
    with fc.Plan(H, W, F, kh, kw) as p:          # host-output destination: the staging buffer is what is probed
        p.set_option("tune_placement", 2)
        p.set_image(img)
        got = p.convolve(ks)
        assert p.get_option("tuned_candidates") == 2
        for g, r in zip(got, ref):
            assert util.rel_err(g, r) < TIGHT


def test_placement_tuning_is_automatic_for_large_launches_only(fc):
    """tune_placement left at its default (-1): a plan whose launches write 2 GiB of maps and more tunes at its first convolve when the
    device is mostly free (csrc/placement.cpp: placement_auto_candidates), a small plan never does, and 0 switches it off; the maps are
    the same either way (bit-exact: the candidates differ in where the intermediate lies, not in what is computed)"""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    H = W = 4096
    kh = kw = 63
    n = 40                                        # 40 maps of 4160 x 4160 floats = 2.8 GB per launch
    g = torch.Generator(device="cpu").manual_seed(5)
    img = torch.rand((1, W, H), generator=g, dtype=torch.float32).to(dev)
    ker = torch.rand((n, 1, kw, kh), generator=g, dtype=torch.float32).to(dev)
    free_b, total_b = torch.cuda.mem_get_info(dev)
    outs = []
    for opt in (None, 0):
        with fc.Plan(H, W, 1, kh, kw) as p:
            if opt is not None:
                p.set_option("tune_placement", opt)
            out = torch.empty((n, p.info.fft_w, p.info.fft_h), dtype=torch.float32, device=dev)
            p.set_image_device(img.data_ptr())
            p.convolve_packed_device(n, ker.data_ptr(), kh, kw, out.data_ptr())
            p.synchronize()
            tuned = p.get_option("tuned_candidates")
            if opt == 0:
                assert tuned == 0
            elif free_b >= 0.7 * total_b:         # (a device that other processes fill is allowed to skip it)
                assert tuned == 5 and 0 <= p.get_option("tuned_best") < 5
            outs.append(out)
    assert torch.equal(outs[0], outs[1])
    with fc.Plan(1024, 1024, 1, 63, 63) as p:     # 16 maps of 1088 x 1088: 76 MB per launch
        k16 = torch.rand((16, 1, 63, 63), dtype=torch.float32, device=dev)
        i16 = torch.rand((1, 1024, 1024), dtype=torch.float32, device=dev)
        out = torch.empty((16, p.info.fft_w, p.info.fft_h), dtype=torch.float32, device=dev)
        p.set_image_device(i16.data_ptr())
        p.convolve_packed_device(16, k16.data_ptr(), 63, 63, out.data_ptr())
        p.synchronize()
        assert p.get_option("tuned_candidates") == 0


def test_external_spectrum_buffer_roundtrip(fc, oracle):
    """the multi-GPU hand-off: spectrum produced into caller memory by one plan, consumed by another"""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    H, W, F, kh, kw = 120, 100, 1, 9, 9
    img, ks = util.synth(31, H, W, F, kh, kw, 2)
    with fc.Plan(H, W, F, kh, kw) as src, fc.Plan(H, W, F, kh, kw) as dst:
        buf = torch.empty(src.info.spectrum_bytes, dtype=torch.uint8, device=dev)
        src.use_spectrum_buffer(buf.data_ptr(), buf.numel())
        src.set_image(img)
        src.synchronize()
        buf2 = buf.clone()                       # stands in for the broadcast
        dst.use_spectrum_buffer(buf2.data_ptr(), buf2.numel())
        with pytest.raises(fc.FFTConvError):
            dst.convolve(ks)                     # not marked valid yet
        dst.mark_spectrum_valid()
        got = dst.convolve(ks)
        for g, r in zip(got, oracle.conv_fft(img, kh, kw, ks)):
            assert util.rel_err(g, r) < TIGHT
        with pytest.raises(fc.FFTConvError):
            dst.use_spectrum_buffer(buf2.data_ptr(), 16)


# ---------------------------------------------------------------- BASELINE.json full sizes

def _device_run(fc, torch, img, ks_packed, kh, kw):
    dev = torch.device("cuda", 0)
    F, W, H = img.shape
    n = ks_packed.shape[0]
    with fc.Plan(H, W, F, kh, kw) as p:
        out = torch.empty((n, p.info.fft_w, p.info.fft_h), dtype=torch.float32, device=dev)
        p.set_image_device(img.data_ptr())
        p.convolve_packed_device(n, ks_packed.data_ptr(), kh, kw, out.data_ptr())
        p.synchronize()
    return out


def test_cfg2_full_size_vs_oracle(fc, oracle):
    """BASELINE configs[1]: 1024x1024 image, 16 kernels of 63x63 -> 16 maps of 1088x1088"""
    img, ks = util.synth(2, 1024, 1024, 1, 63, 63, 16)
    got = fc.cudaConvolutionFFT(img, 63, 63, ks)
    ref = oracle.conv_fft(img, 63, 63, ks)
    for g, r in zip(got, ref):
        assert util.rel_err(g, r) < TOL


def test_cfg3_full_size_properties(fc, oracle):
    """BASELINE configs[2] geometry (4096x4096 image, 127x127 kernels, 4224x4224 maps), checked
    through size-independent properties plus one map against the oracle."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    H = W = 4096
    kh = kw = 127
    g = torch.Generator(device="cpu").manual_seed(1234 + 3)
    img = torch.rand((1, W, H), generator=g, dtype=torch.float32)
    ks = torch.rand((4, 1, kw, kh), generator=g, dtype=torch.float32)
    ks[1].zero_(); ks[1, 0, 5, 9] = 1.0          # delta at (h=9, w=5)
    ks[3] = 2.0 * ks[0] - 3.0 * ks[2]            # linear combination
    out = _device_run(fc, torch, img.to(dev), ks.to(dev), kh, kw)
    fw, fh = out.shape[1], out.shape[2]
    assert (fh, fw) == (4224, 4224)
    o = out.double()
    scale = float(o[0].abs().max())
    # delta kernel -> the image shifted by (9, 5), zero elsewhere
    want = torch.zeros((fw, fh), dtype=torch.float64, device=dev)
    want[5:5 + W, 9:9 + H] = img[0].to(dev).double()
    assert float((o[1] - want).abs().max()) < TOL
    # linearity
    assert float((o[3] - (2.0 * o[0] - 3.0 * o[2])).abs().max()) / scale < TOL
    # checksum: sum(map) = sum(image) * sum(kernel)
    for j in (0, 2):
        s_map = float(o[j].sum())
        s_ref = float(img.double().sum()) * float(ks[j].double().sum())
        assert abs(s_map - s_ref) / abs(s_ref) < 1e-5
    # outside the linear support (4222 x 4222) the window is ~0
    assert float(o[0][4222:, :].abs().max()) / scale < TOL and float(o[0][:, 4222:].abs().max()) / scale < TOL
    # one full map against the oracle
    img_np = np.asfortranarray(np.transpose(img.numpy(), (2, 1, 0)))
    k_np = np.asfortranarray(np.transpose(ks[0].numpy(), (2, 1, 0)))
    ref = oracle.conv_fft(img_np, kh, kw, [k_np])[0]
    assert util.rel_err(out[0].cpu().numpy().T, ref) < TOL


# ---- the vendor's FFT library on the same device as a third, independent statement of the path (test-only: SURVEY 8(c)) ----------
# The reference's arithmetic lives in cuFFT (src/cudaConvolutionFFT.cu:128,136,167,255,273), which cannot run here; its
# counterpart on this platform, rocFFT behind hipFFT, is what torch.fft uses on a ROCm device.  This restates the reference's own
# sequence on it -- zero-pad to [F][FFT_W][FFT_H] (cuh:24-30), R2C with the halved dimension = H (the plan geometry of :122-142),
# product scaled by 1 / (FFT_W * FFT_H) (cuh:62-65, :270), C2R, sum over the features (cuh:84-90) -- in fp32 as the reference runs
# it and in fp64, and holds BOTH the product path and the CPU oracle against it.  The library never links or calls it.
def _vendor_fft_conv(torch, data, mkh, mkw, kernels, dtype):
    dev = torch.device("cuda", 0)
    H, W, F = data.shape
    fh, fw = util.ceil16(H + mkh - 1), util.ceil16(W + mkw - 1)
    d = torch.zeros((F, fw, fh), dtype=dtype, device=dev)
    d[:, :W, :H] = torch.from_numpy(np.ascontiguousarray(np.transpose(data, (2, 1, 0)))).to(dev, dtype)
    D = torch.fft.rfft2(d)                                   # last dimension (h) halved: cuFFT's n = {FFT_W, FFT_H}
    outs = []
    for k in kernels:
        kh, kw = k.shape[0], k.shape[1]
        kp = torch.zeros((F, fw, fh), dtype=dtype, device=dev)
        kp[:, :kw, :kh] = torch.from_numpy(np.ascontiguousarray(np.transpose(k, (2, 1, 0)))).to(dev, dtype)
        prod = D * torch.fft.rfft2(kp)
        # torch's irfft2 normalises by 1 / (FFT_W * FFT_H): the reference's explicit scale of the product (:270)
        outs.append(torch.fft.irfft2(prod, s=(fw, fh)).sum(dim=0).cpu().numpy().T)      # [w][h] -> h x w
    return outs


@pytest.mark.parametrize("shape", [
    (64, 8, 5, 10, 4, 3),          # the demo's problem (demoCudaConvolutionFFT.m:37-42)
    (256, 256, 1, 31, 31, 1),      # cfg1
    (1024, 1024, 1, 63, 63, 2),    # cfg2's geometry
    (2048, 2048, 1, 63, 63, 1),    # cfg5's
    (4096, 4096, 1, 127, 127, 1),  # cfg3's
    (4096, 4096, 1, 63, 63, 1),    # cfg4's (window 4160 cropped from a 4224 transform)
    (300, 260, 3, 31, 17, 2),
])
def test_matches_the_vendor_fft_library_on_the_device(fc, oracle, shape):
    torch = pytest.importorskip("torch")
    H, W, F, kh, kw, n = shape
    rng = np.random.default_rng(sum(shape) + 11)
    data = rng.random((H, W, F), dtype=np.float32)
    ks = [rng.random((kh, kw, F), dtype=np.float32) for _ in range(n)]
    got = fc.cudaConvolutionFFT(data, kh, kw, ks)
    ref32 = _vendor_fft_conv(torch, data, kh, kw, ks, torch.float32)
    ref64 = _vendor_fft_conv(torch, data, kh, kw, ks, torch.float64)
    orc = oracle.conv_fft(data, kh, kw, ks)
    for g, r32, r64, o in zip(got, ref32, ref64, orc):
        assert g.shape == r32.shape == o.shape
        assert util.rel_err(g, r64) < TIGHT            # the product against rocFFT in double precision
        assert util.rel_err(g, r32) < TOL              # ... and against the reference's own precision on the vendor library (north_star's bar)
        assert util.rel_err(o, r64) < 1e-6             # the CPU oracle against it: two independent float64 statements (the oracle returns fp32 maps)


def test_cfg3_headline_launch_geometry(fc, oracle):
    """The launch geometry bench.py times (BASELINE configs[2]): 4096x4096 image, 127x127 kernels,
    84 kernels in one call = one full 64-map launch (four walks of 16 maps per workgroup of the
    spectral-row kernel) + a 20-map launch (shorter walks), the persistent output kernel over
    64 x 528 and 20 x 528 tiles.  Every map is checked on the device -- checksum identity
    sum(map) = sum(image) * sum(kernel), a delta kernel (shifted image), linear combinations of
    earlier kernels -- and three sampled maps (first launch, launch boundary, second launch)
    against the oracle (src/cudaConvolutionFFT.cu:204-291 is the loop this batches)."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    H = W = 4096
    kh = kw = 127
    n = 84
    g = torch.Generator(device="cpu").manual_seed(1234 + 33)
    img = torch.rand((1, W, H), generator=g, dtype=torch.float32)
    ks = torch.rand((n, 1, kw, kh), generator=g, dtype=torch.float32)
    ks[17].zero_(); ks[17, 0, 100, 3] = 1.0               # delta at (h = 3, w = 100): inside the first launch
    ks[70].zero_(); ks[70, 0, 0, 126] = 1.0               # delta at (h = 126, w = 0): second launch
    combos = {40: (2.0, 5, -3.0, 31), 66: (0.5, 63, 1.5, 64), 83: (-1.0, 0, 4.0, 79)}   # j -> a * k[p] + b * k[q], across walks and launches
    for j, (a, p_, b, q_) in combos.items():
        ks[j] = a * ks[p_] + b * ks[q_]
    out = _device_run(fc, torch, img.to(dev), ks.to(dev), kh, kw)
    assert tuple(out.shape) == (n, 4224, 4224)
    img_d = img.to(dev)
    # checksum identity, all maps
    s_img = float(img.double().sum())
    s_ker = ks.double().sum(dim=(1, 2, 3))
    s_map = out.sum(dim=(1, 2), dtype=torch.float64).cpu()
    want = s_img * s_ker
    assert float(((s_map - want).abs() / want.abs()).max()) < 1e-5
    # delta kernels -> the image shifted, zero elsewhere
    for j, (dh, dw) in {17: (3, 100), 70: (126, 0)}.items():
        w_ = torch.zeros((4224, 4224), dtype=torch.float32, device=dev)
        w_[dw:dw + W, dh:dh + H] = img_d[0]
        assert float((out[j] - w_).abs().max()) < TOL
        del w_
    # linear combinations against the maps of their terms
    for j, (a, p_, b, q_) in combos.items():
        d = (out[j].double() - (a * out[p_].double() + b * out[q_].double())).abs().max()
        assert float(d) / float(out[j].abs().max()) < TOL
    # nothing outside the linear support (4222 x 4222), any map
    scale = float(out.abs().amax())
    assert float(out[:, 4222:, :].abs().amax()) / scale < TOL and float(out[:, :, 4222:].abs().amax()) / scale < TOL
    # sampled maps against the oracle
    idx = [15, 63, 64]
    img_np = np.asfortranarray(np.transpose(img.numpy(), (2, 1, 0)))
    k_np = [np.asfortranarray(np.transpose(ks[j].numpy(), (2, 1, 0))) for j in idx]
    ref = oracle.conv_fft(img_np, kh, kw, k_np)
    for j, r in zip(idx, ref):
        assert util.rel_err(out[j].cpu().numpy().T, r) < TOL


def _device_run_opts(fc, torch, img, ks_packed, kh, kw, opts):
    dev = torch.device("cuda", 0)
    F, W, H = img.shape
    n = ks_packed.shape[0]
    with fc.Plan(H, W, F, kh, kw) as p:
        for k, v in opts.items():
            p.set_option(k, v)
            assert p.get_option(k) == v
        out = torch.empty((n, p.info.fft_w, p.info.fft_h), dtype=torch.float32, device=dev)
        if opts.get("defer_prepare"):        # kernels' columns ride in the image's column launch (k_fast_cols_fwd_pair)
            p.prepare_kernels_packed_device(n, ks_packed.data_ptr(), kh, kw)
        p.set_image_device(img.data_ptr())
        p.convolve_packed_device(n, ks_packed.data_ptr(), kh, kw, out.data_ptr())
        p.synchronize()
    return out


# (H, W, F, K, n): cfg4's share geometry (window 4160 cropped from 4224, a full launch + a partial one), cfg3's, cfg5's (2112),
# 4-column output tiles (M = 3072), a small transform whose launch would be sliced (static fall-back inside the dynamic
# plan), F > 1, and a launch of fewer tiles than workgroups
DYNAMIC_TILE_SHAPES = [(4096, 4096, 1, 63, 70), (4096, 4096, 1, 127, 9), (2048, 2048, 1, 63, 33), (6000, 700, 1, 63, 5),
                       (1024, 1024, 1, 63, 16), (1500, 1400, 3, 31, 7), (256, 256, 1, 31, 1)]


@pytest.mark.parametrize("shape", DYNAMIC_TILE_SHAPES)
@pytest.mark.parametrize("defer", [0, 1])
def test_dynamic_tile_queue_matches_static_deal(fc, oracle, shape, defer):
    """Plan option "dynamic_tiles" (fast_cols.hpp: TileQueue; the default of the output kernel from M = 432 on): the persistent column kernels take their tiles from a
    queue in device memory -- robust where a step shares the GPU with a collective (src/cudaConvFFTDataStreams.cu:279-289,338-447).
    A tile is computed by the same code whoever takes it, so the maps must equal the static deal's BIT FOR BIT; the first
    and the last map also go against the oracle.  defer = 1: the image's and the kernels' column passes in one launch."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    H, W, F, K, n = shape
    g = torch.Generator(device="cpu").manual_seed(77 + H + n)
    img = torch.rand((F, W, H), generator=g, dtype=torch.float32)
    ks = torch.rand((n, F, K, K), generator=g, dtype=torch.float32)
    base = {"defer_prepare": 1} if defer else {}
    ref_out = _device_run_opts(fc, torch, img.to(dev), ks.to(dev), K, K, dict(base, dynamic_tiles=0))
    dyn_out = _device_run_opts(fc, torch, img.to(dev), ks.to(dev), K, K, dict(base, dynamic_tiles=1))
    assert torch.equal(ref_out, dyn_out)
    again = _device_run_opts(fc, torch, img.to(dev), ks.to(dev), K, K, dict(base, dynamic_tiles=2))   # counters zeroed per launch; 2: the forward column kernels through their counters too
    assert torch.equal(dyn_out, again)
    if H * W <= 2048 * 2048 or not defer:
        idx = sorted({0, n - 1})
        img_np = np.asfortranarray(np.transpose(img.numpy(), (2, 1, 0)))
        k_np = [np.asfortranarray(np.transpose(ks[j].numpy(), (2, 1, 0))) for j in idx]
        for j, r in zip(idx, oracle.conv_fft(img_np, K, K, k_np)):
            assert util.rel_err(dyn_out[j].cpu().numpy().T, r) < TOL


def test_dynamic_tile_queue_host_entries(fc, oracle):
    """the queue behind the host-array entries (plan convolve with pointer arrays, host output) and across a change of the option"""
    rng = np.random.default_rng(5)
    img = rng.random((1800, 1700, 1), dtype=np.float32)
    ks = [rng.random((40, 50, 1), dtype=np.float32) for _ in range(5)] + [rng.random((17, 9, 1), dtype=np.float32)]
    ref = oracle.conv_fft(img, 40, 50, ks)
    with fc.Plan(1800, 1700, 1, 40, 50) as p:
        p.set_image(img)
        assert p.get_option("dynamic_tiles") == 1       # the default from 864-point transforms on (here 1920 along h)
        for dyn in (2, 0, 1):
            p.set_option("dynamic_tiles", dyn)
            if dyn:
                p.set_image(img)                 # the image's column pass through the queue too
            for g, r in zip(p.convolve(ks), ref):
                assert util.rel_err(g, r) < TOL


def test_cfg4_headline_launch_geometry(fc, oracle):
    """The launch geometry bench.py times for BASELINE configs[3] (one rank's share, and the 1-GPU denominator):
    4096x4096 image, 63x63 kernels, 84 kernels in one packed call = a full 64-map launch of the multi-map row
    kernel (16-map walks) + a 20-map launch, the 4160 x 4160 window CROPPED from the 4224 x 4224 transform (the
    `wout < L` store branch of the row kernel, the cropped rows of the output kernel).  All maps on the device by
    checksum identity, delta kernels and linear combinations; three maps across the launch boundary against the
    oracle (src/cudaConvolutionFFT.cu:204-291 is the loop this batches)."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    H = W = 4096
    kh = kw = 63
    n = 84
    g = torch.Generator(device="cpu").manual_seed(1234 + 44)
    img = torch.rand((1, W, H), generator=g, dtype=torch.float32)
    ks = torch.rand((n, 1, kw, kh), generator=g, dtype=torch.float32)
    ks[9].zero_(); ks[9, 0, 62, 5] = 1.0                  # delta at (h = 5, w = 62): first launch
    ks[77].zero_(); ks[77, 0, 1, 62] = 1.0                # delta at (h = 62, w = 1): second launch
    combos = {33: (2.0, 4, -3.0, 30), 65: (0.5, 63, 1.5, 64), 82: (-1.0, 1, 4.0, 80)}
    for j, (a, p_, b, q_) in combos.items():
        ks[j] = a * ks[p_] + b * ks[q_]
    with fc.Plan(H, W, 1, kh, kw) as p:
        assert (p.info.fft_h, p.info.fft_w) == (4160, 4160) and (p.info.transform_h, p.info.transform_w) == (4224, 4224)
    out = _device_run(fc, torch, img.to(dev), ks.to(dev), kh, kw)
    assert tuple(out.shape) == (n, 4160, 4160)
    img_d = img.to(dev)
    s_img = float(img.double().sum())
    s_ker = ks.double().sum(dim=(1, 2, 3))
    s_map = out.sum(dim=(1, 2), dtype=torch.float64).cpu()
    want = s_img * s_ker
    assert float(((s_map - want).abs() / want.abs()).max()) < 1e-5
    for j, (dh, dw) in {9: (5, 62), 77: (62, 1)}.items():
        w_ = torch.zeros((4160, 4160), dtype=torch.float32, device=dev)
        w_[dw:dw + W, dh:dh + H] = img_d[0]
        assert float((out[j] - w_).abs().max()) < TOL
        del w_
    for j, (a, p_, b, q_) in combos.items():
        d = (out[j].double() - (a * out[p_].double() + b * out[q_].double())).abs().max()
        assert float(d) / float(out[j].abs().max()) < TOL
    scale = float(out.abs().amax())     # nothing outside the linear support (4158 x 4158); here written as exact zeros
    assert float(out[:, 4158:, :].abs().amax()) / scale < TOL and float(out[:, :, 4158:].abs().amax()) / scale < TOL
    idx = [15, 63, 64]
    img_np = np.asfortranarray(np.transpose(img.numpy(), (2, 1, 0)))
    k_np = [np.asfortranarray(np.transpose(ks[j].numpy(), (2, 1, 0))) for j in idx]
    ref = oracle.conv_fft(img_np, kh, kw, k_np)
    for j, r in zip(idx, ref):
        assert util.rel_err(out[j].cpu().numpy().T, r) < TOL


def test_cfg5_headline_streamed_geometry(fc, oracle):
    """BASELINE configs[4] as bench.py --images runs it on every rank: 2048x2048 images STREAMED from pinned host
    memory through ImageStreamedConvolver (H2D of image i + 1 on the side stream beside the maps of image i, two
    device buffers), 64 kernels of 63x63 per image -> 64 maps of 2112 x 2112 in one launch (2112 = 8.12.22, two rows
    per row workgroup).  Three images; after EACH image every map is checked on the device (checksum identity; a
    delta kernel = the shifted image; a linear combination), and three maps of the last image go against the oracle."""
    import importlib
    torch = pytest.importorskip("torch")
    mg = importlib.import_module(fc.__name__ + ".multi_gpu")
    dev = torch.device("cuda", 0)
    H = W = 2048
    kh = kw = 63
    n = 64
    g = torch.Generator(device="cpu").manual_seed(1234 + 55)
    imgs = [torch.rand((1, W, H), generator=g, dtype=torch.float32).pin_memory() for _ in range(3)]
    ks = torch.rand((n, 1, kw, kh), generator=g, dtype=torch.float32)
    ks[20].zero_(); ks[20, 0, 7, 40] = 1.0               # delta at (h = 40, w = 7)
    ks[50] = 2.0 * ks[3] - 0.5 * ks[63]
    kern_d = ks.to(dev)
    stream = torch.cuda.current_stream(dev)
    s_ker = ks.double().sum(dim=(1, 2, 3))
    with fc.Plan(H, W, 1, kh, kw, gpuId=0, stream=stream.cuda_stream) as plan:
        assert (plan.info.fft_h, plan.info.fft_w) == (2112, 2112)
        engine = mg.HipPlanEngine(torch, fc, plan, dev, kern_d, kh, kw, first=0, main_stream=stream, overlap=True)
        conv = mg.ImageStreamedConvolver(engine, n)
        seen = []

        def check(i, out):
            torch.cuda.synchronize(dev)
            img_d = imgs[i].to(dev)
            want = float(imgs[i].double().sum()) * s_ker
            got = out.sum(dim=(1, 2), dtype=torch.float64).cpu()
            assert float(((got - want).abs() / want.abs()).max()) < 1e-5, i
            w_ = torch.zeros((2112, 2112), dtype=torch.float32, device=dev)
            w_[7:7 + W, 40:40 + H] = img_d[0]
            assert float((out[20] - w_).abs().max()) < TOL, i
            d = (out[50].double() - (2.0 * out[3].double() - 0.5 * out[63].double())).abs().max()
            assert float(d) / float(out[50].abs().max()) < TOL, i
            seen.append(i)

        out = conv.run(imgs, on_result=check)
        assert seen == [0, 1, 2] and conv.last_buf == 0
        torch.cuda.synchronize(dev)
        idx = [0, 31, 63]
        img_np = np.asfortranarray(np.transpose(imgs[2].numpy(), (2, 1, 0)))
        k_np = [np.asfortranarray(np.transpose(ks[j].numpy(), (2, 1, 0))) for j in idx]
        ref = oracle.conv_fft(img_np, kh, kw, k_np)
        for j, r in zip(idx, ref):
            assert util.rel_err(out[j].cpu().numpy().T, r) < TOL


def test_cfg4_and_cfg5_geometry_vs_oracle(fc, oracle):
    """BASELINE configs[3] (4160x4160 maps, 63x63 kernels) and configs[4] (2048x2048 image ->
    2112x2112): one map each against the oracle."""
    for seed, (H, W, kh, kw) in {4: (4096, 4096, 63, 63), 5: (2048, 2048, 63, 63)}.items():
        img, ks = util.synth(seed, H, W, 1, kh, kw, 1)
        got = fc.cudaConvolutionFFT(img, kh, kw, ks)
        ref = oracle.conv_fft(img, kh, kw, ks)
        assert util.rel_err(got[0], ref[0]) < TOL


# ---- host-output streaming (pinned ring, copy stream, host copy threads) --------------------------

@pytest.mark.parametrize("ring", [
    {},                                                                      # default: direct copies by 2 host threads
    {"host_stream": 1, "host_threads": 3},
    {"host_stream": 1, "host_threads": 1},
    {"host_stream": 2},                                                      # pinned ring, default shape
    {"host_stream": 2, "host_chunk_kb": 4, "host_slots": 3, "host_threads": 2},   # many wraps of a tiny ring
    {"host_stream": 2, "host_chunk_kb": 64, "host_slots": 2, "host_threads": 1},
    {"host_stream": 0},                                                      # blocking copy-out
])
def test_host_output_streaming_matches_oracle(fc, oracle, ring):
    """More batches than staging buffers (batch_maps = 2, 9 kernels in ragged groups): every
    map must arrive complete and in its own buffer whatever the ring shape."""
    rng = np.random.default_rng(77)
    H, W, F, kh, kw = 120, 70, 2, 9, 6
    data = rng.random((H, W, F), dtype=np.float32)
    ks = [rng.random((kh, kw, F), dtype=np.float32) for _ in range(7)]
    ks += [rng.random((kh - 3, kw - 1, F), dtype=np.float32) for _ in range(2)]
    ref = oracle.conv_fft(data, kh, kw, ks)
    with fc.Plan(H, W, F, kh, kw) as plan:
        plan.set_option("batch_maps", 2)
        plan.set_option("host_min_kb", 0)       # these 41-KB maps through the streaming machinery all the same
        for k, v in ring.items():
            plan.set_option(k, v)
        plan.set_image(data)
        for rep in range(2):         # second call reuses ring and staging
            got = plan.convolve(ks)
            for g, r in zip(got, ref):
                assert util.rel_err(g, r) < TIGHT
        # caller buffers, pre-filled with garbage
        bufs = [np.full(ref[0].shape, np.nan, dtype=np.float32, order="F") for _ in ks]
        plan.convolve(ks, out=bufs)
        for g, r in zip(bufs, ref):
            assert util.rel_err(g, r) < TIGHT


@pytest.mark.parametrize("shape", [
    (64, 8, 5, 10, 4, 3, 0),          # the reference's demo: everything read in place / one copy
    (256, 256, 1, 31, 31, 4, 0),      # cfg1's image (256 KiB: in place), 324-KiB maps
    (500, 262, 1, 9, 7, 29, 0),       # image 512 KB (above the in-place limit: one asynchronous copy), 545-KB maps: 29 of
                                      # them = 4 copies of 7 maps + 1 (two halves of the pinned buffer alternate)
    (300, 200, 2, 12, 12, 6, 2),      # F = 2, cropped ("same") maps
    (90, 130, 1, 40, 50, 150, 0),     # kernel groups above 512 KiB (75 and 74 x 8 KB): the plain per-kernel copies beside pinned maps
])
def test_small_host_arrays_through_pinned_staging(fc, oracle, shape):
    """The small-call path (plan option host_pinned, default on): host image / kernels / maps of a few hundred KB travel
    through pinned buffers of the plan.  Same bits as the plain copies (host_pinned = 0) and the oracle's maps, also when a
    call is repeated with other data on the same plan (the buffers are refilled behind the GPU work that read them)."""
    H, W, F, kh, kw, n, region = shape
    rng = np.random.default_rng(H * 1009 + W)
    pristine = [rng.standard_normal((H, W, F)).astype(np.float32) for _ in range(2)]
    ks = [rng.standard_normal((kh, kw, F)).astype(np.float32) for _ in range(n)]
    if n > 2:
        ks[n // 2] = rng.standard_normal((max(1, kh - 1), max(1, kw - 2), F)).astype(np.float32)   # a mixed cell: three groups
    got = {}
    for pinned in (1, 0):
        with fc.Plan(H, W, F, kh, kw) as plan:
            assert plan.get_option("host_pinned") == 1
            plan.set_option("host_pinned", pinned)
            if region:
                plan.set_option("output_region", region)
            res = []
            for src in pristine:
                d = src.copy()
                plan.set_image(d)
                d[:] = np.nan                      # the caller's array is its own again as soon as set_image returns
                res.append(plan.convolve(ks))
            got[pinned] = res
    for a, b in zip(got[1], got[0]):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    if region == 0:
        for d, res in zip(pristine, got[1]):
            for g, r in zip(res, oracle.conv_fft(d, kh, kw, ks)):
                assert util.rel_err(g, r) < TIGHT


def test_one_shot_small_calls_back_to_back(fc, oracle):
    """cached one-shot calls with changing small inputs (the MATLAB user's loop): every call returns its own maps"""
    rng = np.random.default_rng(5150)
    H, W, F, kh, kw, n = 64, 8, 5, 10, 4, 3
    for rep in range(6):
        data = rng.standard_normal((H, W, F)).astype(np.float32)
        ks = [rng.standard_normal((kh, kw, F)).astype(np.float32) for _ in range(n)]
        got = fc.cudaConvolutionFFT(data, kh, kw, ks)
        for g, r in zip(got, oracle.conv_fft(data, kh, kw, ks)):
            assert util.rel_err(g, r) < TIGHT


def test_host_output_into_pinned_buffers(fc, oracle):
    """Buffers the caller pinned itself: plain DMA in the direct mode, no ring hop in the ring mode."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(78)
    H, W, kh, kw = 200, 90, 11, 7
    data = rng.random((H, W, 1), dtype=np.float32)
    ks = [rng.random((kh, kw, 1), dtype=np.float32) for _ in range(5)]
    ref = oracle.conv_fft(data, kh, kw, ks)
    fh, fw = ref[0].shape
    pinned = [torch.full((fw, fh), float("nan"), dtype=torch.float32).pin_memory() for _ in ks]
    bufs = [t.numpy().T for t in pinned]            # FFT_H x FFT_W Fortran-order views
    assert all(b.flags.f_contiguous for b in bufs)
    with fc.Plan(H, W, 1, kh, kw) as plan:
        plan.set_option("batch_maps", 2)
        plan.set_option("host_min_kb", 0)
        plan.set_image(data)
        for mode in (1, 2):
            for b in bufs:
                b.fill(np.nan)
            plan.set_option("host_stream", mode)
            plan.convolve(ks, out=bufs)
            for g, r in zip(bufs, ref):
                assert util.rel_err(g, r) < TIGHT


def test_host_output_streaming_large_maps(fc, oracle):
    """cfg2-sized maps (4.7 MB each, several ring chunks per map) through the one-shot entry."""
    img, ks = util.synth(2, 1024, 1024, 1, 63, 63, 5)
    got = fc.cudaConvolutionFFT(img, 63, 63, ks)
    ref = oracle.conv_fft(img, 63, 63, ks)
    for g, r in zip(got, ref):
        assert util.rel_err(g, r) < TOL


# ---- flip_kernels: the demo's "Flip Kernel (Required)" step done on the device ---------------------

def test_flip_kernels_option_equals_flipping_by_hand(fc, oracle):
    """demoCudaConvolutionFFT.m:63-69 flips every kernel (end:-1:1, end:-1:1, :) before the call so
    that the convolution acts as template matching; with flip_kernels = 1 the plan does it."""
    rng = np.random.default_rng(5)
    H, W, F, kh, kw = 90, 75, 3, 8, 5
    data = rng.random((H, W, F), dtype=np.float32)
    ks = [rng.random((kh, kw, F), dtype=np.float32) for _ in range(4)]
    ks.append(rng.random((kh - 1, kw - 2, F), dtype=np.float32))
    flipped = [np.ascontiguousarray(k[::-1, ::-1, :]) for k in ks]
    with fc.Plan(H, W, F, kh, kw) as plan:
        plan.set_image(data)
        by_hand = plan.convolve(flipped)
        plan.set_option("flip_kernels", 1)
        on_device = plan.convolve(ks)
        plan.set_option("flip_kernels", 0)
        plain = plan.convolve(ks)
    ref = oracle.conv_fft(data, kh, kw, flipped)
    for a, b, c, r in zip(on_device, by_hand, plain, ref):
        assert np.array_equal(a, b)
        assert not np.array_equal(a, c)
        assert util.rel_err(a, r) < TIGHT
    # template matching: planting kernel 0 in the data makes its correlation peak there
    data2 = data.copy()
    data2[20:20 + kh, 30:30 + kw, :] = 3.0 * ks[0]
    with fc.Plan(H, W, F, kh, kw) as plan:
        plan.set_option("flip_kernels", 1)
        plan.set_image(data2)
        m = plan.convolve([ks[0]])[0]
    assert np.unravel_index(np.argmax(m), m.shape) == (20 + kh - 1, 30 + kw - 1)


# ---- block-wise (overlap-add) one-shot path for sizes beyond one plan ------------------------------

@pytest.mark.parametrize("shape", [
    (150, 40, 2, 9, 7, 3),      # h tiled only (max_transform = 64: blocks of 56 x W)
    (40, 170, 1, 5, 11, 2),     # w tiled only
    (130, 140, 2, 12, 10, 3),   # both: 3 x 3 blocks, ragged edges
    (57, 57, 1, 9, 9, 1),       # one sample more than a block
])
def test_blockwise_one_shot_matches_oracle(fc, oracle, shape):
    """fftconv_convolution_fft falls back to overlap-add over ordinary plans when the padded size
    does not fit one plan; fftconv_plan_options.max_transform makes small problems take that path."""
    H, W, F, kh, kw, n = shape
    rng = np.random.default_rng(sum(shape))
    data = rng.standard_normal((H, W, F)).astype(np.float32)
    ks = [rng.standard_normal((kh, kw, F)).astype(np.float32) for _ in range(n)]
    if n > 1:
        ks[1] = rng.standard_normal((kh - 2, kw - 1, F)).astype(np.float32)      # ragged cell
    ref = oracle.conv_fft(data, kh, kw, ks)
    direct = fc.cudaConvolutionFFT(data, kh, kw, ks)
    small = {"max_transform": 64}
    with pytest.raises(fc.FFTConvError) as ei:
        fc.Plan(H, W, F, kh, kw, options=dict(small, blockwise=1))   # opted out of block-wise plans: the limit is reported
    assert ei.value.status == -5
    got = fc.cudaConvolutionFFT(data, kh, kw, ks, options=small)
    for g, d, r in zip(got, direct, ref):
        assert g.shape == r.shape
        assert util.rel_err(g, r) < TIGHT
        assert util.rel_err(g, d) < TIGHT
    # kernels beyond MAX_KERNEL cannot be folded block-wise: rejected, not wrapped
    with pytest.raises(fc.FFTConvError) as ei:
        fc.cudaConvolutionFFT(data, kh - 1, kw, ks, options=small)
    assert ei.value.status == -4


def test_blockwise_one_shot_beyond_the_single_pass_limit(fc, oracle):
    """a really long dimension (W + kw - 1 = 21000 > the LDS-resident single pass): no test hook"""
    rng = np.random.default_rng(21)
    H, W, kh, kw = 24, 20990, 5, 11
    data = rng.random((H, W, 1), dtype=np.float32)
    ks = [rng.random((kh, kw, 1), dtype=np.float32) for _ in range(2)]
    with pytest.raises(fc.FFTConvError) as ei:
        fc.Plan(H, W, 1, kh, kw, options={"blockwise": 1})
    assert ei.value.status == -5
    ref = oracle.conv_fft(data, kh, kw, ks)
    got = fc.cudaConvolutionFFT(data, kh, kw, ks)
    for g, r in zip(got, ref):
        assert util.rel_err(g, r) < TIGHT
    # the same size through the two-step API (cudaFFTData / cudaConvFFTData, src/cudaFFTData.cu:72-103,
    # src/cudaConvFFTData.cu:92-98: the reference plans cuFFT for any size): the handle is a block-wise plan
    h = fc.cudaFFTData(data, kh, kw)
    assert h.get_option("blockwise") > 1 and (h.info.fft_h, h.info.fft_w) == (util.ceil16(H + kh - 1), util.ceil16(W + kw - 1))
    for rep in range(2):
        for g, r in zip(fc.cudaConvFFTData(h, ks), ref):
            assert util.rel_err(g, r) < TIGHT
    h.destroy()


@pytest.mark.parametrize("shape", [
    (150, 40, 2, 9, 7, 3),      # h tiled only
    (130, 140, 2, 12, 10, 3),   # both dimensions, ragged edge blocks
])
def test_blockwise_plan_api_every_entry(fc, oracle, shape):
    """Block-wise plans behind the plan API (max_transform = 64 makes small problems take it): host and
    device-resident images, pointer-array and packed device-resident outputs, a second image through the same plan,
    the multi-device handle (the spectrum copied between plans is every block's), and what has no block-wise form
    fails with a message instead of computing something else"""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    H, W, F, kh, kw, n = shape
    rng = np.random.default_rng(sum(shape) + 1)
    small = {"max_transform": 64}
    data = rng.standard_normal((H, W, F)).astype(np.float32)
    data2 = rng.standard_normal((H, W, F)).astype(np.float32)
    ks = [rng.standard_normal((kh, kw, F)).astype(np.float32) for _ in range(n)]
    ref, ref2 = oracle.conv_fft(data, kh, kw, ks), oracle.conv_fft(data2, kh, kw, ks)
    fh, fw = util.ceil16(H + kh - 1), util.ceil16(W + kw - 1)
    with fc.Plan(H, W, F, kh, kw, options=small) as p:
        assert p.get_option("blockwise") > 1 and (p.info.fft_h, p.info.fft_w) == (fh, fw) and p.info.map_bytes == fh * fw * 4
        p.set_image(data)
        for g, r in zip(p.convolve(ks), ref):
            assert g.shape == (fh, fw) and util.rel_err(g, r) < TIGHT
        # device-resident image, packed device-resident kernels and maps (what bench.py uses)
        img_d = torch.from_numpy(np.ascontiguousarray(np.transpose(data2, (2, 1, 0)))).to(dev)
        k_d = torch.from_numpy(np.ascontiguousarray(np.stack([np.transpose(k, (2, 1, 0)) for k in ks]))).to(dev)
        out = torch.full((n, fw, fh), float("nan"), dtype=torch.float32, device=dev)
        p.set_image_device(img_d.data_ptr())
        p.prepare_kernels_packed_device(n, k_d.data_ptr(), kh, kw)
        p.convolve_packed_device(n, k_d.data_ptr(), kh, kw, out.data_ptr())
        p.synchronize()
        for j, r in enumerate(ref2):
            assert util.rel_err(out[j].cpu().numpy().T, r) < TIGHT
        # "output_region" on a block-wise plan (overlap-add here): the region is cropped out of the full-window maps on delivery
        for region, (oh, ow, o_h, o_w) in {1: (H + kh - 1, W + kw - 1, 0, 0), 2: (H, W, (kh - 1) // 2, (kw - 1) // 2),
                                           3: (H - kh + 1, W - kw + 1, kh - 1, kw - 1)}.items():
            p.set_option("output_region", region)
            assert p.get_option("output_region") == region and (p.info.out_h, p.info.out_w) == (oh, ow) and p.info.out_map_bytes == oh * ow * 4
            for g, r in zip(p.convolve(ks), ref2):
                assert g.shape == (oh, ow) and util.rel_err(g, r[o_h:o_h + oh, o_w:o_w + ow]) < TIGHT * max(1.0, np.abs(r).max() / np.abs(r[o_h:o_h + oh, o_w:o_w + ow]).max())
            od = torch.full((n, ow, oh), float("nan"), dtype=torch.float32, device=dev)
            p.convolve_packed_device(n, k_d.data_ptr(), kh, kw, od.data_ptr())
            p.synchronize()
            for j, r in enumerate(ref2):
                assert util.rel_err(od[j].cpu().numpy().T, r[o_h:o_h + oh, o_w:o_w + ow]) < TIGHT * max(1.0, np.abs(r).max() / np.abs(r[o_h:o_h + oh, o_w:o_w + ow]).max())
        p.set_option("output_region", 0)
        assert (p.info.out_h, p.info.out_w) == (fh, fw)
        with pytest.raises(fc.FFTConvError) as ei:          # a kernel beyond MAX_KERNEL cannot be folded block-wise
            p.convolve([np.zeros((kh + 1, kw, F), np.float32)])
        assert ei.value.status == -4
    # several plans from one process: each is block-wise, the spectrum copy moves every block's spectrum
    with fc.MultiPlan(H, W, F, kh, kw, [0, 0], options=small) as mp:
        mp.set_image(data)
        for g, r in zip(mp.convolve(ks), ref):
            assert util.rel_err(g, r) < TIGHT


# ---- block-wise, overlap-save: blocks whose transform has specialised kernels; the output kernel stores each block's
# rectangle of the maps (no block maps, no summing pass).  max_transform = 288 / 576 makes small problems take it; large
# single-pass sizes take it by themselves where the planner's cost model says blocks are faster. -------------------------

@pytest.mark.parametrize("case", [
    ((700, 500, 2, 9, 13, 5), 288, 6),       # 3 x 2 blocks of 288 x 288, ragged edge blocks, ragged kernels, F = 2
    ((1000, 300, 1, 31, 17, 3), 576, 2),     # blocks along h only (one 576-point block covers w)
    ((300, 1500, 3, 16, 16, 4), 576, 3),     # ... along w only
    ((560, 1130, 1, 1, 1, 2), 576, 2),       # 1 x 1 kernels: blocks without history
])
def test_overlap_save_blocks_every_entry(fc, oracle, case):
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    shape, mt, nblk = case
    H, W, F, kh, kw, n = shape
    rng = np.random.default_rng(sum(shape) + 7)
    opts = {"max_transform": mt}
    data = rng.standard_normal((H, W, F)).astype(np.float32)
    data2 = rng.standard_normal((H, W, F)).astype(np.float32)
    ks = [rng.standard_normal((kh, kw, F)).astype(np.float32) for _ in range(n)]
    if n > 2:
        ks[1] = rng.standard_normal((max(1, kh - 2), max(1, kw - 1), F)).astype(np.float32)      # ragged cell: three groups of kernels
    ref = oracle.conv_fft(data, kh, kw, ks)
    fh, fw = util.ceil16(H + kh - 1), util.ceil16(W + kw - 1)
    for g, r in zip(fc.cudaConvolutionFFT(data, kh, kw, ks, options=opts), ref):      # one-shot entry
        assert g.shape == r.shape and util.rel_err(g, r) < TIGHT
    with fc.Plan(H, W, F, kh, kw, options=opts) as p:
        assert p.get_option("blockwise") >= nblk and p.get_option("overlap_save") == 1
        assert (p.info.fft_h, p.info.fft_w) == (fh, fw) and p.info.map_bytes == fh * fw * 4
        p.set_image(data)
        for rep in range(2):
            for g, r in zip(p.convolve(ks), ref):
                assert g.shape == (fh, fw) and util.rel_err(g, r) < TIGHT
        # device-resident image, packed device-resident kernels and maps (what bench.py uses); NaN-filled maps: every element is written
        same = [rng.standard_normal((kh, kw, F)).astype(np.float32) for _ in range(n)]
        ref2 = oracle.conv_fft(data2, kh, kw, same)
        img_d = torch.from_numpy(np.ascontiguousarray(np.transpose(data2, (2, 1, 0)))).to(dev)
        k_d = torch.from_numpy(np.ascontiguousarray(np.stack([np.transpose(k, (2, 1, 0)) for k in same]))).to(dev)
        out = torch.full((n, fw, fh), float("nan"), dtype=torch.float32, device=dev)
        p.set_image_device(img_d.data_ptr())
        p.convolve_packed_device(n, k_d.data_ptr(), kh, kw, out.data_ptr())
        p.synchronize()
        for j, r in enumerate(ref2):
            assert util.rel_err(out[j].cpu().numpy().T, r) < TIGHT
        p.set_option("flip_kernels", 1)                                   # correlation through the blocks
        p.convolve_packed_device(n, k_d.data_ptr(), kh, kw, out.data_ptr())
        p.synchronize()
        p.set_option("flip_kernels", 0)
        for j, r in enumerate(oracle.conv_fft(data2, kh, kw, [k[::-1, ::-1, :].copy() for k in same])):
            assert util.rel_err(out[j].cpu().numpy().T, r) < TIGHT
        with pytest.raises(fc.FFTConvError) as ei:          # a kernel beyond MAX_KERNEL would wrap into the stored part: rejected
            p.convolve([np.zeros((kh + 1, kw, F), np.float32)])
        assert ei.value.status == -4
    with fc.Plan(H, W, F, kh, kw, options=dict(opts, kernel_path=1)) as p:     # generic kernels only: blocks are summed (overlap-add)
        assert p.get_option("blockwise") > 1 and p.get_option("overlap_save") == 0
        p.set_image(data)
        for g, r in zip(p.convolve(ks), ref):
            assert util.rel_err(g, r) < TIGHT
    with fc.MultiPlan(H, W, F, kh, kw, [0, 0], options=opts) as mp:      # the spectrum copied between plans is every block's
        mp.set_image(data)
        for g, r in zip(mp.convolve(ks), ref):
            assert util.rel_err(g, r) < TIGHT


def test_large_sizes_run_in_blocks_where_the_cost_model_says_so(fc, oracle):
    """4900 x 4900 with 63 x 63 kernels: one pass would transform 5120 x 5120 (4-column output kernel); the default plan runs
    2 x 2 blocks of 2560 x 2560 instead.  Same maps as the one-pass plan (blockwise = 1) and as the oracle."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    H = W = 4900
    K, n = 63, 2
    rng = np.random.default_rng(4900)
    data = rng.random((H, W, 1), dtype=np.float32)
    ks = [rng.random((K, K, 1), dtype=np.float32) for _ in range(n)]
    img_d = torch.from_numpy(np.ascontiguousarray(np.transpose(data, (2, 1, 0)))).to(dev)
    k_d = torch.from_numpy(np.ascontiguousarray(np.stack([np.transpose(k, (2, 1, 0)) for k in ks]))).to(dev)
    outs = []
    for opts in (None, {"blockwise": 1}):
        with fc.Plan(H, W, 1, K, K, options=opts) as p:
            i = p.info
            assert (i.fft_h, i.fft_w) == (4976, 4976)
            assert (p.get_option("blockwise") > 1) == (opts is None) and p.get_option("overlap_save") == (1 if opts is None else 0)
            assert (i.transform_h <= 4608) == (opts is None)
            out = torch.full((n, i.fft_w, i.fft_h), float("nan"), dtype=torch.float32, device=dev)
            p.set_image_device(img_d.data_ptr())
            p.convolve_packed_device(n, k_d.data_ptr(), K, K, out.data_ptr())
            p.synchronize()
            outs.append(out)
    assert bool(torch.isfinite(outs[0]).all())
    assert float((outs[0] - outs[1]).abs().max() / outs[1].abs().max()) < 2e-6
    ref = oracle.conv_fft(data, K, K, ks[:1])[0]
    assert util.rel_err(outs[0][0].cpu().numpy().T, ref) < TIGHT
    # the options of a one-pass plan hold on the plan the planner turned block-wise by itself (overlap-save): "output_region"
    # full (demoCudaConvolutionFFT.m:149) and same, packed device maps and host maps, and the window again afterwards
    with fc.Plan(H, W, 1, K, K) as p:
        assert p.get_option("blockwise") > 1 and p.get_option("overlap_save") == 1
        p.set_image_device(img_d.data_ptr())
        for region, (oh, ow, o_h, o_w) in {1: (H + K - 1, W + K - 1, 0, 0), 2: (H, W, (K - 1) // 2, (K - 1) // 2)}.items():
            p.set_option("output_region", region)
            assert (p.info.out_h, p.info.out_w) == (oh, ow) and p.info.out_map_bytes == oh * ow * 4
            od = torch.full((n, ow, oh), float("nan"), dtype=torch.float32, device=dev)
            p.convolve_packed_device(n, k_d.data_ptr(), K, K, od.data_ptr())
            p.synchronize()
            assert torch.equal(od, outs[0][:, o_w:o_w + ow, o_h:o_h + oh])
        got = p.convolve(ks[:1])[0]                      # region 2 ("same"), host map
        assert got.shape == (H, W) and np.array_equal(got, outs[0][0].cpu().numpy().T[o_h:o_h + H, o_w:o_w + W])
        p.set_option("output_region", 0)
        assert (p.info.out_h, p.info.out_w) == (4976, 4976)


# ---- output_region: full / same / valid parts of the padded window ---------------------------------

@pytest.mark.parametrize("shape", [(64, 8, 5, 10, 4, 3), (300, 260, 1, 31, 17, 5), (1024, 1024, 1, 63, 63, 2)])
def test_output_regions_match_slices_of_the_window(fc, oracle, shape):
    """the demo crops the linear convolution out of the window by hand
    (demoCudaConvolutionFFT.m:149: cvg(1:n+cn-1, 1:m+cm-1)); "output_region" returns that ("full"),
    the data-sized centred part ("same") or the part free of zero padding ("valid") directly"""
    torch = pytest.importorskip("torch")
    H, W, F, kh, kw, n = shape
    rng = np.random.default_rng(sum(shape) + 1)
    data = rng.standard_normal((H, W, F)).astype(np.float32)
    ks = [rng.standard_normal((kh, kw, F)).astype(np.float32) for _ in range(n)]
    ref = oracle.conv_fft(data, kh, kw, ks)
    regions = {1: (H + kh - 1, W + kw - 1, 0, 0), 2: (H, W, (kh - 1) // 2, (kw - 1) // 2), 3: (H - kh + 1, W - kw + 1, kh - 1, kw - 1)}
    with fc.Plan(H, W, F, kh, kw) as plan:
        plan.set_image(data)
        plan.set_option("host_min_kb", 0)       # small maps through the copy threads as well
        for region, (oh, ow, fh, fw) in regions.items():
            plan.set_option("output_region", region)
            assert (plan.info.out_h, plan.info.out_w) == (oh, ow) and plan.info.out_map_bytes == oh * ow * 4
            for mode in (1, 0):                      # streamed and blocking copy-out
                plan.set_option("host_stream", mode)
                got = plan.convolve(ks)
                for g, r in zip(got, ref):
                    assert g.shape == (oh, ow)
                    assert util.rel_err(g, r[fh:fh + oh, fw:fw + ow]) < TIGHT * max(1.0, np.abs(r).max() / np.abs(r[fh:fh + oh, fw:fw + ow]).max())
            # packed device path: [n][out_w][out_h]
            kd = torch.from_numpy(np.ascontiguousarray(np.stack([np.transpose(k, (2, 1, 0)) for k in ks]))).cuda()   # [n][F][kw][kh]
            od = torch.full((n, ow, oh), float("nan"), dtype=torch.float32, device="cuda")
            plan.convolve_packed_device(n, kd.data_ptr(), kh, kw, od.data_ptr())
            plan.synchronize()
            for j, r in enumerate(ref):
                assert np.array_equal(od[j].cpu().numpy().T, got[j])
        plan.set_option("output_region", 0)
        assert (plan.info.out_h, plan.info.out_w) == (plan.info.fft_h, plan.info.fft_w)
        back = plan.convolve(ks[:1])[0]
        assert util.rel_err(back, ref[0]) < TIGHT


@pytest.mark.parametrize("shape", [(64, 8, 5, 10, 4, 3), (300, 260, 1, 31, 17, 2), (1024, 1024, 1, 63, 63, 1)])
def test_output_region_pow2_window(fc, oracle, shape):
    """"output_region" 4: the next-power-of-two window of the reference's computeFFTsize
    (src/cudaConvFFTData.h:67-94: align to 16, then to a power of two) -- the convolution in the
    top-left corner, exact zeros in the rest; what a float64 fft2/ifft2 at that size gives too"""
    H, W, F, kh, kw, n = shape
    rng = np.random.default_rng(sum(shape) + 4)
    data = rng.standard_normal((H, W, F)).astype(np.float32)
    ks = [rng.standard_normal((kh, kw, F)).astype(np.float32) for _ in range(n)]
    ph, pw = fc.fft_size_pow2(H + kh - 1), fc.fft_size_pow2(W + kw - 1)
    assert ph & (ph - 1) == 0 and pw & (pw - 1) == 0 and ph >= util.ceil16(H + kh - 1) and ph < 2 * util.ceil16(H + kh - 1)
    D = np.fft.fft2(data.astype(np.float64), s=(ph, pw), axes=(0, 1))
    want = [np.real(np.fft.ifft2(D * np.fft.fft2(k.astype(np.float64), s=(ph, pw), axes=(0, 1)), axes=(0, 1))).sum(axis=2) for k in ks]
    with fc.Plan(H, W, F, kh, kw) as plan:
        plan.set_image(data)
        plan.set_option("output_region", 4)
        assert (plan.info.out_h, plan.info.out_w) == (ph, pw)
        got = plan.convolve(ks)
        for g, r in zip(got, want):
            assert g.shape == (ph, pw)
            assert util.rel_err(g, r) < TIGHT
            assert not g[util.ceil16(H + kh - 1):, :].any() and not g[:, util.ceil16(W + kw - 1):].any()   # beyond the ceil16 window: exact zeros


def test_pow2_size_function(fc):
    # computeFFTsize: iAlignUp(n, 16), already a power of two -> itself, else the next one
    assert [fc.fft_size_pow2(n) for n in (1, 16, 17, 33, 64, 65, 286, 1086, 4222, 4096)] == [16, 16, 32, 64, 64, 128, 512, 2048, 8192, 4096]


def test_output_region_rejects_empty_and_unknown(fc):
    with fc.Plan(20, 20, 1, 31, 5) as plan:
        with pytest.raises(fc.FFTConvError):
            plan.set_option("output_region", 3)      # valid region empty: kernel taller than the data
        with pytest.raises(fc.FFTConvError):
            plan.set_option("output_region", 7)
        plan.set_option("output_region", 1)
        assert (plan.info.out_h, plan.info.out_w) == (50, 24)


# ---- the image spectrum in the reference's own order (cudaFFTData's gpuArray) -----------------------

@pytest.mark.parametrize("shape", [
    (64, 8, 5, 10, 4),          # the demo problem: generic kernels
    (256, 256, 1, 31, 31),      # cfg1: 288 x 288, both specialised kernels (register-order spectrum rows)
    (1024, 1024, 1, 63, 63),    # cfg2's 1088 window: exact_window keeps the transform off 1152 -- on the 1088-point kernels (round 4)
    (4096, 1024, 1, 63, 63),    # cfg4's 4160 window along h (M = 2080 kernels), 1088 along w
    (600, 4096, 2, 40, 63),     # 4160 along w (row kernel), F = 2
    (500, 4096, 2, 40, 127),    # 4224 along w only, F = 2
    (300, 200, 3, 21, 9),
])
def test_spectrum_export_import_in_reference_order(fc, oracle, shape):
    """fftconv_plan_export_spectrum == numpy.fft.rfft2 of the zero-padded [F][FFT_W][FFT_H] planes, the
    layout of the complex gpuArray cudaFFTData returns (src/cudaFFTData.cu:90-103; cuFFT geometry
    src/cudaConvolutionFFT.cu:122-142); importing it into a fresh plan convolves like set_image."""
    H, W, F, kh, kw = shape
    rng = np.random.default_rng(sum(shape))
    data = rng.random((H, W, F), dtype=np.float32)
    ks = [rng.random((kh, kw, F), dtype=np.float32) for _ in range(2)]
    fh, fw = util.ceil16(H + kh - 1), util.ceil16(W + kw - 1)
    padded = np.zeros((F, fw, fh), dtype=np.float64)
    padded[:, :W, :H] = np.transpose(data, (2, 1, 0))
    want = np.fft.rfft2(padded, axes=(1, 2))
    with fc.Plan(H, W, F, kh, kw, options={"exact_window": 1}) as p:
        assert p.info.exact_window == 1 and (p.info.transform_h, p.info.transform_w) == (fh, fw)
        if (fh, fw) in ((1088, 1088), (4160, 1088)):       # the BASELINE windows of cfg2 / cfg4: no generic kernel runs
            assert p.get_option("specialised_kernels") == 3
        if fw == 4160:
            assert p.get_option("specialised_kernels") & 1
        p.set_image(data)
        spec = p.export_spectrum()
        assert spec.shape == (F, fw, fh // 2 + 1)
        assert np.abs(spec - want).max() / np.abs(want).max() < 1e-5
        ref = oracle.conv_fft(data, kh, kw, ks)
        for g, r in zip(p.convolve(ks), ref):
            assert util.rel_err(g, r) < TIGHT
    with fc.Plan(H, W, F, kh, kw, options={"exact_window": 1}) as q:
        with pytest.raises(fc.FFTConvError) as ei:
            q.export_spectrum()                       # nothing to export yet
        assert ei.value.status == -9
        q.import_spectrum(want.astype(np.complex64))  # numpy's spectrum in, never saw the image
        for g, r in zip(q.convolve(ks), ref):
            assert util.rel_err(g, r) < TIGHT


def test_spectrum_export_needs_the_window_transform(fc):
    with fc.Plan(1024, 1024, 1, 63, 63) as p:          # default plan: 1152 x 1152 transform, 1088 window
        assert p.info.exact_window == 0
        p.set_image(np.zeros((1024, 1024, 1), np.float32))
        with pytest.raises(fc.FFTConvError) as ei:
            p.export_spectrum()
        assert ei.value.status == -5


def test_verbose_option_prints_the_reference_debug_lines(fc):
    """plan option "verbose": the reference's compile-time `debug` prints (src/cudaConvolutionFFT.cu:9,60,68,87,100,114,
    240,258) at run time, to stderr; silent by default"""
    import subprocess
    import sys
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r); import util; fc = util.load_package()\n"
        "d = np.random.default_rng(0).random((64, 8, 5), dtype=np.float32)\n"
        "ks = [np.random.default_rng(1).random((10, 4, 5), dtype=np.float32) for _ in range(3)]\n"
        "with fc.Plan(64, 8, 5, 10, 4) as p:\n"
        "    p.set_image(d); p.convolve(ks)\n"
        "    sys.stderr.write('--- verbose on\\n'); sys.stderr.flush()\n"
        "    p.set_option('verbose', 1); assert p.get_option('verbose') == 1\n"
        "    p.set_image(d); p.convolve(ks)\n"
    ) % (util.ROOT + "/tests",)
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    quiet, loud = r.stderr.split("--- verbose on\n")
    assert "fftconv:" not in quiet
    for want in ("Using GPU : 0", "Data size: h=64, w=8, f=5", "FFT size: h=80, w=16", "Kernel size: h=10, w=4", "N Kernel: 3", "FFT done"):
        assert want in loud, (want, loud)


def test_host_output_copy_threads_start_and_stop_300_times(fc, oracle):
    """The incident of round 2 (a one-shot call on 92-KB maps died in hipStreamCreateWithFlags, called by freshly
    started copy threads, about once in 50 starts on some boxes): 300 plans, each starting 1-4 copy threads for
    small host maps (host_min_kb = 0 forces the threaded path the default spares maps under 1 MiB), every map checked"""
    H, W, F, kh, kw, n = 140, 150, 1, 13, 11, 6          # 152 x 160 maps = 97 KB
    data, ks = util.synth(91, H, W, F, kh, kw, n)
    ref = oracle.conv_fft(data, kh, kw, ks)
    for i in range(300):
        with fc.Plan(H, W, F, kh, kw) as p:
            p.set_option("host_min_kb", 0)
            p.set_option("host_stream", 1 if i % 5 else 2)
            p.set_option("host_threads", 1 + i % 4)
            p.set_option("batch_maps", 2)
            p.set_image(data)
            got = p.convolve(ks)
        for g, r in zip(got, ref):
            assert util.rel_err(g, r) < TIGHT, i


def test_full_size_five_feature_maps_vs_oracle(fc, oracle):
    """The reference's own demo runs F = 5 feature planes (demoCudaConvolutionFFT.m:37-42; sumAlongFeatures,
    src/cudaConvFFTData.cuh:70-92).  At cfg3's geometry -- 4096 x 4096 x 5 image, 127 x 127 x 5 kernels, the walk over
    (map, feature) pairs of the multi-map row kernel -- three maps in full against the oracle, plus the all-ones
    identity: a map's sum is the sum over features of sum(image_f) * sum(kernel_f)."""
    H = W = 4096
    kh = kw = 127
    F, n = 5, 3
    img, ks = util.synth(55, H, W, F, kh, kw, n)
    ks[1] = np.asfortranarray(ks[1][:101, :90, :])          # ragged cell: a group of its own
    got = fc.cudaConvolutionFFT(img, kh, kw, ks)
    ref = oracle.conv_fft(img, kh, kw, ks)
    for j, (g, r) in enumerate(zip(got, ref)):
        assert g.shape == (4224, 4224)
        assert util.rel_err(g, r) < TOL, j
        want = sum(float(img[:, :, f].astype(np.float64).sum()) * float(ks[j][:, :, f].astype(np.float64).sum()) for f in range(F))
        assert abs(float(g.astype(np.float64).sum()) - want) / abs(want) < 1e-5


@pytest.mark.parametrize("defer", [1, 0])
def test_deferred_kernel_preparation_in_every_order(fc, oracle, defer):
    """With plan option defer_prepare fftconv_plan_prepare_kernels_packed only records the request; the kernels' column
    pass runs in ONE launch with the next image's column pass (set_image on the same stream) or, if none comes, at the
    convolve.  Without it (the default) the pass is queued at once -- ahead of whatever the caller waits for next, the
    overlap the call exists for.  Every order of the calls must give the maps of a plan that never prepared anything."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    for (H, W, F, kh, kw, n) in [(256, 256, 1, 31, 31, 3), (300, 260, 2, 15, 13, 5), (1024, 1024, 1, 63, 63, 4)]:
        img, ks = util.synth(101, H, W, F, kh, kw, n)
        img2, ks2 = util.synth(102, H, W, F, kh, kw, n)
        ref, ref2, ref12 = oracle.conv_fft(img, kh, kw, ks), oracle.conv_fft(img2, kh, kw, ks), oracle.conv_fft(img2, kh, kw, ks2)
        pack = lambda kk: torch.from_numpy(np.ascontiguousarray(np.stack([np.transpose(k, (2, 1, 0)) for k in kk]))).to(dev)
        img_d = torch.from_numpy(np.ascontiguousarray(np.transpose(img, (2, 1, 0)))).to(dev)
        img2_d = torch.from_numpy(np.ascontiguousarray(np.transpose(img2, (2, 1, 0)))).to(dev)
        k_d, k2_d = pack(ks), pack(ks2)
        side = torch.cuda.Stream(dev)
        with fc.Plan(H, W, F, kh, kw) as p:
            out = torch.empty((n, p.info.fft_w, p.info.fft_h), dtype=torch.float32, device=dev)

            def check(want):
                p.synchronize()
                for j, r in enumerate(want):
                    assert util.rel_err(out[j].cpu().numpy().T, r) < TIGHT, (H, W, j)

            assert p.get_option("defer_prepare") == 0          # the default: launched at once
            p.set_option("defer_prepare", defer)
            # prepare -> set_image (merged launch) -> convolve
            p.prepare_kernels_packed_device(n, k_d.data_ptr(), kh, kw)
            # deferred: recorded, nothing queued yet; default: already on the stream (nothing pending)
            if not defer:
                assert p.get_option("prepare_pending") == 0
            elif (H, W) in ((256, 256), (1024, 1024)):       # lengths with a specialised column pass: the request waits
                assert p.get_option("prepare_pending") == 1
            p.set_image_device(img_d.data_ptr())
            assert p.get_option("prepare_pending") == 0
            p.convolve_packed_device(n, k_d.data_ptr(), kh, kw, out.data_ptr())
            check(ref)
            # prepare -> convolve (no image in between: flushed by the convolve), image spectrum reused
            out.zero_()
            p.prepare_kernels_packed_device(n, k_d.data_ptr(), kh, kw)
            p.convolve_packed_device(n, k_d.data_ptr(), kh, kw, out.data_ptr())
            check(ref)
            # prepared for one set of kernels, convolved with another: the preparation is not used
            out.zero_()
            p.prepare_kernels_packed_device(n, k_d.data_ptr(), kh, kw)
            p.set_image_device(img2_d.data_ptr())
            p.convolve_packed_device(n, k2_d.data_ptr(), kh, kw, out.data_ptr())
            check(ref12)
            # two preparations in a row, then a stream change before the image (the request is flushed on its own stream)
            out.zero_()
            p.prepare_kernels_packed_device(n, k2_d.data_ptr(), kh, kw)
            p.prepare_kernels_packed_device(n, k_d.data_ptr(), kh, kw)
            torch.cuda.synchronize(dev)
            p.set_stream(side.cuda_stream)
            p.set_image_device(img2_d.data_ptr())
            p.synchronize()
            p.set_stream(0)
            p.convolve_packed_device(n, k_d.data_ptr(), kh, kw, out.data_ptr())
            check(ref2)
            # with per-kind profiling the two passes are separate launches and both are counted
            p.set_option("profile", 1)
            p.profile(reset=True)
            p.prepare_kernels_packed_device(n, k_d.data_ptr(), kh, kw)
            p.set_image_device(img_d.data_ptr())
            p.convolve_packed_device(n, k_d.data_ptr(), kh, kw, out.data_ptr())
            check(ref)
            prof = p.profile(reset=True)
            assert prof["kernel_cols"]["launches"] == 1 and prof["image_cols"]["launches"] == 1
            p.set_option("profile", 0)


def test_one_shot_plan_cache(fc, oracle):
    """fftconv_convolution_fft keeps the plans of its last few problems (the reference rebuilds its cuFFT plans and
    buffers in every call, src/cudaConvolutionFFT.cu:127-185,302-310): a second call with ANOTHER image and OTHER
    kernels of the same problem size takes the cached plan and still matches the oracle; an argument error leaves the
    cache usable; the limits and fftconv_cache_clear release plans."""
    fc.cache_clear()
    fc.cache_configure(4)
    s0 = fc.cache_stats()
    H, W, F, kh, kw, n = 200, 176, 2, 15, 9, 3
    img, ks = util.synth(501, H, W, F, kh, kw, n)
    got = fc.cudaConvolutionFFT(img, kh, kw, ks)
    t1 = fc.last_call_timing()
    assert t1["cache_hit"] == 0 and t1["total_ms"] > 0
    for g, r in zip(got, oracle.conv_fft(img, kh, kw, ks)):
        assert util.rel_err(g, r) < TOL
    s1 = fc.cache_stats()
    assert s1["plans"] == s0["plans"] + 1 and s1["misses"] == s0["misses"] + 1 and s1["device_bytes"] > 0
    # same problem size, different image, different kernels (also smaller ones, ragged): cached plan
    img2, ks2 = util.synth(502, H, W, F, kh, kw, n + 2)
    ks2[1] = ks2[1][:7, :5, :].copy()
    got2 = fc.cudaConvolutionFFT(img2, kh, kw, ks2)
    t2 = fc.last_call_timing()
    assert t2["cache_hit"] == 1
    for g, r in zip(got2, oracle.conv_fft(img2, kh, kw, ks2)):
        assert util.rel_err(g, r) < TOL
    s2 = fc.cache_stats()
    assert s2["plans"] == s1["plans"] and s2["hits"] == s1["hits"] + 1
    # an argument error on the cached plan (feature mismatch; a kernel larger than the window): the cache stays usable
    bad = [k[:, :, :1].copy() for k in ks]
    with pytest.raises(fc.FFTConvError) as ei:
        fc.cudaConvolutionFFT(img, kh, kw, bad)
    assert ei.value.status == -3
    with pytest.raises(fc.FFTConvError):
        fc.cudaConvolutionFFT(img, kh, kw, [np.zeros((H + 100, 3, F), dtype=np.float32)])
    assert fc.cache_stats()["plans"] == s1["plans"]
    got3 = fc.cudaConvolutionFFT(img, kh, kw, ks)
    assert fc.last_call_timing()["cache_hit"] == 1
    for g, r in zip(got3, oracle.conv_fft(img, kh, kw, ks)):
        assert util.rel_err(g, r) < TOL
    # different options are different plans; generic kernels give the same maps
    got4 = fc.cudaConvolutionFFT(img, kh, kw, ks, options={"kernel_path": 1})
    assert fc.last_call_timing()["cache_hit"] == 0
    for g, r in zip(got4, got3):
        assert util.rel_err(g, r) < TIGHT
    # the limit evicts the least recently used plan
    fc.cache_configure(2)
    assert fc.cache_stats()["plans"] <= 2
    for hh in (64, 80, 96):
        i3, k3 = util.synth(600 + hh, hh, 48, 1, 5, 5, 1)
        fc.cudaConvolutionFFT(i3, 5, 5, k3)
    assert fc.cache_stats()["plans"] == 2
    i3, k3 = util.synth(696, 96, 48, 1, 5, 5, 1)
    fc.cudaConvolutionFFT(i3, 5, 5, k3)
    assert fc.last_call_timing()["cache_hit"] == 1          # the most recent one is still there
    i3, k3 = util.synth(664, 64, 48, 1, 5, 5, 1)
    fc.cudaConvolutionFFT(i3, 5, 5, k3)
    assert fc.last_call_timing()["cache_hit"] == 0          # the oldest was pushed out
    # off: every call builds and tears down its plan, as the reference does
    fc.cache_configure(0)
    assert fc.cache_stats()["plans"] == 0
    got5 = fc.cudaConvolutionFFT(img, kh, kw, ks)
    assert fc.last_call_timing()["cache_hit"] == 0 and fc.cache_stats()["plans"] == 0
    for g, r in zip(got5, got3):
        assert np.array_equal(g, r)
    fc.cache_configure(4)
    fc.cudaConvolutionFFT(img, kh, kw, ks)
    assert fc.cache_stats()["plans"] == 1
    fc.cache_clear()
    assert fc.cache_stats()["plans"] == 0 and fc.cache_stats()["device_bytes"] == 0


def test_cached_plans_go_when_the_device_runs_out_of_memory(fc, oracle):
    """The one-shot entry keeps idle plans with gigabytes of device scratch, where the reference releases everything between
    calls (src/cudaConvolutionFFT.cu:302-310).  That must never cost a later call its memory: with the device filled up to
    less than the next call needs, the call still succeeds -- the idle plan is released when an allocation fails and the
    request repeated -- and matches the oracle."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    fc.cache_clear()
    fc.cache_configure(4)
    H, W, kh, kw, n = 2048, 2048, 31, 31, 32
    img, ks = util.synth(811, H, W, 1, kh, kw, n)
    fh, fw = util.ceil16(H + kh - 1), util.ceil16(W + kw - 1)
    outs = [np.empty((fh, fw), dtype=np.float32, order="F") for _ in range(n)]
    fc.cudaConvolutionFFT(img, kh, kw, ks, out=outs)
    held = fc.cache_stats()
    assert held["plans"] == 1 and held["device_bytes"] > (1 << 30)
    D = held["device_bytes"]
    torch.cuda.synchronize(dev)
    torch.cuda.empty_cache()
    free, _total = torch.cuda.mem_get_info(dev)
    hog = torch.empty(free - int(0.35 * D), dtype=torch.uint8, device=dev)       # what is left is a third of what the next call needs
    try:
        H2 = 2040                                                                 # another problem size: another plan, about as large
        img2, _ = util.synth(812, H2, H2, 1, kh, kw, 1)
        outs2 = [np.empty((util.ceil16(H2 + kh - 1), util.ceil16(H2 + kw - 1)), dtype=np.float32, order="F") for _ in range(n)]
        fc.cudaConvolutionFFT(img2, kh, kw, ks, out=outs2)
        after = fc.cache_stats()
        assert after["plans"] == 1 and after["misses"] == held["misses"] + 1      # the first plan went, this call's plan is the idle one now
        ref = oracle.conv_fft(img2, kh, kw, ks[:2])
        for g, r in zip(outs2[:2], ref):
            assert util.rel_err(g, r) < TOL
    finally:
        del hog
        torch.cuda.empty_cache()
        fc.cache_clear()


def test_one_shot_plan_cache_blockwise_and_threads(fc, oracle):
    """Block-wise plans are cached like any other; two threads calling the one-shot entry with the same problem at the
    same time never share a plan (the second builds its own)."""
    import threading
    fc.cache_clear()
    fc.cache_configure(4)
    H, W, F, kh, kw, n = 150, 130, 1, 9, 9, 2
    img, ks = util.synth(701, H, W, F, kh, kw, n)
    ref = oracle.conv_fft(img, kh, kw, ks)
    for rep in range(2):
        got = fc.cudaConvolutionFFT(img, kh, kw, ks, options={"max_transform": 64})
        assert fc.last_call_timing()["cache_hit"] == rep
        for g, r in zip(got, ref):
            assert util.rel_err(g, r) < TOL
    res, errs = {}, []

    def work(i):
        try:
            im, kk = util.synth(800 + i, H, W, F, kh, kw, n)
            for _ in range(3):
                res[i] = (im, kk, fc.cudaConvolutionFFT(im, kh, kw, kk))
        except Exception as e:   # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for i, (im, kk, got) in res.items():
        for g, r in zip(got, oracle.conv_fft(im, kh, kw, kk)):
            assert util.rel_err(g, r) < TOL
    assert 1 <= fc.cache_stats()["plans"] <= 4
    fc.cache_clear()


def test_fuzz_slice(fc, oracle):
    """a fixed slice of tools/fuzz_gpu.py: random shapes (tiny, strips, sizes sitting just under the specialised lengths),
    ragged cells, every entry and a random set of plan options per case, each map against the oracle"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_gpu", os.path.join(util.ROOT, "tools", "fuzz_gpu.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    done, failures, worst = fz.run(seed=7, cases=40, quiet=True)
    assert done == 40 and not failures and worst < TIGHT


@pytest.mark.parametrize("shape", [(398, 322, 3, 75, 92, 5), (850, 600, 1, 29, 86, 3), (954, 1171, 2, 113, 82, 2)])
def test_blockwise_with_kernels_too_wide_for_the_specialised_block_length(fc, oracle, shape):
    """max_transform = 288 and kernels wider than the 288-point row kernel takes (72 columns): the blocks are zero-padded
    (overlap-add) and their transform must stay within the cap -- the planner's preference for a longer length with specialised
    kernels (384) once made the block plan's creation fail (found by tools/fuzz_gpu.py, seed 5)"""
    H, W, F, kh, kw, n = shape
    rng = np.random.default_rng(sum(shape))
    data = rng.standard_normal((H, W, F)).astype(np.float32)
    ks = [rng.standard_normal((kh, kw, F)).astype(np.float32) for _ in range(n)]
    ks[-1] = rng.standard_normal((max(1, kh - 3), max(1, kw - 5), F)).astype(np.float32)
    with fc.Plan(H, W, F, kh, kw, options={"max_transform": 288}) as plan:
        assert plan.get_option("blockwise") > 1 and max(plan.info.transform_h, plan.info.transform_w) <= 288
        plan.set_image(data)
        got = plan.convolve(ks)
    for g, r in zip(got, oracle.conv_fft(data, kh, kw, ks)):
        assert util.rel_err(g, r) < TIGHT
