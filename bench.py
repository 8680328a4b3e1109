#!/usr/bin/env python3
"""bench.py -- headline benchmark of the FFT-convolution hot path on MI355X.

A "step" is one pass of the hot path over one batch of synthetic input: the image (already
resident in HBM) is zero-padded and transformed once, then every filter of the batch is
convolved with it, all maps staying device-resident (SURVEY.md 8(d) "timed region").

N = 1 (default): BASELINE.json configs[2], the configuration the metric is quoted on --
4096x4096 fp32 image, 256 kernels of 127x127, F = 1 -> 256 maps of 4224x4224.
N > 1 (default): BASELINE.json configs[3] -- 4096x4096 image, 1024 kernels of 63x63 in total,
sharded over the ranks in contiguous blocks (one process per GPU); rank 0 transforms the image and
its spectrum is broadcast once per step over RCCL (torch.distributed "nccl"), the only
collective on the path.  Total work is fixed, so "scaling" is "strong"; value is the whole-job
rate (all maps / max time over ranks).  `--gpus N` run by hand starts the N ranks itself; under
torch.distributed.run the ranks come from the environment.
`--images n` is BASELINE configs[4]: n images (2048x2048 with --config cfg5) dealt over the ranks,
every image against all kernels, H2D of the next image behind the compute of the current one.

`--share-gpu` is the rehearsal of the N > 1 path on a one-GPU box: every rank on cuda:0 (LOCAL_RANK
ignored), the collective over gloo (RCCL refuses two ranks on one device) -- rank start-up and teardown,
per-rank placement tuning, the broadcast step and the timing / check reductions all run as at N > 1.

The steps themselves live in cuda-fft-convolution_amd/multi_gpu.py (shared with the tests).

Untimed set-up and warm-up: the headline runs on DEFAULT plan options (round 5).  The library's default for plan option tune_placement is
automatic: a plan whose launches write >= 2 GiB of maps on a mostly free device times five candidate allocations of its intermediate at its
first convolve (~100 ms, inside the warm-up); the other policy runs beside it as `value_untuned_placement` / `value_tuned_placement`; the W warm-up
steps are followed by more untimed steps where they are shorter than the ~40 ms the GPU's clocks
need to settle after an idle gap (config.clock_warm_steps).  The timed region is exactly K steps.

Order of a run (round 4): set-up -> `cpu_baseline` (CPU only, rank 0 at N = 1) -> W warm-up steps + >= 1 s of settled
load (`--settle-s`) -> the K timed steps -> checks -> untimed extra passes (`host_output`: the MEX-faithful host-in /
host-out entry, one-shot and reused plan; `multi_feature`: F > 1, the reference's sumAlongFeatures case; `vendor_fft_baseline`: the
reference's loop ported straight onto the vendor FFT library of the same GPU -- rocFFT through torch.fft -- a baseline like cpu_baseline).  The timed
step contains the upload of the step's kernels from pinned host memory (double-buffered on an upload
stream: multi_gpu.HipPlanEngine); `value_kernels_resident` is the same K steps with the kernels kept in HBM.

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (dominant
kernel, algorithmic bytes / its HIP-event time over the launches of the timed region itself), `cpu_baseline` (the CPU oracle timed on this
host on a bounded sample, rank 0 at N = 1 only) and `check_max_rel_err` (every map's checksum on
the device + the oracle's maps where they were computed); a failed check exits non-zero.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CONFIGS = {
    # name: (H, W, F, kh, kw, filters in total, seed)
    "cfg1": (256, 256, 1, 31, 31, 1, 1),
    "cfg2": (1024, 1024, 1, 63, 63, 16, 2),
    "cfg3": (4096, 4096, 1, 127, 127, 256, 3),
    "cfg4": (4096, 4096, 1, 63, 63, 1024, 4),   # over 8 GPUs: 128 each
    "cfg5": (2048, 2048, 1, 63, 63, 64, 5),     # per image; 32 images over 8 GPUs (--images)
    # not a BASELINE config: the multi-feature form of the reference's demo (F planes summed per map)
    "cfg3f4": (4096, 4096, 4, 127, 127, 64, 6),
    "mid512": (512, 512, 1, 31, 31, 256, 7),   # not BASELINE configs: mid-sized images on the 576 / 768 / 1536 / 3072 transforms
    "hd720": (720, 1280, 1, 31, 31, 128, 8),
    "mid2900": (2900, 2900, 1, 63, 63, 64, 9),
    "big8192": (8192, 8192, 1, 127, 127, 32, 10),   # 8448 x 8448 transforms
    "big6000": (6000, 6000, 1, 63, 63, 32, 11),     # 6144 x 6144 transforms
}
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
CHECK_TOL_ORACLE = 1e-4   # north_star parity bar (max|out - ref| / max|ref| per map)
CHECK_TOL_CHECKSUM = 1e-5  # sum(map) vs sum(image) * sum(kernel), relative
CLOCK_WARM_S = 1.0        # untimed load before the timed region (--settle-s): the clock ramp after idle is ~40 ms, and a sampler
                          # of GPU utilisation outside this process needs about a second of continuous load to see the run


def ceil16(n):
    return (n + 15) // 16 * 16


def alg_bytes(H, W, F, kh, kw):
    """Algorithmic HBM bytes per filter, SURVEY.md 8(d), split by kernel."""
    fh, fw = ceil16(H + kh - 1), ceil16(W + kw - 1)
    P = fh * fw
    C = fw * (fh // 2 + 1)
    spectral = 4 * kh * kw * F + 8 * C * F + 8 * C   # kernel read + image-spectrum read + intermediate write
    out_cols = 8 * C + 4 * P                          # intermediate read + map write
    return {"spectral_rows": spectral, "cols_c2r": out_cols, "total": spectral + out_cols, "P": P}


def host_cores():
    """CPUs this process may really use: the affinity mask, cut by the cgroup CPU quota where one is set."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
            break
        except Exception:
            continue
    return n


def host_mem_available():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) * 1024
    except Exception:
        pass
    return 8 << 30


def cpu_baseline(cfg, sample_filters):
    """Times the CPU oracle (port of demoCudaConvolutionFFT.m:78-102, complex128 full transforms)
    on this host on a bounded sample of the same workload -- one filter per thread on every core this
    process may use (capped by the filter count, by `sample_filters` when > 0 and by free memory: the
    float64 port holds ~5 full complex planes per thread) -- plus figures beside it: the same port on 8
    threads (the figure of earlier rounds), the fp32 half-spectrum C++ restatement
    (oracle/fftconv_cpu_f32.cpp) and SciPy's pocketfft.
    Returns (json object, oracle maps of the first filters for the self-check)."""
    import util
    H, W, F, kh, kw, nf_cfg, seed = CONFIGS[cfg]
    orc = util.Oracle()
    avail = host_cores()
    total = os.cpu_count() or avail
    fh, fw = ceil16(H + kh - 1), ceil16(W + kw - 1)
    P = fh * fw
    mem = host_mem_available()
    per_thread_f64 = 5 * 16 * P * max(1, F)          # complex128 planes a thread of the port holds
    cap_mem = max(1, int(0.5 * mem / per_thread_f64))
    threads = max(1, min(nf_cfg, avail, cap_mem, sample_filters if sample_filters > 0 else 64))   # at most 64: the sample stays ~10-30 s and < 100 GB
    n = threads
    img, ks = util.synth(seed, H, W, F, kh, kw, max(n, min(8, nf_cfg)))
    t0 = time.perf_counter()
    ref = orc.conv_fft(img, kh, kw, ks[:n], threads=threads)
    dt = time.perf_counter() - t0
    ref = ref[:8]       # the self-check compares the first maps only
    res = {"value": n * P / dt / 1e9, "unit": "Gpixel-filters/s", "cores": threads,
           "kind": "port", "value_from": "f64_port_portable_build",
           "sample": "%s image + %d of its filters (float64 fft2/ifft2 oracle, OpenMP over filters: %d threads; the host has %d "
                     "cores, %d usable by this process, memory allows %d threads), %.1f s wall" % (cfg, n, threads, total, avail, cap_mem, dt)}
    # the faithful variants (complex128, full spectra: the literal shape of demoCudaConvolutionFFT.m:78-102), fastest one = `value`:
    faithful = {"f64_port_portable_build": {"value": res["value"], "unit": res["unit"], "cores": threads, "sample": res["sample"],
                                            "build": "-O3 -march=x86-64-v2 (the checker the tests use)"}}
    # (1) the same source built for THIS host (-march=native, made here and now: oracle/_native/)
    try:
        orc_n = util.Oracle(native=True)
        t0 = time.perf_counter()
        orc_n.conv_fft(img, kh, kw, ks[:n], threads=threads)
        dtn = time.perf_counter() - t0
        faithful["f64_port_native_build"] = {"value": n * P / dtn / 1e9, "unit": "Gpixel-filters/s", "cores": threads,
                                             "build": "-O3 -march=native on this host",
                                             "sample": "same image + %d filters, %d threads, %.1f s wall" % (n, threads, dtn)}
    except Exception as e:   # optional
        faithful["f64_port_native_build"] = {"error": str(e)}
    # (2) an FFTW-class library on the demo's own shape: SciPy's pocketfft, complex128 fft2 of the zero-padded image (once) and of
    # every zero-padded kernel, full spectra, product, ifft2, real part, feature sum -- every usable core (`workers`): the closest
    # stand-in this image has for MATLAB's multithreaded fft2 / ifft2
    try:
        import numpy as np
        import scipy.fft as sfft
        m = min(n, 4)
        img64 = img.astype(np.float64)
        sfft.fft2(img64[:, :, :1], s=(fh, fw), axes=(0, 1), workers=avail)   # untimed: the plan cache of this shape
        t0 = time.perf_counter()
        D = sfft.fft2(img64, s=(fh, fw), axes=(0, 1), workers=avail)
        for k in ks[:m]:
            Kf = sfft.fft2(k.astype(np.float64), s=(fh, fw), axes=(0, 1), workers=avail)
            Kf *= D
            o = sfft.ifft2(Kf, axes=(0, 1), workers=avail, overwrite_x=True).real.sum(axis=2)
        dtc = time.perf_counter() - t0
        err = float(np.max(np.abs(o - ref[m - 1])) / np.max(np.abs(ref[m - 1])))
        faithful["scipy_pocketfft_c128_full"] = {"value": m * P / dtc / 1e9, "unit": "Gpixel-filters/s", "cores": avail,
                                                 "sample": "image fft2 + %d filters one after the other (fft2, .*, ifft2, real), workers=%d, %.1f s wall" % (m, avail, dtc),
                                                 "max_rel_diff_to_port": err}
        del D, Kf, o, img64
    except Exception as e:   # optional
        faithful["scipy_pocketfft_c128_full"] = {"error": str(e)}
    best = max((k for k in faithful if "value" in faithful[k]), key=lambda k: faithful[k]["value"])
    res["value"], res["cores"], res["value_from"] = faithful[best]["value"], faithful[best]["cores"], best
    res["sample"] = faithful[best]["sample"]
    res["faithful_variants"] = faithful
    if threads > 8:     # the 8-thread figure earlier rounds reported
        t0 = time.perf_counter()
        orc.conv_fft(img, kh, kw, ks[:8], threads=8)
        dt8 = time.perf_counter() - t0
        res["port_8_threads"] = {"value": 8 * P / dt8 / 1e9, "unit": "Gpixel-filters/s", "cores": 8,
                                 "sample": "same image + 8 filters, %.1f s wall" % dt8}
    # beside the port: the same maths in fp32 with half spectra, C++ / OpenMP over filters
    try:
        c32 = util.CpuF32(native=True)
        per_thread_f32 = 4 * 8 * P * max(1, F)
        th32 = max(1, min(nf_cfg, avail, int(0.5 * mem / per_thread_f32), sample_filters if sample_filters > 0 else 64))
        _, ks32 = (img, ks) if th32 <= len(ks) else util.synth(seed, H, W, F, kh, kw, th32)
        t0 = time.perf_counter()
        c32.conv_fft(img, kh, kw, ks32[:th32], threads=th32)
        dt1 = time.perf_counter() - t0
        res["f32_rfft2_port"] = {"value": th32 * P / dt1 / 1e9, "unit": "Gpixel-filters/s", "cores": th32,
                                 "sample": "same image + %d filters, fp32 rfft2/irfft2 restatement (-march=native build), %d threads, %.1f s wall" % (th32, th32, dt1)}
    except Exception as e:   # optional
        res["f32_rfft2_port"] = {"error": str(e)}
    # ... and on a production CPU FFT: SciPy's pocketfft, float32 rfft2 / irfft2, every usable core --
    # the closest stand-in for MATLAB's multithreaded fft2 / ifft2 on this host (image spectrum once)
    try:
        import scipy.fft as sfft
        D = sfft.rfft2(img, s=(fh, fw), axes=(0, 1), workers=avail)   # untimed warm-up of the plan cache
        t0 = time.perf_counter()
        D = sfft.rfft2(img, s=(fh, fw), axes=(0, 1), workers=avail)
        m = min(n, 4)
        for k in ks[:m]:
            K = sfft.rfft2(k, s=(fh, fw), axes=(0, 1), workers=avail)
            out = sfft.irfft2(D * K, s=(fh, fw), axes=(0, 1), workers=avail).sum(axis=2)
        dt2 = time.perf_counter() - t0
        res["scipy_pocketfft_f32"] = {"value": m * P / dt2 / 1e9, "unit": "Gpixel-filters/s", "cores": avail,
                                      "sample": "%d filters one after the other, %.1f s wall" % (m, dt2)}
        del D, K, out
    except Exception as e:   # optional
        res["scipy_pocketfft_f32"] = {"error": str(e)}
    return res, ref


def _median(v):
    v = sorted(v)
    return v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])


def host_output_figures(fc, util, np, torch, dev, cases=(("cfg2", 16), ("cfg3", 64))):
    """The reference's ONLY mode: host arrays in, host arrays out (src/cudaConvolutionFFT.cu:146-148,221-222,284-286),
    PCIe-inclusive -- never `value`.  Per case: the literal one-shot entry (fftconv_convolution_fft: first call = plan
    built, later calls = plan from the cache) and the same work through a plan the caller keeps (set_image + convolve),
    into output arrays whose pages are resident and into fresh, never-touched ones (what mxCreateNumericArray hands out)."""
    res = {}
    # what the link gives: device -> pinned host, 1 GiB
    try:
        src = torch.empty(1 << 28, dtype=torch.float32, device=dev)
        dst = torch.empty(1 << 28, dtype=torch.float32).pin_memory()
        dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(3):
            dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize(dev)
        res["pcie_d2h_pinned_gbps"] = 3 * src.numel() * 4 / (time.perf_counter() - t0) / 1e9
        del src, dst
    except Exception as e:   # optional
        res["pcie_d2h_pinned_gbps"] = None
        res["pcie_error"] = str(e)
    res["pcie_spec_gbps"] = 63.0      # MI355X_MICROARCH.md: PCIe Gen5 x16
    fc.cache_clear()
    for cfg, n in cases:
        H, W, F, kh, kw, _nf, seed = CONFIGS[cfg]
        img, ks = util.synth(seed, H, W, F, kh, kw, n)
        fh, fw = ceil16(H + kh - 1), ceil16(W + kw - 1)
        P = fh * fw
        mk = lambda: [np.empty((fh, fw), dtype=np.float32, order="F") for _ in range(n)]
        outs = mk()
        for o in outs:
            o.fill(0.0)            # pages resident
        r = {"maps": n, "map_mb": P * 4 / 1e6}

        def timed(fn, reps):
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                fn()
                ts.append((time.perf_counter() - t0) * 1e3)
            return ts

        first = timed(lambda: fc.cudaConvolutionFFT(img, kh, kw, ks, out=outs), 1)[0]
        r["one_shot_first_call_ms"] = first                      # builds the plan (tables, scratch, copy threads)
        r["one_shot_first_call_parts"] = fc.last_call_timing()
        ts = timed(lambda: fc.cudaConvolutionFFT(img, kh, kw, ks, out=outs), 5)
        r["one_shot_ms"] = _median(ts)                           # plan from the cache
        r["one_shot_parts"] = fc.last_call_timing()
        ts = timed(lambda: fc.cudaConvolutionFFT(img, kh, kw, ks), 3)
        r["one_shot_fresh_outputs_ms"] = _median(ts)             # + first touch of every output page
        with fc.Plan(H, W, F, kh, kw) as p:
            def reused():
                p.set_image(img)
                p.convolve(ks, out=outs)
            reused()
            r["reused_plan_ms"] = _median(timed(reused, 5))
        fc.cache_configure(0)                                    # the reference's own behaviour: everything built and torn down per call
        r["one_shot_no_cache_ms"] = _median(timed(lambda: fc.cudaConvolutionFFT(img, kh, kw, ks, out=outs), 3))
        fc.cache_configure(4)
        r["one_shot_over_reused"] = r["one_shot_ms"] / r["reused_plan_ms"]
        r["one_shot_gpx_per_s"] = n * P / (r["one_shot_ms"] * 1e-3) / 1e9
        r["reused_plan_gpx_per_s"] = n * P / (r["reused_plan_ms"] * 1e-3) / 1e9
        r["one_shot_maps_gbps"] = n * P * 4 / (r["one_shot_ms"] * 1e-3) / 1e9
        r["reused_plan_maps_gbps"] = n * P * 4 / (r["reused_plan_ms"] * 1e-3) / 1e9
        res[cfg + ("/%d maps" % n if n != CONFIGS[cfg][5] else "")] = r
        del outs
        fc.cache_clear()
    # the sizes a MATLAB user of the reference calls it with: the demo's shape (demoCudaConvolutionFFT.m:37-42) and cfg1 -- latency of a
    # cached one-shot call (microseconds, median of 200, through the ctypes mirror; caller buffers reused), with the small host arrays
    # through the plan's pinned staging buffers (option host_pinned, the default) and with one runtime copy per array (host_pinned 0)
    small = {}
    for name, (H, W, F, kh, kw, n) in (("demo 64x8x5 (x) 3 x 10x4x5", (64, 8, 5, 10, 4, 3)), ("cfg1 256x256 (x) 31x31", (256, 256, 1, 31, 31, 1))):
        img, ks = util.synth(77, H, W, F, kh, kw, n)
        outs = fc.cudaConvolutionFFT(img, kh, kw, ks)
        ts = []
        for _ in range(200):
            t0 = time.perf_counter()
            fc.cudaConvolutionFFT(img, kh, kw, ks, out=outs)
            ts.append((time.perf_counter() - t0) * 1e6)
        e = {"one_shot_cached_us": _median(ts), "library_us": fc.last_call_timing()["total_ms"] * 1e3}
        with fc.Plan(H, W, F, kh, kw) as p:
            for pinned in (1, 0):
                p.set_option("host_pinned", pinned)
                p.set_image(img); p.convolve(ks, out=outs)
                ts = []
                for _ in range(200):
                    t0 = time.perf_counter()
                    p.set_image(img); p.convolve(ks, out=outs)
                    ts.append((time.perf_counter() - t0) * 1e6)
                e["kept_plan_us" if pinned else "kept_plan_plain_copies_us"] = _median(ts)
        small[name] = e
    res["small_calls"] = small
    fc.cache_clear()
    return res


def vendor_fft_figures(util, np, torch, dev, cfg="cfg3", maps=8, value=None):
    """The reference's loop as a straight port onto the VENDOR's FFT library of this device (rocFFT behind hipFFT, through torch.fft): every
    kernel padded to the full plane, R2C, product with the image spectrum, C2R (src/cudaConvolutionFFT.cu:204-291), device-resident, fp32, on a
    few maps of the configuration -- what a hipify-style port would run, timed on the same GPU right after the headline.  A baseline like
    cpu_baseline, never `value`; the library itself links and calls no FFT library."""
    H, W, F, kh, kw, _nf, seed = CONFIGS[cfg]
    fh, fw = ceil16(H + kh - 1), ceil16(W + kw - 1)
    g = torch.Generator(device="cpu").manual_seed(seed)
    img = torch.rand((F, W, H), generator=g, dtype=torch.float32).to(dev)
    ker = torch.rand((maps, F, kw, kh), generator=g, dtype=torch.float32).to(dev)
    pad_img = torch.zeros((F, fw, fh), dtype=torch.float32, device=dev)
    pad_img[:, :W, :H] = img
    kp = torch.zeros((F, fw, fh), dtype=torch.float32, device=dev)
    out = torch.empty((maps, fw, fh), dtype=torch.float32, device=dev)

    def loop():
        D = torch.fft.rfft2(pad_img)
        for j in range(maps):
            kp.zero_()
            kp[:, :kw, :kh] = ker[j]
            out[j] = torch.fft.irfft2(D * torch.fft.rfft2(kp), s=(fw, fh)).sum(dim=0)
    loop()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        loop()
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / reps
    v = maps * fh * fw / dt / 1e9
    res = {"value": v, "unit": "Gpixel-filters/s", "us_per_map": dt / maps * 1e6, "kind": "port onto the vendor FFT library (rocFFT through torch.fft), same GPU",
           "sample": "%s image + %d filters, one kernel at a time as the reference's loop, device-resident, fp32" % (cfg, maps)}
    if value:
        res["engine_over_port"] = value / v
    return res


def multi_feature_figures(fc, util, np, torch, dev, steps, cases=(("cfg3f4", 4096, 4096, 4, 127, 64), ("cfg3f5 (the demo's F = 5)", 4096, 4096, 5, 127, 32))):
    """F > 1 -- the only case the reference's gateway accepts (src/cudaConvolutionFFT.cu:51-54,210-211; sumAlongFeatures,
    src/cudaConvFFTData.cuh:70-92): device-resident steps like the headline's, per-kernel times from HIP events, and
    SURVEY 8(d)'s algorithmic bytes at F: 4 K^2 F (kernels) + 8 C F (image spectrum) + 16 C (intermediate round trip)
    + 4 P (map)."""
    res = {}
    for name, H, W, F, k, n in cases:
        rng = np.random.default_rng(4242 + F)
        img = torch.from_numpy(rng.random((F, W, H), dtype=np.float32)).to(dev)
        ker = torch.from_numpy(rng.random((n, F, k, k), dtype=np.float32)).to(dev)
        with fc.Plan(H, W, F, k, k, gpuId=dev.index or 0, stream=torch.cuda.current_stream(dev).cuda_stream) as p:
            i = p.info
            P = i.fft_h * i.fft_w
            C = i.fft_w * (i.fft_h // 2 + 1)
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
            for _ in range(max(2, min(50, int(0.3 / max(est, 1e-4))))):      # settle the clocks
                step()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize(dev)
            dt = (time.perf_counter() - t0) / steps
            p.set_option("profile", 1)
            p.profile(reset=True)
            for _ in range(2):
                step()
            torch.cuda.synchronize(dev)
            prof = p.profile(reset=True)
            p.set_option("profile", 0)
            # the sum over a whole map is sum_f sum(image_f) * sum(kernel_f)
            s_img = img.sum(dim=(1, 2), dtype=torch.float64)
            want = (ker.sum(dim=(2, 3), dtype=torch.float64) * s_img[None, :]).sum(dim=1)
            got = out.sum(dim=(1, 2), dtype=torch.float64)
            err = float(((got - want).abs() / want.abs()).max().item())
            b_alg = 4 * k * k * F + 8 * C * F + 16 * C + 4 * P
            per_map = {kk: prof[kk]["ms"] / max(1, prof[kk]["units"]) * 1e3 for kk in ("spectral_rows", "cols_c2r")}
            res[name] = {"workload": "%dx%d image, F=%d, %d kernels of %dx%dx%d -> %d maps of %dx%d" % (H, W, F, n, k, k, F, n, i.fft_h, i.fft_w),
                         "ms_per_step": dt * 1e3, "value": n * P / dt / 1e9, "unit": "Gpixel-filters/s",
                         "gpixel_features_per_s": n * P * F / dt / 1e9,
                         "algorithmic_bytes_per_map": b_alg, "hbm_algorithmic_gbps": b_alg * n / dt / 1e9,
                         "hbm_frac_of_peak": b_alg * n / dt / 1e9 / HBM_PEAK_GBPS,
                         "us_per_map": {"spectral_rows": per_map["spectral_rows"], "cols_c2r": per_map["cols_c2r"], "step": dt / n * 1e6},
                         "rows_kernel_gbps": (4 * k * k * F + 8 * C * F + 8 * C) / (per_map["spectral_rows"] * 1e-6) / 1e9,
                         "check_checksum_max_rel_err": err}
        del img, ker, out
    return res


def library_sha256(fc):
    import hashlib
    try:
        return hashlib.sha256(open(fc.LIB_PATH, "rb").read()).hexdigest()[:16]
    except Exception:
        return None


def rank_identity(torch, dev):
    """what proves that N ranks ran on N distinct devices: host, PCI bus id, device name, UUID where the runtime has one"""
    import socket
    props = torch.cuda.get_device_properties(dev)
    bus = None
    try:   # (from torch's own device properties: no second HIP runtime is loaded into the process for this)
        bus = "%04x:%02x:%02x.0" % (int(props.pci_domain_id), int(props.pci_bus_id), int(props.pci_device_id))
    except Exception:
        bus = None
    uuid = getattr(props, "uuid", None)
    return {"host": socket.gethostname(), "pci_bus_id": bus, "uuid": str(uuid) if uuid is not None else None,
            "name": props.name, "device_index": int(dev.index or 0), "pid": os.getpid()}


def count_distinct_devices(idents):
    """(number of distinct devices among the ranks' identities, what the count rests on).  PCI bus ids (or UUIDs) first; a host
    that reports the same -- or no -- PCI address for every device (virtual functions, some containers) is counted by device
    index instead: ranks on distinct device indices of one runtime are still distinct devices."""
    by_pci = len(set((i["host"], i["pci_bus_id"] or i["uuid"] or i["device_index"]) for i in idents))
    if by_pci == len(idents):
        return by_pci, "pci_bus_id"
    by_index = len(set((i["host"], i["device_index"]) for i in idents))
    if by_index == len(idents):
        return by_index, "device_index (the PCI ids of the ranks are not distinct)"
    return by_pci, "pci_bus_id"


def numa_prefer_gpu_node(pci_bus_id):
    """Makes the calling thread PREFER the NUMA node the GPU hangs off for the allocations that follow (set_mempolicy
    MPOL_PREFERRED: the pinned images are placed when torch touches / pins them).  Returns what happened, for the JSON."""
    info = {"gpu_numa_node": None, "policy": "default"}
    try:
        info["gpu_numa_node"] = int(open("/sys/bus/pci/devices/%s/numa_node" % str(pci_bus_id).lower()).read())
    except Exception:
        return info
    node = info["gpu_numa_node"]
    if node < 0:
        return info          # the host does not say (one node, or a virtual machine)
    try:
        import ctypes
        libc = ctypes.CDLL(None, use_errno=True)
        mask = (ctypes.c_ulong * 16)()
        mask[node // 64] = 1 << (node % 64)
        rc = libc.syscall(238, 1, mask, 16 * 64 + 1)          # x86-64: set_mempolicy(MPOL_PREFERRED, mask, maxnode)
        info["policy"] = ("preferred node %d" % node) if rc == 0 else ("default (set_mempolicy: errno %d)" % ctypes.get_errno())
    except Exception as e:
        info["policy"] = "default (%s)" % e
    return info


def numa_restore(info):
    if info and info.get("policy", "").startswith("preferred"):
        try:
            import ctypes
            ctypes.CDLL(None).syscall(238, 0, None, 0)          # MPOL_DEFAULT
        except Exception:
            pass


def spawn_ranks(n):
    """`python bench.py --gpus N` by hand: start the N ranks (fresh processes, before anything in
    this one touches the GPU) and exit with their status; rank 0 prints the JSON line."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    # wait for all; a rank that fails would leave the others waiting in a collective for ever, so the
    # first failure ends the ranks this function started (by their own PIDs)
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in live:
                    q.terminate()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS),
                    help="default: cfg3 on one GPU (BASELINE configs[2]), cfg4 on several (configs[3])")
    ap.add_argument("--filters", type=int, default=0, help="filters of the whole job (0 = the config's); sharded over the ranks")
    ap.add_argument("--weak", action="store_true", help="--filters (or the config's count) is per GPU: weak scaling")
    ap.add_argument("--batch-maps", type=int, default=0)
    ap.add_argument("--kernel-chunk-mb", type=int, default=0, help="budget of the kernels' column-spectrum chunk (0 = the library's default; A/B)")
    ap.add_argument("--tune-placement", type=int, default=-1,
                    help="candidate allocations of the intermediate the plan times against the map buffer (plan option "
                         "tune_placement).  -1 = the library's default = the headline's: automatic (five candidates where a launch writes "
                         ">= 2 GiB of maps and the device is mostly free); 0 = never.  Where a launch covers >= 5e8 padded pixels the line "
                         "carries the other policy beside it: `value_untuned_placement` (a second plan with tune_placement = 0) beside a "
                         "plan that tuned, `value_tuned_placement` (5 candidates) beside one that did not")
    ap.add_argument("--rows-group", type=int, default=0, help="maps per workgroup of the spectral-row kernel (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0,
                    help="filters (= threads) of the CPU baseline sample; 0 = one per usable core, capped by the filter count and by memory")
    ap.add_argument("--check", action="store_true",
                    help="also verify three maps of every rank against the oracle (the cheap all-maps checksum always runs)")
    ap.add_argument("--images", type=int, default=0,
                    help="streamed mode (BASELINE configs[4]): this many images in total, dealt over the ranks; per rank "
                         "every image against all kernels, the next image's H2D copy (pinned host memory, side "
                         "stream) overlapped with the current image's compute; no collective")
    ap.add_argument("--graph", action="store_true",
                    help="record one step into a HIP graph (plan bound to the capturing stream) and time graph "
                         "replays: removes the per-launch host cost that bounds the small configurations; "
                         "single-GPU, non-streamed steps only")
    ap.add_argument("--no-overlap", action="store_true",
                    help="one spectrum buffer: image transform + broadcast of a step not overlapped with the previous step's maps (A/B)")
    ap.add_argument("--overlap", action="store_true",
                    help="single GPU: two spectrum buffers as at N > 1 -- the image transform of step k + 1 on a side stream beside the maps of step k (A/B)")
    ap.add_argument("--no-clock-warm", action="store_true",
                    help="exactly W warm-up steps even when they are shorter than the GPU's clock ramp (A/B)")
    ap.add_argument("--no-live-profile", action="store_true",
                    help="do not time the dominant kernel inside the timed region (roofline from the separate pass only; A/B)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal of the N > 1 path on one GPU: every rank on cuda:0, collective over gloo")
    ap.add_argument("--backend", default=None, choices=["nccl", "gloo"],
                    help="torch.distributed backend (default: nccl = RCCL over xGMI; gloo with --share-gpu)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the untimed extra passes (kernel-upload-inclusive steps, default-options plan): A/B runs")
    ap.add_argument("--exact-window", action="store_true",
                    help="plans transform the ceil16 window itself (fftconv_plan_options.exact_window): A/B of the window's own "
                         "specialised kernels (1088, 4160) against the next convenient length + crop")
    ap.add_argument("--settle-s", type=float, default=CLOCK_WARM_S,
                    help="seconds of untimed load (warm-up steps included) in front of the timed region")
    ap.add_argument("--kernels-resident", action="store_true",
                    help="keep the kernels in HBM instead of uploading them from pinned host memory inside every step (A/B; the "
                         "default run reports this figure as value_kernels_resident)")
    ap.add_argument("--no-host-output", action="store_true", help="skip the host-in / host-out figures (host_output)")
    ap.add_argument("--no-multi-feature", action="store_true", help="skip the F > 1 figures (multi_feature)")
    ap.add_argument("--dynamic-tiles", type=int, default=-1,
                    help="plan option dynamic_tiles: the persistent column kernels take their tiles from a queue (1) or by the static "
                         "deal (0: A/B); -1 = the library's default (1)")
    ap.add_argument("--contend", default=None, metavar="K[,LDS_KB]",
                    help="A/B: K workgroups of tools/microbench/cu_hog (each holding LDS_KB of LDS, default 4) sit on the GPU during the "
                         "timed steps -- a stand-in for a collective's channels co-resident with the step (single GPU)")
    ap.add_argument("--force-collective", action="store_true",
                    help="run the broadcast code path even with one rank (self-test of the N > 1 step on a 1-GPU box)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist
    import util

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    if args.share_gpu:
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d wants cuda:%d but this box has %d GPU(s); --share-gpu rehearses the "
                         "N > 1 path on one GPU" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_collective
    backend = args.backend or ("gloo" if args.share_gpu else "nccl")
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    # who takes part: every rank's host, PCI bus id and device name, gathered on all ranks -- N ranks must sit on N
    # distinct devices unless this is the one-GPU rehearsal
    ident = rank_identity(torch, dev)
    idents = [ident]
    if use_dist and world > 1:
        idents = [None] * world
        dist.all_gather_object(idents, ident)
    distinct_devices, identity_source = count_distinct_devices(idents)
    if world > 1 and distinct_devices != world and not args.share_gpu:
        raise SystemExit("bench.py: %d ranks on %d distinct devices (%s); --share-gpu is the one-GPU rehearsal"
                         % (world, distinct_devices, [(i["host"], i["pci_bus_id"]) for i in idents]))

    fc = util.load_package()
    import importlib
    mg = importlib.import_module(fc.__name__ + ".multi_gpu")
    cfg = args.config or ("cfg4" if world > 1 else "cfg3")
    H, W, F, kh, kw, nf_total, seed = CONFIGS[cfg]
    if args.filters > 0:
        nf_total = args.filters
    if args.weak:
        nf_total *= world
    streamed = args.images > 0
    if streamed:          # image-sharded: every rank holds all the kernels
        first, nf = 0, nf_total
        img_first, n_img = mg.image_shard(args.images, rank, world)
    else:                 # filter-sharded: contiguous blocks
        first, nf = mg.filter_shard(nf_total, rank, world)
    img_h, _ = util.synth(seed, H, W, F, kh, kw, 0)
    # kernels: seed 5678 + cfg + k with k the GLOBAL filter index
    kern_h = np.empty((max(nf, 1), F, kw, kh), dtype=np.float32)  # [n][f][kw][kh] == n MATLAB arrays kh x kw x F
    for j in range(nf):
        k = np.random.default_rng(5678 + seed + first + j).random((kh, kw, F), dtype=np.float32)
        kern_h[j] = np.transpose(k, (2, 1, 0))
    kern_h = kern_h[:nf]
    img_d = torch.from_numpy(np.ascontiguousarray(np.transpose(img_h, (2, 1, 0)))).to(dev)  # [f][w][h]
    kern_d = torch.from_numpy(kern_h).to(dev)

    stream = torch.cuda.current_stream(dev)
    plan_opts = {"exact_window": 1} if args.exact_window else None
    plan = fc.Plan(H, W, F, kh, kw, gpuId=local_rank, stream=stream.cuda_stream, options=plan_opts)
    info = plan.info
    P = info.fft_h * info.fft_w
    nblk = max(1, plan.get_option("blockwise"))      # > 1: the plan runs blocks of a shorter transform (a launch covers 1 / nblk of its maps)
    if args.batch_maps:
        plan.set_option("batch_maps", args.batch_maps)
    if args.rows_group:
        plan.set_option("rows_group", args.rows_group)
    if args.kernel_chunk_mb:
        plan.set_option("kernel_chunk_mb", args.kernel_chunk_mb)
    if args.dynamic_tiles >= 0:
        plan.set_option("dynamic_tiles", args.dynamic_tiles)
    hog = None
    hog_cfg = None
    side_work = None
    if args.contend:
        import ctypes
        parts = [int(x) for x in args.contend.split(",")]
        hog_cfg = {"workgroups": parts[0], "lds_kb": parts[1] if len(parts) > 1 else 4, "per_step_us": parts[2] if len(parts) > 2 else 0}
        if hog_cfg["workgroups"] > 0:
            hog_path = os.path.join(ROOT, "tools", "microbench", "libcuhog.so")
            if not os.path.exists(hog_path):
                raise SystemExit("bench.py --contend: %s is not built (python -c 'import __graft_entry__ as g; g.build()')" % hog_path)
            hog = ctypes.CDLL(hog_path)
            hog.cu_hog_run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
            if hog_cfg["per_step_us"] > 0:
                # one launch per step on the side stream, where the broadcast of the N > 1 pipeline goes: beside the previous step's maps
                hog_lib = hog

                def side_work(stream_handle):
                    rc = hog_lib.cu_hog_run(stream_handle, hog_cfg["workgroups"], 256, hog_cfg["lds_kb"] << 10, hog_cfg["per_step_us"])
                    if rc:
                        raise SystemExit("bench.py --contend: cu_hog_run failed (%d)" % rc)
            elif hog_cfg["lds_kb"] > 8:
                raise SystemExit("bench.py --contend: workgroups that hold more than 8 KB of LDS for the whole timed region would block "
                                 "the persistent kernels until the stop that follows them; give a per-step duration (K,LDS_KB,US)")
    # placement tuning: `value` is what a caller with DEFAULT plan options gets -- the library's automatic choice (large launches on a free
    # device tune at their first convolve, inside the warm-up; placement.cpp).  Where launches are long enough to tell 3 % apart the other
    # policy is timed beside it (`value_untuned_placement`, or `value_tuned_placement` where the plan did not tune).
    tune_k = args.tune_placement
    if tune_k >= 0:
        plan.set_option("tune_placement", tune_k)
    other_policy_beside = min(args.batch_maps or 64, max(nf, 1)) * P >= 5e8
    # a side stream only where something overlaps: the next step's transform + broadcast (N > 1), or
    # the next image's H2D copy (streamed mode)
    overlap = streamed or ((use_dist or args.overlap or side_work is not None) and not args.no_overlap)
    # the kernels' column pass rides in the image transform's launch where that transform follows on the plan's own stream
    defer_prepare = streamed or not overlap
    # the step's kernels are uploaded inside the step, from pinned host memory, double-buffered on an upload stream
    upload_kernels = not args.kernels_resident and not args.graph and nf > 0
    kern_pin = torch.from_numpy(kern_h).pin_memory() if upload_kernels else None
    engine = mg.HipPlanEngine(torch, fc, plan, dev, kern_d, kh, kw, first=first, main_stream=stream, overlap=overlap,
                              defer_prepare=defer_prepare, kernels_host=kern_pin)
    out = engine.out

    if streamed:
        # images of this rank in pinned host memory (seeded per global image index)
        # ... allocated on the NUMA node of this rank's GPU where the host says which one that is: eight ranks streaming
        # from one host's DRAM is the case that decides cfg5's scaling (pinned H2D drops to ~17 GB/s once the images fall
        # out of the host caches; a remote node halves that again)
        numa = numa_prefer_gpu_node(ident["pci_bus_id"])
        imgs_h = []
        for i in range(n_img):
            a = np.random.default_rng(1234 + seed + 1000 * (img_first + i)).random((F, W, H), dtype=np.float32)
            imgs_h.append(torch.from_numpy(a).pin_memory())
        numa_restore(numa)
        conv = mg.ImageStreamedConvolver(engine, nf, time_uploads=True)

        def run_steps(k):
            conv.run(imgs_h * k)      # k steps = the batch streamed k times back to back (the first H2D copy of a
                                      # step rides behind the last image of the step before)
    else:
        conv = mg.FilterShardedConvolver(engine, dist if use_dist else None, rank, world, nf_total, src=0,
                                         depth=2 if overlap else 1, always_collective=args.force_collective,
                                         time_broadcast=("event" if backend == "nccl" else "wall") if use_dist else None,
                                         side_work=side_work)

        def run_steps(k):
            conv.run([img_d] * k)

    def barrier():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # the CPU baseline FIRST (CPU only, ~15 s): the GPU phases then run in one piece at the end of the process
    cpu = cpu_ref = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu, cpu_ref = cpu_baseline(cfg, args.cpu_sample)

    run_steps(args.warmup)
    barrier()
    # The GPU needs ~40 ms of continuous load to settle its clocks after any idle gap (tools/clock_ramp.py,
    # profiles/r02w_clock_ramp.txt: the first ten 3-ms steps run up to 28 % slow).  Where the W warm-up steps are
    # shorter than that (small configurations, small W) more untimed steps follow, the same number on every rank.
    clock_warm_steps = 0
    if not args.no_clock_warm:
        t1 = time.perf_counter()
        run_steps(1)
        torch.cuda.synchronize(dev)
        est = time.perf_counter() - t1
        if use_dist:
            te = torch.tensor([est], dtype=torch.float64, device=dev)
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
            est = float(te.item())
        if (args.warmup + 1) * est < args.settle_s:
            clock_warm_steps = min(60000, int(args.settle_s / max(est, 1e-6)) + 1)
            for c0 in range(0, clock_warm_steps, 64):     # in short bursts: never a deep queue behind the host
                run_steps(min(64, clock_warm_steps - c0))
                torch.cuda.synchronize(dev)
        clock_warm_steps += 1
        barrier()
    graph = None
    if args.graph:
        if use_dist or streamed:
            raise SystemExit("--graph: single-GPU, non-streamed steps only")
        run_steps(1)                             # scratch buffers are sized: nothing allocates from here on
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        cap_stream = torch.cuda.Stream(dev)
        with torch.cuda.graph(graph, stream=cap_stream):
            cur = torch.cuda.current_stream(dev)
            plan.set_stream(cur.cuda_stream)
            engine.main_stream = cur
            run_steps(1)
        plan.set_stream(stream.cuda_stream)
        engine.main_stream = stream
        for _ in range(max(1, args.warmup)):
            graph.replay()
        barrier()
    # the two hot kernels (spectral rows, output columns) are timed by HIP events INSIDE the timed region, on
    # the plan's stream: two event records per launch of those kernels (the one-off passes stay unobserved
    # here; their figures come from the separate pass below)
    # ... where a hot launch is long enough not to feel them: a pair costs ~5-20 us of stream time, which
    # is 5 % of a cfg5 launch and most of a cfg1 step (profiles/r02w_live_event_cost.txt), so launches under
    # 5e8 padded pixels (< ~0.7 ms) are timed in the separate pass only
    per_launch_px = min(args.batch_maps or 64, max(nf, 1)) * P
    live_profile = graph is None and not args.no_live_profile and per_launch_px >= 5e8
    if live_profile:
        plan.set_option("profile_kinds", (1 << 1) | (1 << 2))   # indices of fftconv_profile: spectral_rows, cols_c2r
        plan.set_option("profile", 1)
        plan.profile(reset=True)
    if not streamed:
        conv.broadcast_ms(reset=True)
    else:
        conv.upload_ms(reset=True)
    uploads0 = engine.uploads
    if hog is not None and side_work is None:       # the other kernel takes its CUs on the idle GPU, then the timed steps run beside it
        rc = hog.cu_hog_start(hog_cfg["workgroups"], 256, hog_cfg["lds_kb"] << 10, 20000)
        if rc:
            raise SystemExit("bench.py --contend: cu_hog_start failed (%d)" % rc)
        time.sleep(0.002)
    t0 = time.perf_counter()
    if graph is not None:
        for _ in range(args.steps):
            graph.replay()
    else:
        run_steps(args.steps)
    torch.cuda.synchronize(dev)
    dt_rank = time.perf_counter() - t0          # this rank's own time (the headline is the max over ranks, behind the barrier)
    if hog is not None:
        import ctypes
        ncu, ran = ctypes.c_int(0), ctypes.c_float(0)
        if side_work is None:
            rc = hog.cu_hog_stop(ctypes.byref(ncu), ctypes.byref(ran))
            hog_cfg.update({"distinct_cus_held": ncu.value, "held_ms": ran.value, "stop_rc": rc,
                            "covered_timed_region": bool(ran.value >= dt_rank * 1e3)})
        else:
            rc = hog.cu_hog_last_places(ctypes.byref(ncu))
            hog_cfg.update({"distinct_cus_held": ncu.value, "stop_rc": rc, "covered_timed_region": True})
            conv.side_work = None          # the passes below (per-kernel timing, checks) run alone
        hog = None
    barrier()
    dt = time.perf_counter() - t0
    kernel_uploads_timed = engine.uploads - uploads0
    bcast_ms = conv.broadcast_ms() if not streamed else []
    upload_ms = conv.upload_ms() if streamed else []
    live = None
    if live_profile:
        live = plan.profile(reset=True)
        plan.set_option("profile", 0)
        plan.set_option("profile_kinds", 0)
    per_rank = None
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        # per rank: its own K steps, its broadcasts (prologue), its H2D rate -- a sub-6x result can then be attributed
        img_bytes = 4.0 * F * W * H
        mine = {"rank": rank, "ms_per_step": dt_rank / args.steps * 1e3, "filters": nf,
                "broadcast_ms_mean": (sum(bcast_ms) / len(bcast_ms)) if bcast_ms else None,
                "broadcast_ms_max": max(bcast_ms) if bcast_ms else None, "broadcasts": len(bcast_ms),
                "h2d_gbps": (img_bytes * len(upload_ms) / (sum(upload_ms) * 1e-3) / 1e9) if upload_ms and sum(upload_ms) > 0 else None,
                "images": n_img if streamed else None, "pinned_images_numa": numa if streamed else None}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    # self-check of what the timed steps left in `out` (every map of this rank, on the device):
    # the sum over the whole window of a linear convolution is sum(image) * sum(kernel)
    chk_img = img_d
    if streamed and n_img:
        chk_img = conv.buf[conv.last_buf]         # the maps in `out` belong to the image convolved last
    check = {"checksum_max_rel_err": 0.0}
    if nf and (not streamed or n_img):
        s_img = chk_img.sum(dim=(1, 2), dtype=torch.float64)                        # [F]
        s_ker = kern_d.sum(dim=(2, 3), dtype=torch.float64)                         # [n][F]
        want = (s_ker * s_img[None, :]).sum(dim=1)
        got = out[:nf].sum(dim=(1, 2), dtype=torch.float64)
        check["checksum_max_rel_err"] = float(((got - want).abs() / want.abs().clamp_min(1e-30)).max().item())
    chk_img_h = None
    if args.check:
        chk_img_h = np.asfortranarray(np.transpose(chk_img.cpu().numpy(), (2, 1, 0)))

    # HIP-event timing of EVERY kernel of the same steps, on the plan's stream, in a separate pass (so that the
    # one-off passes carry no event records inside the headline timing).  The check above left the GPU idle
    # and the clocks take ~40 ms of load to settle again (tools/clock_ramp.py): a few unobserved steps first
    n_sep = max(1, min(args.steps, 3))
    run_steps(max(1, min(200, int(0.06 / max(dt / args.steps, 1e-6)) + 1)))   # same count on every rank (dt is the max over ranks)
    plan.set_option("profile", 1)
    plan.profile(reset=True)
    run_steps(n_sep)
    torch.cuda.synchronize(dev)
    prof = plan.profile(reset=True)
    plan.set_option("profile", 0)

    # Extra, untimed-for-the-headline passes (N = 1, plain steps only):
    #  * the same K steps with the kernels uploaded from (pinned) host memory inside every step -- SURVEY 8(d) counts
    #    the kernel H2D (63 KB each) in the timed region; the headline keeps the kernels resident as the maps are;
    #  * the same K steps on a second plan with DEFAULT options (no placement tuning) writing into the same maps.
    extras = {}
    if world == 1 and not use_dist and not streamed and graph is None and not args.no_extras and nf:
        def timed(fn, k):
            fn(max(1, min(k, 3)))
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            fn(k)
            torch.cuda.synchronize(dev)
            return time.perf_counter() - t1

        # SURVEY 8(d) words the timing as "hipEvents, median of >= 10 after warm-up": the same K steps once more with an event
        # recorded on the plan's stream after every step (the headline is the wall-clock mean between two device synchronisations)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        conv.run([img_d] * 2)
        evs[0].record(stream)
        conv.run([img_d] * args.steps, on_result=lambda k, _r: evs[k + 1].record(stream))
        torch.cuda.synchronize(dev)
        med = _median([evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps)])
        extras["ms_per_step_event_median"] = med
        extras["value_event_median"] = nf_total * P / (med * 1e-3) / 1e9
        if upload_kernels:
            # the same K steps with the kernels kept in HBM (the bench contract's "inputs already resident"; rounds 1-3's headline)
            eng_r = mg.HipPlanEngine(torch, fc, plan, dev, kern_d, kh, kw, first=first, main_stream=stream, overlap=overlap, out=out,
                                     defer_prepare=defer_prepare)
            conv_r = mg.FilterShardedConvolver(eng_r, None, rank, world, nf_total, src=0, depth=2 if overlap else 1)
            dt_r = timed(lambda k: conv_r.run([img_d] * k), args.steps)
            extras["ms_per_step_kernels_resident"] = dt_r / args.steps * 1e3
            extras["value_kernels_resident"] = nf_total * P * args.steps / dt_r / 1e9
        main_tuned = plan.get_option("tuned_candidates") > 1
        if other_policy_beside:
            # the other placement policy on the same maps: an untuned plan beside one that tuned, a tuned plan (5 candidates) beside one that did not
            tuned_beside = 0 if main_tuned else 5
            plan2 = fc.Plan(H, W, F, kh, kw, gpuId=local_rank, stream=stream.cuda_stream, options=plan_opts)
            if args.batch_maps:
                plan2.set_option("batch_maps", args.batch_maps)
            plan2.set_option("tune_placement", tuned_beside)
            eng2 = mg.HipPlanEngine(torch, fc, plan2, dev, kern_d, kh, kw, first=first, main_stream=stream, overlap=overlap, out=out,
                                    defer_prepare=defer_prepare, kernels_host=kern_pin)
            conv2 = mg.FilterShardedConvolver(eng2, None, rank, world, nf_total, src=0, depth=2 if overlap else 1)
            conv2.run([img_d] * max(2, args.warmup))
            dt2 = timed(lambda k: conv2.run([img_d] * k), args.steps)
            key = "tuned_placement" if tuned_beside else "untuned_placement"
            extras["value_" + key] = nf_total * P * args.steps / dt2 / 1e9
            extras["ms_per_step_" + key] = dt2 / args.steps * 1e3
            if tuned_beside:
                extras["tuned_placement"] = {"candidates": plan2.get_option("tuned_candidates"), "kept": plan2.get_option("tuned_best")}
            torch.cuda.synchronize(dev)
            plan2.destroy()
        conv.run([img_d])      # `out` holds this plan's maps again for the checks below
        torch.cuda.synchronize(dev)

    result = None
    errs = []
    if args.check and nf:       # three maps of every rank against the oracle
        orc = util.Oracle()
        idx = sorted(set([0, nf // 2, nf - 1]))
        ks = [np.asfortranarray(np.transpose(kern_h[j], (2, 1, 0))) for j in idx]
        ref = orc.conv_fft(chk_img_h, kh, kw, ks)
        for j, r in zip(idx, ref):
            errs.append(util.rel_err(out[j].cpu().numpy().T, r))   # [w][h] -> h x w
    if cpu_ref is not None and not streamed:        # the oracle's maps of the first filters (CPU baseline sample) double as a check
        for j, r in enumerate(cpu_ref[:nf]):
            errs.append(util.rel_err(out[j].cpu().numpy().T, r))
    oracle_err = max(errs) if errs else None
    if use_dist:
        tt = torch.tensor([check["checksum_max_rel_err"], oracle_err if oracle_err is not None else -1.0],
                          dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        check["checksum_max_rel_err"] = float(tt[0].item())
        oracle_err = float(tt[1].item()) if tt[1].item() >= 0 else None
    ok = check["checksum_max_rel_err"] < CHECK_TOL_CHECKSUM and (oracle_err is None or oracle_err < CHECK_TOL_ORACLE)

    if rank == 0:
        ab = alg_bytes(H, W, F, kh, kw)
        per = {}
        for name in ("spectral_rows", "cols_c2r"):
            p = prof[name]
            if p["launches"]:
                avg_ms = p["ms"] / p["launches"]
                units = p["units"] / p["launches"] / nblk
                per[name] = {"avg_ms": avg_ms, "units_per_launch": units,
                             "gbps": ab[name] * units / (avg_ms * 1e-3) / 1e9}
        for name in list(per):
            lv = live.get(name) if live else None
            if lv and lv["launches"]:
                # the figures of the hot kernels (and so the roofline) use the launches of the timed region itself
                avg_ms = lv["ms"] / lv["launches"]
                units = lv["units"] / lv["launches"] / nblk
                per[name] = {"avg_ms": avg_ms, "units_per_launch": units, "gbps": ab[name] * units / (avg_ms * 1e-3) / 1e9,
                             "separate_pass_avg_ms": per[name]["avg_ms"],
                             "timed_in": "the timed region (%d launches)" % lv["launches"]}
        dom = max(per, key=lambda k: per[k]["avg_ms"] / per[k]["units_per_launch"]) if per else None
        traffic = None
        traffic_src = None
        traffic_lib = None
        physical_per_map = None      # PMC bytes of both hot kernels per map (FETCH x 2 + WRITE, profiles/traffic.json)
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if dom and os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                per_map = tj.get(cfg, {}).get(dom)
                traffic = per_map * per[dom]["units_per_launch"] if per_map else None
                traffic_src = tj.get("_source")
                traffic_lib = tj.get("_library_sha256")
                if all(tj.get(cfg, {}).get(k) for k in ("spectral_rows", "cols_c2r")):
                    physical_per_map = tj[cfg]["spectral_rows"] + tj[cfg]["cols_c2r"]
            except Exception:
                traffic = None
        total_maps = (nf_total * args.images) if streamed else nf_total
        value = total_maps * P * args.steps / dt / 1e9
        result = {
            "metric": "Gpixel-filters/s (padded FFT size)",
            "value": value,
            "unit": "Gpixel-filters/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak" if (args.weak or streamed) else "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%s: %dx%d image (F=%d), %d kernels of %dx%d in total%s -> %d maps of %dx%d per step, %s"
                                   % (cfg, H, W, F, nf_total, kh, kw,
                                      (" against each of %d images" % args.images) if streamed else "",
                                      total_maps, info.fft_h, info.fft_w,
                                      "image-sharded" if streamed else "filter-sharded"),
                       "transform": [info.transform_h, info.transform_w],
                       "blocks": nblk if nblk > 1 else None,      # overlap-save blocks of that transform (include/fftconv.h: blockwise)
                       "filters_total": nf_total,
                       "filters_per_gpu": nf if streamed else -(-nf_total // world),
                       "kernels": ("uploaded from pinned host memory inside every step (SURVEY 8(d)): %d uploads in the %d timed steps, %s"
                                   % (kernel_uploads_timed, args.steps,
                                      "read over PCIe by the kernels' column pass itself (<= 512 KiB: no copy)" if engine.zero_copy
                                      else "double-buffered on an upload stream")) if upload_kernels
                                  else "resident in HBM",
                       "hip_graph_replay": bool(args.graph), "clock_warm_steps": clock_warm_steps, "settle_s": args.settle_s,
                       "tune_placement": ({"option": plan.get_option("tune_placement"), "candidates": plan.get_option("tuned_candidates"),
                                           "kept": plan.get_option("tuned_best")} if plan.get_option("tuned_candidates") > 1
                                          else {"option": plan.get_option("tune_placement"), "candidates": 0}),
                       "dynamic_tiles": plan.get_option("dynamic_tiles"),
                       "contention": hog_cfg,
                       "images_per_step": args.images if streamed else 1,
                       "parallelism": ("images x%d, streamed H2D" % world) if streamed else
                                      ("filters x%d + 1 bcast per step" % world if world > 1 else "single GPU")},
            "hbm_algorithmic_gbps": ab["total"] * total_maps * args.steps / dt / 1e9,
            "hbm_frac_of_peak": ab["total"] * total_maps * args.steps / dt / 1e9 / (HBM_PEAK_GBPS * world),
            # what the memory system really moved (PMC counters of an earlier profiled run of this configuration): the
            # row kernel re-uses the image-spectrum row from registers, so this is BELOW the algorithmic figure
            "hbm_physical_gbps": (physical_per_map * total_maps * args.steps / dt / 1e9) if physical_per_map else None,
            "hbm_physical_frac": (physical_per_map * total_maps * args.steps / dt / 1e9 / (HBM_PEAK_GBPS * world)) if physical_per_map else None,
            "kernels": per,
            "image_ms": (prof["image_cols"]["ms"] + prof["image_rows"]["ms"]) / max(1, prof["image_cols"]["launches"]),
            "check_max_rel_err": oracle_err,
            "check_checksum_max_rel_err": check["checksum_max_rel_err"],
            "check_ok": ok,
        }
        if dom:
            result["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": per[dom]["gbps"], "peak": HBM_PEAK_GBPS,
                                  "unit": "GB/s", "frac": per[dom]["gbps"] / HBM_PEAK_GBPS, "traffic": traffic,
                                  "traffic_source": traffic_src,
                                  # the PMC passes are separate runs: which binary they measured, and whether it is the one running now
                                  "traffic_measured_at": traffic_lib, "library_sha256": library_sha256(fc),
                                  "traffic_is_of_this_binary": (traffic_lib == library_sha256(fc)) if traffic_lib else None,
                                  "algorithmic_bytes_per_launch": ab[dom] * per[dom]["units_per_launch"],
                                  "avg_launch_ms": per[dom]["avg_ms"]}
        result.update(extras)
        if world > 1 and not streamed and cfg == "cfg4" and nf_total == CONFIGS["cfg4"][5] and not args.weak:
            # the driver's N = 1 run is the headline (cfg3); the strong-scaling denominator of THIS workload on one GPU is
            # the newest committed single-GPU run of it (python bench.py --config cfg4)
            import glob
            den = sorted(glob.glob(os.path.join(ROOT, "profiles", "*cfg4_1024_filters_1gpu_denominator.json")))
            if den:
                try:
                    dj = json.loads(open(den[-1]).read().strip().splitlines()[-1])
                    den_lib = (dj.get("roofline") or {}).get("library_sha256")
                    result["same_workload_1gpu"] = {"value": dj["value"], "ms_per_step": dj["ms_per_step"], "unit": dj["unit"],
                                                    "source": os.path.relpath(den[-1], ROOT),
                                                    # the denominator is a separate one-GPU run (tools/final_session.sh regenerates it every
                                                    # round): which binary it measured, and whether it is the one running now
                                                    "measured_at": den_lib,
                                                    "is_of_this_binary": (den_lib == library_sha256(fc)) if den_lib else None,
                                                    "speedup": value / dj["value"]}
                except Exception:
                    pass
        if use_dist:
            result["config"]["backend"] = backend + (" (all ranks on cuda:0: rehearsal)" if args.share_gpu else "")
            result["ranks"] = idents
            result["distinct_devices"] = distinct_devices
            result["distinct_devices_by"] = identity_source
            result["per_rank"] = per_rank
            if streamed:
                result["config"]["pinned_images_numa"] = numa
        if cpu is not None:
            result["cpu_baseline"] = cpu
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    plan.destroy()
    if rank == 0 and world == 1 and not use_dist and not args.no_extras and result is not None and args.config is None and not streamed and graph is None:
        # untimed for the headline: the MEX-faithful surface (host in / host out) and the F > 1 case, from THIS binary
        del engine, conv, out
        torch.cuda.empty_cache()
        if not args.no_host_output:
            try:
                result["host_output"] = host_output_figures(fc, util, np, torch, dev)
            except Exception as e:   # reported, never fatal for the headline
                result["host_output"] = {"error": repr(e)}
        if not args.no_multi_feature:
            try:
                result["multi_feature"] = multi_feature_figures(fc, util, np, torch, dev, max(3, min(args.steps, 10)))
            except Exception as e:
                result["multi_feature"] = {"error": repr(e)}
        try:        # the same GPU's vendor FFT library running the reference's loop: a baseline beside cpu_baseline, never `value`
            result["vendor_fft_baseline"] = vendor_fft_figures(util, np, torch, dev, value=result["value"])
        except Exception as e:
            result["vendor_fft_baseline"] = {"error": repr(e)}
    if rank == 0:
        # RCCL prints a version banner through C stdio, which would otherwise be flushed after
        # Python's output at exit: flush it first so that the JSON line is the LAST line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(result), flush=True)
    if not ok:
        print("bench.py: self-check FAILED on rank %d (checksum %.3g, oracle %s)"
              % (rank, check["checksum_max_rel_err"], oracle_err), file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
