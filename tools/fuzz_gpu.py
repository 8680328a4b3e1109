#!/usr/bin/env python3
"""Randomised parity run of the HIP path against the CPU oracle (GPU box, repository root).

Every case draws a shape (tiny / small / medium / sitting just under a specialised transform length), a feature count, a
ragged cell of kernels, an entry (one-shot, plan with host arrays, plan with device-resident packed arrays, the two-step pair)
and a set of plan options (kernel path, walk length, forced block-wise plans, batch size, correlation, cropped regions, the
host copy-out machinery), runs it through the C ABI and compares every map with oracle/ (bar: 1e-5 of the map's maximum; the
north-star bar is 1e-4).  A failing case prints the line that reproduces it (--seed S --only I) and the run exits non-zero.

    python tools/fuzz_gpu.py --seconds 240 --seed 1        # as many cases as fit
    python tools/fuzz_gpu.py --cases 40 --seed 7           # what tests/test_gpu_parity.py::test_fuzz_slice runs
"""
import argparse, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import util

FAST_LENGTHS = [288, 384, 480, 576, 672, 768, 864, 960, 1088, 1152, 1280, 1344, 1536, 1760, 1920, 2112, 2304]
TOL = 1e-5


def draw_case(rng, max_pixels):
    cls = rng.choice(["tiny", "small", "small", "medium", "medium", "fast", "fast", "strip"] + (["large"] * 8 if max_pixels >= 30_000_000 else []))
    if cls == "tiny":
        H, W = int(rng.integers(1, 40)), int(rng.integers(1, 40))
    elif cls == "small":
        H, W = int(rng.integers(8, 300)), int(rng.integers(8, 300))
    elif cls == "medium":
        H, W = int(rng.integers(200, 1300)), int(rng.integers(200, 1300))
    elif cls == "large":    # (--max-mpix >= 30 only) the long transforms and the block-wise planner: one pass up to 8448, blocks from ~4900
        H, W = int(rng.integers(2300, 6600)), int(rng.integers(2300, 6600))
        if rng.random() < 0.3:
            H = int(rng.integers(300, 2300))
    elif cls == "strip":
        H, W = (int(rng.integers(1, 6)), int(rng.integers(100, 3000)))
        if rng.random() < 0.5:
            H, W = W, H
    else:
        H = W = 0
    F = int(rng.choice([1, 1, 1, 2, 3, 5]))
    if cls == "fast":       # data + kernel - 1 lands on, or a few samples under, a specialised length (in one or both dimensions)
        mk = [int(rng.integers(1, 70)), int(rng.integers(1, 70))]
        dims = []
        for d in range(2):
            if d == 0 or rng.random() < 0.7:
                L = int(rng.choice(FAST_LENGTHS))
                dims.append(max(mk[d], L - mk[d] + 1 - int(rng.integers(0, 20))))
            else:
                dims.append(int(rng.integers(mk[d], 900)))
        H, W = dims
        mkh, mkw = mk
    else:
        mkh = int(rng.integers(1, min(H, 130) + 1))
        mkw = int(rng.integers(1, min(W, 130) + 1))
    fh, fw = util.ceil16(H + mkh - 1), util.ceil16(W + mkw - 1)
    while fh * fw * F > max_pixels and F > 1:
        F -= 1
    n = int(rng.integers(1, 13))
    n = max(1, min(n, int(max_pixels * 4 // (fh * fw * F))))
    if cls == "large":
        F, n = 1, min(n, 2)
    # a ragged cell: runs of equal sizes (the library groups consecutive kernels of one size into one launch set)
    sizes = []
    while len(sizes) < n:
        kh = mkh if rng.random() < 0.6 else int(rng.integers(1, mkh + 1))
        kw = mkw if rng.random() < 0.6 else int(rng.integers(1, mkw + 1))
        sizes += [(kh, kw)] * int(rng.integers(1, 5))
    sizes = sizes[:n]
    entry = str(rng.choice(["one_shot", "plan_host", "plan_host", "plan_device", "two_step"]))
    opts = {"kernel_path": int(rng.choice([0, 0, 0, 1, 2])), "rows_group": int(rng.choice([0, 0, 1, 2, 3, 5]))}
    if cls == "large":
        opts = {"kernel_path": 0, "rows_group": 0, "blockwise": int(rng.random() < 0.25)}
    if rng.random() < 0.2 and min(fh, fw) > 80:
        opts["max_transform"] = int(rng.choice([64, 96, 128, 288, 576]))      # forces a block-wise plan
        if opts["max_transform"] < max(mkh, mkw) + 16:
            del opts["max_transform"]
    runtime = {"batch_maps": int(rng.choice([0, 0, 1, 2, 5])), "flip_kernels": int(rng.random() < 0.2),
               "host_pinned": int(rng.random() < 0.7), "host_min_kb": int(rng.choice([1024, 0])), "host_stream": int(rng.choice([1, 1, 0, 2]))}
    region = int(rng.choice([0, 0, 0, 1, 2, 3]))
    if region == 3 and (H < mkh or W < mkw):
        region = 0
    if cls == "large":
        opts.pop("max_transform", None)
        if not opts["blockwise"]:           # (the planner may choose blocks by itself: regions are cropped on delivery there, round 5)
            if rng.random() < 0.2:
                opts["max_transform"] = int(rng.choice([1152, 2112]))
    elif "max_transform" not in opts and rng.random() < 0.15:
        opts["exact_window"] = 1            # the window's own transform (the BASELINE windows 1088 / 4160 have kernels of their own)
    return dict(cls=str(cls), H=H, W=W, F=F, mkh=mkh, mkw=mkw, sizes=sizes, entry=entry, opts=opts, runtime=runtime, region=region)


def run_case(fc, oracle, torch, case, rng):
    H, W, F, mkh, mkw = case["H"], case["W"], case["F"], case["mkh"], case["mkw"]
    data = rng.standard_normal((H, W, F)).astype(np.float32)
    ks = [rng.standard_normal((kh, kw, F)).astype(np.float32) for kh, kw in case["sizes"]]
    entry, opts, rt, region = case["entry"], dict(case["opts"]), dict(case["runtime"]), case["region"]
    blockwise_forced = "max_transform" in opts
    if opts.get("exact_window"):        # windows with a prime factor above 17 have no transform of their own: the documented refusal, then without
        try:
            with fc.Plan(H, W, F, mkh, mkw, options=opts):
                pass
        except fc.FFTConvError as e:
            if e.status != -5:
                raise
            opts.pop("exact_window")
    if entry in ("one_shot", "two_step"):
        region = 0                      # (the one-shot and two-step entries return the window)
    if entry == "plan_device" and len(set(case["sizes"])) != 1:
        entry = "plan_host"             # the packed device entry takes kernels of one size
    flip = rt["flip_kernels"] and entry in ("plan_host", "plan_device") and not blockwise_forced
    ref_k = [np.ascontiguousarray(k[::-1, ::-1, :]) for k in ks] if flip else ks
    ref = oracle.conv_fft(data, mkh, mkw, ref_k)
    if entry == "one_shot":
        got = fc.cudaConvolutionFFT(data, mkh, mkw, ks, options=opts)
    elif entry == "two_step":
        h = fc.cudaFFTData(data, mkh, mkw)
        try:
            got = fc.cudaConvFFTData(h, ks)
        finally:
            h.destroy() if hasattr(h, "destroy") else None
    else:
        with fc.Plan(H, W, F, mkh, mkw, options=opts) as plan:
            for k, v in rt.items():
                if k == "flip_kernels":
                    v = int(flip)
                plan.set_option(k, v)
            if region:
                plan.set_option("output_region", region)
            if entry == "plan_host":
                plan.set_image(data)
                got = plan.convolve(ks)
                if rng.random() < 0.3:          # the plan again with another image: staging buffers, spectra and rings are reused
                    data = rng.standard_normal((H, W, F)).astype(np.float32)
                    ref = oracle.conv_fft(data, mkh, mkw, ref_k)
                    plan.set_image(data)
                    got = plan.convolve(ks)
            else:
                kh, kw = case["sizes"][0]
                dev = torch.device("cuda", 0)
                img_d = torch.from_numpy(np.ascontiguousarray(np.transpose(data, (2, 1, 0)))).to(dev)
                k_d = torch.from_numpy(np.ascontiguousarray(np.stack([np.transpose(k, (2, 1, 0)) for k in ks]))).to(dev)
                i = plan.info
                oh, ow = (i.out_h, i.out_w) if region else (i.fft_h, i.fft_w)
                out = torch.full((len(ks), ow, oh), float("nan"), dtype=torch.float32, device=dev)
                plan.set_image_device(img_d.data_ptr())
                plan.convolve_packed_device(len(ks), k_d.data_ptr(), kh, kw, out.data_ptr())
                plan.synchronize()
                got = [out[j].cpu().numpy().T for j in range(len(ks))]
    sl = {0: None, 1: (H + mkh - 1, W + mkw - 1, 0, 0), 2: (H, W, (mkh - 1) // 2, (mkw - 1) // 2), 3: (H - mkh + 1, W - mkw + 1, mkh - 1, mkw - 1)}[region]
    worst = 0.0
    for g, r in zip(got, ref):
        if sl is not None:
            oh, ow, fh0, fw0 = sl
            rr = r[fh0:fh0 + oh, fw0:fw0 + ow]
            assert g.shape == rr.shape, (g.shape, rr.shape)
            scale = max(np.abs(r).max(), 1e-30)
            worst = max(worst, float(np.abs(g.astype(np.float64) - rr).max() / scale))
        else:
            assert g.shape == r.shape, (g.shape, r.shape)
            worst = max(worst, util.rel_err(g, r))
    return worst, entry


def run(seed=1, cases=0, seconds=0.0, only=-1, max_mpix=3.0, quiet=False):
    """returns (cases run, indices of the failing ones, worst relative error of the passing ones)"""
    args = argparse.Namespace(seed=seed, cases=cases, seconds=seconds, only=only, max_mpix=max_mpix, quiet=quiet)
    return _run(args)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=0.0)
    ap.add_argument("--cases", type=int, default=0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--only", type=int, default=-1, help="run just this case index of the seed's sequence")
    ap.add_argument("--max-mpix", type=float, default=3.0, help="padded pixels x features per map, millions")
    ap.add_argument("--quiet", action="store_true")
    done, failures, worst = _run(ap.parse_args())
    sys.exit(1 if failures else 0)


def _run(args):
    import torch
    fc = util.load_package()
    oracle = util.Oracle()
    t0 = time.time()
    failures, done, worst_all = [], 0, 0.0
    by_entry = {}
    i = 0
    while True:
        if args.cases and i >= args.cases:
            break
        if args.seconds and time.time() - t0 > args.seconds:
            break
        if not args.cases and not args.seconds and args.only < 0 and i >= 20:
            break
        rng = np.random.default_rng([args.seed, i])
        case = draw_case(rng, int(args.max_mpix * 1e6))
        if args.only >= 0 and i != args.only:
            i += 1
            if i > args.only:
                break
            continue
        try:
            err, entry = run_case(fc, oracle, torch, case, rng)
            ok = err < TOL
            msg = "%.2e" % err
        except fc.FFTConvError as e:
            ok, entry, msg, err = False, case["entry"], "FFTConvError: %s" % e, float("nan")
        except AssertionError as e:
            ok, entry, msg, err = False, case["entry"], "shape mismatch %s" % (e,), float("nan")
        done += 1
        by_entry[entry] = by_entry.get(entry, 0) + 1
        if ok:
            worst_all = max(worst_all, err)
        if not ok:
            failures.append(i)
        if not ok or not args.quiet:
            print("%s case %d (--seed %d --only %d): %s %dx%dx%d maxk %dx%d n %d %s opts %s runtime %s region %d -> %s" % (
                "ok  " if ok else "FAIL", i, args.seed, i, case["cls"], case["H"], case["W"], case["F"], case["mkh"], case["mkw"], len(case["sizes"]),
                entry, case["opts"], case["runtime"], case["region"], msg), flush=True)
        i += 1
    print("# %d cases in %.0f s, %d failed %s; worst relative error of the passing ones %.2e; entries %s" % (
        done, time.time() - t0, len(failures), failures, worst_all, by_entry), flush=True)
    return done, failures, worst_all


if __name__ == "__main__":
    main()
