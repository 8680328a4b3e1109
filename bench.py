#!/usr/bin/env python3
"""bench.py -- headline benchmark of the FFT-convolution hot path on MI355X.

A "step" is one pass of the hot path over one batch of synthetic input: the image (already
resident in HBM) is zero-padded and transformed once, then every filter of the batch is
convolved with it, all maps staying device-resident (SURVEY.md 8(d) "timed region").

Workload at N = 1 (default): BASELINE.json configs[2], the configuration the metric is quoted
on -- 4096x4096 fp32 image, 256 kernels of 127x127, F = 1 -> 256 maps of 4224x4224.
N > 1: the filters are sharded over the ranks (one process per GPU): every rank owns
`--filters` kernels of the same image; rank 0 transforms the image and its spectrum is
broadcast once over RCCL (torch.distributed "nccl"), the only collective on the path.  Per-GPU
work is fixed, so "scaling" is "weak"; value is the whole-job rate (all ranks' maps / max time).

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (dominant
kernel, algorithmic bytes / live HIP-event time) and `cpu_baseline` (the CPU oracle timed on
this host on a bounded sample, rank 0 at N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CONFIGS = {
    # name: (H, W, F, kh, kw, filters per GPU, seed)
    "cfg1": (256, 256, 1, 31, 31, 1, 1),
    "cfg2": (1024, 1024, 1, 63, 63, 16, 2),
    "cfg3": (4096, 4096, 1, 127, 127, 256, 3),
    "cfg4": (4096, 4096, 1, 63, 63, 128, 4),   # 1024 kernels over 8 GPUs
    "cfg5": (2048, 2048, 1, 63, 63, 64, 5),
    # not a BASELINE config: the multi-feature form of the reference's demo (F planes summed per map)
    "cfg3f4": (4096, 4096, 4, 127, 127, 64, 6),
    "mid512": (512, 512, 1, 31, 31, 256, 7),   # not BASELINE configs: mid-sized images on the 576 / 768 / 1536 / 3072 transforms
    "hd720": (720, 1280, 1, 31, 31, 128, 8),
    "mid2900": (2900, 2900, 1, 63, 63, 64, 9),
    "big8192": (8192, 8192, 1, 127, 127, 32, 10),   # 8448 x 8448 transforms
    "big6000": (6000, 6000, 1, 63, 63, 32, 11),     # 6144 x 6144 transforms
}
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


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


def cpu_baseline(cfg, sample_filters):
    """Times the CPU oracle (port of demoCudaConvolutionFFT.m:78-102, complex128 full transforms)
    on this host on a bounded sample of the same workload."""
    import util
    H, W, F, kh, kw, _, seed = CONFIGS[cfg]
    orc = util.Oracle()
    cores = orc.num_threads(0)
    n = max(1, min(sample_filters, cores))
    img, ks = util.synth(seed, H, W, F, kh, kw, n)
    t0 = time.perf_counter()
    orc.conv_fft(img, kh, kw, ks, threads=0)
    dt = time.perf_counter() - t0
    P = ceil16(H + kh - 1) * ceil16(W + kw - 1)
    res = {"value": n * P / dt / 1e9, "unit": "Gpixel-filters/s", "cores": min(cores, n),
           "kind": "port",
           "sample": "%s image + %d of its filters (float64 fft2/ifft2 oracle, OpenMP over filters), %.1f s wall"
                     % (cfg, n, dt)}
    # sanity number beside the port: the same maths on a production CPU FFT -- SciPy's
    # pocketfft, float32 rfft2 / irfft2, all host cores -- the closest stand-in for MATLAB's
    # multithreaded fft2 / ifft2 on this host (image spectrum computed once, as the GPU path does)
    try:
        import numpy as np
        import scipy.fft as sfft
        workers = os.cpu_count() or 1
        fh, fw = ceil16(H + kh - 1), ceil16(W + kw - 1)
        t0 = time.perf_counter()
        D = sfft.rfft2(img, s=(fh, fw), axes=(0, 1), workers=workers)
        m = min(n, 4)
        for k in ks[:m]:
            K = sfft.rfft2(k, s=(fh, fw), axes=(0, 1), workers=workers)
            out = sfft.irfft2(D * K, s=(fh, fw), axes=(0, 1), workers=workers).sum(axis=2)
        dt2 = time.perf_counter() - t0
        res["scipy_pocketfft_f32"] = {"value": m * P / dt2 / 1e9, "unit": "Gpixel-filters/s", "cores": workers,
                                      "sample": "%d filters, %.1f s wall" % (m, dt2)}
        del D, K, out
    except Exception as e:   # the sanity number is optional
        res["scipy_pocketfft_f32"] = {"error": str(e)}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="cfg3", choices=sorted(CONFIGS))
    ap.add_argument("--filters", type=int, default=0, help="filters per GPU (0 = the config's)")
    ap.add_argument("--batch-maps", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=8)
    ap.add_argument("--check", action="store_true", help="verify a few maps against the oracle after timing")
    ap.add_argument("--images", type=int, default=0,
                    help="streamed mode (BASELINE configs[4]): each rank convolves this many images per step, every "
                         "image with its own kernels' maps, the next image's H2D copy (pinned host memory, side "
                         "stream) overlapped with the current image's compute; no collective")
    ap.add_argument("--graph", action="store_true",
                    help="record one step into a HIP graph (plan bound to the capturing stream) and time graph "
                         "replays: removes the per-launch host cost that bounds the small configurations; "
                         "single-GPU, non-streamed steps only")
    ap.add_argument("--no-overlap", action="store_true", help="blocking broadcast, no kernel-column overlap (A/B)")
    ap.add_argument("--force-collective", action="store_true",
                    help="run the broadcast code path even with one rank (self-test of the N > 1 step on a 1-GPU box)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import util

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_collective
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    fc = util.load_package()
    H, W, F, kh, kw, nf, seed = CONFIGS[args.config]
    if args.filters > 0:
        nf = args.filters
    img_h, _ = util.synth(seed, H, W, F, kh, kw, 0)
    # kernels: seed 5678+cfg+k with k the GLOBAL filter index (rank-sharded contiguous blocks)
    import numpy as np
    kern_h = np.empty((nf, F, kw, kh), dtype=np.float32)  # [n][f][kw][kh] == n MATLAB arrays kh x kw x F
    for j in range(nf):
        k = np.random.default_rng(5678 + seed + rank * nf + j).random((kh, kw, F), dtype=np.float32)
        kern_h[j] = np.transpose(k, (2, 1, 0))
    img_d = torch.from_numpy(np.ascontiguousarray(np.transpose(img_h, (2, 1, 0)))).to(dev)  # [f][w][h]
    kern_d = torch.from_numpy(kern_h).to(dev)

    stream = torch.cuda.current_stream(dev)
    plan = fc.Plan(H, W, F, kh, kw, gpuId=local_rank, stream=stream.cuda_stream)
    info = plan.info
    P = info.fft_h * info.fft_w
    spec = torch.empty(info.spectrum_bytes, dtype=torch.uint8, device=dev)
    plan.use_spectrum_buffer(spec.data_ptr(), spec.numel())
    out = torch.empty((nf, info.fft_w, info.fft_h), dtype=torch.float32, device=dev)
    if args.batch_maps:
        plan.set_option("batch_maps", args.batch_maps)

    streamed = args.images > 0
    if streamed:
        # images of this rank in pinned host memory (seeded per global image index), two device
        # buffers, a side stream for the copies
        n_img = args.images
        imgs_h = []
        for i in range(n_img):
            a = np.random.default_rng(1234 + seed + 1000 * (rank * n_img + i)).random((F, W, H), dtype=np.float32)
            imgs_h.append(torch.from_numpy(a).pin_memory())
        img_buf = [torch.empty((F, W, H), dtype=torch.float32, device=dev) for _ in range(2)]
        copy_stream = torch.cuda.Stream(dev)
        copied = [torch.cuda.Event() for _ in range(2)]
        consumed = [torch.cuda.Event() for _ in range(2)]

    def step_streamed():
        main = torch.cuda.current_stream(dev)
        with torch.cuda.stream(copy_stream):
            img_buf[0].copy_(imgs_h[0], non_blocking=True)
            copied[0].record(copy_stream)
        for i in range(n_img):
            b = i & 1
            if i + 1 < n_img:            # next image's H2D while this one is convolved
                with torch.cuda.stream(copy_stream):
                    if i >= 1:
                        copy_stream.wait_event(consumed[1 - b])   # its buffer was read by image i-1's FFT
                    img_buf[1 - b].copy_(imgs_h[i + 1], non_blocking=True)
                    copied[1 - b].record(copy_stream)
            main.wait_event(copied[b])
            plan.set_image_device(img_buf[b].data_ptr())
            consumed[b].record(main)
            plan.convolve_packed_device(nf, kern_d.data_ptr(), kh, kw, out.data_ptr())

    def step():
        if streamed:
            return step_streamed()
        if rank == 0:
            plan.set_image_device(img_d.data_ptr())
        if use_dist and not streamed:
            # the kernels' column transforms do not need the image: they run while the spectrum
            # travels (rank 0: after its image pass; the others: from the start of the step)
            if args.no_overlap:
                dist.broadcast(spec, src=0)
            else:
                work = dist.broadcast(spec, src=0, async_op=True)
                plan.prepare_kernels_packed_device(nf, kern_d.data_ptr(), kh, kw)
                work.wait()        # the plan's stream waits for the broadcast, the host does not
        plan.mark_spectrum_valid()
        plan.convolve_packed_device(nf, kern_d.data_ptr(), kh, kw, out.data_ptr())

    def barrier():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    run_step = step
    graph = None
    if args.graph:
        if use_dist or streamed:
            raise SystemExit("--graph: single-GPU, non-streamed steps only")
        step()                                   # scratch buffers are sized: nothing allocates from here on
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        cap_stream = torch.cuda.Stream(dev)
        with torch.cuda.graph(graph, stream=cap_stream):
            plan.set_stream(torch.cuda.current_stream(dev).cuda_stream)
            step()
        plan.set_stream(stream.cuda_stream)
        run_step = graph.replay
        for _ in range(max(1, args.warmup)):
            run_step()
        barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step()
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # per-kernel HIP-event timing of the same steps, on the plan's stream (separate pass so the
    # event records do not sit inside the headline timing)
    plan.set_option("profile", 1)
    plan.profile(reset=True)
    for _ in range(max(1, min(args.steps, 3))):
        step()
    torch.cuda.synchronize(dev)
    prof = plan.profile(reset=True)
    plan.set_option("profile", 0)

    result = None
    if rank == 0:
        ab = alg_bytes(H, W, F, kh, kw)
        per = {}
        for name in ("spectral_rows", "cols_c2r"):
            p = prof[name]
            if p["launches"]:
                avg_ms = p["ms"] / p["launches"]
                units = p["units"] / p["launches"]
                per[name] = {"avg_ms": avg_ms, "units_per_launch": units,
                             "gbps": ab[name] * units / (avg_ms * 1e-3) / 1e9}
        dom = max(per, key=lambda k: prof[k]["ms"]) if per else None
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if dom and os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                per_map = tj.get(args.config, {}).get(dom)
                traffic = per_map * per[dom]["units_per_launch"] if per_map else None
            except Exception:
                traffic = None
        total_maps = nf * world * (args.images if streamed else 1)
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
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%s: %dx%d image (F=%d), %d kernels of %dx%d per GPU -> %d maps of %dx%d, %s"
                                   % (args.config, H, W, F, nf, kh, kw, total_maps, info.fft_h, info.fft_w,
                                      "image-sharded" if streamed else "filter-sharded"),
                       "transform": [info.transform_h, info.transform_w],
                       "filters_per_gpu": nf,
                       "hip_graph_replay": bool(args.graph),
                       "images_per_gpu_per_step": args.images if streamed else 1,
                       "parallelism": ("images x%d, streamed H2D" % world) if streamed else
                                      ("filters x%d + 1 bcast" % world if world > 1 else "single GPU")},
            "hbm_algorithmic_gbps": ab["total"] * total_maps * args.steps / dt / 1e9,
            "hbm_frac_of_peak": ab["total"] * total_maps * args.steps / dt / 1e9 / (HBM_PEAK_GBPS * world),
            "kernels": per,
            "image_ms": (prof["image_cols"]["ms"] + prof["image_rows"]["ms"]) / max(1, prof["image_cols"]["launches"]),
        }
        if dom:
            result["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": per[dom]["gbps"], "peak": HBM_PEAK_GBPS,
                                  "unit": "GB/s", "frac": per[dom]["gbps"] / HBM_PEAK_GBPS, "traffic": traffic,
                                  "algorithmic_bytes_per_launch": ab[dom] * per[dom]["units_per_launch"],
                                  "avg_launch_ms": per[dom]["avg_ms"]}
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args.config, args.cpu_sample)
        if args.check:
            orc = util.Oracle()
            idx = sorted(set([0, nf // 2, nf - 1]))
            ks = [np.asfortranarray(np.transpose(kern_h[j], (2, 1, 0))) for j in idx]
            img_chk = img_h
            if streamed:   # the maps in `out` belong to the last image of the step
                img_chk = np.asfortranarray(np.transpose(imgs_h[-1].numpy(), (2, 1, 0)))
            ref = orc.conv_fft(img_chk, kh, kw, ks)
            errs = []
            for j, r in zip(idx, ref):
                g = out[j].cpu().numpy().T  # [w][h] -> h x w
                errs.append(util.rel_err(g, r))
            result["check_max_rel_err"] = max(errs)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    plan.destroy()
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


if __name__ == "__main__":
    main()
