"""One rank of the multi-process tests (started by tests/test_distributed.py): runs the shared
orchestration of cuda-fft-convolution_amd/multi_gpu.py over torch.distributed and writes this
rank's maps to FC_OUT/rank<r>.npz.

FC_ENGINE = emu   host emulator of the kernel bodies as engine (CPU tier, gloo)
          = hip   HIP plans as engines, every rank on cuda:0 (GPU tier: two processes share the one
                  test GPU, gloo moves the spectrum)
          = hip_nccl  HIP plans, rank r on cuda:r, the broadcast over RCCL ("nccl"): needs >= world GPUs
FC_CASE   = H,W,F,kh,kw,N,n_images   problem (seeded like util.synth)
"""
import ctypes
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.environ["FC_ROOT"]
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import util  # noqa: E402

fc = util.load_package()
mg = importlib.import_module(fc.__name__ + ".multi_gpu")
rank = int(os.environ["RANK"])
world = int(os.environ["WORLD_SIZE"])
H, W, F, kh, kw, N, NIMG = [int(v) for v in os.environ["FC_CASE"].split(",")]
SEED = 11
fh, fw = util.ceil16(H + kh - 1), util.ceil16(W + kw - 1)


def image(i):
    """i-th image of the case, H x W x F (Fortran order)"""
    return np.asfortranarray(np.random.default_rng(1234 + SEED + 1000 * i).random((H, W, F), dtype=np.float32))


_, KS = util.synth(SEED, H, W, F, kh, kw, N)


class EmuEngine:
    """engine protocol of multi_gpu.py on the host emulator (tests/emu)"""

    def __init__(self):
        self.emu = ctypes.CDLL(os.path.join(ROOT, "tests", "emu", "libfftconv_emu.so"))
        self.emu.emu_spectrum_elems.restype = ctypes.c_long
        self.nspec = self.emu.emu_spectrum_elems(H, W, F, kh, kw)
        self.sync = mg.NullSync()

    def new_spectrum(self):
        return torch.full((2 * self.nspec,), float("nan"), dtype=torch.float32)   # only rank src ever computes it

    def new_image_buffer(self):
        return np.zeros((H, W, F), dtype=np.float32, order="F")

    def upload(self, buf, host_image):
        buf[...] = host_image

    def compute_spectrum(self, spec, img):
        img = np.asfortranarray(img)
        assert self.emu.emu_image_spectrum(ctypes.c_void_p(img.ctypes.data), H, W, F, kh, kw, ctypes.c_void_p(spec.data_ptr())) == 0

    def convolve(self, spec, first, count):
        outs = [np.zeros((fh, fw), np.float32, order="F") for _ in range(count)]
        if count:
            kp = (ctypes.c_void_p * count)(*[KS[first + j].ctypes.data for j in range(count)])
            khs = (ctypes.c_int * count)(*[kh] * count)
            kws = (ctypes.c_int * count)(*[kw] * count)
            op = (ctypes.c_void_p * count)(*[o.ctypes.data for o in outs])
            assert self.emu.emu_convolve_spectrum(ctypes.c_void_p(spec.data_ptr()), H, W, F, kh, kw, count, kp, khs, kws, op) == 0
        return outs


def to_host_maps(res, count):
    if isinstance(res, list):
        return [np.array(m) for m in res]
    return [res[j].cpu().numpy().T.copy() for j in range(count)]   # device [w][h] -> h x w


def main():
    engine_kind = os.environ.get("FC_ENGINE", "emu")
    gpu = 0
    if engine_kind == "hip_nccl":      # one GPU per rank, RCCL: what bench.py does on a multi-GPU node
        gpu = rank
        torch.cuda.set_device(gpu)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", gpu))
        engine_kind = "hip"
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    saved = {}

    # ---- filter sharding: NIMG distinct images one after the other (pipelined: two spectrum buffers)
    first, count = mg.filter_shard(N, rank, world)
    if engine_kind == "hip":
        dev = torch.device("cuda", gpu)
        torch.cuda.set_device(dev)
        stream = torch.cuda.current_stream(dev)
        plan = fc.Plan(H, W, F, kh, kw, gpuId=gpu, stream=stream.cuda_stream)
        kern = np.stack([np.transpose(KS[first + j], (2, 1, 0)) for j in range(count)]) if count else np.zeros((0, F, kw, kh), np.float32)
        kern_d = torch.from_numpy(np.ascontiguousarray(kern)).to(dev)
        engine = mg.HipPlanEngine(torch, fc, plan, dev, kern_d, kh, kw, first=first, main_stream=stream)
        imgs = [torch.from_numpy(np.ascontiguousarray(np.transpose(image(i), (2, 1, 0)))).to(dev) for i in range(NIMG)]
    else:
        engine = EmuEngine()
        imgs = [image(i) for i in range(NIMG)]
    conv = mg.FilterShardedConvolver(engine, dist, rank, world, N, src=0, depth=2)

    def keep(k, res):
        for j, m in enumerate(to_host_maps(res, count)):
            saved["f_%d_%d" % (k, first + j)] = m

    conv.run(imgs if rank == 0 else [None] * NIMG, on_result=keep)

    # ---- image streaming: the images dealt over the ranks, every image against all N kernels
    ifirst, icount = mg.image_shard(NIMG, rank, world)
    if engine_kind == "hip":
        kern_all = torch.from_numpy(np.ascontiguousarray(np.stack([np.transpose(k, (2, 1, 0)) for k in KS]))).to(dev)
        engine2 = mg.HipPlanEngine(torch, fc, plan, dev, kern_all, kh, kw, first=0, main_stream=stream)
        host = [torch.from_numpy(np.ascontiguousarray(np.transpose(image(ifirst + i), (2, 1, 0)))).pin_memory() for i in range(icount)]
    else:
        engine2 = EmuEngine()
        host = [image(ifirst + i) for i in range(icount)]
    sconv = mg.ImageStreamedConvolver(engine2, N)

    def keep2(i, res):
        for j, m in enumerate(to_host_maps(res, N)):
            saved["s_%d_%d" % (ifirst + i, j)] = m

    for rep in range(2):        # twice: the buffers are reused across run() calls (write-after-read across steps)
        sconv.run(host, on_result=keep2)

    np.savez(os.path.join(os.environ["FC_OUT"], "rank%d.npz" % rank), first=first, count=count, **saved)
    dist.barrier()
    dist.destroy_process_group()
    if engine_kind == "hip":
        plan.destroy()


if __name__ == "__main__":
    main()
