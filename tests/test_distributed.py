"""The N > 1 paths (filter sharding + single spectrum broadcast; image streaming) through the ONE
orchestration of cuda-fft-convolution_amd/multi_gpu.py, world_size 2:
  CPU tier   gloo, compute by the test-only host emulator of the kernel bodies;
  GPU tier   (-m gpu) two processes sharing the one test GPU, gloo moving the spectrum, HIP plans
             as engines -- the same classes bench.py drives over RCCL with one GPU per rank;
plus the in-library multi-device entry (fftconv_multi_*) with the test GPU listed twice."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import util


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_world(tmp_path, engine, case, world=2, timeout=600):
    port = free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   FC_ROOT=util.ROOT, FC_OUT=str(tmp_path), FC_ENGINE=engine, FC_CASE=",".join(str(v) for v in case),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(util.ROOT, "tests", "dist_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for p in procs:
        out, _ = p.communicate(timeout=timeout)
        assert p.returncode == 0, out
    return [np.load(tmp_path / ("rank%d.npz" % r)) for r in range(world)]


def check_world(oracle, ranks, case, tol):
    """every filter-sharded map of every image and every streamed map against the oracle"""
    import importlib
    mg = importlib.import_module("cuda-fft-convolution_amd.multi_gpu")
    H, W, F, kh, kw, N, NIMG = case
    world = len(ranks)
    _, ks = util.synth(11, H, W, F, kh, kw, N)
    refs = []
    for i in range(NIMG):
        img = np.asfortranarray(np.random.default_rng(1234 + 11 + 1000 * i).random((H, W, F), dtype=np.float32))
        refs.append(oracle.conv_fft(img, kh, kw, ks))
    seen_f = seen_s = 0
    for r, z in enumerate(ranks):
        first, count = mg.filter_shard(N, r, world)
        assert (int(z["first"]), int(z["count"])) == (first, count)
        for i in range(NIMG):
            for j in range(first, first + count):
                assert util.rel_err(z["f_%d_%d" % (i, j)], refs[i][j]) < tol, ("filter-sharded", r, i, j)
                seen_f += 1
        ifirst, icount = mg.image_shard(NIMG, r, world)
        for i in range(ifirst, ifirst + icount):
            for j in range(N):
                assert util.rel_err(z["s_%d_%d" % (i, j)], refs[i][j]) < tol, ("streamed", r, i, j)
                seen_s += 1
    assert seen_f == NIMG * N and seen_s == NIMG * N


def test_filter_shard_partition():
    import importlib
    mg = importlib.import_module("cuda-fft-convolution_amd.multi_gpu")
    for n in (0, 1, 5, 8, 256, 1024, 1023):
        for world in (1, 2, 3, 4, 8):
            blocks = [mg.filter_shard(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0
            for (f0, c0), (f1, c1) in zip(blocks, blocks[1:]):
                assert f0 + c0 == f1
            assert blocks[-1][0] + blocks[-1][1] == n
            counts = [c for _, c in blocks]
            assert max(counts) - min(counts) <= 1
    assert mg.filter_shard(1024, 3, 8) == (384, 128)      # cfg4: 128 filters per GPU
    assert mg.image_shard(32, 7, 8) == (28, 4)            # cfg5: 4 images per GPU
    with pytest.raises(ValueError):
        mg.filter_shard(4, 2, 2)


def test_convolver_refuses_to_overwrite_an_unconsumed_spectrum():
    import importlib
    mg = importlib.import_module("cuda-fft-convolution_amd.multi_gpu")

    class Eng:
        sync = mg.NullSync()

        def new_spectrum(self):
            return [None]

        def compute_spectrum(self, spec, image):
            spec[0] = image

        def convolve(self, spec, first, count):
            return spec[0]

    c = mg.FilterShardedConvolver(Eng(), None, 0, 1, 4, depth=2)
    with pytest.raises(RuntimeError):
        c.convolve()
    c.submit("a"); c.submit("b")
    with pytest.raises(RuntimeError):
        c.submit("c")
    assert c.convolve() == "a" and c.convolve() == "b"
    assert c.run(["x", "y", "z"]) == "z"


def test_world2_gloo_emulator_matches_oracle(tmp_path, oracle):
    subprocess.run(["make", "-C", os.path.join(util.ROOT, "tests", "emu")], check=True,
                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    case = (40, 36, 2, 7, 5, 5, 3)      # 5 filters -> blocks 3 + 2; 3 images -> 2 + 1
    ranks = run_world(tmp_path, "emu", case)
    check_world(oracle, ranks, case, 1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [
    (300, 260, 1, 63, 63, 9, 3),     # cfg4-shaped (63 x 63 kernels, F = 1), 9 filters -> 5 + 4; generic row kernel + fast columns
    (512, 512, 1, 31, 31, 7, 3),     # both hot kernels specialised (576 x 576 transforms), multi-map walk
])
def test_world2_one_gpu_hip_engines_match_oracle(tmp_path, oracle, case):
    """two processes, HIP plans: the filter-sharded step with two spectrum buffers in flight (cfg4's
    form) and the streamed image batch (cfg5's form), every map of every image against the oracle"""
    ranks = run_world(tmp_path, "hip", case)
    check_world(oracle, ranks, case, 1e-5)


@pytest.mark.gpu
def test_in_library_multi_device_plan(fftconv, oracle):
    """fftconv_multi_*: one process, one plan per listed device (the test GPU twice), spectrum copied
    from the first plan to the second, contiguous kernel blocks, ragged kernel sizes"""
    H, W, F, kh, kw, n = 200, 180, 2, 15, 11, 7
    data, ks = util.synth(71, H, W, F, kh, kw, n)
    ks[2] = np.asfortranarray(ks[2][:9, :7, :])
    ks[5] = np.asfortranarray(ks[5][:4, :, :])
    ref = oracle.conv_fft(data, kh, kw, ks)
    for devs in ([0], [0, 0], [0, 0, 0]):
        with fftconv.MultiPlan(H, W, F, kh, kw, devs) as mp:
            assert len(mp) == len(devs)
            blocks = [mp.shard(n, g) for g in range(len(devs))]
            assert blocks[0][0] == 0 and sum(c for _, c in blocks) == n
            mp.set_image(data)
            for rep in range(2):
                got = mp.convolve(ks)
                for g, r in zip(got, ref):
                    assert util.rel_err(g, r) < 1e-5
        got = fftconv.cudaConvolutionFFTMulti(data, kh, kw, ks, devs)
        for g, r in zip(got, ref):
            assert util.rel_err(g, r) < 1e-5
    with pytest.raises(fftconv.FFTConvError):
        fftconv.MultiPlan(H, W, F, kh, kw, [0, 99])
    with pytest.raises(fftconv.FFTConvError):
        fftconv.MultiPlan(H, W, F, kh, kw, [])


@pytest.mark.gpu
@pytest.mark.parametrize("args", [["--config", "cfg1", "--force-collective"], ["--config", "cfg2", "--force-collective", "--no-overlap"],
                                  ["--config", "cfg1", "--images", "3"], ["--config", "cfg2", "--graph"]])
def test_bench_steps_self_check(args):
    """bench.py's own steps (the shared orchestration, broadcast path forced on one rank, streamed
    mode, graph replay) pass the bench's self-check: a non-zero exit would mean wrong maps"""
    import json
    r = subprocess.run([sys.executable, os.path.join(util.ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--check",
                        "--no-cpu-baseline"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert j["check_ok"] and j["check_max_rel_err"] < 1e-4 and j["check_checksum_max_rel_err"] < 1e-5
