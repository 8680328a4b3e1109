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


def test_streamed_convolver_remembers_the_buffer_of_the_last_image():
    """bench.py checks the maps of the LAST image a run() convolved: with an odd number of images per call and an
    even number of calls the device buffer holding it alternates (the self-check of round 2 read the wrong one)"""
    import importlib
    mg = importlib.import_module("cuda-fft-convolution_amd.multi_gpu")

    class Eng:
        sync = mg.NullSync()

        def __init__(self):
            self.n = 0

        def new_spectrum(self):
            return [None]

        def new_image_buffer(self):
            self.n += 1
            return {"id": self.n, "img": None}

        def upload(self, buf, host_image):
            buf["img"] = host_image

        def compute_spectrum(self, spec, image):
            spec[0] = image["img"]

        def convolve(self, spec, first, count):
            return spec[0]

    for n_img, calls in ((3, 2), (3, 3), (4, 2), (1, 5), (5, 4)):
        c = mg.ImageStreamedConvolver(Eng(), 2)
        for call in range(calls):
            imgs = ["c%d_i%d" % (call, i) for i in range(n_img)]
            seen = []
            last = c.run(imgs, on_result=lambda i, r: seen.append(r))
            assert seen == imgs and last == imgs[-1]                      # every image convolved, in order
            assert c.buf[c.last_buf]["img"] == imgs[-1], (n_img, calls)   # ... and last_buf names the buffer that holds the last one
        assert c.n_run == n_img * calls


@pytest.mark.parametrize("world", [2, 3])
def test_world_gloo_emulator_matches_oracle(tmp_path, oracle, world):
    subprocess.run(["make", "-C", os.path.join(util.ROOT, "tests", "emu")], check=True,
                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    case = (40, 36, 2, 7, 5, 5, 3)      # 5 filters -> blocks 3 + 2 (2 + 2 + 1 on three ranks); 3 images -> 2 + 1 (1 + 1 + 1)
    ranks = run_world(tmp_path, "emu", case, world=world)
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


def _bench(args, timeout=900):
    import json
    r = subprocess.run([sys.executable, os.path.join(util.ROOT, "bench.py")] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
def test_bench_streamed_check_reads_the_last_image_odd_images_even_steps():
    """(advisor, round 2) 3 images per rank and an even step count: the last image of the timed region sits in the
    OTHER device buffer than after one step; the self-check has to follow it"""
    for steps in ("2", "3"):
        j = _bench(["--config", "cfg1", "--images", "3", "--steps", steps, "--warmup", "1", "--check", "--no-cpu-baseline"])
        assert j["check_ok"] and j["check_max_rel_err"] < 1e-4 and j["check_checksum_max_rel_err"] < 1e-5


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu_cfg4_full_size():
    """Rehearsal of what the driver runs at N > 1, at FULL cfg4 size on the one test GPU: bench.py starts its two ranks
    itself, both on cuda:0 (--share-gpu: gloo moves the spectrum), each rank tunes its placement, runs the
    filter-sharded step with the broadcast and two spectrum buffers (64 + 64 kernels of 63 x 63 on the 4096 x 4096
    image: 4160 windows cropped from 4224 transforms), the timing / check reductions run over the ranks, and three
    maps of every rank are compared with the oracle"""
    j = _bench(["--gpus", "2", "--share-gpu", "--config", "cfg4", "--filters", "128", "--steps", "3", "--warmup", "1", "--check",
                "--no-cpu-baseline"], timeout=1200)
    assert j["n_gpus"] == 2 and j["check_ok"]
    assert j["check_max_rel_err"] < 1e-4 and j["check_checksum_max_rel_err"] < 1e-5
    assert j["config"]["filters_total"] == 128 and j["config"]["filters_per_gpu"] == 64
    # round 4: the N > 1 line proves who took part and where its time went
    assert len(j["ranks"]) == 2 and all(r["name"] and r["host"] for r in j["ranks"])
    assert j["distinct_devices"] == 1                      # the rehearsal: both ranks on the one GPU (--share-gpu allows it)
    pr = j["per_rank"]
    assert [r["rank"] for r in pr] == [0, 1] and all(r["filters"] == 64 and r["ms_per_step"] > 0 for r in pr)
    assert all(r["broadcasts"] >= 3 and r["broadcast_ms_mean"] > 0 for r in pr)      # one broadcast per timed step, timed on every rank
    assert max(r["ms_per_step"] for r in pr) <= j["ms_per_step"] * 1.05 + 0.5
    assert "uploaded from pinned host memory inside every step" in j["config"]["kernels"]
    assert "gloo" in j["config"]["backend"]


@pytest.mark.gpu
def test_bench_four_ranks_on_one_gpu_uneven_shards():
    """four ranks on the one GPU (the box allows six processes on the card), 18 filters -> blocks of 5, 5, 4, 4: the
    uneven split, the reductions over more than two ranks and the teardown of four children"""
    j = _bench(["--gpus", "4", "--share-gpu", "--config", "cfg2", "--filters", "18", "--steps", "3", "--warmup", "1", "--check",
                "--no-cpu-baseline"], timeout=900)
    assert j["n_gpus"] == 4 and j["check_ok"] and j["config"]["filters_total"] == 18
    assert j["check_max_rel_err"] < 1e-4 and j["check_checksum_max_rel_err"] < 1e-5


@pytest.mark.gpu
def test_bench_streamed_two_ranks_on_one_gpu_cfg5_full_size():
    """cfg5's form at full size with two ranks on the one GPU: 2048 x 2048 images streamed (3 per rank over 2 ranks ->
    6 in total, odd per rank), 64 kernels of 63 x 63 each, no collective on the data path"""
    j = _bench(["--gpus", "2", "--share-gpu", "--config", "cfg5", "--images", "6", "--steps", "2", "--warmup", "1", "--check",
                "--no-cpu-baseline"], timeout=1200)
    assert j["n_gpus"] == 2 and j["check_ok"] and j["scaling"] == "weak"
    assert j["check_max_rel_err"] < 1e-4 and j["check_checksum_max_rel_err"] < 1e-5
    # per rank: its images, the rate of its H2D copies (an event pair around every image's copy), where its pinned images live
    pr = j["per_rank"]
    assert [r["images"] for r in pr] == [3, 3] and all(r["h2d_gbps"] and r["h2d_gbps"] > 1.0 for r in pr)
    assert all("policy" in r["pinned_images_numa"] for r in pr) and "pinned_images_numa" in j["config"]


def _gpu_count():
    try:
        return util.load_package().device_count()
    except Exception:
        return 0


needs_two_gpus = pytest.mark.skipif(_gpu_count() < 2, reason="needs at least two GPUs (the test box has one; the driver's scaling node has eight)")


@pytest.mark.gpu
@needs_two_gpus
@pytest.mark.parametrize("case", [(300, 260, 1, 63, 63, 9, 3), (512, 512, 1, 31, 31, 7, 3)])
def test_world_nccl_one_gpu_per_rank_matches_oracle(tmp_path, oracle, case):
    """the same two convolvers over RCCL ("nccl"), one GPU per rank, as bench.py runs them on a multi-GPU node:
    every map of every image against the oracle"""
    world = min(_gpu_count(), 4)
    ranks = run_world(tmp_path, "hip_nccl", case, world=world)
    check_world(oracle, ranks, case, 1e-5)


@pytest.mark.gpu
@needs_two_gpus
def test_bench_nccl_two_gpus_cfg4_share():
    """bench.py --gpus 2 over RCCL on two GPUs: cfg4's sharded form at full size, checked"""
    j = _bench(["--gpus", "2", "--config", "cfg4", "--filters", "128", "--steps", "3", "--warmup", "1", "--check", "--no-cpu-baseline"],
               timeout=1200)
    assert j["n_gpus"] == 2 and j["check_ok"] and "nccl" in j["config"]["backend"]


@pytest.mark.gpu
@needs_two_gpus
@pytest.mark.parametrize("transport", [0, 1])
def test_in_library_multi_device_plan_on_distinct_devices(fftconv, oracle, transport):
    """fftconv_multi_* over DISTINCT devices: the spectrum really crosses xGMI -- by hipMemcpyPeerAsync (transport 0,
    src/cudaConvFFTDataStreams.cu:282-287) or by one ncclBroadcast (transport 1, north_star's collective)"""
    H, W, F, kh, kw, n = 200, 180, 2, 15, 11, 7
    data, ks = util.synth(71, H, W, F, kh, kw, n)
    ref = oracle.conv_fft(data, kh, kw, ks)
    devs = list(range(min(_gpu_count(), 4)))
    with fftconv.MultiPlan(H, W, F, kh, kw, devs) as mp:
        assert mp.get_option("peer_direct") == len(devs) - 1, mp.warning
        mp.set_option("spectrum_transport", transport)
        for rep in range(2):
            mp.set_image(data)
            assert mp.get_option("transport_used") == transport, mp.last_message()
            for g, r in zip(mp.convolve(ks), ref):
                assert util.rel_err(g, r) < 1e-5


@pytest.mark.gpu
def test_multi_plan_options_and_rccl_fallback_on_one_gpu(fftconv, oracle):
    """one GPU listed twice cannot form an RCCL communicator: the broadcast transport falls back to the copies, says
    why, and the maps are right; peer_direct counts same-device destinations as direct"""
    H, W, F, kh, kw, n = 120, 100, 1, 9, 9, 4
    data, ks = util.synth(72, H, W, F, kh, kw, n)
    ref = oracle.conv_fft(data, kh, kw, ks)
    with fftconv.MultiPlan(H, W, F, kh, kw, [0, 0]) as mp:
        assert mp.warning == "" and mp.get_option("peer_direct") == 1
        mp.set_option("spectrum_transport", 1)
        mp.set_image(data)
        assert mp.get_option("transport_used") == 0 and "listed twice" in mp.last_message()
        for g, r in zip(mp.convolve(ks), ref):
            assert util.rel_err(g, r) < 1e-5
        with pytest.raises(fftconv.FFTConvError):
            mp.set_option("spectrum_transport", 2)
        with pytest.raises(fftconv.FFTConvError):
            mp.get_option("no_such_option")
