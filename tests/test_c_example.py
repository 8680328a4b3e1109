"""examples/demo_planted_template.c: a plain C99 program over include/fftconv.h and libfftconv.so --
the reference's demo problem (demoCudaConvolutionFFT.m:37-69,124-129) through the one-shot entry and
through the multi-device entry.  CPU tier: it compiles and links against the public header and the
library alone and fails loudly without a GPU; GPU tier: it finds the planted template."""
import os
import shutil
import subprocess

import pytest

import util

SRC = os.path.join(util.ROOT, "examples", "demo_planted_template.c")
PKG = os.path.join(util.ROOT, "cuda-fft-convolution_amd")


@pytest.fixture(scope="module")
def demo_binary(tmp_path_factory):
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    util.load_package().load_library()      # the library exists (built by __graft_entry__.build())
    exe = str(tmp_path_factory.mktemp("cdemo") / "demo")
    rocm_lib = "/opt/rocm/lib"
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(util.ROOT, "include"), SRC,
           "-L", PKG, "-lfftconv", "-Wl,-rpath," + PKG, "-Wl,-rpath-link," + rocm_lib, "-lm", "-o", exe]
    subprocess.run(cmd, check=True)
    return exe


def test_c_example_builds_and_fails_loudly_without_gpu(demo_binary, fftconv):
    if fftconv.device_count() > 0:
        pytest.skip("a GPU is present")
    r = subprocess.run([demo_binary], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 2 and "fftconv status -6" in r.stderr      # no CPU fallback behind the C ABI either


@pytest.mark.gpu
@pytest.mark.parametrize("nplans", [1, 2])
def test_c_example_finds_the_planted_template(demo_binary, nplans):
    r = subprocess.run([demo_binary, str(nplans)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().endswith("OK") and "at (13, 4)" in r.stdout


@pytest.mark.gpu
def test_python_port_of_the_reference_demo():
    """examples/demo_cuda_convolution_fft.py: demoCudaConvolutionFFT.m step by step (set-up, conv2 and
    fft2/ifft2 CPU references, the three-kernel call) with the figures' residuals as numbers"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("demo_ccf", os.path.join(util.ROOT, "examples", "demo_cuda_convolution_fft.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for seed in (0, 1):
        ok, res = mod.main(seed=seed)
        assert ok, res
