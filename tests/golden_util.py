import glob
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_cases():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "case_*.npz")))


def load_case(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    n = len([k for k in z.files if k.startswith("kernel_")])
    kernels = [z["kernel_%d" % i] for i in range(n)]
    expect = [z["expect_%d" % i] for i in range(n)]
    return z["data"], int(z["maxk"][0]), int(z["maxk"][1]), kernels, expect
