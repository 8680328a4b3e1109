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


# demoCudaConvolutionFFT.m:57-61 plants kernel(:,:,1) in the data -- data(5:(4+cn), 2:(1+cm), 1),
# data(21:(20+cn), 1:cm, 2), data(1:cn, (m-(cm-1)):m, k) -- sets kernel(:,:,k) = kernel(:,:,1) and :63-69
# flips every kernel, so the call correlates: a template planted at 0-based (r, c) answers at
# (r + cn - 1, c + cm - 1) of the full window.  The only numeric structure the reference holds.
DEMO_PLANTED = {0: (4, 1), 1: (20, 0), 4: (0, 4)}   # channel (0-based) -> planted offset (0-based)


def demo_planted_checks(conv, data, cn, cm, ks, tol):
    """conv(kernels) -> maps.  (1) cvcell{1}: its two largest responses sit where channels 1 and k hold
    the template their own kernel channel matches; (2) probing channel c alone with the flipped
    template gives its global maximum, sum(template^2) = sum_{v=1..cn*cm} v^2 exactly, at the
    planted offset of every one of the three plantings."""
    m = np.asarray(conv([ks[0]])[0], dtype=np.float64)
    peak = {c: (r + cn - 1, q + cm - 1) for c, (r, q) in DEMO_PLANTED.items()}
    top2 = {tuple(int(v) for v in np.unravel_index(i, m.shape)) for i in np.argsort(m.ravel())[::-1][:2]}
    assert top2 == {peak[0], peak[4]}
    energy = float(sum(v * v for v in range(1, cn * cm + 1)))   # 22140 for the demo's 10 x 4 template
    for c, p in peak.items():
        probe = np.zeros_like(ks[0])
        probe[:, :, c] = ks[0][:, :, 0]            # flipped template in channel c only
        mm = np.asarray(conv([probe])[0], dtype=np.float64)
        assert tuple(int(v) for v in np.unravel_index(np.argmax(mm), mm.shape)) == p
        assert abs(mm[p] - energy) <= tol * energy
        nb = mm[max(0, p[0] - 1):p[0] + 2, max(0, p[1] - 1):p[1] + 2]
        assert (nb < mm[p]).sum() == nb.size - 1   # strict local maximum
