#!/usr/bin/env python3
"""Rewrites csrc/fast_paths.hpp in place with alternative row / column configurations (a configuration search: build the result
with tools/build_variant.sh, measure with tools/profile_shape.py under FFTCONV_LIB, then `git checkout` the header).
usage: config_variant.py rows:L=R1.R2.R3.NT.RPW[,...] cols:M=R1.R2.R3.T.NT[,...]"""
import os, re, sys
p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cuda-fft-convolution_amd", "csrc", "fast_paths.hpp")
s = open(p).read()
for arg in sys.argv[1:]:
    kind, spec = arg.split(":")
    for item in spec.split(","):
        key, val = item.split("=")
        v = [int(x) for x in val.split(".")]
        if kind == "rows":
            L = int(key); r1, r2, r3, nt, rpw = v
            assert r1 * r2 * r3 == L and rpw * r1 * r2 <= nt and r3 % 2 == 0, item
            pat = re.compile(r"(    X\(%d, [^\n]*\\\n)+" % L)
            m = pat.search(s); assert m, item
            # NZ2 variants: what 31 / 63 / 127-wide kernels need at this R3 (kernel width <= NZ2 * R3), and the unpruned form
            nz = sorted(set([min(r2, max(1, -(-k // r3))) for k in (31, 63, 127)] + [min(3, r2), r2]))
            new = "".join("    X(%d, %d, %d, %d, %d, %d, %d) \\\n" % (L, r1, r2, r3, nt, rpw, z) for z in nz)
            tail = s[m.end():]
            if not m.group(0).rstrip().endswith("\\"):       # last line of a macro (no continuation)
                new = new.rstrip()[:-1].rstrip() + "\n"
            s = s[:m.start()] + new + tail
        else:
            M = int(key); r1, r2, r3, t, nt = v
            assert r1 * r2 * r3 == M and r1 * r2 * t == nt and r3 % 2 == 0, item
            s2, n = re.subn(r"X\(%d, \d+, \d+, \d+, \d+, \d+\)" % M, "X(%d, %d, %d, %d, %d, %d)" % (M, r1, r2, r3, t, nt), s)
            assert n == 1, item
            s = s2
open(p, "w").write(s)
