#!/usr/bin/env python3
"""Register / spill / occupancy table of every kernel from the build's csrc/*.rpt files
(hipcc -Rpass-analysis=kernel-resource-usage, written by csrc/Makefile).
usage: rpt_summary.py [--all] [substring ...]     (default: kernels that spill or use scratch)"""
import glob, os, re, subprocess, sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "cuda-fft-convolution_amd", "csrc")


def kernels(csrc=CSRC):
    out = []
    for f in sorted(glob.glob(os.path.join(csrc, "*.rpt"))):
        cur = None
        for line in open(f, errors="replace"):
            m = re.search(r"remark:\s+Function Name: (\S+)", line)
            if m:
                cur = {"unit": os.path.basename(f)[:-4], "mangled": m.group(1)}
                out.append(cur)
                continue
            m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass", line)
            if m and cur is not None:
                cur[m.group(1).strip()] = m.group(2)
    names = "\n".join(k["mangled"] for k in out)
    try:
        dem = subprocess.run(["c++filt"], input=names, capture_output=True, text=True).stdout.splitlines()
    except Exception:
        dem = names.splitlines()
    for k, d in zip(out, dem):
        d = d.replace("fc::(anonymous namespace)::", "").replace("fc::", "")
        k["name"] = re.sub(r"\(.*$", "", d).replace("void ", "")
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    show_all = "--all" in sys.argv
    for k in kernels():
        spill, scratch = int(k.get("VGPRs Spill", 0)), int(k.get("ScratchSize", 0))
        if args and not any(a in k["name"] for a in args):
            continue
        if not args and not show_all and not spill and not scratch:
            continue
        print("%-24s vgpr %3s  scratch %4d  spill %3d  occ %s  sgpr %3s  %s" % (k["unit"], k.get("VGPRs"), scratch, spill, k.get("Occupancy"),
                                                                               k.get("TotalSGPRs"), k["name"]))
