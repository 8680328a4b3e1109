#!/bin/bash
# SQ counters of the co-residency probe (tools/microbench/fused_roles[_nomem]): dispatch order of the fused kernel is
# all C, all R, C 1 per CU, R 1 per CU, C + R (two dispatches each), then the product-shaped rows1 / rows2 / cols8.
export TMPDIR=/tmp
BIN=${1:-fused_roles}
OUT=gpurun_out/pmc_$BIN; mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INST_LEVEL_LDS SQ_IFETCH"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -o p$i -- ./tools/microbench/$BIN 64 2 16 pmc > /dev/null 2> $OUT/p$i.err || echo "pass $i failed"
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
rows = []
for p in sorted(glob.glob(out + "/**/*counter_collection.csv", recursive=True)):
    rows += list(csv.DictReader(open(p)))
# per (dispatch order within its kernel name, kernel) -> counters
by = collections.OrderedDict()
for r in rows:
    name = r["Kernel_Name"]
    short = "fused" if "pk_fused" in name else ("rows1" if "Li192ELi1E" in name or "192, 1>" in name else "rows2" if "pk_rows" in name else "cols8" if "pk_cols" in name else None)
    if short is None: continue
    key = (short, int(r["Dispatch_Id"]))
    by.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
# dispatch ids differ between passes only if the order differs: group by order of appearance per pass instead
seq = collections.defaultdict(lambda: collections.defaultdict(list))
for (short, did), cs in by.items():
    for c, v in cs.items():
        seq[short][c].append((did, v))
labels = {"fused": ["all C", "all C", "all R", "all R", "C 1/CU", "C 1/CU", "R 1/CU", "R 1/CU", "C + R", "C + R"]}
for short, cs in seq.items():
    n = max(len(v) for v in cs.values())
    for i in range(n):
        lab = labels.get(short, [short] * n)[i] if i < len(labels.get(short, [short] * n)) else short
        if short == "fused" and i % 2 == 0: continue     # the second dispatch of each mode
        print("%-8s %-8s " % (short, lab) + "  ".join("%s=%.4g" % (c, sorted(v)[i][1]) for c, v in sorted(cs.items()) if i < len(v)))
PY
