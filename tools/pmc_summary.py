#!/usr/bin/env python3
"""Summarises a rocprofv3 counter_collection.csv: per kernel name, mean of each counter per dispatch."""
import collections, csv, re, sys
for path in sys.argv[1:]:
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); name = re.split(r"[(<]", name)[0][-40:] + " grid=" + r["Grid_Size"]
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for name, cs in agg.items():
        print(name)
        for c, v in sorted(cs.items()):
            print("    %-28s n=%-4d mean=%.4g" % (c, len(v), sum(v) / len(v)))
