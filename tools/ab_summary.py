#!/usr/bin/env python3
"""Medians per variant of a tools/config_search_run.sh listing: ab_summary.py file [file...]  (column 'cols' = the output kernel's us per launch)"""
import collections, re, statistics, sys
d = collections.defaultdict(lambda: collections.defaultdict(list))
order = []
for path in sys.argv[1:]:
    for l in open(path):
        m = re.match(r'(\w+) (\S+) K=(\d+) (?:F=\d+ )?n=(\d+) window (\S+) transform (\S+)( x\d+ blocks)? +([\d.]+) us/step +([\d.]+) Gpx/s \| rows ([\d.]+) us x(\d+) +cols_c2r ([\d.]+) us', l)
        if not m:
            continue
        key = "%s K=%s n=%s transform %s%s" % (m.group(2), m.group(3), m.group(4), m.group(6), m.group(7) or "")
        if m.group(1) not in order:
            order.append(m.group(1))
        d[key][m.group(1)].append((float(m.group(8)), float(m.group(10)), float(m.group(12))))
for key, v in d.items():
    print(key)
    for name in order:
        if name not in v:
            continue
        med = [statistics.median(x[i] for x in v[name]) for i in range(3)]
        c = [x[2] for x in v[name]]
        print("   %-8s n %d  step %9.1f  rows %8.1f  cols median %8.1f  min %8.1f  max %8.1f us" % (name, len(c), med[0], med[1], med[2], min(c), max(c)))
