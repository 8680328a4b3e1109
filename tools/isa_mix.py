#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -S listing: isa_mix.py file.s substring [substring...]"""
import collections, re, sys
lines = open(sys.argv[1]).read().splitlines()
pats = sys.argv[2:]
starts = [i for i, l in enumerate(lines) if re.match(r'^_Z\S+:', l)]
for si, st in enumerate(starts):
    name = lines[st].split(':')[0]
    if not all(p in name for p in pats): continue
    end = starts[si + 1] if si + 1 < len(starts) else len(lines)
    c = collections.Counter()
    for l in lines[st:end]:
        l = l.strip()
        if l.startswith('.') or l.startswith(';'): continue
        m = re.match(r'^([a-z_0-9]+)(\s|$)', l)
        if m: c[m.group(1)] += 1
        if l.startswith('s_endpgm'): break
    g = collections.Counter()
    for k, v in c.items():
        if k.startswith(('ds_', 'global_', 'scratch_', 'buffer_', 'flat_')): g[k] += v
        elif k.startswith('v_'): g['VALU'] += v
        elif k.startswith('s_waitcnt'): g['s_waitcnt'] += v
        elif k.startswith('s_barrier'): g['s_barrier'] += v
        elif k.startswith('s_'): g['SALU'] += v
    print(name[:150], sum(c.values()))
    for k, v in sorted(g.items(), key=lambda x: -x[1]): print("    %-28s %d" % (k, v))
    vc = collections.Counter({k: v for k, v in c.items() if k.startswith('v_')})
    print("    top VALU:", vc.most_common(12))
