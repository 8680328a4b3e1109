#!/bin/bash
# A/B of the static tile deal against the dynamic tile queue of the persistent column kernels (plan option "dynamic_tiles",
# fast_cols.hpp: TileQueue), alone and with K workgroups of another kernel co-resident (tools/microbench/cu_hog: the stand-in
# for a collective's channels; small LDS = the hot kernels still fit beside it, large LDS = the output kernel's 148-KB
# workgroup does not).  One bench line per run: cfg4's share of a rank (128 filters) and cfg5 (one image, 64 filters).
# Every run uses the N > 1 pipeline shape on the one GPU (--overlap: two spectrum buffers, the next step's image transform -- and,
# with a hog, one hog launch of HOG_US microseconds where the broadcast would be -- on the side stream beside this step's maps).
# usage (through gpurun, from the repo root): bash tools/contention_ab.sh > gpurun_out/<tag>_contention_ab.txt
STEPS=${STEPS:-10}
run() {   # cfg-args, dynamic, contend
  python3 bench.py $1 --overlap --no-cpu-baseline --no-extras --tune-placement 0 --steps $STEPS --warmup 3 --dynamic-tiles $2 ${3:+--contend $3} 2>/dev/null | tail -1 | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=j['kernels']; c=j['config'].get('contention') or {}
print('%-22s dyn %s  hog %-10s held %3s CUs  %7.1f Gpx/s  %8.3f ms/step  %s  %s' % ('$1'.replace('--config ','').replace('--filters ','n='), '$2', '${3:-none}', c.get('distinct_cus_held','-'), j['value'], j['ms_per_step'], {n:(round(v['avg_ms']/v['units_per_launch']*1e3,2)) for n,v in k.items()}, 'ok' if j['check_ok'] and (not c or c.get('covered_timed_region')) else 'CHECK/COVER FAILED'))
"
}
# the stand-in's time per step: what a ring broadcast of the configuration's spectrum takes at ~100 GB/s per link (cfg4: 69 MB;
# cfg5 has no collective -- its images are sharded -- so its hog is a generic short neighbour)
for cfg in "--config cfg4 --filters 128" "--config cfg5"; do
  case "$cfg" in *cfg4*) HOG_US=${HOG_US_CFG4:-800};; *) HOG_US=${HOG_US_CFG5:-150};; esac
  for rep in 1 2; do
    for dyn in 0 1; do run "$cfg" $dyn; done
  done
  for hog in 8,4,${HOG_US:-800} 16,4,${HOG_US:-800} 32,4,${HOG_US:-800} 8,64,${HOG_US:-800} 16,64,${HOG_US:-800} 32,64,${HOG_US:-800}; do
    for dyn in 0 1; do run "$cfg" $dyn $hog; done
  done
done
