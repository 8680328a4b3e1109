# round 5, session f: maps per launch (the intermediate's size) at cfg3 -- 64 (5-GiB cap, default), 128, 256
set -o pipefail
T=gpurun_out/r05f; mkdir -p $T
for rep in 1 2; do for b in 64 128 256; do python3 bench.py --batch-maps $b --no-cpu-baseline --no-extras --steps 20 2>/dev/null | tail -1 | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernels']
print('cfg3 batch_maps $b:', round(j['value'],1), 'Gpx/s', round(j['ms_per_step'],3), 'ms/step', {n:(round(v['avg_ms']/v['units_per_launch']*1e3,2)) for n,v in k.items()}, 'ok' if j['check_ok'] else 'CHECK FAILED')
"; done; done | tee $T/batch_maps.txt
for b in 64 128; do python3 bench.py --config cfg4 --filters 128 --batch-maps $b --no-cpu-baseline --no-extras --steps 20 2>/dev/null | tail -1 | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernels']
print('cfg4/128 batch_maps $b:', round(j['value'],1), 'Gpx/s', round(j['ms_per_step'],3), 'ms/step', {n:(round(v['avg_ms']/v['units_per_launch']*1e3,2)) for n,v in k.items()}, 'ok' if j['check_ok'] else 'CHECK FAILED')
"; done | tee -a $T/batch_maps.txt
