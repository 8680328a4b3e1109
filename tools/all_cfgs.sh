# one bench line per BASELINE config (device-resident timing), defaults
for c in cfg1 cfg2 cfg3 "cfg4 --filters 128" cfg5; do python bench.py --config $c --no-cpu-baseline --steps ${STEPS:-10} --warmup 5 --check 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=j['kernels']
print('$c', round(j['value'],1), 'Gpx/s', round(j['ms_per_step'],4), 'ms/step frac', round(j['hbm_frac_of_peak'],3), {n:(round(v['avg_ms']/v['units_per_launch']*1e3,2)) for n,v in k.items()}, 'err', j.get('check_max_rel_err'))
"; done
