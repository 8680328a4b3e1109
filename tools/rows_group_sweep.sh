# A/B of the multi-map row kernel: rows_group x batch_maps on a config (arguments group:batch ...)
CFG=${CFG:-cfg3}
run() { python bench.py --config $CFG --rows-group $1 --batch-maps $2 --no-cpu-baseline --steps 3 --warmup 1 --check 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=j['kernels']
print('$CFG group $1 batch $2', round(j['value'],1), {n:(round(v['avg_ms']/v['units_per_launch']*1e3,2)) for n,v in k.items()}, 'image_ms', round(j['image_ms'],3), 'err', j.get('check_max_rel_err'))
"; }
for a in "$@"; do run ${a%%:*} ${a##*:}; done
