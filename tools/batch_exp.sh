# per-kernel times of a config for several maps-per-launch values: tools/batch_exp.sh cfg3 "32 64 128 256"
CFG=${1:-cfg3}
for b in ${2:-32 64 128 256}; do
python bench.py --config $CFG --batch-maps $b --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=j['kernels']
print('$CFG batch_maps $b', round(j['value'],1), 'Gpx/s', round(j['ms_per_step'],3), 'ms', {n:(round(v['avg_ms']/v['units_per_launch']*1e3,2)) for n,v in k.items()}, 'chk', j['check_checksum_max_rel_err'])
"
done
