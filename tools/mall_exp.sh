mkdir -p gpurun_out/r02e
for b in 1 2 3 4 8 64; do
python bench.py --filters 64 --batch-maps $b --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=j['kernels']
print('batch_maps $b', round(j['value'],1), 'Gpx/s', {n:(round(v['avg_ms']/v['units_per_launch']*1e3,2)) for n,v in k.items()}, 'chk', j['check_checksum_max_rel_err'])
"
done
