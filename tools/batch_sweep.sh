for b in 1 2 3 4 6 8 16 32; do python bench.py --batch-maps $b --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=j['kernels']
print($b, round(j['value'],1), {n:(round(v['avg_ms']/v['units_per_launch']*1e3,2)) for n,v in k.items()})
"; done > gpurun_out/s2_sweep.txt 2>&1
