# A/B of the multi-map row kernel: FFTCONV_ROWS_GROUP x batch_maps on cfg3
run() { FFTCONV_ROWS_GROUP=$1 python bench.py --batch-maps $2 --no-cpu-baseline --steps 3 --warmup 1 --check 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=j['kernels']
print('group $1 batch $2', round(j['value'],1), {n:(round(v['avg_ms']/v['units_per_launch']*1e3,2)) for n,v in k.items()}, 'err', j.get('check_max_rel_err'))
"; }
run 0 32; run 2 32; run 4 32; run 8 32; run 8 64; run 16 64; run 4 64; run 16 128; run 32 128
