set -o pipefail
mkdir -p gpurun_out/r04c
python bench.py --steps 20 --warmup 5 > gpurun_out/r04c/bench.json 2> gpurun_out/r04c/bench.err; echo "bench rc $?"
python tools/size_sweep.py > gpurun_out/r04c/size_sweep.txt 2> gpurun_out/r04c/size_sweep.err; echo "sweep rc $?"; tail -3 gpurun_out/r04c/size_sweep.txt
STEPS=20 bash tools/all_cfgs.sh > gpurun_out/r04c/all_configs.txt 2>&1; cat gpurun_out/r04c/all_configs.txt
for c in cfg3f4 mid512 hd720 mid2900 big6000 big8192; do python bench.py --config $c --no-cpu-baseline --no-extras --steps 10 --warmup 3 --check 2>/dev/null | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=j['kernels']
print('$c', j['config']['transform'], round(j['value'],1), 'Gpx/s', round(j['ms_per_step'],4), 'ms/step frac', round(j['hbm_frac_of_peak'],3), {n:(round(v['avg_ms']/v['units_per_launch']*1e3,2)) for n,v in k.items()}, 'ok' if j['check_ok'] else 'CHECK FAILED')
" | tee -a gpurun_out/r04c/other_configs.txt; done
