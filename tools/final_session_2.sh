# second GPU call of the round's end (a gpurun call is at most 20 minutes): one line per BASELINE config, the strong-scaling denominator of cfg4 on one
# GPU, other sizes, the size sweep.  usage: TAG=r05z bash tools/final_session_2.sh
set -o pipefail
TAG=${TAG:-r05z}
STEPS=20 bash tools/all_cfgs.sh > gpurun_out/${TAG}_all_configs.txt 2>&1; cat gpurun_out/${TAG}_all_configs.txt
# the denominator of the N > 1 strong-scaling line: the same workload (cfg4, 1024 filters) on ONE GPU, from THIS library (bench.py
# stamps it with the library's hash and the N > 1 line says whether it matches: same_workload_1gpu.is_of_this_binary)
python bench.py --config cfg4 --no-cpu-baseline --no-extras --steps 10 --warmup 3 > gpurun_out/${TAG}_cfg4_1024_filters_1gpu_denominator.json 2> gpurun_out/${TAG}_cfg4_denominator.err; echo "denominator rc $?"
for c in cfg3f4 mid512 hd720 mid2900 big6000 big8192; do python bench.py --config $c --no-cpu-baseline --no-extras --steps 10 --warmup 3 --check 2>/dev/null | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=j['kernels']
print('$c', j['config']['transform'], round(j['value'],1), 'Gpx/s', round(j['ms_per_step'],4), 'ms/step frac', round(j['hbm_frac_of_peak'],3), {n:(round(v['avg_ms']/v['units_per_launch']*1e3,2)) for n,v in k.items()}, 'ok' if j['check_ok'] else 'CHECK FAILED')
" | tee -a gpurun_out/${TAG}_all_configs.txt; done
python tools/size_sweep.py > gpurun_out/${TAG}_size_sweep.txt 2> gpurun_out/${TAG}_size_sweep.err; echo "sweep rc $?"; tail -2 gpurun_out/${TAG}_size_sweep.txt
