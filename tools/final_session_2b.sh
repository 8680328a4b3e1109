# the second half of tools/final_session_2.sh on its own (other sizes appended to ${TAG}_all_configs.txt, then the size sweep): for a round end whose
# first half has already run on the final library.  usage: TAG=r05z bash tools/final_session_2b.sh
set -o pipefail
TAG=${TAG:-r05z}
for c in cfg3f4 mid512 hd720 mid2900 big6000 big8192; do python bench.py --config $c --no-cpu-baseline --no-extras --steps 10 --warmup 3 --check 2>/dev/null | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=j['kernels']
print('$c', j['config']['transform'], round(j['value'],1), 'Gpx/s', round(j['ms_per_step'],4), 'ms/step frac', round(j['hbm_frac_of_peak'],3), {n:(round(v['avg_ms']/v['units_per_launch']*1e3,2)) for n,v in k.items()}, 'ok' if j['check_ok'] else 'CHECK FAILED')
" | tee -a gpurun_out/${TAG}_all_configs.txt; done
python tools/size_sweep.py > gpurun_out/${TAG}_size_sweep.txt 2> gpurun_out/${TAG}_size_sweep.err; echo "sweep rc $?"; tail -2 gpurun_out/${TAG}_size_sweep.txt
