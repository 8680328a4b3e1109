set -o pipefail
python -m pytest tests -m gpu -q > gpurun_out/r04z_gputests.log 2>&1; rc=$?; tail -3 gpurun_out/r04z_gputests.log
if [ $rc -ne 0 ]; then grep -n "Error\|FAILED\|assert" gpurun_out/r04z_gputests.log | tail -20; exit $rc; fi
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/profile_round.sh r04z ${COMMIT:-unknown} > gpurun_out/r04z_profile_round.log 2>&1; echo "profile_round rc $?"
python - <<'PY'
import json
j=json.loads(open('gpurun_out/prof_r04z/bench.json').read().strip().splitlines()[-1])
print('bench', round(j['value'],1), 'frac', round(j['roofline']['frac'],3), j['config']['tune_placement'], 'resident', round(j.get('value_kernels_resident',0),1))
PY
STEPS=20 bash tools/all_cfgs.sh > gpurun_out/r04z_all_configs.txt 2>&1; cat gpurun_out/r04z_all_configs.txt
for c in cfg3f4 mid512 hd720 mid2900 big6000 big8192; do python bench.py --config $c --no-cpu-baseline --no-extras --steps 10 --warmup 3 --check 2>/dev/null | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=j['kernels']
print('$c', j['config']['transform'], round(j['value'],1), 'Gpx/s', round(j['ms_per_step'],4), 'ms/step frac', round(j['hbm_frac_of_peak'],3), {n:(round(v['avg_ms']/v['units_per_launch']*1e3,2)) for n,v in k.items()}, 'ok' if j['check_ok'] else 'CHECK FAILED')
" | tee -a gpurun_out/r04z_all_configs.txt; done
python tools/size_sweep.py > gpurun_out/r04z_size_sweep.txt 2> gpurun_out/r04z_size_sweep.err; echo "sweep rc $?"; tail -2 gpurun_out/r04z_size_sweep.txt
