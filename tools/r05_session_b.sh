# round 5, session b: GPU tier at the tree's head, the default bench line (headline on default plan options, cpu_baseline with the
# native-built port and pocketfft complex128), five fresh processes of default against tuned placement
set -o pipefail
T=gpurun_out/r05b; mkdir -p $T
python -m pytest tests -m gpu -q -x > $T/gputests.log 2>&1; rc=$?; tail -3 $T/gputests.log
if [ $rc -ne 0 ]; then grep -n "Error\|FAILED\|assert" $T/gputests.log | tail -20; exit $rc; fi
python bench.py > $T/bench.json 2> $T/bench.err; echo "bench rc $?"
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r05b/bench.json').read().strip().splitlines()[-1])
print('bench', round(j['value'],1), 'tuned beside', round(j.get('value_tuned_placement',0),1), 'frac', round(j['roofline']['frac'],3), 'resident', round(j.get('value_kernels_resident',0),1))
c=j['cpu_baseline']; print('cpu', c['value'], c['cores'], c['value_from']); print({k:(v.get('value'),v.get('cores')) for k,v in c['faithful_variants'].items()})
PY
for i in 1 2 3 4 5; do python bench.py --no-cpu-baseline --no-host-output --no-multi-feature --steps 20 2>/dev/null | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('fresh process $i: default', round(j['value'],1), 'tuned', round(j.get('value_tuned_placement',0),1), j.get('tuned_placement'), 'frac', round(j['roofline']['frac'],4), 'cols us/map', round(j['kernels']['cols_c2r']['avg_ms']/j['kernels']['cols_c2r']['units_per_launch']*1e3,2), 'rows', round(j['kernels']['spectral_rows']['avg_ms']/j['kernels']['spectral_rows']['units_per_launch']*1e3,2))
" | tee -a $T/default_vs_tuned_fresh_processes.txt; done
