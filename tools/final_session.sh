# First GPU call at the end of a round: the whole GPU tier, smoke, the profiled default bench (rocprofv3 stats + PMC passes + the plain
# line).  tools/final_session_2.sh is the second call (all configs, denominator, other sizes, size sweep).
# usage (through gpurun, from the repo root): TAG=r05z COMMIT=<hash> bash tools/final_session.sh
set -o pipefail
TAG=${TAG:-r05z}
python -m pytest tests -m gpu -q > gpurun_out/${TAG}_gputests.log 2>&1; rc=$?; tail -3 gpurun_out/${TAG}_gputests.log
if [ $rc -ne 0 ]; then grep -n "Error\|FAILED\|assert" gpurun_out/${TAG}_gputests.log | tail -20; exit $rc; fi
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/profile_round.sh $TAG ${COMMIT:-unknown} > gpurun_out/${TAG}_profile_round.log 2>&1; echo "profile_round rc $?"
python - <<PY
import json
j=json.loads(open('gpurun_out/prof_$TAG/bench.json').read().strip().splitlines()[-1])
print('bench', round(j['value'],1), 'frac', round(j['roofline']['frac'],3), 'other placement policy beside', round(j.get('value_untuned_placement', j.get('value_tuned_placement', 0)),1), j['config'].get('tune_placement'), 'resident', round(j.get('value_kernels_resident',0),1), 'cpu', round(j['cpu_baseline']['value'],4), j['cpu_baseline']['value_from'])
PY
