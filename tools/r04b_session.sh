set -o pipefail
mkdir -p gpurun_out/r04b
python -m pytest tests -m gpu -x -q > gpurun_out/r04b/gputests.log 2>&1; rc=$?; tail -4 gpurun_out/r04b/gputests.log
if [ $rc -ne 0 ]; then grep -n "Error\|FAILED\|assert" gpurun_out/r04b/gputests.log | tail -20; exit $rc; fi
python bench.py --steps 20 --warmup 5 > gpurun_out/r04b/bench.json 2> gpurun_out/r04b/bench.err; echo "bench rc $?"; tail -c 600 gpurun_out/r04b/bench.err
for c in "cfg2" "cfg2 --exact-window" "cfg4 --filters 128" "cfg4 --filters 128 --exact-window" "cfg2 --filters 256" "cfg2 --filters 256 --exact-window"; do
  python bench.py --config $c --no-cpu-baseline --no-extras --steps 20 --warmup 5 --check 2>/dev/null | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=j['kernels']
print('$c', j['config']['transform'], round(j['value'],1), 'Gpx/s', round(j['ms_per_step']*1e3,1), 'us/step frac', round(j['hbm_frac_of_peak'],3), {n:(round(v['avg_ms']/v['units_per_launch']*1e3,2)) for n,v in k.items()}, 'ok' if j['check_ok'] else 'CHECK FAILED')
" | tee -a gpurun_out/r04b/native_window_ab.txt
done
