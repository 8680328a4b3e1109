#!/bin/bash
# engine clock and board power sampled while the default bench steps run, for the product library and (optionally) an
# instrumented one: bash tools/clock_power_watch.sh [lib.so]   (round 3: the row kernel WITHOUT its stores -- a wrong-result
# timing build, FC_ROWSM_DBG = 2 -- holds a higher clock: the part is power-bound)
LIB=$1
if [ -n "$LIB" ]; then export FFTCONV_LIB=$PWD/$LIB; fi
python3 bench.py --steps 1500 --warmup 5 --no-cpu-baseline --no-extras > /tmp/cpw_bench.json 2>/dev/null &
BP=$!
sleep 8
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Power" | head -4 | tr '\n' ' '; echo
  sleep 1
done
wait $BP
python3 -c "
import json
j=json.loads(open('/tmp/cpw_bench.json').read().strip().splitlines()[-1]); print('bench:', round(j['value'],1), 'Gpx/s', round(j['ms_per_step'],3), 'ms/step', {k: round(v['avg_ms']/v['units_per_launch']*1e3,2) for k,v in j['kernels'].items()}, 'check_ok', j['check_ok'])"
