# usage: ab_cfg.sh <cfg> <lib or "-"> [extra bench args]
CFG=$1; LIB=$2; shift; shift
if [ "$LIB" != "-" ]; then export FFTCONV_LIB=$PWD/$LIB; fi
python3 bench.py --config $CFG --no-cpu-baseline --no-extras --steps 20 --warmup 5 --check "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=j['kernels']
print('$CFG', '$LIB', round(j['value'],1), 'Gpx/s', round(j['ms_per_step']*1e3,1), 'us/step', {n:(round(v['avg_ms']/v['units_per_launch']*1e3,2)) for n,v in k.items()}, 'ok' if j['check_ok'] else 'CHECK FAILED')
"
