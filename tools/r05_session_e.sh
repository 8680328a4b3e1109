# round 5, session e: native windows on the new configurations (tests + A/B against the convenient length), first-round stagger of the
# row kernel (variants st1 / st3 / st6), 8-column tiles for M = 576 / 1056 (variant c8), the intermediate behind a spacer
set -o pipefail
T=gpurun_out/r05e; mkdir -p $T
rocm-smi --showmemorypartition --showcomputepartition > $T/partition_modes.txt 2>&1
python -m pytest tests -m gpu -x -q -k "spectrum or exact or fast_paths or native or headline" > $T/tests.log 2>&1; rc=$?; tail -3 $T/tests.log
if [ $rc -ne 0 ]; then grep -n "Error\|FAILED\|assert" $T/tests.log | tail -20; exit $rc; fi
line() { python3 bench.py $1 --no-cpu-baseline --no-extras --steps 20 --warmup 5 --check 2>/dev/null | tail -1 | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernels']
print('$1', j['config']['transform'], round(j['value'],1), 'Gpx/s', round(j['ms_per_step']*1e3,1), 'us/step frac', round(j['hbm_frac_of_peak'],3), {n:(round(v['avg_ms']/v['units_per_launch']*1e3,2)) for n,v in k.items()}, 'ok' if j['check_ok'] else 'CHECK FAILED')
"; }
for rep in 1 2; do for c in "--config cfg2" "--config cfg2 --exact-window" "--config cfg4 --filters 128" "--config cfg4 --filters 128 --exact-window" "--config cfg2 --filters 256" "--config cfg2 --filters 256 --exact-window"; do line "$c"; done; done | tee $T/native_window_ab.txt
SHAPES="4096 4096 127 64;4096 4096 63 128;2048 2048 63 64;1024 1024 63 16;1024 1024 63 64;512 512 31 64;256 256 31 16" REPS=2 bash tools/config_search_run.sh st1 st3 st6 c8 > $T/stagger_and_c8.txt 2> $T/stagger_and_c8.err; echo "stagger rc $?"; cat $T/stagger_and_c8.txt
for sp in 0 8192 24576 49152; do for i in 1 2 3 4; do python bench.py --no-cpu-baseline --no-extras --steps 15 --y-spacer-mb $sp 2>/dev/null | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('spacer $sp MB, fresh process $i:', round(j['value'],1), 'Gpx/s  cols us/map', round(j['kernels']['cols_c2r']['avg_ms']/j['kernels']['cols_c2r']['units_per_launch']*1e3,2), 'rows', round(j['kernels']['spectral_rows']['avg_ms']/j['kernels']['spectral_rows']['units_per_launch']*1e3,2))
"; done; done | tee $T/intermediate_spacer.txt
