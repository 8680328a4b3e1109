# same-box A/B of the plan's placement tuning (bench.py --tune-placement K): fresh process per line
for rep in 1 2 3 4 5; do
  for k in 0 3; do
    python3 bench.py --config cfg3 --tune-placement $k --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('cfg3 tune_placement $k', round(j['value'],1), 'Gpx/s', round(j['ms_per_step'],3), 'ms', {k:round(v['avg_ms']/v['units_per_launch']*1e3,2) for k,v in j['kernels'].items()}, 'frac', round(j['roofline']['frac'],3), j['config']['tune_placement'], j['check_ok'])"
  done
done
for k in 0 3 0 3; do
  python3 bench.py --config cfg4 --filters 128 --tune-placement $k --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('cfg4/128 tune_placement $k', round(j['value'],1), 'Gpx/s', round(j['ms_per_step'],3), 'ms', {k:round(v['avg_ms']/v['units_per_launch']*1e3,2) for k,v in j['kernels'].items()}, j['config']['tune_placement'], j['check_ok'])"
done
