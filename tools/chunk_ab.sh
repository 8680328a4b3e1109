# same-box A/B of the kernels' column-spectrum chunk budget -> gpurun_out/chunk_ab.txt
mkdir -p gpurun_out
for rep in 1 2 3; do
  for c in 512 160; do
    python3 bench.py --config cfg3 --kernel-chunk-mb $c --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('cfg3 chunk_mb $c', round(j['value'],1), 'Gpx/s', round(j['ms_per_step'],3), 'ms', {k:(round(v['avg_ms']/v['units_per_launch']*1e3,2), round(v.get('separate_pass_avg_ms',0)/v['units_per_launch']*1e3,2)) for k,v in j['kernels'].items()}, j['roofline']['frac'])" | tee -a gpurun_out/chunk_ab.txt
  done
done
for rep in 1 2; do
  for c in 512 160 80; do
    python3 bench.py --config cfg4 --filters 128 --kernel-chunk-mb $c --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('cfg4/128 chunk_mb $c', round(j['value'],1), 'Gpx/s', round(j['ms_per_step'],3), 'ms', {k:(round(v['avg_ms']/v['units_per_launch']*1e3,2), round(v.get('separate_pass_avg_ms',0)/v['units_per_launch']*1e3,2)) for k,v in j['kernels'].items()})" | tee -a gpurun_out/chunk_ab.txt
  done
done
