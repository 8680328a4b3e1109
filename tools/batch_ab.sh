# same-box A/B of maps per launch: tools/batch_ab.sh  -> gpurun_out/batch_ab.txt
mkdir -p gpurun_out
for rep in 1 2 3; do
  for b in 64 128 256; do
    python3 bench.py --config cfg3 --batch-maps $b --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('cfg3 batch_maps $b', round(j['value'],1), 'Gpx/s', round(j['ms_per_step'],3), 'ms', {k:round(v['avg_ms']/v['units_per_launch']*1e3,2) for k,v in j['kernels'].items()})" | tee -a gpurun_out/batch_ab.txt
  done
done
for rep in 1 2; do
  for b in 64 128; do
    python3 bench.py --config cfg4 --filters 128 --batch-maps $b --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('cfg4/128 batch_maps $b', round(j['value'],1), 'Gpx/s', round(j['ms_per_step'],3), 'ms', {k:round(v['avg_ms']/v['units_per_launch']*1e3,2) for k,v in j['kernels'].items()})" | tee -a gpurun_out/batch_ab.txt
  done
done
