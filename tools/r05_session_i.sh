#!/bin/bash
# round 5, session i: the default bench line under the product of the time (ab/base.so = f28a207b1704a74d) and under this tree's library, alternating, same box
# (two end-of-round sessions landed on boxes 4-5 % slower in BOTH kernels -- the row kernel had not changed: is it the box or the library?)
export TMPDIR=/tmp
OUT=gpurun_out/r05v; mkdir -p $OUT
cp cuda-fft-convolution_amd/libfftconv.so cuda-fft-convolution_amd/ab/new.so
for rep in 1 2 3; do for v in base new; do
  FFTCONV_LIB=$PWD/cuda-fft-convolution_amd/ab/$v.so python bench.py --no-cpu-baseline --no-extras 2> $OUT/${v}_$rep.err | tail -1 > $OUT/${v}_$rep.json
  python - <<PY
import json
j=json.loads(open('$OUT/${v}_$rep.json').read().strip().splitlines()[-1]); r=j['roofline']
print('$v', $rep, 'value', round(j['value'],1), 'frac', round(r['frac'],4), 'launch ms', round(r['avg_launch_ms'],4), r['library_sha256'], {k:(round(x['avg_ms']/x['units_per_launch']*1e3,2)) for k,x in j['kernels'].items()})
PY
done; done 2>&1 | tee $OUT/summary.txt
/opt/rocm/bin/rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -v "^=\|^$" | head -30 > $OUT/smi.txt
