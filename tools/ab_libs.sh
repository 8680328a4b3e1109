# per-kernel times of cfg3 with alternate builds of the library (FFTCONV_LIB), e.g. ablation builds under cuda-fft-convolution_amd/ab/
for lib in "$@"; do
  echo "== $lib"
  FFTCONV_LIB=$PWD/$lib python bench.py --config ${CFG:-cfg3} --rows-group ${GROUP:-0} --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=j['kernels']
print(round(j['value'],1), {n:(round(v['avg_ms']/v['units_per_launch']*1e3,2)) for n,v in k.items()})
"
done
