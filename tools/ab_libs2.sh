# per-kernel times of cfg3 with alternate builds of the library (FFTCONV_LIB), settled clocks, checked:
#   tools/ab_libs2.sh lib1.so lib2.so ...   (each twice, alternating)
for rep in 1 2; do
for lib in "$@"; do
  FFTCONV_LIB=$PWD/$lib python3 bench.py --config ${CFG:-cfg3} --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python3 -c "
import sys,json
j=json.loads(sys.stdin.read())
print('$lib', round(j['value'],1), {n:(round(v['avg_ms']/v['units_per_launch']*1e3,2)) for n,v in j['kernels'].items()}, j['check_ok'], j.get('check_checksum_max_rel_err'))
"
done
done
