# per-kernel times of a config under different environment switches: tools/env_ab.sh "VAR=val VAR2=val" "..." 
for e in "$@"; do
  echo "== $e"
  env $e python bench.py --config ${CFG:-cfg3} --no-cpu-baseline --steps 3 --warmup 1 --check 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=j['kernels']
print(round(j['value'],1), {n:(round(v['avg_ms']/v['units_per_launch']*1e3,2)) for n,v in k.items()}, 'err', j.get('check_max_rel_err'))
"
done
