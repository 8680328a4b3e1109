# kernel trace of one config: tools/trace_cfg.sh cfg1 [extra bench args]  -> gpurun_out/trace_<cfg>/
CFG=$1; shift
OUT=gpurun_out/trace_$CFG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 bench.py --config $CFG --steps 20 --warmup 3 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/err.txt
find $OUT -name "*.db" -delete
python3 - <<PY
import csv,glob,json
f=glob.glob('$OUT/**/t_kernel_trace.csv',recursive=True)[0]
rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r['Start_Timestamp']))
# the timed region: last 20+3(profile) steps; print one step in the middle of the timed region
idx=[i for i,r in enumerate(rows) if 'k_rows_fwd' in r['Kernel_Name'] or 'k_fast_rows_fwd' in r['Kernel_Name'] or 'k_image' in r['Kernel_Name']]
print('$CFG', json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])['ms_per_step'], 'ms/step (under rocprof)')
if len(idx) > 12:
    a,b=idx[10],idx[11]
    a-=1
    prev=None
    for r in rows[a:b]:
        s=int(r['Start_Timestamp']);e=int(r['End_Timestamp'])
        print("  %-48s grid %7s wg %4s dur %7.2f us gap %6.2f us"%(r['Kernel_Name'].split('(')[0][-48:], r['Grid_Size_X'], r['Workgroup_Size_X'], (e-s)/1e3, (s-prev)/1e3 if prev else 0))
        prev=e
    print('  step period %.2f us'%((int(rows[idx[11]]['Start_Timestamp'])-int(rows[idx[10]]['Start_Timestamp']))/1e3))
PY
