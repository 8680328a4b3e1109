# hardware counters of the output kernel in its fast and slow placement states (tools/microbench/pmc_states.cpp)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_states
i=0
for set in "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum" \
           "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_BUSY_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" \
           "TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum GRBM_UTCL2_BUSY TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_states/p$i -o s -- tools/microbench/pmc_states > gpurun_out/pmc_states/p$i.txt 2>&1
  tail -2 gpurun_out/pmc_states/p$i.txt
  python3 - <<PY
import csv, glob, collections
f = glob.glob('gpurun_out/pmc_states/p$i/**/s_counter_collection.csv', recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if 'k_fast_cols<' in r['Kernel_Name']]
by = collections.OrderedDict()
for r in rows:
    by.setdefault(r['Counter_Name'], []).append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
for name, v in by.items():
    v.sort()
    last = [x[1] for x in v[-8:]]
    fa, sl = sum(last[:4]) / 4, sum(last[4:]) / 4
    print("  %-48s fast %.4g  slow %.4g  (slow/fast %.3f)" % (name, fa, sl, sl / fa if fa else float('nan')))
PY
done
find gpurun_out/pmc_states -name "*.csv" -size +1M -delete; find gpurun_out/pmc_states -name "*.db" -delete
