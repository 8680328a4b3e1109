#!/bin/bash
# SQ-level counters of the two hot kernels (issue/LDS/bank-conflict picture), one counter group per pass
export TMPDIR=/tmp
OUT=gpurun_out/pmc_sq; mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -o p$i -- python3 bench.py --steps 1 --warmup 1 --no-clock-warm --no-cpu-baseline --no-extras > /dev/null 2> $OUT/p$i.err || echo "pass $i failed"
done
python3 tools/pmc_summary.py $(find $OUT -name "*counter_collection.csv") | grep -A6 "k_fast_rows_multi\|k_fast_cols grid" > $OUT/summary.txt
cat $OUT/summary.txt
