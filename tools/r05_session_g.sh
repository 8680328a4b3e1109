#!/bin/bash
# round 5, session g: the padded LDS image of the output kernel (fast_cols.hpp: col_layout) against the dense one, same box.
# variants: base = the product of the time, cnopad = this tree with -DFC_COLS_NO_BLOCK_PAD=1, cpad = this tree
export TMPDIR=/tmp
OUT=gpurun_out/r05t; mkdir -p $OUT
AB=$PWD/cuda-fft-convolution_amd/ab
FFTCONV_LIB=$AB/cpad.so timeout -k 10 300 python -m pytest tests/test_fast_paths.py -m gpu -x -q -k "all_variants or one_dimension or dynamic" > $OUT/parity.log 2>&1; echo "parity rc $?"; tail -2 $OUT/parity.log
SHAPES="${SHAPES:-4096 4096 127 64;4096 4096 63 128}" REPS=${REPS:-4} bash tools/config_search_run.sh cnopad cpad > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
for v in cnopad cpad; do
  FFTCONV_LIB=$AB/$v.so rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS --output-format csv -d $OUT/pmc_$v -o p -- python3 bench.py --steps 1 --warmup 1 --no-clock-warm --no-cpu-baseline --no-extras > /dev/null 2> $OUT/pmc_$v.err || echo "pmc $v failed"
  echo "== $v"; python3 tools/pmc_summary.py $(find $OUT/pmc_$v -name "*counter_collection.csv") | grep -A4 "k_fast_cols grid"
  find $OUT/pmc_$v -name "*.csv" -size +1M -delete; find $OUT/pmc_$v -name "*.db" -delete
done 2>&1 | tee $OUT/counters.txt
