#!/bin/bash
# round 5, session h: every output-kernel configuration on its padded LDS image (fast_cols.hpp: FC_COL_LAYOUTS) against the product before it
# (ab/base.so) and the dense image of the same tree (ab/cnopad.so): parity of every specialised length, then the size list, then counters of cfg5 / big sizes
export TMPDIR=/tmp
OUT=gpurun_out/r05u; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_fast_paths.py -m gpu -x -q > $OUT/parity.log 2>&1; rc=$?; echo "parity rc $rc"; tail -2 $OUT/parity.log
[ $rc -ne 0 ] && exit $rc
cp cuda-fft-convolution_amd/libfftconv.so cuda-fft-convolution_amd/ab/cpad.so
REPS=${REPS:-2} bash tools/config_search_run.sh cnopad cpad > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
