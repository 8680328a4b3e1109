# tools/debug/loop_blockwise.sh <lib.so> <iterations> <tag>: loops the one-shot test that hit the copy-thread incident
LIB=$1; N=$2; TAG=$3
export ABRT_TRACE_FILE=$PWD/gpurun_out/native_bt_$TAG.txt; rm -f $ABRT_TRACE_FILE
export FFTCONV_LIB=$PWD/$LIB
for i in $(seq 1 $N); do
  LD_PRELOAD=$PWD/tools/debug/abrt_trace.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -p no:faulthandler -k "blockwise" > gpurun_out/blk_$TAG.log 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then echo "$TAG: iteration $i rc=$rc"; exit 1; fi
done
echo "$TAG: $N iterations clean"
