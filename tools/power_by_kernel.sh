#!/bin/bash
# board power and engine clock while each hot kernel of cfg3 runs ALONE, back to back (tools/microbench/coreside <maps> <reps> power),
# sampled twice a second with rocm-smi: which of the two kernels is at the power cap?
./tools/microbench/coreside 64 12 power > /tmp/pbk.txt 2>&1 &
P=$!
sleep 4
for i in $(seq 1 40); do
  kill -0 $P 2>/dev/null || break
  echo "t=$(date +%s.%N | cut -c1-14) $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'sclk|Graphics Package Power' | sed 's/.*sclk clock level: 1: //; s/.*Power (W): /W /' | tr '\n' ' ')"
  sleep 0.5
done
wait $P
cat /tmp/pbk.txt
