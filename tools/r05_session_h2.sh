#!/bin/bash
# round 5, session h2/h3: longer same-box A/Bs (fresh process per measurement, REPS each) of the padded LDS image on the shapes SHAPES names
export TMPDIR=/tmp
OUT=gpurun_out/r05u; mkdir -p $OUT
cp cuda-fft-convolution_amd/libfftconv.so cuda-fft-convolution_amd/ab/cpad.so
SHAPES="${SHAPES:-700 700 63 128;1450 1450 63;1650 1650 63;2200 2200 63;2950 2950 63;3400 3400 63;6000 6000 63 32}" REPS=${REPS:-6} bash tools/config_search_run.sh cnopad cpad > $OUT/${NAME:-ab2}.txt 2>&1
tail -3 $OUT/${NAME:-ab2}.txt
