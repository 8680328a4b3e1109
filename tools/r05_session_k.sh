#!/bin/bash
# round 5, session k: the pair row kernel (fast_rows_pair.hpp, -DFC_ROWS_PAIR=1 build = ab/pair.so) against the product: parity of the specialised paths, then same-box A/B
export TMPDIR=/tmp
OUT=gpurun_out/r05k2; mkdir -p $OUT
AB=$PWD/cuda-fft-convolution_amd/ab
cp cuda-fft-convolution_amd/libfftconv.so $AB/new.so
FFTCONV_LIB=$AB/pair.so timeout -k 10 400 python -m pytest tests/test_fast_paths.py -m gpu -x -q -k "${TESTS:-all_variants or one_dimension or multi_map or other_configs}" > $OUT/parity.log 2>&1; rc=$?; echo "parity rc $rc"; tail -3 $OUT/parity.log
[ $rc -ne 0 ] && exit $rc
IFS=';' read -ra LIST <<< "${SHAPES:-4096 4096 127 64;4096 4096 63 128}"
for rep in $(seq 1 ${REPS:-4}); do for a in "${LIST[@]}"; do for v in new pair; do
  echo -n "$v "; FFTCONV_LIB=$AB/$v.so python tools/profile_shape.py $a 2>&1 | grep -v amdgpu.ids | sed 's/F=1 //; s/spec 3: //; s/kernel_cols.*spectral_rows/rows/; s/image_cols.*//' | cut -c1-200
done; done; done > $OUT/${NAME:-ab}.txt 2>&1
cat $OUT/${NAME:-ab}.txt
