#!/bin/bash
# round 5, session j: the row kernel's first round de-phased (kernels_rows_multi.inc: FC_ROWS_STAGGER_TICKS / _RAMP) against the product, same box, fresh process per line;
# VARIANTS names the ab/*.so builds; the STAGGER line of a variant says how workgroups map to CUs
export TMPDIR=/tmp
OUT=gpurun_out/r05x; mkdir -p $OUT
AB=$PWD/cuda-fft-convolution_amd/ab
cp cuda-fft-convolution_amd/libfftconv.so $AB/new.so
for rep in $(seq 1 ${REPS:-4}); do for v in new ${VARIANTS:-stg3 stg5}; do
  echo -n "$v "; STAGGER=1 FFTCONV_LIB=$AB/$v.so python tools/profile_shape.py ${SHAPE:-4096 4096 127 64} 2>&1 | grep -v amdgpu.ids | sed 's/F=1 //; s/spec 3: //; s/kernel_cols.*spectral_rows/rows/; s/image_cols.*//' | cut -c1-${CUT:-420}
done; done > $OUT/${NAME:-ab}.txt 2>&1
cat $OUT/${NAME:-ab}.txt
