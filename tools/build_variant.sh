#!/bin/bash
# Builds an A/B variant of libfftconv.so with extra compiler flags into cuda-fft-convolution_amd/ab/<name>.so
# (run with FFTCONV_LIB=<path> python bench.py ...).  tools/build_variant.sh name -DFC_ROWSM_DBG=1 ...
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/cuda-fft-convolution_amd/csrc
OBJ=/tmp/fc_variant_$NAME
mkdir -p $OBJ $ROOT/cuda-fft-convolution_amd/ab
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -Wno-unused-function -DFC_INSTRUMENT $*"
pids=""
for src in $SRC/kernels*.hip; do      # kernels.hip + one translation unit per kernel family and configuration group
  f=$(basename $src .hip)
  /opt/rocm/bin/hipcc $FLAGS -I$SRC -c $src -o $OBJ/$f.o & pids="$pids $!"
done
for f in fftconv_api plan_cache host_ring blockwise placement fftconv_multi; do      # the host units (csrc/Makefile: HOSTUNITS)
  /opt/rocm/bin/hipcc $FLAGS -c $SRC/$f.cpp -o $OBJ/$f.o & pids="$pids $!"
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/cuda-fft-convolution_amd/ab/$NAME.so $OBJ/*.o -ldl
echo built cuda-fft-convolution_amd/ab/$NAME.so
