# measures specialised lengths under the base library and the variant libraries given (cuda-fft-convolution_amd/ab/<name>.so):
# [SHAPES="H W K [maps];..."] [REPS=n] bash tools/config_search_run.sh setA setB > gpurun_out/config_search.txt
ALL="256 256 31 256;500 500 63 128;700 700 63 128;1024 1024 63;1250 1250 63;1450 1450 63;1650 1650 63;1850 1850 63;2048 2048 63;2200 2200 63;2450 2450 63;2700 2700 63;2950 2950 63;3400 3400 63;3750 3750 63;4400 4400 63;4900 4900 63 32;5500 5500 63 32;6000 6000 63 32;6900 6900 63 16;7500 7500 63 16;8192 8192 63 16"
IFS=';' read -ra LIST <<< "${SHAPES:-$ALL}"
for rep in $(seq 1 ${REPS:-1}); do
for a in "${LIST[@]}"; do
  for v in base "$@"; do
    echo -n "$v "; FFTCONV_LIB=$PWD/cuda-fft-convolution_amd/ab/$v.so python tools/profile_shape.py $a 2>&1 | grep -v amdgpu.ids | sed 's/F=1 //; s/spec 3: //; s/kernel_cols.*spectral_rows/rows/; s/image_cols.*//' | cut -c1-200
  done
done
done
