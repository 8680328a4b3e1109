set -o pipefail
mkdir -p gpurun_out/r04e
python -m pytest tests/test_fast_paths.py tests/test_gpu_fuzz.py -m gpu -x -q -k "big_square or length_pair or random_shapes or multi_map or other_configs" > gpurun_out/r04e/gputests.log 2>&1; rc=$?; tail -4 gpurun_out/r04e/gputests.log
if [ $rc -ne 0 ]; then grep -n "Error\|FAILED\|assert" gpurun_out/r04e/gputests.log | tail -20; exit $rc; fi
for a in "4800 4800 63" "5400 5400 63" "6000 6000 63 32" "6600 6600 63 32" "8192 8192 127 32" "8192 8192 127 32 2"; do python tools/profile_shape.py $a 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04e/big_shapes.txt; done
