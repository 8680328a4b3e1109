# round 5, session k (second half): the tile-queue policy after the fix -- default options against dynamic_tiles 0 / 2 over the sizes, the dynamic-tile tests, one line per config
mkdir -p gpurun_out/r05k
python -m pytest tests -m gpu -x -q -k "dynamic_tile or cfg1 or cfg2 or headline or smoke" > gpurun_out/r05k/tests.log 2>&1; rc=$?; tail -3 gpurun_out/r05k/tests.log; if [ $rc -ne 0 ]; then grep -n "Error\|FAILED\|assert" gpurun_out/r05k/tests.log | tail; exit $rc; fi
for rep in 1 2; do for a in "256 256 31 1" "256 256 31 16" "300 300 31 64" "500 500 127 64" "640 640 63 64" "800 800 63 64" "1024 1024 63 16" "1000 1000 63 64" "1400 1400 63 64" "2048 2048 63 64" "4096 4096 127 64"; do for d in default 0 2; do
  echo -n "dynamic_tiles $d: "; if [ $d = default ]; then python tools/profile_shape.py $a 2>&1 | grep -v amdgpu.ids | sed 's/F=1 //; s/spec 3: //' | cut -c1-250; else DYN=$d python tools/profile_shape.py $a 2>&1 | grep -v amdgpu.ids | sed 's/F=1 //; s/spec 3: //' | cut -c1-250; fi
done; done; done > gpurun_out/r05k/dynamic_tiles_policy_check.txt 2>&1; cat gpurun_out/r05k/dynamic_tiles_policy_check.txt
STEPS=20 bash tools/all_cfgs.sh 2>&1 | tee gpurun_out/r05k/all_configs.txt
