mkdir -p gpurun_out/r05k
for rep in 1 2; do for a in "256 256 31 1" "256 256 31 16" "300 300 31 64" "500 500 127 64" "640 640 63 64" "800 800 63 64" "1024 1024 63 16" "1000 1000 63 64" "1200 1200 63 64" "1400 1400 63 64" "1600 1600 63 64" "1800 1800 63 64" "2048 2048 63 64" "2200 2200 63 64"; do for d in 0 1; do
  echo -n "dynamic_tiles $d: "; DYN=$d python tools/profile_shape.py $a 2>&1 | grep -v amdgpu.ids | sed 's/F=1 //; s/spec 3: //' | cut -c1-250
done; done; done > gpurun_out/r05k/dynamic_tiles_by_size.txt 2>&1; cat gpurun_out/r05k/dynamic_tiles_by_size.txt
