set -o pipefail
mkdir -p gpurun_out/r04d
python -m pytest tests -m gpu -x -q > gpurun_out/r04d/gputests.log 2>&1; rc=$?; tail -4 gpurun_out/r04d/gputests.log
if [ $rc -ne 0 ]; then grep -n "Error\|FAILED\|assert" gpurun_out/r04d/gputests.log | tail -20; exit $rc; fi
STEPS=20 bash tools/all_cfgs.sh > gpurun_out/r04d/all_configs.txt 2>&1; cat gpurun_out/r04d/all_configs.txt
python tools/size_sweep.py > gpurun_out/r04d/size_sweep.txt 2> gpurun_out/r04d/size_sweep.err; echo "sweep rc $?"; tail -2 gpurun_out/r04d/size_sweep.txt
