# one gpurun session: GPU tests, the per-config table and the size sweep; results under gpurun_out/<tag>/
# usage (through gpurun): bash tools/gpu_session.sh <tag> [pytest -k expression]
set -o pipefail
TAG=${1:-session}; KEXPR=${2:-}
mkdir -p gpurun_out/$TAG
if [ -n "$KEXPR" ]; then python -m pytest tests -m gpu -x -q -k "$KEXPR" > gpurun_out/$TAG/gputests.log 2>&1; else python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/gputests.log 2>&1; fi
rc=$?; tail -4 gpurun_out/$TAG/gputests.log
if [ $rc -ne 0 ]; then grep -n "Error\|FAILED\|assert" gpurun_out/$TAG/gputests.log | tail -20; exit $rc; fi
STEPS=20 bash tools/all_cfgs.sh > gpurun_out/$TAG/all_configs.txt 2>&1; cat gpurun_out/$TAG/all_configs.txt
python tools/size_sweep.py > gpurun_out/$TAG/size_sweep.txt 2> gpurun_out/$TAG/size_sweep.err; echo "sweep rc $?"; tail -2 gpurun_out/$TAG/size_sweep.txt
