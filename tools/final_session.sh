set -o pipefail
python -m pytest tests -m gpu -x -q > gpurun_out/r04z_gputests.log 2>&1; rc=$?; tail -3 gpurun_out/r04z_gputests.log
if [ $rc -ne 0 ]; then grep -n "Error\|FAILED\|assert" gpurun_out/r04z_gputests.log | tail -20; exit $rc; fi
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/profile_round.sh r04z e70f812 > gpurun_out/r04z_profile_round.log 2>&1; echo "profile_round rc $?"; tail -3 gpurun_out/r04z_profile_round.log
