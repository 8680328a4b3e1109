set -o pipefail
mkdir -p gpurun_out/r05a
python -m pytest tests -m gpu -x -q -k "dynamic_tile or cfg4_headline or cfg3_headline or smoke" > gpurun_out/r05a/tests.log 2>&1; rc=$?; tail -5 gpurun_out/r05a/tests.log
if [ $rc -ne 0 ]; then grep -n "Error\|FAILED\|assert" gpurun_out/r05a/tests.log | tail -20; exit $rc; fi
timeout -k 10 900 bash tools/contention_ab.sh > gpurun_out/r05a/contention_ab.txt 2> gpurun_out/r05a/contention_ab.err; echo "ab rc $?"; cat gpurun_out/r05a/contention_ab.txt
