# round 5, session d: configuration search for the native windows (exact_window plans: 4160 rows / M = 2080 columns, 1088 rows /
# M = 544 columns; variants built by /tmp/build_sets.sh with tools/config_variant.py + tools/build_variant.sh), the stage-2 -> stage-3
# hand-over microbench (LDS round trip against ds_bpermute), a fresh per-phase timeline of the row kernel
set -o pipefail
T=gpurun_out/r05d; mkdir -p $T
export EXACT=1
SHAPES="4096 4096 63 64;1024 1024 63 64;1024 1024 63 16" REPS=2 bash tools/config_search_run.sh nw1 nw2 nw3 nw4 nw5 nw6 nw7 > $T/native_window_search.txt 2> $T/native_window_search.err; echo "search rc $?"
unset EXACT
cat $T/native_window_search.txt
./tools/microbench/stage3_exchange > $T/stage3_exchange.txt 2>&1; echo "exchange rc $?"; cat $T/stage3_exchange.txt
FFTCONV_LIB=$PWD/cuda-fft-convolution_amd/ab/tl.so python tools/rows_timeline.py > $T/rows_timeline.txt 2>&1; echo "timeline rc $?"; grep -v amdgpu.ids $T/rows_timeline.txt
