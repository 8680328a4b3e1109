#!/bin/bash
# Host sanitizers on the kernel bodies (GPU AddressSanitizer is not available on the pool): builds the
# test-only emulator with -fsanitize=address,undefined and runs every path mode over a set of shapes
# (BASELINE-sized one-dimension cases, every specialised length of round 4, the native windows under
# exact_window, the 4-column output tiles) against the oracle.  Run from the repository root.
set -e
OBJ=/tmp/fc_emu_asan; mkdir -p $OBJ
FLAGS="-O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -Icuda-fft-convolution_amd/csrc -Itests/emu"
pids=""
for f in emu emu_rows_g0 emu_rows_g1 emu_rows_g2 emu_cols_g0 emu_cols_g1; do
  g++ $FLAGS -c tests/emu/$f.cpp -o $OBJ/$f.o & pids="$pids $!"
done
for p in $pids; do wait $p; done
g++ -shared -fsanitize=address,undefined -o /tmp/libfftconv_emu_asan.so $OBJ/*.o
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python3 tools/asan_emu.py
