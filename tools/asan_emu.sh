#!/bin/bash
# Host sanitizers on the kernel bodies (GPU AddressSanitizer is not available on the pool): builds the
# test-only emulator with -fsanitize=address,undefined and runs every path mode over a set of shapes
# (BASELINE-sized one-dimension cases included) against the oracle.  Run from the repository root.
set -e
g++ -O1 -g -std=c++17 -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer -Icuda-fft-convolution_amd/csrc \
    -o /tmp/libfftconv_emu_asan.so tests/emu/emu.cpp
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python3 tools/asan_emu.py
