#!/bin/bash
set -u
O=gpurun_out/r6h; mkdir -p $O
export PT_LIB=build_ab/libptrace_fc.so
export PT_GRID_FORCE_CELLS=1
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "cliff" > $O/pytest_fc1.txt 2>&1; echo "rc $?"
grep -A32 "flight position" $O/pytest_fc1.txt | cut -c1-200
