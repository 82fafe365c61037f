#!/bin/bash
# experiment: the cells build (entries gathered from L2; far rays handed to the wave) for a scene whose entries fit the LDS
set -u
O=gpurun_out/r6g; mkdir -p $O
export PT_LIB=build_ab/libptrace_fc.so
for FC in "" 1; do
  if [ -n "$FC" ]; then export PT_GRID_FORCE_CELLS=1; fi
  echo "== PT_GRID_FORCE_CELLS=$FC" >> $O/fc.txt
  AB_CASES=c2grid,c2band8 timeout -k 10 300 python tools/ab_kernels.py --child >> $O/fc.txt 2>&1
  timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "cliff or tune_measures" > $O/pytest_fc$FC.txt 2>&1; echo "rc $?" >> $O/fc.txt
  grep -E "as built:|\(0, |\(5, |\(11, |\(15, |\(20, |\(29, |passed|failed" $O/pytest_fc$FC.txt | cut -c1-250 >> $O/fc.txt
done
cat $O/fc.txt
