#!/bin/bash
# PMC evidence for the launches bench.py does not time (run on the GPU box from the repo root):
#   bash profiles/collect_probes.sh <tag> "<program + args>"
# Every counter group is its own rocprofv3 invocation (never combined with a trace domain); the
# program comes directly after `--` (no env/bash hop).  Condensed by profiles/pmc_dispatches.py.
set -u
TAG=$1
shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 "$@" > $OUT/kt.log 2>&1
i=0
for grp in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS_F32" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32" \
  "GRBM_GUI_ACTIVE FETCH_SIZE" \
  "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc$i -- python3 "$@" > $OUT/pmc$i.log 2>&1
done
python3 profiles/pmc_dispatches.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/kt.log | grep -v "^$" | tail -20 >> $OUT/summary.txt
