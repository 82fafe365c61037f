#!/bin/bash
# Collects the rocprofv3 evidence bench.py's roofline numbers are checked against.
# Run on the GPU box from the repo root:   bash profiles/collect.sh <tag>
# Kernel-trace/stats and each PMC group run as SEPARATE rocprofv3 invocations (never combined).
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# round 5: the classes of the VALU stream beside FMA / MUL / TRANS (what are the other two thirds?).  gfx950 exposes
# SQ_INSTS_VALU_ADD_F32, _FMA_F32, _MUL_F32, _TRANS_F32, _INT32, _CVT and the MFMA families; a name the box does not
# know fails only its own pass (see pmc6.log).
PT_PMC_CLASSES=${PT_PMC_CLASSES:-SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_ADD_F16 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FMA_F64}
if [ "${PT_COLLECT_MAIN:-1}" = "1" ]; then
BENCH="python3 bench.py --no-cpu-baseline --no-work-count"   # the default workload (config 2, 16 steps = 1024 spp); keeps the list-walk leg: the scalar list kernel is in the same trace
timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $BENCH > $OUT/kt.log 2>&1
# the other BASELINE configs (one launch each after pt_tune): kernel trace only
timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_configs -- python3 tools/config_sweep.py config3 config4 config5 default > $OUT/kt_configs.log 2>&1
# the reference's own operating point: the animation loop replayed from a hipGraph (bench.py --config default)
timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_frames -- python3 bench.py --config default --frames 200 > $OUT/kt_frames.log 2>&1
i=0
for grp in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS_F32" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32" \
  "GRBM_GUI_ACTIVE FETCH_SIZE" \
  "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
  "$PT_PMC_CLASSES" ; do
  i=$((i+1))
  echo "pmc pass $i ($grp)" >> $OUT/progress.txt
  timeout -k 5 600 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc$i -- $BENCH > $OUT/pmc$i.log 2>&1
done
python3 profiles/summarize.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
fi  # PT_COLLECT_MAIN=0 skips the config-2 part (PT_COLLECT_CONFIGS="4" re-collects one config)
# the other BASELINE configs through the same bench line, each with its own kernel trace and PMC passes: the records
# bench.py attaches to `--config 3 / 4 / 5` lines (profiles/summarize.py --merge <dirs> writes profiles/pmc_traffic.json)
for CFG in ${PT_COLLECT_CONFIGS-3 4 5 default}; do  # (PT_COLLECT_CONFIGS="" collects config 2 only)
  OC=gpurun_out/prof_${TAG}_c$CFG
  mkdir -p $OC
  BC="python3 bench.py --config $CFG --no-cpu-baseline --no-work-count --no-list-walk --no-first-frame"
  # the reference's own operating point: only the replayed animation loop + the group's trace launch on its own
  # (the longest pt_trace_kernel_small_t1 dispatches are then the groups of 16 frames, which summarize.py keeps)
  if [ "$CFG" = "default" ]; then BC="python3 bench.py --config default --no-cpu-baseline --no-extra-legs --frames 640"; fi
  timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OC/kt -- $BC > $OC/kt.log 2>&1
  i=0
  for grp in \
    "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
    "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS_F32" \
    "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32" \
    "GRBM_GUI_ACTIVE FETCH_SIZE" \
    "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
    "$PT_PMC_CLASSES" ; do
    i=$((i+1))
    echo "config $CFG pmc pass $i ($grp)" >> $OC/progress.txt
    timeout -k 5 600 rocprofv3 --pmc $grp --output-format csv -d $OC/pmc$i -- $BC > $OC/pmc$i.log 2>&1
  done
  python3 profiles/summarize.py $OC > $OC/summary.txt 2>&1
  tail -40 $OC/summary.txt
done
