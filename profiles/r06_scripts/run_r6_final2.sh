#!/bin/bash
# round 6 final, part 2..4: rocprofv3 kernel traces + PMC passes (profiles/collect.sh); $1 = main | c34 | c5d
set -u
case "${1:-main}" in
  main) PT_COLLECT_CONFIGS="" timeout -k 10 1150 bash profiles/collect.sh r06_final > gpurun_out/collect_r06_final_main.txt 2>&1 ;;
  c34)  PT_COLLECT_MAIN=0 PT_COLLECT_CONFIGS="3 4" timeout -k 10 1150 bash profiles/collect.sh r06_final > gpurun_out/collect_r06_final_c34.txt 2>&1 ;;
  c5d)  PT_COLLECT_MAIN=0 PT_COLLECT_CONFIGS="5 default" timeout -k 10 1150 bash profiles/collect.sh r06_final > gpurun_out/collect_r06_final_c5d.txt 2>&1 ;;
esac
echo "rc $?"
tail -5 gpurun_out/collect_r06_final_${1:-main}.txt
find gpurun_out/prof_r06_final* -name "*_counter_collection.csv" -size +4M -delete
find gpurun_out/prof_r06_final* -name "*.db" -delete 2>/dev/null
du -sh gpurun_out/prof_r06_final* | tail -8
