#!/bin/bash
# round 6, third call: the flight's timing test, cold clocks in the 8-rank rehearsal, the non-temporal item store A/B + its WRITE_SIZE
set -u
O=gpurun_out/r6c; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "cliff" > $O/pytest_cliff.txt 2>&1 || { tail -60 $O/pytest_cliff.txt; exit 1; }
grep -A40 "flight position" $O/pytest_cliff.txt | head -50
for B in 0 200; do
  timeout -k 10 600 python tools/rehearse_ranks.py --ranks 8 --config 2 --steps 20 --warmup 5 --busy-ms $B --again > $O/rehearse_8ranks_busy$B.json 2> $O/rehearse_busy$B.err || { tail -20 $O/rehearse_busy$B.err; exit 1; }
  python - $O/rehearse_8ranks_busy$B.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("busy", d["untimed_busy_ms_before_each_timed_region"], "match", d["gather_matches_single_gpu"], "eff", d["predicted"]["strong_scaling_efficiency"], "per-rank kernel", [x["kernel_ms"] for x in d["per_rank"]], "rank 0 again", d["rank_0_again_on_a_warm_device"]["kernel_ms"], "single", d["single_context_same_box"]["kernel_ms"])
PY
done
AB_CASES=c2grid,c5grid16,c4small,defsmall,def1small,c2rank8_0 timeout -k 10 900 python tools/ab_kernels.py ray_tracer_webgl_amd/libptrace.so build_ab/libptrace_nt.so 3 > $O/ab_nt_store.txt 2>&1
cat $O/ab_nt_store.txt
for rep in 1 2; do for L in ray_tracer_webgl_amd/libptrace.so build_ab/libptrace_nt.so; do
  PT_LIB=$L timeout -k 10 300 python bench.py --config default --no-cpu-baseline --no-extra-legs --frames 1920 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$L frames/s', d['value'], 'group trace ms', d['roofline']['avg_launch_ms'])" >> $O/ab_nt_frames.txt
done; done
cat $O/ab_nt_frames.txt
for L in base nt; do
  LIB=ray_tracer_webgl_amd/libptrace.so; [ $L = nt ] && LIB=build_ab/libptrace_nt.so
  PT_LIB=$LIB timeout -k 5 300 rocprofv3 --pmc WRITE_SIZE FETCH_SIZE --output-format csv -d $O/pmc_c5_$L/pmc1 -- python3 tools/pmc_config5.py > $O/pmc_c5_$L.log 2>&1
  python3 profiles/pmc_dispatches.py $O/pmc_c5_$L pt_trace > $O/pmc_c5_$L.txt 2>&1; tail -8 $O/pmc_c5_$L.txt
done
rm -rf $O/pmc_c5_base/pmc1 $O/pmc_c5_nt/pmc1
