#!/bin/bash
set -u
O=gpurun_out/r6e; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_properties.py -x -q -m gpu -s -k "fixed_slice or config5_as_bench" > $O/pytest_tune.txt 2>&1 || { tail -60 $O/pytest_tune.txt; exit 1; }
grep -E "pt_tune kept|passed|failed" $O/pytest_tune.txt
for cfg in 5 2; do
  timeout -k 10 400 python bench.py --config $cfg --no-cpu-baseline --no-list-walk > $O/bench_config$cfg.json 2> $O/bench_config$cfg.err || { tail -20 $O/bench_config$cfg.err; exit 1; }
  python - $O/bench_config$cfg.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(d["config"]["workload"][:40], "value", d["value"], "ms/step", d["ms_per_step"], "sec/frame", d["sec_to_converged_frame"], "match", d["gather_matches_single_gpu"], "first_frame", d["first_frame"]["ms"], "autotune", d["first_frame"]["autotune_ms"], "busy", d["untimed_busy_ms_before_the_timed_region"], "counters", (d["roofline"]["counters"] or {}).get("stale"))
PY
done
for L in base nt; do
  LIB=ray_tracer_webgl_amd/libptrace.so; [ $L = nt ] && LIB=build_ab/libptrace_nt.so
  PT_LIB=$LIB timeout -k 5 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_c5_$L/pmc1 -- python3 tools/pmc_config5.py > $O/pmc_c5_$L.log 2>&1 || { tail -5 $O/pmc_c5_$L.log; exit 1; }
  python3 profiles/pmc_dispatches.py $O/pmc_c5_$L pt_trace > $O/pmc_c5_$L.txt 2>&1; tail -6 $O/pmc_c5_$L.txt
done
rm -rf $O/pmc_c5_base/pmc1 $O/pmc_c5_nt/pmc1
# far rays handed to the wave in the LDS-staged grid build too?
AB_CASES=c2grid,c2band8,c2rank8_0,c2rank8_5 timeout -k 10 900 python tools/ab_kernels.py ray_tracer_webgl_amd/libptrace.so build_ab/libptrace_ho.so 3 > $O/ab_handover.txt 2>&1
cat $O/ab_handover.txt
PT_LIB=build_ab/libptrace_ho.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "cliff or flies_out or tune_measures or far_and_nan" > $O/pytest_handover.txt 2>&1 || { tail -40 $O/pytest_handover.txt; exit 1; }
grep -E "as built:|passed|failed|\(0, |\(11, |\(20, |\(29, " $O/pytest_handover.txt | cut -c1-300
