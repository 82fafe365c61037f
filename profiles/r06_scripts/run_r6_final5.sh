#!/bin/bash
# round 6 final, last part: every committed bench line and rehearsal once more on the final build (records of this build attached)
set -u
O=gpurun_out/r6final5
mkdir -p $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1 || { tail -20 $O/smoke.txt; exit 1; }
tail -4 $O/smoke.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err || { tail -20 $O/bench_driver_flags.err; exit 1; }
timeout -k 10 300 python bench.py > $O/bench_config2.json 2> $O/bench_config2.err || { tail -20 $O/bench_config2.err; exit 1; }
for cfg in 3 4 5 default; do
  timeout -k 10 500 python bench.py --config $cfg > $O/bench_config$cfg.json 2> $O/bench_config$cfg.err || { tail -20 $O/bench_config$cfg.err; exit 1; }
done
for N in 4 6; do
  timeout -k 10 500 python bench.py --gpus $N --backend gloo --same-device --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_${N}ranks_one_device.json 2> $O/bench_${N}ranks.err || { tail -20 $O/bench_${N}ranks.err; exit 1; }
done
timeout -k 10 600 python tools/rehearse_ranks.py --ranks 8 --config 2 --steps 20 --warmup 5 --again > $O/rehearse_8ranks_config2.json 2> $O/rehearse8.err || { tail -20 $O/rehearse8.err; exit 1; }
timeout -k 10 600 python tools/rehearse_ranks.py --ranks 8 --config 2 --steps 20 --warmup 5 --again --busy-ms 0 > $O/rehearse_8ranks_config2_cold.json 2>> $O/rehearse8.err || { tail -20 $O/rehearse8.err; exit 1; }
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6final5/bench_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    c = r.get("counters") or {}
    print("%-30s value %-10s ms/step %-8s sec/frame %-7s gather %-5s %s frac %s | stale %s issue %s lanes %s fp32 %s prior/this %s | first %s" % (
        f.split("/")[-1], d["value"], d["ms_per_step"], d.get("sec_to_converged_frame"), d.get("gather_matches_single_gpu"),
        r.get("kernel"), r.get("frac"), c.get("stale"), c.get("valu_issue_frac"), c.get("lane_utilisation"), c.get("fp32_flop_frac"), c.get("prior_over_this_run_kernel_ms"), d.get("first_frame_ms")))
for f in sorted(glob.glob("gpurun_out/r6final5/rehearse_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], "match", d["gather_matches_single_gpu"], "eff", d["predicted"]["strong_scaling_efficiency"], [x["kernel_ms"] for x in d["per_rank"]], d["single_context_same_box"]["kernel_ms"])
PY
