#!/bin/bash
# round 6 final, part 1: the whole GPU suite, smoke, every bench line, the N > 1 rehearsals
set -u
O=gpurun_out/r6final
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
rc=$?
tail -4 $O/pytest.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1 || { tail -20 $O/smoke.txt; exit 1; }
tail -4 $O/smoke.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err || { tail -20 $O/bench_driver_flags.err; exit 1; }
echo "driver flags done" >> $O/progress.txt
timeout -k 10 300 python bench.py > $O/bench_config2.json 2> $O/bench_config2.err || { tail -20 $O/bench_config2.err; exit 1; }
echo "config 2 done" >> $O/progress.txt
for cfg in 3 4 5 default; do
  timeout -k 10 500 python bench.py --config $cfg > $O/bench_config$cfg.json 2> $O/bench_config$cfg.err || { tail -20 $O/bench_config$cfg.err; exit 1; }
  echo "config $cfg done" >> $O/progress.txt
done
for N in 4 6; do
  timeout -k 10 500 python bench.py --gpus $N --backend gloo --same-device --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_${N}ranks_one_device.json 2> $O/bench_${N}ranks.err || { tail -20 $O/bench_${N}ranks.err; exit 1; }
  echo "$N ranks done" >> $O/progress.txt
done
timeout -k 10 600 python tools/rehearse_ranks.py --ranks 8 --config 2 --steps 20 --warmup 5 --again > $O/rehearse_8ranks_config2.json 2> $O/rehearse8.err || { tail -20 $O/rehearse8.err; exit 1; }
timeout -k 10 600 python tools/rehearse_ranks.py --ranks 8 --config 2 --steps 20 --warmup 5 --again --busy-ms 0 > $O/rehearse_8ranks_config2_cold.json 2>> $O/rehearse8.err || { tail -20 $O/rehearse8.err; exit 1; }
make -s -C examples render_bands && python tools/write_scene_bin.py write config2 /tmp/c2.bin > $O/render_bands_8_config2.txt && \
  timeout -k 10 600 examples/render_bands /tmp/c2_bands8.f32 8 4 /tmp/c2.bin >> $O/render_bands_8_config2.txt 2>&1 && \
  python tools/write_scene_bin.py check config2 /tmp/c2_bands8.f32 >> $O/render_bands_8_config2.txt 2>&1 || { tail $O/render_bands_8_config2.txt; exit 1; }
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6final/bench_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    print("%-36s value %-10s %s  ms/step %-9s sec/frame %-8s gather %-5s kernel %s frac %s first_frame %s stale %s" % (
        f.split("/")[-1], d["value"], d["unit"], d["ms_per_step"], d.get("sec_to_converged_frame"), d.get("gather_matches_single_gpu"),
        r.get("kernel"), r.get("frac"), d.get("first_frame_ms"), (r.get("counters") or {}).get("stale")))
for f in sorted(glob.glob("gpurun_out/r6final/rehearse_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], "match", d["gather_matches_single_gpu"], "eff", d["predicted"]["strong_scaling_efficiency"], [x["kernel_ms"] for x in d["per_rank"]], d["single_context_same_box"]["kernel_ms"])
d = json.loads(open("gpurun_out/r6final/bench_driver_flags.json").read().strip().splitlines()[-1])
print(json.dumps(d["first_frame"], indent=1)[:1800])
PY
tail -3 $O/render_bands_8_config2.txt
