#!/bin/bash
# round 6, second call: the whole -m gpu suite on the new build, then the N > 1 rehearsals and the band launch's sweeps
set -u
O=gpurun_out/r6b; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu --durations=12 > $O/pytest_gpu.txt 2>&1 || { tail -40 $O/pytest_gpu.txt; exit 1; }
tail -16 $O/pytest_gpu.txt
# (a) bench.py's N > 1 path with as many rank processes as the box admits on its card (six), uneven shares at 4, the driver's flags
for N in 4 6; do
  timeout -k 10 500 python bench.py --gpus $N --backend gloo --same-device --steps 20 --warmup 5 > $O/bench_${N}ranks_one_device.json 2> $O/bench_${N}ranks.err || { tail -20 $O/bench_${N}ranks.err; exit 1; }
  python - $O/bench_${N}ranks_one_device.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("bench --gpus %d same-device: gather_matches_single_gpu %s, segments_match %s, per-rank kernel ms %s" % (d["n_gpus"], d["gather_matches_single_gpu"], d["gather_check"]["segments_match"], d["per_rank_render_kernel_ms"]))
PY
done
# ... the eight-rank shape in one process (eight contexts one after the other, assembled like the collective's layout)
timeout -k 10 600 python tools/rehearse_ranks.py --ranks 8 --config 2 --steps 20 --warmup 5 > $O/rehearse_8ranks_config2.json 2> $O/rehearse8.err || { tail -20 $O/rehearse8.err; exit 1; }
python - $O/rehearse_8ranks_config2.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("rehearse 8 ranks: rows", d["rows_per_rank"], "match", d["gather_matches_single_gpu"], d["gather_check"]["segments_match"], "predicted", d["predicted"]["strong_scaling_efficiency"], "per-rank wall", [x["wall_ms"] for x in d["per_rank"]], "single", d["single_context_same_box"]["wall_ms"])
PY
# ... and from plain C: eight contexts in one process at config 2's size
make -s -C examples render_bands && python tools/write_scene_bin.py write config2 /tmp/c2.bin && \
  timeout -k 10 600 examples/render_bands /tmp/c2_bands8.f32 8 4 /tmp/c2.bin > $O/render_bands_8_config2.txt 2>&1 && \
  python tools/write_scene_bin.py check config2 /tmp/c2_bands8.f32 >> $O/render_bands_8_config2.txt 2>&1
tail -4 $O/render_bands_8_config2.txt
# (b) one rank's 1/8-frame launch under the driver's flags (80 passes of 16 spp, rank 0 of 8, 4-row bands): what wins?
S=$O/band_sweep.txt; : > $S
F="SW_BAND=4:0:8,SW_PASSES=80,SW_SPP=16"
run() { echo "== $1  $2" >> $S; timeout -k 10 600 python tools/sweep_knobs.py "$1" "$2" >> $S 2>&1 || exit 1; }
run "$F" "PT_PER_CU=3,2,3"
run "$F" "PT_QUEUE_CHUNK=32,64,128,256,512"
run "$F" "SW_CARRY=8,12,16,20"
run "$F" "PT_BVH_BLOCK=256,512,1024"
run "$F,PT_QUEUE_STATIC=1" "PT_QUEUE_GROUPED=1,0"
run "$F,PT_PER_CU=2" "PT_QUEUE_CHUNK=64,128,256"
run "$F" "PT_COST_FEEDBACK=1,0"
cat $S
