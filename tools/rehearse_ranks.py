"""Dev tool: an N-rank strong-scaling run of bench.py REHEARSED ON ONE DEVICE IN ONE PROCESS.

    python tools/rehearse_ranks.py [--ranks 8] [--config 2] [--steps 20] [--warmup 5] [--band-rows 4]

Why one process: a GPU box of this pool admits at most six processes on its card, so `bench.py --gpus 8 --same-device`
(eight rank processes) cannot run there; `--gpus 4` and `--gpus 6` can, and do (profiles/r06_bench_*ranks_one_device.json).
What the eight-rank run adds to them — the UNEVEN shares (1080 rows in 4-row bands over 8 ranks: six ranks with 34 bands,
two with 33, so two ranks send a band of padding), the full-size gather layout and every rank's own launch shape — is
rehearsed here: one pt_ctx per rank (PtParams.band_*), each driven through exactly the per-rank sequence bench.py runs under
the driver's flags (pt_tune, warm-up steps, the timed steps in launches of 16 x N steps, then the committed 1024-spp workload
once more), the ranks' buffers put together by dist.assemble_rows (the layout and the permutation the all_gather path uses),
the frame hashed against tests/golden/full_frame_digests.json.  The ranks run ONE AFTER THE OTHER on the one device, so the
line holds each rank's own device time, and a PREDICTION of the N-GPU step time (the slowest rank; the gather is 33 MB over
xGMI: ~0.1 ms) — a prediction, not a measurement: the driver's 8-GPU run is the only measurement there can be.
"""
import argparse
import importlib.util
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--config", default="2", choices=["2", "3", "5"])
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--band-rows", type=int, default=4)
    ap.add_argument("--busy-ms", type=float, default=None,
                    help="untimed launches before each rank's timed steps until the device has been busy this long (default: bench.py's own CLOCK_WARMUP_MS; "
                         "0 = only the --warmup steps: a rank that starts on an idle device then shows what the clocks' ramp costs a 20-ms region)")
    ap.add_argument("--again", action="store_true", help="run rank 0 once more at the end (a warm device): is its first figure the clocks or the rank?")
    args = ap.parse_args()

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    bench.ipc_env()
    import torch

    from ray_tracer_webgl_amd import abi, dist as ptdist, scenes
    from ray_tracer_webgl_amd.tracer import PathTracer

    cfg = bench.BENCH_CONFIGS[args.config]
    world, band_rows = args.ranks, args.band_rows
    width, height, converged = cfg["width"], cfg["height"], cfg["spp"]
    spp_pass, pps = 16, 4
    steps, spp_step = bench.plan_steps(converged, spp_pass, pps, args.steps)
    steps_frame, _ = bench.plan_steps(converged, spp_pass, pps, None)
    spl = max(1, min(16 * world, steps))  # bench.py: a rank of N takes 16 x N steps per launch
    ppl = spl * pps
    sc = getattr(scenes, cfg["scene"])(width, height, spp_pass, steps * pps, 50)

    def run_steps(pt, p, k, first_time, spl_use):
        done = 0
        while done < k:
            n = min(spl_use, k - done)
            q = p.copy()
            q.time = float(first_time)
            q.first_pass = done * pps
            pt.set_params(q)
            pt.render_passes(n * pps)
            done += n

    busy_ms = bench.CLOCK_WARMUP_MS if args.busy_ms is None else args.busy_ms
    per_rank, parts, seg_timed, seg_frame, again = [], [], 0, 0, None
    turns = list(range(world + 1)) + ([world + 1] if args.again else [])
    for rank in turns:  # turn `world` is the whole frame on one context: the N = 1 reference of the same box
        n_ranks = 1 if rank == world else world
        r = rank if rank < world else 0
        p = sc.params.copy()
        p.band_rows, p.band_index, p.band_count = ptdist.band_of(r, n_ranks, band_rows)
        p.time_step = abi.PT_TIME_STEP_DECORRELATED
        pt = PathTracer(width, height, device=0, use_torch=True)
        pt.set_spheres(sc.spheres)
        pt.set_params(p)
        spl_here = spl if rank != world else max(1, min(16, steps))
        ppl_here = spl_here * pps
        pt.reserve_passes(ppl if rank != world else ppl_here)
        pt.tune(min(ppl_here, 8))
        run_steps(pt, p, args.warmup, 1000.0, spl_here)
        torch.cuda.synchronize()
        warm_ms = bench.keep_busy(pt, lambda k: run_steps(pt, p, k, 1000.0, spl_here), args.warmup, busy_ms) if busy_ms > 0 else 0.0
        pt.reset()
        t0 = time.perf_counter()
        run_steps(pt, p, steps, 0.0, spl_here)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        st = pt.stats()
        row = {"rank": r, "of": n_ranks, "rows": int(st.local_rows), "wall_ms": round(wall * 1e3, 3), "kernel_ms": round(st.render_kernel_ms, 3),
               "launches": int(st.render_launches), "segments": int(st.segments), "passes_per_launch": ppl if rank != world else ppl_here,
               "untimed_busy_ms_before": round(warm_ms, 1)}
        if rank == world + 1:
            again = row
        elif rank < world:
            seg_timed += st.segments
            # the committed workload once more, as bench.py does after a timed region that is not the committed frame
            pt.reset()
            run_steps(pt, p, steps_frame, 0.0, spl_here)
            torch.cuda.synchronize()
            seg_frame += pt.stats().segments
            parts.append(pt.accum_tensor[: abi.local_rows(height, band_rows, r, world)].clone())
            per_rank.append(row)
        else:
            single = row
        pt.close()
        del pt
        torch.cuda.empty_cache()
    full = ptdist.assemble_rows(parts, height, band_rows)
    want = bench.load_digests().get(cfg["digest"])
    got = bench.frame_digest(full)
    slowest = max(x["wall_ms"] for x in per_rank)
    shares = [x["rows"] for x in per_rank]
    out = {
        "what": "one-device, one-process REHEARSAL of bench.py --gpus %d --steps %d --warmup %d (config %s): the ranks run one after the other" % (world, args.steps, args.warmup, args.config),
        "ranks": world, "band_rows": band_rows, "rows_per_rank": shares,
        "padded_rows_per_rank": ptdist.band_layout(height, band_rows, world)[0],
        "steps": steps, "spp": steps * spp_step, "steps_per_launch": spl, "passes_per_launch": ppl,
        "per_rank": per_rank,
        "single_context_same_box": single,
        "rank_0_again_on_a_warm_device": again,
        "untimed_busy_ms_before_each_timed_region": busy_ms,
        "gather_check": {"sha256": got, "expected": want["sha256"] if want else None, "key": cfg["digest"],
                         "matches": bool(want) and got == want["sha256"],
                         "segments_match": bool(want) and int(seg_frame) == int(want["segments"]),
                         "frame": "the committed %d-spp workload rendered by every rank after its timed steps, assembled by dist.assemble_rows" % converged},
        "gather_matches_single_gpu": bool(want) and got == want["sha256"],
        "predicted": {
            "ms_per_step": round(slowest / steps, 4),
            "mray_s": round(seg_timed / (slowest * 1e-3) / 1e6, 1),
            "strong_scaling_efficiency": round(single["wall_ms"] / world / slowest, 4),
            "rank_imbalance": round(slowest / (sum(x["wall_ms"] for x in per_rank) / world), 4),
            "is": "slowest rank's wall time for the timed steps (host clock around its launches, one rank alone on the device) against the "
                  "single context's / N on the same box; the gather (%.1f MB over xGMI, ~0.1 ms) and the barrier are not in it.  A PREDICTION."
                  % (full.numel() * 4 / 1e6),
        },
    }
    print(json.dumps(out), flush=True)
    return 0 if out["gather_matches_single_gpu"] else 1


if __name__ == "__main__":
    raise SystemExit(main())
