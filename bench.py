#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on BASELINE.json's config.

  metric   Mray/s (ray segments = hit_world invocations per second) at 1920x1080
  workload config 2: Shirley cover scene (484 spheres), 1920x1080, 50 bounces, 64-spp passes;
           the default 16 steps are exactly the config's 1024 spp "converged frame"
  step     one pass of the hot path = 64 samples for every pixel (u_time = step index); steps are
           enqueued `--passes-per-launch` at a time through pt_render_passes (one persistent
           kernel launch works through all their (pixel, pass) items from one queue)
  N > 1    strong scaling: the same frame, rows dealt to ranks in interleaved 8-row bands, no
           collective while rendering, ONE all_gather of the radiance buffers (RCCL over xGMI)
           at the end of the timed region

Prints ONE JSON line on rank 0.  `roofline` prices the path-tracing kernel against the FP32
vector peak (the path has no dense contraction and ~1e5 FLOP per HBM byte) and
also states the HBM figure north_star asks for; `cpu_baseline` is the CPU oracle timed on this
host's cores over a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: 256 CU x 4 SIMD x 32 lanes x 2 x 2.4 GHz
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FLOP_PER_SPHERE_TEST = 20      # SURVEY.md §8d algorithmic work unit


def usable_cores():
    """Threads this process may actually run at once: min(affinity, cgroup CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--passes-per-launch", type=int, default=16)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp-per-step", type=int, default=64)
    ap.add_argument("--max-depth", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-list-walk", action="store_true",
                    help="skip the extra (untimed-region) launch that measures the reference's linear list walk")
    ap.add_argument("--cpu-strip", type=int, default=960, help="width of the CPU baseline's column strip")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --same-device rehearses the N>1 path on a one-GPU box (not a benchmark)")
    ap.add_argument("--same-device", action="store_true", help="every rank uses cuda:0 (rehearsal only)")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the RCCL code path even with one rank (rehearsal of the collectives on a one-GPU box)")
    args = ap.parse_args()

    import numpy as np
    import torch

    from ray_tracer_webgl_amd import abi, dist as ptdist, scenes
    from ray_tracer_webgl_amd.tracer import PathTracer

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    use_dist = world > 1 or args.force_dist
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if use_dist:
        import torch.distributed as dist

        if args.force_dist and "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    def gather(t):
        if args.backend == "gloo":  # rehearsal path: collectives on host copies
            return ptdist.gather_rows(t.cpu(), p.height, band_rows, rank, world)
        return ptdist.gather_rows(t, p.height, band_rows, rank, world)

    sc = scenes.config2(args.width, args.height, args.spp_per_step, args.steps, args.max_depth)
    p = sc.params.copy()
    band_rows = 8
    p.band_rows, p.band_index, p.band_count = ptdist.band_of(rank, world, band_rows)

    pt = PathTracer(p.width, p.height, device=local_rank, use_torch=True)
    pt.set_spheres(sc.spheres)
    pt.set_params(p)
    ppl = max(1, min(args.passes_per_launch, max(args.steps, 1)))
    pt.reserve_passes(ppl)
    # set-up, like reserving the workspace: settle how the scan reads the sphere list (LDS walk
    # or scalar-load walk, bit-identical images) by measuring both once on this scene
    pt.tune(ppl)

    def sync_all():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    def run_steps(k, first_time):
        done = 0
        while done < k:
            n = min(ppl, k - done)
            q = p.copy()
            q.time = float(first_time + done)  # pass j of this launch uses u_time = time + j
            pt.set_params(q)
            pt.render_passes(n)  # asynchronous on torch's current stream
            done += n

    # warmup (untimed), then clear accumulation and statistics
    run_steps(args.warmup, 1000.0)
    if use_dist:
        gather(pt.accum_tensor)  # also sets up the RCCL channels outside the timed region
    sync_all()
    pt.reset()

    sync_all()
    t0 = time.perf_counter()
    run_steps(args.steps, 0.0)
    full = gather(pt.accum_tensor) if use_dist else pt.accum_tensor
    sync_all()
    t1 = time.perf_counter()

    st = pt.stats()
    cdev = "cuda" if args.backend == "nccl" else "cpu"
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=cdev)
    totals = torch.tensor([float(st.segments), float(st.render_kernel_ms), float(st.render_launches)],
                          dtype=torch.float64, device=cdev)
    if use_dist:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
        seg_all = totals[:1].clone()
        dist.all_reduce(seg_all, op=dist.ReduceOp.SUM)
    else:
        seg_all = totals[:1]
    wall = float(elapsed.item())
    segments = float(seg_all.item())

    if rank == 0:
        n_sph = len(sc.spheres)
        mrays = segments / wall / 1e6
        # dominant kernel: pt_trace_kernel, timed with HIP events on its launch stream (rank 0)
        avg_ms = st.render_kernel_ms / max(st.render_launches, 1)
        seg_per_launch = st.segments / max(st.render_launches, 1)
        flop_per_launch = FLOP_PER_SPHERE_TEST * n_sph * seg_per_launch
        achieved_tf = flop_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        local_pix = st.local_rows * p.width
        # algorithmic HBM bytes per launch of the trace kernel: one 16-B slab store per
        # (pixel, pass) item + the scene (48 B/sphere); the fold kernel's traffic is separate
        passes_per_launch_avg = args.steps / max(st.render_launches, 1)
        hbm_bytes = int(passes_per_launch_avg * local_pix * 16 + n_sph * 48)
        traffic = None
        executed = None
        kernel_name = {abi.PT_GEOM_LDS: "pt_trace_kernel", abi.PT_GEOM_SCALAR: "pt_trace_kernel_scalar",
                       abi.PT_GEOM_BVH: "pt_trace_kernel_bvh"}.get(st.geometry_path, "?")
        prof = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(prof):
            try:
                rec = json.load(open(prof))
                if rec.get("kernel") == kernel_name:  # counters of another kernel say nothing about this one
                    traffic = rec.get("pt_trace_kernel_hbm_bytes_per_launch")
                    if "valu_insts_per_launch" in rec:
                        executed = {
                            "valu_instructions_per_launch": rec["valu_insts_per_launch"],
                            "valu_issue_frac": round(rec["valu_issue_frac"], 4),
                            "source": "profiles/pmc_traffic.json (rocprofv3 --pmc SQ_INSTS_VALU, GRBM_GUI_ACTIVE of this command)",
                            "note": "wave64 VALU instructions issued x 2 cycles / (1024 SIMDs x kernel cycles): the share of "
                                    "the chip's vector issue slots the kernel really fills",
                        }
            except Exception:
                traffic = None
        roofline = {
            "kernel": kernel_name,
            "geometry_path": abi.GEOM_NAMES.get(st.geometry_path, "?") + (" (autotuned)" if st.geometry_tuned else ""),
            "bound": "valu",
            "achieved": round(achieved_tf, 3),
            "peak": FP32_VALU_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": round(achieved_tf / FP32_VALU_PEAK_TFLOPS, 4),
            "traffic": traffic,
            "traffic_source": "profiles/pmc_traffic.json (rocprofv3 --pmc WRITE_SIZE + 2*FETCH_SIZE of this command)" if traffic else None,
            "avg_launch_ms": round(avg_ms, 4),
            "launches": int(st.render_launches),
            "flop_per_launch": flop_per_launch,
            "note": ("fp32 vector peak: no dense contraction on this path; ALGORITHMIC work = 20 FLOP per ray-sphere test x %d "
                     "spheres x segments (SURVEY.md 8d), i.e. what the reference's loop does for these rays" % n_sph) +
                    ("; the hierarchy walk skips tests whose outcome is provably 'miss', so frac can exceed 1 - see `executed`"
                     if st.geometry_path == abi.PT_GEOM_BVH else ""),
            "executed": executed,
            "hbm": {
                "achieved": round(hbm_bytes / (avg_ms * 1e-3) / 1e9, 3) if avg_ms > 0 else 0.0,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(hbm_bytes / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6) if avg_ms > 0 else 0.0,
                "algorithmic_bytes_per_launch": hbm_bytes,
            },
        }
        # Beside the headline (fastest path, normally the hierarchy walk): the same frame with the
        # reference's own algorithm, the linear walk over the whole list, after the timed region.
        list_walk = None
        if st.geometry_path == abi.PT_GEOM_BVH and world == 1 and not args.no_list_walk:
            pt.set_geometry_path(abi.PT_GEOM_SCALAR)
            pt.reset()
            run_steps(min(args.steps, ppl), 0.0)  # settle the tile order for this path
            sync_all()
            pt.reset()
            sync_all()
            tl0 = time.perf_counter()
            run_steps(args.steps, 0.0)
            sync_all()
            tl1 = time.perf_counter()
            sl = pt.stats()
            l_ms = sl.render_kernel_ms / max(sl.render_launches, 1)
            l_tf = FLOP_PER_SPHERE_TEST * n_sph * (sl.segments / max(sl.render_launches, 1)) / (l_ms * 1e-3) / 1e12
            list_walk = {
                "kernel": "pt_trace_kernel_scalar",
                "value": round(sl.segments / (tl1 - tl0) / 1e6, 3),
                "unit": "Mray/s",
                "sec_to_converged_frame": round(tl1 - tl0, 4),
                "avg_launch_ms": round(l_ms, 4),
                "roofline_frac": round(l_tf / FP32_VALU_PEAK_TFLOPS, 4),
                "segments": int(sl.segments),
                "note": "every sphere tested for every ray, as static/shader.frag:175-196 does; same image bits, same segment count",
            }
            pt.set_geometry_path(abi.PT_GEOM_AUTO)
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            from oracle import oracle

            cores = usable_cores()
            strip = min(args.cpu_strip, p.width)
            x0 = (p.width - strip) // 2
            cp = sc.params.copy()
            tc0 = time.perf_counter()
            _, cseg = oracle.render(sc.spheres, cp, 1, window=(x0, x0 + strip, 0, p.height), nthreads=cores)
            tc1 = time.perf_counter()
            cpu = {
                "value": round(cseg / (tc1 - tc0) / 1e6, 3),
                "unit": "Mray/s",
                "cores": cores,
                "kind": "port",
                "sample": "one %d-spp pass of the centre %dx%d column strip of the same frame (%d segments, %.1f s)"
                          % (args.spp_per_step, strip, p.height, cseg, tc1 - tc0),
            }
        out = {
            "metric": "Mray/s (ray segments/s) at %dx%d" % (p.width, p.height),
            "value": round(mrays, 3),
            "unit": "Mray/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(wall / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "config2: Shirley cover scene (%d spheres), %dx%d, %d bounces, %d spp/step x %d steps = %d spp"
                            % (n_sph, p.width, p.height, args.max_depth, args.spp_per_step, args.steps,
                               args.spp_per_step * args.steps),
                "partition": "%d rank(s), interleaved %d-row bands, one all_gather at frame end" % (world, band_rows),
                "passes_per_launch": ppl,
            },
            "sec_to_converged_frame": round(wall, 4),
            "segments": int(segments),
            "nominal_mray_s": round(p.width * p.height * args.spp_per_step * args.steps * args.max_depth / wall / 1e6, 1),
            "roofline": roofline,
            "list_walk": list_walk,
            "cpu_baseline": cpu,
        }
        if cpu:
            out["gpu_over_cpu"] = round(mrays / cpu["value"], 1) if cpu["value"] else None
        print(json.dumps(out), flush=True)
        assert tuple(full.shape) == (p.height, p.width, 4)
    pt.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
