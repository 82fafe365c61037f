#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on BASELINE.json's config.

  metric   Mray/s (ray segments = hit_world invocations per second) at 1920x1080
  workload config 2: Shirley cover scene (484 spheres), 1920x1080, 50 bounces, 1024 spp
  step     one pass of the hot path over one batch = 64 samples for every pixel, rendered as
           `--passes-per-step` (4) seeds of `--spp-per-pass` (16) samples each (u_time = (4 * step + j) x 0.3618:
           the reference also accumulates many low-spp frames with distinct u_time, README.md:6);
           the default 16 steps are exactly the config's 1024 spp "converged frame".  Steps are
           enqueued `--steps-per-launch` at a time through pt_render_passes (one persistent kernel
           launch works through all their (pixel, pass) items from one queue).
  N > 1    the image's rows are dealt to the ranks in interleaved 4-row bands (the path shards by
           pixel: no collective while rendering), ONE all_gather of the radiance buffers (RCCL over
           xGMI) at the end of the timed region.  Default `--scaling weak`: per-GPU work is fixed —
           a rank renders 1/N of the rows for N x 4 passes per step, so a step is N x 64 spp of the
           whole frame (the same pass seeds a single GPU would use for that many passes; the
           gathered image is bit-identical to the single-GPU one).  `--scaling strong` keeps the
           frame fixed (4 passes per step whatever N).  Either way the line carries `fixed_frame`:
           the config's own 1024-spp frame split over the N ranks, timed after the main region
           with the same barriers (= the strong-scaling point; `sec_to_converged_frame` is that).
           `python bench.py --gpus N` from a bare shell starts its N ranks itself (the parent never
           touches the GPU); under torch.distributed.run it uses the ranks it is given.

Prints ONE JSON line on rank 0.  `roofline` prices the path-tracing kernel against the FP32
vector peak (the path has no dense contraction and ~1e5 FLOP per HBM byte):
`frac` is EXECUTED lane-level fp32 work (from the kernel's own tallies, measuring twin run after
the timed region) over the peak, always <= 1; what the reference's linear loop would have done for
the same rays is `algorithmic_speedup_vs_list_walk`, and the linear walk itself is timed beside
it (`list_walk`).  Everything is normalised PER PASS so any --steps gives the same ratios.
`cpu_baseline` is the CPU oracle (rebuilt -O3 -march=native on this host) timed on this host's
cores over a bounded sample of the same workload.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: 256 CU x 4 SIMD x 32 lanes x 2 x 2.4 GHz
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FLOP_PER_SPHERE_TEST = 20      # SURVEY.md §8d algorithmic work unit (one literal ray-sphere test)
# lane-level fp32 operations per executed unit of the walk kernels
FLOP_PER_NODE_STEP = 20        # hierarchy: 6 fma + 6 min/max + 2 reductions over one box
FLOP_PER_CELL_STEP = 10        # grid: min3, two compares, three adds, bookkeeping
FLOP_PER_LEAF_ROUND = 4 * FLOP_PER_SPHERE_TEST
FLOP_PER_EXACT = 10            # sqrt, two divisions, sums, compares
FLOP_PER_SEGMENT_SHADE = 150   # scatter + RNG + camera share
CONVERGED_SPP = 1024           # BASELINE config 2


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


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (fresh
    processes, one per GPU, rendezvous on 127.0.0.1), relay rank 0's JSON line, fail if any rank
    fails.  This parent makes no GPU call and replaces no process."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr, text=True))
    out0, _ = procs[0].communicate()
    rc = procs[0].returncode
    deadline = time.time() + 120
    for p in procs[1:]:
        if rc != 0:
            p.terminate()  # exactly the children started above
        try:
            p.wait(timeout=max(1.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
        rc = rc or p.returncode
    for line in (out0 or "").splitlines():  # stdout carries the JSON line only (gloo chats on stdout)
        (sys.stdout if line.startswith("{") else sys.stderr).write(line + "\n")
    sys.stdout.flush()
    return rc


def spawn_selftest(mode):
    """What a rank does under --spawn-selftest: no GPU, no rendering — join the gloo group the
    parent set up, prove every rank is there, let rank 0 print the line the parent relays."""
    import torch
    import torch.distributed as dist

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t)
    if mode == "fail" and rank == world - 1:
        os._exit(3)  # a rank that dies after the rendezvous: the parent must report failure
    if rank == 0:
        print(json.dumps({"selftest": True, "ranks": dist.get_world_size(), "rank_sum": float(t.item()),
                          "local_rank": int(os.environ["LOCAL_RANK"])}), flush=True)
    if mode == "ok":
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--steps-per-launch", type=int, default=16)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp-per-pass", type=int, default=16)
    ap.add_argument("--passes-per-step", type=int, default=4)
    ap.add_argument("--max-depth", type=int, default=50)
    ap.add_argument("--band-rows", type=int, default=4,
                    help="N > 1: rows per interleaved band (4: rank shares of the work within 1 %% of each other at N = 8)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = passes per step grow with N (fixed work per GPU); strong = the same frame for every N")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-list-walk", action="store_true",
                    help="skip the extra (untimed-region) launch that measures the reference's linear list walk")
    ap.add_argument("--no-work-count", action="store_true", help="skip the measuring-twin launch (roofline.frac = null)")
    ap.add_argument("--cpu-strip", type=int, default=960, help="width of the CPU baseline's column strip")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --same-device rehearses the N>1 path on a one-GPU box (not a benchmark)")
    ap.add_argument("--same-device", action="store_true", help="every rank uses cuda:0 (rehearsal only)")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the RCCL code path even with one rank (rehearsal of the collectives on a one-GPU box)")
    ap.add_argument("--spawn-selftest", default="", choices=["", "ok", "fail"],
                    help="CPU-only check of the self-launch path: ranks rendezvous over gloo and report (tests/test_dist_cpu.py)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args.gpus))  # before anything touches the GPU
    if args.spawn_selftest:
        raise SystemExit(spawn_selftest(args.spawn_selftest))

    import numpy as np
    import torch

    from ray_tracer_webgl_amd import abi, dist as ptdist, scenes
    from ray_tracer_webgl_amd.tracer import PathTracer

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print("bench.py: --gpus %d but the launcher gave %d rank(s); using the launcher's" % (args.gpus, world), file=sys.stderr)
    use_dist = world > 1 or args.force_dist
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    ranks_seen = 1
    if use_dist:
        import torch.distributed as dist

        if args.force_dist and "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
        ranks_seen = dist.get_world_size()

    pps_frame = max(1, args.passes_per_step)  # passes per step of the single-GPU workload
    pps = pps_frame * (world if args.scaling == "weak" else 1)  # passes per step on this rank's rows
    spp_step = args.spp_per_pass * pps

    def gather(t):
        if args.backend == "gloo":  # rehearsal path: collectives on host copies
            return ptdist.gather_rows(t.cpu(), p.height, band_rows, rank, world)
        return ptdist.gather_rows(t, p.height, band_rows, rank, world)

    sc = scenes.config2(args.width, args.height, args.spp_per_pass, args.steps * pps, args.max_depth)
    p = sc.params.copy()
    band_rows = args.band_rows
    p.band_rows, p.band_index, p.band_count = ptdist.band_of(rank, world, band_rows)
    p.time_step = abi.PT_TIME_STEP_DECORRELATED  # independent passes (include/ptrace.h)

    pt = PathTracer(p.width, p.height, device=local_rank, use_torch=True)
    pt.set_spheres(sc.spheres)
    pt.set_params(p)
    spl = max(1, min(args.steps_per_launch, max(args.steps, 1)))
    ppl = spl * pps  # passes per launch
    pt.reserve_passes(ppl)
    # set-up, like reserving the workspace: settle how PHASE 1 looks at the sphere list (list
    # walks, hierarchy, grid: bit-identical images) by measuring each once on this scene
    pt.tune(ppl)

    def sync_all():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    def run_steps(k, first_time, pps_use=None, spl_use=None):
        pps_use = pps_use or pps
        done = 0
        while done < k:
            n = min(spl_use or spl, k - done)
            q = p.copy()
            q.time = float(first_time)
            q.first_pass = done * pps_use  # pass j of this launch uses u_time = time + (first_pass + j) * time_step
            pt.set_params(q)
            pt.render_passes(n * pps_use)  # asynchronous on torch's current stream
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
    mine = torch.tensor([float(st.segments), float(st.render_kernel_ms)], dtype=torch.float64, device=cdev)
    if use_dist:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
        per_rank = [torch.zeros_like(mine) for _ in range(ranks_seen)]
        dist.all_gather(per_rank, mine)
    else:
        per_rank = [mine]
    wall = float(elapsed.item())
    segments = float(sum(float(x[0]) for x in per_rank))
    per_rank_kernel_ms = [round(float(x[1]), 3) for x in per_rank]

    # the config's own frame (1024 spp) split over the ranks: the strong-scaling point.  At N = 1, or
    # under --scaling strong, the main region already is that workload.
    fixed_frame = None
    if world > 1 and args.scaling == "weak":
        k_ff = max(1, CONVERGED_SPP // (args.spp_per_pass * pps_frame))
        spl_ff = max(1, min(args.steps_per_launch, k_ff, ppl // pps_frame))  # within the reserved passes
        pt.reset()
        run_steps(k_ff, 1000.0, pps_frame, spl_ff)  # settle the tile order for this launch shape
        gather(pt.accum_tensor)
        sync_all()
        pt.reset()
        sync_all()
        f0 = time.perf_counter()
        run_steps(k_ff, 0.0, pps_frame, spl_ff)
        full = gather(pt.accum_tensor)
        sync_all()
        f1 = time.perf_counter()
        sf = pt.stats()
        ff = torch.tensor([f1 - f0], dtype=torch.float64, device=cdev)
        fseg = torch.tensor([float(sf.segments)], dtype=torch.float64, device=cdev)
        dist.all_reduce(ff, op=dist.ReduceOp.MAX)
        dist.all_reduce(fseg)
        fixed_frame = {
            "spp": k_ff * args.spp_per_pass * pps_frame,
            "sec": round(float(ff.item()), 5),
            "value": round(float(fseg.item()) / float(ff.item()) / 1e6, 3),
            "unit": "Mray/s",
            "scaling": "strong",
            "note": "same frame as N = 1 (%d launches of <= %d passes over 1/%d of the rows per rank + the gather), "
                    "barrier + synchronize on both sides, max over ranks" % ((k_ff + spl_ff - 1) // spl_ff, spl_ff * pps_frame, world),
        }

    if rank == 0:
        n_sph = len(sc.spheres)
        n_passes = args.steps * pps
        mrays = segments / wall / 1e6
        # dominant kernel: the trace kernel, timed with HIP events on its launch stream (rank 0)
        avg_ms = st.render_kernel_ms / max(st.render_launches, 1)
        ms_per_pass = st.render_kernel_ms / max(n_passes, 1)
        seg_per_pass = st.segments / max(n_passes, 1)
        alg_flop_per_pass = FLOP_PER_SPHERE_TEST * n_sph * seg_per_pass
        local_pix = st.local_rows * p.width
        # algorithmic HBM bytes per pass of the trace kernel: one 16-B slab store per (pixel, pass)
        # item (+ the scene once per launch, 48 B/sphere); the fold kernel's traffic is separate
        hbm_bytes_per_pass = local_pix * 16 + n_sph * 48 / max(ppl, 1)
        kernel_name = {abi.PT_GEOM_LDS: "pt_trace_kernel", abi.PT_GEOM_SCALAR: "pt_trace_kernel_scalar",
                       abi.PT_GEOM_BVH: "pt_trace_kernel_bvh", abi.PT_GEOM_GRID: "pt_trace_kernel_grid"}.get(st.geometry_path, "?")
        walk = st.geometry_path in (abi.PT_GEOM_BVH, abi.PT_GEOM_GRID)
        chosen_path = st.geometry_path

        # executed work: the measuring twin of the same kernel, same launch shape, after the timed region
        executed = None
        exec_flop_per_pass = None
        if walk and not args.no_work_count:
            pt.set_geometry_path(chosen_path)
            pt.reset()
            pt.set_count_work(True)
            q = p.copy()
            q.time = 0.0
            pt.set_params(q)
            pt.render_passes(ppl)
            sw = pt.stats()
            pt.set_count_work(False)
            w = [float(x) for x in sw.work]
            seg_c = max(float(sw.segments), 1.0)
            f_walk = FLOP_PER_NODE_STEP if chosen_path == abi.PT_GEOM_BVH else FLOP_PER_CELL_STEP
            n_always = sw.grid_always if chosen_path == abi.PT_GEOM_GRID else sw.bvh_outliers
            flop = (w[1] * f_walk + w[3] * FLOP_PER_LEAF_ROUND + w[5] * FLOP_PER_EXACT +
                    seg_c * (FLOP_PER_SEGMENT_SHADE + FLOP_PER_SPHERE_TEST * n_always))
            exec_flop_per_pass = flop / ppl
            per64 = 64.0 / seg_c
            executed = {
                "source": "the kernel's own tallies (measuring twin %s_count, one %d-pass launch after the timed region)" % (kernel_name, ppl),
                "per_64_segments": {
                    "walk_iterations": round(w[0] * per64, 3), "walk_lanes_active": round(w[1] / max(w[0], 1.0), 2),
                    "leaf_rounds": round(w[2] * per64, 3), "leaf_lanes_active": round(w[3] / max(w[2], 1.0), 2),
                    "exact_evaluations": round(w[4] * per64, 3), "exact_lanes_active": round(w[5] / max(w[4], 1.0), 2),
                    "wave_steps": round(w[6] * per64, 4), "lanes_carried": round(w[7] * per64, 3),
                },
                "lane_utilisation": {
                    "walk": round(w[1] / max(64.0 * w[0], 1.0), 3), "leaf": round(w[3] / max(64.0 * w[2], 1.0), 3),
                    "exact": round(w[5] / max(64.0 * w[4], 1.0), 3),
                    "shade": round(seg_c / max(64.0 * w[6], 1.0), 3),
                },
                "flop_model": "lanes x (walk step %d, leaf round %d, exact %d) + segments x (%d shade/RNG/camera + %d per always-tested sphere)"
                              % (f_walk, FLOP_PER_LEAF_ROUND, FLOP_PER_EXACT, FLOP_PER_SEGMENT_SHADE, FLOP_PER_SPHERE_TEST),
                "literal_tests_per_segment": round((4.0 * w[3]) / seg_c + n_always, 2),
            }
        if not walk:  # the list walks execute exactly the algorithmic work
            exec_flop_per_pass = alg_flop_per_pass
        achieved_tf = (exec_flop_per_pass / (ms_per_pass * 1e-3) / 1e12) if (exec_flop_per_pass and ms_per_pass > 0) else None

        # a prior rocprofv3 PMC profile of the same kernel and launch shape, if one is committed
        prior = None
        prof = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(prof):
            try:
                rec = json.load(open(prof))
                if rec.get("kernel") == kernel_name and rec.get("spp_per_pass") == args.spp_per_pass and rec.get("passes_per_launch") == ppl:
                    prior = rec
            except Exception:
                prior = None
        roofline = {
            "kernel": kernel_name,
            "geometry_path": abi.GEOM_NAMES.get(st.geometry_path, "?") + (" (autotuned)" if st.geometry_tuned else ""),
            "bound": "valu",
            "achieved": round(achieved_tf, 3) if achieved_tf is not None else None,
            "peak": FP32_VALU_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": round(achieved_tf / FP32_VALU_PEAK_TFLOPS, 4) if achieved_tf is not None else None,
            "traffic": prior.get("hbm_bytes_per_pass") if prior else None,
            "traffic_source": ("PRIOR profile %s (rocprofv3 --pmc WRITE_SIZE + 2*FETCH_SIZE, same kernel and launch shape), per pass"
                               % prior.get("profile", "profiles/pmc_traffic.json")) if prior else None,
            "per_pass": {
                "kernel_ms": round(ms_per_pass, 4),
                "segments": round(seg_per_pass, 1),
                "executed_flop": round(exec_flop_per_pass, 1) if exec_flop_per_pass else None,
                "algorithmic_flop": alg_flop_per_pass,
                "algorithmic_hbm_bytes": round(hbm_bytes_per_pass, 1),
            },
            "avg_launch_ms": round(avg_ms, 4),
            "launches": int(st.render_launches),
            "passes_per_launch": ppl,
            "algorithmic_speedup_vs_list_walk": round(alg_flop_per_pass / (ms_per_pass * 1e-3) / 1e12 / FP32_VALU_PEAK_TFLOPS, 4)
                                                if ms_per_pass > 0 else None,
            "note": ("fp32 vector peak: no dense contraction on this path.  frac = EXECUTED lane-level fp32 work / peak (<= 1). "
                     "algorithmic_speedup_vs_list_walk = 20 FLOP x %d spheres x segments / time / peak: what the reference's linear "
                     "loop (static/shader.frag:175-196) would have needed for the same rays; it exceeds 1 exactly because the walk "
                     "kernels skip tests whose outcome is provably 'miss'" % n_sph),
            "executed": executed,
            "valu_issue_frac_prior_profile": prior.get("valu_issue_frac") if prior else None,
            "hbm": {
                "achieved": round(hbm_bytes_per_pass / (ms_per_pass * 1e-3) / 1e9, 3) if ms_per_pass > 0 else 0.0,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(hbm_bytes_per_pass / (ms_per_pass * 1e-3) / 1e9 / HBM_PEAK_GBS, 6) if ms_per_pass > 0 else 0.0,
            },
        }
        # Beside the headline (fastest path): the same frame with the reference's own algorithm,
        # the linear walk over the whole list, after the timed region.
        list_walk = None
        if walk and world == 1 and not args.no_list_walk:
            pt.set_geometry_path(abi.PT_GEOM_SCALAR)
            pt.reset()
            k_lw = min(args.steps, spl)
            run_steps(k_lw, 0.0)  # settle the tile order for this path
            sync_all()
            pt.reset()
            sync_all()
            tl0 = time.perf_counter()
            run_steps(k_lw, 0.0)
            sync_all()
            tl1 = time.perf_counter()
            sl = pt.stats()
            l_ms_pass = sl.render_kernel_ms / max(k_lw * pps, 1)
            l_tf = FLOP_PER_SPHERE_TEST * n_sph * (sl.segments / max(k_lw * pps, 1)) / (l_ms_pass * 1e-3) / 1e12
            list_walk = {
                "kernel": "pt_trace_kernel_scalar",
                "value": round(sl.segments / (tl1 - tl0) / 1e6, 3),
                "unit": "Mray/s",
                "steps": k_lw,
                "sec_to_converged_frame": round((tl1 - tl0) / k_lw * (CONVERGED_SPP / spp_step), 4),
                "kernel_ms_per_pass": round(l_ms_pass, 4),
                "roofline_frac": round(l_tf / FP32_VALU_PEAK_TFLOPS, 4),
                "note": "every sphere tested for every ray, as static/shader.frag:175-196 does; same image bits, same segment "
                        "count per pass; executed work = algorithmic work, so this IS a roofline fraction",
            }
            pt.set_geometry_path(abi.PT_GEOM_AUTO)
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cores = usable_cores()
            flags, native = build_native_oracle()
            from oracle import oracle

            strip = min(args.cpu_strip, p.width)
            x0 = (p.width - strip) // 2
            cp = sc.params.copy()
            cp.samples_per_pixel = spp_step  # one step's worth of samples in one pass (the longest streams)
            tc0 = time.perf_counter()
            _, cseg = oracle.render(sc.spheres, cp, 1, window=(x0, x0 + strip, 0, p.height), nthreads=cores)
            tc1 = time.perf_counter()
            cpu = {
                "value": round(cseg / (tc1 - tc0) / 1e6, 3),
                "unit": "Mray/s",
                "cores": cores,
                "kind": "port",
                "build": flags,
                "native_build": native,
                "sample": "one %d-spp pass of the centre %dx%d column strip of the same frame (%d segments, %.1f s)"
                          % (spp_step, strip, p.height, cseg, tc1 - tc0),
            }
        out = {
            "metric": "Mray/s (ray segments/s) at %dx%d" % (p.width, p.height),
            "value": round(mrays, 3),
            "unit": "Mray/s",
            "n_gpus": world,
            "ranks": ranks_seen,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(wall / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "config2: Shirley cover scene (%d spheres), %dx%d, %d bounces, %d spp/step (%d passes x %d spp) x %d steps = %d spp"
                            % (n_sph, p.width, p.height, args.max_depth, spp_step, pps, args.spp_per_pass, args.steps, spp_step * args.steps),
                "partition": "%d rank(s), interleaved %d-row bands, one all_gather at the end of the timed region%s"
                             % (world, band_rows, "; per-GPU work fixed: 1/%d of the rows x %d passes per step" % (world, pps)
                                if (world > 1 and args.scaling == "weak") else ""),
                "steps_per_launch": spl,
                "passes_per_launch": ppl,
                "spp_per_pass": args.spp_per_pass,
            },
            "sec_to_converged_frame": fixed_frame["sec"] if fixed_frame else round(wall / args.steps * (CONVERGED_SPP / spp_step), 4),
            "fixed_frame": fixed_frame,
            "converged_frame_spp": CONVERGED_SPP,
            "segments": int(segments),
            "nominal_mray_s": round(p.width * p.height * spp_step * args.steps * args.max_depth / wall / 1e6, 1),
            "per_rank_render_kernel_ms": per_rank_kernel_ms,
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


def build_native_oracle():
    """the CPU baseline is the oracle built -O3 -march=native (-ffp-contract=off
    stays: it is part of the arithmetic contract).  The shipped libpt_oracle.so is a portable
    -march=x86-64-v3 build because it travels to other hosts; the baseline leg rebuilds it for THIS
    host.  Returns (flags, native?)."""
    src = os.path.join(ROOT, "oracle", "pt_oracle.c")
    out = os.path.join(ROOT, "oracle", "libpt_oracle_native.so")
    flags = "-O3 -std=c11 -fPIC -march=native -ffp-contract=off -fno-fast-math -fno-math-errno -fno-builtin-sin -fno-builtin-cos"
    try:
        subprocess.check_call(["gcc"] + flags.split() + [src, "-o", out, "-shared", "-lm", "-lpthread"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        os.environ["PT_ORACLE_LIB"] = out
        return "gcc " + flags, True
    except Exception:
        return "shipped build: gcc -O2 -march=x86-64-v3 -ffp-contract=off (native rebuild failed)", False


if __name__ == "__main__":
    main()
