#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on BASELINE.json's configs.

  metric   Mray/s (ray segments = hit_world invocations per second)
  workload --config 2 (default; the config the metric is quoted on): Shirley cover scene (484 spheres),
           1920x1080, 50 bounces, 1024 spp;  --config 3: the same scene at 3840x2160, 4096 spp (BASELINE's
           8-GPU case; fits one GPU too);  --config default: the reference's own operating point, see
           frame_loop_bench() below (State::default, 1280x702, 1 spp per frame, depth 8, temporal blend).
  step     one pass of the hot path over one batch = 64 samples for every pixel of the frame, rendered
           as `--passes-per-step` (4) seeds of `--spp-per-pass` (16) samples each (u_time = (4 * step + j)
           x 0.3618: the reference also accumulates many low-spp frames with distinct u_time,
           README.md:6).  The default step count is exactly the config's converged frame (16 steps =
           1024 spp for config 2, 64 steps = 4096 spp for config 3).  Steps are enqueued
           `--steps-per-launch` at a time through pt_render_passes (one persistent kernel launch works
           through all their (pixel, pass) items from one queue).
  N > 1    the image's rows are dealt to the ranks in interleaved 4-row bands (the path shards by
           pixel: no collective while rendering), ONE all_gather of the radiance buffers (RCCL over
           xGMI) at the end of the timed region.  Default `--scaling strong`: the SAME frame for every
           N (north_star: the 1024-spp cover-scene frame "at 1/2/4/8 GPUs"): `value`, `ms_per_step`
           and `sec_to_converged_frame` are on BASELINE's workload whatever N is, and rank 0 checks
           the SHA-256 of the gathered frame against the committed single-GPU digest
           (`gather_matches_single_gpu`) — the RCCL parity check.  The weak-scaling point (per-GPU
           work fixed: 1/N of the rows x N x as many passes) is timed after the main region and
           reported beside it as `weak_series`; `--scaling weak` makes it the main region instead.
           `python bench.py --gpus N` from a bare shell starts its N ranks itself (the parent never
           touches the GPU); under torch.distributed.run it uses the ranks it is given.

Prints ONE JSON line on rank 0.  `roofline` prices the path-tracing kernel against the FP32
vector peak (the path has no dense contraction and ~1e5 FLOP per HBM byte):
`frac` is EXECUTED lane-level fp32 work (from the kernel's own tallies, measuring twin run after
the timed region) over the peak, always <= 1; beside this MODEL figure the line carries the
counter-derived ones of the committed PMC profile of the same kernel and launch shape
(`counters`: valu_issue_frac, lane_utilisation, fp32_flop_frac — labelled PRIOR, they are not
this run's).  What the reference's linear loop would have done for the same rays is
`algorithmic_speedup_vs_list_walk`, and the linear walk itself is timed beside it (`list_walk`).
Everything is normalised PER PASS so any --steps gives the same ratios.  `cpu_baseline` is the CPU
oracle (rebuilt -O3 -march=native on this host) timed on this host's cores over a bounded sample
of the same workload.  `first_frame_ms` is what the steady-state figure hides: scene upload +
structure builds + path autotune + one cold converged frame.
"""
import argparse
import hashlib
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

# BASELINE.json configs bench.py can time: frame size, converged sample count, committed digest key
# (tests/golden/full_frame_digests.json: the single-GPU frame rendered as passes of 16 spp with
# decorrelated pass times, exactly what the default flags of this script render)
BENCH_CONFIGS = {
    "2": {"name": "config2", "scene": "config2", "what": "Shirley cover scene", "width": 1920, "height": 1080, "spp": 1024,
          "digest": "config2_1920x1080_64x16spp_decorrelated", "cpu_strip": 960},
    "3": {"name": "config3", "scene": "config3", "what": "Shirley cover scene", "width": 3840, "height": 2160, "spp": 4096,
          "digest": "config3_3840x2160_256x16spp_decorrelated", "cpu_strip": 960},
    # the stress configs of BASELINE.json (not the metric's config): the same line for them
    "4": {"name": "config4", "scene": "config4", "what": "closed room with an emissive sphere", "width": 1024, "height": 1024,
          "spp": 8192, "digest": "config4_1024x1024_512x16spp_decorrelated", "cpu_strip": 1024},
    "5": {"name": "config5", "scene": "config5", "what": "random field + ground", "width": 1920, "height": 1080, "spp": 256,
          "digest": "config5_1920x1080_16x16spp_decorrelated", "cpu_strip": 40},
}


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


# Rendezvous and every collective: a rank that never arrives fails the others after 3 min instead of the
# backends' 10-30 min defaults (the driver's limit for a run is 10 min).  Not tighter: the first `import torch`
# on a fresh box pages the image in for 1-2 min and the ranks need not finish that together.  The launchers
# (spawn_ranks below, torch.distributed.run) notice a DEAD rank within seconds; this bound is for a live one that hangs.
RDZV_TIMEOUT_S = 180


def ipc_env():
    """HSA_ENABLE_IPC_MODE_LEGACY=0 for every launch style (self-started ranks, torchrun, a bare
    rank), set before torch / HIP is loaded.  RCCL shares its xGMI transport buffers between the
    ranks' processes through HIP IPC handles; this pool's host driver exports device memory as
    dmabuf only, and with the ROCr runtime's legacy IPC mode (its default) the first communicator
    fails with `hipIpcGetMemHandle: invalid argument`.  A value the caller exported is kept."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def spawn_ranks(n, job_timeout_s=None):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (fresh
    processes, one per GPU, rendezvous on 127.0.0.1), relay rank 0's JSON line, and SUPERVISE them:
    every child is polled; the first one that ends with a non-zero code (a bad device, an
    out-of-memory kill, a GPU fault, at start-up or in the middle of a frame) ends the job — the
    other ranks, which would otherwise sit in the rendezvous or in a collective waiting for it, are
    terminated (exactly the children started here, each in its own session), and the parent
    returns that code within seconds with the rank's number and its last stderr lines.  This
    parent makes no GPU call and replaces no process."""
    import collections
    import signal
    import threading

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs, tails, out0, readers = [], [], [], []

    def pump(stream, keep, sink):
        for line in stream:
            keep.append(line)
            if sink is not None:
                sink.write(line)
                sink.flush()

    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, text=True, start_new_session=True,
                             stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=subprocess.PIPE)
        procs.append(p)
        tails.append(collections.deque(maxlen=12))
        readers.append(threading.Thread(target=pump, args=(p.stderr, tails[r], sys.stderr), daemon=True))
        if r == 0:
            readers.append(threading.Thread(target=pump, args=(p.stdout, out0, None), daemon=True))
    for t in readers:
        t.start()
    print("bench.py: started ranks " + " ".join("%d:pid%d" % (r, p.pid) for r, p in enumerate(procs)), file=sys.stderr, flush=True)

    def stop_all(grace=5.0):
        """SIGTERM, then SIGKILL, to the sessions of the children started above — nobody else."""
        for sig in (signal.SIGTERM, signal.SIGKILL):
            for p in procs:
                if p.poll() is None:
                    try:
                        os.killpg(p.pid, sig)  # start_new_session: the child leads its own process group
                    except (ProcessLookupError, PermissionError):
                        pass
            end = time.time() + grace
            while time.time() < end and any(p.poll() is None for p in procs):
                time.sleep(0.05)

    rc, failed = 0, None
    deadline = time.time() + job_timeout_s if job_timeout_s else None
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                failed, rc = bad[0], codes[bad[0]]
                break
            if all(c == 0 for c in codes):
                break
            if deadline and time.time() > deadline:
                failed, rc = -1, 124
                break
            time.sleep(0.02)
    finally:
        stop_all()
    for t in readers:
        t.join(timeout=2.0)
    if failed is not None:
        if failed < 0:
            print("bench.py: ranks still running after %d s (--job-timeout); all %d terminated" % (job_timeout_s, n), file=sys.stderr)
        else:
            print("bench.py: rank %d (pid %d) ended with code %d; the other %d rank(s) were terminated.  Its last stderr lines:"
                  % (failed, procs[failed].pid, rc, n - 1), file=sys.stderr)
            for line in tails[failed]:
                sys.stderr.write("    [rank %d] %s" % (failed, line))
        sys.stderr.flush()
        return rc if rc > 0 else 1  # a signal's negative code is still a failure
    for line in "".join(out0).splitlines():  # stdout carries the JSON line only (gloo chats on stdout)
        (sys.stdout if line.startswith("{") else sys.stderr).write(line + "\n")
    sys.stdout.flush()
    return 0


def spawn_selftest(mode):
    """What a rank does under --spawn-selftest: no GPU, no rendering — join the gloo group the
    parent set up, prove every rank is there, let rank 0 print the line the parent relays.  The
    failing modes rehearse a rank that dies where the others cannot see it: `die-early` before the
    rendezvous (rank 0 waits in init_process_group), `die-in-collective` after it (rank 0 waits in
    an all_reduce the dead rank never joins), `fail` after the collectives."""
    import datetime

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if mode == "die-early" and rank == world - 1:
        print("selftest: rank %d leaves before the rendezvous" % rank, file=sys.stderr, flush=True)
        os._exit(5)
    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=RDZV_TIMEOUT_S))
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t)
    if mode == "die-in-collective":
        if rank == world - 1:
            print("selftest: rank %d leaves while the others wait in a collective" % rank, file=sys.stderr, flush=True)
            os._exit(7)
        dist.all_reduce(t)  # never completes: the last rank is gone
    if mode == "hang":  # nobody dies, nobody finishes: only the job's own time limit ends this
        if rank == world - 1:
            time.sleep(3600)
        dist.all_reduce(t)
    if mode == "fail" and rank == world - 1:
        os._exit(3)  # a rank that dies after the rendezvous: the parent must report failure
    if rank == 0:
        print(json.dumps({"selftest": True, "ranks": dist.get_world_size(), "rank_sum": float(t.item()),
                          "local_rank": int(os.environ["LOCAL_RANK"]),
                          "ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}), flush=True)
    if mode == "ok":
        dist.barrier()
        dist.destroy_process_group()
    return 0


class stdout_to_stderr:
    """RCCL prints a version banner to STDOUT when its first communicator comes up; rank 0's stdout must
    carry the JSON line and nothing else.  Sends file descriptor 1 to stderr for the duration."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def frame_digest(full):
    """SHA-256 of the (height, width, 4) fp32 radiance frame, as tests/golden/make_full_digests.py
    computes it."""
    import numpy as np

    a = full.detach().cpu().numpy() if hasattr(full, "detach") else full
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float32).tobytes()).hexdigest()


def load_digests():
    try:
        return json.load(open(os.path.join(ROOT, "tests", "golden", "full_frame_digests.json")))
    except Exception:
        return {}


def setup_times(pt):
    """The library's own host clocks inside pt_create / pt_set_spheres / pt_reserve_passes (include/ptrace_dev.h), ms."""
    import ctypes as C

    names = ["pt_create.hip_runtime_start", "pt_create.device_and_properties", "pt_create.stream_and_counters", "pt_create.kernel_attributes_and_code_object_load",
             "pt_create.buffers", "pt_create.total", "pt_set_spheres.split_records", "pt_set_spheres.hierarchy_build", "pt_set_spheres.grid_build",
             "pt_set_spheres.allocations_and_uploads", "pt_set_spheres.total", "pt_reserve_passes.total",
             "pt_create.stream_and_counters.hipStreamCreate", "pt_create.stream_and_counters.first_hipMalloc", "pt_create.stream_and_counters.first_hipMemsetAsync"]
    try:
        fn = pt.lib.pt_debug_setup_times
    except AttributeError:
        return {}
    fn.restype = C.c_long
    fn.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_size_t]
    buf = (C.c_double * len(names))()
    n = fn(pt._ctx, buf, len(names))
    return {"library_ms": {names[k]: round(buf[k], 3) for k in range(max(n, 0))}}


# A timed region of N ranks is 1/N as long as the single-GPU one (config 2 at N = 8: 20 ms), and so is everything before
# it: a device that comes out of idle has not reached its clocks by then, and the rank that starts coldest sets the job's time
# (one-device rehearsal, profiles/r06_band_rehearsal.txt: the first of eight ranks 19.9 ms, the same rank on a warm device
# 18.4).  After the W warm-up steps the launches of those steps are therefore repeated — untimed, same launch shape — until
# the device has been busy this long, for every N alike (N = 1 reaches it with two repeats of its own warm-up).
CLOCK_WARMUP_MS = 200.0


def keep_busy(pt, run, steps, busy_ms, limit=256):
    """run(steps) again and again (untimed) until the context's kernel time since its last reset reaches busy_ms; returns it"""
    import torch

    n = 0
    while True:
        torch.cuda.synchronize()
        done = float(pt.stats().render_kernel_ms)
        if done >= busy_ms or n >= limit:
            return done
        run(max(int(steps), 1))
        n += 1


def plan_steps(converged_spp, spp_per_pass, passes_per_step, steps):
    """(steps, spp per step): the default step count is the config's converged frame."""
    spp_step = spp_per_pass * passes_per_step
    if steps is None:
        steps = max(1, converged_spp // spp_step)
    return steps, spp_step


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--config", default="2", choices=["2", "3", "4", "5", "default"],
                    help="BASELINE config: 2 = cover scene 1920x1080 1024 spp (the metric's config), 3 = the same at "
                         "3840x2160 4096 spp, 4 = closed room 1024x1024 8192 spp, 5 = 10 000-sphere field 1920x1080 256 spp, "
                         "default = the reference's own 1-spp frame loop on State::default")
    ap.add_argument("--steps", type=int, default=None, help="default: the config's converged frame (16 / 64)")
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--steps-per-launch", type=int, default=None,
                    help="default: 16, and 16 x N for a rank of N in strong scaling (a rank's rows are 1/N of the frame: "
                         "the same slab memory as the single-GPU launch, and a longer launch loses less to its start and drain)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--spp-per-pass", type=int, default=16)
    ap.add_argument("--passes-per-step", type=int, default=4)
    ap.add_argument("--max-depth", type=int, default=50)
    ap.add_argument("--band-rows", type=int, default=4,
                    help="N > 1: rows per interleaved band (4: rank shares of the work within 1 %% of each other at N = 8)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong = the same frame for every N (BASELINE's workload); weak = passes per step grow with N")
    ap.add_argument("--clock-warmup-ms", type=float, default=CLOCK_WARMUP_MS,
                    help="after the --warmup steps, repeat them (untimed) until the device has been busy this long: the clocks of a device "
                         "that was idle a moment ago (0 = only the --warmup steps)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-list-walk", action="store_true",
                    help="skip the extra (untimed-region) launch that measures the reference's linear list walk")
    ap.add_argument("--no-work-count", action="store_true", help="skip the measuring-twin launch (roofline.frac = null)")
    ap.add_argument("--no-first-frame", action="store_true", help="skip the cold first-frame measurement")
    ap.add_argument("--no-weak-series", action="store_true", help="N > 1: skip the weak-scaling point after the main region")
    ap.add_argument("--cpu-strip", type=int, default=None,
                    help="width of the CPU baseline's column strip (default per config: 10-30 s of CPU work)")
    ap.add_argument("--frames", type=int, default=640, help="--config default: frames per timed replay series (640 = ten groups of 64)")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="--config default: only the replayed animation loop (no host-issued ticks, no paused 25-spp frames): what profiles/collect.sh runs under the PMC passes")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --same-device rehearses the N>1 path on a one-GPU box (not a benchmark)")
    ap.add_argument("--same-device", action="store_true", help="every rank uses cuda:0 (rehearsal only)")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the RCCL code path even with one rank (rehearsal of the collectives on a one-GPU box)")
    ap.add_argument("--job-timeout", type=int, default=540,
                    help="self-started ranks (--gpus N from a bare shell): end the job after this many seconds (0 = never)")
    ap.add_argument("--spawn-selftest", default="", choices=["", "ok", "fail", "die-early", "die-in-collective", "hang"],
                    help="CPU-only check of the self-launch path: ranks rendezvous over gloo and report (tests/test_dist_cpu.py)")
    args = ap.parse_args()

    ipc_env()  # before torch / HIP is loaded, for every launch style
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args.gpus, args.job_timeout or None))  # before anything touches the GPU
    if args.spawn_selftest:
        raise SystemExit(spawn_selftest(args.spawn_selftest))
    if args.config == "default":
        raise SystemExit(frame_loop_bench(args))

    import numpy as np
    import torch

    from ray_tracer_webgl_amd import abi, dist as ptdist, scenes
    from ray_tracer_webgl_amd.tracer import PathTracer

    cfg = BENCH_CONFIGS[args.config]
    width = args.width or cfg["width"]
    height = args.height or cfg["height"]
    converged_spp = cfg["spp"]

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
        import datetime

        import torch.distributed as dist

        limit = datetime.timedelta(seconds=RDZV_TIMEOUT_S)
        if args.force_dist and "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        with stdout_to_stderr():
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=limit)
            else:
                dist.init_process_group("gloo", timeout=limit)
            ranks_seen = dist.get_world_size()
            # bring the communicator up here (RCCL initialises lazily, and talks while it does)
            t_up = torch.zeros(1, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(t_up, async_op=True).wait(timeout=limit)  # a rank that died since the rendezvous: an error here, not a hang
            if args.backend == "nccl":
                torch.cuda.synchronize()

    pps_frame = max(1, args.passes_per_step)  # passes per step of the single-GPU workload
    weak = args.scaling == "weak" and world > 1
    pps = pps_frame * (world if weak else 1)  # passes per step on this rank's rows
    steps, spp_step_frame = plan_steps(converged_spp, args.spp_per_pass, pps_frame, args.steps)
    spp_step = args.spp_per_pass * pps

    def gather(t):
        if args.backend == "gloo":  # rehearsal path: collectives on host copies
            return ptdist.gather_rows(t.cpu(), p.height, band_rows, rank, world)
        return ptdist.gather_rows(t, p.height, band_rows, rank, world)

    sc = getattr(scenes, cfg["scene"])(width, height, args.spp_per_pass, steps * pps, args.max_depth)
    p = sc.params.copy()
    band_rows = args.band_rows
    p.band_rows, p.band_index, p.band_count = ptdist.band_of(rank, world, band_rows)
    p.time_step = abi.PT_TIME_STEP_DECORRELATED  # independent passes (include/ptrace.h)

    spl_frame = 16  # steps per launch of the single-GPU workload
    spl_want = args.steps_per_launch or (spl_frame * world if (world > 1 and not weak) else spl_frame)
    spl = max(1, min(spl_want, max(steps, 1)))
    ppl = spl * pps  # passes per launch
    # the weak-scaling point after a strong main region renders world x the passes per step, spl_frame steps per
    # launch (over 1/world of the rows: the same slab memory as the single-GPU launch)
    want_weak_series = world > 1 and not weak and not args.no_weak_series
    spl_weak = max(1, min(args.steps_per_launch or spl_frame, max(steps, 1)))
    reserve = max(ppl, spl_weak * pps_frame * world) if want_weak_series else ppl

    def sync_all():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    # ---- set-up + the cold first frame (what the steady-state figures below do not show) ----------
    sync_all()
    tf0 = time.perf_counter()
    pt = PathTracer(p.width, p.height, device=local_rank, use_torch=True)
    tfa = time.perf_counter()
    pt.set_spheres(sc.spheres)  # upload + hierarchy and grid builds
    tfb = time.perf_counter()
    pt.set_params(p)
    tfc = time.perf_counter()
    pt.reserve_passes(reserve)
    tf1 = time.perf_counter()
    # settle how PHASE 1 looks at the sphere list (list walks, hierarchy, grid: bit-identical
    # images) by measuring each usable path once on this scene, on a short launch
    tune_passes = min(ppl, 8)
    pt.tune(tune_passes)
    tf2 = time.perf_counter()

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

    first_frame = None
    if not args.no_first_frame:
        k_frame = max(1, converged_spp // spp_step)
        run_steps(k_frame, 500.0)
        torch.cuda.synchronize()
        tf3 = time.perf_counter()
        first_frame = {
            "ms": round((tf3 - tf0) * 1e3, 2),
            "set_scene_ms": round((tf1 - tf0) * 1e3, 2),
            "autotune_ms": round((tf2 - tf1) * 1e3, 2),
            "cold_frame_ms": round((tf3 - tf2) * 1e3, 2),
            # set_scene_ms = the four calls below (host clock around each); inside pt_create and pt_set_spheres the library's own
            # host clocks (include/ptrace_dev.h pt_debug_setup_times)
            "set_scene_breakdown": dict(
                {"context_ms": round((tfa - tf0) * 1e3, 2), "set_spheres_ms": round((tfb - tfa) * 1e3, 2),
                 "set_params_ms": round((tfc - tfb) * 1e3, 2), "reserve_passes_ms": round((tf1 - tfc) * 1e3, 2)},
                **setup_times(pt)),
            "note": "context (PathTracer: pt_create + torch buffer + stream binding; in a fresh process most of pt_create is the HIP runtime coming up and the code "
                    "object loading: a second context takes 3 ms, tools/cold_start.py) + scene upload + structure builds + workspace (%d slabs); pt_tune (the grid rebuilt for the margin class that covers the camera; one cold and one measured %d-pass launch per "
                    "geometry path whose outcome is open: none on an even grid); the first %d-spp frame with no tile-order feedback from a launch of its own shape (rank 0's share)"
                    % (reserve, tune_passes, k_frame * spp_step),
        }
        pt.reset()

    # warmup (untimed), then clear accumulation and statistics
    run_steps(args.warmup, 1000.0)
    clock_warmup_ms = keep_busy(pt, lambda k: run_steps(k, 1000.0), args.warmup, args.clock_warmup_ms) if args.clock_warmup_ms > 0 else None
    if use_dist:
        gather(pt.accum_tensor)  # also sets up the RCCL channels outside the timed region
    sync_all()
    pt.reset()

    sync_all()
    t0 = time.perf_counter()
    run_steps(steps, 0.0)
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

    # RCCL parity: the gathered frame against the committed single-GPU digest of the config's converged frame
    # (passes of 16 spp, decorrelated pass times, 50 bounces: tests/golden/full_frame_digests.json).  When the timed
    # region rendered exactly that frame it is hashed as it stands; under any other --steps (the driver's own
    # flags: 20 steps = 1280 spp) every rank renders the committed workload ONCE MORE after the timed region,
    # through the same partition and the same gather, and rank 0 hashes that — so the line carries a verdict
    # (true / false, never null) whatever --steps / --warmup / --scaling say, at the config's default size.
    gather_check = None
    total_passes = steps * pps
    digest_sized = (width == cfg["width"] and height == cfg["height"] and args.spp_per_pass == 16 and args.max_depth == 50)
    timed_is_digest = digest_sized and not weak and total_passes * args.spp_per_pass == converged_spp
    want = load_digests().get(cfg["digest"]) if digest_sized else None
    if want:
        if timed_is_digest:
            check_frame, check_seg, check_src = full, segments, "the timed region's frame"
        else:
            pt.reset()
            run_steps(max(1, converged_spp // spp_step_frame), 0.0, pps_frame, spl_weak if weak else spl)
            check_frame = gather(pt.accum_tensor) if use_dist else pt.accum_tensor
            sync_all()
            seg_t = torch.tensor([float(pt.stats().segments)], dtype=torch.float64, device=cdev)
            if use_dist:
                dist.all_reduce(seg_t)
            check_seg = float(seg_t.item())
            check_src = ("the committed %d-spp workload rendered again after the timed region (the timed region ran %d spp%s), same partition, same gather"
                         % (converged_spp, total_passes * args.spp_per_pass, ", weak scaling" if weak else ""))
        if rank == 0:
            got = frame_digest(check_frame[: p.height])
            if os.environ.get("PT_BENCH_DUMP"):  # dev: keep the frame that was hashed
                np.save(os.environ["PT_BENCH_DUMP"], check_frame[: p.height].detach().cpu().numpy())
            gather_check = {"sha256": got, "expected": want["sha256"], "key": cfg["digest"],
                            "matches": got == want["sha256"],
                            "segments_match": int(check_seg) == int(want["segments"]),
                            "frame": check_src}
    elif rank == 0:
        gather_check = {"matches": None, "frame": "no committed digest for this size / pass shape (non-default --width/--height/--spp-per-pass/--max-depth)"}

    # the other scaling mode, timed after the main region with the same barriers (N > 1 only)
    weak_series = None
    if want_weak_series:
        pps_w = pps_frame * world
        k_w = min(steps, spl_weak)
        pt.reset()
        run_steps(k_w, 1000.0, pps_w, spl_weak)  # settle the tile order for this launch shape
        gather(pt.accum_tensor)
        sync_all()
        pt.reset()
        sync_all()
        f0 = time.perf_counter()
        run_steps(k_w, 0.0, pps_w, spl_weak)
        gather(pt.accum_tensor)
        sync_all()
        f1 = time.perf_counter()
        sf = pt.stats()
        ff = torch.tensor([f1 - f0], dtype=torch.float64, device=cdev)
        fseg = torch.tensor([float(sf.segments)], dtype=torch.float64, device=cdev)
        dist.all_reduce(ff, op=dist.ReduceOp.MAX)
        dist.all_reduce(fseg)
        weak_series = {
            "spp": k_w * args.spp_per_pass * pps_w,
            "sec": round(float(ff.item()), 5),
            "value": round(float(fseg.item()) / float(ff.item()) / 1e6, 3),
            "unit": "Mray/s",
            "scaling": "weak",
            "note": "per-GPU work fixed: 1/%d of the rows x %d passes per step (%d steps, one launch per rank + the gather), "
                    "barrier + synchronize on both sides, max over ranks" % (world, pps_w, k_w),
        }

    if rank == 0:
        n_sph = len(sc.spheres)
        n_passes = steps * pps
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
                       abi.PT_GEOM_SMALL: "pt_trace_kernel_small_t%d" % (n_sph % 4),  # one build per list length modulo four (pt_kernels_small.hip)
                       abi.PT_GEOM_BVH: "pt_trace_kernel_bvh", abi.PT_GEOM_GRID: "pt_trace_kernel_grid"}.get(st.geometry_path, "?")
        # which build of a walk kernel: everything staged in the LDS, the nodes / cell records only, or nothing
        # (pt_api.hip bind_hierarchy / bind_grid: what fits beside a 1024-thread workgroup's 60 KiB of parked state)
        lds_room = 163776 - 15 * 4 * 1024
        if st.geometry_path == abi.PT_GEOM_GRID:
            kernel_name += {1: "", 2: "_cells", 3: "_gmem"}.get(int(st.grid_kernel_build), "")  # (the library says which build it launches)
        elif st.geometry_path == abi.PT_GEOM_BVH:
            if (2 * int(st.bvh_nodes) + int(st.bvh_slots)) * 16 > lds_room:
                kernel_name += "_nodes" if int(st.bvh_nodes) * 16 <= lds_room else "_gmem"
        walk = st.geometry_path in (abi.PT_GEOM_BVH, abi.PT_GEOM_GRID)
        chosen_path = st.geometry_path

        # executed work: the measuring twin of the same kernel, same launch shape, after the timed region
        executed = None
        exec_flop_per_pass = None
        if walk and not args.no_work_count:
            pt.set_geometry_path(chosen_path)
            # the launch the twin repeats: ppl passes from u_time 0 with the timed kernel (the timed region itself when that
            # was exactly one such launch)
            if steps == spl and not weak:
                seg_timed_launch = int(st.segments)
            else:
                pt.reset()
                q = p.copy()
                q.time = 0.0
                pt.set_params(q)
                pt.render_passes(ppl)
                seg_timed_launch = int(pt.stats().segments)
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
                # same passes, same seeds as one launch of the timed kernel: the twin must have shaded exactly as many segments
                "twin_segments_equal_timed_kernel": int(sw.segments) == seg_timed_launch,
            }
        if not walk:  # the list walks execute exactly the algorithmic tests — and, like the walk kernels, scatter / RNG / camera per segment
            exec_flop_per_pass = alg_flop_per_pass + FLOP_PER_SEGMENT_SHADE * seg_per_pass
        achieved_tf = (exec_flop_per_pass / (ms_per_pass * 1e-3) / 1e12) if (exec_flop_per_pass and ms_per_pass > 0) else None

        # a prior rocprofv3 PMC profile of the same kernel, config and launch shape, if one is committed
        # (profiles/pmc_traffic.json: one record per kernel + config, written by profiles/summarize.py)
        identity = build_identity()
        prior = prior_pmc_record(kernel_name, args.config, args.spp_per_pass, ppl) if world == 1 else None
        # (this run's time of a launch of the record's shape: kernel time per pass x passes per launch — under --steps 20 the timed
        # region's launches are one of 64 passes and one of 16, whose plain average would not compare with the record's 64)
        counters = counters_of(prior, ms_per_pass * ppl, identity)
        prior = fresh(prior, identity)  # (a record of another build: no traffic figure either)
        roofline = {
            "kernel": kernel_name,
            "geometry_path": abi.GEOM_NAMES.get(st.geometry_path, "?") + (" (autotuned)" if st.geometry_tuned else ""),
            "bound": "valu",
            "achieved": round(achieved_tf, 3) if achieved_tf is not None else None,
            "peak": FP32_VALU_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": round(achieved_tf / FP32_VALU_PEAK_TFLOPS, 4) if achieved_tf is not None else None,
            "frac_is": "a MODEL: executed units from the measuring twin's tallies x the flop_model weights below; the counter-derived "
                       "figure of the same kernel is counters.fp32_flop_frac",
            "traffic": prior.get("hbm_bytes_per_pass") if prior else None,
            "traffic_source": ("PRIOR profile %s (rocprofv3 --pmc WRITE_SIZE + 2*FETCH_SIZE, same kernel and launch shape), per pass"
                               % prior.get("profile", "profiles/pmc_traffic.json")) if prior else None,
            "counters": counters,
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
            k_lw = min(steps, spl)
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
                "sec_to_converged_frame": round((tl1 - tl0) / k_lw * (converged_spp / spp_step), 4),
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

            strip = min(args.cpu_strip or cfg["cpu_strip"], p.width)
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
                "algorithm": "linear list walk (static/shader.frag:175-196), scalar code, pthread workers over rows",
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
            "steps": steps,
            "warmup": args.warmup,
            "untimed_busy_ms_before_the_timed_region": round(clock_warmup_ms, 1) if clock_warmup_ms is not None else None,
            "ms_per_step": round(wall / steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak" if weak else "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "%s: %s (%d spheres), %dx%d, %d bounces, %d spp/step (%d passes x %d spp) x %d steps = %d spp"
                            % (cfg["name"], cfg["what"], n_sph, p.width, p.height, args.max_depth, spp_step, pps, args.spp_per_pass, steps, spp_step * steps),
                "partition": "%d rank(s), interleaved %d-row bands, one all_gather at the end of the timed region%s"
                             % (world, band_rows, "; per-GPU work fixed: 1/%d of the rows x %d passes per step" % (world, pps)
                                if weak else ("; the same frame for every N" if world > 1 else "")),
                "steps_per_launch": spl,
                "passes_per_launch": ppl,
                "spp_per_pass": args.spp_per_pass,
            },
            "sec_to_converged_frame": round(wall / steps * (converged_spp / spp_step), 4),
            "converged_frame_spp": converged_spp,
            "gather_matches_single_gpu": gather_check["matches"] if gather_check else None,  # None only at non-default sizes
            "gather_check": gather_check,
            "weak_series": weak_series,
            "first_frame_ms": first_frame["ms"] if first_frame else None,
            "first_frame": first_frame,
            "segments": int(segments),
            "nominal_mray_s": round(p.width * p.height * spp_step * steps * args.max_depth / wall / 1e6, 1),
            "per_rank_render_kernel_ms": per_rank_kernel_ms,
            "roofline": roofline,
            "list_walk": list_walk,
            "cpu_baseline": cpu,
            "build": identity,
        }
        if cpu:
            out["gpu_over_cpu"] = round(mrays / cpu["value"], 1) if cpu["value"] else None
            out["list_walk_over_cpu"] = round(list_walk["value"] / cpu["value"], 1) if (list_walk and cpu["value"]) else None
            out["gpu_over_cpu_note"] = ("gpu_over_cpu divides the fastest GPU path (culling structure) by a CPU that walks the whole list; "
                                        "list_walk_over_cpu is like for like (both test every sphere)")
        print(json.dumps(out), flush=True)
        assert tuple(full.shape) == (p.height, p.width, 4)
    pt.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def prior_pmc_record(kernel, config, spp_per_pass, passes_per_launch):
    """The committed rocprofv3 PMC record of this kernel, config and launch shape (profiles/pmc_traffic.json), or None."""
    prof = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        doc = json.load(open(prof))
    except Exception:
        return None
    for rec in (doc.get("records") or [doc]):
        if (rec.get("kernel") == kernel and str(rec.get("config", "2")) == str(config) and
                rec.get("spp_per_pass") == spp_per_pass and rec.get("passes_per_launch") == passes_per_launch):
            return rec
    return None


def build_identity():
    from ray_tracer_webgl_amd import _lib

    return _lib.build_identity()


def counters_of(prior, this_kernel_ms=None, identity=None):
    """Counter-derived figures of a PRIOR profile record (profiles/summarize.py wrote it from the PMC passes):
         valu_issue_frac  = 2 x SQ_INSTS_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)
         lane_utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)
         fp32_flop_frac   = (2 FMA + MUL + ADD) x 64 x lane_utilisation / kernel time / peak
    A record is only as good as the build it was collected on: it carries the sha256 of the sources of that build
    (`csrc_sha256`, _lib.build_identity), and a record of ANOTHER build — or of none: the records of rounds 1-5 — is refused:
    `stale` true, no figures.  `prior_kernel_ms / this_run_kernel_ms` is printed either way (HIP events of this run against
    the profiler's kernel trace of the record)."""
    if not prior:
        return None
    identity = identity or build_identity()
    ratio = round(prior["kernel_ms"] / this_kernel_ms, 4) if (prior.get("kernel_ms") and this_kernel_ms) else None
    head = {
        "source": "PRIOR profile %s (rocprofv3 --pmc, same kernel and launch shape; not this run)" % prior.get("profile", "profiles/pmc_traffic.json"),
        "record_csrc_sha256": prior.get("csrc_sha256"),
        "this_build_csrc_sha256": identity["csrc_sha256"],
        "kernel_ms": prior.get("kernel_ms"),
        "this_run_kernel_ms": round(this_kernel_ms, 4) if this_kernel_ms else None,
        "prior_over_this_run_kernel_ms": ratio,
    }
    if prior.get("csrc_sha256") != identity["csrc_sha256"]:
        head["stale"] = True
        head["note"] = ("the record was collected on another build of the kernels (or carries no build identity): its figures are "
                        "not this library's and are not shown; re-collect with profiles/collect.sh")
        return head
    head["stale"] = False
    lu = None
    if prior.get("sq_thread_cycles_valu") and prior.get("sq_active_inst_valu"):
        lu = prior["sq_thread_cycles_valu"] / (64.0 * prior["sq_active_inst_valu"])
    ff32 = None
    if lu and prior.get("kernel_ms") and prior.get("sq_insts_valu_fma_f32") is not None:
        flop = (2.0 * prior["sq_insts_valu_fma_f32"] + prior.get("sq_insts_valu_mul_f32", 0.0) + prior.get("sq_insts_valu_add_f32", 0.0)) * 64.0 * lu
        ff32 = flop / (prior["kernel_ms"] * 1e-3) / 1e12 / FP32_VALU_PEAK_TFLOPS
    head.update({
        "valu_issue_frac": round(prior["valu_issue_frac"], 4) if prior.get("valu_issue_frac") else None,
        "cycles_per_valu_per_simd": round(2.0 / prior["valu_issue_frac"], 3) if prior.get("valu_issue_frac") else None,
        "lane_utilisation": round(lu, 4) if lu else None,
        "fp32_flop_frac": round(ff32, 4) if ff32 else None,
        "fp32_flop_frac_is": "(2 x FMA_F32 + MUL_F32 + ADD_F32 wave-instructions) x 64 x lane_utilisation / kernel time / %.1f TFLOP/s" % FP32_VALU_PEAK_TFLOPS,
        "salu_per_valu": round(prior["sq_insts_salu"] / prior["valu_insts_per_launch"], 3) if prior.get("sq_insts_salu") and prior.get("valu_insts_per_launch") else None,
        "lds_bank_conflict_frac": round(prior["sq_lds_bank_conflict"] / prior["sq_lds_idx_active"], 4) if prior.get("sq_lds_idx_active") else None,
    })
    return head


def fresh(prior, identity=None):
    """the record itself when it was collected on THIS build, else None (its traffic figure is as stale as its counters)"""
    identity = identity or build_identity()
    return prior if (prior and prior.get("csrc_sha256") == identity["csrc_sha256"]) else None


def frame_loop_bench(args):
    """--config default: the reference at its OWN operating point.  State::default (9 spheres, at most
    15 by static/shader.frag:103), 1280x702 (images/14.png; MAX_CANVAS_SIZE 1280, src/dom.rs:13),
    depth 8, one 1-spp frame per animation tick blended into RGBA8 ping-pong textures by the
    shader's render() rule (src/state.rs:127-135, src/lib.rs:65-104, src/webgl.rs:180-205) —
    app.FrameLoop's "reference" mode, with the per-frame work (trace + fold + blend) replayed from
    hipGraphs.  A step is one frame; `value` is frames per second, with Mray/s beside it; the
    25-spp paused mode (src/webgl.rs:342-346) is timed beside it.  Like the main line it carries
    `roofline` (the trace launch of one group of 64 frames: pt_trace_kernel_small_t1, a LIST kernel, so
    executed work = algorithmic work) and `cpu_baseline` (the oracle replaying the same ticks: one
    1-spp pass + the shader's blend per tick, on this host's cores)."""
    from ray_tracer_webgl_amd.app import frame_loop_benchmark

    out = frame_loop_benchmark(args.frames, args.warmup, extra_legs=not args.no_extra_legs)
    g = out["group_trace_kernel"]
    t_launch = g["avg_launch_ms"] * 1e-3
    # a list kernel tests every sphere for every segment (static/shader.frag:175-196): 20 FLOP each, + ~150 per segment of scatter / RNG / camera
    flop = g["segments_per_launch"] * (FLOP_PER_SPHERE_TEST * g["n_spheres"] + FLOP_PER_SEGMENT_SHADE)
    achieved = flop / t_launch / 1e12 if t_launch > 0 else None
    identity = build_identity()
    prior_any = prior_pmc_record(g["kernel"], "default", g["spp_per_pass"], g["passes_per_launch"])
    prior = fresh(prior_any, identity)
    n_pix, k = g["pixels"], g["passes_per_launch"]
    # HBM bytes of ONE GROUP of frames as built: the trace kernel stores one 16-B slab entry per (pixel, frame); the blend kernel
    # reads them, reads the previous RGBA8 texture and writes the last two frames' textures and the canvas (src/webgl.rs:186-204)
    group_bytes = {"trace_slab_stores": 16 * n_pix * k, "blend_slab_loads": 16 * n_pix * k, "blend_rgba8_textures": 4 * n_pix * 4}
    group_ms = out["animation"]["device_ms_per_frame"] * k if out["animation"].get("device_ms_per_frame") else None
    out["roofline"] = {
        "kernel": g["kernel"],
        "geometry_path": "small (the whole list from SGPRs; lists of <= 16 spheres)",
        "bound": "valu",
        "achieved": round(achieved, 3) if achieved else None,
        "peak": FP32_VALU_PEAK_TFLOPS,
        "unit": "TFLOP/s",
        "frac": round(achieved / FP32_VALU_PEAK_TFLOPS, 4) if achieved else None,
        "frac_is": "algorithmic = executed for a list kernel: segments x (20 FLOP x %d spheres + %d scatter / RNG / camera) per launch / "
                   "the launch's duration (HIP events on the launch stream around the kernel alone, %d launches of the group's own shape "
                   "through pt_render_passes after the timed region)" % (g["n_spheres"], FLOP_PER_SEGMENT_SHADE, g["launches"]),
        "avg_launch_ms": g["avg_launch_ms"],
        "launches": g["launches"],
        "passes_per_launch": k,
        "spp_per_pass": g["spp_per_pass"],
        "segments_per_launch": g["segments_per_launch"],
        "gray_s_of_the_kernel": round(g["segments_per_launch"] / t_launch / 1e9, 2) if t_launch > 0 else None,
        "traffic": prior.get("pt_trace_kernel_hbm_bytes_per_launch") if prior else None,
        "traffic_source": ("PRIOR profile %s (rocprofv3 --pmc WRITE_SIZE + 2*FETCH_SIZE of the group's trace launch)" % prior.get("profile")) if prior else None,
        "counters": counters_of(prior_any, g["avg_launch_ms"], identity),
        "hbm": {
            "algorithmic_bytes_per_group": group_bytes,
            "achieved": round(sum(group_bytes.values()) / (group_ms * 1e-3) / 1e9, 1) if group_ms else None,
            "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(sum(group_bytes.values()) / (group_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if group_ms else None,
            "note": "all three kernels of a group (trace, blend, advance) over the group's device time: the loop is not HBM-bound either",
        },
    }
    out["build"] = identity
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = frame_loop_cpu_baseline(out)
        if out["cpu_baseline"] and out["cpu_baseline"]["value"]:
            out["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
    print(json.dumps(out), flush=True)
    return 0


def frame_loop_cpu_baseline(line, budget_s=12.0, min_frames=24):
    """The CPU oracle replaying the animation loop's ticks as the reference issues them (src/lib.rs:65-104): per tick one 1-spp
    pass of State::default at u_time = now, then the shader's render() blend with the previous RGBA8 texture
    (static/shader.frag:387-404) — oracle.render + oracle.blend_rgba8, the checker's own functions, pthread workers over rows on
    this host's usable cores.  Bounded: whole frames until `budget_s` seconds have passed."""
    import numpy as np

    cores = usable_cores()
    flags, native = build_native_oracle()
    from oracle import oracle
    from ray_tracer_webgl_amd.state import State

    w, h = 1280, 702
    st = State(w, h)
    st.set_flags(is_paused=False)
    spheres = st.spheres()
    tex = [np.zeros((h, w, 4), np.uint8), np.zeros((h, w, 4), np.uint8)]
    now0, dt = 3000.0, 16.7
    frames, seg_total = 0, 0
    t0 = time.perf_counter()
    while True:
        now = now0 + dt * frames
        st.update_position(now if frames == 0 else dt)
        st.update_render_globals()
        v, p = st.view(), st.to_params(now)
        acc, seg = oracle.render(spheres, p, 1, nthreads=cores)
        tex[v.even_odd_count % 2] = oracle.blend_rgba8(acc, p.samples_per_pixel, p, tex[(v.even_odd_count + 1) % 2])
        frames += 1
        seg_total += seg
        if frames >= min_frames and time.perf_counter() - t0 >= budget_s:
            break
    t1 = time.perf_counter()
    st.close()
    return {
        "value": round(frames / (t1 - t0), 2),
        "unit": "frames/s",
        "mray_s": round(seg_total / (t1 - t0) / 1e6, 2),
        "cores": cores,
        "kind": "port",
        "algorithm": "the oracle's frame loop: one 1-spp pass (linear list walk, scalar code, pthread workers over rows) + the shader's RGBA8 blend per tick",
        "build": flags,
        "native_build": native,
        "sample": "%d whole 1280x702 frames of the same loop (%d segments, %.1f s)" % (frames, seg_total, t1 - t0),
    }


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
