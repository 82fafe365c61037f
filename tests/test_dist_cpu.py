"""The N>1 path on CPU: world_size-2 (and 3, and 8 with bench.py's 4-row bands) `gloo` process groups exercise the row-band
partition and the all_gather/de-interleave of ray_tracer_webgl_amd/dist.py.  The per-rank render
is injected (the oracle stands in for the HIP path here — tests may do that, the product never
does), so what is under test is exactly the multi-rank logic bench.py runs over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, band_rows, width, height, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle
        from ray_tracer_webgl_amd import abi, dist as ptdist, scenes

        sc = scenes.config1(width, height, 2, 6)
        sc.n_passes = 2

        def render_fn(scene, params):
            return oracle.render(scene.spheres, params, scene.n_passes, nthreads=2)

        local, seg, _ = ptdist.render_band(sc, rank, world, band_rows, render_fn=render_fn)
        assert local.shape[0] == abi.local_rows(height, band_rows, rank, world)
        full = ptdist.gather_rows(local, height, band_rows, rank, world)
        again = ptdist.gather_rows(local, height, band_rows, rank, world)  # cached buffers and permutation
        assert torch.equal(full, again) and full.data_ptr() != again.data_ptr()
        segs = torch.tensor([seg], dtype=torch.int64)
        dist.all_reduce(segs)
        np.save(os.path.join(out_dir, "full_%d.npy" % rank), full.numpy())
        if rank == 0:
            np.save(os.path.join(out_dir, "segs.npy"), segs.numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,band_rows,width,height", [(2, 8, 48, 37), (3, 4, 40, 30), (2, 8, 32, 5), (8, 4, 24, 70)])
def test_row_band_render_and_gather_gloo(tmp_path, world, band_rows, width, height, ora):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, band_rows, width, height, str(tmp_path)), nprocs=world, join=True)
    from ray_tracer_webgl_amd import scenes

    sc = scenes.config1(width, height, 2, 6)
    ref, seg = ora.render(sc.spheres, sc.params, 2)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "full_%d.npy" % r))
        assert got.shape == ref.shape
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), "rank %d image differs" % r
    assert int(np.load(os.path.join(str(tmp_path), "segs.npy"))[0]) == seg


def test_gather_world_size_one():
    from ray_tracer_webgl_amd import dist as ptdist

    local = torch.arange(5 * 3 * 4, dtype=torch.float32).reshape(5, 3, 4)
    full = ptdist.gather_rows(local, 5, 8, 0, 1)
    assert torch.equal(full, local)


def test_band_helpers():
    from ray_tracer_webgl_amd import abi, dist as ptdist

    assert ptdist.band_of(3, 8) == (8, 3, 8)
    assert ptdist.max_local_rows(1080, 8, 8) == 136  # 135 bands of 8 rows: ranks 0..6 get 17
    assert sum(abi.local_rows(1080, 8, r, 8) for r in range(8)) == 1080
    assert ptdist.max_local_rows(5, 8, 2) == 5


def test_assemble_rows_is_the_collectives_layout_at_the_eight_rank_shape():
    """dist.assemble_rows — the gather of ONE process that drives every rank's context (a host like examples/render_bands.c;
    the one-device rehearsal of an 8-rank run, tools/rehearse_ranks.py) — lays the parts out as all_gather_into_tensor lays out
    the ranks' padded buffers and de-interleaves with the same permutation (band_layout): 1080 rows in 4-row bands over eight
    ranks are 34 x 6 / 33 x 2 bands, so two ranks send a band of padding.  Every image row must come back from the rank
    that owns it, whatever lies in the parts beyond a rank's own rows."""
    from ray_tracer_webgl_amd import abi, dist as ptdist

    height, width, band, world = 1080, 3, 4, 8
    shares = [abi.local_rows(height, band, r, world) for r in range(world)]
    assert shares == [136] * 6 + [132] * 2 and ptdist.band_layout(height, band, world)[0] == 136
    image = torch.arange(height * width * 4, dtype=torch.float32).reshape(height, width, 4)
    parts = []
    for r in range(world):
        ys = torch.as_tensor(np.asarray(abi.owned_rows(height, band, r, world), dtype=np.int64))
        part = torch.full((140, width, 4), -1.0)  # (more rows than the rank owns, junk behind them: a caller's buffer may be larger)
        part[: len(ys)] = image[ys]
        parts.append(part)
    assert torch.equal(ptdist.assemble_rows(parts, height, band), image)
    pad_rows, perm = ptdist.band_layout(height, band, world)
    assert sorted(perm.tolist()) == sorted(r * pad_rows + l for r in range(world) for l in range(shares[r]))
    # the degenerate partitions: one rank; more ranks than bands (ranks without a row)
    assert torch.equal(ptdist.assemble_rows([image], height, band), image)
    small = image[:5]
    parts = [small[np.asarray(abi.owned_rows(5, 4, r, 3), dtype=np.int64)] if abi.local_rows(5, 4, r, 3) else torch.zeros((0, width, 4)) for r in range(3)]
    assert torch.equal(ptdist.assemble_rows(parts, 5, 4), small)


def _run_bench(*extra):
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--backend", "gloo"] + list(extra),
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r.returncode, [json.loads(ln) for ln in lines], r.stderr


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` from a bare shell (no torchrun, WORLD_SIZE unset): the parent
    starts N ranks, they rendezvous on 127.0.0.1, rank 0's JSON line comes back through the
    parent.  --spawn-selftest keeps the ranks off the GPU, so this runs on CPU."""
    rc, lines, err = _run_bench("--gpus", "3", "--spawn-selftest", "ok")
    assert rc == 0, err[-2000:]
    assert len(lines) == 1 and lines[0]["ranks"] == 3 and lines[0]["rank_sum"] == 6.0 and lines[0]["local_rank"] == 0


def test_bench_reports_a_failed_rank():
    rc, lines, err = _run_bench("--gpus", "2", "--spawn-selftest", "fail")
    assert rc != 0


def test_bench_workload_plan_and_digest_helpers():
    """The N-independent workload bench.py times by default (strong scaling): config 2 = 16 steps of
    4 x 16 spp = 1024 spp, config 3 = 64 steps = 4096 spp, whatever the rank count; the frame
    digest is the one tests/golden/make_full_digests.py records (sha256 of the fp32 bytes)."""
    import hashlib
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.plan_steps(1024, 16, 4, None) == (16, 64)
    assert bench.plan_steps(4096, 16, 4, None) == (64, 64)
    assert bench.plan_steps(1024, 16, 4, 5) == (5, 64)
    for key in ("2", "3"):
        cfg = bench.BENCH_CONFIGS[key]
        assert cfg["digest"] in bench.load_digests(), "no committed digest for --config %s" % key
    a = np.arange(2 * 3 * 4, dtype=np.float32).reshape(2, 3, 4)
    assert bench.frame_digest(a) == hashlib.sha256(a.tobytes()).hexdigest()
    assert bench.frame_digest(torch.from_numpy(a)) == bench.frame_digest(a)
    ap_default = [ln for ln in open(os.path.join(ROOT, "bench.py")) if '"--scaling"' in ln]
    assert ap_default and 'default="strong"' in ap_default[0]


def _pids_started(err):
    import re

    m = re.search(r"started ranks (.*)", err)
    assert m, err[-2000:]
    return [int(x) for x in re.findall(r"pid(\d+)", m.group(1))]


def _gone(pid):
    """No such process any more (a zombie cannot be: the parent reaped its children, and it has ended too)."""
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return True
    except PermissionError:
        return False
    try:
        return open("/proc/%d/stat" % pid).read().split(")")[-1].split()[0] == "Z"
    except OSError:
        return True


@pytest.mark.parametrize("mode,code", [("die-early", 5), ("die-in-collective", 7)])
def test_bench_fails_fast_when_a_rank_dies_where_rank0_cannot_see_it(mode, code):
    """a rank that exits BEFORE the rendezvous (rank 0 sits in init_process_group) or
    while rank 0 waits for it in a collective must end the whole job in seconds — with its code,
    its number and its last words — and leave no process behind."""
    import time

    t0 = time.time()
    rc, lines, err = _run_bench("--gpus", "3", "--spawn-selftest", mode)
    took = time.time() - t0
    assert rc == code, (rc, err[-2000:])
    assert took < 30.0, "the parent took %.1f s to notice the dead rank" % took
    assert not lines, "no JSON line from a failed job"
    assert "rank 2" in err and "ended with code %d" % code in err and "[rank 2] selftest: rank 2 leaves" in err, err[-2000:]
    pids = _pids_started(err)
    assert len(pids) == 3
    deadline = time.time() + 5.0
    while time.time() < deadline and not all(_gone(p) for p in pids):
        time.sleep(0.1)
    assert all(_gone(p) for p in pids), "orphan rank processes: %s" % [p for p in pids if not _gone(p)]


def test_bench_job_timeout_ends_ranks_that_never_finish():
    """A live rank that hangs (here: rank 0 waits in a collective, the dead rank's exit code is 0 so
    nothing looks failed) is ended by --job-timeout, not by the driver's run limit."""
    import time

    t0 = time.time()
    rc, lines, err = _run_bench("--gpus", "2", "--spawn-selftest", "hang", "--job-timeout", "8")
    assert rc == 124 and time.time() - t0 < 40.0, (rc, err[-2000:])
    assert all(_gone(p) for p in _pids_started(err))


def test_ipc_mode_is_set_for_every_launch_style():
    """HSA_ENABLE_IPC_MODE_LEGACY=0 reaches self-started ranks AND a rank started by a launcher
    (bench.py sets it in main() before torch is imported; a caller's own value is kept)."""
    rc, lines, err = _run_bench("--gpus", "2", "--spawn-selftest", "ok")
    assert rc == 0 and lines[0]["ipc_mode_legacy"] == "0", err[-2000:]
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k != "HSA_ENABLE_IPC_MODE_LEGACY"}
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--spawn-selftest", "ok"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    line = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")][0]
    assert line["ipc_mode_legacy"] == "0"
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "1"
    env["MASTER_PORT"] = str(_free_port())
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--spawn-selftest", "ok"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    line = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")][0]
    assert line["ipc_mode_legacy"] == "1"


def test_frame_loop_cpu_baseline_replays_the_reference_ticks():
    """bench.py --config default's `cpu_baseline` leg (the oracle replaying the animation loop: one 1-spp pass +
    the shader's blend per tick) on a tiny budget: whole frames, frames/s, the cores it used.  CPU only."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    keep = os.environ.get("PT_ORACLE_LIB")
    try:
        c = bench.frame_loop_cpu_baseline({}, budget_s=0.2, min_frames=2)
    finally:
        if keep is None:
            os.environ.pop("PT_ORACLE_LIB", None)
        else:
            os.environ["PT_ORACLE_LIB"] = keep
    assert c["unit"] == "frames/s" and c["value"] > 0 and c["mray_s"] > 0 and c["cores"] >= 1 and c["kind"] == "port"
    assert c["sample"].split()[0].isdigit() and int(c["sample"].split()[0]) >= 2


def test_a_counter_record_of_another_build_is_refused():
    """VERDICT r5 #4.  bench.py attaches the committed rocprofv3 counter record of the timed kernel (profiles/pmc_traffic.json)
    to its line, labelled PRIOR.  A record is only as good as the build it was collected on: profiles/summarize.py stamps it
    with the sha256 of the kernel sources the profiled bench line reported (_lib.build_identity), and a record of another
    build — or of none, like the records of rounds 1-5 — is refused: `stale` true, no figures, no traffic; the ratio of the
    record's kernel time to this run's is printed either way.  A record of THIS build yields the figures, with the fp32
    fraction counting FMA, MUL and ADD (the figure the committed summaries are recomputed to).  CPU only."""
    import importlib.util

    from ray_tracer_webgl_amd import _lib

    spec = importlib.util.spec_from_file_location("bench_mod3", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    ident = _lib.build_identity()
    assert len(ident["csrc_sha256"]) == 64 and ident == _lib.build_identity() and ident["abi"] == 5
    rec = {"kernel": "pt_trace_kernel_grid", "config": "2", "kernel_ms": 111.5, "valu_issue_frac": 0.786,
           "sq_thread_cycles_valu": 4.32191e12, "sq_active_inst_valu": 1.07898e11, "sq_insts_valu_fma_f32": 2.33195e10,
           "sq_insts_valu_mul_f32": 1.18834e10, "sq_insts_valu_add_f32": 1.16189e10, "sq_insts_salu": 2.74e10,
           "valu_insts_per_launch": 1.06215e11, "sq_lds_bank_conflict": 1.3e9, "sq_lds_idx_active": 1.32e10,
           "hbm_bytes_per_pass": 35579005, "profile": "profiles/x_summary.txt"}
    for foreign in (dict(rec), dict(rec, csrc_sha256="0" * 64)):
        c = bench.counters_of(foreign, 109.5, ident)
        assert c["stale"] is True and "valu_issue_frac" not in c and "fp32_flop_frac" not in c
        assert c["prior_over_this_run_kernel_ms"] == round(111.5 / 109.5, 4) and c["this_build_csrc_sha256"] == ident["csrc_sha256"]
        assert bench.fresh(foreign, ident) is None
    own = dict(rec, csrc_sha256=ident["csrc_sha256"])
    c = bench.counters_of(own, 109.5, ident)
    assert c["stale"] is False and c["valu_issue_frac"] == 0.786 and bench.fresh(own, ident) is own
    assert abs(c["lane_utilisation"] - 0.6259) < 1e-3
    assert abs(c["fp32_flop_frac"] - 0.160) < 2e-3  # (2 FMA + MUL + ADD): VERDICT r5's recomputation of the round-5 record
    assert bench.counters_of(None, 1.0, ident) is None
    # every record committed under profiles/ names the build it belongs to, or is refused by the line
    import json

    doc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    for r in doc["records"]:
        c = bench.counters_of(r, r.get("kernel_ms"), ident)
        assert c["stale"] == (r.get("csrc_sha256") != ident["csrc_sha256"])


def test_the_committed_bench_lines_and_counter_records_belong_to_this_build():
    """The evidence committed under profiles/ for this round — the bench lines (`r06_bench_config*.json`, the driver-flags line) and
    every record of `pmc_traffic.json` — was collected on the sources in this tree: each names the sha256 of the kernel and host
    sources it ran on (`build.csrc_sha256`, `_lib.build_identity`), the lines carry their kernel's counter record un-refused
    (`counters.stale` false) and a true digest verdict.  A later change to `csrc/` or `include/ptrace*.h` makes this fail until the
    evidence is collected again (`profiles/collect.sh`, `profiles/summarize.py --merge`): that is its purpose.  CPU only."""
    import json

    from ray_tracer_webgl_amd import _lib

    ident = _lib.build_identity()["csrc_sha256"]
    doc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert {str(r["config"]) for r in doc["records"]} == {"2", "3", "4", "5", "default"}
    for r in doc["records"]:
        assert r.get("csrc_sha256") == ident, "the counter record of config %s was collected on another build" % r["config"]
    for name in ("config2", "config3", "config4", "config5", "configdefault", "driver_flags"):
        path = os.path.join(ROOT, "profiles", "r06_bench_%s.json" % name)
        d = json.loads([ln for ln in open(path) if ln.startswith("{")][-1])
        assert d["build"]["csrc_sha256"] == ident, name
        c = d["roofline"]["counters"]
        assert c and c["stale"] is False and c["record_csrc_sha256"] == ident and c["valu_issue_frac"] > 0, name
        assert 0.9 < c["prior_over_this_run_kernel_ms"] < 1.1, (name, c["prior_over_this_run_kernel_ms"])
        if name != "configdefault":
            assert d["gather_matches_single_gpu"] is True and d["gather_check"]["segments_match"] is True, name
