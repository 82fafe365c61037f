"""Dev tool (GPU): would a per-workgroup LDS cache of the HOTTEST cells' entry runs pay on BASELINE config 5?

    python tools/config5_cache_model.py            (the measuring twin's gather statistics + the model)
    python tools/config5_cache_model.py --time     (one timing run of the product kernel; PT_LIB / PT_PER_CU / PT_BVH_BLOCK from the environment)

Config 5 (10 000-sphere field) runs pt_trace_kernel_grid_cells: the cell records (22 KB) are staged in the LDS, the entry runs
(24 k entries x 16 B = 0.4 MB) are gathered per lane from L2 — 4.4 L2 requests per wave-level load, leaf + exact 43-46 % of the
wave time (profiles/r04_phase_clocks.txt).  The structural alternative VERDICT r4 #4 asks to price: give up one of the three
512-thread workgroups per CU (6 -> 4 waves per SIMD) and use the freed LDS as a cache of the most-visited cells' runs.

Measured here, with the twin (every 8th wave sampled; include/ptrace_dev.h pt_debug_cell_hist):
  * how the leaf rounds' lanes distribute over the runs -> the hit rate of a cache of the best runs for a byte budget
    (greedy by visits per byte; an upper bound for any static choice);
  * how many DISTINCT runs one leaf round's lanes gather (the coherence of one wave-level gather);
and, with the product kernel in fresh child processes, what the occupancy the cache would cost is worth: 3 / 2 workgroups of 512
threads per CU and 1 of 1024 (the PT_DEV_KNOBS build's PT_PER_CU / PT_BVH_BLOCK).
The model is OPTIMISTIC on purpose: every cached lane-gather is assumed to take its whole share of the leaf + exact wave time away
(in reality a wave waits for its slowest lane, so a round is only faster when ALL its lanes hit).
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ray_tracer_webgl_amd import abi, scenes  # noqa: E402
from ray_tracer_webgl_amd.tracer import PathTracer  # noqa: E402

PASSES, SPP = 16, 16


def scene():
    sc = scenes.config5(1920, 1080, SPP, PASSES, 50)
    sc.params.time_step = abi.PT_TIME_STEP_DECORRELATED
    return sc


def time_once():
    sc = scene()
    pt = PathTracer(1920, 1080)
    pt.set_spheres(sc.spheres)
    pt.set_params(sc.params)
    pt.reserve_passes(PASSES)
    pt.set_geometry_path(abi.PT_GEOM_GRID)
    ms = []
    for rep in range(4):
        pt.reset()
        pt.render_passes(PASSES)
        if not pt.wait(60.0):
            print("WATCHDOG", file=sys.stderr, flush=True)
            os._exit(3)
        ms.append(pt.stats().render_kernel_ms)
    print("%.3f" % min(ms[1:]), flush=True)
    pt.close()


def timed(env):
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--time"], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    if r.returncode != 0:
        raise SystemExit("timing child failed (%d): %s" % (r.returncode, r.stderr[-800:]))
    return float(r.stdout.strip().splitlines()[-1])


def host_grid(spheres):
    """The grid as the gathering kernels see it: cell records and entry runs in Morton order (pt_build_grid_runs)."""
    from ray_tracer_webgl_amd import _lib
    lib = _lib.load()
    fn = lib.pt_build_grid_runs
    fn.restype, fn.argtypes = lib.pt_build_grid.restype, lib.pt_build_grid.argtypes
    ptr, n, keep = abi.spheres_as_ctypes(spheres)
    counts, geom, margin, dg = np.zeros(8, np.uint32), np.zeros(12, np.float32), np.zeros(4, np.float32), C.c_float(0)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    assert fn(ptr, n, vp(counts), vp(geom), vp(margin), C.byref(dg), None, 0, None, 0, None, 0) == 0
    cells = np.zeros(int(counts[0]) * int(counts[1]) * int(counts[2]), np.uint32)
    entries = np.zeros((counts[5], 4), np.float32)
    index = np.zeros(counts[5], np.uint32)
    assert fn(ptr, n, vp(counts), vp(geom), vp(margin), C.byref(dg), vp(cells), cells.size, vp(entries), entries.size, vp(index), index.size) == 0
    return counts, cells


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--time":
        return time_once()
    sc = scene()
    counts, cells = host_grid(sc.spheres)
    first, length = (cells & np.uint32(0xffffff)).astype(np.int64), (cells >> np.uint32(24)).astype(np.int64)
    nonempty = length > 0
    print("# config 5: grid %d x %d x %d = %d cells (%d non-empty), %d entries (%d in cells, %d always-tested), %.1f entries per non-empty cell, longest run %d"
          % (counts[0], counts[1], counts[2], cells.size, int(nonempty.sum()), counts[5], counts[3], counts[4], length[nonempty].mean(), length.max()))

    pt = PathTracer(1920, 1080)
    pt.set_spheres(sc.spheres)
    pt.set_params(sc.params)
    pt.reserve_passes(PASSES)
    pt.set_geometry_path(abi.PT_GEOM_GRID)
    pt.render_passes(PASSES)  # tile order
    pt.reset()
    pt.set_count_work(True)   # the plain twin first: its phase clock is not disturbed by the histogram's atomics
    pt.render_passes(PASSES)
    ctr = np.zeros(128, np.uint64)
    pt.lib.pt_debug_counters.restype = C.c_long
    pt.lib.pt_debug_counters.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    pt.lib.pt_debug_counters(pt._ctx, ctr.ctypes.data_as(C.c_void_p), 128)
    ph = ctr[24:32].astype(np.float64)
    leaf_share = float(ph[5] / ph.sum())
    pt.reset()
    pt.set_count_work(2)      # ... then the twin with the gather histogram
    pt.render_passes(PASSES)
    st = pt.stats()
    n_ent = int(st.grid_entries)
    buf = np.zeros(n_ent + 66, np.uint32)
    pt.lib.pt_debug_cell_hist.restype = C.c_long
    pt.lib.pt_debug_cell_hist.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    k = pt.lib.pt_debug_cell_hist(pt._ctx, buf.ctypes.data_as(C.c_void_p), buf.size)
    assert k == buf.size, k
    pt.close()
    hist, coh, lanes = buf[:n_ent].astype(np.float64), buf[n_ent:n_ent + 65].astype(np.float64), float(buf[n_ent + 65])
    rounds = coh.sum()
    print("# the twin (%d x %d spp, every 8th wave): %.4g sampled leaf rounds with %.1f lanes on average; leaf + exact = %.1f %% of the wave time (phase clock)"
          % (PASSES, SPP, rounds, lanes / max(rounds, 1), 100 * leaf_share))
    mean_distinct = float((coh * np.arange(65)).sum() / max(rounds, 1))
    print("# coherence of ONE wave-level gather: %.1f distinct entry runs among a leaf round's lanes on average (lanes / runs = %.2f); "
          "rounds whose lanes all read ONE run: %.1f %%, at most 4 runs: %.1f %%, more than 16: %.1f %%"
          % (mean_distinct, lanes / max(rounds, 1) / max(mean_distinct, 1e-9), 100 * coh[1] / rounds, 100 * coh[1:5].sum() / rounds, 100 * coh[17:].sum() / rounds))
    # visits per cell: a cell's rounds start at first, first + 4, ... (each round reads four entries)
    visits = np.zeros(cells.size)
    for c in np.nonzero(nonempty)[0]:
        visits[c] = hist[first[c]:first[c] + length[c]:4].sum()
    bytes_of = ((length + 3) // 4 * 4) * 16.0
    order = np.argsort(-(visits / np.maximum(bytes_of, 1.0)))
    cum_b, cum_v = np.cumsum(bytes_of[order]), np.cumsum(visits[order])
    total_v = visits.sum()
    share_sorted = np.sort(visits[nonempty])[::-1] / total_v
    print("# how the gathers spread over the %d non-empty cells: the hottest 1 %% of the cells take %.1f %% of the leaf-round lanes, the hottest 10 %% %.1f %%, the hottest 25 %% %.1f %%"
          % (int(nonempty.sum()), 100 * share_sorted[: max(1, len(share_sorted) // 100)].sum(), 100 * share_sorted[: len(share_sorted) // 10].sum(),
             100 * share_sorted[: len(share_sorted) // 4].sum()))

    print("\n## what the occupancy is worth (product kernel pt_trace_kernel_grid_cells, %d x %d spp, min of 3 launches, fresh process each)" % (PASSES, SPP))
    knobs = os.path.join(ROOT, "build_ab", "libptrace_knobs.so")
    t3 = timed({"PT_LIB": knobs})
    t2 = timed({"PT_LIB": knobs, "PT_PER_CU": "2"})
    t1k = timed({"PT_LIB": knobs, "PT_BVH_BLOCK": "1024"})
    print("  3 workgroups of 512 threads per CU (6 waves per SIMD; as shipped): %.2f ms" % t3)
    print("  2 workgroups of 512 threads per CU (4 waves per SIMD):             %.2f ms  (x %.3f)" % (t2, t2 / t3))
    print("  1 workgroup of 1024 threads per CU (4 waves per SIMD):             %.2f ms  (x %.3f)" % (t1k, t1k / t3))

    print("\n## the cache's hit rate for the LDS each shape frees (cells 22 KB + parking 60 B per lane stay), greedy by visits per byte")
    print("   shape                                   cache per workgroup   cells cached   hit rate   optimistic time   vs 6 waves")
    lds_cu, cells_b, park = 160 * 1024, 22 * 1024, 60
    for name, wgs, threads, t_shape in (("2 x 512 threads, 4 waves per SIMD", 2, 512, t2), ("1 x 1024 threads, 4 waves per SIMD", 1, 1024, t1k),
                                         ("3 x 512 threads, 6 waves per SIMD", 3, 512, t3)):
        free = lds_cu // wgs - cells_b - park * threads
        n_fit = int(np.searchsorted(cum_b, max(free, 0), side="right"))
        hit = cum_v[n_fit - 1] / total_v if n_fit > 0 else 0.0
        t_opt = t_shape * (1.0 - hit * leaf_share)
        print("   %-38s %8.1f KB          %6d       %5.1f %%     %8.2f ms       x %.3f" % (name, max(free, 0) / 1024.0, n_fit, 100 * hit, t_opt, t_opt / t3))
    print("   (optimistic time = the shape's measured time x (1 - hit rate x the leaf + exact share of the wave time): every cached LANE-gather is credited\n"
          "    with its whole share, although a wave-level gather is only as fast as its slowest lane — with %.1f distinct runs per round a round is\n"
          "    all-hits with probability ~ hit rate ^ %.0f)" % (mean_distinct, mean_distinct))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
