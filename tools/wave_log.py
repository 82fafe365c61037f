"""Dev tool: when do the waves of a launch start, see the queue dry, and end?  (measuring twin)

    python tools/wave_log.py [ranks] [passes] [spp]      (WL_CONFIG=config5 / config4 / default: that scene — the last two through
                                                          the small-list kernel's twin; WL_DECO=1: decorrelated pass times)
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_webgl_amd import abi, dist as ptdist, scenes  # noqa: E402
from ray_tracer_webgl_amd.tracer import PathTracer  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 64
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 16
which = os.environ.get("WL_CONFIG", "config2")
if which == "config4":      # the closed room through the small-list kernel's twin
    sc = scenes.config4(1024, 1024, spp, passes, 50)
elif which == "default":    # State::default at the reference's own size
    sc = scenes.default_scene(1280, 702, spp, 8, passes)
else:
    sc = (scenes.config5 if which == "config5" else scenes.config2)(1920, 1080, spp, passes, 50)
W, H = sc.params.width, sc.params.height
path = abi.PT_GEOM_SMALL if which in ("config4", "default") else abi.PT_GEOM_GRID
if os.environ.get("WL_DECO"):
    sc.params.time_step = abi.PT_TIME_STEP_DECORRELATED
p = sc.params.copy()
p.band_rows, p.band_index, p.band_count = ptdist.band_of(0, n, 4)
pt = PathTracer(W, H)
pt.set_spheres(sc.spheres)
pt.set_params(p)
pt.reserve_passes(passes)
pt.set_geometry_path(path)
pt.tune(1)  # (the grid fitted to the camera, as bench.py has it)
pt.set_params(p)
for _ in range(2):
    pt.reset()
    pt.render_passes(passes)
pt.reset()
pt.set_count_work(True)
pt.render_passes(passes)
st = pt.stats()
buf = np.zeros((20000, 4), np.uint64)
pt.lib.pt_debug_wave_log.restype = C.c_long
pt.lib.pt_debug_wave_log.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
k = pt.lib.pt_debug_wave_log(pt._ctx, buf.ctypes.data_as(C.c_void_p), len(buf))
w = buf[:k, :3].astype(np.float64) * 1e-5  # ms
t0 = w[:, 0].min()
start, dry, end = w[:, 0] - t0, np.where(w[:, 1] > 0, w[:, 1] - t0, np.nan), w[:, 2] - t0
T = end.max()
print("%d waves, kernel %.2f ms (events %.2f ms); starts within %.3f ms; queue dry first seen at %.2f ms (median %.2f)"
      % (k, T, st.render_kernel_ms, start.max(), np.nanmin(dry), np.nanmedian(dry)))
for q in (0.5, 1.0, 1.5, 2.0, 3.0, 4.0):
    print("  waves still running %.1f ms before the end: %5d of %d" % (q, int((end > T - q).sum()), k))
# ---- residency: which of the launched waves were on the machine from the start, and where (HW_ID | XCC_ID << 32) ----
hw = buf[:k, 3]
slot, simd, cu = (hw & np.uint64(15)).astype(int), ((hw >> np.uint64(4)) & np.uint64(3)).astype(int), ((hw >> np.uint64(8)) & np.uint64(15)).astype(int)
sh, se, xcc = ((hw >> np.uint64(12)) & np.uint64(1)).astype(int), ((hw >> np.uint64(13)) & np.uint64(7)).astype(int), ((hw >> np.uint64(32)) & np.uint64(15)).astype(int)
first_end = end.min()
late = start > 0.05
print("  residency: %d of %d waves start within 0.05 ms; %d start later, %d of them after the first wave has ENDED (%.3f ms)"
      % (int((~late).sum()), k, int(late.sum()), int((start > first_end).sum()), first_end))
edges = [0.0, 0.05, 0.25 * T, 0.5 * T, 0.75 * T, 0.9 * T, 0.95 * T, T + 1e-9]
hist, _ = np.histogram(start, bins=edges)
print("  start-time histogram (ms): " + ", ".join("[%.2f, %.2f) %d" % (edges[i], edges[i + 1], hist[i]) for i in range(len(hist))))
cu_key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
simd_key = cu_key * 4 + simd
res_cu = np.bincount(cu_key[~late], minlength=1)
res_simd = np.bincount(simd_key[~late], minlength=1)
cus = np.unique(cu_key)
print("  at t = 0: %d distinct CUs over %d XCCs hold waves; waves per CU: %s; waves per SIMD: %s"
      % (int((res_cu > 0).sum()), len(np.unique(xcc[~late])),
         dict(zip(*[x.tolist() for x in np.unique(res_cu[res_cu > 0], return_counts=True)])),
         dict(zip(*[x.tolist() for x in np.unique(res_simd[res_simd > 0], return_counts=True)]))))
print("  distinct (XCC, SE, SH, CU, SIMD, slot) at t = 0: %d (= waves really resident; launched %d); wave slots in use: %s"
      % (len(np.unique(simd_key[~late] * 16 + slot[~late])), k, sorted(np.unique(slot[~late]).tolist())))
per_xcc = {int(x): int(((xcc == x) & ~late).sum()) for x in np.unique(xcc)}
per_se = {"%d.%d" % (x, s_): int(((xcc == x) & (se == s_) & ~late).sum()) for x in np.unique(xcc)[:2] for s_ in np.unique(se)}
print("  resident waves per XCC: %s; per (XCC.SE) of the first two XCCs: %s; CUs per (XCC.SE): %s"
      % (per_xcc, per_se, {"%d.%d" % (x, s_): len(np.unique(cu_key[(xcc == x) & (se == s_)])) for x in np.unique(xcc)[:2] for s_ in np.unique(se)}))
if late.any():
    wg = np.arange(k) // (8 if path == abi.PT_GEOM_GRID else 4)  # waves per workgroup: 512-thread walk kernels, 256-thread small-list ones
    print("  late waves: workgroup numbers %d ... %d (of %d); they start at %.3f ... %.3f ms; late waves per CU: %s"
          % (wg[late].min(), wg[late].max(), wg.max() + 1, start[late].min(), start[late].max(),
             dict(zip(*[x.tolist() for x in np.unique(np.bincount(cu_key[late])[np.bincount(cu_key[late]) > 0], return_counts=True)]))))
print("  wave end times: median %.2f, 90 %% %.2f, 99 %% %.2f, max %.2f" % (np.median(end), np.quantile(end, 0.9), np.quantile(end, 0.99), T))
lost = (T - end).sum() / (T * k)
print("  wave-time lost to the drain (sum of (T - end) / (T x waves)): %.3f" % lost)
ctr = np.zeros(128, np.uint64)
pt.lib.pt_debug_counters.restype = C.c_long
pt.lib.pt_debug_counters.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
pt.lib.pt_debug_counters(pt._ctx, ctr.ctypes.data_as(C.c_void_p), 128)
if os.environ.get("WL_JSON"):
    import json
    json.dump({"what": "the measuring twin's raw counters (pt_kernel_args.h PT_CTR_*) of ONE launch: %s, %d rank(s), %d passes of %d spp" % (which, n, passes, spp),
               "config": which, "ranks": n, "passes": passes, "spp": spp, "waves": int(k), "segments": int(st.segments), "kernel_ms": st.render_kernel_ms,
               "grid_always": int(st.grid_always), "grid_cells": [int(x) for x in st.grid_cells], "grid_entries": int(st.grid_entries),
               "counters": [int(x) for x in ctr]}, open(os.environ["WL_JSON"], "w"), indent=1)
bins = ctr[64:128].astype(np.float64)
first = int((buf[:k, 0].min() >> np.uint64(16)) & np.uint64(63))
order = [(first + j) % 64 for j in range(64)]
print("  segments per 0.655 ms bin from the first wave's start (Gray/s):")
print("   " + " ".join("%.1f" % (bins[b] / 0.65536e-3 / 1e9) for b in order if bins[b] > 0))
ph = ctr[24:32].astype(np.float64)
names = ["refill", "camera ray", "park + always-tested", "per-ray constants + entry", "advance / node loops", "leaf + exact", "literal + unpark", "shade"]
print("  wave-time shares by phase (s_memtime): " + ", ".join("%s %.3f" % (n_, v / ph.sum()) for n_, v in zip(names, ph)))
w = ctr[8:16].astype(np.float64)
per = 64.0 / max(float(st.segments), 1.0)
print("  per 64 segments: %.2f advance trips x %.1f lanes, %.2f leaf rounds x %.1f, %.2f exact evaluations x %.1f (of them for the always-tested "
      "group: %.2f x %.1f), %.3f wave steps" % (w[0] * per, w[1] / max(w[0], 1), w[2] * per, w[3] / max(w[2], 1), w[4] * per, w[5] / max(w[4], 1),
                                                 float(ctr[21]) * per, float(ctr[22]) / max(float(ctr[21]), 1.0), w[6] * per))
lit = ctr[16:21]
print("  PHASE 3 (literal loop): %d irregular lane-steps, %d handed over by the walk, %d wave steps; last irregular pixel (slab index) %d, last handed-over %d"
      % (lit[0], lit[1], lit[2], int(lit[3]) - 1, int(lit[4]) - 1))
