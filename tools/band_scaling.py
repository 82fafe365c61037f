"""Dev tool: what one rank of an N-rank strong-scaling run does, on one GPU: the band partition
of config 2 for N = 1, 2, 4, 8 (rank 0's rows), one launch of the whole 1024-spp frame, per
geometry path and pass shape.

    python tools/band_scaling.py [path ...] [--spp 16|64]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_webgl_amd import dist as ptdist, scenes  # noqa: E402
from ray_tracer_webgl_amd.tracer import PathTracer  # noqa: E402

argv = sys.argv[1:]
spps = [16, 64]
if "--spp" in argv:
    i = argv.index("--spp")
    spps = [int(argv[i + 1])]
    argv = argv[:i] + argv[i + 2:]
band_rows = 8
if "--band-rows" in argv:
    i = argv.index("--band-rows")
    band_rows = int(argv[i + 1])
    argv = argv[:i] + argv[i + 2:]
config3 = "--config3" in argv  # the 4K frame of BASELINE's 8-GPU case: only the N = 8 bands (the whole frame's slabs would be 34 GB)
if config3:
    argv.remove("--config3")
paths = [int(x) for x in (argv or ["4", "3"])]
W, H, TOTAL = (3840, 2160, 4096) if config3 else (1920, 1080, 1024)
for spp in spps:
    passes = TOTAL // spp
    sc = (scenes.config3 if config3 else scenes.config2)(W, H, spp, passes, 50)
    if config3:
        from ray_tracer_webgl_amd import abi
        sc.params.time_step = abi.PT_TIME_STEP_DECORRELATED
    base = None
    for path in paths:
        total_seg = None
        for n, r in ([] if config3 else [(1, 0), (2, 0), (4, 0)]) + [(8, k) for k in range(8)]:
            p = sc.params.copy()
            p.band_rows, p.band_index, p.band_count = ptdist.band_of(r, n, band_rows)
            pt = PathTracer(W, H)
            pt.set_spheres(sc.spheres)
            pt.set_params(p)
            pt.reserve_passes(passes)
            pt.set_geometry_path(path)
            ts = []
            for rep in range(2 if config3 else 4):
                pt.reset()
                t0 = time.perf_counter()
                pt.render_passes(passes)
                pt.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            t = min(ts[1:])
            seg = pt.stats().segments
            if n == 1:
                base, total_seg = t, seg
            if config3:
                print("config 3, band_rows %d  path %d  %d x %d spp  rank %d of %d: %.2f ms, %d segments (%.1f Gray/s)" % (
                    band_rows, path, passes, spp, r, n, t, seg, seg / t / 1e6), flush=True)
                pt.close()
                continue
            print("band_rows %d  path %d  %d x %d spp  rank %d of %d: %.2f ms  (ideal %.2f, efficiency %.3f; %.4f of the segments, "
                  "%.3f of the N=1 rate)" % (band_rows, path, passes, spp, r, n, t, base / n, base / n / t, seg / total_seg,
                                             (seg / t) / (total_seg / base)), flush=True)
            pt.close()
