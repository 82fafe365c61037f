"""Dev tool: what one rank of an N-rank strong-scaling run does, on one GPU: the band partition
of config 2 for N = 1, 2, 4, 8 (rank 0), 16-pass launch, per geometry path."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_webgl_amd import scenes, dist as ptdist
from ray_tracer_webgl_amd.tracer import PathTracer

paths = [int(x) for x in (sys.argv[1:] or ["3", "2"])]
sc = scenes.config2(1920, 1080, 64, 16, 50)
base = None
for path in paths:
    for n in (1, 2, 4, 8):
        p = sc.params.copy()
        p.band_rows, p.band_index, p.band_count = ptdist.band_of(0, n, 8)
        pt = PathTracer(1920, 1080)
        pt.set_spheres(sc.spheres); pt.set_params(p); pt.reserve_passes(16)
        pt.set_geometry_path(path)
        ts = []
        for rep in range(4):
            pt.reset()
            t0 = time.perf_counter(); pt.render_passes(16); pt.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        t = min(ts[1:])
        if n == 1:
            base = t
        print("path %d  ranks %d: %.1f ms  (ideal %.1f, efficiency %.2f)" % (path, n, t, base / n, base / n / t), flush=True)
        pt.close()
