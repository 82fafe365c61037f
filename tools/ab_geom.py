"""Dev tool: LDS walk vs scalar-load walk vs hierarchy walk, interleaved in one process."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_webgl_amd import scenes
from ray_tracer_webgl_amd.tracer import PathTracer

cases = {"config2": (scenes.config2(1920, 1080, 64, 16, 50), 16), "config5": (scenes.config5(1920, 1080, 64, 2, 50), 2),
         "config4": (scenes.config4(1024, 1024, 64, 16, 50), 16), "default": (scenes.default_scene(1280, 702, 25, 8, 16), 16),
         "config5_2k": (scenes.config5(1920, 1080, 64, 4, 50, n=2000), 4)}
for k in (100, 250, 500, 700, 1000, 1400):
    cases["field_%d" % k] = (scenes.config5(1920, 1080, 64, 4, 50, n=k), 4)
for name in sys.argv[1:] or list(cases):
    sc, n = cases[name]
    pt = PathTracer(sc.params.width, sc.params.height)
    pt.set_spheres(sc.spheres); pt.set_params(sc.params); pt.reserve_passes(n)
    modes = {"lds": 1, "scalar": 2, "bvh": 3}
    if len(sc.spheres) > 10232:
        del modes["lds"]
    res = {m: [] for m in modes}
    for rep in range(3):
        for mode in modes:
            pt.set_geometry_path(modes[mode])
            pt.reset(); pt.render_passes(n); pt.synchronize(); pt.reset()
            t0 = time.perf_counter(); pt.render_passes(n); pt.synchronize()
            res[mode].append((time.perf_counter() - t0) * 1e3)
    segs = pt.stats().segments
    st = pt.stats()
    med = {m: sorted(v)[1] for m, v in res.items()}
    print("%-10s %5d spheres (tree: %d nodes, %d slots, %d outliers, depth %d): " % (
        name, len(sc.spheres), st.bvh_nodes, st.bvh_slots, st.bvh_outliers, st.bvh_depth) +
        "  ".join("%s %.1f ms (%.2f Gray/s)" % (m, med[m], segs / med[m] / 1e6) for m in med), flush=True)
    pt.close()
