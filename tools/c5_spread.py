"""Dev tool: config 5 (10 001 spheres), grid walk: a series of identical launches on one context, every
launch's kernel time printed (the first runs on the identity tile order, the later ones on the order the
previous launch's costs produced) — where does the run-to-run spread of this config come from?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_webgl_amd import abi, scenes
from ray_tracer_webgl_amd.tracer import PathTracer

for spp, passes in ((64, 4), (16, 16)):
    sc = scenes.config5(1920, 1080, spp, passes, 50)
    sc.params.time_step = abi.PT_TIME_STEP_DECORRELATED
    pt = PathTracer(1920, 1080)
    pt.set_geometry_path(abi.PT_GEOM_GRID)
    pt.set_spheres(sc.spheres); pt.set_params(sc.params); pt.reserve_passes(passes)
    out = []
    for rep in range(7):
        pt.reset(); pt.render_passes(passes); out.append(pt.stats().render_kernel_ms)
    print("config5 %2d x %2d spp: %s ms" % (passes, spp, " ".join("%.1f" % x for x in out)), flush=True)
    pt.close()
