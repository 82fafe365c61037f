import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_webgl_amd import abi, scenes
from ray_tracer_webgl_amd.tracer import PathTracer
for spp, n in ((64, 4), (32, 8), (16, 16)):
    for force in (True, False):
        for deco in (True, False):
            sc = scenes.config5(1920, 1080, spp, n, 50)
            if deco: sc.params.time_step = abi.PT_TIME_STEP_DECORRELATED
            pt = PathTracer(1920, 1080)
            if force: pt.set_geometry_path(abi.PT_GEOM_GRID)
            pt.set_spheres(sc.spheres); pt.set_params(sc.params); pt.reserve_passes(n)
            if not force: pt.tune(n)
            ms = []
            for rep in range(5):
                pt.reset(); k0 = pt.stats().render_kernel_ms; pt.render_passes(n); ms.append(pt.stats().render_kernel_ms)
            print(spp, n, "forced" if force else "auto", "deco" if deco else "t=1", abi.GEOM_NAMES.get(pt.stats().geometry_path), " ".join("%.1f" % m for m in ms), flush=True)
            pt.close()
