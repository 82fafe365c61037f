"""Dev tool: does the TILE ORDER matter to a statically dealt launch of short items (the reference's frame groups: 16 passes of 1 spp)?
Such launches keep the order they find (no cost feedback of their own).  Times 16 x 1 spp on State::default at 1280x702 (a) with
the identity order of a fresh context, (b) after two 8-spp launches have left a cost-sorted order behind."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_webgl_amd import abi, scenes
from ray_tracer_webgl_amd.tracer import PathTracer

def timed(pt, p, n, reps=5):
    pt.set_params(p)
    ms = []
    for _ in range(reps):
        pt.reset(); pt.render_passes(n)
        assert pt.wait(60.0)
        ms.append(pt.stats().render_kernel_ms)
    return min(ms[1:])

for scene_name in ("default", "config2"):
    sc = scenes.default_scene(1280, 702, 1, 8, 16) if scene_name == "default" else scenes.config2(1920, 1080, 1, 16, 50)
    sc.params.time_step = abi.PT_TIME_STEP_DECORRELATED
    pt = PathTracer(sc.params.width, sc.params.height)
    pt.set_spheres(sc.spheres)
    pt.reserve_passes(16)
    pt.set_geometry_path(abi.PT_GEOM_SMALL if scene_name == "default" else abi.PT_GEOM_GRID)
    p1 = sc.params.copy()
    a = timed(pt, p1, 16)
    p8 = sc.params.copy(); p8.samples_per_pixel = 8
    pt.set_params(p8)
    for _ in range(3):
        pt.reset(); pt.render_passes(2); pt.synchronize()
    b = timed(pt, p1, 16)
    p1b = p1.copy(); p1b.time = 77.0
    c = timed(pt, p1b, 16)
    print("%s 16 x 1 spp: identity tile order %.4f ms; cost-sorted order (left by 8-spp launches) %.4f ms, other seeds %.4f ms" % (scene_name, a, b, c), flush=True)
    pt.close()
