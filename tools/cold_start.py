"""Dev tool: where the first frame's set-up time goes (context, scene upload + structure builds, slab reservation)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t00 = time.perf_counter()
from ray_tracer_webgl_amd import abi, scenes
from ray_tracer_webgl_amd.tracer import PathTracer
sc = scenes.config2(1920, 1080, 16, 64, 50)
for rep in range(2):
    t = [time.perf_counter()]
    pt = PathTracer(1920, 1080); t.append(time.perf_counter())
    pt.set_spheres(sc.spheres); t.append(time.perf_counter())
    pt.set_params(sc.params); t.append(time.perf_counter())
    pt.reserve_passes(64); t.append(time.perf_counter())
    pt.synchronize(); t.append(time.perf_counter())
    pt.tune(8); t.append(time.perf_counter())
    pt.render_passes(64); pt.synchronize(); t.append(time.perf_counter())
    names = ["create", "set_spheres", "set_params", "reserve_passes(64)", "synchronize", "tune(8)", "first 64-pass launch"]
    print("context %d: " % rep + ", ".join("%s %.1f ms" % (n, (b - a) * 1e3) for n, a, b in zip(names, t, t[1:])), flush=True)
    pt.close()
