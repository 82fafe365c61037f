"""Dev tool: time pt_render_passes(n) for several n on config 2 (tail / load-balance study)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ray_tracer_webgl_amd import scenes
from ray_tracer_webgl_amd.tracer import PathTracer

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sc = scenes.config2(1920, 1080, spp, 16, 50)
pt = PathTracer(1920, 1080)
pt.set_spheres(sc.spheres)
pt.set_params(sc.params)
pt.reserve_passes(16)
pt.render_passes(1); pt.synchronize(); pt.reset()
for n in (1, 2, 4, 8, 16):
    pt.reset()
    t0 = time.perf_counter()
    pt.render_passes(n)
    pt.synchronize()
    dt = time.perf_counter() - t0
    st = pt.stats()
    print("spp/pass %d passes/launch %2d: %.2f ms total, %.2f ms/pass, %.1f Mray/s, kernel %.2f ms" % (
        spp, n, dt * 1e3, dt * 1e3 / n, st.segments / dt / 1e6, st.render_kernel_ms))
