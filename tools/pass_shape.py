"""Dev tool: the same samples per pixel as passes of 64 / 32 / 16 / 8 spp in one launch (finer work
items shorten a launch's drain; more items cost queue traffic).  Usage: pass_shape.py config5 [config4 ...]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_webgl_amd import scenes
from ray_tracer_webgl_amd.tracer import PathTracer

makers = {"config2": (lambda spp, n: scenes.config2(1920, 1080, spp, n, 50), 1024),
          "config4": (lambda spp, n: scenes.config4(1024, 1024, spp, n, 50), 1024),
          "config5": (lambda spp, n: scenes.config5(1920, 1080, spp, n, 50), 256)}
for name in sys.argv[1:]:
    make, total = makers[name]
    for spp in (64, 32, 16, 8):
        n = total // spp
        sc = make(spp, n)
        pt = PathTracer(sc.params.width, sc.params.height)
        pt.set_spheres(sc.spheres)
        pt.set_params(sc.params)
        pt.reserve_passes(n)
        pt.tune(n)
        best = 1e9
        for _ in range(3):
            pt.reset()
            t0 = time.perf_counter()
            pt.render_passes(n)
            pt.synchronize()
            best = min(best, time.perf_counter() - t0)
        st = pt.stats()
        print("%-8s %4d passes x %2d spp: %.1f ms  (%.3e segments per launch)" % (name, n, spp, best * 1e3, st.segments), flush=True)
        pt.close()
