"""Dev tool: interleaved A/B timing of env-var-selected kernel behaviours in ONE process on ONE
GPU (separate runs differ by device and clock state).  usage: ab_env.py VAR val1,val2,... [passes...]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_webgl_amd import scenes
from ray_tracer_webgl_amd.tracer import PathTracer

var = sys.argv[1]
vals = sys.argv[2].split(",")
plist = [int(x) for x in sys.argv[3:]] or [1, 2, 16]
sc = scenes.config2(1920, 1080, 64, 16, 50)
pt = PathTracer(1920, 1080)
pt.set_spheres(sc.spheres)
pt.set_params(sc.params)
pt.reserve_passes(16)
if os.environ.get('PT_GEOM'):
    pt.set_geometry_path(int(os.environ['PT_GEOM']))
for n in plist:
    res = {v: [] for v in vals}
    for rep in range(3):
        for v in vals:
            if v == "unset":
                os.environ.pop(var, None)
            else:
                os.environ[var] = v
            pt.reset()
            pt.render_passes(n); pt.synchronize()  # sets the tile order for this shape
            pt.reset()
            t0 = time.perf_counter()
            pt.render_passes(n)
            pt.synchronize()
            res[v].append((time.perf_counter() - t0) * 1e3)
    for v in vals:
        r = sorted(res[v])
        print("%s=%-6s passes %2d: median %.2f ms  min %.2f  max %.2f" % (var, v, n, r[len(r) // 2], r[0], r[-1]), flush=True)
