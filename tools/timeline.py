"""Dev tool: per-wave timeline of one launch (needs `make -C ray_tracer_webgl_amd/csrc timeline`,
run with PT_LIB=ray_tracer_webgl_amd/libptrace_timeline.so)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ray_tracer_webgl_amd import scenes
from ray_tracer_webgl_amd.tracer import PathTracer

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sc = scenes.config5(1920, 1080, 64, 4, 50) if os.environ.get('PT_SCENE') == 'config5' else scenes.config2(1920, 1080, 64, 16, 50)
pt = PathTracer(1920, 1080)
pt.set_spheres(sc.spheres); pt.set_params(sc.params); pt.reserve_passes(16)
if os.environ.get('PT_GEOM'):
    pt.set_geometry_path(int(os.environ['PT_GEOM']))
pt.render_passes(n); pt.synchronize(); pt.reset()
pt.render_passes(n); pt.synchronize()
lib = pt.lib
lib.pt_debug_timeline.restype = C.c_long
buf = np.zeros(8 * 65536, dtype=np.uint64)
got = lib.pt_debug_timeline(pt._ctx, buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size))
t = buf[:got].reshape(-1, 8).astype(np.float64)
t = t[t[:, 3] > 0]
t0 = t[:, 0].min()
us = lambda x: (x - t0) / 100.0  # 100 MHz
end = us(t[:, 3]); dry = us(np.where(t[:, 1] > 0, t[:, 1], t[:, 3])); coop = us(np.where(t[:, 2] > 0, t[:, 2], t[:, 3]))
print("waves %d  kernel span %.1f us" % (len(t), end.max()))
for name, v in (("first-exhausted-lane", dry), ("tail-mode start", coop), ("wave exit", end)):
    q = np.percentile(v, [0, 10, 50, 90, 99, 100])
    print("%-22s min %.0f p10 %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f us" % ((name,) + tuple(q)))
print("iterations/wave: total p50 %.0f max %.0f | after queue dry p50 %.0f max %.0f | tail-mode p50 %.0f max %.0f" % (
    np.median(t[:, 4]), t[:, 4].max(), np.median(t[:, 5]), t[:, 5].max(), np.median(t[:, 6]), t[:, 6].max()))
busy = (end - 0).sum() / (len(t) * end.max())
print("mean wave lifetime / kernel span = %.3f ; time in scan-mode after dry (mean) %.0f us ; in tail mode %.0f us" % (
    busy, (coop - dry).mean(), (end - coop).mean()))
pt.stats()
