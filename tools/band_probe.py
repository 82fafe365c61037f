"""Dev tool: where does the short-launch rate loss come from?  Full frame and rank 0's band of 8
with growing pass counts, plus the measuring twin's utilisation figures."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_webgl_amd import abi, dist as ptdist, scenes
from ray_tracer_webgl_amd.tracer import PathTracer

def run(n, passes, spp, chunk=None):
    sc = scenes.config2(1920, 1080, spp, passes, 50)
    p = sc.params.copy()
    p.band_rows, p.band_index, p.band_count = ptdist.band_of(0, n, 4)
    pt = PathTracer(1920, 1080)
    pt.set_spheres(sc.spheres); pt.set_params(p); pt.reserve_passes(passes); pt.set_geometry_path(abi.PT_GEOM_GRID)
    ts = []
    for rep in range(4):
        pt.reset(); t0 = time.perf_counter(); pt.render_passes(passes); pt.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    st = pt.stats()
    t = min(ts[1:]); seg = st.segments; kms = st.render_kernel_ms
    pt.reset(); pt.set_count_work(True); pt.render_passes(passes); w = list(pt.stats().work)
    print("ranks %d passes %3d x %d spp: wall %.2f ms, %.1f Mray/s | shade util %.3f walk util %.3f leaf %.3f exact %.3f steps/64seg %.3f carried %.2f"
          % (n, passes, spp, t, seg / t / 1e3, seg / max(64.0 * w[6], 1), w[1] / max(64.0 * w[0], 1), w[3] / max(64.0 * w[2], 1), w[5] / max(64.0 * w[4], 1), w[6] * 64.0 / seg, w[7] * 64.0 / seg), flush=True)
    pt.close()

for passes in (2, 4, 8, 16, 64):
    run(1, passes, 16)
for passes in (16, 32, 64, 128):
    run(8, passes, 16)
