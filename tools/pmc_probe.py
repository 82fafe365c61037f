"""Dev tool for rocprofv3 --pmc runs: a few launches of the full frame and of rank 0's band of 8
(grid walk, 64 passes of 16 spp) — compare instructions per segment and cycles."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_webgl_amd import abi, dist as ptdist, scenes
from ray_tracer_webgl_amd.tracer import PathTracer
for n, passes in ((1, 64), (8, 64), (1, 8)):
    sc = scenes.config2(1920, 1080, 16, passes, 50)
    p = sc.params.copy()
    p.band_rows, p.band_index, p.band_count = ptdist.band_of(0, n, 4)
    pt = PathTracer(1920, 1080)
    pt.set_spheres(sc.spheres); pt.set_params(p); pt.reserve_passes(passes); pt.set_geometry_path(abi.PT_GEOM_GRID)
    for rep in range(3):
        pt.reset(); pt.render_passes(passes); pt.synchronize()
    st = pt.stats()
    print("ranks %d passes %d: segments %d kernel %.3f ms" % (n, passes, st.segments, st.render_kernel_ms), flush=True)
    pt.close()
