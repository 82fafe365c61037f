"""Dev tool for rocprofv3 --pmc passes: three launches of config 5's timed shape (grid walk, 16 passes of 16 spp, decorrelated
pass times), nothing else — is the leaf round bound by the LATENCY of its L2 gathers or by the THROUGHPUT of the texture
addresser / vector L1 (TA_TA_BUSY, TCP_GATE_EN*, TCP_*_STALL_CYCLES against GRBM_GUI_ACTIVE)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_webgl_amd import abi, scenes
from ray_tracer_webgl_amd.tracer import PathTracer
sc = scenes.config5(1920, 1080, 16, 16, 50)
sc.params.time_step = abi.PT_TIME_STEP_DECORRELATED
pt = PathTracer(1920, 1080)
pt.set_spheres(sc.spheres); pt.set_params(sc.params); pt.reserve_passes(16); pt.set_geometry_path(abi.PT_GEOM_GRID)
for rep in range(3):
    pt.reset(); pt.render_passes(16); pt.synchronize()
    st = pt.stats()
    print("config5 launch %d: %d segments, kernel %.3f ms" % (rep, st.segments, st.render_kernel_ms), flush=True)
pt.close()
