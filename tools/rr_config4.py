"""Dev tool: BASELINE config 4 (1024x1024, 16 passes x 64 spp per launch) with and without Russian
roulette: kernel time, segments, and the frame means (the image statistics are in tests/test_gpu_roulette.py)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_webgl_amd import abi, scenes
from ray_tracer_webgl_amd.tracer import PathTracer

sc = scenes.config4(1024, 1024, 64, 16, 50)
sc.params.time_step = abi.PT_TIME_STEP_DECORRELATED
for rr in (0, 3, 5, 8):
    pt = PathTracer(1024, 1024)
    pt.set_russian_roulette(rr)
    pt.set_spheres(sc.spheres); pt.set_params(sc.params); pt.reserve_passes(16); pt.tune(4)
    ms = []
    for rep in range(3):
        pt.reset(); pt.render_passes(16); st = pt.stats(); ms.append(st.render_kernel_ms)
    a = pt.accum()
    print("roulette after %d bounces: kernel %.1f ms per 16 x 64 spp (%s), %.3e segments (%.1f per path), %.0f Mray/s, frame mean rgb %s  -> 8192 spp in %.2f s"
          % (rr, min(ms[1:]), abi.GEOM_NAMES[st.geometry_path], st.segments, st.segments / (1024 * 1024 * 1024.0), st.segments / min(ms[1:]) / 1e3,
             np.round(a[..., :3].mean((0, 1)) / a[0, 0, 3], 5), min(ms[1:]) * 8 / 1e3), flush=True)
    pt.close()
