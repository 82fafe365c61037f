"""Dev tool for rocprofv3 --pmc passes over the BASELINE configs other than 2: a fixed, deterministic
sequence of forced-path launches (PT_GEOM_AUTO's choice may differ between profiler passes, which
would break the per-dispatch join of profiles/pmc_dispatches.py).

    config 4 (9 spheres, 1024x1024, 16 x 64 spp)      : small-list kernel x2, scalar list walk x2
    State::default (9 spheres, 1280x702, 16 x 25 spp)  : small-list kernel x2, scalar list walk x2
    State::default at the reference's operating point (1 spp, depth 8, one pass per launch) : small x3
    config 5 (10 001 spheres, 1920x1080, 4 x 64 spp)   : grid x3;  the same 256 spp as 16 x 16 spp : grid x3
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_webgl_amd import abi, scenes
from ray_tracer_webgl_amd.tracer import PathTracer

NAMES = dict(abi.GEOM_NAMES)
want = sys.argv[1:]
cases = [("config4", scenes.config4(1024, 1024, 64, 16, 50), 16, [(abi.PT_GEOM_SMALL, 2), (abi.PT_GEOM_SCALAR, 2)]),
         ("default", scenes.default_scene(1280, 702, 25, 8, 16), 16, [(abi.PT_GEOM_SMALL, 2), (abi.PT_GEOM_SCALAR, 2)]),
         ("default1spp", scenes.default_scene(1280, 702, 1, 8, 1), 1, [(abi.PT_GEOM_SMALL, 3)]),
         ("config5", scenes.config5(1920, 1080, 64, 4, 50), 4, [(abi.PT_GEOM_GRID, 3)]),
         ("config5_16spp", scenes.config5(1920, 1080, 16, 16, 50), 16, [(abi.PT_GEOM_GRID, 3)])]
for name, sc, n, plan in cases:
    if want and name not in want:
        continue
    pt = PathTracer(sc.params.width, sc.params.height)
    pt.set_spheres(sc.spheres)
    pt.set_params(sc.params)
    pt.reserve_passes(n)
    for path, reps in plan:
        pt.set_geometry_path(path)
        for rep in range(reps):
            pt.reset()
            pt.render_passes(n)
            pt.synchronize()
            st = pt.stats()
            print("%-12s %-7s launch %d: %d passes x %d spp, %d segments, kernel %.3f ms" % (
                name, NAMES.get(st.geometry_path), rep, n, sc.params.samples_per_pixel, st.segments, st.render_kernel_ms), flush=True)
    pt.close()
