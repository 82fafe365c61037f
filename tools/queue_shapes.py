"""Dev tool: kernel time of pt_render_passes over launch shapes that stress the work queue — few samples per item, from a
few to hundreds of items per resident lane (State::default at the reference's size; the cover scene at 1920x1080).

    PT_LIB=build.so python tools/queue_shapes.py [scene:spp:passes ...]      (scene = default | config4 | config2; default: a fixed table)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_webgl_amd import abi, scenes  # noqa: E402
from ray_tracer_webgl_amd.tracer import PathTracer  # noqa: E402

shapes = [("default", spp, n) for spp in (1, 2, 4, 8, 25) for n in (4, 16, 64)] + [("config2", spp, n) for spp in (1, 2, 4, 16) for n in (8, 16, 64)]
if len(sys.argv) > 1:
    shapes = [(a.split(":")[0], int(a.split(":")[1]), int(a.split(":")[2])) for a in sys.argv[1:]]
for name, spp, n in shapes:
    sc = (scenes.default_scene(1280, 702, spp, 8, n) if name == "default" else
          scenes.config4(1024, 1024, spp, n, 50) if name == "config4" else scenes.config2(1920, 1080, spp, n, 50))
    sc.params.time_step = abi.PT_TIME_STEP_DECORRELATED
    pt = PathTracer(sc.params.width, sc.params.height)
    pt.set_spheres(sc.spheres)
    pt.set_params(sc.params)
    pt.reserve_passes(n)
    pt.set_geometry_path(abi.PT_GEOM_SMALL if name in ("default", "config4") else abi.PT_GEOM_GRID)  # (no autotuning: the same kernel in every build)
    ms = []
    for rep in range(4):
        pt.reset()
        pt.render_passes(n)
        if not pt.wait(60.0):
            print("WATCHDOG", name, spp, n, file=sys.stderr, flush=True)
            os._exit(3)
        ms.append(pt.stats().render_kernel_ms)
    print("%-8s %2d spp x %2d passes: %8.3f ms  (%d segments)" % (name, spp, n, min(ms[1:]), pt.stats().segments), flush=True)
    pt.close()
