"""Dev tool: one launch of each BASELINE config at reduced pass counts; prints Mray/s."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_webgl_amd import scenes
from ray_tracer_webgl_amd.tracer import PathTracer

cases = [("config2", scenes.config2(1920, 1080, 64, 16, 50), 16),
         ("config4", scenes.config4(1024, 1024, 64, 16, 50), 16),
         ("config5", scenes.config5(1920, 1080, 64, 4, 50), 4),
         ("config3", scenes.config3(3840, 2160, 64, 8, 50), 8),
         ("default", scenes.default_scene(1280, 702, 25, 8, 16), 16)]
# one launch per config after pt_tune (the timed launch is the LAST trace-kernel dispatch of each
# config in a rocprofv3 kernel trace; the ones before it are pt_tune's trials of every usable path)
for name, sc, n in cases:
    if len(sys.argv) > 1 and name not in sys.argv[1:]:
        continue
    pt = PathTracer(sc.params.width, sc.params.height)
    pt.set_spheres(sc.spheres)
    pt.set_params(sc.params)
    pt.reserve_passes(n)
    pt.tune(n)
    t0 = time.perf_counter()
    pt.render_passes(n)
    pt.synchronize()
    dt = time.perf_counter() - t0
    st = pt.stats()
    tests = st.segments * len(sc.spheres)
    from ray_tracer_webgl_amd import abi
    print("%-8s %5d spheres %dx%d %d passes x %d spp [%s]: %.1f ms, %.1f Mray/s, %.2f Ttests/s, %.1f TFLOP/s(20/test)" % (
        name, len(sc.spheres), sc.params.width, sc.params.height, n, sc.params.samples_per_pixel,
        abi.GEOM_NAMES.get(st.geometry_path), dt * 1e3,
        st.segments / dt / 1e6, tests / dt / 1e12, 20 * tests / dt / 1e12), flush=True)
    pt.close()
