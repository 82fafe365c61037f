"""Dev tool: one 16-pass launch of a config per geometry path and carry setting, kernel time from
the library's HIP events (third of three launches), then the measuring twin's work counters.

    python tools/ab_paths.py [config2|config5|...] [passes] [spp]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_webgl_amd import abi, scenes  # noqa: E402
from ray_tracer_webgl_amd.tracer import PathTracer  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "config2"
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 16
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 64
variants = [(abi.PT_GEOM_BVH, 0), (abi.PT_GEOM_BVH, 8), (abi.PT_GEOM_GRID, 0), (abi.PT_GEOM_GRID, 4), (abi.PT_GEOM_GRID, 8),
            (abi.PT_GEOM_GRID, 16)]
if os.environ.get("AB_VARIANTS"):
    variants = [tuple(int(x) for x in v.split(":")) for v in os.environ["AB_VARIANTS"].split(",")]
sc = scenes.CONFIGS[name]()
sc.params.samples_per_pixel = spp
p = sc.params
for path, carry in variants:
    pt = PathTracer(p.width, p.height)
    pt.set_geometry_path(path)
    pt.set_carry_lanes(carry)
    if os.environ.get("AB_REFILL_MIN"):
        pt.set_refill_min(int(os.environ["AB_REFILL_MIN"]))
    pt.set_spheres(sc.spheres)
    pt.set_params(p)
    pt.reserve_passes(passes)
    ms = []
    for rep in range(3):
        pt.reset()
        pt.render_passes(passes)
        st = pt.stats()
        ms.append(st.render_kernel_ms)
    seg = st.segments
    pt.reset()
    pt.set_count_work(True)
    pt.render_passes(passes)
    w = list(pt.stats().work)
    per = 64.0 / max(seg, 1)
    print("%s path %-6s carry %2d: %8.2f ms (%.2f, %.2f)  %.0f Mray/s | per 64 seg: walk %.2f it x %.1f lanes, leaf %.2f x %.1f, "
          "exact %.2f x %.1f, steps %.3f, carried %.2f"
          % (name, abi.GEOM_NAMES[st.geometry_path], carry, ms[2], ms[0], ms[1], seg / ms[2] / 1e3,
             w[0] * per, w[1] / max(w[0], 1), w[2] * per, w[3] / max(w[2], 1), w[4] * per, w[5] / max(w[4], 1),
             w[6] * per, w[7] * per), flush=True)
    pt.close()
