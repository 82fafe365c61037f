"""Command-line renderer: one of the BASELINE scenes -> PNG (the reference's "Save Image" path,
src/dom.rs:126-143, without a browser).

    python -m ray_tracer_webgl_amd.render --config config2 --width 1920 --height 1080 --out cover.png
"""
import argparse
import time

from . import image_io, scenes
from . import abi
from .tracer import render_scene


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--config", default="config2", choices=sorted(scenes.CONFIGS))
    ap.add_argument("--width", type=int)
    ap.add_argument("--height", type=int)
    ap.add_argument("--spp-per-pass", type=int)
    ap.add_argument("--passes", type=int)
    ap.add_argument("--max-depth", type=int)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--geometry", default="auto", choices=["auto", "lds", "scalar", "bvh", "grid", "small"],
                    help="how the kernel looks at the sphere list (same image bits on every path; auto: the library measures "
                         "the usable ones — the grid walk on scenes of hundreds of spheres, the small-list kernels up to 16)")
    ap.add_argument("--out", default="render.png")
    ap.add_argument("--checkpoint", help="also save the fp32 accumulation buffer (.npz)")
    args = ap.parse_args(argv)

    make = scenes.CONFIGS[args.config]
    sc = make()
    if args.width and args.height:
        sc = make(args.width, args.height)
    p = sc.params
    if args.spp_per_pass:
        p.samples_per_pixel = args.spp_per_pass
    if args.max_depth:
        p.max_depth = args.max_depth
    if args.passes:
        sc.n_passes = args.passes
    t0 = time.perf_counter()
    geom = {"auto": abi.PT_GEOM_AUTO, "lds": abi.PT_GEOM_LDS, "scalar": abi.PT_GEOM_SCALAR, "bvh": abi.PT_GEOM_BVH,
            "grid": abi.PT_GEOM_GRID, "small": abi.PT_GEOM_SMALL}[args.geometry]
    per = min(sc.n_passes, 16)
    # (pt_tune first, as bench.py does: the grid fitted to this camera, PT_GEOM_AUTO settled before the frame's launches)
    pt, acc = render_scene(sc, device=args.device, passes_per_launch=per, geometry_path=geom, tune=min(per, 8))
    dt = time.perf_counter() - t0
    st = pt.stats()
    frame = pt.resolve(gamma=True)
    image_io.write_png(args.out, frame)
    if args.checkpoint:
        image_io.save_accum(args.checkpoint, acc, st.total_spp)
    print("%s: %dx%d, %d spheres, %d spp, depth %d: %.2f s, %.0f Mray/s -> %s" % (
        sc.name, p.width, p.height, len(sc.spheres), st.total_spp, p.max_depth, dt, st.segments / dt / 1e6, args.out))
    pt.close()


if __name__ == "__main__":
    main()
