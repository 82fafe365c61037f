"""Dev tool: many more seeds of tests/test_gpu_fuzz.py's generator than the suite runs.

    python tools/long_fuzz.py <first seed> <last seed> [bvh | grid | small]     (no third argument: a random path per scene)
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from oracle import oracle
from ray_tracer_webgl_amd.tracer import render_scene
from test_gpu_fuzz import random_scene

lo, hi = int(sys.argv[1]), int(sys.argv[2])
bvh = len(sys.argv) > 3 and sys.argv[3] in ("bvh", "grid")  # force a walk kernel on scenes that get a structure
grid = len(sys.argv) > 3 and sys.argv[3] == "grid"
force_small = len(sys.argv) > 3 and sys.argv[3] == "small"  # force the small-list kernels on lists of 1 .. 16 spheres
used = 0
bad = 0
refitted = 0
classes = {}
for seed in range(lo, hi):
    rng = np.random.default_rng(seed)
    n = int(rng.choice([16, 17, 33, 40, 130, 400, 1500, 4000] if bvh else [1, 2, 5, 9, 17, 40, 130, 400]))
    if force_small:  # every list length the small-list kernels exist for (one build per length modulo four)
        n = int(rng.integers(1, 17))
    width, height = int(rng.integers(9, 200)), int(rng.integers(5, 120))
    if n >= 400:
        width, height = min(width, 64), min(height, 40)
    spp, depth, passes = int(rng.integers(1, 9)), int(rng.choice([1, 3, 8, 50])), int(rng.integers(1, 5))
    sc = random_scene(rng, n, width, height, spp, depth, passes)
    path = int(rng.choice([0, 1, 2, 3, 4, 5, 5]))  # auto, LDS, scalar, hierarchy, grid, small (falls back beyond 16 spheres)
    if force_small:
        path = 5
    if bvh:
        path = 4 if grid else 3
        if grid and seed % 2 == 0:  # mostly small spheres: what a grid is for (the rest: wild radii, many always-tested)
            small = rng.random(n) < 0.9
            sc.spheres["radius"][small] = (np.sign(sc.spheres["radius"][small]) * rng.uniform(0.05, 0.4, small.sum())).astype(np.float32)
        if grid and seed % 5 == 0:  # a flat field: one layer of cells
            sc.spheres["center"][:, 1] = np.float32(0.3)
        if seed % 3 != 1:  # spread the field out so that the tree has something to cull
            sc.spheres["center"] *= np.float32(rng.choice([2.0, 4.0, 15.0, 100.0]))
        if seed % 7 == 0:  # and far from the origin: the margin works in the frame of the scene
            sc.spheres["center"] += np.float32(rng.choice([50.0, 3000.0]))
            sc.params.camera_origin[0] += float(sc.spheres["center"][0][0]) * 0  # camera stays: distant views
    from ray_tracer_webgl_amd.tracer import PathTracer
    if grid and seed % 2 == 1:
        # ... with the grid fitted to the scene's camera first (pt_tune: another margin class, another set of entries) — every
        # fourth scene with the camera PLACED for one of the seven classes in turn and the class taken unmeasured (round 6)
        if seed % 4 == 3:
            import ctypes as C
            import math
            from ray_tracer_webgl_amd import abi, scenes
            from test_grid import build as grid_build
            rc_g, g = grid_build(sc.spheres)
            if rc_g == 0:
                want = [2.5, 3.0, 4.0, 5.5, 8.0, 12.0, 16.0][(seed // 4) % 7]
                c0, s0 = g["c0"].astype(np.float64), float(g["s0"])
                dvec = rng.normal(size=3)
                dvec /= np.linalg.norm(dvec)
                la = abi.PtLookAtIn()
                la.width, la.height = sc.params.width, sc.params.height
                la.look_from = abi.d3(*(c0 + dvec * (want / 1.01 - 1.0) * 0.97 * s0))
                la.look_at = abi.d3(*(c0 + rng.uniform(-0.2, 0.2, 3) * s0))
                la.vup = abi.d3(0, 1, 0)
                la.vfov_radians = math.radians(rng.uniform(15, 60))
                la.focus_distance = max(want - 1.0, 0.5) * s0
                la.aperture = float(rng.choice([0.0, 0.02 * s0]))
                assert scenes._lib().pt_camera_look_at(C.byref(la), C.byref(sc.params)) == 0
        t = PathTracer(sc.params.width, sc.params.height)
        t.set_geometry_path(path)
        if seed % 4 == 3:
            t.set_grid_fit(True)
        t.set_spheres(sc.spheres)
        t.set_params(sc.params)
        t.reserve_passes(passes)
        before = t.stats().grid_entries
        t.tune(1)
        classes[round(float(t.stats().grid_near_factor), 1)] = classes.get(round(float(t.stats().grid_near_factor), 1), 0) + 1
        refitted += int(t.stats().grid_entries != before)
        t.set_params(sc.params)
        t.render_passes(passes)
        got = t.accum()
    else:
        t, got = render_scene(sc, passes_per_launch=int(rng.integers(1, passes + 1)), geometry_path=path)
    used += int(t.stats().geometry_path == path or path == 0)
    ref, seg = oracle.render(sc.spheres, sc.params, passes)
    ok = np.array_equal(got.view(np.uint32), ref.view(np.uint32)) and t.stats().segments == seg
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, n, width, height, spp, depth, passes, path, flush=True)
    t.close()
    if seed % 50 == 0:
        print("seed", seed, "ok so far, bad =", bad, flush=True)
print("done", lo, hi, "bad =", bad, "forced path really used:", used, "grids refitted by pt_tune:", refitted, "margin classes walked after pt_tune:", dict(sorted(classes.items())))
