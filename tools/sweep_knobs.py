"""Dev tool: one-variable sweeps of the scheduling knobs on config 2's timed launch (grid walk, 64 x 16 spp),
each point in a fresh child process on the PT_DEV_KNOBS build (build_ab/libptrace_knobs.so)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def child():
    from ray_tracer_webgl_amd import abi, scenes
    from ray_tracer_webgl_amd.tracer import PathTracer
    spp, n = int(os.environ.get("SW_SPP", "16")), int(os.environ.get("SW_PASSES", "64"))
    if os.environ.get("SW_CONFIG", "config2") == "default":  # State::default at the reference's size, depth 8
        sc = scenes.default_scene(1280, 702, spp, 8, n)
    elif os.environ.get("SW_CONFIG", "config2") == "config5":
        sc = scenes.config5(1920, 1080, spp, n, 50)
    else:
        sc = scenes.config2(1920, 1080, spp, n, 50)
    sc.params.time_step = abi.PT_TIME_STEP_DECORRELATED
    band = os.environ.get("SW_BAND")  # "rows:index:count": one rank's interleaved row bands (bench.py --gpus N, strong scaling)
    if band:
        sc.params.band_rows, sc.params.band_index, sc.params.band_count = [int(x) for x in band.split(":")]
    pt = PathTracer(sc.params.width, sc.params.height)
    pt.set_geometry_path(int(os.environ.get("SW_PATH", abi.PT_GEOM_GRID)))
    if os.environ.get("SW_CARRY"): pt.set_carry_lanes(int(os.environ["SW_CARRY"]))
    if os.environ.get("SW_REFILL"): pt.set_refill_min(int(os.environ["SW_REFILL"]))
    pt.set_spheres(sc.spheres); pt.set_params(sc.params); pt.reserve_passes(n)
    ms = []
    for rep in range(4):
        if band:  # as in bench.py: every measured launch follows a warm-up of another seed (the tile order it finds)
            q = sc.params.copy(); q.time = 1000.0 + rep
            pt.set_params(q); pt.render_passes(n); pt.set_params(sc.params)
        pt.reset(); pt.render_passes(n)
        if not pt.wait(float(os.environ.get("AB_DEADLINE_S", "60"))):  # watchdog: poll with a deadline, never block in the driver
            print("WATCHDOG: launch did not finish", file=sys.stderr, flush=True); os._exit(3)
        ms.append(pt.stats().render_kernel_ms)
    print("%.3f" % min(ms[1:]), flush=True)
    pt.close()

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    child()
else:
    lib = os.environ.get("PT_KNOBS_LIB") or os.path.join(ROOT, "build_ab", "libptrace_knobs.so")  # make -C ray_tracer_webgl_amd/csrc variant NAME=knobs DEFS=-DPT_DEV_KNOBS
    points = [("base", {})] + [("carry %d" % v, {"SW_CARRY": str(v)}) for v in (6, 8, 10, 16, 20)] + \
             [("refill_min %d" % v, {"SW_REFILL": str(v)}) for v in (1, 2, 6, 8)] + \
             [("block %d" % v, {"PT_BVH_BLOCK": str(v)}) for v in (256, 1024)] + \
             [("chunk %d" % v, {"PT_QUEUE_CHUNK": str(v)}) for v in (128, 256, 1024, 2048)] + \
             [("base again", {})]
    if len(sys.argv) > 1:  # e.g. sweep_knobs.py SW_CONFIG=config5,SW_SPP=16,SW_PASSES=16 PT_QUEUE_CHUNK=32,64,128
        fixed = dict(kv.split("=") for kv in sys.argv[1].split(","))
        var, vals = sys.argv[2].split("=")
        points = [("%s %s" % (var, v), dict(fixed, **{var: v})) for v in vals.split(",")]
    for name, env in points:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, PT_LIB=lib, **env),
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        print("%-16s %s" % (name, out.stdout.strip() or ("FAILED " + out.stderr[-300:])), flush=True)
        if out.returncode == 3:  # the watchdog fired: no further GPU work behind a kernel that may never end
            raise SystemExit(3)
