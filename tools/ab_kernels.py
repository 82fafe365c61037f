"""Dev tool: A/B of two builds of libptrace.so on ONE device (devices differ by up to ~10 %, and so do
the boxes of two gpurun calls — never compare across calls).

    python tools/ab_kernels.py libA.so libB.so [rounds]      driver: alternates A B A B ..., one child process per run
    PT_LIB=lib.so python tools/ab_kernels.py --child          one run: forced-path launches of the BASELINE configs

Prints per (config, path) the kernel time of the library's HIP events (min over the repetitions).

WATCHDOG: every launch of a child is waited for by polling an event with a deadline (pt_debug_wait,
include/ptrace_dev.h; AB_DEADLINE_S, default 60 s per launch); a launch that does not finish makes the child
report the case and exit with status 3 at once, and the driver stops — no further GPU work is started behind a
kernel that may never end.  Every build runs in a fresh child process; nothing is ever re-executed in place."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def watchdog(pt, what):
    """Wait for the launch by polling (never a blocking synchronise); on timeout report and leave at once."""
    deadline = float(os.environ.get("AB_DEADLINE_S", "60"))
    try:
        ok = pt.wait(deadline)
    except AttributeError:  # a build from before the watchdog existed (a known-good baseline)
        return
    if not ok:
        print("WATCHDOG: %s did not finish within %.0f s (%s)" % (what, deadline, os.environ.get("PT_LIB", "default library")),
              file=sys.stderr, flush=True)
        os._exit(3)


def child():
    from ray_tracer_webgl_amd import abi, scenes
    from ray_tracer_webgl_amd.tracer import PathTracer

    sel = os.environ.get("AB_CASES", "").split(",") if os.environ.get("AB_CASES") else None
    cases = [("c2grid", scenes.config2(1920, 1080, 16, 64, 50), 64, abi.PT_GEOM_GRID, 3),
             ("c2bvh", scenes.config2(1920, 1080, 16, 64, 50), 64, abi.PT_GEOM_BVH, 2),
             ("c2scalar", scenes.config2(1920, 1080, 16, 16, 50), 16, abi.PT_GEOM_SCALAR, 2),
             ("c2band8", scenes.config2(1920, 1080, 16, 8, 50), 8, abi.PT_GEOM_GRID, 3),
             ("c4scalar", scenes.config4(1024, 1024, 64, 8, 50), 8, abi.PT_GEOM_SCALAR, 2),
             ("c4lds", scenes.config4(1024, 1024, 64, 8, 50), 8, abi.PT_GEOM_LDS, 2),
             ("c5grid", scenes.config5(1920, 1080, 64, 4, 50), 4, abi.PT_GEOM_GRID, 3),
             ("c5grid16", scenes.config5(1920, 1080, 16, 16, 50), 16, abi.PT_GEOM_GRID, 3),
             ("c5grid32", scenes.config5(1920, 1080, 32, 8, 50), 8, abi.PT_GEOM_GRID, 3),  # (holds a pixel whose paths leave the real numbers)
             ("default", scenes.default_scene(1280, 702, 25, 8, 16), 16, abi.PT_GEOM_SCALAR, 3),
             ("default1", scenes.default_scene(1280, 702, 1, 8, 1), 1, abi.PT_GEOM_SCALAR, 5),
             ("c4small", scenes.config4(1024, 1024, 64, 8, 50), 8, abi.PT_GEOM_SMALL, 2),
             ("defsmall", scenes.default_scene(1280, 702, 25, 8, 16), 16, abi.PT_GEOM_SMALL, 3),
             ("def1small", scenes.default_scene(1280, 702, 1, 8, 1), 1, abi.PT_GEOM_SMALL, 5)]
    # one rank of eight as bench.py runs it in strong scaling: its 4-row bands, ONE 64-pass launch, the tile order fed by a
    # warm-up launch of ANOTHER seed (u_time 1000), rank 0 and rank 5
    cases += [("c2rank8_%d" % r, scenes.config2(1920, 1080, 16, 64, 50), 64, abi.PT_GEOM_GRID, 3, (4, r, 8)) for r in (0, 5)]
    for case in cases:
        name, sc, n, path, reps = case[:5]
        band = case[5] if len(case) > 5 else None
        if sel and name not in sel:
            continue
        sc.params.time_step = abi.PT_TIME_STEP_DECORRELATED
        if band:
            sc.params.band_rows, sc.params.band_index, sc.params.band_count = band
        pt = PathTracer(sc.params.width, sc.params.height)
        try:
            pt.set_geometry_path(path)
        except Exception:  # a build that does not know this path
            pt.close()
            continue
        pt.set_spheres(sc.spheres)
        pt.set_params(sc.params)
        pt.reserve_passes(n)
        ms = []
        for rep in range(reps + 1):  # the first launch settles the tile order
            if band:  # every measured launch is preceded by a warm-up of another seed, as in bench.py
                q = sc.params.copy()
                q.time = 1000.0 + rep
                pt.set_params(q)
                pt.render_passes(n)
                pt.set_params(sc.params)
            pt.reset()
            pt.render_passes(n)
            watchdog(pt, name)
            ms.append(pt.stats().render_kernel_ms)
        import hashlib
        digest = hashlib.sha256(pt.accum().tobytes()).hexdigest()[:16]  # the builds must agree on every bit of the frame
        print("%s %.4f %d %s" % (name, min(ms[1:]), pt.stats().segments, digest), flush=True)
        pt.close()


def main():
    libs = [os.path.abspath(x) for x in sys.argv[1:3]]
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    res = {}
    for r in range(rounds):
        for lib in libs:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, PT_LIB=lib),
                                 stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
            if out.returncode != 0:
                print("child failed for", lib, "(status %d%s)" % (out.returncode, ": WATCHDOG, stopping" if out.returncode == 3 else ""),
                      out.stderr[-1500:], flush=True)
                return out.returncode
            for line in out.stdout.splitlines():
                k, ms, seg, dig = line.split()
                res.setdefault(k, {}).setdefault(lib, []).append((float(ms), (int(seg), dig)))
    print("%-10s %s" % ("case", "  ".join("%26s" % os.path.basename(x) for x in libs)) + "   B/A")
    for k, v in res.items():
        if any(lib not in v for lib in libs):
            only = [lib for lib in libs if lib in v][0]
            print("%-10s only in %s: %s" % (k, os.path.basename(only), " ".join("%.3f" % x[0] for x in v[only])))
            continue
        a = min(x[0] for x in v[libs[0]])
        b = min(x[0] for x in v[libs[1]])
        segs = {x[1] for lib in libs for x in v[lib]}
        print("%-10s %26s  %26s   %.4f%s" % (k, " ".join("%.3f" % x[0] for x in v[libs[0]]), " ".join("%.3f" % x[0] for x in v[libs[1]]),
                                             b / a, "" if len(segs) == 1 else "  !! segment counts / frame digests differ: %s" % segs))
    return 0


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child()
    else:
        raise SystemExit(main())
