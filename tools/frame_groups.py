"""Dev tool: the reference's frame loop (State::default, 1280x702, depth 8) at several samples per frame — ms per frame
when a series is replayed from hipGraphs in groups (pt_render_frames) and when single frames are issued from the host.

    PT_LIB=build_ab/libptrace_knobs.so PT_GROUP_STATIC=0 python tools/frame_groups.py [spp ...]

With the PT_DEV_KNOBS build: PT_GROUP_STATIC=0 sends a group's items through the shared queue whenever the launch's own
rule would (prepare_launch), =1 (the default) deals every group statically; PT_FEWER_X10_1 / _2 = items per lane (x 10)
a statically dealt launch of 1- / 2-sample items is sized for.
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_webgl_amd.app import FrameLoop  # noqa: E402

spps = [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8, 25]
for spp in spps:
    loop = FrameLoop(1280, 702, mode="reference")
    loop.state.set_quality(spp, 8)
    loop.state.set_flags(is_paused=False)
    n = max(32, (640 // spp) // 16 * 16)
    res = []
    for kind in ("groups", "single"):
        loop.tracer.clear_textures()
        best = None
        for rep in range(3):
            if kind == "groups":
                loop.frames(32, 100.0, 16.5)  # capture + warm
                loop.tracer.synchronize()
                t0 = time.perf_counter()
                loop.frames(n, 5000.0 + 100 * rep, 16.5)
            else:
                for k in range(8):
                    loop.frame(100.0 + 16.5 * k)
                loop.tracer.synchronize()
                t0 = time.perf_counter()
                for k in range(n):
                    loop.frame(9000.0 + 16.5 * k)
            if not loop.tracer.wait(60.0):
                print("WATCHDOG", spp, kind, file=sys.stderr, flush=True)
                os._exit(3)
            dt = (time.perf_counter() - t0) / n * 1e3
            best = dt if best is None else min(best, dt)
        res.append(best)
    print("%2d spp per frame: %.4f ms per frame in groups (%d frames), %.4f ms issued one by one" % (spp, res[0], n, res[1]), flush=True)
    loop.close()
