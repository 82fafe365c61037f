"""Dev tool (GPU): what would overlapping consecutive frame groups be worth?

Two contexts on ONE device, each with a stream of its own, replay the reference's animation loop at the same time: group k of
one context overlaps the drain and the blend of group k of the other.  The aggregate frames/s against ONE context replaying the
same total is the upper bound of what a pipelined pt_render_frames (trace of group k + 1 beside the blend of group k) can gain.

    python tools/frames_overlap_probe.py [frames per context]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ray_tracer_webgl_amd.app import FrameLoop  # noqa: E402


def series(loops, n, k0):
    for lp in loops:
        assert lp.frames(n, 3000.0 + 16.7 * k0, 16.7) == n


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
    for n_ctx in (1, 2, 3):
        loops = [FrameLoop(1280, 702, device=0, mode="reference") for _ in range(n_ctx)]
        for lp in loops:
            lp.state.set_flags(is_paused=False)
            lp.tracer.clear_textures()
        series(loops, 85, 0)
        for lp in loops:
            lp.tracer.synchronize()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            # interleave the submissions in chunks of one group so that neither stream runs ahead of the other on the host side
            per = n // n_ctx // 64 * 64
            for k in range(0, per, 64):
                series(loops, 64, 1000 + k)
            for lp in loops:
                lp.tracer.synchronize()
            best = min(best, time.perf_counter() - t0)
        total = per * n_ctx
        print("%d context(s): %d frames in %.2f ms = %.0f frames/s (%.4f ms per frame)" % (n_ctx, total, best * 1e3, total / best, best * 1e3 / total), flush=True)
        for lp in loops:
            lp.close()


if __name__ == "__main__":
    main()
