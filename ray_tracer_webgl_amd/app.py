"""FrameLoop — the reference's requestAnimationFrame closure (src/lib.rs:65-104) as a headless
harness over the C ABI: update_position -> should_render -> update_render_globals ->
run_setters -> render -> (save).  Two display modes:

  * "reference": every frame is one fresh pass at u_time = `now`, blended with the previous
    RGBA8 frame by the shader's render() rule (static/shader.frag:387-404) into ping-pong
    textures (src/webgl.rs:186-204) — the reference's on-screen behaviour, 8-bit quantisation
    and gamma-space averaging included.  The textures live on the device (pt_render_frame); a run
    of ticks at a constant frame interval is replayed from hipGraphs (`frames`, pt_render_frames:
    groups of 64, 16 and 4 frames, each traced by one launch) with the per-frame state counted on the device.
  * "linear": passes accumulate as fp32 linear radiance (north_star's "accumulated radiance")
    and are resolved at read-out; a camera change (render_count reset to 0,
    src/state.rs:343-346) clears the accumulation.
"""
import time

import numpy as np

from .state import State
from .tracer import PathTracer


class FrameLoop:
    # frames a grid may stay LOOSER than the camera needs (PtStats.grid_fit_stale == 2: a few per cent of speed) before it
    # is refitted; a grid that is too SMALL (== 1: every primary ray tested against the whole list) is refitted at once
    LOOSE_FRAMES = 32

    def __init__(self, width, height, device=0, mode="reference", host_spheres=None):
        assert mode in ("reference", "linear")
        self.mode = mode
        self.state = State(width, height)
        if host_spheres is not None:  # another scene than State::default's nine spheres (abi.PtHostSphere records, f64 like src/glsl.rs:27-40)
            self.state.set_spheres(host_spheres)
        self.tracer = PathTracer(width, height, device=device)
        self.tracer.set_spheres(self.state.spheres())  # set_geometry, once (src/lib.rs:57)
        self.grid_refits = 0
        self._loose = 0
        # two RGBA8 textures cleared to 0 (alpha 0 = "no data", shader.frag:391): in HBM, owned by the context
        self.tracer.clear_textures()
        self._canvas = None
        self.prev_now = 0.0
        self.frames_rendered = 0

    def close(self):
        self.tracer.close()
        self.state.close()

    @property
    def canvas(self):
        """The last frame drawn, RGBA8 (read back from the device on demand)."""
        if self.mode == "reference":
            return self.tracer.read_canvas()
        return self._canvas

    @property
    def textures(self):
        return [self.tracer.read_texture(0), self.tracer.read_texture(1)]

    def _keep_the_grid_fitted(self):
        """The reference moves its camera every tick a key is held (State::update_position, src/state.rs:411-441); the grid
        of a large scene is fitted to where rays START (pt_tune / pt_refit_grid).  After the tick's uniforms are up: a camera
        that has left the fitted region gets a grid that serves it before the frame is traced (pt_refit_grid: ~2 ms of host
        work for 10 000 spheres, the stream is drained), a grid looser than needed is tightened once the camera has stayed
        inside for LOOSE_FRAMES frames.  Speed only: the frame's bits do not depend on the grid.  Scenes without a grid
        (State::default's nine spheres) answer 0 from host arithmetic."""
        fit = self.tracer.grid_fit()
        if fit == 1 or (fit == 2 and self._loose + 1 >= self.LOOSE_FRAMES):
            self.tracer.refit_grid()
            self.grid_refits += 1
            self._loose = 0
        elif fit == 2:
            self._loose += 1
        else:
            self._loose = 0

    def frame(self, now_ms, should_save=False):
        """One rAF tick.  Returns True when a frame was rendered."""
        st = self.state
        dt = now_ms - self.prev_now
        st.update_position(dt)                      # src/lib.rs:73
        if not st.should_render(should_save):       # :77-82
            return False
        st.update_render_globals()                  # :93
        self.prev_now = now_ms                      # update_moving_fps_array, src/state.rs:402
        v = st.view()
        p = st.to_params(now_ms)                    # uniforms.run_setters, :96
        if self.mode == "reference":
            self.tracer.set_params(p)
            self._keep_the_grid_fitted()
            # webgl::render: previous frame = textures[(even_odd + 1) % 2] (src/webgl.rs:186-190), draw
            # to the canvas (:193-194) and, when averaging, to the other texture (:197-204)
            self.tracer.render_frame(v.even_odd_count)
        else:
            if v.render_count <= 1:  # accumulation restarts after any camera change
                self.tracer.reset()
            self.tracer.set_params(p)
            self._keep_the_grid_fitted()
            self.tracer.render()
            self._canvas = self.tracer.resolve_rgba8(True)
        self.frames_rendered += 1
        return True

    def frames(self, n, first_now_ms, interval_ms):
        """n rAF ticks at a constant frame interval, i.e. exactly what n calls of
        frame(first_now_ms + k * interval_ms) do; returns how many of them drew a frame.  When nothing but
        the clock changes between the ticks (averaging on, no movement key held: update_position changes
        nothing, should_render stays true while unpaused) they are ONE pt_render_frames call: the first
        tick's uniforms go up once, the graph of one frame is replayed n times, u_time / render_count /
        even-odd advance on the device as the n calls would advance them.  Otherwise the ticks differ in
        more than the clock.  BIT-identical to the n single ticks when `first_now_ms`, `interval_ms` and their products are
        exact in fp32 (the tests' 100.0 and 16.5; any whole or half millisecond below 2^23): the device computes
        u_time = fp32(first_now_ms) + float(k) * fp32(interval_ms), a single tick uploads fp32(first_now_ms + k * interval_ms)
        rounded from f64 (gl.uniform1f(now as f32), src/webgl.rs:322).  For a real rAF interval like 16.6667 the two differ in
        the last bit for some k: other seeds, so other — statistically equivalent — samples.  Ticks that differ in more than
        the clock are never replayed: with should_average off only the first one draws (update_render_globals
        clears should_render, src/state.rs:443-447), with a key held every tick moves the camera
        (update_position, src/state.rs:411-441) — and they are issued one by one.  Reference mode only."""
        assert self.mode == "reference" and n >= 1
        st = self.state
        v0 = st.view()
        assert not v0.is_paused, "a paused State renders one frame per camera change, not a series"
        if not v0.should_average or st.keys != 0:
            return sum(1 for k in range(n) if self.frame(first_now_ms + k * interval_ms))
        st.update_position(first_now_ms - self.prev_now)
        if not st.should_render(False):
            return 0
        st.update_render_globals()
        v = st.view()
        p = st.to_params(first_now_ms)
        p.time_step = float(interval_ms)  # frame k: u_time = time + float(k) * interval (fp32, like the kernel's pass time)
        p.first_pass = 0
        self.tracer.set_params(p)
        self._keep_the_grid_fitted()  # (before the series' graphs are captured: a refit re-captures them by itself)
        self.tracer.render_frames(v.even_odd_count, v.max_render_count, n)
        for _ in range(n - 1):                       # the host's copy of the counters follows
            st.update_render_globals()
        self.prev_now = first_now_ms + (n - 1) * interval_ms
        self.frames_rendered += n
        return n


def frame_loop_benchmark(n_frames=400, warmup=16, width=1280, height=702, device=0, extra_legs=True):
    """bench.py --config default: the reference at its own operating point (State::default, 9 spheres,
    1280x702 = images/14.png, depth 8; src/state.rs:127-135).  Times (i) the animation loop — 1 spp per
    tick, blended into the RGBA8 textures, replayed from hipGraphs in groups of 64 — and (ii) the 25-spp frames the
    reference draws while paused (src/webgl.rs:342-346), plus (iii) the same ticks issued one by one
    from the host (uniform upload + three launches per frame) for comparison.  Returns the JSON dict."""
    from . import abi

    out = {}
    loop = FrameLoop(width, height, device=device, mode="reference")
    st = loop.state
    n_sph = st.view().n_spheres

    def run(kind, n):
        """kind: "graph" = the animation loop replayed from hipGraphs (groups of 64, 16 and 4 frames); "host" = the same ticks issued one
        by one; "paused" = what the reference draws while paused: ONE 25-spp frame after every camera
        change (render_count == 0, src/lib.rs:77-82, src/webgl.rs:342-346) — here after a yaw nudge"""
        paused = kind == "paused"
        st.set_flags(is_paused=paused)
        loop.tracer.clear_textures()
        now0, dt = 3000.0, 16.7

        def ticks(k0, count):
            if kind == "graph":
                assert loop.frames(count, now0 + dt * k0, dt) == count
            else:
                for k in range(k0, k0 + count):
                    if paused:
                        st.set_camera_angles(-90.0 + 0.25 * (1 + (k & 1)), 0.0)  # a camera change: render_count = 0
                    assert loop.frame(now0 + dt * k)

        ticks(0, max(85 if kind == "graph" else 1, warmup))  # first capture of every group size (64 + 16 + 4 + 1), tile order, clocks
        loop.tracer.synchronize()
        loop.tracer.reset()
        t0 = time.perf_counter()
        ticks(1000, n)
        loop.tracer.synchronize()
        t1 = time.perf_counter()
        s = loop.tracer.stats()
        spp = st.to_params(0.0).samples_per_pixel
        return {
            "frames": n, "spp_per_frame": int(spp),
            "ms_per_frame": round((t1 - t0) / n * 1e3, 4),
            "frames_per_s": round(n / (t1 - t0), 1),
            "mray_s": round(s.segments / (t1 - t0) / 1e6, 1),
            "segments_per_frame": round(s.segments / n, 1),
            "device_ms_per_frame": round(s.render_kernel_ms / n, 4) if kind == "graph" else None,
            "geometry_path": abi.GEOM_NAMES.get(s.geometry_path, "?"),
            "how": {"graph": "%d frames replayed from hipGraphs in groups of 64, 16 and 4 (one trace launch renders a group's frames as its passes, one "
                             "kernel runs their blends, one advance), per-frame state on the device" % n,
                    "host": "uniform upload + trace + blend issued per frame from the host",
                    "paused": "one frame per camera change, issued from the host (uniform upload + trace + blend)"}[kind],
        }

    anim = run("graph", n_frames)
    anim_host = run("host", min(n_frames, 200)) if extra_legs else None
    paused = run("paused", max(8, n_frames // 8)) if extra_legs else None
    st.set_flags(is_paused=False)
    canvas = loop.canvas
    # The dominant kernel of the animation loop on its own: the trace launch of one GROUP of 64 frames (64 passes of 1 spp,
    # pass k at u_time = now + k * interval, exactly the launch pt_render_frames captures), timed with HIP events on the
    # launch stream around that kernel alone (PtStats.render_kernel_ms) — what bench.py's `roofline` prices.
    group = 64
    p = st.to_params(3000.0)
    p.time_step, p.first_pass = 16.7, 0
    loop.tracer.set_params(p)
    loop.tracer.reserve_passes(group)
    for rep in range(3):
        loop.tracer.reset()
        n_launch = 4 if rep < 2 else 24
        for _ in range(n_launch):
            loop.tracer.render_passes(group)
        loop.tracer.synchronize()
    sg = loop.tracer.stats()
    n_pix = width * height
    group_kernel = {
        "kernel": "pt_trace_kernel_small_t%d" % (n_sph % 4),
        "passes_per_launch": group, "spp_per_pass": 1, "launches": int(sg.render_launches),
        "avg_launch_ms": round(sg.render_kernel_ms / max(sg.render_launches, 1), 5),
        "segments_per_launch": round(sg.segments / max(sg.render_launches, 1), 1),
        "n_spheres": int(n_sph), "pixels": n_pix,
    }
    loop.close()
    out = {
        "metric": "frames/s of the reference's animation loop (1 spp per tick, temporal blend) at %dx%d" % (width, height),
        "value": anim["frames_per_s"],
        "unit": "frames/s",
        "n_gpus": 1,
        "steps": n_frames,
        "warmup": warmup,
        "ms_per_step": anim["ms_per_frame"],
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "default: State::default (%d spheres; the shader holds at most 15, static/shader.frag:103), %dx%d, depth 8, "
                        "1 spp per frame blended into RGBA8 ping-pong textures (src/state.rs:127-135, src/lib.rs:65-104, "
                        "src/webgl.rs:180-205)" % (n_sph, width, height),
            "step": "one animation frame",
            "passes_per_launch": group, "spp_per_pass": 1,  # the trace launch of one group of frames (profiles/summarize.py keys on these)
        },
        "reference_claim": "\"less than a second\" for a decent render (README.md:6): the only performance statement the reference makes",
        "animation": anim,
        "animation_issued_per_frame_from_the_host": anim_host,
        "paused_25spp": paused,
        "group_trace_kernel": group_kernel,
        "canvas_mean_rgb": [round(float(x), 3) for x in canvas[..., :3].reshape(-1, 3).mean(0)],
    }
    return out
