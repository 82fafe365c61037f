"""FrameLoop — the reference's requestAnimationFrame closure (src/lib.rs:65-104) as a headless
harness over the C ABI: update_position -> should_render -> update_render_globals ->
run_setters -> render -> (save).  Two display modes:

  * "reference": every frame is one fresh pass at u_time = `now`, blended with the previous
    RGBA8 frame by the shader's render() rule (static/shader.frag:387-404) into ping-pong
    textures (src/webgl.rs:186-204) — the reference's on-screen behaviour, 8-bit quantisation
    and gamma-space averaging included.
  * "linear": passes accumulate as fp32 linear radiance (north_star's "accumulated radiance")
    and are resolved at read-out; a camera change (render_count reset to 0,
    src/state.rs:343-346) clears the accumulation.
"""
import numpy as np

from .state import State
from .tracer import PathTracer


class FrameLoop:
    def __init__(self, width, height, device=0, mode="reference"):
        assert mode in ("reference", "linear")
        self.mode = mode
        self.state = State(width, height)
        self.tracer = PathTracer(width, height, device=device)
        self.tracer.set_spheres(self.state.spheres())  # set_geometry, once (src/lib.rs:57)
        # two RGBA8 textures cleared to 0 (alpha 0 = "no data", shader.frag:391)
        self.textures = [np.zeros((height, width, 4), np.uint8), np.zeros((height, width, 4), np.uint8)]
        self.canvas = None
        self.prev_now = 0.0
        self.frames_rendered = 0

    def close(self):
        self.tracer.close()
        self.state.close()

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
            self.tracer.reset()
            self.tracer.set_params(p)
            self.tracer.render()
            prev = self.textures[(v.even_odd_count + 1) % 2]   # src/webgl.rs:186-190
            out = self.tracer.blend_rgba8(prev)                # draw to canvas :193-194
            self.canvas = out
            if v.should_average:                               # draw to the FBO :197-204
                self.textures[v.even_odd_count % 2] = out
        else:
            if v.render_count <= 1:  # accumulation restarts after any camera change
                self.tracer.reset()
            self.tracer.set_params(p)
            self.tracer.render()
            self.canvas = self.tracer.resolve_rgba8(True)
        self.frames_rendered += 1
        return True
