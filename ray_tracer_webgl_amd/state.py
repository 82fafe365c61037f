"""State — Python handle on the reference's `State` object as mirrored by the C ABI
(`pt_state_*`, include/ptrace.h; C++ in csrc/pt_host.hpp).  Same members and update rules as
src/state.rs: camera (origin, yaw/pitch, fov, focus, aperture), derived basis, render
bookkeeping (`render_count`, `even_odd_count`, `should_render`), movement (`KeydownMap`,
`update_position` with autofocus through the f64 pick ray)."""
import ctypes as C

import numpy as np

from . import abi
from ._lib import load


class State:
    def __init__(self, width, height):
        self.lib = load()
        self._h = C.c_void_p()
        rc = self.lib.pt_state_create(C.byref(self._h), int(width), int(height))
        if rc != 0:
            raise ValueError("pt_state_create failed: %d" % rc)
        self.keys = 0  # the KeydownMap as last set through set_keys (abi.KEY_* mask)

    def close(self):
        if self._h:
            self.lib.pt_state_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ok(self, rc):
        if rc < 0:
            raise ValueError("pt_state call failed: %d" % rc)
        return rc

    def view(self):
        v = abi.PtStateView()
        self._ok(self.lib.pt_state_get(self._h, C.byref(v)))
        return v

    # src/state.rs:349-358
    def set_fov(self, radians):
        self._ok(self.lib.pt_state_set_fov(self._h, float(radians)))

    def set_camera_angles(self, yaw, pitch):
        self._ok(self.lib.pt_state_set_camera_angles(self._h, float(yaw), float(pitch)))

    def set_camera_origin(self, origin):
        self._ok(self.lib.pt_state_set_camera_origin(self._h, abi.d3(*origin)))

    def set_lens(self, aperture, focus_distance):
        self._ok(self.lib.pt_state_set_lens(self._h, float(aperture), float(focus_distance)))

    def set_quality(self, samples_per_pixel, max_depth):
        self._ok(self.lib.pt_state_set_quality(self._h, int(samples_per_pixel), int(max_depth)))

    def set_flags(self, is_paused, should_average=True, last_frame_weight=1.0):
        self._ok(self.lib.pt_state_set_flags(self._h, int(is_paused), int(should_average), float(last_frame_weight)))

    def set_keys(self, mask):
        self._ok(self.lib.pt_state_set_keys(self._h, int(mask)))
        self.keys = int(mask)

    # src/state.rs:411-450
    def update_position(self, dt_ms):
        self._ok(self.lib.pt_state_update_position(self._h, float(dt_ms)))

    def update_render_globals(self):
        self._ok(self.lib.pt_state_update_render_globals(self._h))

    def resize(self, width, height):
        self._ok(self.lib.pt_state_resize(self._h, int(width), int(height)))

    def should_render(self, should_save=False):  # src/lib.rs:77-82
        return bool(self._ok(self.lib.pt_state_should_render(self._h, int(should_save))))

    def set_spheres(self, host_spheres):
        n = len(host_spheres)
        arr = (abi.PtHostSphere * n)(*host_spheres)
        self._ok(self.lib.pt_state_set_spheres(self._h, arr, n))

    def spheres(self):
        """The f32 sphere records webgl::set_geometry would upload (src/webgl.rs:225-274)."""
        n = self._ok(self.lib.pt_state_spheres(self._h, None, 0))
        out = (abi.PtSphere * max(n, 1))()
        self._ok(self.lib.pt_state_spheres(self._h, out, n))
        return np.frombuffer(bytes(out), dtype=abi.SPHERE_DTYPE)[:n].copy()

    def to_params(self, now_ms):
        """Uniforms::run_setters (src/webgl.rs:279-593): State -> uniform block."""
        p = abi.PtParams()
        p.band_rows, p.band_index, p.band_count = 8, 0, 1
        self._ok(self.lib.pt_state_to_params(self._h, float(now_ms), C.byref(p)))
        return p


def adjusted_screen_dimensions(raw_width, raw_height):
    """dom::get_adjusted_screen_dimensions (src/dom.rs:277-291)."""
    w, h = C.c_uint32(), C.c_uint32()
    rc = load().pt_adjusted_screen_dimensions(float(raw_width), float(raw_height), C.byref(w), C.byref(h))
    if rc != 0:
        raise ValueError("bad window size")
    return w.value, h.value
