"""Deterministic synthetic scenes for the BASELINE.json configs.

The reference ships exactly one scene — State::default's 9 spheres (src/state.rs:148-257),
available here as `default_scene()` through the C ABI — and no generators.  Everything else in
this file is synthetic input made by the build: a SplitMix64 stream (state seeded with the listed
constant, double in [0,1) = (x >> 11) * 2^-53) produces f64 scene values which are narrowed to
f32 exactly like webgl::set_geometry does (src/webgl.rs:225-274), so the oracle and the HIP path
consume identical bytes.
"""
import ctypes as C
import math
from dataclasses import dataclass, field

import numpy as np

from . import abi
from .abi import PT_DIFFUSE, PT_EMISSIVE, PT_GLASS, PT_METAL, SPHERE_DTYPE

MASK64 = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed):
        self.state = seed & MASK64

    def next_u64(self):
        self.state = (self.state + 0x9E3779B97F4A7C15) & MASK64
        z = self.state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
        return z ^ (z >> 31)

    def uniform(self, lo=0.0, hi=1.0):
        x = (self.next_u64() >> 11) * (1.0 / 9007199254740992.0)
        return lo + (hi - lo) * x


@dataclass
class Scene:
    name: str
    spheres: np.ndarray  # SPHERE_DTYPE
    params: abi.PtParams  # camera + width/height + spp per pass + depth + background; time = 0
    n_passes: int  # passes of params.samples_per_pixel samples; pass p uses u_time = p
    note: str = ""
    extra: dict = field(default_factory=dict)

    @property
    def total_spp(self):
        return self.n_passes * self.params.samples_per_pixel


def _sphere(center, radius, mtype, albedo, fuzz=0.0, ri=0.0):
    return (center, radius, mtype, albedo, fuzz, ri)


def _pack(items):
    out = np.zeros(len(items), dtype=SPHERE_DTYPE)
    for i, (c, r, t, a, fz, ri) in enumerate(items):
        out[i]["center"] = np.asarray(c, dtype=np.float64).astype(np.float32)
        out[i]["radius"] = np.float32(r)
        out[i]["type"] = t
        out[i]["albedo"] = np.asarray(a, dtype=np.float64).astype(np.float32)
        out[i]["fuzz"] = np.float32(fz)
        out[i]["refraction_index"] = np.float32(ri)
        out[i]["uuid"] = i  # glsl::set_sphere_uuids, src/glsl.rs:84-88
    return out


def _base_params(spp, max_depth, background=abi.PT_BG_SKY):
    p = abi.PtParams()
    p.time = 0.0
    p.samples_per_pixel = spp
    p.max_depth = max_depth
    p.render_count = 0
    p.should_average = 1
    p.last_frame_weight = 1.0
    p.background_mode = background
    p.band_rows = 8
    p.band_index = 0
    p.band_count = 1
    return p


def _state_camera(lib, p, width, height):
    cam = abi.PtCameraIn()
    rc = lib.pt_default_camera(width, height, C.byref(cam))
    assert rc == 0
    rc = lib.pt_camera_from_state(C.byref(cam), C.byref(p))
    assert rc == 0
    return cam


def _look_at(lib, p, width, height, look_from, look_at, vfov_deg, aperture, focus):
    la = abi.PtLookAtIn()
    la.width, la.height = width, height
    la.look_from = abi.d3(*look_from)
    la.look_at = abi.d3(*look_at)
    la.vup = abi.d3(0.0, 1.0, 0.0)
    la.vfov_radians = math.radians(vfov_deg)
    la.focus_distance = focus
    la.aperture = aperture
    rc = lib.pt_camera_look_at(C.byref(la), C.byref(p))
    assert rc == 0
    return la


def _lib():
    from ._lib import load

    return load()


# ---------------------------------------------------------------------------------------------
def default_scene(width=1280, height=702, spp=25, max_depth=8, n_passes=1):
    """State::default: 9 spheres incl. two negative radii (src/state.rs:148-257), camera
    (0,0,1) yaw -90 fov pi/3 aperture 0 focus .75 (src/state.rs:98-125), depth 8 (:128),
    25 spp while paused (src/webgl.rs:342-346).  1280x702 is images/14.png's size."""
    lib = _lib()
    host = (abi.PtHostSphere * 16)()
    n = lib.pt_default_scene(host, 16)
    assert n == 9
    dev = (abi.PtSphere * n)()
    assert lib.pt_narrow_spheres(host, n, dev) == 0
    spheres = np.frombuffer(bytes(dev), dtype=SPHERE_DTYPE).copy()
    p = _base_params(spp, max_depth)
    _state_camera(lib, p, width, height)
    return Scene("default", spheres, p, n_passes, "State::default scene")


def config1(width=400, height=225, spp=16, max_depth=8):
    """BASELINE config 1: 3-sphere Lambertian scene, 400x225, 16 spp, 8 bounces, u_time = 0."""
    items = [
        _sphere((0.0, -100.5, -1.0), 100.0, PT_DIFFUSE, (0.8, 0.8, 0.0)),
        _sphere((0.0, 0.0, -1.0), 0.5, PT_DIFFUSE, (0.7, 0.3, 0.3)),
        _sphere((-1.0, 0.0, -1.0), 0.5, PT_DIFFUSE, (0.8, 0.8, 0.8)),
    ]
    p = _base_params(spp, max_depth)
    _state_camera(_lib(), p, width, height)
    return Scene("config1_lambert3", _pack(items), p, 1, "3 diffuse spheres, State::default camera")


def cover_spheres(seed=0x5EED0002):
    """Shirley 'In One Weekend' final scene, SplitMix64-seeded."""
    rng = SplitMix64(seed)
    items = [_sphere((0.0, -1000.0, 0.0), 1000.0, PT_DIFFUSE, (0.5, 0.5, 0.5))]
    for a in range(-11, 11):
        for b in range(-11, 11):
            choose = rng.uniform()
            cx = a + 0.9 * rng.uniform()
            cz = b + 0.9 * rng.uniform()
            c = (cx, 0.2, cz)
            if math.sqrt((cx - 4.0) ** 2 + (0.2 - 0.2) ** 2 + cz**2) <= 0.9:
                continue
            if choose < 0.8:
                alb = tuple(rng.uniform() * rng.uniform() for _ in range(3))
                items.append(_sphere(c, 0.2, PT_DIFFUSE, alb))
            elif choose < 0.95:
                alb = tuple(rng.uniform(0.5, 1.0) for _ in range(3))
                fz = rng.uniform(0.0, 0.5)
                items.append(_sphere(c, 0.2, PT_METAL, alb, fuzz=fz))
            else:
                items.append(_sphere(c, 0.2, PT_GLASS, (1.0, 1.0, 1.0), ri=1.5))
    items.append(_sphere((0.0, 1.0, 0.0), 1.0, PT_GLASS, (1.0, 1.0, 1.0), ri=1.5))
    items.append(_sphere((-4.0, 1.0, 0.0), 1.0, PT_DIFFUSE, (0.4, 0.2, 0.1)))
    items.append(_sphere((4.0, 1.0, 0.0), 1.0, PT_METAL, (0.7, 0.6, 0.5), fuzz=0.0))
    return _pack(items)


def config2(width=1920, height=1080, spp_per_pass=64, n_passes=16, max_depth=50):
    """BASELINE config 2 (the metric's config): cover scene, 1920x1080, 1024 spp as 16 passes of
    64 (u_time = pass index, SURVEY.md §7 'seed stalls'), 50 bounces."""
    p = _base_params(spp_per_pass, max_depth)
    _look_at(_lib(), p, width, height, (13.0, 2.0, 3.0), (0.0, 0.0, 0.0), 20.0, 0.1, 10.0)
    return Scene("config2_cover", cover_spheres(), p, n_passes, "Shirley cover scene")


def config3(width=3840, height=2160, spp_per_pass=64, n_passes=64, max_depth=50):
    """BASELINE config 3: cover scene at 3840x2160, 4096 spp, row bands over 8 GPUs."""
    s = config2(width, height, spp_per_pass, n_passes, max_depth)
    s.name = "config3_cover_4k"
    return s


def config4(width=1024, height=1024, spp_per_pass=64, n_passes=128, max_depth=50):
    """BASELINE config 4 (build extension): closed room of six huge diffuse
    spheres, glass + metal spheres inside, one emissive sphere (type 3), black background."""
    R = 1000.0
    items = [
        _sphere((-(R + 1.0), 0.0, 0.0), R, PT_DIFFUSE, (0.65, 0.05, 0.05)),  # left, red
        _sphere(((R + 1.0), 0.0, 0.0), R, PT_DIFFUSE, (0.12, 0.45, 0.15)),  # right, green
        _sphere((0.0, -(R + 1.0), 0.0), R, PT_DIFFUSE, (0.73, 0.73, 0.73)),  # floor
        _sphere((0.0, (R + 1.0), 0.0), R, PT_DIFFUSE, (0.73, 0.73, 0.73)),  # ceiling
        _sphere((0.0, 0.0, -(R + 1.0)), R, PT_DIFFUSE, (0.73, 0.73, 0.73)),  # back
        _sphere((0.0, 0.0, (R + 4.0)), R, PT_DIFFUSE, (0.73, 0.73, 0.73)),  # behind the camera
        _sphere((-0.45, -0.65, 0.3), 0.35, PT_GLASS, (1.0, 1.0, 1.0), ri=1.5),
        _sphere((0.45, -0.6, -0.3), 0.4, PT_METAL, (0.8, 0.85, 0.88), fuzz=0.05),
        _sphere((0.0, 0.72, 0.0), 0.2, PT_EMISSIVE, (15.0, 15.0, 15.0)),
    ]
    p = _base_params(spp_per_pass, max_depth, background=abi.PT_BG_BLACK)
    _look_at(_lib(), p, width, height, (0.0, 0.0, 3.6), (0.0, 0.0, 0.0), 40.0, 0.0, 3.6)
    return Scene("config4_room", _pack(items), p, n_passes, "enclosed room, emissive light")


def field_spheres(n=10000, seed=0x5EED0005):
    rng = SplitMix64(seed)
    items = [_sphere((0.0, -1000.0, 0.0), 1000.0, PT_DIFFUSE, (0.5, 0.5, 0.5))]
    for _ in range(n):
        c = (rng.uniform(-50.0, 50.0), rng.uniform(0.2, 20.0), rng.uniform(-50.0, 50.0))
        r = rng.uniform(0.1, 0.5)
        m = rng.uniform()
        if m < 0.7:
            items.append(_sphere(c, r, PT_DIFFUSE, tuple(rng.uniform() for _ in range(3))))
        elif m < 0.9:
            alb = tuple(rng.uniform(0.5, 1.0) for _ in range(3))
            items.append(_sphere(c, r, PT_METAL, alb, fuzz=rng.uniform(0.0, 0.5)))
        else:
            items.append(_sphere(c, r, PT_GLASS, (1.0, 1.0, 1.0), ri=1.5))
    return _pack(items)


def config5(width=1920, height=1080, spp_per_pass=64, n_passes=4, max_depth=50, n=10000):
    """BASELINE config 5: 10 000-sphere random field (+ground), 1920x1080, 256 spp."""
    p = _base_params(spp_per_pass, max_depth)
    _look_at(_lib(), p, width, height, (60.0, 15.0, 60.0), (0.0, 8.0, 0.0), 40.0, 0.0, 80.0)
    return Scene("config5_field", field_spheres(n), p, n_passes, "%d-sphere field" % n)


CONFIGS = {
    "default": default_scene,
    "config1": config1,
    "config2": config2,
    "config3": config3,
    "config4": config4,
    "config5": config5,
}
