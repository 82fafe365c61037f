"""ctypes mirrors of the POD structs in include/ptrace.h.

Field order, types and sizes must match the header exactly (tests/test_abi.py checks the sizes
against a C program compiled from the header).  Reference citations: PtSphere mirrors the fields
webgl::set_geometry uploads (src/webgl.rs:225-274); PtParams mirrors the uniform block
(static/shader.frag:79-102) as uploaded by Uniforms::run_setters (src/webgl.rs:629-633);
PtCameraIn mirrors the camera members of State (src/state.rs:31-57).
"""
import ctypes as C

import numpy as np

PT_ABI_VERSION = 5

PT_OK = 0
PT_ERR_INVALID = -1
PT_ERR_NO_DEVICE = -2
PT_ERR_HIP = -3
PT_ERR_NOT_READY = -4
PT_ERR_CAPACITY = -5

# static/shader.frag:45-47, src/glsl.rs:10-24 (+ build extension 3)
PT_DIFFUSE = 0
PT_METAL = 1
PT_GLASS = 2
PT_EMISSIVE = 3

PT_BG_SKY = 0
PT_BG_BLACK = 1

PT_GEOM_AUTO, PT_GEOM_LDS, PT_GEOM_SCALAR, PT_GEOM_BVH, PT_GEOM_GRID, PT_GEOM_SMALL = 0, 1, 2, 3, 4, 5
PT_OPT_GEOMETRY_PATH, PT_OPT_COUNT_WORK, PT_OPT_CARRY_LANES, PT_OPT_REFILL_MIN, PT_OPT_RUSSIAN_ROULETTE, PT_OPT_GRID_FIT = 1, 2, 3, 4, 5, 6
PT_TIME_STEP_DECORRELATED = 0.3618034  # include/ptrace.h
PT_STREAM_LEGACY = 1  # include/ptrace.h: hipStreamLegacy, the default (NULL) stream by name
GEOM_NAMES = {0: "auto", 1: "lds", 2: "scalar", 3: "bvh", 4: "grid", 5: "small"}

f3 = C.c_float * 3
d3 = C.c_double * 3


class PtSphere(C.Structure):
    _fields_ = [
        ("center", f3),
        ("radius", C.c_float),
        ("type", C.c_int32),
        ("albedo", f3),
        ("fuzz", C.c_float),
        ("refraction_index", C.c_float),
        ("uuid", C.c_int32),
        ("_pad", C.c_int32),
    ]


# numpy view of the same 48-byte record, for bulk scene generation
SPHERE_DTYPE = np.dtype(
    [
        ("center", "<f4", 3),
        ("radius", "<f4"),
        ("type", "<i4"),
        ("albedo", "<f4", 3),
        ("fuzz", "<f4"),
        ("refraction_index", "<f4"),
        ("uuid", "<i4"),
        ("_pad", "<i4"),
    ]
)
assert SPHERE_DTYPE.itemsize == C.sizeof(PtSphere) == 48


class PtParams(C.Structure):
    _fields_ = [
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("time", C.c_float),
        ("samples_per_pixel", C.c_int32),
        ("max_depth", C.c_int32),
        ("camera_origin", f3),
        ("horizontal", f3),
        ("vertical", f3),
        ("lower_left_corner", f3),
        ("u", f3),
        ("v", f3),
        ("lens_radius", C.c_float),
        ("render_count", C.c_int32),
        ("should_average", C.c_int32),
        ("last_frame_weight", C.c_float),
        ("background_mode", C.c_int32),
        ("band_rows", C.c_uint32),
        ("band_index", C.c_uint32),
        ("band_count", C.c_uint32),
        ("time_step", C.c_float),
        ("first_pass", C.c_uint32),
    ]

    def copy(self):
        out = PtParams()
        C.memmove(C.byref(out), C.byref(self), C.sizeof(PtParams))
        return out


class PtCameraIn(C.Structure):
    _fields_ = [
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("camera_origin", d3),
        ("yaw_degrees", C.c_double),
        ("pitch_degrees", C.c_double),
        ("vup", d3),
        ("fov_radians", C.c_double),
        ("focus_distance", C.c_double),
        ("aperture", C.c_double),
    ]


class PtLookAtIn(C.Structure):
    _fields_ = [
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("look_from", d3),
        ("look_at", d3),
        ("vup", d3),
        ("vfov_radians", C.c_double),
        ("focus_distance", C.c_double),
        ("aperture", C.c_double),
    ]


class PtStats(C.Structure):
    _fields_ = [
        ("segments", C.c_uint64),
        ("samples", C.c_uint64),
        ("sphere_tests", C.c_uint64),
        ("render_kernel_ms", C.c_double),
        ("render_launches", C.c_uint32),
        ("total_spp", C.c_uint32),
        ("n_spheres", C.c_uint32),
        ("local_rows", C.c_uint32),
        ("geometry_path", C.c_uint32),
        ("geometry_tuned", C.c_uint32),
        ("bvh_nodes", C.c_uint32),
        ("bvh_slots", C.c_uint32),
        ("bvh_outliers", C.c_uint32),
        ("bvh_depth", C.c_uint32),
        ("grid_cells", C.c_uint32 * 3),
        ("grid_entries", C.c_uint32),
        ("grid_always", C.c_uint32),
        ("grid_fit_stale", C.c_uint32),  # 0 fits / 1 camera outside the near region (far path: refit) / 2 looser than needed
        ("work", C.c_uint64 * 8),
        ("grid_near_factor", C.c_float),
        ("grid_need_factor", C.c_float),
        ("far_rays", C.c_uint64),
        ("grid_kernel_build", C.c_uint32),  # 1 pt_trace_kernel_grid (all staged in the LDS) / 2 _grid_cells / 3 _grid_gmem / 0 no grid
        ("_pad2", C.c_uint32),
    ]


class PtHostSphere(C.Structure):
    """src/glsl.rs:27-40: f64 centre/radius/albedo, f32 fuzz/refraction_index."""

    _fields_ = [
        ("center", d3),
        ("radius", C.c_double),
        ("type", C.c_int32),
        ("uuid", C.c_int32),
        ("albedo", d3),
        ("fuzz", C.c_float),
        ("refraction_index", C.c_float),
    ]


class PtCenterHit(C.Structure):
    _fields_ = [
        ("t", C.c_double),
        ("hit_point", d3),
        ("normal", d3),
        ("front_face", C.c_int32),
        ("uuid", C.c_int32),
    ]


class PtStateView(C.Structure):
    """Snapshot of the reference's State (src/state.rs:31-94), f64 like the original."""

    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32),
        ("samples_per_pixel", C.c_uint32), ("max_depth", C.c_uint32),
        ("aspect_ratio", C.c_double),
        ("camera_origin", d3), ("camera_front", d3), ("vup", d3),
        ("yaw", C.c_double), ("pitch", C.c_double), ("camera_field_of_view", C.c_double),
        ("u", d3), ("v", d3), ("w", d3),
        ("aperture", C.c_double), ("lens_radius", C.c_double), ("focus_distance", C.c_double),
        ("viewport_height", C.c_double), ("viewport_width", C.c_double),
        ("horizontal", d3), ("vertical", d3), ("lower_left_corner", d3),
        ("cursor_point", d3),
        ("selected_object", C.c_int32),
        ("is_paused", C.c_int32), ("should_average", C.c_int32), ("should_render", C.c_int32),
        ("even_odd_count", C.c_uint32), ("render_count", C.c_uint32), ("max_render_count", C.c_uint32),
        ("last_frame_weight", C.c_float),
        ("n_spheres", C.c_uint32),
    ]


KEY_W, KEY_A, KEY_S, KEY_D, KEY_SPACE, KEY_SHIFT = 1, 2, 4, 8, 16, 32


def spheres_as_ctypes(spheres):
    """numpy SPHERE_DTYPE array (or PtSphere ctypes array) -> (PtSphere pointer, n, keepalive)."""
    if isinstance(spheres, np.ndarray):
        arr = np.ascontiguousarray(spheres, dtype=SPHERE_DTYPE)
        return arr.ctypes.data_as(C.POINTER(PtSphere)), int(arr.shape[0]), arr
    n = len(spheres)
    return C.cast(spheres, C.POINTER(PtSphere)), n, spheres


def local_rows(height, band_rows, band_index, band_count):
    """Rows y in [0,height) with (y // band_rows) % band_count == band_index (include/ptrace.h)."""
    if band_count <= 1 or band_rows == 0:
        return int(height)
    full, rem = divmod(int(height), band_rows * band_count)
    n = full * band_rows
    lo = band_index * band_rows
    n += max(0, min(rem - lo, band_rows))
    return n


def owned_rows(height, band_rows, band_index, band_count):
    """Ascending global row indices owned by band_index."""
    ys = np.arange(height, dtype=np.int64)
    if band_count <= 1 or band_rows == 0:
        return ys
    return ys[(ys // band_rows) % band_count == band_index]
