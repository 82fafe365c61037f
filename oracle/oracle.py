"""ctypes binding of the CPU oracle (oracle/libpt_oracle.so).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the
product package (ray_tracer_webgl_amd/ must not import this module)."""
import ctypes as C
import os
import subprocess

import numpy as np

from ray_tracer_webgl_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpt_oracle.so")
_lib = None


class OraHit(C.Structure):
    _fields_ = [("hit", C.c_int32), ("index", C.c_int32), ("t", C.c_float), ("point", C.c_float * 3),
                ("normal", C.c_float * 3), ("front_face", C.c_int32)]


class OraScatter(C.Structure):
    _fields_ = [("did_scatter", C.c_int32), ("attenuation", C.c_float * 3), ("origin", C.c_float * 3),
                ("direction", C.c_float * 3), ("seed_after", C.c_float)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def load():
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("PT_ORACLE_LIB", LIB_PATH)  # bench.py's cpu_baseline leg: the -march=native rebuild
    if not os.path.exists(path):
        build()
        path = LIB_PATH
    L = C.CDLL(path)
    fp = C.POINTER(C.c_float)
    sp = C.POINTER(abi.PtSphere)
    pp = C.POINTER(abi.PtParams)
    L.ora_base_hash.restype = C.c_uint32
    L.ora_base_hash.argtypes = [C.c_uint32, C.c_uint32]
    L.ora_hash1.restype = C.c_float
    L.ora_hash1.argtypes = [fp]
    L.ora_hash2.argtypes = [fp, fp]
    L.ora_hash3.argtypes = [fp, fp]
    L.ora_sincos2pi.argtypes = [C.c_float, fp, fp]
    L.ora_cbrt.restype = C.c_float
    L.ora_cbrt.argtypes = [C.c_float]
    L.ora_v_position.restype = C.c_float
    L.ora_v_position.argtypes = [C.c_uint32, C.c_uint32]
    L.ora_init_seed.restype = C.c_float
    L.ora_init_seed.argtypes = [C.c_float, C.c_float, C.c_float]
    L.ora_random_in_unit_sphere.argtypes = [fp, fp]
    L.ora_random_in_unit_circle.argtypes = [fp, fp]
    L.ora_hit_sphere.argtypes = [sp, fp, fp, C.c_float, C.c_float, C.POINTER(OraHit)]
    L.ora_hit_world.argtypes = [sp, C.c_uint32, fp, fp, C.POINTER(OraHit)]
    L.ora_scatter.argtypes = [sp, C.c_uint32, fp, fp, C.c_float, C.POINTER(OraScatter)]
    L.ora_ray_color.argtypes = [sp, C.c_uint32, pp, fp, fp, fp, fp, C.POINTER(C.c_uint64)]
    L.ora_camera_ray.argtypes = [pp, C.c_float, C.c_float, fp, fp, fp]
    L.ora_local_rows.restype = C.c_uint32
    L.ora_local_rows.argtypes = [pp]
    L.ora_render_pass.restype = C.c_uint64
    L.ora_render_pass.argtypes = [sp, C.c_uint32, pp, C.c_float, fp] + [C.c_uint32] * 5
    L.ora_render_passes.restype = C.c_uint64
    L.ora_render_passes.argtypes = [sp, C.c_uint32, pp, C.c_uint32, fp] + [C.c_uint32] * 5
    L.ora_resolve.argtypes = [fp, C.c_size_t, C.c_uint32, C.c_int, fp]
    L.ora_resolve_rgba8.argtypes = [fp, C.c_size_t, C.c_uint32, C.c_int, C.c_void_p]
    L.ora_blend_rgba8.argtypes = [fp, C.c_size_t, C.c_uint32, pp, C.c_void_p, C.c_void_p]
    L.ora_camera_from_state.argtypes = [C.POINTER(abi.PtCameraIn), pp]
    L.ora_camera_look_at.argtypes = [C.POINTER(abi.PtLookAtIn), pp]
    L.ora_center_hit_f64.argtypes = [C.POINTER(abi.PtHostSphere), C.c_uint32, C.POINTER(abi.PtCameraIn),
                                     C.POINTER(abi.PtCenterHit)]
    _lib = L
    return L


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def render(spheres, params, n_passes=1, window=None, nthreads=None, accum=None):
    """Render n_passes passes (u_time = params.time + k) of the owned rows on the CPU.

    window = (x0, x1, y0, y1) in global pixel coordinates restricts the computed pixels.
    Returns (accum ndarray (local_rows, width, 4) float32, segments)."""
    L = load()
    ptr, n, keep = abi.spheres_as_ctypes(spheres)
    p = params.copy()
    rows = L.ora_local_rows(C.byref(p))
    if accum is None:
        accum = np.zeros((rows, p.width, 4), dtype=np.float32)
    x0, x1, y0, y1 = window if window is not None else (0, p.width, 0, p.height)
    if nthreads is None:
        nthreads = os.cpu_count() or 1
    seg = L.ora_render_passes(ptr, n, C.byref(p), int(n_passes), accum.ctypes.data_as(C.POINTER(C.c_float)),
                              int(x0), int(x1), int(y0), int(y1), int(nthreads))
    return accum, int(seg)


def resolve(accum, total_spp, gamma=True):
    L = load()
    a = np.ascontiguousarray(accum, dtype=np.float32)
    out = np.empty_like(a)
    L.ora_resolve(a.ctypes.data_as(C.POINTER(C.c_float)), a.size // 4, int(total_spp), 1 if gamma else 0,
                  out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def resolve_rgba8(accum, total_spp, gamma=True):
    L = load()
    a = np.ascontiguousarray(accum, dtype=np.float32)
    out = np.empty(a.shape[:-1] + (4,), dtype=np.uint8)
    L.ora_resolve_rgba8(a.ctypes.data_as(C.POINTER(C.c_float)), a.size // 4, int(total_spp), 1 if gamma else 0,
                        out.ctypes.data_as(C.c_void_p))
    return out


def blend_rgba8(accum, total_spp, params, prev):
    L = load()
    a = np.ascontiguousarray(accum, dtype=np.float32)
    prev = np.ascontiguousarray(prev, dtype=np.uint8)
    out = np.empty_like(prev)
    p = params.copy()
    L.ora_blend_rgba8(a.ctypes.data_as(C.POINTER(C.c_float)), a.size // 4, int(total_spp), C.byref(p),
                      prev.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    return out
