"""Loader for libptrace.so (the C ABI of include/ptrace.h).

The library is built in-tree by `make -C ray_tracer_webgl_amd/csrc` (or
`__graft_entry__.build()`).  There is deliberately no fallback: if the shared object is missing
or a symbol is absent, importing users get an ImportError / AttributeError, never a CPU path.
"""
import ctypes as C
import os

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libptrace.so")

_lib = None

# every symbol include/ptrace.h declares: name -> (restype, argtypes)
_vp = C.c_void_p
_ctx = C.c_void_p
SIGNATURES = {
    "pt_create": (C.c_int, [C.POINTER(_ctx), C.c_int, C.c_uint32, C.c_uint32]),
    "pt_create_on_stream": (C.c_int, [C.POINTER(_ctx), C.c_int, C.c_uint32, C.c_uint32, _vp]),
    "pt_destroy": (C.c_int, [_ctx]),
    "pt_resize": (C.c_int, [_ctx, C.c_uint32, C.c_uint32]),
    "pt_set_spheres": (C.c_int, [_ctx, C.POINTER(abi.PtSphere), C.c_uint32]),
    "pt_set_params": (C.c_int, [_ctx, C.POINTER(abi.PtParams)]),
    "pt_render": (C.c_int, [_ctx]),
    "pt_render_passes": (C.c_int, [_ctx, C.c_uint32]),
    "pt_reserve_passes": (C.c_int, [_ctx, C.c_uint32]),
    "pt_reset_accum": (C.c_int, [_ctx]),
    "pt_synchronize": (C.c_int, [_ctx]),
    "pt_resolve": (C.c_int, [_ctx, _vp, C.c_int]),
    "pt_resolve_rgba8": (C.c_int, [_ctx, _vp, C.c_int]),
    "pt_accum_ptr": (C.c_int, [_ctx, C.POINTER(_vp), C.POINTER(C.c_size_t)]),
    "pt_bind_accum": (C.c_int, [_ctx, _vp, C.c_size_t]),
    "pt_read_accum": (C.c_int, [_ctx, _vp, C.c_size_t]),
    "pt_load_accum": (C.c_int, [_ctx, _vp, C.c_size_t]),
    "pt_set_stream": (C.c_int, [_ctx, _vp]),
    "pt_blend_rgba8": (C.c_int, [_ctx, _vp, _vp]),
    "pt_clear_textures": (C.c_int, [_ctx]),
    "pt_render_frame": (C.c_int, [_ctx, C.c_uint32]),
    "pt_render_frames": (C.c_int, [_ctx, C.c_uint32, C.c_uint32, C.c_uint32]),
    "pt_read_canvas": (C.c_int, [_ctx, _vp]),
    "pt_read_texture": (C.c_int, [_ctx, C.c_int, _vp]),
    "pt_write_texture": (C.c_int, [_ctx, C.c_int, _vp]),
    "pt_get_stats": (C.c_int, [_ctx, C.POINTER(abi.PtStats)]),
    "pt_set_option": (C.c_int, [_ctx, C.c_int, C.c_int]),
    "pt_tune": (C.c_int, [_ctx, C.c_uint32]),
    "pt_refit_grid": (C.c_int, [_ctx, C.c_int]),
    "pt_grid_fit": (C.c_int, [_ctx]),
    "pt_build_bvh": (C.c_int, [C.POINTER(abi.PtSphere), C.c_uint32, _vp, C.c_size_t, _vp, C.c_size_t, _vp, C.c_size_t, _vp, _vp,
                               _vp, C.c_size_t, C.POINTER(C.c_float), _vp, C.c_size_t]),
    "pt_build_grid": (C.c_int, [C.POINTER(abi.PtSphere), C.c_uint32, _vp, _vp, _vp, C.POINTER(C.c_float), _vp, C.c_size_t,
                                _vp, C.c_size_t, _vp, C.c_size_t]),
    "pt_grid_walk_constants": (C.c_int, [C.POINTER(abi.PtSphere), C.c_uint32, _vp]),
    "pt_last_error": (C.c_char_p, [_ctx]),
    "pt_abi_version": (C.c_int, []),
    "pt_device_count": (C.c_int, []),
    "pt_probe": (C.c_int, [_ctx, C.c_int, _vp, C.c_size_t, _vp, C.c_size_t, C.c_uint32]),
    "pt_local_rows": (C.c_uint32, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "pt_band_row": (C.c_uint32, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "pt_camera_from_state": (C.c_int, [C.POINTER(abi.PtCameraIn), C.POINTER(abi.PtParams)]),
    "pt_camera_look_at": (C.c_int, [C.POINTER(abi.PtLookAtIn), C.POINTER(abi.PtParams)]),
    "pt_narrow_spheres": (C.c_int, [C.POINTER(abi.PtHostSphere), C.c_uint32, C.POINTER(abi.PtSphere)]),
    "pt_set_sphere_uuids": (C.c_int, [C.POINTER(abi.PtHostSphere), C.c_uint32]),
    "pt_default_scene": (C.c_int, [C.POINTER(abi.PtHostSphere), C.c_uint32]),
    "pt_default_camera": (C.c_int, [C.c_uint32, C.c_uint32, C.POINTER(abi.PtCameraIn)]),
    "pt_state_create": (C.c_int, [C.POINTER(_vp), C.c_uint32, C.c_uint32]),
    "pt_state_destroy": (C.c_int, [_vp]),
    "pt_state_get": (C.c_int, [_vp, C.POINTER(abi.PtStateView)]),
    "pt_state_set_fov": (C.c_int, [_vp, C.c_double]),
    "pt_state_set_camera_angles": (C.c_int, [_vp, C.c_double, C.c_double]),
    "pt_state_set_camera_origin": (C.c_int, [_vp, C.POINTER(C.c_double)]),
    "pt_state_set_lens": (C.c_int, [_vp, C.c_double, C.c_double]),
    "pt_state_set_quality": (C.c_int, [_vp, C.c_uint32, C.c_uint32]),
    "pt_state_set_flags": (C.c_int, [_vp, C.c_int, C.c_int, C.c_float]),
    "pt_state_set_keys": (C.c_int, [_vp, C.c_uint32]),
    "pt_state_update_position": (C.c_int, [_vp, C.c_double]),
    "pt_state_update_render_globals": (C.c_int, [_vp]),
    "pt_state_resize": (C.c_int, [_vp, C.c_uint32, C.c_uint32]),
    "pt_state_should_render": (C.c_int, [_vp, C.c_int]),
    "pt_state_set_spheres": (C.c_int, [_vp, C.POINTER(abi.PtHostSphere), C.c_uint32]),
    "pt_state_spheres": (C.c_int, [_vp, C.POINTER(abi.PtSphere), C.c_uint32]),
    "pt_state_to_params": (C.c_int, [_vp, C.c_double, C.POINTER(abi.PtParams)]),
    "pt_adjusted_screen_dimensions": (C.c_int, [C.c_double, C.c_double, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "pt_center_hit": (
        C.c_int,
        [C.POINTER(abi.PtHostSphere), C.c_uint32, C.POINTER(abi.PtCameraIn), C.POINTER(abi.PtCenterHit)],
    ),
}


def _elf_dynamic_strings(path, tags):
    """The strings of an ELF64 shared object's dynamic section for the given tags (1 = DT_NEEDED, 14 = DT_SONAME)."""
    import struct

    out = []
    with open(path, "rb") as f:
        hdr = f.read(64)
        if hdr[:4] != b"\x7fELF" or hdr[4] != 2 or hdr[5] != 1:
            return out
        shoff, = struct.unpack_from("<Q", hdr, 0x28)
        shentsize, shnum = struct.unpack_from("<HH", hdr, 0x3A)
        f.seek(shoff)
        secs = [struct.unpack_from("<IIQQQQIIQQ", f.read(shentsize)) for _ in range(shnum)]
        for sec in secs:
            if sec[1] != 6:  # SHT_DYNAMIC
                continue
            strtab = secs[sec[6]]  # sh_link: its string table
            f.seek(sec[4])
            dyn = f.read(sec[5])
            for k in range(0, len(dyn) - 15, 16):
                tag, val = struct.unpack_from("<qQ", dyn, k)
                if tag in tags:
                    f.seek(strtab[4] + val)
                    out.append(f.read(256).split(b"\0", 1)[0].decode("ascii", "replace"))
    return out


def _share_torch_hip_runtime(lib_path):
    """One HIP runtime per process.  libptrace.so needs libamdhip64.so.7; PyTorch ships its own copy of that
    library.  Whichever is loaded first serves both (same SONAME) — as long as it is PyTorch's: with the system
    copy loaded first, a later `import torch` brings its own runtime as a second one, and the second runtime to
    touch the GPU finds none ("No HIP GPUs are available" from PathTracer(use_torch=True), dist.py or bench.py,
    depending on import order).  So if PyTorch is installed but not imported yet, its copy is loaded here, before
    libptrace.so, exactly as if `import torch` had come first — but ONLY when its SONAME is the one libptrace.so asks
    for (DT_NEEDED): a PyTorch built for another ROCm major (libamdhip64.so.6) could never serve libptrace, and
    preloading it would CREATE the two-runtime process this function exists to avoid.  Without PyTorch, with a
    mismatching one, or with PT_NO_TORCH_HIP_PRELOAD set, the system runtime the library was built against is used
    (and PathTracer(use_torch=True) reports the mismatch when it is asked to share buffers with torch)."""
    import importlib.util
    import sys

    if "torch" in sys.modules or os.environ.get("PT_NO_TORCH_HIP_PRELOAD"):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if not os.path.exists(path):
        return
    try:
        wanted = [n for n in _elf_dynamic_strings(lib_path, (1,)) if n.startswith("libamdhip64.so")]
        offered = _elf_dynamic_strings(path, (14,))
    except Exception:  # (struct.error on a truncated or odd ELF is none of OSError / ValueError / IndexError: no match, never an abort)
        return
    if not wanted or not offered or offered[0] != wanted[0]:
        return  # another ROCm major (or an unreadable file): leave it alone
    try:
        C.CDLL(path, mode=C.RTLD_GLOBAL)
    except OSError:
        pass  # (an unusable copy: the system runtime serves libptrace.so, and use_torch will say what is wrong)


def build_identity():
    """What the library in use was built from: sha256 over the kernel and host sources (csrc/*.hip, *.hpp, *.h, *.cpp, the
    Makefile's flags, include/ptrace*.h: name and content, sorted) and, for the record, over the shared object itself.
    profiles/summarize.py writes it into every counter record it derives from a profiled bench line and bench.py refuses
    to attach a record of another build to a later line (a stale issue fraction would look like a measurement)."""
    import hashlib

    h = hashlib.sha256()
    csrc = os.path.join(_HERE, "csrc")
    files = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith((".hip", ".hpp", ".h", ".cpp")) or f == "Makefile"]
    inc = os.path.join(os.path.dirname(_HERE), "include")
    files += [os.path.join(inc, f) for f in ("ptrace.h", "ptrace_dev.h") if os.path.exists(os.path.join(inc, f))]
    for path in files:
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    lib_path = os.environ.get("PT_LIB", LIB_PATH)
    lib_sha = None
    if os.path.exists(lib_path):
        with open(lib_path, "rb") as f:
            lib_sha = hashlib.sha256(f.read()).hexdigest()
    return {"csrc_sha256": h.hexdigest(), "lib_sha256": lib_sha, "lib": os.path.relpath(lib_path, os.path.dirname(_HERE)),
            "abi": abi.PT_ABI_VERSION}


def load():
    """Return the ctypes handle of libptrace.so with every signature declared."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("PT_LIB", LIB_PATH)  # dev: A/B another build of the same ABI
    if not os.path.exists(path):
        raise ImportError(
            "libptrace.so not found at %s — build it with `make -C ray_tracer_webgl_amd/csrc` "
            "(there is no CPU fallback)" % path
        )
    _share_torch_hip_runtime(path)
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI lost a symbol
        fn.restype = res
        fn.argtypes = args
    if lib.pt_abi_version() != abi.PT_ABI_VERSION:
        raise ImportError("libptrace.so ABI %d != expected %d" % (lib.pt_abi_version(), abi.PT_ABI_VERSION))
    _lib = lib
    return lib
