"""CPU-side checks of the drop-in boundary: libptrace.so loads, exports every symbol
include/ptrace.h declares, the ctypes struct mirrors match the C layout, the library refuses to
run without a GPU (no CPU fallback), and its pure-host entry points (camera derivation
src/state.rs:319-347, default scene :148-257, narrowing src/webgl.rs:225-274, pick ray
src/glsl.rs:213-239) agree with the oracle's independent restatements."""
import ctypes as C
import math
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest

from ray_tracer_webgl_amd import abi, scenes
from ray_tracer_webgl_amd._lib import LIB_PATH, SIGNATURES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ptrace.h")


DEV_HEADER = os.path.join(ROOT, "include", "ptrace_dev.h")


def declared_symbols(header=HEADER):
    text = open(header).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(lib):
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), "libptrace.so does not export %s" % s
        assert s in SIGNATURES, "no ctypes signature for %s" % s
    out = subprocess.check_output(["nm", "-D", "--defined-only", LIB_PATH], text=True)
    exported = set(re.findall(r" T (pt_[a-z0-9_]+)", out))
    assert set(syms) <= exported
    # the developer diagnostics live in their own header (outside the versioned ABI); together the two
    # headers declare EVERYTHING the library exports as pt_*
    dev = declared_symbols(DEV_HEADER)
    assert dev and set(dev) <= exported and not set(dev) & set(syms)
    assert exported == set(syms) | set(dev), sorted(exported ^ (set(syms) | set(dev)))
    # nothing but the C ABI is exported as pt_*, and no oracle symbol leaked into the product
    assert not re.search(r"\bora_", out)


def test_struct_layouts_match_header():
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "ptrace.h"
int main(void){
 printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(PtSphere), sizeof(PtParams), sizeof(PtCameraIn),
   sizeof(PtLookAtIn), sizeof(PtStats), sizeof(PtHostSphere), sizeof(PtCenterHit));
 printf("%zu %zu %zu %zu\n", offsetof(PtParams,lens_radius), offsetof(PtParams,band_rows),
   offsetof(PtSphere,albedo), offsetof(PtHostSphere,albedo));
 return 0; }'''
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(td, "t")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        l1, l2 = subprocess.check_output([exe], text=True).strip().split("\n")
    sizes = [int(x) for x in l1.split()]
    mine = [C.sizeof(t) for t in (abi.PtSphere, abi.PtParams, abi.PtCameraIn, abi.PtLookAtIn, abi.PtStats,
                                  abi.PtHostSphere, abi.PtCenterHit)]
    assert sizes == mine
    offs = [int(x) for x in l2.split()]
    assert offs == [abi.PtParams.lens_radius.offset, abi.PtParams.band_rows.offset,
                    abi.PtSphere.albedo.offset, abi.PtHostSphere.albedo.offset]


def test_no_cpu_fallback_without_gpu(lib):
    if lib.pt_device_count() > 0:
        pytest.skip("a GPU is present")
    ctx = C.c_void_p()
    rc = lib.pt_create(C.byref(ctx), 0, 64, 64)
    assert rc == abi.PT_ERR_NO_DEVICE and not ctx.value
    assert b"no HIP device" in lib.pt_last_error(None)
    from ray_tracer_webgl_amd.tracer import PathTracer, PtError

    with pytest.raises(PtError):
        PathTracer(64, 64)


def test_product_package_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "ray_tracer_webgl_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")) or f == "Makefile":
                text = open(os.path.join(dp, f), errors="ignore").read()
                # comments may MENTION the oracle; nothing may include, import, link or load it
                assert not re.search(r'#\s*include\s*[<"][^>"]*oracle', text), f
                assert not re.search(r"^\s*(import|from)\s+oracle\b", text, flags=re.M), f
                assert "libpt_oracle" not in text and "oracle/" not in text.replace("oracle/pt_oracle.c", ""), f


def test_camera_from_state_matches_oracle(lib, ora):
    L = ora.load()
    rng = np.random.default_rng(11)
    for _ in range(200):
        cam = abi.PtCameraIn()
        cam.width, cam.height = int(rng.integers(16, 4000)), int(rng.integers(16, 2200))
        cam.camera_origin = abi.d3(*rng.normal(0, 5, 3))
        cam.yaw_degrees = float(rng.uniform(-180, 180))
        cam.pitch_degrees = float(rng.uniform(-89, 89))
        cam.vup = abi.d3(0, 1, 0)
        cam.fov_radians = float(rng.uniform(0.05, 2.3))
        cam.focus_distance = float(rng.uniform(0.1, 20))
        cam.aperture = float(rng.uniform(0, 0.5))
        a, b = abi.PtParams(), abi.PtParams()
        assert lib.pt_camera_from_state(C.byref(cam), C.byref(a)) == 0
        assert L.ora_camera_from_state(C.byref(cam), C.byref(b)) == 0
        assert bytes(a) == bytes(b)


def test_camera_look_at_matches_oracle(lib, ora):
    L = ora.load()
    la = abi.PtLookAtIn()
    la.width, la.height = 1920, 1080
    la.look_from, la.look_at, la.vup = abi.d3(13, 2, 3), abi.d3(0, 0, 0), abi.d3(0, 1, 0)
    la.vfov_radians, la.focus_distance, la.aperture = math.radians(20), 10.0, 0.1
    a, b = abi.PtParams(), abi.PtParams()
    assert lib.pt_camera_look_at(C.byref(la), C.byref(a)) == 0
    assert L.ora_camera_look_at(C.byref(la), C.byref(b)) == 0
    assert bytes(a) == bytes(b)
    assert abs(a.lens_radius - 0.05) < 1e-8
    assert lib.pt_camera_look_at(None, C.byref(a)) == abi.PT_ERR_INVALID


def test_default_scene_and_narrowing(lib):
    host = (abi.PtHostSphere * 16)()
    assert lib.pt_default_scene(host, 16) == 9  # src/state.rs:148-257
    assert lib.pt_default_scene(None, 0) == 9
    assert [host[i].uuid for i in range(9)] == list(range(9))  # set_sphere_uuids
    assert host[4].radius == -0.15 and host[5].radius == -0.1  # negative radii
    assert host[3].type == abi.PT_GLASS and host[3].refraction_index == 1.5
    assert tuple(host[0].center) == (0.0, -100.5, -1.0) and host[0].radius == 100.0
    dev = (abi.PtSphere * 9)()
    assert lib.pt_narrow_spheres(host, 9, dev) == 0
    assert dev[2].center[0] == np.float32(-1.1) and dev[4].radius == np.float32(-0.15)
    sc = scenes.default_scene(320, 176, 4)
    assert len(sc.spheres) == 9 and sc.params.max_depth == 8
    cam = abi.PtCameraIn()
    assert lib.pt_default_camera(400, 225, C.byref(cam)) == 0
    assert cam.yaw_degrees == -90.0 and cam.fov_radians == math.pi / 3 and cam.focus_distance == 0.75


def test_center_hit_matches_oracle_f64(lib, ora):
    L = ora.load()
    host = (abi.PtHostSphere * 16)()
    n = lib.pt_default_scene(host, 16)
    rng = np.random.default_rng(2)
    hits = 0
    for _ in range(300):
        cam = abi.PtCameraIn()
        lib.pt_default_camera(640, 360, C.byref(cam))
        cam.yaw_degrees = float(rng.uniform(-180, 180))
        cam.pitch_degrees = float(rng.uniform(-60, 60))
        cam.camera_origin = abi.d3(*rng.uniform(-2, 2, 3))
        a, b = abi.PtCenterHit(), abi.PtCenterHit()
        ra = lib.pt_center_hit(host, n, C.byref(cam), C.byref(a))
        rb = L.ora_center_hit_f64(host, n, C.byref(cam), C.byref(b))
        assert ra == rb
        if ra == 1:
            hits += 1
            assert a.uuid == b.uuid and a.front_face == b.front_face
            assert a.t == b.t and tuple(a.hit_point) == tuple(b.hit_point) and tuple(a.normal) == tuple(b.normal)
    assert hits > 50


def test_local_rows_helpers(lib):
    for h in (1, 7, 8, 45, 1080, 2160):
        for band in (1, 4, 8):
            for world in (1, 2, 3, 8):
                tot = 0
                for r in range(world):
                    n = lib.pt_local_rows(h, band, r, world)
                    ys = abi.owned_rows(h, band, r, world)
                    assert n == abi.local_rows(h, band, r, world) == len(ys)
                    assert [lib.pt_band_row(band, r, world, l) for l in range(n)] == [int(y) for y in ys]
                    tot += n
                assert tot == h


def test_scene_generators_are_deterministic():
    a, b = scenes.cover_spheres(), scenes.cover_spheres()
    assert a.tobytes() == b.tobytes()
    assert 470 <= len(a) <= 488  # ~485 spheres
    types = set(a["type"].tolist())
    assert types == {abi.PT_DIFFUSE, abi.PT_METAL, abi.PT_GLASS}
    f = scenes.field_spheres()
    assert len(f) == 10001 and f.tobytes() == scenes.field_spheres().tobytes()
    c4 = scenes.config4()
    assert (c4.spheres["type"] == abi.PT_EMISSIVE).sum() == 1 and c4.params.background_mode == abi.PT_BG_BLACK
    assert scenes.config2().total_spp == 1024 and scenes.config3().total_spp == 4096
    assert scenes.config5().total_spp == 256 and scenes.config4().total_spp == 8192


def test_rust_binding_lists_the_render_abi():
    """bindings/rust/ptrace_sys.rs (source-only, row f4) declares every device-facing symbol of
    the header with the same argument count."""
    text = open(os.path.join(ROOT, "bindings", "rust", "ptrace_sys.rs")).read()
    header = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for name in ["pt_create", "pt_create_on_stream", "pt_grid_fit", "pt_refit_grid", "pt_destroy", "pt_resize", "pt_set_spheres", "pt_set_params", "pt_render",
                 "pt_render_passes", "pt_reserve_passes", "pt_reset_accum", "pt_synchronize", "pt_resolve",
                 "pt_resolve_rgba8", "pt_blend_rgba8", "pt_accum_ptr", "pt_bind_accum", "pt_read_accum", "pt_load_accum",
                 "pt_set_stream", "pt_set_option", "pt_tune", "pt_clear_textures", "pt_render_frame", "pt_render_frames",
                 "pt_read_canvas", "pt_read_texture", "pt_write_texture",
                 "pt_last_error", "pt_abi_version", "pt_device_count", "pt_local_rows", "pt_band_row"]:
        r = re.search(r"pub fn %s\(([^)]*)\)" % name, text)
        h = re.search(r"\b%s\s*\(([^)]*)\)" % name, header)
        assert r and h, name
        n_rust = len([a for a in r.group(1).split(",") if a.strip()])
        n_c = len([a for a in h.group(1).split(",") if a.strip() and a.strip() != "void"])
        assert n_rust == n_c, "%s: %d rust args vs %d C args" % (name, n_rust, n_c)
    # #[repr(C)] mirrors carry the same number of fields as the C structs
    def c_fields(struct):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), header, flags=re.S).group(1)
        n = 0
        for decl in body.split(";"):
            decl = decl.strip()
            if decl:
                n += len(decl.split(","))
        return n
    def rust_fields(struct):
        body = re.search(r"pub struct %s \{(.*?)\n\}" % struct, text, flags=re.S).group(1)
        return len(re.findall(r"pub \w+:", body))
    assert c_fields("PtSphere") == rust_fields("PtSphere")
    assert c_fields("PtParams") == rust_fields("PtParams")


def test_one_hip_runtime_per_process_whatever_the_import_order():
    """libptrace.so needs libamdhip64.so.7 and PyTorch ships its own copy.  Loaded in the order libptrace ->
    torch the process would hold two HIP runtimes, and the second one to touch the GPU finds none (seen on the
    GPU box as "No HIP GPUs are available" from PathTracer(use_torch=True)).  _lib.load() loads PyTorch's copy
    first when PyTorch is installed but not imported yet: one runtime mapped either way (no GPU needed to see
    which libraries the loader mapped)."""
    import subprocess
    import sys

    code = (
        "import sys\n"
        "sys.path.insert(0, %r)\n"
        "from ray_tracer_webgl_amd import _lib\n"
        "_lib.load()\n"
        "assert 'torch' not in sys.modules\n"
        "import torch\n"
        "libs = sorted({ln.split()[-1] for ln in open('/proc/self/maps') if 'libamdhip64' in ln})\n"
        "print(len(libs), libs)\n"
    ) % ROOT
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.split()[0] == "1", r.stdout
