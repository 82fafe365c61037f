"""Dev tool: a BASELINE scene in the file format examples/render_bands.c and examples/render.c read
("PTSC", u32 n_spheres, u32 sizeof(PtSphere), u32 sizeof(PtParams), u32 n_passes, PtParams, PtSphere[n]), and the check of
the frame such a program wrote against the committed digest of the single-GPU frame.

    python tools/write_scene_bin.py write config2 scene.bin            (bench.py's pass shape: 64 passes of 16 spp, decorrelated pass times)
    python tools/write_scene_bin.py check config2 frame.f32            (sha256 of the fp32 frame against tests/golden/full_frame_digests.json)
"""
import ctypes as C
import hashlib
import json
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from ray_tracer_webgl_amd import abi, scenes  # noqa: E402

SHAPES = {"config2": (lambda: scenes.config2(1920, 1080, 16, 64, 50), "config2_1920x1080_64x16spp_decorrelated"),
          "config3": (lambda: scenes.config3(3840, 2160, 16, 256, 50), "config3_3840x2160_256x16spp_decorrelated"),
          "config5": (lambda: scenes.config5(1920, 1080, 16, 16, 50), "config5_1920x1080_16x16spp_decorrelated")}

mode, name, path = sys.argv[1], sys.argv[2], sys.argv[3]
make, key = SHAPES[name]
sc = make()
sc.params.time_step = abi.PT_TIME_STEP_DECORRELATED
if mode == "write":
    sph = np.ascontiguousarray(sc.spheres)
    with open(path, "wb") as f:
        f.write(b"PTSC" + struct.pack("<4I", len(sph), C.sizeof(abi.PtSphere), C.sizeof(abi.PtParams), sc.n_passes))
        f.write(bytes(sc.params))
        f.write(sph.tobytes())
    print("%s: %d spheres, %dx%d, %d passes of %d spp -> %s" % (name, len(sph), sc.params.width, sc.params.height, sc.n_passes,
                                                                  sc.params.samples_per_pixel, path))
else:
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "full_frame_digests.json"))).get(key)
    got = hashlib.sha256(np.fromfile(path, dtype=np.float32).tobytes()).hexdigest()
    ok = bool(want) and got == want["sha256"]
    print("%s: sha256 %s %s the committed single-GPU digest (%s)" % (path, got[:16], "==" if ok else "!=", key))
    raise SystemExit(0 if ok else 1)
