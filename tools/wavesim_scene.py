"""Dev tool: dump a scenes.py scene (spheres + camera) as the flat binary tools/wavesim.cpp reads.

    python tools/wavesim_scene.py config2 480 270 /tmp/c2.scene

Layout: u32 n, width, height, max_depth; 19 floats camera {origin, horizontal, vertical, llc, u, v,
lens_radius}; i32 background; n x PtSphere (48 B).  Needs libptrace.so (camera derivation), no GPU.
"""
import os
import struct
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_webgl_amd import scenes  # noqa: E402

name, w, h, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
sc = scenes.CONFIGS[name](w, h)
p = sc.params
cam = list(p.camera_origin) + list(p.horizontal) + list(p.vertical) + list(p.lower_left_corner) + list(p.u) + list(p.v)
cam.append(p.lens_radius)
with open(out, "wb") as f:
    f.write(struct.pack("<4I", len(sc.spheres), w, h, p.max_depth))
    f.write(struct.pack("<19f", *cam))
    f.write(struct.pack("<i", p.background_mode))
    f.write(sc.spheres.tobytes())
print("%s: %d spheres, %dx%d, depth %d -> %s" % (name, len(sc.spheres), w, h, p.max_depth, out))
