"""Extracts a few dozen SKY pixels of the reference's own screenshot of State::default
(/root/reference/images/14.png, 1280x702 RGBA8; SURVEY.md §2 "Gallery") into a small JSON fixture.

Sky pixels carry no Monte-Carlo noise and no material: their value is
sqrt(mix(white, (0.5,0.7,1.0), 0.5*(normalize(d).y+1))) for the camera ray through that pixel, so
they pin — against the reference's real output — the camera derivation (src/state.rs:98-125), the
pixel -> v_position -> st mapping and row orientation (static/shader.vert:8, shader.frag:410),
background() (:289-294) and the sqrt gamma (:380).  Only pixel VALUES are stored (data), not the
image.  Run where /root/reference exists:  python tests/golden/make_sky_fixture.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/images/14.png"


sys.path.insert(0, os.path.dirname(HERE))
from _png import read_png  # noqa: E402


def main():
    img = read_png(SRC)
    h, w, _ = img.shape
    assert (w, h) == (1280, 702)
    pts = []
    # top band of the image (pure sky) on a coarse grid, and the left edge above the horizon
    for y_top in (2, 40, 80, 120, 160):
        for x in (5, 160, 400, 640, 880, 1120, 1274):
            pts.append((x, y_top))
    for y_top in (220, 280, 340, 380):
        pts.append((8, y_top))
        pts.append((1270, y_top))
    rows = []
    for x, y_top in pts:
        r, g, b = (int(v) for v in img[y_top, x, :3])
        rows.append({"x": x, "y_from_bottom": h - 1 - y_top, "rgb": [r, g, b]})
    out = {"source": "images/14.png of the reference (1280x702)", "width": w, "height": h, "pixels": rows}
    json.dump(out, open(os.path.join(HERE, "reference_sky_pixels.json"), "w"), indent=1)
    print("wrote", len(rows), "sky pixels")

    # Silhouette mask: which pixels of the screenshot show unobstructed sky.  A pixel counts as sky
    # when it is within 2.5/255 of the analytic sky colour of the camera ray through its centre
    # (State::default camera, src/state.rs:98-125, evaluated here in numpy doubles, independently of
    # the oracle).  Row 0 of the stored mask is the BOTTOM image row.
    import math
    origin = np.array([0.0, 0.0, 1.0])
    yaw = -90.0 * math.pi / 180.0
    front = np.array([math.cos(yaw), 0.0, math.sin(yaw)])
    wv = origin - (origin + front)
    wv /= np.linalg.norm(wv)
    u = np.cross([0.0, 1.0, 0.0], wv)
    u /= np.linalg.norm(u)
    v = np.cross(wv, u)
    vh = 2.0 * math.tan(math.pi / 6.0)
    vw = vh * (w / h)
    horiz, vert = 0.75 * vw * u, 0.75 * vh * v
    llc = origin - horiz / 2 - vert / 2 - 0.75 * wv
    S = ((2 * np.arange(w) + 1) / w - 1 + 1) * 0.5
    T = ((2 * np.arange(h) + 1) / h - 1 + 1) * 0.5
    d = llc[None, None, :] + S[None, :, None] * horiz[None, None, :] + T[:, None, None] * vert[None, None, :] - origin
    t = 0.5 * (d[..., 1] / np.linalg.norm(d, axis=-1) + 1.0)
    sky = np.sqrt(np.stack([(1 - t) + 0.5 * t, (1 - t) + 0.7 * t, (1 - t) + 1.0 * t], -1)) * 255.0
    bottom_up = img[::-1, :, :3].astype(np.float64)
    mask = np.abs(bottom_up - sky).max(axis=-1) <= 2.5
    np.savez_compressed(os.path.join(HERE, "reference_sky_mask.npz"), packed=np.packbits(mask), shape=np.array(mask.shape))
    print("wrote sky mask: %.2f %% of the pixels are sky" % (100.0 * mask.mean()))

    # Mirror pixels: interior pixels of the metal spheres (fuzz 0) whose reflection goes straight to
    # the sky.  Their colour is albedo * sky(reflected direction), free of Monte-Carlo noise, so
    # they pin hit point, normal orientation (also for the NEGATIVE radii, src/state.rs:200,213)
    # and reflect() against the reference's output.  Pixels are chosen with the oracle (5x5
    # neighbourhood on the same sphere, every neighbour a 2-segment hit->sky path); only their
    # coordinates and the screenshot's RGB are stored.
    import ctypes as C
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import oracle
    from ray_tracer_webgl_amd import abi, scenes

    L = oracle.load()
    sc = scenes.default_scene(w, h, spp=1, max_depth=8)
    ptr, n, keep = abi.spheres_as_ctypes(sc.spheres)
    first = np.zeros((h, w), np.int32)
    L.ora_first_hit_map.argtypes = [C.POINTER(abi.PtSphere), C.c_uint32, C.POINTER(abi.PtParams), C.c_void_p]
    L.ora_first_hit_map(ptr, n, C.byref(sc.params), first.ctypes.data_as(C.c_void_p))

    def centre_path(x, y):
        vx, vy = np.float32((2 * x + 1) / np.float32(w)) - np.float32(1), np.float32((2 * y + 1) / np.float32(h)) - np.float32(1)
        s_, t_ = np.float32((vx + np.float32(1)) * np.float32(0.5)), np.float32((vy + np.float32(1)) * np.float32(0.5))
        seed = C.c_float(0.0)
        o, dd = (C.c_float * 3)(), (C.c_float * 3)()
        L.ora_camera_ray(C.byref(sc.params), float(s_), float(t_), C.byref(seed), o, dd)
        col, seg = (C.c_float * 3)(), C.c_uint64()
        L.ora_ray_color(ptr, n, C.byref(sc.params), o, dd, C.byref(seed), col, C.byref(seg))
        return seg.value, np.sqrt(np.array(col[:], dtype=np.float64)) * 255.0

    rng = np.random.default_rng(14)
    mirror = []
    for sid, want in ((2, 120), (4, 25), (5, 25)):
        ys, xs = np.where(first == sid)
        order = rng.permutation(len(ys))
        got = 0
        for k in order:
            y, x = int(ys[k]), int(xs[k])
            if y < 2 or x < 2 or y >= h - 2 or x >= w - 2 or not (first[y - 2:y + 3, x - 2:x + 3] == sid).all():
                continue
            if any(centre_path(x + dx, y + dy)[0] != 2 for dx in (-2, 0, 2) for dy in (-2, 0, 2)):
                continue
            r, g, b = (int(v) for v in img[h - 1 - y, x, :3])
            mirror.append({"x": x, "y_from_bottom": y, "sphere": sid, "rgb": [r, g, b]})
            got += 1
            if got == want:
                break
    json.dump({"source": "images/14.png of the reference (1280x702)", "width": w, "height": h, "pixels": mirror},
              open(os.path.join(HERE, "reference_mirror_pixels.json"), "w"), indent=1)
    print("wrote", len(mirror), "mirror pixels")


if __name__ == "__main__":
    main()
