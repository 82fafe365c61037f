"""Extracts small windows of SHADED pixels (diffuse ground, diffuse spheres, glass, ground next to
the spheres) of the reference's own screenshot of State::default (/root/reference/images/14.png,
1280x702 RGBA8; SURVEY.md §2 "Gallery") into tests/golden/reference_shaded_windows.npz.

Unlike sky and mirror pixels these depend on the Monte-Carlo half of the path (sampling, scatter,
the RNG as an estimator) AND on how the reference puts frames on the screen: one 1-spp frame per
animation tick (src/state.rs:127, src/webgl.rs:342-346 while not paused), blended with the
previous RGBA8 frame in GAMMA space by the shader's render() (static/shader.frag:387-404,
src/webgl.rs:186-204).  tests/test_reference_pins.py replays exactly that with the oracle and
compares window statistics.  Only pixel VALUES of a few windows are stored (data), not the image.
Run where /root/reference exists:  python tests/golden/make_shaded_fixture.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from _png import read_png  # noqa: E402

SRC = "/root/reference/images/14.png"
# name -> (x0, x1, y0, y1), y counted from the BOTTOM image row (static/shader.frag:410)
WINDOWS = {
    "open_ground_left": (100, 148, 60, 84),
    "open_ground_right": (1080, 1128, 70, 94),
    "ground_far": (200, 248, 250, 274),
    "ground_between_spheres": (600, 648, 240, 264),
    "ground_under_glass": (860, 908, 236, 260),
    "centre_sphere_diffuse": (616, 664, 330, 354),
    "glass_showing_sky": (860, 908, 300, 324),
    "glass_showing_ground": (850, 898, 390, 414),
    "glass_centre": (856, 904, 346, 370),
}


def main():
    img = read_png(SRC)
    h, w, _ = img.shape
    assert (w, h) == (1280, 702)
    bottom_up = img[::-1, :, :3]
    names = sorted(WINDOWS)
    boxes = np.array([WINDOWS[n] for n in names], np.int32)
    pix = np.stack([bottom_up[y0:y1, x0:x1] for (x0, x1, y0, y1) in boxes]).astype(np.uint8)
    np.savez_compressed(os.path.join(HERE, "reference_shaded_windows.npz"), names=np.array(names), boxes=boxes, pixels=pix,
                        size=np.array([w, h], np.int32))
    for n, p in zip(names, pix):
        print("%-24s mean %s  std %s" % (n, p.reshape(-1, 3).mean(0).round(1), p.reshape(-1, 3).std(0).round(1)))


if __name__ == "__main__":
    main()
