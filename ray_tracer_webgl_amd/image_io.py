"""Image write-out for resolved frames (the reference's "Save Image" path, src/dom.rs:126-143:
canvas.toDataURL -> PNG).  Frames here have row 0 at the BOTTOM (static/shader.frag:410), files
have row 0 at the top, so rows are flipped on the way out.  Pure-Python PNG/PPM encoders: the
image has no imaging library."""
import struct
import zlib

import numpy as np


def to_rgb8(frame):
    """(H, W, >=3) float [0,1] gamma-encoded or uint8 -> (H, W, 3) uint8, clamped like the
    reference's RGBA8 framebuffer (src/webgl.rs:109-119)."""
    a = np.asarray(frame)
    if a.dtype == np.uint8:
        return np.ascontiguousarray(a[..., :3])
    v = np.nan_to_num(a[..., :3].astype(np.float32), nan=0.0, posinf=1.0, neginf=0.0)
    return np.ascontiguousarray((np.clip(v, 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8))


def write_png(path, frame, flip=True):
    rgb = to_rgb8(frame)
    if flip:
        rgb = rgb[::-1]
    h, w, _ = rgb.shape
    raw = b"".join(b"\x00" + rgb[y].tobytes() for y in range(h))

    def chunk(tag, data):
        c = struct.pack(">I", len(data)) + tag + data
        return c + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
    png += chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b"")
    with open(path, "wb") as f:
        f.write(png)


def write_ppm(path, frame, flip=True):
    rgb = to_rgb8(frame)
    if flip:
        rgb = rgb[::-1]
    h, w, _ = rgb.shape
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (w, h))
        f.write(rgb.tobytes())


def save_accum(path, accum, total_spp):
    """Checkpoint of the fp32 accumulation state (the reference's accumulation state is its
    ping-pong textures + render_count, src/state.rs:443-450)."""
    np.savez_compressed(path, accum=np.asarray(accum, dtype=np.float32), total_spp=np.uint32(total_spp))


def load_accum(path):
    z = np.load(path)
    return z["accum"], int(z["total_spp"])
