"""Row f3: image write-out of a resolved frame (the reference's "Save Image",
src/dom.rs:126-143: canvas.toDataURL -> PNG) and the accumulation checkpoint (the reference's
accumulation state is its ping-pong textures + render_count, src/state.rs:443-450).  Host-only
part here; the device round trips are in test_gpu_parity.py."""
import numpy as np

from _png import read_png
from ray_tracer_webgl_amd import image_io


def _frame(h=37, w=53, seed=5):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, (h, w, 4), dtype=np.uint8)


def test_png_round_trip_and_row_flip(tmp_path):
    f = _frame()
    path = tmp_path / "a.png"
    image_io.write_png(str(path), f)
    img = read_png(str(path))
    assert img.shape == (37, 53, 3)
    # frames have row 0 at the BOTTOM (static/shader.frag:410), files at the top
    assert np.array_equal(img, f[::-1, :, :3])
    image_io.write_png(str(path), f, flip=False)
    assert np.array_equal(read_png(str(path)), f[..., :3])


def test_ppm_round_trip(tmp_path):
    f = _frame(11, 7)
    path = tmp_path / "a.ppm"
    image_io.write_ppm(str(path), f)
    raw = open(path, "rb").read()
    head = b"P6\n7 11\n255\n"
    assert raw.startswith(head)
    body = np.frombuffer(raw[len(head):], np.uint8).reshape(11, 7, 3)
    assert np.array_equal(body, f[::-1, :, :3])


def test_float_frames_quantise_like_the_rgba8_framebuffer():
    """clamp to [0,1], round to nearest of 255 steps (src/webgl.rs:109-119); NaN -> 0"""
    f = np.array([[[-0.5, 0.0, 0.002], [0.5, 1.0, 7.0], [np.nan, np.inf, 0.9980]]], np.float32)
    q = image_io.to_rgb8(f)
    assert q.tolist() == [[[0, 0, 1], [128, 255, 255], [0, 255, 254]]]


def test_accumulation_checkpoint_round_trip(tmp_path):
    rng = np.random.default_rng(1)
    acc = rng.random((9, 13, 4), dtype=np.float32) * 100.0
    acc[..., 3] = 48.0
    path = str(tmp_path / "ck.npz")
    image_io.save_accum(path, acc, 48)
    got, spp = image_io.load_accum(path)
    assert spp == 48 and got.dtype == np.float32
    assert np.array_equal(got.view(np.uint32), acc.view(np.uint32))
