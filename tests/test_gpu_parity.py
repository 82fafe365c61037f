"""GPU parity: the HIP path (through the C ABI of include/ptrace.h) against the CPU oracle on the
same seeded inputs.  The bar is BIT-EXACT fp32 (north_star allows 1e-4 per channel; both sides
implement the same pinned arithmetic, so any difference is a bug).  Every test calls through
libptrace.so; the oracle is only the checker.

Sizes: the oracle finishes each case in seconds.  BASELINE's full-size configs are covered by
size-independent properties in test_gpu_properties.py.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

from ray_tracer_webgl_amd import abi, scenes
from ray_tracer_webgl_amd.tracer import PathTracer, PtError, render_scene

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")

PROBE_HASH, PROBE_SINCOS, PROBE_CBRT, PROBE_UNIT_SPHERE, PROBE_DIVSQRT, PROBE_BASE_HASH, PROBE_FAST_ARITH = range(7)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bit_equal(got, ref, what=""):
    g, r = bits(got), bits(ref)
    if not np.array_equal(g, r):
        bad = np.argwhere(g != r)
        diff = np.abs(np.asarray(got, dtype=np.float64) - np.asarray(ref, dtype=np.float64))
        raise AssertionError("%s: %d of %d values differ (max abs %.3g), first at %s: %r vs %r" % (
            what, len(bad), g.size, np.nanmax(diff), tuple(bad[0]),
            np.asarray(got)[tuple(bad[0])], np.asarray(ref)[tuple(bad[0])]))


@pytest.fixture(scope="module")
def pt():
    t = PathTracer(64, 64)
    yield t
    t.close()


# ----------------------------------------------------------- single PT-SPEC functions (§8a rows)
def test_probe_base_hash(pt, ora):
    L = ora.load()
    rng = np.random.default_rng(0)
    xy = rng.integers(0, 2**32, (4096, 2), dtype=np.uint64).astype(np.uint32)
    out = pt.probe(PROBE_BASE_HASH, xy.view(np.float32).ravel(), 1, len(xy)).view(np.uint32)
    ref = np.array([L.ora_base_hash(int(x), int(y)) for x, y in xy], dtype=np.uint32)
    assert np.array_equal(out, ref)


def test_probe_hash_streams(pt, ora):
    L = ora.load()
    seeds = np.concatenate([np.linspace(0, 50, 1500), np.linspace(1e3, 2e5, 500), [0.0, 4194304.0]]).astype(np.float32)
    out = pt.probe(PROBE_HASH, seeds, 9, len(seeds)).reshape(-1, 9)
    ref = np.zeros_like(out)
    for i, s0 in enumerate(seeds):
        s = C.c_float(float(s0))
        h1 = L.ora_hash1(C.byref(s))
        ref[i, 0:2] = (s.value, h1)
        o2 = (C.c_float * 2)()
        L.ora_hash2(C.byref(s), o2)
        ref[i, 2:5] = (s.value, o2[0], o2[1])
        o3 = (C.c_float * 3)()
        L.ora_hash3(C.byref(s), o3)
        ref[i, 5:9] = (s.value, o3[0], o3[1], o3[2])
    assert_bit_equal(out, ref, "hash1/2/3 + seed stepping")


def test_probe_sincos_cbrt_unit_sphere(pt, ora):
    L = ora.load()
    rng = np.random.default_rng(1)
    u = np.concatenate([rng.random(20000), [0, 0.125, 0.25, 0.375, 0.5, 0.625, 0.75, 0.875, 1.0]]).astype(np.float32)
    out = pt.probe(PROBE_SINCOS, u, 2, len(u)).reshape(-1, 2)
    ref = np.zeros_like(out)
    s, c = C.c_float(), C.c_float()
    for i, v in enumerate(u):
        L.ora_sincos2pi(float(v), C.byref(s), C.byref(c))
        ref[i] = (s.value, c.value)
    assert_bit_equal(out, ref, "sincos2pi")
    x = np.concatenate([rng.random(20000), rng.integers(0, 2**31, 4000) / 2.0**31, [0.0, 1.0, 2.0**-31]]).astype(np.float32)
    out = pt.probe(PROBE_CBRT, x, 1, len(x))
    ref = np.array([L.ora_cbrt(float(v)) for v in x], dtype=np.float32)
    assert_bit_equal(out, ref, "cbrt")
    seeds = rng.uniform(0, 5000, 8000).astype(np.float32)
    out = pt.probe(PROBE_UNIT_SPHERE, seeds, 4, len(seeds)).reshape(-1, 4)
    ref = np.zeros_like(out)
    o3 = (C.c_float * 3)()
    for i, s0 in enumerate(seeds):
        sd = C.c_float(float(s0))
        L.ora_random_in_unit_sphere(C.byref(sd), o3)
        ref[i] = (o3[0], o3[1], o3[2], sd.value)
    assert_bit_equal(out, ref, "random_in_unit_sphere")


def test_probe_ieee_div_sqrt_fma(pt):
    """/ , sqrt and fma on the device are IEEE correctly rounded (vs numpy float32 / float64)."""
    rng = np.random.default_rng(2)
    a = np.concatenate([rng.normal(0, 100, 30000), 10.0 ** rng.uniform(-30, 30, 10000), [1e-40, 3e-39, 0.0]])
    b = np.concatenate([rng.normal(0, 3, 30000), 10.0 ** rng.uniform(-30, 30, 10000), [3.0, 1e-3, 1.0]])
    a, b = a.astype(np.float32), b.astype(np.float32)
    b[b == 0] = 1.0
    out = pt.probe(PROBE_DIVSQRT, np.stack([a, b], 1).ravel(), 3, len(a)).reshape(-1, 3)
    with np.errstate(all="ignore"):
        ref_div = (a / b).astype(np.float32)
        ref_sqrt = np.sqrt(np.abs(a)).astype(np.float32)
        ref_fma = (a.astype(np.float64) * b.astype(np.float64) + a.astype(np.float64)).astype(np.float32)
    assert_bit_equal(out[:, 0], ref_div, "division")
    assert_bit_equal(out[:, 1], ref_sqrt, "sqrt")
    # double-rounding can differ from a true fma in rare halfway cases; allow those only
    mism = bits(out[:, 2]) != bits(ref_fma)
    assert mism.mean() < 1e-4


def _log_uniform(rng, n, lo_exp, hi_exp, signed=True):
    v = np.ldexp(rng.uniform(1.0, 2.0, n), rng.integers(lo_exp, hi_exp, n)).astype(np.float32)
    return v * rng.choice([-1.0, 1.0], n).astype(np.float32) if signed else v


def test_probe_unscaled_div_sqrt_match_the_operators(pt):
    """pt_kernels.hip's div_core / sqrt_core (the compiler's correctly rounded expansions without
    their range scaling) return the operators' bits on the operand ranges their call sites
    guard; sqrt_rn and hit_root fall back outside them, so they match everywhere (hit_root up to
    roots below MIN_T, which the caller only ever compares with MIN_T)."""
    rng = np.random.default_rng(77)
    n = 1 << 21
    # 1. guarded ranges: |b| in [2^-20, 2^20], |n| in [2^-103, 2^76); x in [2^-96, 2^127]
    num = _log_uniform(rng, n, -103, 76)
    den = _log_uniform(rng, n, -20, 20)
    den[: n // 4] = rng.uniform(0.25, 4.0, n // 4).astype(np.float32)  # typical |d|^2
    num[: n // 8] = rng.normal(0, 3, n // 8).astype(np.float32)
    num[np.abs(num) < 2.0 ** -103] = 1.0
    x = _log_uniform(rng, n, -96, 127, signed=False)
    x[: n // 4] = rng.uniform(0, 1, n // 4).astype(np.float32) ** 2  # typical discriminants
    x[x < 2.0 ** -96] = 2.0 ** -96
    out = pt.probe(PROBE_FAST_ARITH, np.stack([num, den, np.ones_like(num)], 1).ravel(), 10, n).reshape(-1, 10)
    assert out[:, 7].all()
    assert_bit_equal(out[:, 1], out[:, 0], "div_core vs /")
    with np.errstate(all="ignore"):
        assert_bit_equal(out[:, 0], (num / den).astype(np.float32), "/ vs numpy")
    out = pt.probe(PROBE_FAST_ARITH, np.stack([x, np.ones_like(x), np.ones_like(x)], 1).ravel(), 10, n).reshape(-1, 10)
    assert_bit_equal(out[:, 3], out[:, 2], "sqrt_core vs sqrt")
    assert_bit_equal(out[:, 8], out[:, 9], "inv_sqrt_rn vs 1 / sqrt")
    assert_bit_equal(out[:, 2], np.sqrt(x.astype(np.float64)).astype(np.float32), "sqrt vs numpy")
    # 2. sqrt_rn: any operand (tiny, denormal, zero, negative, inf, NaN)
    odd = np.concatenate([_log_uniform(rng, 200000, -149, 128), np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, 2.0 ** -96, 2.0 ** -97], np.float32)])
    out = pt.probe(PROBE_FAST_ARITH, np.stack([odd, np.ones_like(odd), np.ones_like(odd)], 1).ravel(), 10, len(odd)).reshape(-1, 10)
    both_nan = np.isnan(out[:, 4]) & np.isnan(out[:, 2])
    assert np.array_equal(bits(out[:, 4])[~both_nan], bits(out[:, 2])[~both_nan])
    both_nan = np.isnan(out[:, 8]) & np.isnan(out[:, 9])
    assert np.array_equal(bits(out[:, 8])[~both_nan], bits(out[:, 9])[~both_nan])  # inv_sqrt_rn, any operand
    # 3. hit_root(half_b, disc, a) against the literal two-root formula, operands of every size
    m = 1 << 20
    hb = _log_uniform(rng, m, -140, 120)
    disc = _log_uniform(rng, m, -140, 127, signed=False)
    a = _log_uniform(rng, m, -30, 30, signed=False)
    k = m // 2  # half of them shaped like real candidates: disc ~ hb^2 (self-intersections: far root ~ 0)
    hb[:k] = rng.normal(0, 2, k).astype(np.float32)
    disc[:k] = (hb[:k].astype(np.float64) ** 2 * (1.0 + rng.choice([0.0, 1e-7, -1e-7, 1e-3, 0.5], k))).astype(np.float32)
    a[:k] = rng.uniform(0.5, 2.0, k).astype(np.float32)
    # hit_root's precondition: disc = fma(-a, c, half_b * half_b), so a finite disc means the square did not overflow
    disc[np.abs(hb) >= 2.0 ** 63] = np.inf
    extra = np.array([[1.0, 0.0, 1.0], [-1.0, 1.0, 1.0], [1.0, 1.0, 1.0], [0.0, 0.0, 1.0], [1e30, 1e20, 1.0], [1.0, np.inf, 1.0],
                      [1.0, np.nan, 1.0], [1.0, 1.0, 0.0], [1.0, 1.0, 1e-30], [1e-40, 1e-44, 1.0]], np.float32)
    inp = np.concatenate([np.stack([hb, disc, a], 1), extra]).astype(np.float32)
    out = pt.probe(PROBE_FAST_ARITH, inp.ravel(), 10, len(inp)).reshape(-1, 10)
    fast, plain = out[:, 5], out[:, 6]
    same = (bits(fast) == bits(plain)) | (np.isnan(fast) & np.isnan(plain))
    with np.errstate(invalid="ignore"):
        below = (fast < 0.001) & (plain < 0.001)
    bad = ~(same | below)
    assert not bad.any(), (int(bad.sum()), inp[bad][:5], fast[bad][:5], plain[bad][:5])
    assert same.mean() > 0.9  # the fast form is what normally runs


# -------------------------------------------------------------------------------- full frames
def _check_scene(ora, sc, n_passes=None, window=None, passes_per_launch=None, geometry_path=None):
    n_passes = n_passes or sc.n_passes
    sc.n_passes = n_passes
    t, got = render_scene(sc, passes_per_launch=passes_per_launch, geometry_path=geometry_path)
    try:
        st = t.stats()
        ref, seg = ora.render(sc.spheres, sc.params, n_passes, window=window)
        if window is None:
            assert_bit_equal(got, ref, sc.name)
            assert st.segments == seg, "segment count %d vs oracle %d" % (st.segments, seg)
        else:
            x0, x1, y0, y1 = window
            assert_bit_equal(got[y0:y1, x0:x1], ref[y0:y1, x0:x1], sc.name + " window")
        assert st.total_spp == n_passes * sc.params.samples_per_pixel
        return t, got, ref
    finally:
        pass


def test_config1_full_frame_bit_exact(ora):
    """BASELINE config 1: 3-sphere Lambertian, 400x225, 16 spp, 8 bounces — HIP vs oracle vs golden."""
    sc = scenes.config1()
    t, got, ref = _check_scene(ora, sc)
    z = np.load(os.path.join(GOLDEN, "config1_accum.npz"))
    assert_bit_equal(got, z["accum"], "config1 vs committed golden")
    assert t.stats().segments == int(z["segments"])
    # read-out paths: fp32 resolve, RGBA8 quantisation
    assert_bit_equal(t.resolve(True), ora.resolve(ref, 16, True), "resolve gamma")
    assert_bit_equal(t.resolve(False), ora.resolve(ref, 16, False), "resolve linear")
    assert np.array_equal(t.resolve_rgba8(True), ora.resolve_rgba8(ref, 16, True))
    t.close()


def test_default_scene_golden(ora):
    """The reference's own scene (State::default: metal, glass, negative radii), 2 passes."""
    sc = scenes.default_scene(320, 176, spp=4, max_depth=8)
    sc.n_passes = 2
    t, got, ref = _check_scene(ora, sc)
    z = np.load(os.path.join(GOLDEN, "default_320x176_accum.npz"))
    assert_bit_equal(got, z["accum"], "default scene vs committed golden")
    t.close()


def test_cover_scene_small_frame(ora):
    """Config 2's scene and camera (lens, 484 spheres, depth 50) at 96x54 — golden + oracle."""
    sc = scenes.config2(96, 54, 4, 2, 50)
    t, got, ref = _check_scene(ora, sc)
    z = np.load(os.path.join(GOLDEN, "cover_96x54_accum.npz"))
    assert_bit_equal(got, z["accum"], "cover 96x54 vs committed golden")
    t.close()


def test_cover_scene_full_resolution_window(ora):
    """Config 2 at its real 1920x1080 size, one 64-spp pass; oracle checks a 48x40 window."""
    sc = scenes.config2(1920, 1080, 64, 1, 50)
    t, got, ref = _check_scene(ora, sc, window=(930, 978, 500, 540))
    assert np.all(got[..., 3] == 64.0)
    t.close()


def test_room_scene_emissive_black_background(ora):
    """Config 4 (extension): enclosed room, emissive sphere, black background, deep bounces."""
    sc = scenes.config4(96, 96, 8, 2, 50)
    t, got, ref = _check_scene(ora, sc)
    assert got[..., :3].max() > 0
    t.close()


def test_field_scene_10k_spheres_window(ora):
    """Config 5: 10 001 spheres = the whole 160 KiB LDS list, 1024-thread workgroups."""
    sc = scenes.config5(256, 144, 4, 1, 50)
    t, got, ref = _check_scene(ora, sc, window=(100, 132, 60, 84))
    t.close()


def test_ragged_sizes_and_edge_tiles(ora):
    """Width/height not multiples of the 8x8 tile; 1x1 image; single sphere; many passes."""
    for (w, h, spp, passes) in [(13, 7, 3, 2), (1, 1, 5, 3), (65, 9, 1, 4), (8, 8, 2, 1)]:
        sc = scenes.config1(w, h, spp, 6)
        sc.n_passes = passes
        t, got, ref = _check_scene(ora, sc)
        t.close()
    sc = scenes.config1(40, 24, 4, 8)
    sc.spheres = sc.spheres[:1].copy()
    t, *_ = _check_scene(ora, sc)
    t.close()
    sc = scenes.config1(40, 24, 4, 8)
    sc.spheres = sc.spheres[:0].copy()  # empty scene: every ray escapes to the sky
    t, got, ref = _check_scene(ora, sc)
    t.close()


def test_max_depth_one_and_large_time(ora):
    sc = scenes.default_scene(64, 36, spp=3, max_depth=1)
    t, *_ = _check_scene(ora, sc)
    t.close()
    sc = scenes.default_scene(64, 36, spp=3, max_depth=8)
    sc.params.time = 12345.5  # coarse seed steps
    t, *_ = _check_scene(ora, sc)
    t.close()
    sc.params.time = 4194304.0  # seed stalls: every hash call returns the same value
    t, *_ = _check_scene(ora, sc)
    t.close()


def test_nan_and_degenerate_geometry(ora):
    """Zero-radius and coincident spheres, a sphere containing the camera."""
    sc = scenes.config1(48, 27, 4, 8)
    sp = np.concatenate([sc.spheres, sc.spheres[1:2], sc.spheres[1:2]])
    sp[3]["radius"] = 0.0
    sp[4]["type"] = abi.PT_GLASS
    sp[4]["refraction_index"] = 1.5
    sp[4]["center"] = (0.0, 0.0, 1.0)
    sp[4]["radius"] = 0.5  # camera origin (0,0,1) is at its centre
    sc.spheres = sp
    t, *_ = _check_scene(ora, sc)
    t.close()


def test_candidate_queue_overflow_falls_back_to_literal_loop(ora):
    """More than six candidate spheres along one ray (nested + stacked spheres): the per-lane
    queue overflows and the kernel continues with the shader's loop verbatim."""
    sc = scenes.config1(64, 36, 4, 8)
    items = []
    for k in range(12):  # nested glass shells around the view axis, decreasing radius
        items.append(((0.0, 0.0, -1.0), 0.9 - 0.05 * k, abi.PT_GLASS if k % 2 else abi.PT_DIFFUSE, 1.5))
    for k in range(10):  # a row of spheres behind each other
        items.append(((0.3, 0.0, -1.5 - 0.7 * k), 0.3, abi.PT_METAL, 0.0))
    sp = np.zeros(len(items) + 1, dtype=abi.SPHERE_DTYPE)
    sp[0] = sc.spheres[0]
    for i, (c, r, t, ri) in enumerate(items, start=1):
        sp[i]["center"], sp[i]["radius"], sp[i]["type"] = c, r, t
        sp[i]["albedo"], sp[i]["refraction_index"], sp[i]["fuzz"] = (0.8, 0.7, 0.9), ri, 0.1
    sp["uuid"] = np.arange(len(sp))
    sc.spheres = sp
    sc.n_passes = 2
    t, *_ = _check_scene(ora, sc)
    t.close()


def test_irregular_scene_and_rays_take_the_literal_path(ora):
    """Huge / non-finite sphere data switches the conservative rejections off for the whole
    scene; a zero or NaN camera makes every ray irregular (|d|^2 == 0 -> NaN roots, which the
    shader's comparisons ACCEPT as hits).  Both must still match the oracle bit for bit, NaNs included."""
    sc = scenes.default_scene(48, 27, spp=3, max_depth=6)
    sp = np.concatenate([sc.spheres, sc.spheres[:2]])
    sp[9]["center"] = (1e20, 0.0, 0.0)
    sp[9]["radius"] = 1e19
    sp[10]["radius"] = np.float32("nan")
    sp["uuid"] = np.arange(len(sp))
    sc.spheres = sp
    t, got, ref = _check_scene(ora, sc)
    t.close()
    sc = scenes.default_scene(40, 24, spp=2, max_depth=5)
    for k in range(3):
        sc.params.horizontal[k] = 0.0
        sc.params.vertical[k] = 0.0
        sc.params.lower_left_corner[k] = sc.params.camera_origin[k]  # direction == 0 exactly
    t, got, ref = _check_scene(ora, sc)
    t.close()
    sc = scenes.default_scene(40, 24, spp=2, max_depth=5)
    sc.params.camera_origin[1] = float("nan")
    t, got, ref = _check_scene(ora, sc)
    t.close()
    sc = scenes.default_scene(40, 24, spp=2, max_depth=5)
    sc.params.camera_origin[0] = 3e15  # finite but outside the regular range
    t, got, ref = _check_scene(ora, sc)
    t.close()
    # NaN and zero-direction rays are answered without the loop (every sphere accepts a NaN root: the last
    # one wins) — also in a scene whose spheres are not finite themselves
    for kind in ("nan", "zero"):
        sc = scenes.default_scene(40, 24, spp=2, max_depth=5)
        sp = np.concatenate([sc.spheres, sc.spheres[:3]])
        sp[9]["center"] = (np.float32("inf"), 0.0, 0.0)
        sp[10]["radius"] = np.float32("nan")
        sp[11]["center"] = (np.float32("nan"), np.float32("-inf"), 1e30)
        sp["uuid"] = np.arange(len(sp))
        sc.spheres = sp
        if kind == "nan":
            sc.params.camera_origin[2] = float("nan")
        else:
            for k in range(3):
                sc.params.horizontal[k] = 0.0
                sc.params.vertical[k] = 0.0
                sc.params.lower_left_corner[k] = sc.params.camera_origin[k]
        t, got, ref = _check_scene(ora, sc)
        t.close()


def test_passes_batched_equals_separate_launches(ora):
    sc = scenes.default_scene(96, 54, spp=4, max_depth=8)
    sc.n_passes = 6
    t1, a1 = render_scene(sc)  # one launch, 6 passes
    t2, a2 = render_scene(sc, passes_per_launch=1)  # six launches
    t3, a3 = render_scene(sc, passes_per_launch=4)  # 4 + 2
    assert_bit_equal(a1, a2, "batched vs separate")
    assert_bit_equal(a1, a3, "batched vs 4+2")
    assert t1.stats().segments == t2.stats().segments == t3.stats().segments
    ref, seg = ora.render(sc.spheres, sc.params, 6)
    assert_bit_equal(a1, ref, "6 passes vs oracle")
    for t in (t1, t2, t3):
        t.close()
    # the same with pass times that are not whole numbers (PtParams.time_step / first_pass): u_time is
    # time + float(first_pass + p) * step whichever launch renders pass p
    sc = scenes.config2(96, 54, 3, 7, 12)
    sc.params.time, sc.params.time_step = 2.5, abi.PT_TIME_STEP_DECORRELATED
    ref, seg = ora.render(sc.spheres, sc.params, 7)
    for ppl in (7, 1, 3):
        t, a = render_scene(sc, passes_per_launch=ppl)
        assert_bit_equal(a, ref, "time_step, %d passes per launch" % ppl)
        assert t.stats().segments == seg
        t.close()
    one = sc.params.copy()
    one.time_step = 1.0
    other, _ = ora.render(sc.spheres, one, 7)
    assert not np.array_equal(other, ref)


def test_frame_loop_reproduces_the_shaded_windows_of_the_reference_screenshot():
    """The product, driven the way the reference drives WebGL (app.FrameLoop in reference mode:
    one 1-spp frame per tick, blended into RGBA8 ping-pong textures by pt_blend_rgba8), lands on
    the reference's own screenshot of State::default: the window means of
    tests/golden/reference_shaded_windows.npz within 1.5/255 (tests/test_reference_pins.py does
    the same with the oracle and explains why a converged linear render is 5-16/255 brighter)."""
    from ray_tracer_webgl_amd.app import FrameLoop

    z = np.load(os.path.join(GOLDEN, "reference_shaded_windows.npz"))
    w, h = (int(v) for v in z["size"])
    loop = FrameLoop(w, h, mode="reference")
    loop.state.set_flags(is_paused=False)  # 1 spp per frame (src/state.rs:127)
    for k in range(320):
        assert loop.frame(3000.0 + 16.7 * k) is True
    canvas = loop.canvas[..., :3].astype(np.float64)
    worst = 0.0
    for name, box, shot in zip(z["names"], z["boxes"], z["pixels"]):
        x0, x1, y0, y1 = (int(v) for v in box)
        d = np.abs(canvas[y0:y1, x0:x1].mean((0, 1)) - shot.astype(np.float64).mean((0, 1))).max()
        worst = max(worst, float(d))
        assert d <= 1.5, (str(name), d)
    loop.close()


def test_row_band_partition_bit_exact(ora):
    sc = scenes.config2(160, 90, 4, 2, 50)
    tf, full = render_scene(sc)
    seg_full = tf.stats().segments
    out = np.zeros_like(full)
    seg = 0
    for world, band in [(3, 8), (8, 8), (2, 4)]:
        out[:] = 0
        seg = 0
        for r in range(world):
            t, part = render_scene(sc, band=(band, r, world))
            ys = abi.owned_rows(90, band, r, world)
            assert part.shape[0] == len(ys)
            out[ys] = part
            seg += t.stats().segments
            t.close()
        assert_bit_equal(out, full, "row bands world=%d" % world)
        assert seg == seg_full
    p = sc.params.copy()
    p.band_rows, p.band_index, p.band_count = 8, 1, 3
    ref, _ = ora.render(sc.spheres, p, 2)
    t, part = render_scene(sc, band=(8, 1, 3))
    assert_bit_equal(part, ref, "band 1/3 vs oracle")
    t.close()
    tf.close()


def test_temporal_blend_rgba8(ora):
    """static/shader.frag:387-404 running mean over RGBA8 ping-pong frames."""
    sc = scenes.default_scene(64, 36, spp=4, max_depth=8)
    t, acc = render_scene(sc)
    rng = np.random.default_rng(4)
    prev = rng.integers(0, 256, (36, 64, 4), dtype=np.uint8)
    prev[::3, ::2, 3] = 0  # alpha 0 -> "no data", rendered straight
    for rc, avg, w in [(0, 1, 1.0), (1, 1, 1.0), (2, 1, 1.0), (37, 1, 0.5), (5, 0, 1.0)]:
        p = sc.params.copy()
        p.render_count, p.should_average, p.last_frame_weight = rc, avg, w
        t.set_params(p)
        got = t.blend_rgba8(prev)
        ref = ora.blend_rgba8(acc, 4, p, prev)
        assert np.array_equal(got, ref), (rc, avg, w)
    t.close()


def test_temporal_blend_rgba8_at_the_edges_of_its_fast_forms(ora):
    """The blend kernels take sqrt and the three kinds of division through the unscaled forms of pt_arith.hpp where those
    are the compiler's own sequences, and through the plain operators elsewhere (pt_kernels.hip blend_texel).  Operands on
    both sides of every guard: colours of 0, below 2^-96 (subnormal too), huge, infinite, NaN and negative; every byte
    value in every channel against alpha 0 / 1 / 255; weights of 1e-30 (numerators below 2^-103), 1e30, 0; render counts
    whose total leaves [2^-20, 2^20).  Bit for bit the oracle's statement-by-statement blend."""
    sc = scenes.default_scene(64, 36, spp=4, max_depth=8)
    t, acc0 = render_scene(sc)
    rng = np.random.default_rng(44)
    acc = np.array(acc0, np.float32)
    h, w = acc.shape[:2]
    special = np.array([0.0, 1e-45, 1e-40, 2.0 ** -100, 2.0 ** -96 * 4 * 0.999, 2.0 ** -96 * 4, 2.0 ** -94, 1e-20, 1e-10, 0.5, 4.0, 7.99, 1e10,
                        3e38, np.inf, np.nan, -0.0, -1e-30, -1.0], np.float32)
    pick = rng.random((h, w, 3)) < 0.5
    acc[..., :3] = np.where(pick, rng.choice(special, (h, w, 3)), acc[..., :3])
    acc[..., 3] = 4.0
    prev = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    prev[:, :16, 0] = np.arange(h * 16).reshape(h, 16) % 256   # every byte value, in every channel
    prev[:, 16:32, 1] = (np.arange(h * 16).reshape(h, 16) + 97) % 256
    prev[:, 32:48, 2] = (np.arange(h * 16).reshape(h, 16) + 191) % 256
    prev[..., 3] = rng.choice(np.array([0, 1, 128, 255], np.uint8), (h, w))
    t.load_accum(acc)
    for rc, avg, wt in [(0, 1, 1.0), (1, 1, 1.0), (2, 1, 1.0), (37, 1, 0.5), (5, 0, 1.0), (3, 1, 1e-30), (3, 1, 1e-38), (2, 1, 1e30), (2, 1, 0.0),
                        (1 << 20, 1, 1.0), ((1 << 20) - 1, 1, 0.5), (2_000_000_000, 1, 1.0), (2, 1, 3e38), (7, 1, float("inf")), (7, 1, float("nan")),
                        (9, 1, -9.0), (9, 1, -1.0)]:
        p = sc.params.copy()
        p.render_count, p.should_average, p.last_frame_weight = rc, avg, wt
        t.set_params(p)
        got = t.blend_rgba8(prev)
        ref = ora.blend_rgba8(acc, 4, p, prev)
        assert np.array_equal(got, ref), (rc, avg, wt, int((got != ref).sum()), np.argwhere(got != ref)[:4])
    t.close()


def test_torch_owned_accumulation_and_stream(ora):
    import torch

    sc = scenes.config1(64, 40, 4, 8)
    sc.n_passes = 2
    t, acc = render_scene(sc, use_torch=True)
    ref, _ = ora.render(sc.spheres, sc.params, 2)
    assert_bit_equal(acc, ref, "torch-bound accumulation")
    assert t.accum_tensor.is_cuda and tuple(t.accum_tensor.shape) == (40, 64, 4)
    torch.cuda.synchronize()
    t.close()


def test_torch_work_is_ordered_after_the_render(ora):
    """use_torch=True: the kernels must run on the stream torch orders its own work on — torch's
    default stream is the NULL stream, which the ABI spells PT_STREAM_LEGACY — so that a .cpu() or a
    collective issued right after render_passes, WITHOUT a device-wide synchronise, sees the finished
    frame.  (Round 3's digest check in bench.py found the multi-rank gather reading an empty buffer.)"""
    sc = scenes.config2(320, 180, 8, 6, 50)
    p = sc.params.copy()
    pt = PathTracer(p.width, p.height, use_torch=True)
    pt.set_spheres(sc.spheres)
    pt.set_params(p)
    pt.reserve_passes(6)
    pt.set_geometry_path(abi.PT_GEOM_SCALAR)  # the slow path: a long launch
    pt.render_passes(6)
    got = pt.accum_tensor.cpu().numpy()       # stream-ordered copy, no pt.synchronize()
    ref, _ = ora.render(sc.spheres, p, 6, window=(100, 132, 60, 76))
    assert_bit_equal(got[60:76, 100:132], ref[60:76, 100:132], "torch copy right after the launch")
    assert (got[..., 3] == 48.0).all()
    pt.close()


def test_error_behaviour(pt):
    lib = pt.lib
    assert lib.pt_render(pt._ctx) == abi.PT_ERR_NOT_READY
    assert b"pt_set_spheres" in lib.pt_last_error(pt._ctx)
    sc = scenes.config1(64, 64, 2, 4)
    pt.set_spheres(sc.spheres)
    bad = sc.params.copy()
    bad.samples_per_pixel = 0
    assert lib.pt_set_params(pt._ctx, C.byref(bad)) == abi.PT_ERR_INVALID
    bad = sc.params.copy()
    bad.width = 32
    assert lib.pt_set_params(pt._ctx, C.byref(bad)) == abi.PT_ERR_INVALID
    pt.set_params(sc.params)
    assert lib.pt_render_passes(pt._ctx, 5) == abi.PT_ERR_CAPACITY  # nothing reserved beyond 1
    out = np.zeros((64, 64, 4), np.float32)
    assert lib.pt_resolve(pt._ctx, out.ctypes.data_as(C.c_void_p), 1) == abi.PT_ERR_NOT_READY
    big = np.zeros(65529, dtype=abi.SPHERE_DTYPE)
    ptr, n, keep = abi.spheres_as_ctypes(big)
    assert lib.pt_set_spheres(pt._ctx, ptr, n) == abi.PT_ERR_CAPACITY
    ctx = C.c_void_p()
    assert lib.pt_create(C.byref(ctx), 99, 8, 8) == abi.PT_ERR_NO_DEVICE
    assert lib.pt_create(C.byref(ctx), 0, 0, 8) == abi.PT_ERR_INVALID
    with pytest.raises(PtError):
        PathTracer(0, 0)
    pt.set_spheres(sc.spheres)
    pt.render()
    pt.reset()
    assert pt.stats().segments == 0 and pt.stats().total_spp == 0


def test_refused_repartition_leaves_the_context_as_it_was(ora):
    """pt_set_params validates before it commits: a row partition the caller-bound accumulation
    buffer cannot hold is refused and the previous partition keeps rendering correctly."""
    import torch

    sc = scenes.config1(64, 48, 2, 4)
    p = sc.params.copy()
    p.band_rows, p.band_index, p.band_count = 8, 0, 3   # rows 0-7, 24-31: 16 rows
    t = PathTracer(64, 48)
    t.set_spheres(sc.spheres)
    t.set_params(p)
    assert t.local_rows == 16
    buf = torch.zeros((16, 64, 4), dtype=torch.float32, device="cuda")
    t._check(t.lib.pt_bind_accum(t._ctx, C.c_void_p(buf.data_ptr()), buf.numel() * 4))
    full = sc.params.copy()                              # all 48 rows do not fit the bound buffer
    assert t.lib.pt_set_params(t._ctx, C.byref(full)) == abi.PT_ERR_CAPACITY
    assert t.stats().local_rows == 16
    t.render()                                           # still the 16-row band, inside the buffer
    t.synchronize()
    torch.cuda.synchronize()
    ref, _ = ora.render(sc.spheres, p, 1)
    assert_bit_equal(buf.cpu().numpy(), ref, "band render after a refused repartition")
    t._check(t.lib.pt_bind_accum(t._ctx, None, 0))
    t.close()


def test_save_image_png_matches_the_rgba8_read_out(ora, tmp_path):
    """Row f3: the reference's "Save Image" (src/dom.rs:126-143) — the resolved RGBA8 frame as a
    PNG, file row 0 = top of the image = the LAST frame row (static/shader.frag:410)."""
    from _png import read_png
    from ray_tracer_webgl_amd import image_io

    sc = scenes.default_scene(96, 54, spp=4, max_depth=8)
    t, acc = render_scene(sc)
    rgba = t.resolve_rgba8()
    assert np.array_equal(rgba, ora.resolve_rgba8(acc, 4))
    path = str(tmp_path / "frame.png")
    image_io.write_png(path, rgba)
    img = read_png(path)
    assert img.shape == (54, 96, 3)
    assert np.array_equal(img, rgba[::-1, :, :3])
    # the sky is at the top of the file: bluer than the ground rows at the bottom
    assert img[0, :, 2].mean() > img[-1, :, 2].mean()
    # float read-out goes through the same quantisation
    image_io.write_png(path, t.resolve())
    assert np.array_equal(read_png(path), rgba[::-1, :, :3])
    t.close()


@pytest.mark.parametrize("path", [abi.PT_GEOM_LDS, abi.PT_GEOM_BVH])
def test_checkpoint_and_resume_is_bit_exact(ora, tmp_path, path):
    """Row f3: render 4 passes, checkpoint, resume in a NEW context, render 4 more == 8
    uninterrupted passes, bit for bit, sample count included (src/state.rs:443-450 semantics:
    the accumulation state is the buffer + its frame count)."""
    from ray_tracer_webgl_amd import image_io

    sc = scenes.config2(96, 54, 2, 8, 12)
    t8, full = render_scene(sc, geometry_path=path)
    sc4 = scenes.config2(96, 54, 2, 4, 12)
    t4, half = render_scene(sc4, geometry_path=path)
    ck = str(tmp_path / "ck.npz")
    image_io.save_accum(ck, half, t4.stats().total_spp)
    t4.close()
    acc, spp = image_io.load_accum(ck)
    assert spp == 8
    t = PathTracer(96, 54)
    t.set_geometry_path(path)
    t.set_spheres(sc.spheres)
    q = sc.params.copy()
    q.time = 4.0                                     # passes 4..7
    t.set_params(q)
    t.reserve_passes(4)
    t.load_accum(acc)
    assert t.stats().total_spp == 8
    t.render_passes(4)
    got = t.accum()
    assert_bit_equal(got, full, "resumed vs uninterrupted")
    assert t.stats().total_spp == 16
    assert_bit_equal(t.resolve(), t8.resolve(), "resolved frames")
    ref, _ = ora.render(sc.spheres, sc.params, 8)
    assert_bit_equal(got, ref, "resumed vs oracle")
    with pytest.raises(ValueError):
        t.load_accum(acc[:10])
    t.close()
    t8.close()


def test_resize_and_reuse(ora):
    sc = scenes.config1(48, 32, 2, 8)
    t = PathTracer(48, 32)
    t.set_spheres(sc.spheres)
    t.set_params(sc.params)
    t.render()
    a = t.accum()
    assert t.lib.pt_resize(t._ctx, 80, 24) == 0
    t.width, t.height = 80, 24
    sc2 = scenes.config1(80, 24, 2, 8)
    t.set_params(sc2.params)
    t.render()
    b = t.accum()
    ref_a, _ = ora.render(sc.spheres, sc.params, 1)
    ref_b, _ = ora.render(sc2.spheres, sc2.params, 1)
    assert_bit_equal(a, ref_a, "before resize")
    assert_bit_equal(b, ref_b, "after resize")
    t.close()


def test_frame_loop_reference_mode_matches_oracle_simulation(ora):
    """The reference's rAF closure (src/lib.rs:65-104) driven headless: each frame is one fresh
    pass at u_time = now, blended into RGBA8 ping-pong textures by the shader's render() rule.
    The oracle simulates the same sequence (render pass -> blend with the other texture)."""
    from ray_tracer_webgl_amd.app import FrameLoop

    w, h = 96, 54
    loop = FrameLoop(w, h, mode="reference")
    loop.state.set_flags(is_paused=False)      # unpaused: 1 spp per frame, renders every tick
    loop.state.set_quality(2, 8)
    tex = [np.zeros((h, w, 4), np.uint8), np.zeros((h, w, 4), np.uint8)]
    spheres = loop.state.spheres()
    for k, now in enumerate([16.0, 33.0, 50.0, 66.5, 83.0]):
        if k == 3:
            loop.state.set_camera_angles(-80.0, 5.0)   # camera change: render_count restarts
        assert loop.frame(now) is True
        v = loop.state.view()
        p = loop.state.to_params(now)
        acc, _ = ora.render(spheres, p, 1)
        expect = ora.blend_rgba8(acc, p.samples_per_pixel, p, tex[(v.even_odd_count + 1) % 2])
        assert np.array_equal(loop.canvas, expect), "frame %d" % k
        tex[v.even_odd_count % 2] = expect
    assert loop.state.view().render_count == 2 and loop.frames_rendered == 5
    loop.close()
    # linear mode: fp32 accumulation, restart on camera change
    loop = FrameLoop(w, h, mode="linear")
    loop.state.set_flags(is_paused=False)
    loop.state.set_quality(3, 8)
    for now in (1.0, 2.0, 3.0):
        loop.frame(now)
    assert loop.tracer.stats().total_spp == 9
    loop.state.set_fov(1.2)
    loop.frame(4.0)
    assert loop.tracer.stats().total_spp == 3
    p = loop.state.to_params(4.0)
    acc, _ = ora.render(spheres, p, 1)
    assert np.array_equal(loop.canvas, ora.resolve_rgba8(acc, 3, True))
    loop.close()


def test_frames_replayed_from_one_graph_match_single_ticks(ora):
    """pt_render_frames: captured frames replayed with u_time, render_count and the even/odd texture choice
    counted on the device — in groups of 16 and 4 (ONE trace launch renders a group's frames as its passes,
    their blends follow in order) and one by one for the remainder — must draw what n single ticks with
    host-made uniforms draw (pt_render_frame), which in turn is the oracle's simulation of
    src/lib.rs:65-104 + src/webgl.rs:180-205.  Interval and start time are exact in fp32, so the
    device's time + float(k) * interval is the host's float(now_k).  n = 19: a group of 16 and three singles; then 9 = 4 + 4 + 1 and 10 = 4 + 4 + 1 + 1."""
    from ray_tracer_webgl_amd.app import FrameLoop

    w, h, n = 96, 54, 19
    a = FrameLoop(w, h, mode="reference")
    a.state.set_flags(is_paused=False)
    a.state.set_quality(2, 8)
    spheres = a.state.spheres()
    tex = [np.zeros((h, w, 4), np.uint8), np.zeros((h, w, 4), np.uint8)]
    for k in range(n):
        now = 100.0 + 16.5 * k
        assert a.frame(now) is True
        v, p = a.state.view(), a.state.to_params(now)
        acc, _ = ora.render(spheres, p, 1)
        expect = ora.blend_rgba8(acc, p.samples_per_pixel, p, tex[(v.even_odd_count + 1) % 2])
        tex[v.even_odd_count % 2] = expect
        assert np.array_equal(a.canvas, expect), "tick %d" % k
    b = FrameLoop(w, h, mode="reference")
    b.state.set_flags(is_paused=False)
    b.state.set_quality(2, 8)
    assert b.frames(n, 100.0, 16.5) == n
    assert np.array_equal(b.canvas, a.canvas)
    ta, tb = a.textures, b.textures
    assert np.array_equal(ta[0], tb[0]) and np.array_equal(ta[1], tb[1])
    va, vb = a.state.view(), b.state.view()
    assert (va.render_count, va.even_odd_count) == (vb.render_count, vb.even_odd_count) == (n, n)
    assert b.tracer.stats().segments == a.tracer.stats().segments
    # a second series continues where the first stopped (same graphs, re-armed counter) ...
    assert b.frames(9, 100.0 + 16.5 * n, 16.5) == 9
    for k in range(n, n + 9):
        assert a.frame(100.0 + 16.5 * k) is True
    assert np.array_equal(b.canvas, a.canvas)
    tb = b.textures
    assert np.array_equal(a.textures[0], tb[0]) and np.array_equal(a.textures[1], tb[1])
    # ... and a camera change between series re-captures with the new uniforms
    for loop in (a, b):
        loop.state.set_camera_angles(-75.0, 4.0)
    assert b.frames(10, 500.0, 16.5) == 10
    for k in range(10):
        assert a.frame(500.0 + 16.5 * k) is True
    assert np.array_equal(b.canvas, a.canvas) and b.state.view().render_count == 10
    a.close()
    b.close()


def test_frames_equal_single_ticks_without_averaging_and_with_a_key_held():
    """FrameLoop.frames(n) is n calls of frame() by contract.  With should_average off only the first tick
    draws (update_render_globals clears should_render, src/state.rs:443-447); with a movement key held
    every tick moves the camera (update_position, src/state.rs:411-441) and has its own uniforms.  Neither
    series may be replayed from one graph: frames() issues them tick by tick, and state, canvas and the
    number of frames drawn equal the single ticks'."""
    from ray_tracer_webgl_amd.app import FrameLoop

    w, h, n = 64, 36, 5
    for setup in ("no_average", "key_held"):
        loops = [FrameLoop(w, h, mode="reference"), FrameLoop(w, h, mode="reference")]
        for lp in loops:
            lp.state.set_quality(2, 6)
            if setup == "no_average":
                lp.state.set_flags(is_paused=False, should_average=False)
            else:
                lp.state.set_flags(is_paused=False)
                lp.state.set_keys(abi.KEY_W | abi.KEY_A)
        a, b = loops
        drawn_a = sum(1 for k in range(n) if a.frame(200.0 + 16.5 * k))
        drawn_b = b.frames(n, 200.0, 16.5)
        assert drawn_a == drawn_b == (1 if setup == "no_average" else n), (setup, drawn_a, drawn_b)
        va, vb = a.state.view(), b.state.view()
        assert (va.render_count, va.even_odd_count, va.should_render) == (vb.render_count, vb.even_odd_count, vb.should_render)
        assert tuple(va.camera_origin) == tuple(vb.camera_origin)
        assert np.array_equal(a.canvas, b.canvas), setup
        assert a.frames_rendered == b.frames_rendered
        a.close()
        b.close()


def test_resize_clears_the_frame_textures_growing_or_not(ora):
    """update_render_dimensions_to_match_window re-specifies both textures as empty on every resize
    (src/state.rs:382-396).  A frame with render_count > 1 drawn after a SHRINKING (or same-size) resize must
    therefore find alpha 0 = "no data" (static/shader.frag:391) and draw the new frame unblended — as a fresh
    context of that size does — not average with texels of the old layout."""
    sc = scenes.default_scene(96, 54, spp=2, max_depth=6)
    small = scenes.default_scene(64, 36, spp=2, max_depth=6)
    t = PathTracer(96, 54)
    t.set_spheres(sc.spheres)
    p = sc.params.copy()
    p.render_count, p.should_average = 5, 1
    t.set_params(p)
    t.render_frame(0)
    t.render_frame(1)
    for (w, h, prm) in ((64, 36, small.params), (64, 36, small.params)):   # shrink, then the same size again
        assert t.lib.pt_resize(t._ctx, w, h) == abi.PT_OK
        t.width, t.height = w, h
        q = prm.copy()
        q.render_count, q.should_average = 5, 1
        t.set_params(q)
        t.render_frame(2)
        got = t.read_canvas()
        fresh = PathTracer(w, h)
        fresh.set_spheres(sc.spheres)
        fresh.set_params(q)
        fresh.render_frame(2)
        assert np.array_equal(got, fresh.read_canvas())
        acc, _ = ora.render(sc.spheres, q, 1)
        assert np.array_equal(got, ora.blend_rgba8(acc, q.samples_per_pixel, q, np.zeros((h, w, 4), np.uint8)))
        fresh.close()
    t.close()


def test_frames_on_the_legacy_stream_are_refused_with_a_reason():
    """A context bound to torch's default stream runs on hipStreamLegacy (PT_STREAM_LEGACY), which HIP does
    not capture: pt_render_frames says so (PT_ERR_INVALID + message) instead of failing inside the capture;
    single ticks (pt_render_frame) work there."""
    sc = scenes.default_scene(48, 27, spp=1, max_depth=4)
    t = PathTracer(48, 27, use_torch=True)
    t.set_spheres(sc.spheres)
    t.set_params(sc.params)
    rc = t.lib.pt_render_frames(t._ctx, 0, 100000, 2)
    assert rc == abi.PT_ERR_INVALID and b"legacy default stream" in t.lib.pt_last_error(t._ctx)
    t.render_frame(0)
    t.synchronize()
    assert t.read_canvas()[..., 3].min() == 255
    t.close()


def test_a_frame_after_a_captured_but_unreplayed_launch_finds_every_tile(ora):
    """pt_render_passes captured into a caller's hipGraph enqueues its tile-order kernel INTO THE GRAPH: until
    the first replay nothing has written the order.  A frame issued directly in between (frames never run the
    order kernel themselves) must still visit every tile exactly once — the order array holds the identity from
    its allocation, and a captured order kernel does not count as having run."""
    import torch

    sc = scenes.default_scene(96, 54, spp=1, max_depth=6)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        t = PathTracer(96, 54, use_torch=True)
        t.set_spheres(sc.spheres)
        t.set_params(sc.params)
        t.reserve_passes(2)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            t.render_passes(2)                       # first launch of this context: captured, never run
        t.render_frame(0)                            # directly, before any replay
        torch.cuda.current_stream().synchronize()
    acc, _ = ora.render(sc.spheres, sc.params, 1)
    expect = ora.blend_rgba8(acc, sc.params.samples_per_pixel, sc.params, np.zeros((54, 96, 4), np.uint8))
    assert np.array_equal(t.read_canvas(), expect)
    t.close()


def test_frames_while_the_autotuner_is_still_measuring(ora):
    """pt_render_frames plans its launch (path choice, occupancy, the autotuner's event queries) BEFORE it
    begins capturing, so a series issued while PT_GEOM_AUTO's trial launches are still in flight captures
    cleanly, and equals the single ticks."""
    from ray_tracer_webgl_amd.app import FrameLoop

    w, h, n = 96, 54, 4
    a, b = FrameLoop(w, h, mode="reference"), FrameLoop(w, h, mode="reference")
    for lp in (a, b):
        lp.state.set_flags(is_paused=False)
        lp.state.set_quality(1, 6)
    p = b.state.to_params(50.0)
    b.tracer.set_params(p)
    b.tracer.reserve_passes(4)
    b.tracer.render_passes(4)    # cold launch + the first trial of PT_GEOM_AUTO, not waited for
    b.tracer.render_passes(4)
    assert b.frames(n, 100.0, 16.5) == n
    for k in range(n):
        assert a.frame(100.0 + 16.5 * k)
    assert np.array_equal(a.canvas, b.canvas)
    # the same uniforms again: the cached graph is reused (the plan it bakes in is unchanged), same frames
    b.tracer.clear_textures(); a.tracer.clear_textures()
    a.close()
    b.close()


@pytest.mark.parametrize("should_average", [1, 0])
def test_a_group_of_frames_equals_its_single_ticks_at_the_abi(should_average):
    """pt_render_frames at the C ABI, below FrameLoop: nine frames (two groups of four — each one trace launch of four
    passes and ONE blend kernel that runs each pixel's four blends in registers — and one single frame)
    against nine pt_render_frame calls whose uniforms the host steps itself: time + k * interval (exact in
    fp32), render_count clamped at max_render_count on the way, even_odd_count + k.  With averaging the
    canvas and BOTH textures must agree (the group writes only its last two frames' textures: the others
    would have been overwritten); without it no texture changes and the canvas shows the last frame unblended."""
    sc = scenes.default_scene(96, 54, spp=1, max_depth=6)
    n, t0, dt, rc0, rc_max, e0 = 9, 200.0, 16.5, 3, 7, 5
    seed_tex = np.random.default_rng(5).integers(0, 256, (2, 54, 96, 4), dtype=np.uint8)
    seed_tex[..., 3] = 255
    out = []
    for grouped in (True, False):
        t = PathTracer(96, 54)
        t.set_spheres(sc.spheres)
        t.write_texture(0, seed_tex[0])
        t.write_texture(1, seed_tex[1])
        p = sc.params.copy()
        p.should_average, p.last_frame_weight, p.render_count = should_average, 1.0, rc0
        p.time, p.time_step, p.first_pass = t0, dt, 0
        if grouped:
            t.set_params(p)
            t.render_frames(e0, rc_max, n)
        else:
            for k in range(n):
                q = p.copy()
                q.time, q.render_count = t0 + dt * k, min(rc0 + k, rc_max)
                t.set_params(q)
                t.render_frame(e0 + k)
        out.append((t.read_canvas(), t.read_texture(0), t.read_texture(1), t.stats().segments))
        t.close()
    (ca, a0, a1, sa), (cb, b0, b1, sb) = out
    assert sa == sb
    assert np.array_equal(ca, cb), "canvas"
    assert np.array_equal(a0, b0) and np.array_equal(a1, b1), "textures"
    if not should_average:
        assert np.array_equal(a0, seed_tex[0]) and np.array_equal(a1, seed_tex[1])


def test_frame_entry_points_refuse_what_they_cannot_do():
    pt = PathTracer(16, 16)
    assert pt.lib.pt_render_frame(pt._ctx, 0) == abi.PT_ERR_NOT_READY  # no scene / uniforms yet
    assert pt.lib.pt_render_frames(pt._ctx, 0, 100000, 2) == abi.PT_ERR_NOT_READY
    assert pt.lib.pt_read_canvas(pt._ctx, None) == abi.PT_ERR_INVALID
    assert pt.lib.pt_read_texture(pt._ctx, 2, None) == abi.PT_ERR_INVALID
    assert pt.lib.pt_clear_textures(None) == abi.PT_ERR_INVALID
    pt.close()


def test_render_is_hip_graph_capturable(ora):
    """pt_render_passes neither allocates nor synchronises (after pt_reserve_passes), so a
    frame can be captured into a hipGraph and replayed; each replay adds the same passes."""
    import torch

    sc = scenes.default_scene(96, 54, spp=4, max_depth=8)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        t = PathTracer(96, 54, use_torch=True)   # binds the side stream as its launch stream
        t.set_spheres(sc.spheres)
        t.set_params(sc.params)
        t.reserve_passes(2)
        t.render_passes(2)                        # warm-up outside the capture
        torch.cuda.current_stream().synchronize()
        t.reset()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            t.render_passes(2)
    g.replay()
    torch.cuda.synchronize()
    ref, _ = ora.render(sc.spheres, sc.params, 2)
    got = t.accum_tensor.cpu().numpy()
    assert_bit_equal(got[..., :3], ref[..., :3], "graph replay 1")
    g.replay()
    torch.cuda.synchronize()
    got2 = t.accum_tensor.cpu().numpy()
    # the fold is sequential fp32: ((p0 + p1) + p0) + p1
    p0, _ = ora.render(sc.spheres, sc.params, 1)
    q = sc.params.copy()
    q.time = 1.0
    p1, _ = ora.render(sc.spheres, q, 1)
    expect = ((p0 + p1) + p0) + p1
    assert_bit_equal(got2[..., :3], expect[..., :3], "graph replay 2 adds the same passes again")
    # read-out after replays the host never saw: the divisor comes from the device-side sample
    # count (accum.w = 4 launches' worth of 4 spp), not from host bookkeeping
    assert np.all(got2[..., 3] == 16.0)
    assert_bit_equal(t.resolve(gamma=True), ora.resolve(expect, 16, gamma=True), "resolve after two replays")
    assert np.array_equal(t.resolve_rgba8(), ora.resolve_rgba8(expect, 16))
    assert t.stats().total_spp == 16
    t.close()


def test_resolve_before_any_replay_of_a_captured_launch_is_empty_not_wrong(ora):
    """A capture that was never replayed has rendered nothing: read-out gives zeros (no host-side
    sample count can claim otherwise)."""
    import torch

    sc = scenes.default_scene(64, 36, spp=2, max_depth=4)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        t = PathTracer(64, 36, use_torch=True)
        t.set_spheres(sc.spheres)
        t.set_params(sc.params)
        t.render_passes(1)
        torch.cuda.current_stream().synchronize()
        t.reset()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            t.render_passes(1)
    torch.cuda.synchronize()
    out = t.resolve()
    assert np.all(out[..., :3] == 0.0)
    assert t.stats().total_spp == 0
    g.replay()
    torch.cuda.synchronize()
    ref, _ = ora.render(sc.spheres, sc.params, 1)
    assert_bit_equal(t.resolve(), ora.resolve(ref, 2), "resolve after the first replay")
    t.close()


def test_hierarchy_walk_is_hip_graph_capturable(ora):
    """The same with the cover scene and the hierarchy walk forced: the captured launch carries
    the hierarchy kernel (dynamic LDS, occupancy-driven workgroup size) and replays bit-exactly."""
    import torch

    sc = scenes.config2(96, 54, 2, 2, 50)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        t = PathTracer(96, 54, use_torch=True)
        t.set_spheres(sc.spheres)
        t.set_params(sc.params)
        t.set_geometry_path(abi.PT_GEOM_BVH)
        t.reserve_passes(2)
        t.render_passes(2)
        torch.cuda.current_stream().synchronize()
        t.reset()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            t.render_passes(2)
    g.replay()
    torch.cuda.synchronize()
    ref, _ = ora.render(sc.spheres, sc.params, 2)
    assert_bit_equal(t.accum_tensor.cpu().numpy()[..., :3], ref[..., :3], "hierarchy graph replay")
    assert t.stats().geometry_path == abi.PT_GEOM_BVH
    t.close()


def test_sphere_list_beyond_lds_capacity(ora):
    """n > 10 232 spheres do not fit the 160 KiB LDS: the scalar-load walk of the padded
    global copy (pt_trace_kernel_scalar) is used instead.  Window-checked against the oracle."""
    sc = scenes.config5(160, 90, 2, 1, 20, n=12000)
    assert len(sc.spheres) == 12001
    t, got, ref = _check_scene(ora, sc, window=(60, 84, 30, 46), geometry_path=abi.PT_GEOM_LDS)  # falls back
    assert t.stats().geometry_path == abi.PT_GEOM_SCALAR
    t.close()
    # left to itself PT_GEOM_AUTO does not even try the list walks on a structured scene this size
    t2, got2 = render_scene(sc)
    assert t2.stats().geometry_path in (abi.PT_GEOM_BVH, abi.PT_GEOM_GRID)
    assert_bit_equal(got2, got, "culling structure vs the list walk, whole frame")
    t2.close()


@pytest.mark.parametrize("path", [abi.PT_GEOM_LDS, abi.PT_GEOM_SCALAR, abi.PT_GEOM_BVH, abi.PT_GEOM_GRID, abi.PT_GEOM_SMALL,
                                  abi.PT_GEOM_AUTO])
def test_geometry_paths_are_bit_identical(ora, path):
    """LDS walk, scalar-load walk, hierarchy walk, grid walk and the autotuned choice give the same
    bits (and the same segment counts) on scenes that exercise every phase of hit_world."""
    for sc in (scenes.default_scene(96, 54, spp=4, max_depth=8), scenes.config2(96, 54, 4, 2, 50),
               scenes.config4(48, 48, 4, 2, 50)):
        sc.n_passes = 6
        t, got = render_scene(sc, passes_per_launch=1, geometry_path=path)  # 6 launches: AUTO tries them all
        ref, seg = ora.render(sc.spheres, sc.params, 6)
        assert_bit_equal(got, ref, "%s path %d" % (sc.name, path))
        st = t.stats()
        assert st.segments == seg
        has_tree = len(sc.spheres) >= 16
        assert (st.bvh_nodes > 0) == has_tree and (st.grid_entries > 0) == has_tree
        small = len(sc.spheres) <= 16
        if path in (abi.PT_GEOM_BVH, abi.PT_GEOM_GRID):
            # scenes without a culling structure (fewer than 16 spheres) fall back to the scalar walk
            assert st.geometry_path == (path if has_tree else abi.PT_GEOM_SCALAR)
        elif path == abi.PT_GEOM_SMALL:
            # a list longer than 16 spheres falls back to the scalar walk
            assert st.geometry_path == (path if small else abi.PT_GEOM_SCALAR)
        elif path != abi.PT_GEOM_AUTO:
            assert st.geometry_path == path
        else:
            assert st.geometry_tuned == 1
            assert st.geometry_path in (abi.PT_GEOM_LDS, abi.PT_GEOM_SCALAR) + ((abi.PT_GEOM_BVH, abi.PT_GEOM_GRID) if has_tree else ()) + \
                ((abi.PT_GEOM_SMALL,) if small else ())
        t.close()
    assert t.lib.pt_set_option(None, 1, 1) == abi.PT_ERR_INVALID


def test_lens_off_with_a_negative_zero_direction_component(ora):
    """start_sample skips the lens arithmetic when lens_radius is 0 (the offset is a vector of zeros
    then) — except where a component of the ray direction is exactly -0, whose sign the offset's
    zeros decide: those waves must take the full form.  A camera whose rays all have d.x == -0
    (-0 components in llc / horizontal / vertical, origin.x == +0) next to the ordinary case."""
    for neg_zero in (True, False):
        sc = scenes.default_scene(64, 40, spp=3, max_depth=6)
        p = sc.params
        z = -0.0 if neg_zero else 0.0
        p.camera_origin = abi.f3(0.0, 0.1, 1.5)
        p.lower_left_corner = abi.f3(z, -0.9, 0.5)
        p.horizontal = abi.f3(z, 0.0, -1.0)     # s sweeps -z ...
        p.vertical = abi.f3(z, 1.8, 0.0)        # ... t sweeps +y: every direction lies in the y-z plane
        p.u = abi.f3(0.3, -0.2, 0.9)
        p.v = abi.f3(-0.7, 0.6, 0.1)
        p.lens_radius = 0.0
        sc.n_passes = 2
        for path in (abi.PT_GEOM_SMALL, abi.PT_GEOM_LDS):
            t, got = render_scene(sc, geometry_path=path)
            ref, seg = ora.render(sc.spheres, sc.params, 2)
            assert_bit_equal(got, ref, "lens off, d.x = %r, path %d" % (z, path))
            assert t.stats().segments == seg
            t.close()


def test_small_list_kernel_on_random_scenes(ora):
    """PT_GEOM_SMALL (at most 16 spheres straight from SGPRs, candidates finished group by group in
    list order): duplicates and concentric spheres (ties must go to the LATER list entry), negative
    radii, every material, cameras inside spheres, lists of 1 ... 16 spheres (group padding)."""
    from test_gpu_fuzz import random_scene

    for seed, n in enumerate([1, 2, 3, 4, 5, 7, 8, 9, 12, 13, 15, 16, 16, 9, 6, 11]):
        rng = np.random.default_rng(31000 + seed)
        sc = random_scene(rng, n, int(rng.integers(20, 120)), int(rng.integers(12, 70)), int(rng.integers(1, 6)),
                          int(rng.choice([1, 3, 8, 50])), 2)
        t, got = render_scene(sc, geometry_path=abi.PT_GEOM_SMALL)
        ref, seg = ora.render(sc.spheres, sc.params, 2)
        assert_bit_equal(got, ref, "small list, seed %d, %d spheres" % (seed, n))
        st = t.stats()
        assert st.segments == seg and st.geometry_path == abi.PT_GEOM_SMALL
        t.close()
    # tie order: two coincident spheres, different albedo — the image depends on which is later
    base = scenes.config1(120, 72, 4, 8)
    a = base.spheres[1:2].copy()
    b = a.copy()
    b["albedo"] = (0.1, 0.9, 0.1)
    imgs = []
    for order in ((a, b), (b, a)):
        sc = scenes.config1(120, 72, 4, 8)
        sc.spheres = np.concatenate([base.spheres[:1], order[0], base.spheres[2:], base.spheres[2:], order[1]])
        sc.spheres["uuid"] = np.arange(len(sc.spheres))
        sc.n_passes = 2
        t, got = render_scene(sc, geometry_path=abi.PT_GEOM_SMALL)
        ref, seg = ora.render(sc.spheres, sc.params, 2)
        assert_bit_equal(got, ref, "tie order")
        imgs.append(got)
        t.close()
    assert not np.array_equal(imgs[0], imgs[1])


def test_hierarchy_in_global_memory(ora):
    """10 001 spheres: nodes + slots (335 KB) exceed the LDS, pt_trace_kernel_bvh_gmem walks them
    in global memory / L2.  Window-checked against the oracle, whole frame against the list walk."""
    sc = scenes.config5(160, 90, 2, 1, 20)
    t, got, ref = _check_scene(ora, sc, window=(60, 84, 30, 46), geometry_path=abi.PT_GEOM_BVH)
    st = t.stats()
    assert st.geometry_path == abi.PT_GEOM_BVH and (2 * st.bvh_nodes + st.bvh_slots) * 16 > 160 * 1024
    t2, got2 = render_scene(sc, geometry_path=abi.PT_GEOM_SCALAR)
    assert_bit_equal(got, got2, "hierarchy walk vs list walk, whole frame")
    assert t2.stats().segments == st.segments
    t.close()
    t2.close()


def test_grid_kernels_for_scenes_beyond_the_lds(ora):
    """10 001 spheres: the cell records fit the LDS, the entries (0.7 MB) do not:
    pt_trace_kernel_grid_cells gathers them from global memory / L2.  Window-checked against the
    oracle, whole frame against the list walk; then the measuring twin of the same kernel."""
    sc = scenes.config5(160, 90, 2, 1, 20)
    t, got, ref = _check_scene(ora, sc, window=(60, 84, 30, 46), geometry_path=abi.PT_GEOM_GRID)
    st = t.stats()
    assert st.geometry_path == abi.PT_GEOM_GRID and st.grid_entries * 16 > 160 * 1024
    assert st.grid_cells[0] * st.grid_cells[1] * st.grid_cells[2] * 4 < 163776 - 15 * 4 * 1024  # the cell records fit beside a 1024-thread workgroup's parked state
    t2, got2 = render_scene(sc, geometry_path=abi.PT_GEOM_SCALAR)
    assert_bit_equal(got, got2, "grid walk vs list walk, whole frame")
    assert t2.stats().segments == st.segments
    t.reset()
    t.set_count_work(True)
    t.render_passes(1)
    assert_bit_equal(t.accum(), got, "measuring twin")
    w = t.stats().work
    assert w[0] > 0 and w[1] >= w[0] and w[2] > 0 and w[4] > 0 and w[6] > 0
    t.close()
    t2.close()


def test_grid_in_global_memory(ora):
    """60 001 spheres: 31 000 cell records (0.12 MB) and 0.13 M entries (2 MB) — neither fits the LDS, so
    pt_trace_kernel_grid_gmem reads both from global memory / L2.  Window-checked against the oracle,
    whole frame against the list walk."""
    sc = scenes.config5(128, 72, 2, 1, 12, n=60000)
    t, got, ref = _check_scene(ora, sc, window=(56, 72, 30, 40), geometry_path=abi.PT_GEOM_GRID)
    st = t.stats()
    assert st.geometry_path == abi.PT_GEOM_GRID
    assert st.grid_cells[0] * st.grid_cells[1] * st.grid_cells[2] * 4 > 163776 - 15 * 4 * 1024  # more cell records than fit beside the parked state
    t2, got2 = render_scene(sc, geometry_path=abi.PT_GEOM_SCALAR)
    assert_bit_equal(got, got2, "grid walk (global memory) vs list walk, whole frame")
    assert t2.stats().segments == st.segments
    t.close()
    t2.close()


def _literal_counters(t):
    """(irregular lane-steps, lane-steps handed over by the grid walk, wave steps that ran PHASE 3) of the last
    measuring-twin launch (dev interface, include/ptrace_dev.h)."""
    import ctypes as C
    ctr = np.zeros(128, np.uint64)
    t.lib.pt_debug_counters.restype = C.c_long
    t.lib.pt_debug_counters.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    assert t.lib.pt_debug_counters(t._ctx, ctr.ctypes.data_as(C.c_void_p), 128) == 128
    return int(ctr[16]), int(ctr[17]), int(ctr[18])


def test_far_and_nan_rays_in_a_scene_of_thousands(ora):
    """What the grid walk cannot take in a 10 001-sphere scene.  (a) Regular rays that start far outside
    the scene and reach the grid are handed over whole; the kernels whose entries do not fit the LDS
    look at them wave-wide (64 spheres at a time) instead of running the literal loop.  (b) A ray with
    a NaN in it is accepted by every sphere in turn (static/shader.frag:153-161 reject nothing that is
    NaN): the literal loop's answer — the LAST sphere, root NaN — is written down without the loop.
    Both against the oracle (window) and against the list walk (whole frame), then the counters of
    the measuring twin say that these rays really came by."""
    from ray_tracer_webgl_amd.scenes import _lib, _look_at
    far = scenes.config5(160, 90, 2, 1, 12)
    _look_at(_lib(), far.params, 160, 90, (900.0, 225.0, 900.0), (0.0, 8.0, 0.0), 3.0, 0.0, 500.0)  # 15x as far, 1/13 the angle; |d|^2 stays below 1e6
    nan = scenes.config5(96, 54, 2, 1, 6)
    nan.params.camera_origin[1] = float("nan")
    zero = scenes.config5(96, 54, 2, 1, 6)
    for k in range(3):
        zero.params.horizontal[k] = 0.0
        zero.params.vertical[k] = 0.0
        zero.params.lower_left_corner[k] = zero.params.camera_origin[k]  # direction == 0 exactly, then NaN hit points
    for sc, window, want in ((far, (60, 84, 30, 46), 1), (nan, (40, 56, 20, 30), 0), (zero, (40, 56, 20, 30), 0)):
        t, got, ref = _check_scene(ora, sc, window=window, geometry_path=abi.PT_GEOM_GRID)
        st = t.stats()
        assert st.geometry_path == abi.PT_GEOM_GRID and st.grid_entries * 16 > 160 * 1024  # pt_trace_kernel_grid_cells
        t2, got2 = render_scene(sc, geometry_path=abi.PT_GEOM_SCALAR)
        assert_bit_equal(got, got2, "grid walk vs list walk, whole frame")
        assert t2.stats().segments == st.segments
        t.reset()
        t.set_count_work(True)
        t.render_passes(1)
        assert_bit_equal(t.accum(), got, "measuring twin")
        irregular, handed_over, literal_steps = _literal_counters(t)
        if want == 1:
            assert handed_over > 1000, (irregular, handed_over)
            assert st.far_rays > 1000 and st.grid_fit_stale == 1  # the PRODUCT kernel's own tally said so too (PtStats.far_rays)
        else:
            assert irregular > 1000, (irregular, handed_over)
        t.close()
        t2.close()


@pytest.mark.parametrize("path", [abi.PT_GEOM_BVH, abi.PT_GEOM_GRID])
def test_measuring_twin_equals_the_timed_kernel(ora, path):
    """bench.py's `roofline.executed` tallies come from a DIFFERENT binary than the one it times (the
    `_count` twin of the walk kernel: same body + wave-uniform counters, different register
    allocation).  The tallies only describe the timed kernel if the twin takes the same control
    flow: same image bits, same segment count, on the bench's own scene (structure staged in LDS)."""
    sc = scenes.config2(256, 144, 4, 3, 50)
    t, got = render_scene(sc, geometry_path=path)
    st = t.stats()
    assert st.geometry_path == path and sum(st.work) == 0  # the timed kernel tallies nothing
    t.reset()
    t.set_count_work(True)
    t.set_params(sc.params)
    t.render_passes(3)
    twin = t.accum()
    sw = t.stats()
    assert_bit_equal(twin, got, "measuring twin vs timed kernel")
    assert sw.segments == st.segments
    w = sw.work
    assert w[6] > 0 and w[6] * 64 >= sw.segments  # wave steps x 64 lanes bound the shaded segments
    assert w[1] <= 64 * w[0] and w[3] <= 64 * w[2] and w[5] <= 64 * w[4]
    ref, seg = ora.render(sc.spheres, sc.params, 3, window=(100, 116, 60, 68))
    assert_bit_equal(twin[60:68, 100:116], ref[60:68, 100:116], "twin vs oracle window")
    t.close()


def test_small_list_twin_and_remainder_builds_agree(ora):
    """The small-list kernel exists once per list length modulo four (pt_kernels_small.hip: the last group
    tests exactly its own spheres) and once for any length (the measuring twin and the roulette build of
    pt_kernels_extra.hip: padding tested and masked).  Same bits and segment counts from both, on lists of
    every remainder, and the twin's phase clock runs (tools/wave_log.py reads it for config 4)."""
    import ctypes as C
    from test_gpu_fuzz import random_scene

    for seed, n in enumerate([9, 10, 11, 12, 1, 6]):
        rng = np.random.default_rng(47000 + seed)
        sc = random_scene(rng, n, 96, 54, 2, 8, 2)
        t, got = render_scene(sc, geometry_path=abi.PT_GEOM_SMALL)
        st = t.stats()
        assert st.geometry_path == abi.PT_GEOM_SMALL
        t.reset()
        t.set_count_work(True)
        t.set_params(sc.params)
        t.render_passes(2)
        assert_bit_equal(t.accum(), got, "small-list twin vs the build for %d spheres" % n)
        assert t.stats().segments == st.segments
        ctr = np.zeros(128, np.uint64)
        t.lib.pt_debug_counters.restype = C.c_long
        t.lib.pt_debug_counters.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        assert t.lib.pt_debug_counters(t._ctx, ctr.ctypes.data_as(C.c_void_p), 128) == 128
        assert ctr[24:32].sum() > 0 and ctr[24 + 7] > 0  # the phase clock ran; shading took some of it
        ref, seg = ora.render(sc.spheres, sc.params, 2)
        assert_bit_equal(got, ref, "small list of %d spheres vs oracle" % n)
        t.close()


@pytest.mark.parametrize("path", [abi.PT_GEOM_BVH, abi.PT_GEOM_GRID])
def test_carrying_stragglers_is_scheduling_only(ora, path):
    """PT_OPT_CARRY_LANES decides when a wave stops waiting for its last walks (they continue in
    the next wave step beside the fresh ones); images and segment counts cannot depend on it."""
    sc = scenes.config2(160, 90, 4, 2, 50)
    ref, seg = ora.render(sc.spheres, sc.params, 2)
    carried = {}
    for lanes in (0, 8, 24, 64):
        t = PathTracer(160, 90)
        t.set_geometry_path(path)
        t.set_carry_lanes(lanes)
        t.set_count_work(True)
        t.set_spheres(sc.spheres)
        t.set_params(sc.params)
        t.reserve_passes(2)
        t.render_passes(2)
        assert_bit_equal(t.accum(), ref, "carry_lanes %d" % lanes)
        st = t.stats()
        assert st.segments == seg and st.geometry_path == path
        carried[lanes] = st.work[7]
        t.close()
    assert carried[0] == 0 and carried[8] > 0 and carried[24] >= carried[8]
    t = PathTracer(8, 8)
    assert t.lib.pt_set_option(t._ctx, abi.PT_OPT_CARRY_LANES, 65) == abi.PT_ERR_INVALID
    t.close()


@pytest.mark.parametrize("path", [abi.PT_GEOM_LDS, abi.PT_GEOM_GRID])
def test_deferred_refill_is_scheduling_only(ora, path):
    """PT_OPT_REFILL_MIN: a busy wave runs the item decode only once a few lanes wait for an item.
    Which step a lane gets its next item in cannot change what the item computes."""
    sc = scenes.config2(160, 90, 2, 3, 50)
    ref, seg = ora.render(sc.spheres, sc.params, 3)
    for lanes in (1, 4, 16, 64):
        t = PathTracer(160, 90)
        t.set_geometry_path(path)
        t.set_refill_min(lanes)
        t.set_spheres(sc.spheres)
        t.set_params(sc.params)
        t.reserve_passes(3)
        t.render_passes(3)
        assert_bit_equal(t.accum(), ref, "refill_min %d" % lanes)
        assert t.stats().segments == seg
        t.close()
    t = PathTracer(8, 8)
    assert t.lib.pt_set_option(t._ctx, abi.PT_OPT_REFILL_MIN, 0) == abi.PT_ERR_INVALID
    t.close()


def test_pt_tune_settles_the_path_and_leaves_a_clean_context(ora):
    sc = scenes.config2(128, 72, 4, 2, 50)
    t = PathTracer(128, 72)
    t.set_spheres(sc.spheres)
    t.set_params(sc.params)
    t.reserve_passes(2)
    t.tune(2)
    st = t.stats()
    assert st.geometry_tuned == 1 and st.total_spp == 0 and st.segments == 0
    t.render_passes(2)
    ref, seg = ora.render(sc.spheres, sc.params, 2)
    assert_bit_equal(t.accum(), ref, "after pt_tune")
    assert t.stats().segments == seg and t.stats().geometry_path in (abi.PT_GEOM_LDS, abi.PT_GEOM_SCALAR, abi.PT_GEOM_BVH,
                                                                           abi.PT_GEOM_GRID)
    t.close()


def test_plain_c_example_matches_python_path(tmp_path):
    """examples/render.c drives the same ABI from plain C (no Python/torch in the process): its
    PPM must equal the frame the ctypes path resolves for the same calls."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "examples")])
    out = str(tmp_path / "c.ppm")
    subprocess.check_call([os.path.join(root, "examples", "render"), out, "160", "88", "3"])
    data = open(out, "rb").read()
    header, pix = data.split(b"255\n", 1)
    img = np.frombuffer(pix, dtype=np.uint8).reshape(88, 160, 3)[::-1]
    sc = scenes.default_scene(160, 88, spp=25, max_depth=8)
    sc.n_passes = 3
    t, _ = render_scene(sc)
    assert np.array_equal(t.resolve_rgba8(True)[..., :3], img)
    t.close()


def test_plain_c_animation_matches_the_python_frame_loop(tmp_path):
    """examples/animate.c: the reference's animation loop from plain C — n ticks as ONE pt_render_frames
    call (one hipGraph replayed n times, per-frame state on the device).  Its canvas must be the canvas of
    app.FrameLoop driven tick by tick with host-made uniforms."""
    import subprocess

    from ray_tracer_webgl_amd.app import FrameLoop

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "examples"), "animate"])
    out = str(tmp_path / "a.ppm")
    w, h, n = 128, 72, 12
    subprocess.check_call([os.path.join(root, "examples", "animate"), out, str(w), str(h), str(n)])
    data = open(out, "rb").read()
    header, pix = data.split(b"255\n", 1)
    img = np.frombuffer(pix, dtype=np.uint8).reshape(h, w, 3)[::-1]
    loop = FrameLoop(w, h, mode="reference")
    loop.state.set_flags(is_paused=False)
    for k in range(n):
        assert loop.frame(3000.0 + 16.5 * k) is True
    assert np.array_equal(loop.canvas[..., :3], img)
    loop.close()


def test_plain_c_flight_with_a_key_held_follows_the_camera_with_refits(tmp_path):
    """examples/fly.c: the reference's rAF loop WITH a movement key held (State::update_position, src/state.rs:411-441)
    on a 1 500-sphere field, from plain C — per tick update_position, update_render_globals, run_setters, then the two
    lines a host of a large scene adds (pt_grid_fit == 1 -> pt_refit_grid) and webgl::render (pt_render_frame).  Its canvas
    after 30 ticks of flying backwards out of the scene must be the Python FrameLoop's (same State, same key, same clock;
    FrameLoop is pinned to the oracle's frame loop, refits included, by the flight test above), it must have refitted as
    often, and practically none of its segments may have taken the far path; with refits switched off the canvas is the
    same bytes and the far path's share shows what the two lines are for."""
    import ctypes as C
    import re
    import struct
    import subprocess

    from ray_tracer_webgl_amd.app import FrameLoop

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "examples"), "fly"])
    sph = np.ascontiguousarray(scenes.field_spheres(1500))
    sc = scenes.config1(128, 72, 1, 8)  # (the file format carries uniforms; fly.c takes only the spheres)
    scene = str(tmp_path / "field.bin")
    with open(scene, "wb") as f:
        f.write(b"PTSC" + struct.pack("<4I", len(sph), C.sizeof(abi.PtSphere), C.sizeof(abi.PtParams), 1))
        f.write(bytes(sc.params))
        f.write(sph.tobytes())
    w, h, ticks, dt = 128, 72, 30, 11500.0
    runs = {}
    for refit in (1, 0):
        out = str(tmp_path / ("fly%d.ppm" % refit))
        r = subprocess.run([os.path.join(root, "examples", "fly"), out, scene, str(w), str(h), str(ticks), str(refit)],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        line = [ln for ln in r.stdout.splitlines() if ln.startswith(out)][-1]
        m = re.search(r"(\d+) refits, grid for ([\d.]+) scene radii \(the camera needs ([\d.]+), fit flag (\d), kernel build (\d)\), (\d+) segments, (\d+) of them", line)
        assert m, line
        header, pix = open(out, "rb").read().split(b"255\n", 1)
        img = np.frombuffer(pix, dtype=np.uint8).reshape(h, w, 3)[::-1]
        runs[refit] = (img, int(m.group(1)), float(m.group(2)), float(m.group(3)), int(m.group(4)), int(m.group(5)), int(m.group(6)), int(m.group(7)))
        print(line)
    # the same flight through the Python frame loop
    small = np.abs(sph["radius"]) < 100.0
    mid = 0.5 * (sph["center"][small].astype(np.float64).min(0) + sph["center"][small].astype(np.float64).max(0))
    loop = FrameLoop(w, h, mode="reference", host_spheres=_host_spheres(sph))
    loop.state.set_flags(is_paused=False)
    loop.state.set_camera_origin(mid + np.array([6.5, 2.0, 7.25]))
    loop.state.set_camera_angles(-132.0, -11.5)
    loop.state.set_keys(abi.KEY_S)
    for k in range(ticks):
        assert loop.frame(dt * (k + 1)) is True
    canvas = loop.canvas[..., :3]
    st = loop.tracer.stats()
    img1, refits1, have1, need1, flag1, build1, seg1, far1 = runs[1]
    img0, refits0, have0, need0, flag0, build0, seg0, far0 = runs[0]
    assert np.array_equal(img1, canvas) and np.array_equal(img0, canvas)
    assert refits1 == loop.grid_refits >= 2 and have1 == float(st.grid_near_factor) >= need1 and flag1 == 0 and build1 == 1
    assert seg1 == seg0 == st.segments and far1 == st.far_rays and far1 < 0.002 * seg1
    assert refits0 == 0 and have0 == 3.0 and flag0 == 1 and build0 == 2 and far0 > 20 * max(far1, 1)
    loop.close()


@pytest.mark.parametrize("which", ["small-list", "grid"])
def test_every_launched_wave_is_resident(which):
    """VERDICT r4 #5.  A full-size launch puts `CUs x resident workgroups` on the machine and every one of them must BE
    there from the start: the occupancy query counts 7 waves per SIMD for a kernel of 106 SGPRs, the SIMD holds 6 (the
    trap handler's 16 SGPRs per wave), and through round 4 every seventh workgroup of the list and small-list launches
    started only when another one had ended — for a statically dealt launch, its share of the frame began when everybody
    else was done.  The measuring twins log where and when each wave starts (HW_ID, s_memrealtime): all waves start
    before the first one ends, within 50 us of each other, each in a wave slot of its own."""
    import ctypes as C

    if which == "small-list":
        sc, path, n = scenes.default_scene(1280, 702, 4, 8, 16), abi.PT_GEOM_SMALL, 16
    else:
        sc, path, n = scenes.config2(1920, 1080, 2, 8, 50), abi.PT_GEOM_GRID, 8
    t = PathTracer(sc.params.width, sc.params.height)
    t.set_spheres(sc.spheres)
    t.set_params(sc.params)
    t.reserve_passes(n)
    t.set_geometry_path(path)
    t.set_count_work(True)
    t.render_passes(n)  # (cold: the twin's code object, the machine's clocks — the log read below is the second launch's)
    t.synchronize()
    t.reset()
    t.render_passes(n)
    buf = np.zeros((20000, 4), np.uint64)
    t.lib.pt_debug_wave_log.restype = C.c_long
    t.lib.pt_debug_wave_log.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    k = t.lib.pt_debug_wave_log(t._ctx, buf.ctypes.data_as(C.c_void_p), len(buf))
    t.close()
    import torch  # (plumbing: the device's CU count)

    n_cus = torch.cuda.get_device_properties(0).multi_processor_count
    assert k >= n_cus * 4 * 6, "a full-size launch fills the machine: %d waves on %d CUs" % (k, n_cus)
    start, end, hw = buf[:k, 0].astype(np.int64), buf[:k, 2].astype(np.int64), buf[:k, 3]
    assert (start > 0).all() and (end >= start).all()
    spread_us = (start.max() - start.min()) / 100.0  # 100 MHz ticks
    # the residency facts: every wave starts before the first one ends, each in a wave slot of its own.  The spread of the
    # start times is a clock matter (a warm launch: ~10 us; a cold or busy box: > 100 us, seen once in round 5) — reported,
    # and a failure only when it is structural (a millisecond: waves that waited for others to END)
    print("%s: %d waves start within %.1f us" % (which, k, spread_us))
    assert start.max() < end.min(), "%d of %d waves start after the first one has ended" % (int((start >= end.min()).sum()), k)
    assert spread_us < 1000.0, "waves start %.1f us apart" % spread_us
    where = ((hw >> np.uint64(32)) << np.uint64(16)) | (hw & np.uint64(0xffff))  # XCC | SE, SH, CU, SIMD, wave slot
    assert len(np.unique(where)) == k, "%d waves in %d wave slots" % (k, len(np.unique(where)))


def test_tune_fits_the_grid_to_the_view(ora):
    """pt_set_spheres builds the grid for rays that start within 2 s0 of the scene's middle (it does not know the camera);
    pt_tune, the set-up call that knows scene AND uniforms, rebuilds it — here in its unmeasured mode (PT_OPT_GRID_FIT 1) for
    the smallest margin class that covers the camera: fewer entries for a camera close by, a grid that still serves a camera
    far out (whose rays would otherwise all be tested against the whole list).  Speed only — the image is the oracle's either way."""
    from ray_tracer_webgl_amd import _lib

    lib = _lib.load()
    sph = scenes.field_spheres(1500)
    entries = {}
    for name, cam in (("close", (60.0, 15.0, 60.0)), ("far out", (190.0, 60.0, 190.0))):
        p = scenes._base_params(2, 8)
        scenes._look_at(lib, p, 96, 54, cam, (0.0, 8.0, 0.0), 40.0, 0.0, 80.0)
        sc = scenes.Scene("field", sph, p, 2, "field")
        ref, seg = ora.render(sc.spheres, sc.params, 2)
        t = PathTracer(96, 54)
        t.set_geometry_path(abi.PT_GEOM_GRID)
        t.set_grid_fit(True)  # the class the camera needs, unmeasured (the measured choice: test_tune_measures_the_margin_class)
        t.set_spheres(sc.spheres)
        t.set_params(sc.params)
        t.reserve_passes(2)
        built = t.stats().grid_entries
        if name == "close":  # (the camera far out is only rendered on the fitted grid: on the other every primary ray walks the list)
            t.render_passes(2)
            assert_bit_equal(t.accum(), ref, "grid as built")
            assert t.stats().segments == seg
        t.tune(2)
        fitted = t.stats().grid_entries
        t.reset()
        t.render_passes(2)
        assert_bit_equal(t.accum(), ref, "grid fitted to the %s camera" % name)
        assert t.stats().segments == seg and t.stats().geometry_path == abi.PT_GEOM_GRID
        entries[name] = (built, fitted)
        t.tune(2)  # a second call finds the grid in place
        assert t.stats().grid_entries == fitted
        t.close()
    assert entries["close"][1] < entries["close"][0], entries      # a smaller d_near, smaller margins
    assert entries["far out"][1] > entries["far out"][0], entries  # a larger one


def _host_spheres(sph):
    """abi.PtHostSphere records (f64, src/glsl.rs:27-40) holding exactly the values of f32 PtSphere records"""
    out = []
    for s in sph:
        h = abi.PtHostSphere()
        h.center = abi.d3(*[float(x) for x in s["center"]])
        h.radius = float(s["radius"])
        h.type, h.uuid = int(s["type"]), int(s["uuid"])
        h.albedo = abi.d3(*[float(x) for x in s["albedo"]])
        h.fuzz, h.refraction_index = float(s["fuzz"]), float(s["refraction_index"])
        out.append(h)
    return out


def test_a_camera_that_flies_out_of_the_fitted_region_keeps_a_grid_that_serves_it(ora):
    """VERDICT r5 #2 / ADVICE r5 (medium).  The reference's host moves the camera every tick a key is held
    (State::update_position, src/state.rs:411-441); the grid of a large scene is fitted to ONE region of ray origins, and a
    camera that leaves it sends every primary ray down the far path — exact, but tested against the whole list.  The
    boundary now says so (PtStats.grid_fit_stale / pt_grid_fit: host arithmetic in no launch's way; PtStats.far_rays: what
    the walk really handed over) and FrameLoop acts on it (pt_refit_grid before the tick's frame is traced).  A camera flies
    from inside a 1 500-sphere field to five scene radii out over 30 ticks and back in: every frame's canvas is the oracle's
    frame loop bit for bit (one fresh pass, the shader's blend with the other texture), the grid is refitted exactly when the
    camera crosses into a wider margin class, practically no ray takes the far path in any frame — and WITHOUT the policy
    the same flight ends with most primary rays on it (so the tally would have caught the cliff) — still the oracle's frames
    bit for bit, through the build of the grid kernel that hands far rays to the whole wave (PtStats.grid_kernel_build 2)."""
    import math

    from ray_tracer_webgl_amd.app import FrameLoop
    from test_grid import build as grid_build

    sph = scenes.field_spheres(1500)
    rc, g = grid_build(sph)
    assert rc == 0
    c0, s0 = g["c0"].astype(np.float64), float(g["s0"])
    w, h, ticks = 96, 54, 30
    direction = np.array([0.66, 0.18, 0.73]) / np.linalg.norm([0.66, 0.18, 0.73])

    def fly(loop, k):  # tick k of the flight: 0.2 s0 from the middle ... 5 s0 out, looking back at the field
        pos = c0 + direction * s0 * (0.2 + 4.8 * k / (ticks - 1))
        f = (c0 - pos) / np.linalg.norm(c0 - pos)
        loop.state.set_camera_origin(pos)
        loop.state.set_camera_angles(math.degrees(math.atan2(f[2], f[0])), math.degrees(math.asin(f[1])))

    shares = {}
    for policy in (True, False):
        loop = FrameLoop(w, h, mode="reference", host_spheres=_host_spheres(sph))
        loop.LOOSE_FRAMES = 4
        if not policy:
            loop._keep_the_grid_fitted = lambda: None
        loop.tracer.set_geometry_path(abi.PT_GEOM_GRID)
        loop.state.set_flags(is_paused=False)
        loop.state.set_quality(2, 8)
        spheres = loop.state.spheres()
        assert np.array_equal(spheres["center"], sph["center"]) and np.array_equal(spheres["radius"], sph["radius"])
        tex = [np.zeros((h, w, 4), np.uint8), np.zeros((h, w, 4), np.uint8)]
        factors, far_share, seen = [], [], (0, 0)
        order = list(range(ticks)) + ([ticks - 1 - k for k in range(1, 13)] if policy else [])  # out, and a bit of the way back in
        for i, k in enumerate(order):
            fly(loop, k)
            now = 100.0 + 16.5 * i
            assert loop.frame(now) is True
            # (both flights) the oracle's frame loop: one fresh pass at u_time = now, the shader's blend with the other texture
            v = loop.state.view()
            p = loop.state.to_params(now)
            acc, _ = ora.render(spheres, p, 1)
            expect = ora.blend_rgba8(acc, p.samples_per_pixel, p, tex[(v.even_odd_count + 1) % 2])
            assert np.array_equal(loop.canvas, expect), "tick %d (flight position %d, policy %s)" % (i, k, policy)
            tex[v.even_odd_count % 2] = expect
            st = loop.tracer.stats()
            assert st.geometry_path == abi.PT_GEOM_GRID
            far_share.append((st.far_rays - seen[0]) / max(st.segments - seen[1], 1))
            seen = (st.far_rays, st.segments)
            factors.append(round(float(st.grid_near_factor), 2))
            if policy:
                assert st.grid_fit_stale != 1 and st.grid_kernel_build == 1, (i, k, st.grid_near_factor, st.grid_need_factor, st.grid_kernel_build)
            else:  # a stale view is walked by the build that hands far rays to the whole wave (same bits: the canvas above)
                assert st.grid_kernel_build == (2 if st.grid_fit_stale == 1 else 1), (i, k, st.grid_fit_stale, st.grid_kernel_build)
        shares[policy] = far_share
        print("policy", policy, "grid factors", factors, "far-ray share per frame", ["%.4f" % x for x in far_share])
        if policy:
            # as built (3 s0: a refit never goes below the default class, whether a tighter one pays is pt_tune's measurement) ->
            # refitted on the way out exactly when the camera needed a wider class, never before -> tightened again on the
            # way back in, after LOOSE_FRAMES frames on a grid looser than needed
            out = factors[:ticks]
            assert set(out) == {3.0, 4.0, 5.5, 8.0} and out == sorted(out), out
            assert factors[-1] < 8.0 and loop.grid_refits == len([1 for a, b in zip([3.0] + factors, factors) if a != b]) >= 4, (factors, loop.grid_refits)
            assert max(far_share) < 0.01, far_share
        else:
            assert factors == [3.0] * ticks and loop.grid_refits == 0 and loop.tracer.grid_fit() == 1
        loop.close()
    # the cliff the policy removes: at the far end of the flight the unfitted grid hands the camera's rays to the whole list
    assert shares[False][ticks - 1] > 20 * max(shares[True][ticks - 1], 1e-4), (shares[False][ticks - 1], shares[True][ticks - 1])


def test_a_refit_between_two_replayed_series_recaptures_their_graphs():
    """pt_render_frames replays hipGraphs whose launches bake the grid's buffers and numbers in.  A refit between two series
    (the camera has left the fitted region: FrameLoop.frames refits before it replays) must make the next series capture
    anew — never replay launches that point at the old grid: 9 ticks inside the field, a jump to four scene radii out, 9 more
    ticks (two groups of four and a single frame per series), canvas and both textures against the same ticks issued one by
    one (which test_a_camera_that_flies_out_... pins to the oracle's frame loop, refits included)."""
    import math

    from ray_tracer_webgl_amd.app import FrameLoop
    from test_grid import build as grid_build

    sph = scenes.field_spheres(1500)
    rc, g = grid_build(sph)
    c0, s0 = g["c0"].astype(np.float64), float(g["s0"])
    w, h = 96, 54
    loops = []
    for _ in range(2):
        lp = FrameLoop(w, h, mode="reference", host_spheres=_host_spheres(sph))
        lp.tracer.set_geometry_path(abi.PT_GEOM_GRID)
        lp.state.set_flags(is_paused=False)
        lp.state.set_quality(1, 8)
        loops.append(lp)
    series, ticks = loops
    direction = np.array([0.66, 0.18, 0.73]) / np.linalg.norm([0.66, 0.18, 0.73])
    now, factors = 100.0, []
    for dist_s0 in (0.3, 4.0):
        pos = c0 + direction * s0 * dist_s0
        f = (c0 - pos) / np.linalg.norm(c0 - pos)
        for lp in loops:
            lp.state.set_camera_origin(pos)
            lp.state.set_camera_angles(math.degrees(math.atan2(f[2], f[0])), math.degrees(math.asin(f[1])))
        assert series.frames(9, now, 16.5) == 9
        for k in range(9):
            assert ticks.frame(now + 16.5 * k) is True
        assert np.array_equal(series.canvas, ticks.canvas), "series at %.1f s0" % dist_s0
        ta, tb = series.textures, ticks.textures
        assert np.array_equal(ta[0], tb[0]) and np.array_equal(ta[1], tb[1])
        st = series.tracer.stats()
        factors.append(float(st.grid_near_factor))
        assert st.grid_fit_stale == 0 and st.far_rays < 0.01 * max(st.segments, 1)
        assert st.segments == ticks.tracer.stats().segments
        now += 16.5 * 9
    assert factors == [3.0, 5.5] and series.grid_refits == 1 and ticks.grid_refits == 1, (factors, series.grid_refits)
    for lp in loops:
        lp.close()


def test_tune_measures_the_margin_class():
    """The smallest class that covers the CAMERA is a lower bound, not the answer.  A camera inside a 1 500-sphere field that
    looks across it sees the ground out to the horizon; the rays that bounce off it beyond the near region and come back
    into the grid's box take the far path, and each costs what hundreds of walked segments cost — the grid for 2.5 s0 (which
    the camera alone would allow) renders this view about twice as slowly as the default 3 s0.  pt_tune therefore times its
    candidates: whatever it keeps must be within 15 % of the best of the classes it could have kept, the frame the same bits
    on all of them, the flag must say "fits", and the far-ray tally must show WHY the tight class loses.  (On a scene whose
    staged entries take a good part of the LDS it also times the build that gathers them from L2, and may keep that.)"""
    import math
    import time

    from ray_tracer_webgl_amd import _lib
    from test_grid import build as grid_build

    sph = scenes.field_spheres(1500)
    rc, g = grid_build(sph)
    c0, s0 = g["c0"].astype(np.float64), float(g["s0"])
    p = scenes._base_params(8, 8)
    pos = c0 + np.array([0.66, 0.18, 0.73]) / np.linalg.norm([0.66, 0.18, 0.73]) * 0.2 * s0
    scenes._look_at(_lib.load(), p, 1280, 720, tuple(pos), tuple(c0), 60.0, 0.0, 80.0)
    sc = scenes.Scene("field", sph, p, 2, "field")

    def run(mode):
        t = PathTracer(1280, 720)
        t.set_geometry_path(abi.PT_GEOM_GRID)
        t.set_spheres(sc.spheres)
        t.set_params(sc.params)
        t.reserve_passes(2)
        if mode == "measured":
            t.tune(2)
        elif mode == "need":
            t.set_grid_fit(True)
            t.tune(2)
        ms = []
        for _ in range(4):
            t.reset()
            t.set_params(sc.params)
            t.render_passes(2)
            ms.append(t.stats().render_kernel_ms)
        st = t.stats()
        out = (min(ms[1:]), float(st.grid_near_factor), st.far_rays / max(st.segments, 1), t.accum(), st.grid_fit_stale, int(st.grid_kernel_build))
        t.close()
        return out

    built, need, measured = run("built"), run("need"), run("measured")
    print("as built: %.3f ms at %.1f s0 (far share %.2e); the camera's class: %.3f ms at %.1f s0 (far share %.2e); pt_tune kept %.1f s0 and build %d: %.3f ms"
          % (built[0], built[1], built[2], need[0], need[1], need[2], measured[1], measured[5], measured[0]))
    assert built[1] == 3.0 and need[1] == 2.5 and measured[1] >= 2.5 and measured[4] == 0
    assert built[5] == 1 and need[5] == 1 and measured[5] in (1, 2)  # (untuned and unmeasured contexts stage what fits; pt_tune also times the gathering build)
    assert_bit_equal(need[3], built[3], "2.5 s0 vs 3 s0")
    assert_bit_equal(measured[3], built[3], "the measured class vs 3 s0")
    assert measured[0] <= 1.15 * min(built[0], need[0]), (measured[0], built[0], need[0])
    assert need[2] > 2.0 * built[2] and need[0] > 1.2 * built[0], "this view is the one where the camera's class loses"


def test_no_frame_of_the_flight_is_a_cliff():
    """... and what the flight costs ("no frame slower than 2x the fitted-grid frame").  The same flight at
    1280x720, 8 spp per frame — frames of milliseconds, long against a refit's host work.  At each of the 30 camera positions:
    the frame as FrameLoop issues it (the FIRST one includes the refit when the camera has crossed into a wider class; then
    the best of three more) against the best of three on a context whose grid pt_refit_grid has just fitted to exactly that
    camera.  Steady frames within 1.5x, the frame that carries a refit within 2x + 3 ms of host work (single host-clock
    samples: all but at most two of thirty); and the control — the flight on the grid as built, no policy — IS a cliff at the
    far end (so the bound means something).  Clock-based, hence generous: the fitted frame is 0.5-1 ms, the unfitted far-end
    frame 8-10 ms."""
    import math
    import time

    from ray_tracer_webgl_amd.app import FrameLoop
    from test_grid import build as grid_build

    sph = scenes.field_spheres(1500)
    rc, g = grid_build(sph)
    assert rc == 0
    c0, s0 = g["c0"].astype(np.float64), float(g["s0"])
    w, h, ticks = 1280, 720, 30
    direction = np.array([0.66, 0.18, 0.73]) / np.linalg.norm([0.66, 0.18, 0.73])
    host = _host_spheres(sph)

    def make(policy):
        loop = FrameLoop(w, h, mode="reference", host_spheres=host)
        if not policy:
            loop._keep_the_grid_fitted = lambda: None
        loop.tracer.set_geometry_path(abi.PT_GEOM_GRID)
        loop.state.set_flags(is_paused=False)
        loop.state.set_quality(8, 8)
        return loop

    def fly(loop, k):
        pos = c0 + direction * s0 * (0.2 + 4.8 * k / (ticks - 1))
        f = (c0 - pos) / np.linalg.norm(c0 - pos)
        loop.state.set_camera_origin(pos)
        loop.state.set_camera_angles(math.degrees(math.atan2(f[2], f[0])), math.degrees(math.asin(f[1])))

    now = [100.0]

    def frame_ms(loop):
        now[0] += 16.5
        loop.tracer.synchronize()
        t0 = time.perf_counter()
        assert loop.frame(now[0]) is True
        loop.tracer.synchronize()
        return (time.perf_counter() - t0) * 1e3

    flown, fitted, plain = make(True), make(True), make(False)
    for loop in (flown, fitted, plain):  # (a context's very first frame loads code and probes its tile order: not the flight's cost)
        fly(loop, 0)
        frame_ms(loop)
    rows = []
    for k in range(ticks):
        for loop in (flown, fitted, plain):
            fly(loop, k)
        refits = flown.grid_refits
        first = frame_ms(flown)
        carried_a_refit = flown.grid_refits != refits
        steady = min(frame_ms(flown) for _ in range(3))
        fitted.tracer.set_params(fitted.state.to_params(now[0]))
        fitted.tracer.refit_grid()  # exactly the class this camera needs, tight or not
        frame_ms(fitted)
        best = min(frame_ms(fitted) for _ in range(3))
        frame_ms(plain)
        unfitted = min(frame_ms(plain) for _ in range(2))
        rows.append((k, round(first, 3), carried_a_refit, round(steady, 3), round(best, 3), round(unfitted, 3)))
    print("flight position, first frame ms, carried a refit, steady ms, fitted-grid ms, as-built-grid ms:")
    for r in rows:
        print("  ", r)
    for k, first, refit, steady, best, unfitted in rows:
        assert steady <= 1.5 * best + 0.1, (k, steady, best)  # (each the best of three: a host hiccup does not reach it)
    # the FIRST frame at a position is one host-clock sample (it may carry a refit, a tile-order probe every 64 frames — or a
    # 10-ms scheduling hiccup of the box, seen once in 30): the bound holds for all but at most two of the thirty
    late = [(k, first, refit, best) for k, first, refit, steady, best, unfitted in rows if first > 2.0 * best + (3.0 if refit else 0.3)]
    assert len(late) <= 2, late
    assert sum(1 for r in rows if r[2]) >= 3
    assert rows[-1][5] > 2.0 * rows[-1][4], rows[-1]  # the control: without the policy the far end is the cliff
    for loop in (flown, fitted, plain):
        loop.close()


def test_plain_c_multi_gpu_example_gathers_the_single_gpu_frame(tmp_path, ora):
    """examples/render_bands.c (VERDICT r4 "missing" #2): multi-GPU rendering from ONE plain-C host process, as the
    reference's Rust host would drive it — one pt_ctx per rank with its interleaved row bands (PtParams.band_*), one
    ncclAllGather of the padded per-rank radiance buffers straight through librccl, de-interleaved with pt_band_row.
    On this one-GPU box: (i) one rank through RCCL (ncclCommInitAll + ncclAllGather with a communicator of one),
    (ii) three ranks sharing the device with the gather done by copies (RCCL refuses two ranks on one device; the
    program says REHEARSAL).  Both files must hold the single-context frame, bit for bit (= the oracle's)."""
    import ctypes as C
    import struct
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "examples"), "render_bands"])
    sc = scenes.config2(160, 90, 4, 3, 12)
    scene = str(tmp_path / "scene.bin")
    sph = np.ascontiguousarray(sc.spheres)
    with open(scene, "wb") as f:
        f.write(b"PTSC" + struct.pack("<4I", len(sph), C.sizeof(abi.PtSphere), C.sizeof(abi.PtParams), sc.n_passes))
        f.write(bytes(sc.params))
        f.write(sph.tobytes())
    assert sph.dtype.itemsize == C.sizeof(abi.PtSphere)
    t, single = render_scene(sc)
    t.close()
    ref, seg = ora.render(sc.spheres, sc.params, sc.n_passes)
    assert_bit_equal(single, ref, "single context vs oracle")
    for ranks, how in ((1, "ncclAllGather (RCCL)"), (3, "REHEARSAL")):
        out = str(tmp_path / ("bands%d.f32" % ranks))
        r = subprocess.run([os.path.join(root, "examples", "render_bands"), out, str(ranks), "4", scene],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        line = [ln for ln in r.stdout.splitlines() if ln.startswith(out)][-1]  # (RCCL prints its banner to stdout too)
        assert how in line and "%d segments" % seg in line and "same gathered frame: yes" in line, line
        got = np.fromfile(out, dtype=np.float32).reshape(90, 160, 4)
        assert_bit_equal(got, single, "render_bands with %d rank(s)" % ranks)


def test_frames_probe_a_cost_sorted_tile_order_without_a_trace_in_results_or_statistics(ora):
    """pt_render_frame(s) probe a cost-sorted tile order (one extra pass with the cost feedback on, into scratch) when there is
    none for the scene and view: scheduling only.  The canvas is the oracle's frame loop, the segment statistics count the
    frames' own segments and nothing else, and a second series at the same view does not probe again (same counts)."""
    from ray_tracer_webgl_amd.app import FrameLoop

    w, h, n = 120, 67, 6
    loop = FrameLoop(w, h, mode="reference")
    loop.state.set_flags(is_paused=False)
    loop.tracer.reset()
    assert loop.frames(n, 100.0, 16.5) == n
    chk = FrameLoop(w, h, mode="reference")
    chk.state.set_flags(is_paused=False)
    tex = [np.zeros((h, w, 4), np.uint8), np.zeros((h, w, 4), np.uint8)]
    spheres = chk.state.spheres()
    total = 0
    for k in range(n):
        now = 100.0 + 16.5 * k
        chk.state.update_position(now if k == 0 else 16.5)
        chk.state.update_render_globals()
        v, p = chk.state.view(), chk.state.to_params(now)
        acc, seg = ora.render(spheres, p, 1)
        total += seg
        expect = ora.blend_rgba8(acc, p.samples_per_pixel, p, tex[(v.even_odd_count + 1) % 2])
        tex[v.even_odd_count % 2] = expect
    assert np.array_equal(loop.canvas, expect)
    assert loop.tracer.stats().segments == total, (loop.tracer.stats().segments, total)
    loop.close()
    chk.close()


def test_a_series_with_a_group_of_64_frames_matches_single_ticks():
    """pt_render_frames replays the largest groups first: 85 frames = one group of 64 (ONE trace launch of 64 passes, one
    blend kernel), one of 16, one of 4 and a single frame.  Canvas and both textures must be those of 85 ticks issued
    one by one (pt_render_frame, itself checked against the oracle's frame loop above)."""
    from ray_tracer_webgl_amd.app import FrameLoop

    w, h, n = 72, 40, 85
    a = FrameLoop(w, h, mode="reference")
    a.state.set_flags(is_paused=False)
    for k in range(n):
        assert a.frame(100.0 + 16.5 * k) is True
    b = FrameLoop(w, h, mode="reference")
    b.state.set_flags(is_paused=False)
    assert b.frames(n, 100.0, 16.5) == n
    assert np.array_equal(a.canvas, b.canvas)
    for ta, tb in zip(a.textures, b.textures):
        assert np.array_equal(ta, tb)
    assert a.tracer.stats().segments == b.tracer.stats().segments
    a.close()
    b.close()
