"""Russian roulette (PT_OPT_RUSSIAN_ROULETTE) — the ONE mode of the product that is not bit-exact
against the oracle, by construction: the reference never ends a path early (static/shader.frag:297-339,
SURVEY.md §0 F3), so this opt-in mode draws different samples.  What it must keep is every pixel's
EXPECTATION.  Checked the way tests/test_reference_pins.py checks the oracle itself:

  * against RNG-free float64 expectations (tests/analytic.py): the cosine-lobe closed form for the
    first diffuse bounce, and the exhaustive reflect/refract tree of the glass sphere;
  * against the oracle's (roulette-free) mean on BASELINE config 4 — the deep-bounce scene the mode
    exists for — with the standard errors of both sides estimated from independent passes;
  * and that it is OFF by default and does what it is for: far fewer segments on config 4.
"""
import numpy as np
import pytest

import analytic
from ray_tracer_webgl_amd import abi, scenes
from ray_tracer_webgl_amd.tracer import PathTracer

pytestmark = pytest.mark.gpu


def pass_means(sc, n_passes, roulette, path=abi.PT_GEOM_AUTO):
    """per-pass pixel means (n_passes, h, w, 3) float64 and the segment count, one launch per pass"""
    p = sc.params.copy()
    p.time_step = abi.PT_TIME_STEP_DECORRELATED
    t = PathTracer(p.width, p.height)
    t.set_geometry_path(path)
    t.set_russian_roulette(roulette)
    t.set_spheres(sc.spheres)
    out = []
    seg = 0
    for k in range(n_passes):
        q = p.copy()
        q.first_pass = k
        t.set_params(q)
        t.reset()
        t.render_passes(1)
        a = t.accum()
        assert (a[..., 3] == p.samples_per_pixel).all()
        out.append(a[..., :3].astype(np.float64) / p.samples_per_pixel)
        seg += t.stats().segments
    t.close()
    return np.stack(out), seg


def z_stats(z):
    z = z[np.isfinite(z)]
    return float(np.abs(z).max()), float(z.mean()), float(np.sqrt((z * z).mean())), z.size


def test_roulette_is_off_by_default_and_opt_in(ora):
    sc = scenes.config4(48, 48, 4, 2, 50)
    t = PathTracer(48, 48)
    t.set_spheres(sc.spheres)
    t.set_params(sc.params)
    t.reserve_passes(2)
    t.render_passes(2)
    ref, seg = ora.render(sc.spheres, sc.params, 2)
    assert np.array_equal(t.accum().view(np.uint32), ref.view(np.uint32)) and t.stats().segments == seg
    t.reset()
    t.set_russian_roulette(3)
    t.render_passes(2)
    st = t.stats()
    assert st.segments < 0.5 * seg                     # the closed room: most of a path's 40 segments are gone
    assert not np.array_equal(t.accum().view(np.uint32), ref.view(np.uint32))
    t.set_russian_roulette(0)                          # and off again: the reference's estimator, bit for bit
    t.reset()
    t.render_passes(2)
    assert np.array_equal(t.accum().view(np.uint32), ref.view(np.uint32))
    assert t.lib.pt_set_option(t._ctx, abi.PT_OPT_RUSSIAN_ROULETTE, -1) == abi.PT_ERR_INVALID
    t.set_russian_roulette(2)
    t.set_count_work(True)
    t.set_geometry_path(abi.PT_GEOM_GRID)
    assert t.lib.pt_render_passes(t._ctx, 1) in (abi.PT_OK, abi.PT_ERR_INVALID)  # 9 spheres: no grid, no twin, renders
    t.close()


def test_roulette_keeps_the_cosine_lobe_expectation():
    """Ground sphere alone under the sky: E[pixel] = albedo * mix(white, blue, 1/2 + n.y/3) in closed
    form.  With roulette from the first bounce on, a quarter of the paths end at the ground (q = .75)
    and the rest carry 1/q — the mean must not move."""
    w, h, passes, spp = 64, 36, 24, 64
    sc = scenes.default_scene(w, h, spp=spp, max_depth=8)
    sc.spheres = sc.spheres[:1]
    got, _ = pass_means(sc, passes, roulette=1)
    mean, se = got.mean(0), got.std(0, ddof=1) / np.sqrt(passes)
    ground = sc.spheres
    c, r, alb = ground["center"][0].astype(np.float64), float(ground["radius"][0]), ground["albedo"][0].astype(np.float64)
    zs = []
    for py in range(h):
        for px in range(0, w, 3):
            o, d = analytic.pixel_rays(w, h, px, py, 4)
            val, hit = analytic.diffuse_first_bounce(o, d, c, r, alb)
            o2, d2 = analytic.pixel_rays(w, h, px, py + 2, 2)
            if not (hit.all() and not np.isnan(analytic.hit_sphere(o2, d2, c, r)).any()):
                continue
            zs.append((mean[py, px] - val.mean(0)) / np.maximum(se[py, px], 1e-12))
    zmax, zmean, zrms, count = z_stats(np.concatenate(zs))
    assert count > 300
    assert zrms < 1.35 and abs(zmean) < 0.25 and zmax < 5.5, (zmax, zmean, zrms, count)


def test_roulette_keeps_the_glass_tree_expectation():
    """The lone WHITE glass sphere over the sky: throughput stays 1, so q = 1 and no path dies — but
    every bounce now draws one more random number, which moves the stream under the Schlick coin
    flips.  The exhaustive reflect/refract tree of depth 8 (tests/analytic.py) still has to be the
    mean: the extra draws must not correlate with the decisions they sit between."""
    w, h, passes, spp = 96, 54, 16, 64
    sc = scenes.default_scene(w, h, spp=spp, max_depth=8)
    glass = sc.spheres[3:4].copy()
    assert int(glass["type"][0]) == abi.PT_GLASS
    sc.spheres = glass
    got, seg_rr = pass_means(sc, passes, roulette=1)
    _, seg = pass_means(sc, 2, roulette=0)
    mean, se = got.mean(0), got.std(0, ddof=1) / np.sqrt(passes)
    c, r = glass["center"][0].astype(np.float64), float(glass["radius"][0])
    zs, pixels = [], 0
    for py in range(h):
        for px in range(w):
            o, d = analytic.pixel_rays(w, h, px, py, 6)
            inside = ~np.isnan(analytic.hit_sphere(o, d, c, r))
            o2, d2 = analytic.pixel_rays(w, h, px - 1, py - 1, 2)
            o3, d3 = analytic.pixel_rays(w, h, px + 1, py + 1, 2)
            ring = ~np.isnan(analytic.hit_sphere(np.concatenate([o2, o3]), np.concatenate([d2, d3]), c, r))
            if not (inside.all() and ring.all()):
                continue
            pixels += 1
            m1, _ = analytic.glass_tree(o, d, c, r, 1.5, 8)
            zs.append((mean[py, px] - m1.mean(0)) / np.maximum(se[py, px], 1e-9))
    zmax, zmean, zrms, count = z_stats(np.concatenate(zs))
    assert pixels > 60
    # standard errors from 16 passes are themselves noisy (Student t, 15 degrees of freedom): wider bands than a known sigma
    assert zmax < 7.0 and abs(zmean) < 0.4 and 0.75 < zrms < 1.45, (zmax, zmean, zrms, count)
    assert abs(seg_rr / passes - seg / 2) < 0.02 * seg / 2  # nothing dies at q = 1: the same work per pass


def test_roulette_keeps_the_room_expectation_and_cuts_its_work(ora):
    """BASELINE config 4 (closed room, emissive sphere, depth 50: 40 segments per path on average).
    Roulette after 3 bounces against the ORACLE's roulette-free frame: per-pixel difference of the means
    over the standard errors of both sides (independent passes), and the room's mean radiance."""
    w, h, passes, spp = 40, 40, 24, 32
    sc = scenes.config4(w, h, spp, passes, 50)
    got, seg_rr = pass_means(sc, passes, roulette=3)
    ref = []
    seg = 0
    p = sc.params.copy()
    p.time_step = abi.PT_TIME_STEP_DECORRELATED
    for k in range(passes):
        q = p.copy()
        q.first_pass = k
        a, s_ = ora.render(sc.spheres, q, 1)
        ref.append(a[..., :3].astype(np.float64) / spp)
        seg += s_
    ref = np.stack(ref)
    m_rr, m_ref = got.mean(0), ref.mean(0)
    se = np.sqrt(got.var(0, ddof=1) / passes + ref.var(0, ddof=1) / passes)
    ok = se > 1e-9
    z = ((m_rr - m_ref) / np.where(ok, se, 1.0))[ok]
    zmax, zmean, zrms, count = z_stats(z)
    assert count > 0.9 * w * h * 3
    assert zmax < 6.5 and abs(zmean) < 0.15 and 0.8 < zrms < 1.3, (zmax, zmean, zrms, count)
    # the frame's mean radiance, a 4800-value average: agreement to a fraction of a per cent
    rel = abs(m_rr.mean() - m_ref.mean()) / m_ref.mean()
    assert rel < 0.01, rel
    # ... for a third of the work or less (wall albedo .73: a path now ends after ~6 more bounces, not at depth 50)
    assert seg_rr < 0.4 * seg, (seg_rr, seg)
