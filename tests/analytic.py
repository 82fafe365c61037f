"""RNG-free float64 expectations of small path trees of static/shader.frag, written from the
shader's formulas and independent of the oracle's code: what the ESTIMATOR the oracle (and the
HIP path) implements must converge to, if its sampling routines draw what the shader means them
to draw.  Used by tests/test_reference_pins.py.

  * camera: State::default / update_pipeline (src/state.rs:98-125, :319-347) in doubles
  * sky: background(), static/shader.frag:289-294
  * diffuse first bounce off a convex surface under the sky: scatter's DIFFUSE branch (:212-229)
    draws normalize(n + unit vector) — a cosine-weighted direction about n, whose mean is 2n/3 —
    and background() is linear in the direction's y, so
        E[radiance] = albedo * mix(white, blue, 1/2 + n.y / 3)               (closed form)
  * glass: scatter's GLASS branch (:250-282) picks reflect with probability R (Schlick on the
    ratio, or 1 beyond the critical angle), refract otherwise; the expectation over those
    choices is a binary tree of depth u_max_depth whose leaves are sky colours, or the
    throughput when the depth runs out (:300, :338) — evaluated exhaustively, level by level.
"""
import math

import numpy as np

MIN_T, MAX_T = 0.001, 1e5
BLUE = np.array([0.5, 0.7, 1.0])


def default_camera(width, height):
    """origin, lower-left corner, horizontal, vertical of State::default (lens radius 0)"""
    origin = np.array([0.0, 0.0, 1.0])
    yaw, pitch = math.radians(-90.0), 0.0
    front = np.array([math.cos(yaw) * math.cos(pitch), math.sin(pitch), math.sin(yaw) * math.cos(pitch)])
    w = origin - (origin + front)
    w /= np.linalg.norm(w)
    u = np.cross([0.0, 1.0, 0.0], w)
    u /= np.linalg.norm(u)
    v = np.cross(w, u)
    focus = 0.75
    vh = 2.0 * math.tan((math.pi / 3.0) / 2.0)
    vw = vh * (width / height)
    horizontal, vertical = focus * vw * u, focus * vh * v
    llc = origin - horizontal / 2 - vertical / 2 - focus * w
    return origin, llc, horizontal, vertical


def pixel_rays(width, height, px, py, n_sub):
    """camera rays through an n_sub x n_sub midpoint grid of the pixel's sample footprint
    [centre, centre + 1 px) (static/shader.frag:365-369: the jitter is ADDED to the centre)"""
    origin, llc, hz, vt = default_camera(width, height)
    j = (np.arange(n_sub) + 0.5) / n_sub
    s = ((px + 0.5) + j[None, :]) / width
    t = ((py + 0.5) + j[:, None]) / height
    s, t = np.broadcast_to(s, (n_sub, n_sub)).ravel(), np.broadcast_to(t, (n_sub, n_sub)).ravel()
    d = llc[None, :] + s[:, None] * hz[None, :] + t[:, None] * vt[None, :] - origin[None, :]
    return np.broadcast_to(origin, d.shape).copy(), d


def sky(d):
    t = 0.5 * (d[:, 1] / np.linalg.norm(d, axis=1) + 1.0)
    return (1.0 - t)[:, None] + t[:, None] * BLUE[None, :]


def hit_sphere(o, d, c, r):
    """hit_sphere :145-173 in doubles: root in (MIN_T, MAX_T), nearest first; returns t (nan = miss)"""
    oc = o - c[None, :]
    a = np.einsum("ij,ij->i", d, d)
    hb = np.einsum("ij,ij->i", oc, d)
    cc = np.einsum("ij,ij->i", oc, oc) - r * r
    disc = hb * hb - a * cc
    sq = np.sqrt(np.maximum(disc, 0.0))
    near, far = (-hb - sq) / a, (-hb + sq) / a
    t = np.where((near >= MIN_T) & (near <= MAX_T), near, np.where((far >= MIN_T) & (far <= MAX_T), far, np.nan))
    return np.where(disc < 0, np.nan, t)


def diffuse_first_bounce(o, d, c, r, albedo):
    """E[radiance] of camera rays (o, d) over a lone convex diffuse sphere under the sky: rays
    that miss see the sky; rays that hit scatter once (the bounce cannot return to a convex
    sphere) and see the cosine-lobe mean of the sky, times the albedo."""
    t = hit_sphere(o, d, c, r)
    hit = ~np.isnan(t)
    p = o + d * np.where(hit, t, 0.0)[:, None]
    n = (p - c[None, :]) / r
    front = np.einsum("ij,ij->i", d, n) < 0
    n = np.where(front[:, None], n, -n)
    tbar = 0.5 + n[:, 1] / 3.0
    lobe = (1.0 - tbar)[:, None] + tbar[:, None] * BLUE[None, :]
    return np.where(hit[:, None], np.asarray(albedo)[None, :] * lobe, sky(d)), hit


def glass_tree(o, d, c, r, ri, max_depth):
    """First and second moment of the radiance of camera rays over a lone white glass sphere
    under the sky, over all reflect / refract choices down to max_depth segments."""
    n_rays = len(o)
    weight = np.ones(n_rays)
    owner = np.arange(n_rays)
    m1 = np.zeros((n_rays, 3))
    m2 = np.zeros((n_rays, 3))

    def leaf(idx, w, val):
        np.add.at(m1, idx, w[:, None] * val)
        np.add.at(m2, idx, w[:, None] * val * val)

    for _ in range(max_depth):
        t = hit_sphere(o, d, c, r)
        miss = np.isnan(t)
        leaf(owner[miss], weight[miss], sky(d[miss]))
        o, d, t, weight, owner = o[~miss], d[~miss], t[~miss], weight[~miss], owner[~miss]
        if not len(o):
            break
        p = o + d * t[:, None]
        n = (p - c[None, :]) / r
        front = np.einsum("ij,ij->i", d, n) < 0
        n = np.where(front[:, None], n, -n)
        ratio = np.where(front, 1.0 / ri, ri)
        ud = d / np.linalg.norm(d, axis=1)[:, None]
        cos_t = np.minimum(-np.einsum("ij,ij->i", ud, n), 1.0)
        sin_t = np.sqrt(np.maximum(1.0 - cos_t * cos_t, 0.0))
        r0 = ((1.0 - ratio) / (1.0 + ratio)) ** 2  # reflectance(), :204-207, fed with the RATIO
        refl = np.where(ratio * sin_t > 1.0, 1.0, r0 + (1.0 - r0) * (1.0 - cos_t) ** 5)
        dn = np.einsum("ij,ij->i", ud, n)
        d_refl = ud - 2.0 * dn[:, None] * n
        k = 1.0 - ratio * ratio * (1.0 - dn * dn)
        d_refr = ratio[:, None] * ud - (ratio * dn + np.sqrt(np.maximum(k, 0.0)))[:, None] * n
        keep = refl < 1.0
        o = np.concatenate([p, p[keep]])
        d = np.concatenate([d_refl, d_refr[keep]])
        weight = np.concatenate([weight * refl, (weight * (1.0 - refl))[keep]])
        owner = np.concatenate([owner, owner[keep]])
    leaf(owner, weight, np.ones((len(owner), 3)))  # depth exhausted: `return color`, :338 (white glass: throughput 1)
    return m1, m2
