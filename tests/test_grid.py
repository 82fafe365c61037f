"""The uniform grid of PT_GEOM_GRID (ray_tracer_webgl_amd/csrc/pt_grid.hpp), checked on the host.

The grid kernels may skip a sphere only if the reference's hit_sphere (static/shader.frag:145-173)
could not have produced the winning root, so these tests check
  (i)   the structure pt_set_spheres uploads (registration with the inflation delta_g),
  (ii)  the error bound the inflation rests on: the fp32 root of the literal test lies within
        delta(D) of the sphere's surface (float64 check of float32 emulated roots),
  (iii) a numpy emulation of the kernel's walk — same formulas, fp32, entry test, 3D-DDA, early
        termination — against a brute-force emulation of hit_world over the whole list: the
        walk must return the same (root, index) pair for every regular ray.
No GPU needed: pt_build_grid is the host half of the C ABI.
"""
import ctypes as C

import numpy as np
import pytest

from ray_tracer_webgl_amd import _lib, abi, scenes
from test_bvh import SCENES, f32, fma, literal_disc, random_field, rays_for

PAD = 0xFFFFFFFF
U = 2.0 ** -24
MIN_T = np.float32(0.001)
MAX_T = np.float32(1e5)


def build(spheres, runs=False):
    """runs=True: the layout of the kernels that gather entries from global memory (pt_build_grid_runs,
    include/ptrace_dev.h: the cells' runs in Morton order of their cells)"""
    lib = _lib.load()
    build_fn = lib.pt_build_grid
    if runs:
        build_fn = lib.pt_build_grid_runs
        build_fn.restype, build_fn.argtypes = lib.pt_build_grid.restype, lib.pt_build_grid.argtypes
    ptr, n, keep = abi.spheres_as_ctypes(spheres)
    counts = np.zeros(8, np.uint32)
    geom = np.zeros(12, np.float32)
    margin = np.zeros(4, np.float32)
    dg = C.c_float(0)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = build_fn(ptr, n, vp(counts), vp(geom), vp(margin), C.byref(dg), None, 0, None, 0, None, 0)
    if rc != 0:
        return rc, None
    nc = int(counts[0]) * int(counts[1]) * int(counts[2])
    cells = np.zeros(nc, np.uint32)
    entries = np.zeros((counts[5], 4), np.float32)
    index = np.zeros(counts[5], np.uint32)
    rc = build_fn(ptr, n, vp(counts), vp(geom), vp(margin), C.byref(dg), vp(cells), cells.size, vp(entries),
                  entries.size, vp(index), index.size)
    consts = np.zeros(10, np.float32)  # what pt_render_passes copies into the launch arguments
    assert lib.pt_grid_walk_constants(ptr, n, vp(consts)) == 0
    return rc, dict(r2_near=consts[0], lo_n=consts[1:4].copy(), hi_n=consts[4:7].copy(), inv_h=consts[7:10].copy(),
                    n=counts[:3].astype(np.int64), n_cell_entries=int(counts[3]), n_always=int(counts[4]),
                    n_entries=int(counts[5]), max_entries=int(counts[6]), nonempty=int(counts[7]), lo=geom[0:3], h=geom[3:6],
                    hi=geom[6:9], c0=geom[9:12], s0=margin[0], rmin=margin[1], rmax=margin[2], d_near=margin[3],
                    delta_g=np.float32(dg.value), cells=cells, entries=entries, index=index)


def delta_of(rmin, rmax, D):
    return np.sqrt(rmin * rmin + 32.0 * U * D * D) - rmin + 10.0 * U * rmax


GRID_SCENES = dict(SCENES)
GRID_SCENES["mixed_radii"] = lambda: np.concatenate(
    [random_field(200, 11, extent=10.0, rmax=0.3, giants=1), random_field(6, 12, extent=8.0, rmax=4.0, giants=0)])
GRID_SCENES["flat"] = lambda: _flat()
# the same fields far from the origin: the walk's slab and plane times are evaluated in ABSOLUTE
# coordinates (fma(plane, 1/d, -(o * 1/d))), so their rounding scales with |c0|, not with the scene
GRID_SCENES["field300_far"] = lambda: _moved(SCENES["field300"](), (3.0e3, -1.5e3, 2.0e3))
GRID_SCENES["flat_far"] = lambda: _moved(_flat(), (-2.5e3, 40.0, 1.0e3))
GRID_SCENES["config2_far"] = lambda: _moved(SCENES["config2"](), (500.0, 0.0, -300.0), giants_too=True)


def _moved(sph, by, giants_too=False):
    """translate a scene (float32 arithmetic on the centres, like a host that builds it there)"""
    s = sph.copy()
    big = np.abs(s["radius"]) > 100.0
    sel = np.ones(len(s), bool) if giants_too else ~big
    s["center"][sel] = (s["center"][sel].astype(np.float64) + np.asarray(by)).astype(np.float32)
    if not giants_too:  # keep far-out giants below the field
        s["center"][big, 0] += np.float32(by[0]); s["center"][big, 2] += np.float32(by[2])
    return s


def _flat():
    s = random_field(400, 13, extent=12.0, rmax=0.25, giants=1)
    s["center"][1:, 1] = 0.2
    return s


@pytest.mark.parametrize("name", sorted(GRID_SCENES))
def test_structure(name):
    sph = GRID_SCENES[name]()
    rc, g = build(sph)
    assert rc == 0
    n = len(sph)
    c = np.asarray(sph["center"], np.float64)
    r = np.abs(np.asarray(sph["radius"], np.float64))
    cells, entries, index = g["cells"], g["entries"], g["index"]
    first, count = (cells & 0xFFFFFF).astype(np.int64), (cells >> 24).astype(np.int64)
    # cell records tile the gridded part of the entry array, in cell order, without padding
    assert np.array_equal(first, np.concatenate([[0], np.cumsum(count)[:-1]]))
    assert count.sum() == g["n_cell_entries"] and count.max() == g["max_entries"] and (count > 0).sum() == g["nonempty"]
    assert (g["n_entries"] - g["n_cell_entries"]) % 4 == 0 and g["n_entries"] >= g["n_cell_entries"] + g["n_always"]
    # entries are exact copies of the list records; padding (only behind the always-tested spheres)
    # can never pass the literal test
    real = index != PAD
    assert real[:g["n_cell_entries"]].all() and np.all(np.isneginf(entries[~real, 3]))
    cs = np.asarray(sph["center"], np.float32)
    rr = np.asarray(sph["radius"], np.float32)
    assert np.array_equal(entries[real, :3], cs[index[real]])
    assert np.array_equal(entries[real, 3], (rr * rr)[index[real]])
    # every sphere is gridded or tested for every ray, never both
    always = index[g["n_cell_entries"]:]
    always = always[always != PAD]
    assert len(always) == g["n_always"] and np.all(np.diff(always.astype(np.int64)) > 0)
    gridded = np.unique(index[:g["n_cell_entries"]][index[:g["n_cell_entries"]] != PAD])
    assert not set(gridded.tolist()) & set(always.tolist())
    assert sorted(gridded.tolist() + always.tolist()) == list(range(n))
    # the margin's reference data
    c0, s0 = g["c0"].astype(np.float64), float(g["s0"])
    reach = np.linalg.norm(c[gridded] - c0, axis=1) + r[gridded]
    assert reach.max() <= s0 and r[gridded].min() >= g["rmin"] and r[gridded].max() <= g["rmax"]
    assert g["d_near"] >= 3.0 * s0
    # the registration inflation covers delta(d_near) as the KERNEL bounds it (25 % slack on E') ...
    dk = np.sqrt(float(g["rmin"]) ** 2 + 40.0 * U * float(g["d_near"]) ** 2) - float(g["rmin"]) + 16.0 * U * float(g["rmax"])
    assert g["delta_g"] >= dk and g["delta_g"] >= delta_of(float(g["rmin"]), float(g["rmax"]), float(g["d_near"]))
    # ... and is not absurd next to a cell
    assert g["delta_g"] < 0.6 * float(g["h"].min()) or name in ("clumps",) or name.endswith("_far")
    # registration: a sphere is an entry of EVERY cell its box inflated by ITS delta touches: the kernel's bound with the
    # sphere's own radius in the square root (sqrt(r^2 + x) - r decreases with r) and its own distance bound D_i = d_near - s0
    # + |C_i - c0| (a walking ray starts within d_near - s0 of c0: the kernel's near test, r2_near) + what delta_g carries
    # beyond that term (16 u rmax and the walk's rounding, eps_dda); at r = rmin, |C - c0| = s0 - r that is delta_g itself
    d_near = float(g["d_near"])
    assert float(g["r2_near"]) <= (d_near - s0) ** 2
    dg_walk = float(g["delta_g"]) - (np.sqrt(float(g["rmin"]) ** 2 + 40.0 * U * d_near ** 2) - float(g["rmin"]))
    assert dg_walk >= 16.0 * U * float(g["rmax"])
    D_i = np.minimum(d_near, d_near - s0 + np.linalg.norm(c - c0, axis=1))
    delta_i = np.minimum(float(g["delta_g"]), np.sqrt(r * r + 40.0 * U * D_i ** 2) - r + dg_walk)
    assert np.all(delta_i[gridded] >= np.sqrt(r[gridded] ** 2 + 40.0 * U * D_i[gridded] ** 2) - r[gridded] + 16.0 * U * float(g["rmax"]) - 1e-12)
    lo, h, nn = g["lo"].astype(np.float64), g["h"].astype(np.float64), g["n"]
    member = {}
    for cell in np.nonzero(count)[0]:
        member[int(cell)] = set(index[first[cell]:first[cell] + count[cell]].tolist())
    rng = np.random.default_rng(0)
    for i in rng.choice(gridded, min(len(gridded), 400), replace=False):
        a = np.clip(np.floor((c[i] - r[i] - delta_i[i] - lo) / h), 0, nn - 1).astype(np.int64)
        b = np.clip(np.floor((c[i] + r[i] + delta_i[i] - lo) / h), 0, nn - 1).astype(np.int64)
        for z in range(a[2], b[2] + 1):
            for y in range(a[1], b[1] + 1):
                for x in range(a[0], b[0] + 1):
                    assert int(i) in member.get((z * nn[1] + y) * nn[0] + x, ()), (name, i, x, y, z)
    # all registered boxes lie inside [lo, hi]: near rays need no inflation of the entry test
    assert np.all(c[gridded] - r[gridded, None] - float(g["delta_g"]) >= lo - 1e-9)
    assert np.all(c[gridded] + r[gridded, None] + float(g["delta_g"]) <= g["hi"].astype(np.float64) + 1e-9)
    # cells of about one sphere each, a bounded number of copies
    assert g["n_cell_entries"] <= 8 * len(gridded) + 64 or name.endswith("_far")


def test_giants_and_big_spheres_are_tested_for_every_ray():
    rc, g = build(GRID_SCENES["config2"]())
    assert rc == 0
    always = g["index"][g["n_cell_entries"]:]
    assert sorted(always[always != PAD].tolist()) == [0, 481, 482, 483]  # the ground and the three r = 1 spheres
    assert g["n"][1] == 1 and g["n"][0] >= 16 and g["n"][2] >= 16             # one layer of cells over the flat field
    assert g["s0"] < 20.0


def test_a_field_too_far_from_the_origin_for_fp32_planes_gets_no_grid():
    """At |c0| = 3e5 a float32 coordinate resolves 0.03 — a sixth of config 2's sphere radius: the
    registration inflation (which carries |c0| since round 3) would put every sphere into dozens of
    cells, the builder gives up, and the scene runs through the hierarchy / the list instead."""
    rc, g = build(_moved(SCENES["config2"](), (1.0e5, 0.0, -3.0e5), giants_too=True))
    assert rc == abi.PT_ERR_NOT_READY
    rc, near = build(SCENES["config2"]())
    rc2, far = build(GRID_SCENES["config2_far"]())
    assert rc == 0 and rc2 == 0 and far["delta_g"] > near["delta_g"]  # the |c0| term at work


def test_scenes_without_a_grid():
    assert build(scenes.default_scene(64, 36, 1, 8).spheres)[0] == abi.PT_ERR_NOT_READY  # 9 spheres
    bad = random_field(64, 5)
    bad["center"][7, 1] = np.inf
    assert build(bad)[0] == abi.PT_ERR_NOT_READY


# ---- (ii) the root of the literal test stays within delta(D) of the sphere ------------------------
def exact_root(o, d, c, r2):
    """candidate root of hit_sphere as the kernels evaluate it (fp32), per ray/sphere pair"""
    oc = [f32(o[:, k] - c[:, k]) for k in range(3)]
    hb = fma(oc[2], d[:, 2], fma(oc[1], d[:, 1], f32(oc[0] * d[:, 0])))
    cc = fma(oc[2], oc[2], fma(oc[1], oc[1], fma(oc[0], oc[0], -r2)))
    a = fma(d[:, 2], d[:, 2], fma(d[:, 1], d[:, 1], f32(d[:, 0] * d[:, 0])))
    disc = fma(-a, cc, f32(hb * hb))
    # the kernels' candidate test, as they take it: one bit of sign arithmetic (pt_grid_walk.hpp pass_bit)
    bits = lambda x: np.asarray(x, np.float32).view(np.uint32)
    ok = (((bits(cc) | bits(hb)) & ~bits(f32(disc + np.float32(0.0)))) >> np.uint32(31)).astype(bool)
    with np.errstate(invalid="ignore", divide="ignore"):
        sq = np.sqrt(np.where(ok, disc, 0).astype(np.float32))
        v = f32(f32(-hb - sq) / a)
        far = f32(f32(-hb + sq) / a)
    v = np.where(v < MIN_T, far, v)
    ok &= ~(v < MIN_T)
    return v, ok


def test_root_stays_within_delta_of_the_sphere():
    rng = np.random.default_rng(42)
    worst = 0.0
    n = 200000
    for scale in (1.0, 30.0, 1000.0):
        r = f32(rng.choice([0.05, 0.2, 1.0, 7.0], n) * rng.uniform(0.5, 1.5, n))
        c = f32(rng.uniform(-1, 1, (n, 3)) * scale)
        o = f32(c + rng.normal(size=(n, 3)) * rng.choice([1.0, 3.0, 30.0], (n, 1)) * r[:, None] * rng.choice([1, 10, 100], (n, 1)))
        # aim at the rim: where rounding decides whether and where the ray hits
        u = rng.normal(size=(n, 3))
        u /= np.linalg.norm(u, axis=1)[:, None]
        target = c + u * (r * rng.choice([0.999999, 1.0, 1.000001, 0.9, 0.3], n))[:, None]
        d = f32((target - o) * rng.choice([1.0, 0.01, 20.0], (n, 1)))
        v, ok = exact_root(o, d, c, f32(r * r))
        aa = np.einsum("ij,ij->i", d.astype(np.float64), d.astype(np.float64))
        ok &= (aa > 1e-12) & (aa < 1e6)
        P = o.astype(np.float64) + d.astype(np.float64) * v.astype(np.float64)[:, None]
        dist = np.linalg.norm(P - c.astype(np.float64), axis=1) - r.astype(np.float64)
        D = np.linalg.norm(o.astype(np.float64) - c.astype(np.float64), axis=1)
        bound = np.sqrt(r.astype(np.float64) ** 2 + 32.0 * U * D * D) - r + 10.0 * U * r
        ratio = (np.abs(dist) / bound)[ok]
        assert ok.sum() > n // 20
        worst = max(worst, float(ratio.max()))
    # the analysis' constants are worst-case; the measured worst sits well inside
    assert worst < 0.75, worst


# ---- (iii) the kernel's walk returns hit_world's pair ---------------------------------------------
def brute_force(o, d, sph):
    cs = np.asarray(sph["center"], np.float32)
    r = np.asarray(sph["radius"], np.float32)
    disc, hb, cc = literal_disc(o, d, cs, f32(r * r))
    a = fma(d[:, 2], d[:, 2], fma(d[:, 1], d[:, 1], f32(d[:, 0] * d[:, 0])))[:, None]
    ok = ~(disc < 0) & ~((cc > 0) & (hb >= 0))
    with np.errstate(invalid="ignore", divide="ignore"):
        sq = np.sqrt(np.where(ok, disc, 0).astype(np.float32))
        v = f32(f32(-hb - sq) / a)
        far = f32(f32(-hb + sq) / a)
    v = np.where(v < MIN_T, far, v)
    ok &= ~(v < MIN_T) & (v <= MAX_T)
    v = np.where(ok, v, np.float32(np.inf))
    best = v.min(1)
    # ties -> the LATER sphere of the list (static/shader.frag:159)
    idx = np.where(v == best[:, None], np.arange(v.shape[1])[None, :], -1).max(1)
    idx = np.where(np.isinf(best), -1, idx)
    return np.where(np.isinf(best), MAX_T, best), idx


def walk(g, o, d, sph):
    """pt_trace_kernel_grid's PHASE 1, formula for formula in fp32 (grid_walk, pt_grid_walk.hpp) with the
    entry constants the launch really uses (pt_grid_walk_constants): returns closest, sphere index,
    and the number of entries looked at per ray; far rays that enter the box take the literal loop"""
    n = len(o)
    ent, index = g["entries"], g["index"].astype(np.int64)
    closest = np.full(n, MAX_T, np.float32)
    hit = np.full(n, -1, np.int64)
    looked = np.zeros(n, np.int64)

    def test_entries(rays, pos):
        """literal test + exact evaluation of entries `pos` (one per ray in `rays`)"""
        nonlocal closest, hit
        v, ok = exact_root(o[rays], d[rays], ent[pos, :3], ent[pos, 3])
        idx = index[pos]
        ok &= idx != PAD
        cur, cur_hit = closest[rays], hit[rays]
        wins = ok & ((v < cur) | ((v == cur) & ((cur_hit < 0) | (idx > cur_hit))))
        closest[rays] = np.where(wins, v, cur)
        hit[rays] = np.where(wins, idx, cur_hit)

    allr = np.arange(n)
    for k in range(g["n_cell_entries"], g["n_entries"]):
        test_entries(allr, np.full(n, k))
    # ---- the kernel's entry arithmetic, operation for operation (pt_grid_walk.hpp) -------------------
    # (one liberty: 1/d is the correctly rounded quotient here, v_rcp_f32 — good to 1 ulp — in the
    # kernel; the error budget of pt_grid.hpp counts the reciprocal as a rounded operand either way)
    with np.errstate(divide="ignore"):
        inv = np.clip(f32(1.0) / d, f32(-1e18), f32(1e18)).astype(np.float32)
    pos_dir = inv > 0
    H = np.broadcast_to(g["h"][None, :], d.shape)
    td = f32(H * np.abs(inv))
    p = f32(o - g["c0"][None, :])
    r2 = fma(p[:, 2], p[:, 2], fma(p[:, 1], p[:, 1], f32(p[:, 0] * p[:, 0])))
    near = r2 <= g["r2_near"]
    mm = np.where(near, np.float32(0), f32(np.float32(1.7e-3) * f32(np.sqrt(r2) + g["s0"]))).astype(np.float32)
    oi = f32(o * inv)                                              # oix = o.x * ix
    lo_m = f32(g["lo_n"][None, :] - mm[:, None])                   # K.grid_lo_n - mm
    hi_m = f32(g["hi_n"][None, :] + mm[:, None])
    t1 = fma(lo_m, inv, -oi)                                       # fma(lo_n - mm, ix, -oix)
    t2 = fma(hi_m, inv, -oi)
    tn = np.maximum(np.maximum(np.minimum(t1[:, 0], t2[:, 0]), np.minimum(t1[:, 1], t2[:, 1])),
                    np.maximum(np.minimum(t1[:, 2], t2[:, 2]), np.float32(0)))
    tf = np.minimum(np.minimum(np.maximum(t1[:, 0], t2[:, 0]), np.maximum(t1[:, 1], t2[:, 1])), np.maximum(t1[:, 2], t2[:, 2]))
    enter = tn <= np.minimum(tf, closest)
    literal = enter & ~near
    active = enter & near
    nn = g["n"]
    LO = np.broadcast_to(g["lo"][None, :], d.shape)
    # the cell that holds the entry point: (fma(d, tn, o) - lo) * inv_h, floor, clamp
    fcell = f32(f32(fma(d, np.broadcast_to(tn[:, None], d.shape), o) - LO) * g["inv_h"][None, :])
    fcell = np.where(active[:, None], fcell, np.float32(0))  # rays that do not walk: anything finite
    cell3 = np.clip(np.floor(fcell).astype(np.int64), 0, nn[None, :] - 1)
    # far planes of that cell and their crossing times, never before the entry time: fma(b, ix, -oix)
    bnd = fma(f32(cell3 + pos_dir), H, LO)
    tm = np.maximum(fma(bnd, inv, -oi), tn[:, None])
    rem = np.where(pos_dir, nn[None, :] - 1 - cell3, cell3) + 1
    first, count = (g["cells"] & 0xFFFFFF).astype(np.int64), (g["cells"] >> 24).astype(np.int64)
    for _ in range(int(nn.sum()) + 4):
        rays = np.nonzero(active)[0]
        if not len(rays):
            break
        cidx = (cell3[rays, 2] * nn[1] + cell3[rays, 1]) * nn[0] + cell3[rays, 0]
        tmin = tm[rays].min(1)
        isx = tm[rays, 0] == tmin
        isy = ~isx & (tm[rays, 1] == tmin)
        ax = np.where(isx, 0, np.where(isy, 1, 2))
        t_exit = tmin
        tm[rays, ax] = f32(tm[rays, ax] + td[rays, ax])
        rem[rays, ax] -= 1
        out = rem[rays, ax] == 0
        cell3[rays, ax] += np.where(pos_dir[rays, ax], 1, -1)
        # all entries of the cell
        cmax = int(count[cidx].max()) if len(cidx) else 0
        for k in range(cmax):
            sel = count[cidx] > k
            test_entries(rays[sel], first[cidx[sel]] + k)
            looked[rays[sel]] += 1
        done = out | (closest[rays] < t_exit)
        active[rays[done]] = False
    return closest, hit, looked, literal


def dda_discrepancy(g, o, d, rng):
    """The 3D-DDA of pt_grid_walk.hpp in fp32 (entry slab, entry cell, plane times, the running sums) run to the end of the
    grid, against the same ray and the same planes in float64: the largest distance, along the stepping axis, between where the
    walk BELIEVES a cell boundary is crossed (its fp32 time, put into the exact ray) and the boundary — and by how much the
    entry point lies outside the entry cell.  That is what eps_dda (pt_grid.hpp) must cover.  The reciprocal is perturbed by
    +-1 ulp at random: v_rcp_f32 is good to 1 ulp, numpy's quotient is correctly rounded.  Returns (worst crossing, worst entry)
    in units of u (n_sum + 8) (d_near + diag + |c0|_inf), the builder's eps_dda without its safety factor."""
    n = len(o)
    with np.errstate(divide="ignore"):
        inv = np.clip(f32(1.0) / d, f32(-1e18), f32(1e18)).astype(np.float32)
    wob = rng.integers(-1, 2, inv.shape)
    inv = np.where(wob > 0, np.nextafter(inv, np.float32(np.inf)), np.where(wob < 0, np.nextafter(inv, np.float32(-np.inf)), inv)).astype(np.float32)
    pos_dir = inv > 0
    H = np.broadcast_to(g["h"][None, :], d.shape)
    td = f32(H * np.abs(inv))
    p = f32(o - g["c0"][None, :])
    r2 = fma(p[:, 2], p[:, 2], fma(p[:, 1], p[:, 1], f32(p[:, 0] * p[:, 0])))
    near = r2 <= g["r2_near"]
    oi = f32(o * inv)
    t1 = fma(np.broadcast_to(g["lo_n"][None, :], d.shape), inv, -oi)
    t2 = fma(np.broadcast_to(g["hi_n"][None, :], d.shape), inv, -oi)
    tn = np.maximum(np.maximum(np.minimum(t1[:, 0], t2[:, 0]), np.minimum(t1[:, 1], t2[:, 1])),
                    np.maximum(np.minimum(t1[:, 2], t2[:, 2]), np.float32(0)))
    tf = np.minimum(np.minimum(np.maximum(t1[:, 0], t2[:, 0]), np.maximum(t1[:, 1], t2[:, 1])), np.maximum(t1[:, 2], t2[:, 2]))
    active = near & (tn <= tf) & np.all(np.abs(inv) < 1e17, axis=1)   # (an axis the ray never crosses has no boundary to miss)
    nn = g["n"]
    LO = np.broadcast_to(g["lo"][None, :], d.shape)
    fcell = f32(f32(fma(d, np.broadcast_to(tn[:, None], d.shape), o) - LO) * g["inv_h"][None, :])
    fcell = np.where(active[:, None], fcell, np.float32(0))
    cell3 = np.clip(np.floor(fcell).astype(np.int64), 0, nn[None, :] - 1)
    bnd = fma(f32(cell3 + pos_dir), H, LO)
    tm = np.maximum(fma(bnd, inv, -oi), tn[:, None])
    rem = np.where(pos_dir, nn[None, :] - 1 - cell3, cell3) + 1
    o64, d64, lo64, h64 = o.astype(np.float64), d.astype(np.float64), g["lo"].astype(np.float64), g["h"].astype(np.float64)
    # the entry point against the entry cell (per axis, how far outside)
    P = o64 + d64 * tn.astype(np.float64)[:, None]
    c_lo, c_hi = lo64[None, :] + cell3 * h64[None, :], lo64[None, :] + (cell3 + 1) * h64[None, :]
    outside = np.maximum(np.maximum(c_lo - P, P - c_hi), 0.0).max(1)
    # rays that start inside the grid enter "at" t = 0 in their own cell; rays from outside enter through a face
    worst_entry = float(outside[active].max()) if active.any() else 0.0
    worst_cross = 0.0
    for _ in range(int(nn.sum()) + 4):
        rays = np.nonzero(active)[0]
        if not len(rays):
            break
        tmin = tm[rays].min(1)
        isx = tm[rays, 0] == tmin
        isy = ~isx & (tm[rays, 1] == tmin)
        ax = np.where(isx, 0, np.where(isy, 1, 2))
        # the boundary the walk crosses now, exactly; only crossings that lie ahead of the entry matter (the first plane
        # times are clamped to tn: a plane "crossed" at the entry time is the entry point's business, measured above)
        plane = lo64[ax] + (cell3[rays, ax] + pos_dir[rays, ax]) * h64[ax]
        at = o64[rays, ax] + d64[rays, ax] * tmin.astype(np.float64)
        off = np.abs(at - plane)
        real = tmin > tn[rays]
        if real.any():
            worst_cross = max(worst_cross, float(off[real].max()))
        tm[rays, ax] = f32(tm[rays, ax] + td[rays, ax])
        rem[rays, ax] -= 1
        cell3[rays, ax] += np.where(pos_dir[rays, ax], 1, -1)
        active[rays[rem[rays, ax] == 0]] = False
    diag = float(np.linalg.norm(g["hi"].astype(np.float64) - lo64))
    unit = U * (float(nn.sum()) + 8.0) * (float(g["d_near"]) + diag + float(np.abs(g["c0"]).max()))
    return worst_cross / unit, worst_entry / unit


@pytest.mark.parametrize("name", ["config2", "config5", "mixed_radii", "flat", "clumps", "field300", "field300_far", "flat_far", "config2_far"])
def test_the_walks_boundary_times_stay_within_the_rounding_budget_of_the_registration(name):
    """eps_dda (pt_grid.hpp) is 8 x u (n_sum + 8) (d_near + diag + |c0|) — a first-order bound of the DDA's accumulated
    rounding (derivation in the header) times a safety factor.  Measured here: the fp32 walk against float64 on bounce, camera
    and far-but-walking rays, scenes at the origin and thousands of units away from it: the worst boundary discrepancy is
    0.03 ... 0.065 of the bound WITHOUT its factor (asserted: below a quarter of it, i.e. the factor leaves more than 32 x).
    The entry point may lie outside its (clamped) entry cell by the slab's deliberate widening, 1e-6 (d_near + |c0|) — no
    sphere's box is out there; it stays below the bound without its factor as well."""
    sph = GRID_SCENES[name]()
    rc, g = build(sph)
    assert rc == 0
    rng = np.random.default_rng(5)
    worst = [0.0, 0.0]
    for seed in range(3):
        o, d = rays_for(sph, 4000, seed)
        a = np.einsum("ij,ij->i", d.astype(np.float64), d.astype(np.float64))
        ok = (a > 1e-12) & (a < 1e6)
        wc, we = dda_discrepancy(g, o[ok], d[ok], rng)
        worst = [max(worst[0], wc), max(worst[1], we)]
    assert 0.0 < worst[0] < 0.25 and worst[1] < 1.0, worst


@pytest.mark.parametrize("name", ["config2", "field300", "clumps", "field17_no_giant", "config5", "mixed_radii", "flat",
                                  "field300_far", "flat_far", "config2_far"])
def test_walk_returns_the_pair_hit_world_returns(name):
    sph = GRID_SCENES[name]()
    rc, g = build(sph)
    assert rc == 0
    n_rays = 3000 if len(sph) < 2000 else 400
    total_hits = total_looked = total_lit = 0
    for seed in range(4):
        o, d = rays_for(sph, n_rays, seed)
        a = np.einsum("ij,ij->i", d.astype(np.float64), d.astype(np.float64))
        ok = (a > 1e-12) & (a < 1e6)  # the kernel's regular rays; the others take the literal loop
        o, d = o[ok], d[ok]
        ref_t, ref_i = brute_force(o, d, sph)
        got_t, got_i, looked, literal = walk(g, o, d, sph)
        chk = ~literal  # rays from far away that enter the box run the literal loop: nothing to check
        bad = chk & ((got_i != ref_i) | (got_t.view(np.uint32) != ref_t.view(np.uint32)))
        assert not bad.any(), (name, seed, np.nonzero(bad)[0][:5], got_i[bad][:5], ref_i[bad][:5], got_t[bad][:5], ref_t[bad][:5])
        total_hits += int((ref_i >= 0).sum())
        total_looked += int(looked[chk].sum())
        total_lit += int(literal.sum())
    assert total_hits > 300
    # the walk does cull: it looks at a small part of the scene per ray
    # (`clumps` is the grid's bad case — a dense clump inside one cell of a sparse field; PT_GEOM_AUTO
    # measures and keeps the hierarchy there)
    if len(sph) >= 100 and name != "clumps":
        assert total_looked < 0.1 * 4 * n_rays * len(sph)
    assert total_lit < 0.2 * 4 * n_rays


def test_walk_with_each_spheres_own_inflation_at_the_rims_of_the_big_ones_from_far_origins():
    """Registration inflates sphere i by the delta of ITS radius (pt_grid.hpp): a sphere of 20 x rmin gets a small fraction
    of delta_g.  The case that margin is for: rays from as far away as a walking ray can start (|o - c0| up to 2 s0), aimed at
    the rims of the biggest gridded spheres of a wide scene with very mixed radii — where the shader's own rounding decides
    whether and where the ray hits.  The fp32 emulation of the kernel's walk must return hit_world's pair.
    (A regression guard, not the proof: a margin that is too small shows only when a rounded hit point also falls across a
    cell boundary, which random rays practically never do — this test passes with NO inflation at all.  The margin's size
    is the per-sphere bound |P(v) - C| <= sqrt(r^2 + 32 u D^2) + 10 u r, which test_root_stays_within_delta_of_the_sphere
    measures pair by pair with the sphere's own r.)"""
    rng = np.random.default_rng(77)
    n = 700
    sph = random_field(n, 21, extent=90.0, rmax=0.1, giants=1)
    big = rng.random(n) < 0.3
    big[0] = False
    sph["radius"][big] = rng.uniform(1.0, 2.5, int(big.sum())).astype(np.float32)
    rc, g = build(sph)
    assert rc == 0
    gridded = np.unique(g["index"][:g["n_cell_entries"]])
    r_all = np.abs(np.asarray(sph["radius"], np.float64))
    big_gridded = gridded[r_all[gridded] >= 1.0]
    assert len(big_gridded) > 50 and r_all[gridded].min() < 0.06                 # mixed radii IN the cells
    x40 = 40.0 * U * float(g["d_near"]) ** 2
    first_small = np.sqrt(float(g["rmin"]) ** 2 + x40) - float(g["rmin"])
    first_big = np.sqrt(1.0 + x40) - 1.0
    assert first_big < 0.4 * first_small and first_small > 0.02 * float(g["rmin"])  # the margins really differ, and matter
    c = np.asarray(sph["center"], np.float64)
    c0, s0 = g["c0"].astype(np.float64), float(g["s0"])
    bad_total = hits = 0
    for seed in range(3):
        r2 = np.random.default_rng(100 + seed)
        m = 3000
        u = r2.normal(size=(m, 3)); u /= np.linalg.norm(u, axis=1)[:, None]
        o = c0 + u * (s0 * r2.uniform(1.2, 1.98, (m, 1)))                           # far, but still walking rays
        j = r2.choice(big_gridded, m)
        v = r2.normal(size=(m, 3)); v /= np.linalg.norm(v, axis=1)[:, None]
        # a rim point as seen from o: offset perpendicular to the line of sight
        los = c[j] - o; los /= np.linalg.norm(los, axis=1)[:, None]
        v -= np.einsum("ij,ij->i", v, los)[:, None] * los; v /= np.linalg.norm(v, axis=1)[:, None]
        target = c[j] + v * r_all[j, None] * r2.choice([0.9999, 0.999999, 1.0, 1.000001, 1.0001], (m, 1))
        d = (target - o) * r2.choice([1.0, 0.01, 3.0], (m, 1))
        o32, d32 = f32(o), f32(d)
        ref_t, ref_i = brute_force(o32, d32, sph)
        got_t, got_i, looked, literal = walk(g, o32, d32, sph)
        chk = ~literal
        assert chk.mean() > 0.9                                                      # they do walk
        bad = chk & ((got_i != ref_i) | (got_t.view(np.uint32) != ref_t.view(np.uint32)))
        bad_total += int(bad.sum()); hits += int((ref_i[chk] >= 0).sum())
        assert not bad.any(), (seed, np.nonzero(bad)[0][:5], got_i[bad][:5], ref_i[bad][:5])
    assert hits > 2000


def _morton(x, y, z):
    k = 0
    for b in range(10):
        k |= ((int(x) >> b) & 1) << (3 * b) | ((int(y) >> b) & 1) << (3 * b + 1) | ((int(z) >> b) & 1) << (3 * b + 2)
    return k


@pytest.mark.parametrize("name", ["field300", "config2", "clumps", "flat", "config5"])
def test_morton_runs_hold_the_same_spheres_in_the_same_order(name):
    """The layout for entries gathered from global memory (csrc/pt_grid.hpp morton_runs; config 5's kernel)
    only MOVES the runs: every cell keeps its spheres in their order, the runs follow each other in Morton
    order of their cells without gaps or overlaps, and the always-tested group still follows the cells'
    entries."""
    sph = GRID_SCENES[name]()
    rc, g = build(sph)
    rc2, r = build(sph, runs=True)
    assert rc == 0 and rc2 == 0
    assert np.array_equal(g["n"], r["n"]) and g["n_always"] == r["n_always"] and g["nonempty"] == r["nonempty"]
    assert g["n_cell_entries"] == r["n_cell_entries"] and g["n_entries"] == r["n_entries"]
    cnt_g, cnt_r = g["cells"] >> 24, r["cells"] >> 24
    assert np.array_equal(cnt_g, cnt_r)
    first_g, first_r = g["cells"] & 0xFFFFFF, r["cells"] & 0xFFFFFF
    used = np.zeros(r["n_cell_entries"], bool)
    nx, ny = int(g["n"][0]), int(g["n"][1])
    keys = []
    for c in np.nonzero(cnt_r)[0]:
        a, b, k = int(first_g[c]), int(first_r[c]), int(cnt_r[c])
        assert np.array_equal(g["index"][a:a + k], r["index"][b:b + k])
        assert np.array_equal(g["entries"][a:a + k].view(np.uint32), r["entries"][b:b + k].view(np.uint32))
        assert not used[b:b + k].any()
        used[b:b + k] = True
        keys.append((b, _morton(c % nx, (c // nx) % ny, c // (nx * ny))))
    assert used.all()
    keys.sort()
    assert all(k0[1] < k1[1] for k0, k1 in zip(keys, keys[1:])), "runs are not in Morton order of their cells"
    # the always-tested group: unchanged, behind the cells' entries
    assert np.array_equal(g["index"][g["n_cell_entries"]:], r["index"][r["n_cell_entries"]:])
    assert np.array_equal(g["entries"][g["n_cell_entries"]:].view(np.uint32), r["entries"][r["n_cell_entries"]:].view(np.uint32))
