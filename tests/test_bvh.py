"""The culling hierarchy of PT_GEOM_BVH (ray_tracer_webgl_amd/csrc/pt_bvh.hpp), checked on the host.

The kernels may skip a sphere only if the reference's hit_sphere (static/shader.frag:145-173)
could not accept it, so these tests check (i) the structure pt_set_spheres uploads and (ii) —
with a numpy emulation of the kernel's inflated slab walk against a numpy emulation of the
literal fp32 discriminant — that no sphere whose discriminant is >= 0 is ever skipped.
No GPU needed: pt_build_bvh is the host half of the C ABI.
"""
import ctypes as C

import numpy as np
import pytest

from ray_tracer_webgl_amd import _lib, abi, scenes

INNER = 0xFFFFFFFF


def build(spheres):
    lib = _lib.load()
    ptr, n, keep = abi.spheres_as_ctypes(spheres)
    counts = np.zeros(5, np.uint32)
    margin = np.zeros(4, np.float32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = lib.pt_build_bvh(ptr, n, None, 0, None, 0, None, 0, vp(margin), vp(counts), None, 0, None, None, 0)
    if rc != 0:
        return rc, None
    nodes = np.zeros((counts[0], 8), np.float32)
    slots = np.zeros((counts[1], 4), np.float32)
    index = np.zeros(counts[1], np.uint32)
    nodes16 = np.zeros((counts[0] + 1, 4), np.uint32)
    kscale = C.c_float(0)
    nodes32 = np.zeros((counts[0] + 1, 8), np.float32)
    rc = lib.pt_build_bvh(ptr, n, vp(nodes), nodes.size, vp(slots), slots.size, vp(index), index.size, vp(margin), vp(counts),
                          vp(nodes16), nodes16.size, C.byref(kscale), vp(nodes32), nodes32.size)
    return rc, dict(nodes=nodes, slots=slots, index=index, margin=margin, n_nodes=int(counts[0]), n_slots=int(counts[1]),
                    n_tree_slots=int(counts[2]), n_outliers=int(counts[3]), depth=int(counts[4]), nodes16=nodes16,
                    kscale=float(kscale.value), nodes32=nodes32)


def unpack32(b):
    """fp32 device nodes -> (lo, hi) boxes in the frame x - c0, skip, leaf number"""
    w = b["nodes32"]
    return w[:, 0:3], w[:, 4:7], w[:, 3].view(np.uint32).astype(np.int64), w[:, 7].view(np.uint32).astype(np.int64)


def unpack16(b):
    """packed nodes -> (lo, hi) float32 boxes in the frame (x - c0) * kscale, skip, leaf number"""
    w = b["nodes16"]
    h = np.ascontiguousarray(w[:, :3]).view(np.float16).reshape(len(w), 6).astype(np.float32)  # lo.x lo.y lo.z hi.x hi.y hi.z
    return h[:, 0:3], h[:, 3:6], (w[:, 3] & 0xFFFF).astype(np.int64), (w[:, 3] >> 16).astype(np.int64)


def random_field(n, seed, extent=20.0, rmax=0.6, giants=1):
    rng = np.random.default_rng(seed)
    s = np.zeros(n, dtype=abi.SPHERE_DTYPE)
    s["center"] = rng.uniform(-extent, extent, (n, 3)).astype(np.float32)
    s["radius"] = rng.uniform(0.05, rmax, n).astype(np.float32)
    s["radius"][rng.random(n) < 0.1] *= -1  # negative radii are legal (src/state.rs:200)
    for g in range(giants):
        s["center"][g] = (0.0, -1000.0 - extent, 0.0)
        s["radius"][g] = 1000.0
    s["albedo"] = 0.5
    s["uuid"] = np.arange(n)
    return s


SCENES = {
    "config2": lambda: scenes.config2(64, 36, 1, 1, 8).spheres,
    "config5": lambda: scenes.config5(64, 36, 1, 1, 8).spheres,
    "field300": lambda: random_field(300, 1),
    "field17_no_giant": lambda: random_field(17, 2, giants=0),
    "clumps": lambda: np.concatenate([random_field(64, 3, extent=1.0, giants=0), random_field(64, 4, extent=300.0, giants=0)]),
}


@pytest.mark.parametrize("name", sorted(SCENES))
def test_structure(name):
    sph = SCENES[name]()
    rc, b = build(sph)
    assert rc == 0
    nodes, slots, index = b["nodes"], b["slots"], b["index"]
    n = len(sph)
    skip = nodes[:, 3].view(np.uint32)
    leaf = nodes[:, 7].view(np.uint32)
    # every sphere sits in exactly one slot; padding slots can never pass the literal test
    real = index != INNER
    assert sorted(index[real].tolist()) == list(range(n))
    assert np.all(np.isneginf(slots[~real, 3]))
    c = np.asarray(sph["center"], np.float32)
    r = np.asarray(sph["radius"], np.float32)
    assert np.array_equal(slots[real, :3], c[index[real]])
    assert np.array_equal(slots[real, 3], (r * r)[index[real]])  # fp32 r*r, as the list kernels use
    # outliers: after the tree's slots, ascending index order
    tail = index[b["n_tree_slots"]:]
    tail = tail[tail != INNER]
    assert len(tail) == b["n_outliers"] and np.all(np.diff(tail.astype(np.int64)) > 0)
    # depth-first layout: leaves own consecutive groups of four slots, skip links close subtrees
    is_leaf = leaf != INNER
    assert np.array_equal(leaf[is_leaf], 4 * np.arange(is_leaf.sum(), dtype=np.uint32))
    assert 4 * is_leaf.sum() == b["n_tree_slots"]
    assert np.all(skip > np.arange(len(nodes))) and np.all(skip <= len(nodes))
    assert np.array_equal(skip[is_leaf], np.nonzero(is_leaf)[0] + 1)
    assert skip[0] == len(nodes)
    # containment, in double: leaf boxes hold their spheres, inner boxes hold their subtrees
    lo, hi = nodes[:, 0:3].astype(np.float64), nodes[:, 4:7].astype(np.float64)
    for i in np.nonzero(is_leaf)[0]:
        members = index[leaf[i]:leaf[i] + 4]
        members = members[members != INNER]
        assert len(members) >= 1
        cc, rr = c[members].astype(np.float64), np.abs(r[members].astype(np.float64))[:, None]
        assert np.all(lo[i] <= (cc - rr).min(0)) and np.all(hi[i] >= (cc + rr).max(0))
    for i in np.nonzero(~is_leaf)[0]:
        sub = slice(i + 1, skip[i])
        assert skip[i] > i + 2  # two children at least
        assert np.all(lo[i] <= lo[sub].min(0)) and np.all(hi[i] >= hi[sub].max(0))
        # the left child is i+1, the right child starts where the left subtree ends
        assert skip[skip[i + 1]] == skip[i]
    # the packed form the kernels read: same links, boxes that contain the float boxes
    lo16, hi16, skip16, leaf16 = unpack16(b)
    k = b["kscale"]
    assert k > 0 and np.log2(k) == np.round(np.log2(k)) and np.abs(lo16).max() <= 1024 and np.abs(hi16).max() <= 1024
    assert np.array_equal(skip16[:-1], skip) and skip16[-1] == len(nodes) and leaf16[-1] == 0xFFFF
    assert np.array_equal(leaf16[:-1][is_leaf], leaf[is_leaf] // 4) and np.all(leaf16[:-1][~is_leaf] == 0xFFFF)
    c0d = b["margin"][:3].astype(np.float64)
    assert np.all(lo16[:-1].astype(np.float64) / k + c0d <= lo) and np.all(hi16[:-1].astype(np.float64) / k + c0d >= hi)
    # ... and not much bigger: one binary16 step at the rim of the scene
    assert np.all(lo - (lo16[:-1] / k + c0d) <= b["margin"][3] / 1000.0 + 1e-6)
    lo32, hi32, skip32, leaf32 = unpack32(b)
    assert np.array_equal(skip32, 32 * skip16) and np.array_equal(leaf32, leaf16)  # fp32 form: byte offsets
    assert np.all(lo32[:-1].astype(np.float64) + c0d <= lo) and np.all(hi32[:-1].astype(np.float64) + c0d >= hi)
    assert np.all(lo16[:-1] / np.float32(k) <= lo32[:-1]) and np.all(hi16[:-1] / np.float32(k) >= hi32[:-1])
    # the margin's reference data
    c0, s0 = b["margin"][:3].astype(np.float64), float(b["margin"][3])
    tree = index[:b["n_tree_slots"]]
    tree = tree[tree != INNER]
    reach = np.linalg.norm(c[tree].astype(np.float64) - c0, axis=1) + np.abs(r[tree].astype(np.float64))
    assert reach.max() <= s0


def test_giants_are_left_out_of_the_tree():
    rc, b = build(SCENES["config2"]())
    assert rc == 0 and b["n_outliers"] == 1
    assert b["index"][b["n_tree_slots"]] == 0  # the r = 1000 ground sphere
    assert b["margin"][3] < 20.0               # so the margin's scale is the field, not the ground


def test_scenes_without_a_hierarchy():
    assert build(scenes.default_scene(64, 36, 1, 8).spheres)[0] == abi.PT_ERR_NOT_READY  # 9 spheres
    bad = random_field(64, 5)
    bad["center"][7, 1] = np.inf
    assert build(bad)[0] == abi.PT_ERR_NOT_READY
    bad = random_field(64, 5)
    bad["radius"][3] = np.nan
    assert build(bad)[0] == abi.PT_ERR_NOT_READY


def test_identical_centres_still_split():
    s = random_field(200, 6, giants=0)
    s["center"][:] = (1.0, 2.0, 3.0)
    rc, b = build(s)
    assert rc == 0 and b["depth"] <= 12
    assert sorted(b["index"][b["index"] != INNER].tolist()) == list(range(200))


# ---- no sphere that can pass the literal test is skipped -----------------------------------------
def f32(x):
    return np.asarray(x, np.float32)


def fma(a, b, c):  # one rounding: the product of two floats is exact in double
    return f32(a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64))


def literal_disc(o, d, cs, r2):
    """discriminant of hit_sphere as PT_TEST evaluates it (pt_kernels.hip), rays x spheres"""
    oc = [f32(o[:, k, None] - cs[None, :, k]) for k in range(3)]
    dd = [np.broadcast_to(d[:, k, None], oc[0].shape) for k in range(3)]
    hb = fma(oc[2], dd[2], fma(oc[1], dd[1], f32(oc[0] * dd[0])))
    cc = fma(oc[2], oc[2], fma(oc[1], oc[1], fma(oc[0], oc[0], -np.broadcast_to(r2[None, :], oc[0].shape))))
    a = fma(d[:, 2], d[:, 2], fma(d[:, 1], d[:, 1], f32(d[:, 0] * d[:, 0])))
    disc = fma(-np.broadcast_to(a[:, None], cc.shape), cc, f32(hb * hb))
    return disc, hb, cc


def visited_slots(b, o, d, packed=True):
    """the kernel's walk (same formulas, fp32; packed binary16 boxes or the fp32 boxes of small
    scenes): boolean rays x slots, True where a slot is looked at"""
    lo16, hi16, skip, leaf = unpack16(b) if packed else unpack32(b)
    if not packed:
        skip = skip // 32
    c0, s0 = b["margin"][:3], b["margin"][3]
    kinv = f32(1.0 / b["kscale"]) if packed else f32(1.0)
    p = f32(o - c0[None, :])
    l1 = f32(f32(f32(np.abs(p[:, 0]) + np.abs(p[:, 1])) + np.abs(p[:, 2])) + s0)
    m = fma(f32(np.full(len(o), 1.25e-3)), l1, f32(np.full(len(o), 1e-6)))
    with np.errstate(divide="ignore"):
        inv = np.clip(f32(1.0) / d, f32(-1e18), f32(1e18)).astype(np.float32)
    kk = f32(inv * kinv)
    ah = -f32(f32(p + m[:, None]) * inv)
    al = -f32(f32(p - m[:, None]) * inv)
    seen = np.zeros((len(o), b["n_slots"]), bool)
    seen[:, b["n_tree_slots"]:] = True  # outliers: every ray
    cur = np.zeros(len(o), np.int64)
    for i in range(b["n_nodes"]):
        act = cur == i
        if not act.any():
            continue
        t1 = fma(np.broadcast_to(lo16[i], o.shape), kk, ah)
        t2 = fma(np.broadcast_to(hi16[i], o.shape), kk, al)
        tn = np.maximum(np.minimum(t1, t2).max(1), f32(0))
        tf = np.maximum(t1, t2).min(1)
        through = tn <= tf
        if leaf[i] != 0xFFFF:
            seen[act & through, 4 * leaf[i]:4 * leaf[i] + 4] = True
        cur = np.where(act, np.where(through, i + 1, skip[i]), cur)
    return seen


def rays_for(sph, n, seed):
    """origins on sphere surfaces (bounce rays), at the camera, and far away; unnormalised directions"""
    rng = np.random.default_rng(seed)
    c = np.asarray(sph["center"], np.float64)
    r = np.abs(np.asarray(sph["radius"], np.float64))
    k = rng.integers(0, len(sph), n)
    u = rng.normal(size=(n, 3))
    u /= np.linalg.norm(u, axis=1)[:, None]
    o = c[k] + u * r[k, None]
    d = rng.normal(size=(n, 3)) * rng.choice([1e-3, 0.3, 1.0, 30.0], (n, 1))
    # a third of the rays aim at another sphere's rim: the cases where rounding decides
    aim = rng.random(n) < 0.35
    j = rng.integers(0, len(sph), n)
    v = rng.normal(size=(n, 3))
    v /= np.linalg.norm(v, axis=1)[:, None]
    target = c[j] + v * r[j, None] * rng.choice([0.999999, 1.0, 1.000001, 1.001], (n, 1))
    d[aim] = (target - o)[aim] * rng.choice([1.0, 0.01], (n, 1))[aim]
    far = rng.random(n) < 0.1
    o[far] *= rng.choice([10.0, 100.0, 1e4], (n, 1))[far]
    axis = rng.random(n) < 0.05
    d[axis, rng.integers(0, 3)] = 0.0  # axis-parallel components
    return f32(o), f32(d)


@pytest.mark.parametrize("name", ["config2", "field300", "clumps", "field17_no_giant", "config5"])
def test_walk_reaches_every_sphere_that_can_pass(name):
    sph = SCENES[name]()
    rc, b = build(sph)
    assert rc == 0
    real = b["index"] != INNER
    cs, r2 = b["slots"][:, :3], b["slots"][:, 3]
    total_pass = total_seen = 0
    n_rays = 4000 if len(sph) < 2000 else 500  # rays x spheres matrices: keep them in memory
    for seed in range(4):
        o, d = rays_for(sph, n_rays, seed)
        a = np.einsum("ij,ij->i", d.astype(np.float64), d.astype(np.float64))
        ok = (a > 1e-12) & (a < 1e6)  # the kernel's regular rays; the others take the literal loop
        o, d = o[ok], d[ok]
        disc, hb, cc = literal_disc(o, d, cs[real], r2[real])
        can_pass = ~(disc < 0) & ~((cc > 0) & (hb >= 0))
        seen = visited_slots(b, o, d)[:, real]
        missed = can_pass & ~seen
        assert not missed.any(), (name, seed, np.argwhere(missed)[:5])
        seen32 = visited_slots(b, o, d, packed=False)[:, real]
        missed = can_pass & ~seen32
        assert not missed.any(), (name, seed, "fp32 boxes", np.argwhere(missed)[:5])
        assert seen32.sum() <= seen.sum()
        total_pass += int(can_pass.sum())
        total_seen += int(seen.sum())
    assert total_pass > 500
    # and the walk does cull: it looks at a small part of the scene
    if len(sph) >= 100:
        assert total_seen < 0.35 * 4 * n_rays * real.sum()
