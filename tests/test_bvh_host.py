"""CPU tests of the spatial index builder (r1_bvh.cpp through r1_bvh_describe; no GPU).

Structure: every hittable sphere sits in exactly one leaf slot, child boxes contain their
spheres, depth fits the kernel's traversal stack.  Conservativeness: a numpy re-statement of the
kernel's traversal rule (inflated slab test, r1_trace.hpp::bvh_box) must present every sphere
whose fp32 reference test can offer a hit (clear sign bit of the discriminant and a root
beyond t_min, rayweek1.cpp:192-204, :294-313) — checked against
brute force over all spheres for random and adversarial rays."""
import numpy as np
import pytest

import rays1bench_amd as r1
from rays1bench_amd import binding

F = np.float32
LEAF = 0x80000000


def scene_arrays(sc):
    a = sc.arrays()
    return a


M = ((0, 2, 4), (1, 3, 5))    # node row: {m0x m1x m0y m1y} {m0z m1z e0x e1x} {e0y e1y e0z e1z} {w2 k child0 child1}
E = ((6, 8, 10), (7, 9, 11))
REF = (14, 15)
EMPTY = 0xFFFFFFFF


def walk(nodes):
    """Yields (node, child_slot, ref, m, e, w2, k) for every child reference in the tree."""
    stack = [0]
    seen = set()
    while stack:
        n = stack.pop()
        assert n not in seen
        seen.add(n)
        row = nodes[n]
        refs = row.view(np.uint32)
        for c in (0, 1):
            ref = int(refs[REF[c]])
            yield n, c, ref, row[list(M[c])].astype(np.float64), row[list(E[c])].astype(np.float64), float(row[12]), float(row[13])
            if not ref & LEAF:
                stack.append(ref)
    assert len(seen) == len(nodes)


def leaf_slots(ref):
    """id slots of a leaf reference: two per sphere pair (one may be EMPTY)."""
    first, pairs = ref & 0x0FFFFFFF, (ref >> 28) & 7
    return range(2 * first, 2 * (first + pairs))


def subtree_slots(nodes, ref):
    if ref & LEAF:
        return list(leaf_slots(ref))
    row = nodes[ref].view(np.uint32)
    return subtree_slots(nodes, int(row[REF[0]])) + subtree_slots(nodes, int(row[REF[1]]))


@pytest.mark.parametrize("kind,gw,gh", [("small", 0, 0), ("medium", 0, 0), ("large", 0, 0), ("grid", 64, 40), ("grid", 400, 250)])
def test_tree_structure(kind, gw, gh):
    sc = {"small": r1.create_small_scene, "medium": r1.create_medium_scene, "large": r1.create_large_scene}[kind](1200, 800) \
        if kind != "grid" else r1.create_grid_scene(1920, 1080, gw, gh)
    a = sc.arrays()
    info, nodes, ids = binding.bvh_describe(sc.spheres.contents)
    active = np.nonzero(a["inv_radius"] != 0)[0]
    assert info["spheres"] == len(active)
    real = ids[ids != EMPTY]
    assert sorted(real.tolist()) == active.tolist()         # each hittable sphere exactly once, placeholders never
    assert len(ids) == 2 * info["pairs"]
    assert 1 <= info["depth"] <= info["stack_entries"]
    c = np.stack([a["center_x"], a["center_y"], a["center_z"]], 1).astype(np.float64)
    r = np.sqrt(a["radius_sq"].astype(np.float64))
    n_leaf = 0
    for n, ci, ref, m, e, w2, k in walk(nodes):
        if ref & LEAF:
            n_leaf += ((ref >> 28) & 7) > 0
            slots = [q for q in leaf_slots(ref) if ids[q] != EMPTY]
            assert len(slots) <= (8 if info["spheres"] > 1023 else 4)
        elif len(nodes) <= 2000:
            slots = [q for q in subtree_slots(nodes, ref) if ids[q] != EMPTY]
        else:
            continue  # big trees: leaves only (the inner boxes are unions of their children's by construction)
        for s in slots:
            i = ids[s]
            assert (c[i] - r[i] >= m - e - 1e-12).all() and (c[i] + r[i] <= m + e + 1e-12).all()
        # trees padded from the scene's centre carry ONE A and have every K folded into the half extents (r1_bvh.cpp)
        assert w2 > 0 and (k > 0 if info["pad_local"] else k == 0)
    assert n_leaf == info["leaves"]
    # the pad formula follows the sphere size: measured from one point of the scene for the reference's scenes, per node
    # for the lattices of small spheres (r1_bvh.cpp)
    assert info["pad_local"] == (1 if kind == "grid" and gw == 400 else 0), info
    if kind == "grid" and gw == 400:
        assert info["depth"] <= 24 and info["spheres"] == 100004


def ref_flagged(cx, cy, cz, rsq, o, d):
    """Spheres that can offer a hit: clear sign bit of the reference's fp32 discriminant
    (rayweek1.cpp:192-204) and a root beyond t_min (:294-313), vectorised over spheres."""
    cox, coy, coz = (cx - o[0]).astype(F), (cy - o[1]).astype(F), (cz - o[2]).astype(F)

    def fma(a, b, c):  # exactly rounded fp32 fma through float64 (products of two fp32 are exact in fp64;
        return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F)  # double rounding is harmless for a superset test

    nb = fma(coz, np.full_like(coz, d[2]), fma(coy, np.full_like(coy, d[1]), (cox * d[0]).astype(F)))
    cc = (fma(coz, coz, fma(coy, coy, (cox * cox).astype(F))) - rsq).astype(F)
    discr = ((nb * nb).astype(F) - cc).astype(F)
    ok = ~np.signbit(discr)
    root = np.sqrt(np.where(ok, discr, 0).astype(F)).astype(F)
    t1, t2 = (nb - root).astype(F), (nb + root).astype(F)
    return ok & ((t1 > F(0.001)) | (t2 > F(0.001)))


def traverse(nodes, centre, o, d, jitter=None, pad_local=0):
    """The kernel's visit rule (r1_trace.hpp::bvh_box, fp32 step by step) without distance pruning: leaf slots the ray is shown."""
    o = o.astype(F)

    def fma(a, b, c):  # exactly rounded fp32 fma (the product of two fp32 is exact in fp64; inf/NaN propagate alike)
        return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(F)

    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        inv = (F(1) / d.astype(F)).astype(F)
        if jitter is not None:  # the kernel uses v_rcp_f32 (1 ulp): any reciprocal within one ulp must do
            inv = np.where(np.isfinite(inv), np.nextafter(inv, np.where(jitter > 0, F(np.inf), F(-np.inf)).astype(F)), inv).astype(F)
        ainv = np.abs(inv)
        oi = (o * inv).astype(F)
        r = (o - centre.astype(F)).astype(F)
        r2 = fma(r[2], r[2], fma(r[1], r[1], F(r[0] * r[0])))
        out = []
        stack = [0]
        while stack:
            n = stack.pop()
            row = nodes[n]
            refs = row.view(np.uint32)
            dist2 = r2
            if pad_local:
                sv = fma(F(-2), o, (row[list(M[0])] + row[list(M[1])]).astype(F))
                dist2 = fma(sv[2], sv[2], fma(sv[1], sv[1], F(sv[0] * sv[0])))
            pad = fma(row[12], dist2, row[13])
            pa = (pad * ainv).astype(F)
            for c in (0, 1):
                a = fma(row[list(M[c])], inv, -oi)
                b = fma(row[list(E[c])], ainv, pa)
                tn = np.fmax(np.fmax(F(a[0] - b[0]), F(a[1] - b[1])), F(a[2] - b[2]))
                tf = np.fmin(np.fmin(F(a[0] + b[0]), F(a[1] + b[1])), F(a[2] + b[2]))
                if tn <= tf and tf >= 0:
                    ref = int(refs[REF[c]])
                    if ref & LEAF:
                        out.extend(leaf_slots(ref))
                    else:
                        stack.append(ref)
    return out


@pytest.mark.parametrize("kind", ["large", "grid", "far_tiny"])
def test_traversal_rule_presents_every_sphere_the_reference_flags(kind):
    rng = np.random.default_rng(5)
    if kind == "large":
        sc = r1.create_large_scene(1200, 800)
    elif kind == "grid":
        sc = r1.create_grid_scene(1920, 1080, 120, 80)
    else:
        sc = r1.create_grid_scene(1920, 1080, 120, 80)
    a = sc.arrays()
    info, nodes, ids = binding.bvh_describe(sc.spheres.contents)
    cx, cy, cz, rsq = a["center_x"], a["center_y"], a["center_z"], a["radius_sq"]
    active = a["inv_radius"] != 0
    n_rays = 400
    shown_total = flagged_total = 0
    for q in range(n_rays):
        if kind == "far_tiny":
            # origins hundreds of units away: the discriminant of the 0.05-radius spheres is mostly noise
            o = (rng.normal(0, 1, 3) * 300).astype(F)
            target = np.array([rng.uniform(-10, 10), 0.1, rng.uniform(-10, 10)])
        else:
            o = np.array([rng.uniform(-12, 12), rng.uniform(0.0, 6), rng.uniform(-12, 12)], F)
            target = np.array([rng.uniform(-12, 12), rng.uniform(-0.5, 1.0), rng.uniform(-12, 12)])
        d = (target - o).astype(np.float64)
        d = (d / np.linalg.norm(d)).astype(F)
        if q % 7 == 0:
            d = np.array([0, 0, -1], F) if q % 2 else np.array([1, 0, 0], F)  # axis-parallel: infinite reciprocals
        flagged = set(np.nonzero(ref_flagged(cx, cy, cz, rsq, o, d) & active)[0].tolist())
        shown = set(ids[traverse(nodes, info["centre"], o, d, rng.integers(0, 2, 3) * 2 - 1, info["pad_local"])].tolist()) - {EMPTY}
        assert flagged <= shown, (q, sorted(flagged - shown)[:5])
        shown_total += len(shown)
        flagged_total += len(flagged)
    # and the index is selective (not for far_tiny: from 300 units the reference's own test of a
    # 0.05-radius sphere is rounding noise and the pad must cover all of it)
    if kind != "far_tiny":
        assert shown_total < 0.2 * n_rays * active.sum()
    assert flagged_total > 0


def _raw_scene(c, rad):
    """CScene over numpy arrays (kept alive by the returned tuple)."""
    import ctypes as C
    c = np.asarray(c, F)
    rad = np.asarray(rad, F)
    n = len(rad)
    arrs = {"center_x": c[:, 0].copy(), "center_y": c[:, 1].copy(), "center_z": c[:, 2].copy(), "radius_sq": (rad * rad).astype(F),
            "inv_radius": (F(1) / rad).astype(F), "albedo_r": np.full(n, 0.5, F), "albedo_g": np.full(n, 0.5, F),
            "albedo_b": np.full(n, 0.5, F), "mat_param": np.zeros(n, F)}
    mt = np.zeros(n, np.uint8)
    cs = binding.CScene()
    cs.count = n
    for k, v in arrs.items():
        setattr(cs, k, v.ctypes.data_as(C.POINTER(C.c_float)))
    cs.mat_type = mt.ctypes.data_as(C.POINTER(C.c_uint8))
    return cs, arrs, mt


@pytest.mark.parametrize("seed", range(12))
def test_random_scenes_tree_invariants_and_visit_rule(seed):
    """Random clouds (1..600 spheres, radii over four decades, duplicates, one huge sphere now and then):
    structure invariants, depth bound, and the visit rule against brute force for 60 rays each."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(1, 601))
    c = rng.normal(0, rng.uniform(0.5, 20), (n, 3))
    rad = np.exp(rng.uniform(np.log(1e-3), np.log(10.0), n))
    if seed % 3 == 0 and n > 4:
        c[: n // 4] = c[0]                       # coincident centres
    if seed % 4 == 0:
        rad[-1], c[-1] = 2000.0, (0, -2001, 0)   # a ground sphere
    cs, arrs, mt = _raw_scene(c, rad)
    info, nodes, ids = binding.bvh_describe(cs)
    real = ids[ids != EMPTY]
    assert sorted(real.tolist()) == list(range(n))
    assert 1 <= info["depth"] <= info["stack_entries"]
    cx, cy, cz, rsq = arrs["center_x"], arrs["center_y"], arrs["center_z"], arrs["radius_sq"]
    for q in range(60):
        o = (rng.normal(0, 1, 3) * rng.choice([1.0, 30.0, 400.0])).astype(F)
        target = c[rng.integers(0, n)] + rng.normal(0, 1, 3) * rad.mean()
        d = (target - o).astype(np.float64)
        d = (d / np.linalg.norm(d)).astype(F)
        flagged = set(np.nonzero(ref_flagged(cx, cy, cz, rsq, o, d))[0].tolist())
        shown = set(ids[traverse(nodes, info["centre"], o, d, rng.integers(0, 2, 3) * 2 - 1, info["pad_local"])].tolist()) - {EMPTY}
        assert flagged <= shown, (seed, q, sorted(flagged - shown)[:5])


def test_discriminant_error_constant_behind_the_pad():
    """r1_bvh.cpp's pad rests on |fp32 discriminant of the reference - exact| <= 23 u |c - o|^2 + 2 u r^2
    (it uses 40 u).  Monte-Carlo check of that constant on the reference's operation order
    (rayweek1.cpp:192-202) over three magnitudes of coordinates: the observed worst case is ~7 u."""
    rng = np.random.default_rng(0)

    def fma(a, b, c):
        return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F)

    worst = 0.0
    for scale in (1.0, 30.0, 1000.0):
        n = 300_000
        o = (rng.normal(0, 1, (n, 3)) * scale).astype(F)
        c = (rng.normal(0, 1, (n, 3)) * scale).astype(F)
        d = rng.normal(0, 1, (n, 3))
        d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(F)
        r = np.exp(rng.uniform(np.log(1e-3), np.log(10), n)).astype(F)
        rsq = (r * r).astype(F)
        co = (c - o).astype(F)
        nb = fma(co[:, 2], d[:, 2], fma(co[:, 1], d[:, 1], (co[:, 0] * d[:, 0]).astype(F)))
        cc = (fma(co[:, 2], co[:, 2], fma(co[:, 1], co[:, 1], (co[:, 0] * co[:, 0]).astype(F))) - rsq).astype(F)
        discr = ((nb * nb).astype(F) - cc).astype(F).astype(np.float64)
        O, C, D = o.astype(np.float64), c.astype(np.float64), d.astype(np.float64)
        D = D / np.linalg.norm(D, axis=1, keepdims=True)  # the geometric line of the ray
        V = C - O
        exact = (V * D).sum(1) ** 2 - (V * V).sum(1) + rsq.astype(np.float64)
        u = 2.0 ** -24
        v2 = (V * V).sum(1)
        worst = max(worst, float(((np.abs(discr - exact) - 2 * u * rsq) / (u * v2)).max()))
    assert worst < 23.0, worst


# ---- round 2: the cases VERDICT r01 / ADVICE r01 name ---------------------------------------------


@pytest.mark.parametrize("family", ["degenerate_radii", "camera_inside_big_sphere", "far_cluster"])
def test_visit_rule_on_adversarial_families(family):
    """The three families whose exactness rests on constants nobody had exercised: radii
    1e-12..1e-6 (the r_floor branch of r1_bvh.cpp: radius_sq down to 1e-24), ray origins inside
    an r = 50 sphere, and clusters 1e4..1e5 units from the world origin (fp32 spacing there is
    1e-3..8e-3, comparable with the small radii)."""
    rng = np.random.default_rng({"degenerate_radii": 71, "camera_inside_big_sphere": 72, "far_cluster": 73}[family])
    n = 160
    if family == "degenerate_radii":
        c = rng.uniform(-3, 3, (n, 3))
        rad = np.exp(rng.uniform(np.log(1e-12), np.log(1e-6), n))
        origin = lambda: rng.uniform(-4, 4, 3)
    elif family == "camera_inside_big_sphere":
        c = rng.uniform(-20, 20, (n, 3))
        rad = rng.uniform(0.1, 1.5, n)
        c[0], rad[0] = (0.0, 0.0, 0.0), 50.0
        origin = lambda: rng.uniform(-20, 20, 3)  # always inside sphere 0
    else:
        shift = np.array([3.0e4, -8.0e4, 1.2e4]) * rng.uniform(0.4, 1.2)
        c = rng.uniform(-6, 6, (n, 3)) + shift
        rad = np.exp(rng.uniform(np.log(0.02), np.log(1.0), n))
        origin = lambda: rng.uniform(-9, 9, 3) + shift
    cs, arrs, mt = _raw_scene(c, rad)
    info, nodes, ids = binding.bvh_describe(cs)
    assert sorted(ids[ids != EMPTY].tolist()) == list(range(n))
    cx, cy, cz, rsq = arrs["center_x"], arrs["center_y"], arrs["center_z"], arrs["radius_sq"]
    hits = 0
    for q in range(150):
        o = origin().astype(F)
        i = int(rng.integers(0, n))
        # aim at (or just past the rim of) a sphere so that grazing cases are frequent
        target = np.array([cx[i], cy[i], cz[i]], np.float64) + rng.normal(0, 1, 3) * rad[i] * rng.choice([0.0, 0.7, 1.0, 1.05])
        d = target - o.astype(np.float64)
        d = (d / np.linalg.norm(d)).astype(F)
        flagged = set(np.nonzero(ref_flagged(cx, cy, cz, rsq, o, d))[0].tolist())
        shown = set(ids[traverse(nodes, info["centre"], o, d, rng.integers(0, 2, 3) * 2 - 1, info["pad_local"])].tolist()) - {EMPTY}
        assert flagged <= shown, (family, q, sorted(flagged - shown)[:5])
        hits += len(flagged)
    assert hits > 0  # the family does produce reference candidates (for degenerate radii: rounding-noise hits)


def test_describe_filters_like_set_scene_and_bounds_follow_the_larger_radius():
    """r1_bvh_describe builds exactly what r1_set_scene builds — non-finite centres /
    radius_sq are dropped (they can never be hit), NaN inv_radius is an error — and a caller whose
    radius_sq and inv_radius disagree (or whose inv_radius is negative) still gets boxes that cover
    the radius the exact test reads."""
    import ctypes as C
    c = np.array([[0, 0, 0], [np.nan, 0, 0], [3, 0, 0], [6, 0, 0], [9, 0, 0], [np.inf, 1, 1]], F)
    rad = np.array([1.0, 1.0, 0.5, 0.5, 0.25, 1.0], F)
    cs, arrs, mt = _raw_scene(c, rad)
    arrs["inv_radius"][2] = F(1 / 0.05)   # inv_radius of a 10x smaller sphere: bounds must follow radius_sq
    arrs["inv_radius"][3] = F(-2.0)       # negative inv_radius (the reference would flip the normal, still hit)
    arrs["radius_sq"][4] = F(np.inf)      # can never be hit: dropped
    info, nodes, ids = binding.bvh_describe(cs)
    assert sorted(ids[ids != EMPTY].tolist()) == [0, 2, 3]
    cc = c.astype(np.float64)
    r_true = np.sqrt(arrs["radius_sq"].astype(np.float64))
    for n_, ci, ref, m, e, w2, k in walk(nodes):
        if ref & LEAF:
            for s in leaf_slots(ref):
                if ids[s] != EMPTY:
                    i = ids[s]
                    assert (cc[i] - r_true[i] >= m - e - 1e-12).all() and (cc[i] + r_true[i] <= m + e + 1e-12).all()
    arrs["inv_radius"][0] = F(np.nan)
    info2 = binding.BvhInfo()
    assert binding.lib().r1_bvh_describe(C.byref(cs), 0, C.byref(info2), None, 0, None, 0) == binding.R1_EINVAL


def test_outlier_peeling_keeps_the_lattice_boxes_flat_and_the_depth_bounded():
    """r1_bvh.cpp peels up to one leaf's worth of spheres much larger than the node's median radius into a leaf of their own
    before the centroid split.  Large scene: the root is [ground + the three r = 2 balls | the lattice], and the lattice's box
    is as flat as the lattice (0.5 high, not 2.7); a tower of ever larger spheres cannot push the depth past the stack."""
    sc = r1.create_large_scene(1200, 800)
    a = sc.arrays()
    info, nodes, ids = binding.bvh_describe(sc.spheres.contents)
    root = nodes[0].view(np.uint32)
    big = sorted(np.argsort(a["radius_sq"])[-4:].tolist())
    leaf = [int(r) for r in (root[14], root[15]) if r & LEAF]
    assert len(leaf) == 1 and sorted(ids[list(leaf_slots(leaf[0]))].tolist()) == big
    lattice_child = 1 if (root[14] & LEAF) else 0
    # half height of the lattice's boxes: the root stores its children's half extents plus its own pad constant K (which measures
    # the distance to the ground sphere's centre: ~17), so the flatness is read off the lattice child's own two children
    lattice = nodes[int(root[14 + lattice_child])]
    assert lattice[E[0][1]] < 0.6 and lattice[E[1][1]] < 0.6
    assert (nodes[:, 13] == 0).all() and (nodes[:, 12] == nodes[0][12]).all() and nodes[0][12] > 0  # one A for the tree, K folded into e
    # the root step of the kernels (bvh_advance): child `1 - lattice_child` is the leaf every ray tests
    assert info["root_leaf"] == (1 - lattice_child) + 1
    # adversarial: radii 4^i (every sphere an outlier of the rest) + a crowd of small ones
    rng = np.random.default_rng(9)
    n_small = 600
    rad = np.concatenate([np.full(n_small, 0.01), 0.05 * 4.0 ** np.arange(20)])
    c = np.concatenate([rng.uniform(-1, 1, (n_small, 3)), rng.uniform(-1, 1, (20, 3))])
    cs, arrs, mt = _raw_scene(c, rad)
    info2, nodes2, ids2 = binding.bvh_describe(cs)
    assert sorted(ids2[ids2 != EMPTY].tolist()) == list(range(len(rad)))
    assert 1 <= info2["depth"] <= info2["stack_entries"]
