"""GPU parity tests of the optional spatial index (R1_VARIANT_BVH).

The reference has no acceleration structure; the box tree only chooses which spheres are given
to the reference's per-sphere test, and must never lose a sphere the reference would hit: every
test here requires BIT-IDENTICAL samples, ray counts and pixels against the exhaustive kernels
(R1_VARIANT_REFERENCE: every sphere in the reference's arithmetic) and against the CPU oracle.
"""
import ctypes as C

import numpy as np
import pytest

import rays1bench_amd as r1
from rays1bench_amd import binding
import r1o

pytestmark = pytest.mark.gpu

MAKE = {"small": r1.create_small_scene, "medium": r1.create_medium_scene, "large": r1.create_large_scene}
BVH = binding.VARIANT_BVH


@pytest.fixture(scope="module")
def renderer():
    assert r1.device_count() >= 1, "no HIP device: the product has no CPU fallback"
    r = r1.Renderer(0)
    yield r
    r.close()


def oracle_scene(sc):
    return r1o.SceneArrays.from_c(sc.spheres, sc.camera)


def oparams(p):
    return r1o.make_params(p.width, p.height, p.spp, p.seed, p.max_bounces, p.tile_w, p.tile_h, p.shard, p.num_shards)


def _as_cscene(sa):
    cs = binding.CScene()
    cs.count = sa.count
    for k in r1o.SCENE_F32:
        setattr(cs, k, sa.arrays[k].ctypes.data_as(C.POINTER(C.c_float)))
    cs.mat_type = sa.arrays["mat_type"].ctypes.data_as(C.POINTER(C.c_uint8))
    return cs


def _as_ccamera(sa):
    cc = binding.CCamera()
    C.memmove(C.byref(cc), C.byref(sa.camera), C.sizeof(cc))
    return cc


def same(a, b):
    return a[1] == b[1] and a[2].tobytes() == b[2].tobytes() and a[0].tobytes() == b[0].tobytes()


def pad8(arr):
    n = len(arr["center_x"])
    pad = (-n) % 8 or (8 if n == 0 else 0)
    for k in arr:
        fill = {"center_x": 999999999.0, "center_y": 999999999.0, "center_z": 999999999.0, "mat_type": 255}.get(k, 0)
        arr[k] = np.concatenate([arr[k], np.full(pad, fill, arr[k].dtype)])
    return arr


def random_materials(rng, n):
    mt = rng.integers(0, 3, n).astype(np.uint8)
    return {"mat_type": mt,
            "albedo_r": rng.uniform(0.1, 0.95, n).astype(np.float32), "albedo_g": rng.uniform(0.1, 0.95, n).astype(np.float32),
            "albedo_b": rng.uniform(0.1, 0.95, n).astype(np.float32),
            "mat_param": np.where(mt == 2, rng.uniform(1.1, 2.4, n), rng.uniform(0, 1, n)).astype(np.float32)}


def spheres(c, rad, rng):
    c = np.asarray(c, np.float32)
    rad = np.asarray(rad, np.float32)
    arr = {"center_x": c[:, 0].copy(), "center_y": c[:, 1].copy(), "center_z": c[:, 2].copy(), "radius_sq": rad * rad,
           "inv_radius": (np.float32(1.0) / rad).astype(np.float32)}
    arr.update(random_materials(rng, len(rad)))
    return pad8(arr)


@pytest.mark.parametrize("name,w,h,spp,seed", [("small", 160, 120, 16, 7), ("medium", 333, 211, 5, 123456789), ("large", 256, 192, 12, 0)])
def test_bvh_full_frame_bit_exact_vs_oracle(renderer, name, w, h, spp, seed):
    sc = MAKE[name](w, h)
    renderer.set_scene(sc)
    p = r1.make_params(w, h, spp, seed, variant=BVH)
    img, rays, samples = renderer.render_samples(p)
    oimg, orays, osamples = r1o.render_frame(oracle_scene(sc), oparams(p), want_samples=True)
    differing = (samples.view(np.uint32) != osamples.view(np.uint32)).any(1)
    assert differing.mean() <= 1e-5, differing.sum()  # pow5 vs glibc powf, as for the default kernel
    assert abs(rays - orays) <= 51 * differing.sum()


@pytest.mark.parametrize("name", ["small", "medium", "large"])
def test_bvh_equals_exhaustive_sweep_at_the_baseline_size(renderer, name):
    """1200x800x10 (BASELINE configs 2/3): the tree-driven kernel and the default (grouped
    exhaustive sweep) agree on every one of the 9.6 M samples (the exhaustive sweep pinned with PREFILTER)."""
    w, h, spp = 1200, 800, 10
    renderer.set_scene(MAKE[name](w, h))
    a = renderer.render_samples(r1.make_params(w, h, spp, 10001, variant=binding.VARIANT_PREFILTER))
    assert renderer.launch_info()["kernel"] == binding.VARIANT_PREFILTER
    b = renderer.render_samples(r1.make_params(w, h, spp, 10001, variant=BVH))
    assert same(a, b)


@pytest.mark.parametrize("n_active", [0, 1, 2, 4, 5, 9, 600, 1023, 1024])
def test_bvh_sphere_count_edges(renderer, n_active):
    w, h, spp = 64, 48, 2
    src = r1.create_grid_scene(w, h, 36, 30)
    arr = src.arrays()
    keep = np.nonzero(arr["inv_radius"] != 0)[0][-n_active:] if n_active else np.zeros(0, np.int64)
    sub = pad8({k: v[keep] for k, v in arr.items()})
    sa = r1o.SceneArrays(sub, src.camera_array())
    renderer.set_scene_raw(_as_cscene(sa), _as_ccamera(sa))
    got = renderer.render_samples(r1.make_params(w, h, spp, 9, variant=BVH))
    # 600 spheres: a node table of more than 128 but at most 256 nodes, kept in LDS by the small-scene kernels;
    # 1023: more than 256 nodes, so the big-scene tree kernels run although the hit indices would fit 10 bits
    nodes = renderer.launch_info()["bvh_nodes"]
    assert {600: 128 < nodes <= 256, 1023: nodes > 256}.get(n_active, True), nodes
    ref = renderer.render_samples(r1.make_params(w, h, spp, 9, variant=binding.VARIANT_REFERENCE))
    assert same(got, ref)
    oimg, orays, osamples = r1o.render_frame(sa, oparams(r1.make_params(w, h, spp, 9)), want_samples=True)
    assert got[1] == orays and got[2].tobytes() == osamples.tobytes()


@pytest.mark.parametrize("pick,code", [("first9", 2), ("large", 1), ("first12", 0)])
def test_root_step_shapes(renderer, pick, code):
    """bvh_advance's root step (the root's leaf child and the box of its sibling tested outside the walk's loops) for each shape of root
    r1_bvh_info.root_leaf reports: the leaf is child 1 (nine spheres: [inner | leaf]), child 0 (the large scene's outlier leaf), and no
    such root (both children inner nodes): the tree kernel equals the reference-form sweep and the oracle either way."""
    w, h, spp = 96, 64, 3
    if pick == "large":
        src = r1.create_large_scene(w, h)
        sa = r1o.SceneArrays.from_c(src.spheres, src.camera)
    else:
        src = r1.create_grid_scene(w, h, 36, 30)
        arr = src.arrays()
        keep = np.nonzero(arr["inv_radius"] != 0)[0][:9 if pick == "first9" else 12]
        sa = r1o.SceneArrays(pad8({k: v[keep] for k, v in arr.items()}), src.camera_array())
    cs = _as_cscene(sa)
    info, _, _ = binding.bvh_describe(cs)
    assert info["root_leaf"] == code, info
    renderer.set_scene_raw(cs, _as_ccamera(sa))
    got = renderer.render_samples(r1.make_params(w, h, spp, 21, variant=BVH))
    assert renderer.launch_info()["kernel"] == BVH
    ref = renderer.render_samples(r1.make_params(w, h, spp, 21, variant=binding.VARIANT_REFERENCE))
    assert same(got, ref)
    oimg, orays, osamples = r1o.render_frame(sa, oparams(r1.make_params(w, h, spp, 21)), want_samples=True)
    assert got[1] == orays and got[2].tobytes() == osamples.tobytes()


CASES = ["mixed_radii", "far_camera", "dense_cluster", "tiny_spheres", "nested", "noise_dominated", "coincident", "collinear"]


@pytest.mark.parametrize("case", CASES)
def test_bvh_is_exact_on_adversarial_scenes(renderer, case):
    """Geometry that stresses the pad analysis of r1_bvh.cpp: radius ratios of 300, a scene
    850 units from the world origin, touching/nested/coincident spheres, and spheres so small
    and far that the reference's fp32 discriminant is mostly rounding noise (its 'hits' reach
    well outside the geometric sphere — the tree must still present those spheres)."""
    rng = np.random.default_rng(100 + CASES.index(case))
    w, h, spp = 72, 48, 3
    base = r1.create_small_scene(w, h)
    cam = base.camera_array().copy()
    n = 300
    if case == "mixed_radii":
        c, rad = rng.uniform(-12, 12, (n, 3)), np.exp(rng.uniform(np.log(0.02), np.log(6.0), n))
    elif case == "far_camera":
        shift = np.array([700.0, 260.0, 410.0], np.float32)
        c, rad = rng.uniform(-8, 8, (n, 3)) + shift, rng.uniform(0.2, 0.8, n)
        cam[0:3] += shift
        cam[3:6] += shift
    elif case == "dense_cluster":
        c, rad = rng.normal(0, 1.2, (n, 3)), rng.uniform(0.05, 0.35, n)
    elif case == "tiny_spheres":
        c, rad = rng.uniform(-3, 3, (n, 3)), np.exp(rng.uniform(np.log(1e-3), np.log(0.05), n))
    elif case == "nested":
        centres = rng.uniform(-4, 4, (12, 3))
        c, rad = centres[rng.integers(0, 12, n)], rng.uniform(0.05, 2.5, n)
    elif case == "noise_dominated":
        # radius 2e-3 seen from ~600 units: discriminant error ~ 2^-22 * 3.6e5 >> r^2 = 4e-6
        shift = np.array([-420.0, 380.0, 210.0], np.float32)
        c, rad = rng.uniform(-1.5, 1.5, (n, 3)) + shift, np.full(n, 2e-3)
        c[:40] = rng.uniform(-1.5, 1.5, (40, 3))  # and some near the camera
        rad[:20] = 0.4
        rad[-1] = 300.0  # a big mirror ball behind, sends rays back at the far cluster
        c[-1] = np.array([0.0, -302.0, 0.0])
    elif case == "coincident":
        c, rad = np.repeat(rng.uniform(-3, 3, (n // 6, 3)), 6, axis=0), np.repeat(rng.uniform(0.1, 0.6, n // 6), 6)
    else:  # collinear: all centres on one axis-parallel line (degenerate boxes, one split axis)
        c = np.zeros((n, 3))
        c[:, 0] = rng.uniform(-20, 20, n)
        c[:, 1] = 0.25
        rad = rng.uniform(0.05, 0.3, n)
    arr = spheres(c, rad, rng)
    sa = r1o.SceneArrays(arr, cam)
    renderer.set_scene_raw(_as_cscene(sa), _as_ccamera(sa))
    got = renderer.render_samples(r1.make_params(w, h, spp, 1234, variant=BVH))
    ref = renderer.render_samples(r1.make_params(w, h, spp, 1234, variant=binding.VARIANT_REFERENCE))
    assert same(got, ref)
    oimg, orays, osamples = r1o.render_frame(sa, oparams(r1.make_params(w, h, spp, 1234)), want_samples=True)
    assert got[1] == orays and got[2].tobytes() == osamples.tobytes()


def test_bvh_deep_paths_between_two_huge_spheres(renderer):
    """A gap of two units between a floor and a ceiling sphere of radius 1000, bright Lambertian: paths bounce dozens of
    times before they escape sideways to the sky, so the attenuation stack runs past the 30 entries the tree kernel
    keeps in LDS (R1_STACK_LDS_WORDS; deeper entries live in the global workspace) and is unwound from there with a
    non-zero sky colour.  Bit-identical to the reference-form kernel (whole stack in LDS) and to the oracle."""
    rng = np.random.default_rng(77)
    w, h, spp = 64, 40, 4
    n = 140
    c = np.concatenate([[[0.0, -1001.0, 0.0], [0.0, 1001.0, 0.0]], rng.uniform(-6, 6, (n - 2, 3)) * np.array([1.0, 0.12, 1.0])])
    rad = np.concatenate([[1000.0, 1000.0], rng.uniform(0.05, 0.25, n - 2)])
    arr = spheres(c, rad, rng)
    arr["mat_type"][:2] = 0
    for k, v in (("albedo_r", 0.97), ("albedo_g", 0.93), ("albedo_b", 0.9)):
        arr[k][:2] = v
    cam = r1.create_small_scene(w, h).camera_array().copy()
    cam[0:3] = (0.0, 0.0, 3.0)                       # origin in the middle of the gap
    cam[3:6] = (-2.0, -1.25, 3.0 - 2.0)              # lower_left
    cam[6:9], cam[9:12] = (4.0, 0.0, 0.0), (0.0, 2.5, 0.0)
    sa = r1o.SceneArrays(arr, cam)
    renderer.set_scene_raw(_as_cscene(sa), _as_ccamera(sa))
    got = renderer.render_samples(r1.make_params(w, h, spp, 31, variant=BVH))
    ref = renderer.render_samples(r1.make_params(w, h, spp, 31, variant=binding.VARIANT_REFERENCE))
    assert same(got, ref)
    oimg, orays, osamples = r1o.render_frame(sa, oparams(r1.make_params(w, h, spp, 31)), want_samples=True)
    assert got[1] == orays and got[2].tobytes() == osamples.tobytes()
    n_rays = got[2][:, 3].copy().view(np.uint32)
    deep_and_lit = (n_rays > 33) & (n_rays < 51) & (got[2][:, :3].sum(1) > 0)  # more than 30 stacked attenuations
    assert deep_and_lit.sum() > 20, int(deep_and_lit.sum())  # the deep entries are really unwound with colour


def test_bvh_axis_parallel_rays(renderer):
    """horizontal = vertical = 0 and no lens: every primary ray is exactly (0, 0, -1), so two
    reciprocal direction components are infinite (0 x inf = NaN inside the slab test)."""
    rng = np.random.default_rng(7)
    w, h, spp = 48, 32, 4
    n = 200
    c = rng.uniform(-2, 2, (n, 3))
    c[:, 2] -= 6
    c[:20, :2] = 0  # several exactly on the axis the rays run along
    arr = spheres(c, rng.uniform(0.05, 0.5, n), rng)
    cam = np.zeros(22, np.float32)
    cam[3:6] = (0, 0, -1)                      # lower_left - origin = direction
    cam[12:15], cam[15:18], cam[18:21] = (1, 0, 0), (0, 1, 0), (0, 0, 1)
    sa = r1o.SceneArrays(arr, cam)
    renderer.set_scene_raw(_as_cscene(sa), _as_ccamera(sa))
    got = renderer.render_samples(r1.make_params(w, h, spp, 3, variant=BVH))
    ref = renderer.render_samples(r1.make_params(w, h, spp, 3, variant=binding.VARIANT_REFERENCE))
    assert same(got, ref)
    assert got[1] > w * h * spp  # the axis spheres are hit


@pytest.mark.parametrize("gw,gh,w,h,spp", [(64, 40, 160, 120, 4), (33, 31, 96, 64, 3)])
def test_bvh_big_scene_bit_exact_vs_oracle(renderer, gw, gh, w, h, spp):
    sc = r1.create_grid_scene(w, h, gw, gh)
    renderer.set_scene(sc)
    p = r1.make_params(w, h, spp, 31337, variant=BVH)
    img, rays, samples = renderer.render_samples(p)
    oimg, orays, osamples = r1o.render_frame(oracle_scene(sc), oparams(p), want_samples=True)
    assert rays == orays
    assert samples.tobytes() == osamples.tobytes()
    assert img.tobytes() == oimg.tobytes()


def test_bvh_config5_100k_spheres_equals_the_lds_tiled_sweep(renderer):
    """BASELINE config 5's scene (100 004 spheres): a whole small frame, tree vs exhaustive
    LDS-tiled sweep, every sample; plus oracle spot checks."""
    w, h, spp = 160, 90, 2
    sc = r1.create_grid_scene(w, h, 400, 250)
    renderer.set_scene(sc)
    a = renderer.render_samples(r1.make_params(w, h, spp, 5, variant=binding.VARIANT_PREFILTER))
    assert renderer.launch_info()["kernel"] == binding.VARIANT_PREFILTER
    b = renderer.render_samples(r1.make_params(w, h, spp, 5))  # DEFAULT resolves to the tree above 1 023 spheres
    assert renderer.launch_info()["kernel"] == BVH
    assert same(a, b)
    sa = oracle_scene(sc)
    rng = np.random.default_rng(1)
    xs, ys, ss = rng.integers(0, w, 200), rng.integers(0, h, 200), rng.integers(0, spp, 200)
    rgb, orays = r1o.trace_samples(sa, w, h, 5, xs, ys, ss)
    got = b[2][(ys * w + xs) * spp + ss]
    assert (got[:, 3].copy().view(np.uint32) == orays).all()
    assert got[:, :3].tobytes() == rgb.tobytes()


def test_bvh_shards_and_tiles(renderer):
    """The variant goes through the same tiling/sharding path: tile size and shard count do not
    change the frame."""
    w, h, spp = 200, 120, 3
    renderer.set_scene(r1.create_large_scene(w, h))
    base = renderer.render(r1.make_params(w, h, spp, 77, variant=binding.VARIANT_PREFILTER))
    full = renderer.render(r1.make_params(w, h, spp, 77, variant=BVH))
    assert full[0].tobytes() == base[0].tobytes() and full[1] == base[1]
    odd = renderer.render(r1.make_params(w, h, spp, 77, tile_w=24, tile_h=40, variant=BVH))
    assert odd[0].tobytes() == base[0].tobytes() and odd[1] == base[1]
    total = 0
    for shard in range(3):
        total += renderer.render(r1.make_params(w, h, spp, 77, shard=shard, num_shards=3, variant=BVH))[1]
    assert total == base[1]


# ---- wavefront variant: same device functions, state in HBM between the steps ----


@pytest.mark.parametrize("name,w,h,spp", [("small", 160, 120, 8), ("medium", 200, 120, 5), ("large", 256, 192, 6)])
def test_wavefront_variant_equals_megakernel_and_oracle(renderer, name, w, h, spp):
    sc = MAKE[name](w, h)
    renderer.set_scene(sc)
    a = renderer.render_samples(r1.make_params(w, h, spp, 4242, variant=binding.VARIANT_WAVEFRONT))
    assert renderer.launch_info()["kernel"] == binding.VARIANT_WAVEFRONT
    b = renderer.render_samples(r1.make_params(w, h, spp, 4242, variant=BVH))
    assert same(a, b)
    oimg, orays, osamples = r1o.render_frame(oracle_scene(sc), oparams(r1.make_params(w, h, spp, 4242)), want_samples=True)
    differing = (a[2].view(np.uint32) != osamples.view(np.uint32)).any(1)
    assert differing.mean() <= 1e-5 and abs(a[1] - orays) <= 51 * differing.sum()


def test_wavefront_variant_edges(renderer):
    """Ragged tiles (void slots), shards, bounce limits 1 and 51, a 1x1 frame, an empty scene."""
    renderer.set_scene(r1.create_large_scene(75, 53))
    for kw in (dict(), dict(max_bounces=1), dict(max_bounces=51), dict(tile_w=7, tile_h=5)):
        a = renderer.render_samples(r1.make_params(75, 53, 3, 9, variant=binding.VARIANT_WAVEFRONT, **kw))
        b = renderer.render_samples(r1.make_params(75, 53, 3, 9, variant=BVH, **kw))
        assert same(a, b), kw
    for shard in range(3):
        a = renderer.render(r1.make_params(75, 53, 3, 9, shard=shard, num_shards=3, variant=binding.VARIANT_WAVEFRONT))
        b = renderer.render(r1.make_params(75, 53, 3, 9, shard=shard, num_shards=3, variant=BVH))
        assert a[0].tobytes() == b[0].tobytes() and a[1] == b[1]
    renderer.set_scene(r1.create_large_scene(1, 1))
    assert same(renderer.render_samples(r1.make_params(1, 1, 1, 1, variant=binding.VARIANT_WAVEFRONT)),
                renderer.render_samples(r1.make_params(1, 1, 1, 1, variant=BVH)))
    with pytest.raises(binding.R1Error):  # the whole frame's paths must fit the workspace limit (2^24 slots)
        renderer.set_scene(r1.create_large_scene(4096, 4096))
        renderer.render(r1.make_params(4096, 4096, 2, 1, variant=binding.VARIANT_WAVEFRONT))


# ---- randomised stress: many small scenes, tree vs the reference-form exhaustive kernel ---------------


def test_bvh_random_scene_stress(renderer):
    """40 random scenes (1..400 spheres; uniform, clustered, lattice-like and shell layouts; radii over
    three decades; cameras inside and outside the cloud): the tree kernel must reproduce the
    REFERENCE-form kernel (every sphere, reference arithmetic) on every sample."""
    rng = np.random.default_rng(20261004)
    w, h, spp = 48, 32, 2
    base = r1.create_small_scene(w, h)
    for trial in range(40):
        n = int(rng.integers(1, 401))
        layout = trial % 4
        if layout == 0:
            c = rng.uniform(-10, 10, (n, 3))
        elif layout == 1:
            c = rng.normal(0, 0.8, (n, 3)) + rng.uniform(-6, 6, (1, 3))
        elif layout == 2:
            g = int(np.ceil(np.sqrt(n)))
            ij = np.stack(np.meshgrid(np.arange(g), np.arange(g)), -1).reshape(-1, 2)[:n]
            c = np.concatenate([ij - g / 2 + rng.uniform(0, 0.9, (n, 2)), np.full((n, 1), 0.2)], 1)[:, [0, 2, 1]]
        else:
            u = rng.normal(0, 1, (n, 3))
            c = u / np.linalg.norm(u, axis=1, keepdims=True) * rng.uniform(2, 9)
        rad = np.exp(rng.uniform(np.log(3e-3), np.log(3.0), n))
        if trial % 5 == 0:
            rad[0], c[0] = 500.0, (0, -500.5, 0)  # a ground sphere
        arr = spheres(c, rad, rng)
        cam = base.camera_array().copy()
        if trial % 3 == 0:  # camera in the middle of the cloud
            shift = c.mean(0).astype(np.float32) - cam[0:3]
            cam[0:3] += shift
            cam[3:6] += shift
        sa = r1o.SceneArrays(arr, cam)
        renderer.set_scene_raw(_as_cscene(sa), _as_ccamera(sa))
        got = renderer.render_samples(r1.make_params(w, h, spp, 1000 + trial, variant=BVH))
        ref = renderer.render_samples(r1.make_params(w, h, spp, 1000 + trial, variant=binding.VARIANT_REFERENCE))
        assert same(got, ref), (trial, n, layout)


def test_bvh_300k_spheres_against_oracle_samples(renderer):
    """A 640x480 lattice (307 204 spheres, beyond anything the exhaustive kernels finish quickly): the
    tree kernel against the oracle's brute force on 120 pixel-samples, and determinism of the frame."""
    w, h, spp = 128, 72, 2
    sc = r1.create_grid_scene(w, h, 640, 480)
    renderer.set_scene(sc)
    a = renderer.render_samples(r1.make_params(w, h, spp, 77))
    assert renderer.launch_info()["kernel"] == BVH and renderer.launch_info()["spheres_active"] == 640 * 480 + 4
    b = renderer.render_samples(r1.make_params(w, h, spp, 77, tile_w=16, tile_h=8))
    assert same(a, b)
    sa = oracle_scene(sc)
    rng = np.random.default_rng(3)
    xs, ys, ss = rng.integers(0, w, 120), rng.integers(0, h, 120), rng.integers(0, spp, 120)
    rgb, orays = r1o.trace_samples(sa, w, h, 77, xs, ys, ss)
    got = a[2][(ys * w + xs) * spp + ss]
    assert (got[:, 3].copy().view(np.uint32) == orays).all()
    assert got[:, :3].tobytes() == rgb.tobytes()


def test_non_finite_spheres_are_never_hit(renderer):
    """NaN / inf centres or radius_sq among hittable spheres: the reference's arithmetic never hits them;
    the product drops them at r1_set_scene.  Oracle (raw arithmetic) vs both kernel families."""
    rng = np.random.default_rng(11)
    w, h, spp = 64, 40, 3
    n = 60
    arr = spheres(rng.uniform(-3, 3, (n, 3)), rng.uniform(0.1, 0.5, n), rng)
    arr["center_x"][3], arr["center_y"][7], arr["center_z"][11] = np.nan, np.inf, -np.inf
    arr["radius_sq"][13], arr["radius_sq"][17] = np.inf, np.nan
    sa = r1o.SceneArrays(arr, r1.create_small_scene(w, h).camera_array())
    renderer.set_scene_raw(_as_cscene(sa), _as_ccamera(sa))
    assert renderer.render(r1.make_params(w, h, spp, 2))[1] > w * h * spp
    assert renderer.launch_info()["spheres_active"] == n - 5
    oimg, orays, osamples = r1o.render_frame(sa, oparams(r1.make_params(w, h, spp, 2)), want_samples=True)
    for v in (BVH, binding.VARIANT_PREFILTER, binding.VARIANT_REFERENCE):
        got = renderer.render_samples(r1.make_params(w, h, spp, 2, variant=v))
        assert got[1] == orays and got[2].tobytes() == osamples.tobytes(), v


# ---- per-tile entry nodes --------------------------------------------------------------------------------------


def _look(lookfrom, lookat, vfov, aspect, aperture, focus, vup=(0, 1, 0)):
    """the 22 floats of a camera built as the reference's Camera constructor does (rayweek1.cpp:365-380), in numpy"""
    f = np.float32
    lf, la, up = (np.asarray(x, f) for x in (lookfrom, lookat, vup))
    hh = f(np.tan(np.deg2rad(vfov) / 2))
    hw = f(aspect) * hh
    w = lf - la
    w = w / np.linalg.norm(w)
    u = np.cross(up, w)
    u = u / np.linalg.norm(u)
    v = np.cross(w, u)
    fo = f(focus)
    ll = lf - hw * fo * u - hh * fo * v - fo * w
    return np.concatenate([lf, ll, 2 * hw * fo * u, 2 * hh * fo * v, u, v, w, [f(aperture / 2)]]).astype(f)


ENTRY_CAMERAS = {
    "reference": None,
    "wide lens": ((13, 2, 3), (0, 0, 0), 20, 0.1 * 40, 10.0),           # aperture 4: beams as wide as the lattice cells
    "inside the lattice": ((0.3, 0.25, 0.4), (4, 0.2, 1), 70, 0.05, 3.0),  # the origin between the small spheres
    "looking away": ((13, 2, 3), (26, 4, 6), 30, 0.1, 10.0),            # the lattice behind the camera
    "from above": ((0.5, 30, 0.5), (0, 0, 0), 40, 0.3, 30.0, (0, 0, -1)),
    "grazing": ((-14, 0.21, -14), (14, 0.2, 14), 8, 0.0, 5.0),          # pinhole along the ground through the whole lattice
    "focus behind the origin": ((6, 1, 2), (0, 0.2, 0), 50, 1.5, 0.6),  # focal plane closer than the lens is wide
}


@pytest.mark.parametrize("cam", list(ENTRY_CAMERAS))
@pytest.mark.parametrize("name,w,h,spp,tile", [("large", 192, 128, 3, 32), ("large", 100, 75, 2, 8), ("medium", 128, 96, 4, 16), ("grid", 96, 64, 2, 16)])
def test_entry_nodes_keep_every_hit(renderer, cam, name, w, h, spp, tile):
    """A primary ray starts below the root's inner child at the node all primary rays of its tile stay under (r1_capi.cpp
    compute_entries).  Whatever the camera — a lens wider than the lattice's cells, an origin among the small spheres, a view away
    from the scene, a focal plane closer than the lens radius — the samples are the oracle's to the bit, through the synchronous
    frame (MODE 1) and through the frames-in-flight kernel (MODE 0, tiles summed in the kernel)."""
    sc = r1.create_grid_scene(w, h, 48, 36) if name == "grid" else MAKE[name](w, h)  # (grid: the big-scene kernels, pads measured per node)
    sa = oracle_scene(sc)
    if ENTRY_CAMERAS[cam] is not None:
        sa = r1o.SceneArrays(sa.arrays, _look(*ENTRY_CAMERAS[cam][:2], ENTRY_CAMERAS[cam][2], w / h, *ENTRY_CAMERAS[cam][3:]))
    renderer.set_scene_raw(_as_cscene(sa), _as_ccamera(sa))
    p = r1.make_params(w, h, spp, 99, tile_w=tile, tile_h=tile, variant=BVH)
    img, rays, samples = renderer.render_samples(p)
    oimg, orays, osamples = r1o.render_frame(sa, oparams(p), want_samples=True)
    assert rays == orays
    assert samples.tobytes() == osamples.tobytes()
    assert img.tobytes() == oimg.tobytes()
    hf = binding.HostFrame(w, h)
    renderer.render_async(p, hf)
    renderer.sync()
    assert hf.rays == orays and hf.image.tobytes() == oimg.tobytes()
    hf.close()
