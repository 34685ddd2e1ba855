"""The Sphere extension on the GPU (PT_SPHERE).  The reference has no sphere
(src/Object.hs:191-194 is a TODO), so none of this is reference parity: the
device is checked against the oracle's O_SPHERE definition
(tests/test_oracle_sphere.py pins that one with geometry known answers), to
the same bars as the reference's own forms -- primary hits bit-exact (fp64),
fp32 distances to 1e-5 where well conditioned, images path-wise on the same
Philox stream -- through every object-list path, and BASELINE.json configs[3]
("random-sphere scene") at its full size."""
import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pa():
    import pathtracer_amd
    return pathtracer_amd


def _mismatch_fraction(dev_sum, ora_sum, rel, abs_):
    bad = np.abs(dev_sum - ora_sum) > (rel * np.abs(ora_sum) + abs_)
    return bad.any(axis=-1).mean()


def _mixed_small_scene(pa):
    return [pa.Object((1, 1, 1), 0, False, pa.InfinitePlane((0, 0, 400), (0, 0, -1)), "wall"),
            pa.Object((1, 1, 1), 4.0, True, pa.Sphere((0.0, 150.0, 150.0), 60.0), "lamp"),
            pa.Object((0.9, 0.5, 0.3), 0, False, pa.Sphere((-60.0, 0.0, 200.0), 50.0), "diffuse ball"),
            pa.Object((0.8, 0.8, 0.9), 0, True, pa.Sphere((70.0, -20.0, 220.0), 45.0), "mirror ball"),
            pa.Object((0.3, 0.9, 0.4), 0, False, pa.Disk((0.0, -80.0, 200.0), (0.0, 1.0, 0.0), 150.0), "floor disk")]


@pytest.mark.parametrize("n,seed", [(5, 1), (8, 2), (9, 3), (100, 4), (1000, 1)])
def test_primary_hits_random_spheres(pa, n, seed):
    objs = pa.random_sphere_scene(n, seed)
    H, W = 96, 128
    with pa.Renderer(pa.Params(H, W, 120.0, 1, 4), objs) as r:
        idx, t, nrm = r.primary_hits()
    oidx, ot, onrm = O.primary_hits(O.params(H, W, 120.0, 4), O.scene_from(objs))
    assert np.array_equal(idx, oidx)
    assert np.array_equal(t, ot)
    assert np.array_equal(nrm, onrm)
    assert len(np.unique(oidx)) > (2 if n > 8 else 1)


def test_dev_distance_sphere(pa):
    rng = np.random.default_rng(17)
    n = 20000
    org = (rng.normal(size=(n, 3)) * 60.0).astype(np.float32)
    vec = (rng.normal(size=(n, 3)) * rng.uniform(0.3, 3.0, size=(n, 1))).astype(np.float32)
    with pa.Renderer(pa.Params(8, 8, 500.0, 1, 4)) as small:
        for obj in (pa.Object((1, 1, 1), 0, False, pa.Sphere((5.0, -3.0, 20.0), 25.0)),
                    pa.Object((1, 1, 1), 0, False, pa.Sphere((0.0, 0.0, 0.0), 90.0)),      # most origins inside
                    pa.Object((1, 1, 1), 0, False, pa.Sphere((40.0, 40.0, -10.0), 3.0))):
            dev = small.dev_distance(obj, org, vec)
            arr = O.scene_from([obj])
            c = np.array(obj.form.center); r = obj.form.radius
            bad = hits = 0
            for k in range(n):
                d = O.C.c_double()
                ok = O.L.o_distance_from(O.vec(org[k]), O.vec(vec[k]), arr, O.C.byref(d))
                if ok != (not np.isnan(dev[k])):
                    bad += 1                                   # grazing rays: fp32 and fp64 may disagree on l.l <= r^2
                    continue
                if ok:
                    hits += 1
                    # conditioning: h = sqrt(r^2 - l.l) loses digits when the ray grazes; |oc| >> r amplifies too
                    o64, v64 = org[k].astype(np.float64), vec[k].astype(np.float64)
                    oc = o64 - c
                    m = (oc @ v64) / (v64 @ v64)
                    l2 = (oc - m * v64) @ (oc - m * v64)
                    cond = (oc @ oc + r * r) / max(r * r - l2, 1e-30)
                    tol = max(1e-5, 8 * 6e-8 * cond) * max(abs(d.value), abs(m)) + 1e-6
                    assert abs(dev[k] - d.value) <= tol, (obj.form, k, dev[k], d.value, cond)
            assert bad <= n * 2e-3 and hits > 5, (obj.form, bad, hits)


@pytest.mark.parametrize("flags", [0, 4, 1])
def test_pathwise_parity_small_sphere_scene(pa, flags):
    """Five objects (by-value list, trace_static_kernel / strict kernel): lamp
    sphere, diffuse and mirror balls, a disk, a wall."""
    objs = _mixed_small_scene(pa)
    H, W, spp, B = 64, 80, 12, 6
    with pa.Renderer(pa.Params(H, W, 100.0, spp, B), objs, seed=4, flags=flags) as r:
        r.trace(spp)
        dev = r.read_sum()
        segs = r.segments
    tr = O.Tracer(O.params(H, W, 100.0, B, O.PHILOX, 4), O.scene_from(objs))
    ora = tr.trace(spp)
    assert ora.sum() > 0
    assert _mismatch_fraction(dev, ora, 1e-3, 1e-4) <= 0.03
    assert abs(dev.mean() - ora.mean()) <= 0.02 * ora.mean()
    expect = tr.segments.value - (H * W * spp - H * W if not (flags & 1) else 0)
    assert abs(segs - expect) <= 0.005 * expect


@pytest.mark.parametrize("flags", [0, 2, 4, 8])
@pytest.mark.parametrize("n,seed,bounces", [(8, 6, 8), (40, 7, 6), (300, 8, 5), (1025, 10, 4)])
def test_pathwise_parity_random_spheres(pa, n, seed, bounces, flags):
    """Every object-list path: by value (n <= 8), BVH (default), LDS tiles
    (flags 2), strict kernel (4), scalar stream without the BVH (8)."""
    if n <= 8 and flags in (2, 8):
        pytest.skip("small scenes never take the streamed paths")
    objs = pa.random_sphere_scene(n, seed)
    H, W, spp = 40, 56, 6
    with pa.Renderer(pa.Params(H, W, 60.0, spp, bounces), objs, seed=3, flags=flags) as r:
        r.trace(spp)
        dev = r.read_sum()
    tr = O.Tracer(O.params(H, W, 60.0, bounces, O.PHILOX, 3), O.scene_from(objs))
    ora = tr.trace(spp)
    assert _mismatch_fraction(dev, ora, 1e-3, 1e-4) <= 0.03
    assert abs(dev.mean() - ora.mean()) <= 0.02 * max(ora.mean(), 1e-3)


@pytest.mark.parametrize("n,seed", [(40, 7), (300, 8), (2500, 11)])
def test_spatial_index_equals_linear_scan_with_spheres(pa, n, seed):
    objs = pa.random_sphere_scene(n, seed)
    objs = objs[: n // 2] + pa.random_disk_scene(n, seed + 1)[n // 2:]           # spheres and disks mixed
    H, W, spp = 72, 56, 5
    out = []
    for flags in (0, pa.PT_FLAG_NO_BVH, pa.PT_FLAG_NO_STEAL, pa.PT_FLAG_NO_STEAL | pa.PT_FLAG_NO_BVH):
        with pa.Renderer(pa.Params(H, W, 60.0, spp, 6), objs, seed=3, flags=flags) as r:
            r.trace(spp)
            out.append((r.read_sum_f32(), r.segments, r.primary_hits()))
    assert out[0][1] == out[1][1] == out[2][1] == out[3][1]
    assert np.array_equal(out[2][0], out[3][0])             # strict sample order: bit for bit
    # default mode: the same paths, summed per pixel in the order the lanes happened to take them
    assert np.allclose(out[0][0], out[1][0], rtol=2e-6, atol=1e-6) and np.allclose(out[0][0], out[2][0], rtol=2e-6, atol=1e-6)
    for q in (1, 2, 3):
        for x, y in zip(out[0][2], out[q][2]):
            assert np.array_equal(x, y)


def test_config4_thousand_spheres_full_size(pa):
    """BASELINE.json configs[3] as written: 1,000-object random-sphere scene,
    diffuse/specular mix, 3840x2160, 512 spp, depth 8."""
    H, W, spp, B = 3840, 2160, 512, 8
    objs = pa.random_sphere_scene(1000, 1)
    rng = np.random.default_rng(46)
    px = np.unique(np.concatenate([rng.integers(0, H * W, 160), [0, W - 1, (H - 1) * W, H * W - 1, 63, 64]])).astype(np.int32)
    with pa.Renderer(pa.Params(H, W, 500.0, spp, B), objs, seed=1) as r:
        r.trace(spp)
        dev = r.read_sum().reshape(-1, 3)
        segs = r.segments
    ora, _ = O.trace_pixels(O.params(H, W, 500.0, B, O.PHILOX, 1), O.scene_from(objs), px, spp)
    d = dev[px]
    close = ((np.abs(d - ora) <= 1e-3 * np.abs(ora)) | (np.abs(d - ora) <= 1e-4)).all(axis=1)
    # A pixel's sum moves as soon as ONE of its 512 paths flips a hit in fp32; mirror chains over curved
    # surfaces amplify rounding more than the disks' planes do (measured: 95 % of the spot pixels agree
    # to 1e-3, against > 97 % for the disk variant).  A flipped path moves a sum by at most the largest
    # emittance (5) per channel.
    assert close.mean() >= 0.90, close.mean()
    assert np.abs(d - ora).max() <= 5.0 * 6
    assert np.all(np.isfinite(dev)) and dev.min() >= 0.0
    assert np.abs(d.mean(0) - ora.mean(0)).max() <= 0.02 * max(ora.mean(), 1e-3)
    assert H * W * spp < segs <= H * W * (1 + spp * (B - 1))
