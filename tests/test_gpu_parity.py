"""GPU parity tests: the HIP path (through the C ABI, libpt.so) against the CPU
oracle, which is an fp64 RESTATEMENT of the reference (oracle/pt_oracle.h), on
the same seeded inputs.

Bars (BASELINE.json north_star):
  * integer / index / byte results bit-exact: Philox words, primary-hit object
    indices, RGB8 bytes;
  * primary-hit t and normals within 1e-5 relative;
  * fp32 device functions (intersection, scatter) within the tolerance written
    in each test;
  * images: path-wise agreement with the oracle on the SAME Philox stream, and
    statistical agreement with the oracle on the reference's own StdGen stream
    (thresholds from SURVEY.md 8c, fixed here).
"""
import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pa():
    import pathtracer_amd
    return pathtracer_amd


@pytest.fixture(scope="module")
def small(pa):
    r = pa.Renderer(pa.Params(8, 8, 500.0, 1, 4))
    yield r
    r.close()


# ---------------------------------------------------------------- RNG --------
def test_philox_bit_exact(small):
    rng = np.random.default_rng(1)
    c0 = rng.integers(0, 2**32, 4096, dtype=np.uint64).astype(np.uint32)
    c1 = rng.integers(0, 2**32, 4096, dtype=np.uint64).astype(np.uint32)
    c0[:3] = [0, 0xFFFFFFFF, 0x243F6A88]
    c1[:3] = [0, 0xFFFFFFFF, 0x85A308D3]
    key = 0x13198A2E
    dev = small.dev_philox(c0, c1, key)
    out = (O.C.c_uint32 * 2)()
    for k in range(len(c0)):
        O.L.o_philox2x32_10(int(c0[k]), int(c1[k]), key, out)
        assert (int(dev[k, 0]), int(dev[k, 1])) == (out[0], out[1])
    # Random123 known answer, key 0x13198a2e, ctr (243f6a88, 85a308d3)
    assert (int(dev[2, 0]), int(dev[2, 1])) == (0xDD7CE038, 0xF62A4C12)


# ------------------------------------------------------------ geometry -------
@pytest.mark.parametrize("H,W,depth", [(200, 200, 500.0), (64, 64, 500.0), (301, 257, 500.0), (33, 17, 40.0),
                                        (1, 1, 500.0), (540, 960, 500.0)])
def test_primary_hits_default_scene(pa, H, W, depth):
    with pa.Renderer(pa.Params(H, W, depth, 1, 4)) as r:
        idx, t, nrm = r.primary_hits()
    oidx, ot, onrm = O.primary_hits(O.params(H, W, depth, 4), O.default_scene())
    assert np.array_equal(idx, oidx)                     # bit-exact, tie lines included
    hit = oidx >= 0
    assert np.all(np.abs(t[hit] - ot[hit]) <= 1e-5 * np.abs(ot[hit]))
    assert np.all(np.abs(nrm - onrm) <= 1e-5)
    # the fp64 device path follows the reference's operation order: not just close
    assert np.array_equal(t, ot)


def test_primary_hits_tie_lines(pa):
    """Columns j = W/2-100 and rows i = H/2-100 tie in exact arithmetic
    (SURVEY.md 7): the winner is decided by fp64 rounding and must still match."""
    H = W = 500
    with pa.Renderer(pa.Params(H, W, 500.0, 1, 4)) as r:
        idx, t, _ = r.primary_hits()
    oidx, ot, _ = O.primary_hits(O.params(H, W, 500.0, 4), O.default_scene())
    assert np.array_equal(idx[:, W // 2 - 100], oidx[:, W // 2 - 100])
    assert np.array_equal(idx[H // 2 - 100, :], oidx[H // 2 - 100, :])
    assert np.array_equal(idx, oidx)
    assert idx[H // 2, W // 2] == 0 and t[H // 2, W // 2] == 100.0     # analytic KAT


@pytest.mark.parametrize("n,seed", [(5, 1), (8, 2), (9, 3), (100, 4), (1000, 1)])
def test_primary_hits_random_disks(pa, n, seed):
    objs = pa.random_disk_scene(n, seed)
    H, W = 96, 128
    with pa.Renderer(pa.Params(H, W, 120.0, 1, 4), objs) as r:
        idx, t, nrm = r.primary_hits()
    oidx, ot, onrm = O.primary_hits(O.params(H, W, 120.0, 4), O.scene_from(objs))
    assert np.array_equal(idx, oidx)
    assert np.array_equal(t, ot)
    assert np.array_equal(nrm, onrm)
    assert len(np.unique(oidx)) > (2 if n > 8 else 1)     # the scene is actually visible


def _rand_vecs(rng, n, scale=1.0):
    return (rng.normal(size=(n, 3)) * scale).astype(np.float32)


def test_dev_distance_plane_and_disk(pa, small):
    rng = np.random.default_rng(7)
    n = 20000
    org = _rand_vecs(rng, n, 50.0)
    vec = _rand_vecs(rng, n, 1.0)
    for obj in (pa.infPlane, pa.infPlane2, pa.light, pa.disk,
                pa.Object((1, 1, 1), 0, False, pa.InfinitePlane((3, -2, 9), (0.3, -2.0, 1.1)))):
        dev = small.dev_distance(obj, org, vec)
        arr = O.scene_from([obj])
        # conditioning of t = ((p-o).n) / (v.n_hat) in fp32: both dot products can cancel
        f = obj.form
        pt = np.array(f.center if isinstance(f, pa.Disk) else f.point, np.float64)
        nr = np.array(f.normal, np.float64)
        nh = nr / np.linalg.norm(nr)
        o64, v64 = org.astype(np.float64), vec.astype(np.float64)
        num, den = (pt - o64) @ nr, v64 @ nh
        cond = (np.abs(pt @ nr) + np.abs(o64 * nr).sum(1)) / np.abs(num) + np.abs(v64 * nh).sum(1) / np.abs(den)
        bad = relaxed = hits = 0
        for k in range(n):
            d = O.C.c_double()
            ok = O.L.o_distance_from(O.vec(org[k]), O.vec(vec[k]), arr, O.C.byref(d))
            if ok != (not np.isnan(dev[k])):
                # an fp32/fp64 disagreement is legitimate only on a guard boundary
                bad += 1
                continue
            if ok:
                # north_star's bar: 1e-5 relative.  It holds wherever the fp32 quotient is well conditioned; a ray
                # whose two dot products cancel (cond large) is allowed eps*cond instead, and such rays are COUNTED
                hits += 1
                err = abs(dev[k] - d.value)
                if err > 1e-5 * abs(d.value) + 1e-6:
                    relaxed += 1
                    assert 4 * 6e-8 * cond[k] > 1e-5, (obj.name, k, dev[k], d.value, cond[k])   # only ill-conditioned rays
                    assert err <= 4 * 6e-8 * cond[k] * abs(d.value) + 1e-6, (obj.name, k, dev[k], d.value, cond[k])
        print(f"{obj.name}: {hits} hits, {relaxed} beyond 1e-5 relative (all with cond > 42), "
              f"{bad} guard-boundary disagreements of {n} rays")
        assert bad <= n * 2e-3, (obj.name, bad)
        assert relaxed <= hits * 5e-3, (obj.name, relaxed, hits)          # measured: <= 0.2 % of the hits


def test_dev_diffuse(small):
    rng = np.random.default_rng(8)
    n = 20000
    nrm = _rand_vecs(rng, n, 1.0) * rng.uniform(0.2, 3.0, size=(n, 1)).astype(np.float32)
    nrm[:6] = [[0, 0, -1], [0, 1, 0], [-1, 0, 0], [0, 0, 1], [0, 0, 2.5], [1, 0, -1]]
    u = rng.integers(0, 2**24, size=(n, 2)).astype(np.float32) / np.float32(2**24)
    dev = small.dev_diffuse(nrm, u)
    for k in range(n):
        ref = O.L.o_diffuse_u(float(u[k, 0]), float(u[k, 1]), O.vec(nrm[k])).t()
        ln = np.linalg.norm(nrm[k].astype(np.float64))
        # fp32 rotation with hardware sin/cos: absolute 2e-5 of the vector's length
        assert np.max(np.abs(dev[k] - np.array(ref))) <= 2e-5 * ln, (k, dev[k], ref)


def test_dev_specular(small):
    rng = np.random.default_rng(9)
    n = 20000
    vec = _rand_vecs(rng, n, 1.0) * rng.uniform(0.2, 3.0, size=(n, 1)).astype(np.float32)
    nrm = _rand_vecs(rng, n, 1.0) * rng.uniform(0.2, 3.0, size=(n, 1)).astype(np.float32)
    vec[:4] = [[0, 0, 1], [0, 0, -1], [1, 0, 0], [0.3, 0.4, -0.5]]
    nrm[:4] = [[0, 0, 1], [0, 0, 1], [0, 0, 1], [0, 0, 1]]
    u = rng.integers(0, 2**24, size=(n, 2)).astype(np.float32) / np.float32(2**24)
    dev = small.dev_specular(vec, nrm, u)
    worst = 0.0
    for k in range(n):
        ref = np.array(O.L.o_specular_u(float(u[k, 0]), float(u[k, 1]), 40.0, O.vec(vec[k]), O.vec(nrm[k])).t())
        ln = np.linalg.norm(vec[k].astype(np.float64))
        worst = max(worst, np.max(np.abs(dev[k] - ref)) / ln)
    # polynomial asin + hardware sin/cos/rsq in fp32
    assert worst <= 5e-5, worst
    # propSpecular (test/Spec.hs:144-155): noise 0 => mirror law
    dev0 = small.dev_specular(vec, nrm, u, noise=0.0)
    nh = nrm / np.linalg.norm(nrm, axis=1, keepdims=True)
    vin = vec / np.linalg.norm(vec, axis=1, keepdims=True)
    vout = dev0 / np.linalg.norm(dev0, axis=1, keepdims=True)
    assert np.max(np.abs(np.sum(-vin * nh, axis=1) - np.sum(vout * nh, axis=1))) < 1e-4
    proj = lambda v: v - np.sum(v * nh, axis=1, keepdims=True) * nh
    assert np.max(np.abs(proj(vec) - proj(dev0))) < 1e-4 * 3.0


# ------------------------------------------------------------- images --------
def _mismatch_fraction(dev_sum, ora_sum, rel=2e-4, abs_=1e-5):
    bad = np.abs(dev_sum - ora_sum) > (rel * np.abs(ora_sum) + abs_)
    return bad.any(axis=2).mean()


@pytest.mark.parametrize("flags", [0, 1])
@pytest.mark.parametrize("H,W,spp,bounces", [(64, 64, 8, 4), (200, 200, 16, 4), (48, 80, 4, 8), (31, 29, 5, 30)])
def test_pathwise_parity_same_stream(pa, H, W, spp, bounces, flags):
    """Same Philox draws on both sides: every path takes the same decisions
    unless an fp32/fp64 rounding difference flips a hit, so almost every pixel
    sum must agree to fp32 accuracy."""
    with pa.Renderer(pa.Params(H, W, 500.0, spp, bounces), seed=11, flags=flags) as r:
        r.trace(spp)
        dev = r.read_sum()
        segs = r.segments
    p = O.params(H, W, 500.0, bounces, O.PHILOX, 11)
    tr = O.Tracer(p, O.default_scene())
    ora = tr.trace(spp)
    assert _mismatch_fraction(dev, ora) <= 0.01
    assert abs(dev.mean() - ora.mean()) <= 2e-3 * ora.mean()
    # closest-hit searches: the reference's count, minus the reused primary
    expect = tr.segments.value
    if flags == 0:
        expect = expect - H * W * spp + H * W
    assert abs(segs - expect) <= 0.002 * expect


def test_pathwise_parity_on_the_tie_lines(pa):
    """The pixels of the analytic tie lines (column j = W/2 - 100, row i = H/2 - 100) path by path.
    Their primary hit point lies on two planes at once, so whether the first bounce "sees" the second plane hangs on
    the last bit of that point: the device carries it in fp32, the oracle in fp64, and a plain path-wise comparison
    flips there by construction (the full-size spot checks therefore leave these pixels to the statistical crops).
    Here the oracle is handed the device's own primary segment -- the fp64 hit (bit-exact with the oracle's own, see
    above) rounded to fp32 exactly as the kernel rounds it -- and everything from the first bounce on is compared
    like any other pixel: same Philox draws, same decisions unless fp32/fp64 rounding flips a hit."""
    H = W = 256
    SPP, B = 64, 8
    tie = np.unique(np.concatenate([np.arange(H) * W + (W // 2 - 100), (H // 2 - 100) * W + np.arange(W)])).astype(np.int32)
    with pa.Renderer(pa.Params(H, W, 500.0, SPP, B), seed=7) as r:
        idx, t, _ = r.primary_hits()
        r.trace(SPP)
        dev = r.read_sum().reshape(-1, 3)[tie]
    p = O.params(H, W, 500.0, B, O.PHILOX, 7)
    oidx, ot, _ = O.primary_hits(p, O.default_scene())
    assert np.array_equal(idx, oidx) and np.array_equal(t, ot)
    d64 = O.primary_directions(p, tie)
    # trace_static_kernel / primary_fill_kernel: P0 = (float)(t * d), d0 = (float)d, component by component
    p32 = (d64 * t.reshape(-1)[tie][:, None]).astype(np.float32).astype(np.float64)
    d32 = d64.astype(np.float32).astype(np.float64)
    ora, _ = O.trace_pixels_from_primary(p, O.default_scene(), tie, idx.reshape(-1)[tie], p32, d32, SPP)
    # the tie lines really are what this test is about: with the oracle's own fp64 hit point a good part of them flips
    plain, _ = O.trace_pixels(p, O.default_scene(), tie, SPP)
    flips_plain = (np.abs(dev - plain) > 2e-4 * np.abs(plain) + 1e-5).any(axis=1).mean()
    flips = (np.abs(dev - ora) > 2e-4 * np.abs(ora) + 1e-5).any(axis=1).mean()
    assert flips <= 0.02, (flips, flips_plain)
    assert abs(dev.mean() - ora.mean()) <= 2e-3 * ora.mean()
    print(f"tie-line pixels off: {flips:.4f} with the device's primary point, {flips_plain:.4f} with the oracle's own")


@pytest.mark.parametrize("flags", [0, 2])
@pytest.mark.parametrize("n,seed,bounces", [(6, 5, 6), (8, 6, 8), (40, 7, 6), (300, 8, 5), (1024, 9, 4),
                                            (1025, 10, 4), (2500, 11, 3)])
def test_pathwise_parity_random_disks(pa, n, seed, bounces, flags):
    """Diffuse + specular mix through every object-list path: by-value SGPR list
    (n <= 8), one LDS tile (n <= 1024), LDS tiles streamed in workgroup lockstep
    (n > 1024), scalar-load stream from L2 by default; flags = PT_FLAG_LDS_TILES stages the list in LDS."""
    if flags and n <= 8:
        pytest.skip("small scenes never take the streamed paths")
    objs = pa.random_disk_scene(n, seed)
    H, W, spp = 40, 56, 6
    if n > 1000:
        H, W, spp = 24, 40, 3
    with pa.Renderer(pa.Params(H, W, 60.0, spp, bounces), objs, seed=3, flags=flags) as r:
        r.trace(spp)
        dev = r.read_sum()
    tr = O.Tracer(O.params(H, W, 60.0, bounces, O.PHILOX, 3), O.scene_from(objs))
    ora = tr.trace(spp)
    # specular chains amplify rounding differences more than diffuse ones
    assert _mismatch_fraction(dev, ora, rel=1e-3, abs_=1e-4) <= 0.03
    assert abs(dev.mean() - ora.mean()) <= 0.02 * max(ora.mean(), 1e-3)


def test_statistical_parity_reference_stream(pa):
    """Device (Philox) against the oracle on the reference's own StdGen stream:
    independent estimates of the same image.  z = (mu_dev - mu_ref) /
    sqrt(var/N + var/N_ref) per pixel and channel.  Thresholds:
    fraction |z| > 4 below 1e-3; per-channel mean bias below 0.5 %; RMSE within
    1.2x of the predicted value."""
    H, W, B, N, NREF = 96, 96, 8, 256, 256
    with pa.Renderer(pa.Params(H, W, 500.0, N, B), seed=5) as r:
        r.trace(N)
        mu_dev = r.read_sum() / N
    tr = O.Tracer(O.params(H, W, 500.0, B, O.STDGEN), O.default_scene())
    tr.trace(NREF)
    mu_ref = tr.sum / NREF
    var = np.maximum(tr.sumsq / NREF - mu_ref**2, 0.0) * NREF / (NREF - 1)
    se2 = var / N + var / NREF
    ok = se2 > 0
    z = (mu_dev[ok] - mu_ref[ok]) / np.sqrt(se2[ok])
    assert (np.abs(z) > 4).mean() < 1e-3
    for k in range(3):
        bias = abs(mu_dev[..., k].mean() - mu_ref[..., k].mean()) / mu_ref[..., k].mean()
        assert bias < 5e-3, (k, bias)
    rmse = np.sqrt(np.mean((mu_dev - mu_ref) ** 2))
    pred = np.sqrt(np.mean(se2))
    assert rmse <= 1.2 * pred, (rmse, pred)
    assert rmse >= 0.5 * pred      # and the two estimates really are independent


def test_max_bounces_one_is_black(pa):
    """Analytic KAT: the only light is behind the camera, so
    with maxBounces = 1 every pixel is black."""
    for flags in (0, 1):
        with pa.Renderer(pa.Params(32, 32, 500.0, 4, 1), flags=flags) as r:
            r.trace(4)
            assert not r.read_sum().any()


def test_primary_light_scene(pa):
    """A light in front of the camera: radiance = emittance * white, per sample."""
    objs = [pa.Object((1, 1, 1), 3.0, True, pa.Disk((0, 0, 50), (0, 0, -1), 10.0), "front light"),
            pa.infPlane]
    for flags in (0, 1):
        with pa.Renderer(pa.Params(64, 64, 500.0, 4, 4), objs, flags=flags) as r:
            r.trace(4)
            dev = r.read_sum()
        tr = O.Tracer(O.params(64, 64, 500.0, 4, O.PHILOX, 0), O.scene_from(objs))
        ora = tr.trace(4)
        assert dev[32, 32, 0] == 12.0
        assert _mismatch_fraction(dev, ora) <= 0.01


def test_chunking_is_bit_exact(pa):
    """Under PT_FLAG_NO_STEAL k calls of pt_trace(1) == one call of pt_trace(k), bit
    for bit, and the canvas is the running SUM (traces !! k), element 0 black."""
    P = pa.Params(50, 70, 500.0, 6, 5)
    with pa.Renderer(P, seed=2, flags=pa.PT_FLAG_NO_STEAL) as a, pa.Renderer(P, seed=2, flags=pa.PT_FLAG_NO_STEAL) as b:
        assert not a.read_sum_f32().any()
        a.trace(6)
        for _ in range(6):
            b.trace(1)
        assert np.array_equal(a.read_sum_f32(), b.read_sum_f32())
        assert a.samples_done == b.samples_done == 6
        assert a.segments == b.segments - 5 * 50 * 70      # b recomputed the primary hit per launch
        a.reset()
        assert not a.read_sum_f32().any() and a.samples_done == 0 and a.segments == 0


@pytest.mark.parametrize("H,W,spp,bounces", [(50, 70, 6, 5), (64, 64, 40, 8), (33, 31, 17, 30)])
def test_work_sharing_renders_the_same_paths(pa, H, W, spp, bounces):
    """Default mode (idle lanes help other pixels of their wave) against the strict
    per-lane mode: same Philox-keyed paths, so the same number of closest-hit
    searches and the same sums up to fp32 summation order; chunked calls likewise."""
    P = pa.Params(H, W, 500.0, spp, bounces)
    with pa.Renderer(P, seed=6) as steal, pa.Renderer(P, seed=6, flags=pa.PT_FLAG_NO_STEAL) as strict, \
            pa.Renderer(P, seed=6) as chunked:
        steal.trace(spp)
        strict.trace(spp)
        for n in (1, 2, spp - 3):
            chunked.trace(n)
        a, b, c = steal.read_sum_f32(), strict.read_sum_f32(), chunked.read_sum_f32()
        assert steal.segments == strict.segments
        assert chunked.segments == strict.segments      # the primary hits are searched once and kept for later launches
    assert np.allclose(a, b, rtol=2e-6, atol=1e-6)
    assert np.allclose(c, b, rtol=2e-6, atol=1e-6)
    assert b.sum() > 0


@pytest.mark.parametrize("H,W,spp", [(40, 56, 512), (33, 32, 700), (64, 64, 513)])
def test_sample_slices_of_a_large_launch(pa, H, W, spp):
    """A launch of >= 512 samples runs as two sample slices (the last 1/16 of the
    samples in workgroups of their own, summed into the canvas by a second kernel):
    same paths, same count of searches, sums equal up to fp32 summation order --
    against the strict kernel (never sliced), against two launches that are too
    small to be sliced, and continuing a canvas that already holds samples."""
    P = pa.Params(H, W, 500.0, spp, 8)
    with pa.Renderer(P, seed=9) as sliced, pa.Renderer(P, seed=9, flags=pa.PT_FLAG_NO_STEAL) as strict, \
            pa.Renderer(P, seed=9) as small, pa.Renderer(P, seed=9) as cont:
        sliced.trace(spp)
        strict.trace(spp)
        k = 0
        while k < spp:                                   # launches of < 512 samples: whole
            n = min(200, spp - k)
            small.trace(n); k += n
        cont.trace(5)
        cont.trace(spp - 5)                              # a sliced launch on top of a non-empty canvas
        a, b, c, d = sliced.read_sum_f32(), strict.read_sum_f32(), small.read_sum_f32(), cont.read_sum_f32()
        launches = -(-spp // 200)
        assert sliced.segments == strict.segments
        assert launches > 1
        assert small.segments == strict.segments         # (the primary hits are searched by the first launch only)
        assert cont.segments == strict.segments
        assert sliced.samples_done == spp
    for x in (a, c, d):
        assert np.allclose(x, b, rtol=4e-6, atol=1e-5)
    assert b.sum() > 0


def test_kept_primary_hits_follow_the_scene(pa):
    """The fp64 primary hits are searched by the first pt_trace and kept for later
    launches of the same camera and scene; pt_scene_upload must drop them, and
    pt_reset must not need them redone."""
    H, W, spp = 48, 40, 9
    P = pa.Params(H, W, 120.0, spp, 6)
    other = [pa.Object((1, 1, 1), 0, False, pa.InfinitePlane((0, 0, 300), (0, 0, -1)), "wall"),
             pa.Object((1, 1, 1), 3.0, True, pa.Disk((0.0, 0.0, -5.0), (0.0, 0.0, 1.0), 60.0), "lamp"),
             pa.Object((0.8, 0.3, 0.3), 0, False, pa.Disk((10.0, -10.0, 150.0), (0.2, 0.1, -1.0), 70.0), "disk")]
    with pa.Renderer(P, seed=5) as r, pa.Renderer(P, other, seed=5) as want_other, pa.Renderer(P, seed=5) as want_default:
        r.trace(spp)                                     # default scene: its primary hits are now kept
        first = r.read_sum_f32()
        r.upload(other)
        r.reset()
        r.trace(4); r.trace(spp - 4)                     # second launch reads what the first one of THIS scene kept
        want_other.trace(spp)
        assert np.allclose(r.read_sum_f32(), want_other.read_sum_f32(), rtol=2e-6, atol=1e-6)
        assert r.segments == want_other.segments
        r.upload(pa.objects)
        r.reset()
        r.trace(spp)
        want_default.trace(spp)
        assert np.array_equal(r.read_sum_f32(), want_default.read_sum_f32()) and np.array_equal(first, want_default.read_sum_f32())
        assert r.segments == want_default.segments
    assert first.sum() > 0


def test_traces_generator(pa):
    """Lib.traces: element k = sum of k samples; element 0 black; lazily extended."""
    P = pa.Params(16, 24, 500.0, 3, 4)
    g = pa.traces(P, seed=4)
    e0, e1, e2 = next(g), next(g), next(g)
    g.close()
    assert e0.shape == (16, 24, 3) and e0.dtype == np.float64 and not e0.any()
    with pa.Renderer(P, seed=4) as r:
        r.trace(1)
        assert np.array_equal(r.read_sum(), e1)
        r.trace(1)
        assert np.array_equal(r.read_sum(), e2)


def test_sample_split_union(pa):
    """Multi-GPU sample split: contexts rendering global samples
    {g, g+G, ...} sum to the single-context image up to fp32 summation order."""
    P = pa.Params(40, 40, 500.0, 8, 6)
    with pa.Renderer(P, seed=9) as full:
        full.trace(8)
        want = full.read_sum()
        segs = full.segments
    G = 4
    got = np.zeros_like(want)
    total = 0
    for g in range(G):
        with pa.Renderer(P, seed=9, sample_offset=g, sample_stride=G) as r:
            r.trace(8 // G)
            got += r.read_sum()
            total += r.segments
    assert np.allclose(got, want, rtol=1e-5, atol=1e-5)
    assert total == segs + (G - 1) * 40 * 40


# ------------------------------------------------------------- output --------
def test_sample_split_union_with_sliced_launches(pa):
    """The same union when every shard's launch is large enough to run as two
    sample slices (>= 512 samples) and strides its global sample indices: three
    shards of 520 against one context's 1560 samples."""
    H, W, G, per = 32, 48, 3, 520
    P = pa.Params(H, W, 500.0, G * per, 8)
    with pa.Renderer(P, seed=11) as full:
        full.trace(G * per)
        want = full.read_sum()
        segs = full.segments
    total = np.zeros_like(want)
    tsegs = 0
    for g in range(G):
        with pa.Renderer(P, seed=11, sample_offset=g, sample_stride=G) as r:
            r.trace(per)
            total += r.read_sum()
            tsegs += r.segments
    assert tsegs == segs + (G - 1) * H * W                # every shard searches its primary hits once
    assert np.allclose(total, want, rtol=1e-5, atol=1e-5)
    assert want.sum() > 0


def test_rgb8_reference_mode_bit_exact(pa):
    """repa3ToImage + listToPixelRGB8: round-half-even(255*x), wrap mod 256,
    transposed layout (j*H + i)*3 + k -- compared byte for byte."""
    H, W = 70, 45
    rng = np.random.default_rng(3)
    canvas = rng.uniform(0, 3.0, size=(H, W, 3)).astype(np.float32)
    flat = canvas.reshape(-1)
    ties = (np.arange(0, 600) + 0.5) / 255.0               # exact .5 cases where representable
    flat[: len(ties)] = ties.astype(np.float32)
    flat[len(ties): len(ties) + 4] = [0.0, 1.0, 256.0 / 255.0, 1000.25]
    with pa.Renderer(pa.Params(H, W, 500.0, 1, 4)) as r:
        r.write_sum_f32(canvas, samples_done=3)
        dev = r.read_rgb8(pa.PT_RGB8_REFERENCE)
        mean = r.read_rgb8(pa.PT_RGB8_MEAN_CLAMP)
        gam = r.read_rgb8(pa.PT_RGB8_MEAN_GAMMA)
    p = O.params(H, W, 500.0, 4)
    ora = O.repa3_to_image(p, canvas.astype(np.float64))
    assert dev.shape == (W, H, 3)
    assert np.array_equal(dev, ora)
    want_mean = O.repa3_to_image(p, np.clip(canvas.astype(np.float64) * (1.0 / 3.0), 0, 1))
    assert np.array_equal(mean, want_mean)
    want_g = np.rint(255.0 * np.clip(canvas.astype(np.float64) * (1.0 / 3.0), 0, 1) ** (1 / 2.2))
    assert np.max(np.abs(gam.astype(int) - want_g.transpose(1, 0, 2).astype(int))) <= 1


@pytest.mark.parametrize("H,W", [(1, 1), (64, 64), (65, 63), (200, 3), (3, 200)])
def test_rgb8_ragged_tiles(pa, H, W):
    rng = np.random.default_rng(H * 1000 + W)
    canvas = rng.uniform(0, 1.2, size=(H, W, 3)).astype(np.float32)
    with pa.Renderer(pa.Params(H, W, 500.0, 1, 4)) as r:
        r.write_sum_f32(canvas)
        dev = r.read_rgb8()
    assert np.array_equal(dev, O.repa3_to_image(O.params(H, W, 500.0, 4), canvas.astype(np.float64)))


@pytest.mark.parametrize("flags", [0, 4])
def test_checkpoint_resume_continues_the_same_render(pa, flags):
    """(sums, samples done) is the whole resumable state: a render resumed in a
    NEW context continues with the same global sample indices, so under
    PT_FLAG_NO_STEAL it equals the uninterrupted render bit for bit, and up to
    fp32 summation order otherwise."""
    P = pa.Params(48, 72, 500.0, 12, 8)
    with pa.Renderer(P, seed=6, flags=flags) as whole:
        whole.trace(12)
        want, want_segs = whole.read_sum_f32(), whole.segments
    with pa.Renderer(P, seed=6, flags=flags) as first:
        first.trace(5)
        state = first.checkpoint()
        segs_a = first.segments
    assert state["samples_done"] == 5
    with pa.Renderer(P, seed=6, flags=flags) as second:
        second.restore(state)
        second.trace(7)
        got = second.read_sum_f32()
        assert second.samples_done == 12
        segs_b = second.segments
    if flags & 4:
        assert np.array_equal(got, want)
    else:
        assert np.allclose(got, want, rtol=4e-6, atol=1e-6)
    assert segs_a + segs_b == want_segs + 48 * 72          # the resumed launch recomputed the primary hits
    with pa.Renderer(pa.Params(48, 72, 500.0, 12, 7), seed=6) as other:
        with pytest.raises(ValueError):
            other.restore(state)


# -------------------------------------------------------------- errors -------
def test_error_behaviour(pa):
    with pytest.raises(pa.PtError) as e:
        pa.Renderer(pa.Params(0, 8, 500.0, 1, 4))
    assert e.value.status == -1
    with pytest.raises(pa.PtError) as e:
        pa.Renderer(pa.Params(8, 8, 500.0, 1, 65))
    assert e.value.status == -5
    with pytest.raises(pa.PtError):
        pa.Renderer(pa.Params(8, 8, 500.0, 1, 4), device=99)
    with pytest.raises(pa.PtError):
        pa.Renderer(pa.Params(8, 8, 500.0, 1, 4), sample_offset=2, sample_stride=2)
    with pa.Renderer(pa.Params(8, 8, 500.0, 1, 4)) as r:
        with pytest.raises(pa.PtError):
            r.trace(-1)
        r.upload([])                                       # the empty object list is a scene (src/Lib.hs:83; tested below)
        with pytest.raises(pa.PtError) as e:
            _ = pa._lib.check(pa._lib.lib.pt_scene_upload(r._ctx, None, 3), r._ctx)     # null objects with n > 0
        assert e.value.status == pa._lib.PT_ERR_INVALID


def test_sample_index_limit_and_the_last_usable_samples(pa):
    """The global sample index shares a Philox word with the 6-bit bounce: indices
    below 2^26.  The last usable samples must render exactly what the oracle does
    with the same indices (the kernels shift the index left by 6 in 32 bits), and
    one sample more must be refused with PT_ERR_LIMIT."""
    H, W, B = 24, 40, 6
    top = (1 << 26) - 1                                    # first index beyond a launch must stay below 2^26
    zero = np.zeros((H, W, 3), np.float32)
    for flags in (0, 4):
        with pa.Renderer(pa.Params(H, W, 500.0, 1, B), seed=2, flags=flags) as r:
            r.write_sum_f32(zero, samples_done=top - 5)
            r.trace(5)
            dev = r.read_sum()
            with pytest.raises(pa.PtError) as e:
                r.trace(1)
            assert e.value.status == -5
        px = np.arange(H * W, dtype=np.int32)
        ora, _ = O.trace_pixels(O.params(H, W, 500.0, B, O.PHILOX, 2), O.default_scene(), px, 5, s_begin=top - 5)
        assert _mismatch_fraction(dev, ora.reshape(H, W, 3)) <= 0.01
        assert ora.sum() > 0


# ----------------------------------------------------- committed fixtures ----
def test_against_committed_golden_fixtures(pa):
    """tests/golden/*.npz (written by make_golden.py from the oracle, cross-checked
    analytically in test_oracle_golden.py): primary hits bit-exact, Philox-stream
    image path-wise equal."""
    import os
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for name in ("primary_default_64x64.npz", "primary_default_200x200.npz"):
        f = np.load(os.path.join(G, name))
        with pa.Renderer(pa.Params(int(f["H"]), int(f["W"]), float(f["camera_depth"]), 1, 4)) as r:
            idx, t, nrm = r.primary_hits()
        assert np.array_equal(idx, f["idx"]) and np.array_equal(t, f["t"]) and np.array_equal(nrm, f["normal"])
    f = np.load(os.path.join(G, "image_philox_64x64_8spp_d8_seed11.npz"))
    with pa.Renderer(pa.Params(64, 64, 500.0, 8, 8), seed=11) as r:
        r.trace(8)
        dev = r.read_sum()
    assert _mismatch_fraction(dev, f["sum"]) <= 0.01
    # BASELINE config 1 (200x200, 16 spp, depth 4) against the oracle on the reference's StdGen
    # stream: independent estimates, so only image-wide statistics are comparable at 16 spp
    f = np.load(os.path.join(G, "image_config1_stdgen_200x200_16spp_d4.npz"))
    with pa.Renderer(pa.Params(200, 200, 500.0, 16, 4), seed=0) as r:
        r.trace(16)
        dev = r.read_sum()
    for k in range(3):
        assert abs(dev[..., k].mean() - f["sum"][..., k].mean()) <= 0.01 * f["sum"][..., k].mean()


def test_reset_is_lazy_but_exact(pa):
    """pt_reset defers the zeroing of the canvas to the next pt_trace (which starts
    from zero without reading it) or the next read: results must equal a fresh
    context's, segment counts restart, and the raw device pointer reads zero after
    a reset + read."""
    P = pa.Params(40, 52, 500.0, 5, 6)
    with pa.Renderer(P, seed=8) as a, pa.Renderer(P, seed=8) as fresh:
        a.trace(7)
        a.reset()
        a.trace(5)
        fresh.trace(5)
        assert np.array_equal(a.read_sum_f32(), fresh.read_sum_f32())
        # (the primary hits of this camera and scene are kept across pt_reset: H*W searches are not repeated)
        assert a.segments == fresh.segments - 40 * 52 and a.samples_done == 5
        for _ in range(3):                                 # repeated resets, nothing in between
            a.reset()
        assert not a.read_sum_f32().any() and a.segments == 0
        a.reset()
        assert a.read_rgb8().max() == 0


@pytest.mark.parametrize("steal_flag", [0, 4])
@pytest.mark.parametrize("n,seed,bounces", [(17, 21, 6), (40, 7, 6), (300, 8, 5), (1025, 10, 4), (2500, 11, 6)])
def test_spatial_index_finds_the_linear_scans_object(pa, n, seed, bounces, steal_flag):
    """The BVH only prunes (pt_bvh.hpp): with and without it the same object is
    struck on every segment, so executed searches and fp64 primary hits are equal,
    and canvases are equal bit for bit in the strict sample order
    (PT_FLAG_NO_STEAL) and up to the fp32 summation order of a pixel's samples
    otherwise -- the indexed kernel hands samples to idle lanes at other moments
    than the scan does (closestObjectTo, src/Lib.hs:81-98, is the scan)."""
    objs = pa.random_disk_scene(n, seed)
    H, W, spp = 72, 56, 5
    out = []
    for flags in (steal_flag, steal_flag | pa.PT_FLAG_NO_BVH):
        with pa.Renderer(pa.Params(H, W, 60.0, spp, bounces), objs, seed=3, flags=flags) as r:
            r.trace(2)
            r.trace(spp - 2)
            out.append((r.read_sum_f32(), r.segments, r.primary_hits()))
    (a, sa, pa_), (b, sb, pb) = out
    assert sa == sb
    assert np.array_equal(a, b) if steal_flag else np.allclose(a, b, rtol=2e-6, atol=1e-6)
    for x, y in zip(pa_, pb):
        assert np.array_equal(x, y)
    if n >= 300:                                           # 2 % of the objects are lights
        assert a.sum() > 0


def test_spatial_index_with_coincident_and_degenerate_disks(pa):
    """Exact duplicates (ties go to the lower index), zero and negative radii,
    a zero normal: the index must not change what the scan finds."""
    base = pa.random_disk_scene(60, 31)
    objs = list(base)
    d = base[5]
    objs += [pa.Object(d.color, 0.0, False, pa.Disk(d.form.center, d.form.normal, d.form.radius), "dup of 5")] * 3
    objs += [pa.Object((1, 1, 1), 0.0, False, pa.Disk((0.0, 0.0, 300.0), (0.0, 0.0, -1.0), 0.0), "r = 0"),
             pa.Object((1, 1, 1), 0.0, False, pa.Disk((10.0, 0.0, 300.0), (0.0, 0.0, -1.0), -5.0), "r < 0"),
             pa.Object((1, 1, 1), 0.0, False, pa.Disk((20.0, 0.0, 300.0), (0.0, 0.0, 0.0), 30.0), "n = 0")]
    H, W, spp = 64, 64, 4
    res = []
    for flags in (0, pa.PT_FLAG_NO_BVH, pa.PT_FLAG_NO_STEAL, pa.PT_FLAG_NO_STEAL | pa.PT_FLAG_NO_BVH):
        with pa.Renderer(pa.Params(H, W, 60.0, spp, 6), objs, seed=9, flags=flags) as r:
            r.trace(spp)
            res.append((r.read_sum_f32(), r.segments, r.primary_hits()[0]))
    assert res[0][1] == res[1][1] == res[2][1] == res[3][1]
    assert np.array_equal(res[2][0], res[3][0])             # strict sample order: bit for bit
    assert np.allclose(res[0][0], res[1][0], rtol=2e-6, atol=1e-6) and np.allclose(res[0][0], res[2][0], rtol=2e-6, atol=1e-6)
    assert np.array_equal(res[0][2], res[1][2]) and np.array_equal(res[2][2], res[3][2])
    assert not np.isin(res[0][2], [60, 61, 62]).any()       # a duplicate never wins over object 5


def test_spatial_index_with_32_bit_stacks(pa, monkeypatch):
    """The walk's per-lane stack holds 16-bit node numbers while the tree has fewer than 65,536 nodes (LDS per wave
    is what it saves); PT_BVH_STACK32 selects the 32-bit form that larger trees take.  Same walk, same result."""
    objs = pa.random_disk_scene(2000, 12)
    P = pa.Params(80, 96, 500.0, 5, 6)
    def render(flags):
        with pa.Renderer(P, objs, seed=4, flags=flags) as r:
            r.trace(5)
            return r.read_sum_f32(), r.segments, r.bvh_counters()
    monkeypatch.delenv("PT_BVH_STACK32", raising=False)
    a, sa, ca = render(0)
    monkeypatch.setenv("PT_BVH_STACK32", "1")
    b, sb, cb = render(0)
    assert sa == sb and ca == cb and ca[0] > 0
    assert np.allclose(a, b, rtol=4e-6, atol=1e-6)          # (the order lanes finish in may differ: fp32 summation order)
    scan, ss, _ = render(pa.PT_FLAG_NO_BVH)
    assert ss == sa and np.allclose(scan, b, rtol=4e-6, atol=1e-6)


# ------------------------------------------------- empty and maximum scenes ----
@pytest.mark.parametrize("flags_name", ["0", "PT_FLAG_NO_STEAL", "PT_FLAG_NO_PRIMARY_REUSE", "PT_FLAG_LDS_TILES"])
def test_empty_scene_is_black(pa, flags_name):
    """`objects = []`: closestObjectTo's guard (src/Lib.hs:83) answers Nothing for every ray, interactWith Nothing is
    black (src/Lib.hs:72).  One search per pixel runs and finds nothing; the oracle says the same."""
    flags = 0 if flags_name == "0" else getattr(pa, flags_name)
    H, W = 70, 130
    with pa.Renderer(pa.Params(H, W, 500.0, 4, 8), [], seed=1, flags=flags) as r:
        r.trace(3)
        r.trace(1)
        assert r.samples_done == 4
        assert not r.read_sum_f32().any()
        # executed searches: the primary one -- once per pixel (kept for the scene), once per pixel and launch in
        # the strict-order kernels, once per pixel and sample without the reuse
        per_pixel = {"0": 1, "PT_FLAG_NO_STEAL": 2, "PT_FLAG_LDS_TILES": 1, "PT_FLAG_NO_PRIMARY_REUSE": 4}[flags_name]
        assert r.segments == H * W * per_pixel
        assert not r.read_rgb8(pa.PT_RGB8_REFERENCE).any()
        idx, t, nrm = r.primary_hits()
        if flags == 0:
            r.specialise()                                  # an empty scene has a scene kernel too (all padding)
            r.reset()
            r.trace(2)
            assert not r.read_sum_f32().any() and r.segments == 0       # (the kept primary hits: no search runs)
    empty = (O.OObject * 0)()
    p = O.params(H, W, 500.0, 8, O.PHILOX, 1)
    oidx, ot, _ = O.primary_hits(p, empty)
    assert np.array_equal(idx, oidx) and (idx == -1).all()
    assert np.array_equal(t, ot)
    ora = O.Tracer(p, empty, threads=2)
    assert not ora.trace(4).any() and ora.segments.value == H * W * 4     # the oracle searches for every sample (no reuse)


def test_scene_changes_to_and_from_the_empty_scene(pa):
    H, W = 64, 64
    with pa.Renderer(pa.Params(H, W, 500.0, 2, 8), seed=2) as r:          # the default scene
        r.trace(2)
        lit = r.read_sum_f32()
        assert lit.any()
        r.upload([])
        r.reset()
        r.trace(2)
        assert not r.read_sum_f32().any() and r.segments == H * W
        r.upload(pa.objects)
        r.reset()
        r.trace(2)
        assert np.array_equal(r.read_sum_f32(), lit)


def test_limit_of_65536_objects(pa):
    """MAX_OBJECTS (include/pt.h): 65,536 objects upload, index and render -- checked against the oracle's scan on
    its Philox stream at a small frame -- and one more is refused."""
    n = 65536
    objs = pa.synthetic_scene(n, 5, "disks")
    H, W, spp, B = 48, 64, 2, 6
    with pa.Renderer(pa.Params(H, W, 500.0, spp, B), objs, seed=3) as r:
        r.trace(spp)
        dev = r.read_sum()
        segs = r.segments
        idx, t, _ = r.primary_hits()
        nodes, _ = r.bvh_counters()
        assert nodes > 0                                    # it went through the index
        with pytest.raises(pa.PtError) as e:
            r.upload(list(objs) + [objs[0]])
        assert e.value.status == pa.PT_ERR_LIMIT
        r.trace(1)                                          # the refused upload left the scene in place
    p = O.params(H, W, 500.0, B, O.PHILOX, 3)
    sc = O.scene_from(objs)
    oidx, ot, _ = O.primary_hits(p, sc)
    assert np.array_equal(idx, oidx) and np.array_equal(t, ot)
    tr = O.Tracer(p, sc, threads=8)
    ora = tr.trace(spp)
    assert tr.segments.value - H * W * spp == segs - H * W   # the oracle also counts each sample's primary search
    bad = (np.abs(dev - ora) > 2e-4 * np.abs(ora) + 1e-5).any(axis=2).mean()
    assert bad <= 0.01


def test_box_test_of_the_spatial_index_never_misses(pa, small):
    """The index only prunes, so its box test (centre / half-extent nodes in binary16, nine v_fma_mix_f32: slab_hit,
    DESIGN.md 3.4) may say "reached" too often but never too seldom: wherever the exact segment origin + t*vector,
    0 <= t <= t_best, touches the UNPADDED box (fp64 slab test with exact handling of zero components), the device says
    1.  Random rays and boxes, rays with zero direction components (a pixel of the middle row or column), origins on a
    face, boxes of zero thickness, tiny and large scenes, origins twenty scene sizes away; the pad is the builder's 2e-5
    of the scene's extent."""
    rng = np.random.default_rng(11)
    n = 40000
    false_pos = []
    for scale, offset, far in ((100.0, 0.0, 0.0), (1.0, 0.0, 0.0), (1e-3, 0.0, 0.0), (500.0, 2000.0, 0.0), (3e4, 0.0, 0.0), (100.0, 0.0, 20.0)):
        c = offset + scale * (rng.random((n, 3)) - 0.5)
        h = scale * 0.05 * rng.random((n, 3)) * (rng.random((n, 3)) > 0.1)          # a tenth of the sides: zero thickness
        lo, hi = c - h, c + h
        extent = np.abs(np.concatenate([lo, hi])).max()
        pad = 2e-5 * extent + 1e-30
        org = (offset + scale * (rng.random((n, 3)) - 0.5)).astype(np.float32)
        if far:                                                                      # origins `far` scene sizes away (a hit on an infinite plane)
            u = rng.standard_normal((n, 3))
            org = (org + far * scale * u / np.linalg.norm(u, axis=1, keepdims=True)).astype(np.float32)
        vec = rng.standard_normal((n, 3))
        vec /= np.linalg.norm(vec, axis=1, keepdims=True)
        vec[rng.random((n, 3)) < 0.08] = 0.0                                         # axis-parallel rays
        vec[np.all(vec == 0.0, axis=1)] = (0.0, 0.0, 1.0)
        vec = vec.astype(np.float32)
        k = rng.random(n) < 0.15                                                     # origins exactly on a face
        org[k, 0] = lo[k, 0].astype(np.float32)
        aim = rng.random(n) < 0.5                                                    # half of the rays aimed at their box
        tgt = (c + h * (rng.random((n, 3)) - 0.5))[aim]
        dv = tgt - org[aim].astype(np.float64)
        vec[aim] = (dv / np.maximum(np.linalg.norm(dv, axis=1, keepdims=True), 1e-30)).astype(np.float32)
        t_best = np.where(rng.random(n) < 0.3, np.inf, scale * 2.0 * rng.random(n)).astype(np.float32)
        dev = small.dev_box_test(np.concatenate([lo, hi], axis=1), pad, org, vec, t_best)
        # exact: intersect [0, t_best] with the three slabs of the unpadded box, in fp64 on the fp32 ray
        o64, v64 = org.astype(np.float64), vec.astype(np.float64)
        tn = np.zeros(n); tf = t_best.astype(np.float64).copy()
        ok = np.ones(n, dtype=bool)
        for a in range(3):
            z = v64[:, a] == 0.0
            ok &= ~z | ((o64[:, a] >= lo[:, a]) & (o64[:, a] <= hi[:, a]))
            with np.errstate(divide="ignore", invalid="ignore"):
                t0 = (lo[:, a] - o64[:, a]) / v64[:, a]
                t1 = (hi[:, a] - o64[:, a]) / v64[:, a]
            tn = np.where(z, tn, np.maximum(tn, np.minimum(t0, t1)))
            tf = np.where(z, tf, np.minimum(tf, np.maximum(t0, t1)))
        exact = ok & (tn <= tf)
        missed = exact & (dev == 0)
        assert not missed.any(), (scale, offset, int(missed.sum()), np.flatnonzero(missed)[:5])
        assert exact.sum() > n // 8                                                   # the case is exercised
        false_pos.append(float(((dev == 1) & ~exact).sum()) / n)
    assert max(false_pos) < 0.05, false_pos                                           # and the test still prunes


def test_indexed_kernel_sample_slices(pa):
    """A launch of >= 256 samples on an indexed scene runs as two sample slices (the last 1/16 behind the rest, in a
    plane of its own that is added afterwards: DESIGN.md 3.4); it must trace exactly the samples that two unsliced
    launches of half the size trace -- the same segment count but for the second launch's primary pass, and the same
    canvas up to fp32 summation order -- and an additive third launch must continue the same sample sequence."""
    objs = pa.random_disk_scene(1025, 10)
    H, W = 96, 80
    with pa.Renderer(pa.Params(H, W, 500.0, 256, 6), objs, seed=5) as r:
        r.trace(256)
        a, sa = r.read_sum_f32(), r.segments
        r.trace(64)                                           # unsliced, on top
        a2, sa2 = r.read_sum_f32(), r.segments
    with pa.Renderer(pa.Params(H, W, 500.0, 256, 6), objs, seed=5) as r:
        r.trace(128); r.trace(128)
        b, sb = r.read_sum_f32(), r.segments
        r.trace(64)
        b2, sb2 = r.read_sum_f32(), r.segments
    assert sa + H * W == sb and sa2 + H * W == sb2
    assert np.allclose(a, b, rtol=1e-5, atol=1e-4) and np.allclose(a2, b2, rtol=1e-5, atol=1e-4)
    assert a.sum() > 0
