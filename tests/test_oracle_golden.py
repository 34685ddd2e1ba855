"""The oracle against the committed fixtures (tests/golden/, written by
make_golden.py) and against known answers derived analytically from the
reference's scene and camera (src/Object.hs:104-148, src/Lib.hs:41-51) by code
that shares nothing with the oracle."""
import os

import numpy as np

import oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def analytic_primary(H, W, D):
    """Region map of SURVEY.md 8(c): t0 = 100/v_z (back wall), t1 = -20/v_y
    (y = -20 plane), t2 = 20/v_x (x = 20 plane), light at t = -1/v_z < 0."""
    i, j = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    v = np.stack([H / 2.0 - i, j - W / 2.0, np.full_like(i, D)], axis=-1)
    v /= np.linalg.norm(v, axis=-1, keepdims=True)
    with np.errstate(divide="ignore", invalid="ignore"):
        t = np.stack([100.0 / v[..., 2], -20.0 / v[..., 1], 20.0 / v[..., 0], -1.0 / v[..., 2]], axis=-1)
    t[~((t > 0) & np.isfinite(t))] = np.inf
    idx = np.argmin(t, axis=-1)                       # first minimum wins, as minimumBy
    best = np.min(t, axis=-1)
    srt = np.sort(t, axis=-1)
    tie = np.abs(srt[..., 1] - srt[..., 0]) <= 1e-9 * srt[..., 0]
    return idx, best, tie


def test_primary_hits_match_analytic_regions():
    for H, W, D in [(64, 64, 500.0), (200, 200, 500.0), (500, 500, 500.0), (301, 257, 500.0)]:
        idx, t, nrm = O.primary_hits(O.params(H, W, D, 4), O.default_scene())
        aidx, at, tie = analytic_primary(H, W, D)
        assert np.array_equal(idx[~tie], aidx[~tie])
        assert np.allclose(t[~tie], at[~tie], rtol=1e-12)
        normals = np.array([[0, 0, -1], [0, 1, 0], [-1, 0, 0], [0, 0, 1]], np.float64)
        assert np.array_equal(nrm, normals[idx])
    # tie lines of SURVEY.md 7 exist where predicted, for 500x500: column j = 150, row i = 150
    _, _, tie = analytic_primary(500, 500, 500.0)
    assert tie[:, 150].sum() > 100 and tie[150, :].sum() > 100


def test_centre_pixel_kat():
    """Centre pixel: direction (0,0,1), hits object 0 at t = 100, normal (0,0,-1);
    planes 1 and 2 give -20/0 = -Inf and are rejected, the light gives t = -1."""
    p = O.params(500, 500, 500.0, 30)
    sc = O.default_scene()
    d = O.L.o_primary_direction(O.C.byref(p), 250, 250).t()
    assert d == (0.0, 0.0, 1.0)
    t = O.C.c_double()
    assert O.L.o_closest_object_to(sc, 4, O.vec((0, 0, 0)), O.vec(d), -1, O.C.byref(t)) == 0 and t.value == 100.0
    for k in (1, 2, 3):
        assert O.L.o_distance_from(O.vec((0, 0, 0)), O.vec(d), O.C.byref(sc[k]), O.C.byref(t)) == 0
    O.L.o_distance_from_prime(O.vec((0, 0, 0)), O.vec(d), O.C.byref(sc[1]), O.C.byref(t))
    assert t.value == -np.inf


def test_max_bounces_one_is_black():
    tr = O.Tracer(O.params(32, 32, 500.0, 1, O.STDGEN), O.default_scene(), threads=2)
    assert not tr.trace(3).any() and tr.segments.value == 32 * 32 * 3


def test_radiance_values_are_products_of_scene_colours():
    """Every sample is 0 or emittance * product of plane colours (src/Lib.hs:74,79):
    channel values lie in {2 * 0.1^a} and never exceed 2."""
    tr = O.Tracer(O.params(40, 40, 500.0, 6, O.STDGEN), O.default_scene(), threads=2)
    s = tr.trace(1)
    vals = np.unique(np.round(s, 12))
    allowed = {0.0} | {round(2 * 0.1**a, 12) for a in range(6)}
    assert set(vals.tolist()) <= allowed


def test_fixtures_reproduce():
    for name in ("primary_default_64x64.npz", "primary_default_200x200.npz"):
        f = np.load(os.path.join(G, name))
        idx, t, nrm = O.primary_hits(O.params(int(f["H"]), int(f["W"]), float(f["camera_depth"]), 4),
                                     O.default_scene())
        assert np.array_equal(idx, f["idx"]) and np.array_equal(t, f["t"]) and np.array_equal(nrm, f["normal"])
    for name in ("image_config1_stdgen_200x200_16spp_d4.npz", "image_philox_64x64_8spp_d8_seed11.npz"):
        f = np.load(os.path.join(G, name))
        tr = O.Tracer(O.params(int(f["H"]), int(f["W"]), float(f["camera_depth"]), int(f["bounces"]),
                               int(f["rng"]), int(f["seed"])), O.default_scene(), threads=3)
        tr.trace(int(f["spp"]))
        assert np.array_equal(tr.sum, f["sum"]) and tr.segments.value == int(f["segments"])


def test_stdgen_state_is_carried_between_iterations():
    """traceCanvas threads gen' (src/Lib.hs:65): 2 x trace(3) == trace(6), and the
    two halves differ."""
    p = O.params(20, 20, 500.0, 5, O.STDGEN)
    a, b = O.Tracer(p, O.default_scene(), threads=1), O.Tracer(p, O.default_scene(), threads=2)
    a.trace(6)
    first = b.trace(3).copy()
    b.trace(3)
    assert np.array_equal(a.sum, b.sum)
    assert not np.array_equal(first * 2, b.sum)


def test_repa3_to_image_layout():
    """generateImage fromCoords H W with fromCoords i j = canvas!(i,j,k): byte
    (j*H + i)*3 + k, an image H wide and W tall."""
    H, W = 5, 3
    p = O.params(H, W, 500.0, 4)
    canvas = np.zeros((H, W, 3))
    canvas[4, 1, 2] = 1.0
    canvas[0, 2, 0] = 0.5
    img = O.repa3_to_image(p, canvas)
    assert img.shape == (W, H, 3)
    flat = img.reshape(-1)
    assert flat[(1 * H + 4) * 3 + 2] == 255 and flat[(2 * H + 0) * 3 + 0] == 128
    assert flat.sum() == 255 + 128


def test_crop_fixture_is_what_the_oracle_computes():
    """tests/golden/crops_stdgen_*.npz (the per-pixel statistical test's reference side, make_golden.py): a
    seeded subset of each window recomputed here equals the committed sums exactly -- the pixel-subset tracer
    threads each pixel's own StdGen (src/Lib.hs:34-39, :58-65), so a subset is bit-identical to the whole."""
    g = np.load(os.path.join(G, "crops_stdgen_1920x1080_d8_2048spp.npz"))
    H, W, spp = int(g["H"]), int(g["W"]), int(g["spp"])
    assert (H, W, spp, int(g["bounces"])) == (1920, 1080, 2048, 8)
    rng = np.random.default_rng(3)
    for key in ("centre", "corner", "tie"):
        i0, j0 = (int(x) for x in g[key + "_origin"])
        pick = rng.choice(128 * 128, 24, replace=False)
        px = O.crop_pixels(W, i0, j0)[pick]
        s, sq, _ = O.trace_pixels_stdgen(O.params(H, W, 500.0, 8, O.STDGEN), O.default_scene(), px, spp, threads=4)
        assert np.array_equal(s, g[key + "_sum"].reshape(-1, 3)[pick])
        assert np.array_equal(sq, g[key + "_sumsq"].reshape(-1, 3)[pick])
    # the window over the tie column really contains it (j = W/2 - 100)
    i0, j0 = (int(x) for x in g["tie_origin"])
    assert j0 <= W // 2 - 100 < j0 + 128


def test_pixel_subset_stdgen_equals_the_full_frame_tracer():
    H, W, B, spp = 40, 56, 6, 9
    p = O.params(H, W, 500.0, B, O.STDGEN)
    full = O.Tracer(p, O.default_scene(), threads=2)
    full.trace(spp)
    px = np.array([0, 5, W - 1, 17 * W + 3, H * W - 1], np.int32)
    s, sq, _ = O.trace_pixels_stdgen(p, O.default_scene(), px, spp, threads=2)
    assert np.array_equal(s, full.sum.reshape(-1, 3)[px])
    assert np.array_equal(sq, full.sumsq.reshape(-1, 3)[px])


def test_from_primary_aid_reproduces_the_plain_trace():
    """o_trace_pixels_from_primary (a test aid for the tie-line parity test) handed the oracle's OWN primary segment
    must give the plain trace bit for bit, with exactly one search per path fewer counted."""
    p = O.params(64, 64, 500.0, 6, O.PHILOX, 3)
    objs = O.default_scene()
    px = np.arange(0, 64 * 64, 37, dtype=np.int32)
    plain, segs = O.trace_pixels(p, objs, px, 8)
    idx, t, _ = O.primary_hits(p, objs)
    d = O.primary_directions(p, px)
    forced, segs_f = O.trace_pixels_from_primary(p, objs, px, idx.reshape(-1)[px], d * t.reshape(-1)[px][:, None], d, 8)
    assert np.array_equal(plain, forced)
    assert segs - segs_f == len(px) * 8
