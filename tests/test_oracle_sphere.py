"""The Sphere extension of the oracle (O_SPHERE, oracle/pt_oracle.c).  The
reference has no sphere -- src/Object.hs:191-194 is "TODO: add spheres" -- so
there is NO reference parity to claim here: these are known answers from
elementary geometry, computed by code that shares nothing with the oracle, and
they pin the definition that the device follows."""
import numpy as np
import pytest

import oracle as O


def _sphere(c, r, color=(1, 1, 1), e=0.0, refl=False):
    import pathtracer_amd as pa
    return O.scene_from([pa.Object(color, e, refl, pa.Sphere(tuple(map(float, c)), float(r)), "s")])


def _dist(o, v, c, r):
    sc = _sphere(c, r)
    t = O.C.c_double()
    ok = O.L.o_distance_from(O.vec(o), O.vec(v), O.C.byref(sc[0]), O.C.byref(t))
    return (t.value if ok else None), sc


def test_axis_ray_known_answers():
    # from outside along +z: enters at z = 10 - 2
    t, sc = _dist((0, 0, 0), (0, 0, 1), (0, 0, 10), 2)
    assert t == 8.0
    n = O.L.o_normal_at(O.C.byref(sc[0]), O.L.o_march(O.vec((0, 0, 0)), O.vec((0, 0, 1)), 8.0)).t()
    assert n == (0.0, 0.0, -1.0)                                # outward, toward the ray origin
    # non-unit direction: t is a ray PARAMETER
    t, _ = _dist((0, 0, 0), (0, 0, 4), (0, 0, 10), 2)
    assert t == 2.0
    # origin inside: the far intersection, normal still outward (never flipped)
    t, sc = _dist((0, 0, 10), (0, 0, 1), (0, 0, 10), 2)
    assert t == 2.0
    assert O.L.o_normal_at(O.C.byref(sc[0]), O.vec((0, 0, 12))).t() == (0.0, 0.0, 1.0)
    # sphere behind the origin, ray missing sideways, non-positive radius: Nothing
    assert _dist((0, 0, 0), (0, 0, 1), (0, 0, -10), 2)[0] is None
    assert _dist((0, 0, 0), (0, 0, 1), (5, 0, 10), 2)[0] is None
    assert _dist((0, 0, 0), (0, 0, 1), (0, 0, 10), 0)[0] is None
    assert _dist((0, 0, 0), (0, 0, 1), (0, 0, 10), -3)[0] is None


def test_random_rays_against_the_quadratic():
    rng = np.random.default_rng(5)
    hits = 0
    for _ in range(6000):
        c = rng.uniform(-50, 50, 3); r = rng.uniform(0.5, 30)
        o = rng.uniform(-80, 80, 3); v = rng.normal(size=3) * rng.uniform(0.2, 5)
        t, sc = _dist(o, v, c, r)
        # textbook quadratic |o + t v - c|^2 = r^2 in extended precision
        oc = (o - c).astype(np.longdouble); vv = v.astype(np.longdouble)
        a, b, cc = vv @ vv, 2 * (oc @ vv), oc @ oc - np.longdouble(r) ** 2
        disc = b * b - 4 * a * cc
        want = None
        if disc >= 0:
            roots = sorted([(-b - np.sqrt(disc)) / (2 * a), (-b + np.sqrt(disc)) / (2 * a)])
            pos = [x for x in roots if x > 0]
            want = float(pos[0]) if pos else None
        if want is None or t is None:
            # disagreement only where the ray grazes the sphere (disc ~ 0) or starts on it
            assert (want is None and t is None) or abs(float(disc)) < 1e-6 * float(b * b + 1) or min(abs(float(x)) for x in roots) < 1e-9
            continue
        hits += 1
        assert t == pytest.approx(want, rel=1e-9, abs=1e-9)
        p = o + t * v
        assert np.linalg.norm(p - c) == pytest.approx(r, rel=1e-9)
        n = np.array(O.L.o_normal_at(O.C.byref(sc[0]), O.vec(p)).t())
        assert np.linalg.norm(n) == pytest.approx(1.0, rel=1e-9) and np.allclose(n * r, p - c, atol=1e-9)
    assert hits > 60


def test_sphere_scene_renders_and_light_sphere_emits():
    import pathtracer_amd as pa
    objs = [pa.Object((1, 1, 1), 0, False, pa.InfinitePlane((0, 0, 100), (0, 0, -1)), "wall"),
            pa.Object((1, 1, 1), 3.0, True, pa.Sphere((0.0, 0.0, -30.0), 25.0), "lamp behind the camera"),
            pa.Object((0.5, 0.5, 1), 0, False, pa.Sphere((0.0, 0.0, 60.0), 10.0), "ball")]
    p = O.params(32, 32, 60.0, 6, O.STDGEN)
    tr = O.Tracer(p, O.scene_from(objs), threads=2)
    img = tr.trace(64)
    idx, t, nrm = O.primary_hits(p, O.scene_from(objs))
    assert idx[16, 16] == 2 and t[16, 16] == 50.0 and tuple(nrm[16, 16]) == (0.0, 0.0, -1.0)
    assert set(np.unique(idx)) == {0, 2}
    assert img.sum() > 0 and np.all(np.isfinite(img))
    assert img.max() <= 3.0 * 64                                  # emittance * pixel <= 3 per sample
