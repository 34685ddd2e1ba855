"""pt_synthetic_scene (include/pt.h): the generator of BASELINE.json's synthetic
scenes behind the C ABI (SplitMix64).  Needs no GPU.  Checked
against SplitMix64's published first outputs and against an independent
restatement of the draw order in Python integers."""
import ctypes as C
import math

import numpy as np

MASK = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & MASK

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & MASK
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK
        return z ^ (z >> 31)

    def unit(self):
        return (self.next() >> 11) * (1.0 / 9007199254740992.0)

    def uniform(self, lo, hi):
        return lo + (hi - lo) * self.unit()


def test_splitmix64_known_answers():
    g = SplitMix64(0)                                    # the stream every xoshiro reference seeds from
    assert [g.next() for _ in range(3)] == [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4, 0x06C45D188009454F]


def _expected(n, seed):
    g = SplitMix64(seed)
    out = []
    for _ in range(1, n):
        c = (g.uniform(-400.0, 400.0), g.uniform(-400.0, 400.0), g.uniform(200.0, 1200.0))
        while True:
            x, y, z = g.uniform(-1.0, 1.0), g.uniform(-1.0, 1.0), g.uniform(-1.0, 1.0)
            r2 = x * x + y * y + z * z
            if 1e-4 <= r2 <= 1.0:
                break
        inv = 1.0 / math.sqrt(r2)
        radius = g.uniform(5.0, 40.0)
        col = tuple(g.uniform(0.2, 1.0) for _ in range(3))
        refl = g.unit() < 0.5
        emissive = g.unit() < 0.02
        e = g.uniform(1.0, 5.0)
        out.append((c, (x * inv, y * inv, z * inv), radius, col, refl, e if emissive else 0.0))
    return out


def test_generator_follows_the_stated_draw_order():
    import pathtracer_amd as pa
    from pathtracer_amd import _lib
    n, seed = 200, 1
    objs = pa.synthetic_scene(n, seed, "disks")
    assert len(objs) == n
    back = objs[0]
    assert isinstance(back.form, pa.InfinitePlane) and back.form.point == (0.0, 0.0, 1500.0) and back.form.normal == (0.0, 0.0, -1.0)
    for o, (c, nrm, radius, col, refl, e) in zip(objs[1:], _expected(n, seed)):
        assert isinstance(o.form, pa.Disk)
        assert o.form.center == c and o.form.radius == radius and o.color == col
        assert o.reflective == refl and o.emittance == e
        assert np.allclose(o.form.normal, nrm, rtol=0, atol=1e-15)       # sqrt / divide: last-bit freedom only
        assert abs(np.linalg.norm(o.form.normal) - 1.0) < 1e-12
    # the other forms share centres, radii and materials
    sph = pa.synthetic_scene(n, seed, "spheres")
    rec = pa.synthetic_scene(n, seed, "rectangles")
    for d, s, r in zip(objs[1:], sph[1:], rec[1:]):
        assert isinstance(s.form, pa.Sphere) and s.form.center == d.form.center and s.form.radius == d.form.radius
        assert isinstance(r.form, pa.Rectangle) and r.form.height == 2.0 * d.form.radius and r.form.width == 1.2 * d.form.radius
        assert s.color == d.color == r.color
    # argument checks
    arr = (_lib.pt_object * 2)()
    assert _lib.lib.pt_synthetic_scene(arr, 0, 1, _lib.PT_DISK) == _lib.PT_ERR_INVALID
    assert _lib.lib.pt_synthetic_scene(arr, 2, 1, _lib.PT_INFINITE_PLANE) == _lib.PT_ERR_INVALID
    assert _lib.lib.pt_synthetic_scene(None, 2, 1, _lib.PT_DISK) == _lib.PT_ERR_INVALID


def test_population_statistics_of_configs3():
    import pathtracer_amd as pa
    objs = pa.synthetic_scene(1000, 1, "disks")[1:]
    assert 0.44 < np.mean([o.reflective for o in objs]) < 0.56
    assert 5 <= sum(o.emittance > 0 for o in objs) <= 40
    c = np.array([o.form.center for o in objs])
    assert c[:, :2].min() >= -400 and c[:, :2].max() <= 400 and c[:, 2].min() >= 200 and c[:, 2].max() <= 1200
    assert abs(np.mean([o.form.normal for o in objs], axis=0)).max() < 0.08
