"""Scene and parameter types with the reference's names.

`Params` mirrors src/Params.hs:3-11, `Object`/`Disk`/`InfinitePlane` mirror
src/Object.hs:30-56, and `objects` is the reference's hard-coded scene
(src/Object.hs:104-148) -- here they are run-time values handed to libpt.
"""
from dataclasses import dataclass, field
from typing import Sequence, Tuple, Union

from . import _lib

Vec = Tuple[float, float, float]


@dataclass
class Params:
    height: int = 500           # src/Params.hs:3
    width: int = 500            # src/Params.hs:5
    cameraDepth: float = 500.0  # src/Params.hs:7
    numIters: int = 10          # src/Params.hs:9
    maxBounces: int = 30        # src/Params.hs:11


@dataclass
class InfinitePlane:            # src/Object.hs:55-56
    point: Vec
    normal: Vec


@dataclass
class Disk:                     # src/Object.hs:52-54
    center: Vec
    normal: Vec
    radius: float


@dataclass
class Sphere:                   # EXTENSION: the reference only has "TODO: add spheres" (src/Object.hs:191-194)
    center: Vec
    radius: float
    normal: Vec = (0.0, 0.0, 0.0)       # unused: the normal is (point - center) / radius at the hit


@dataclass
class Rectangle:                # EXTENSION: commented out in the reference (src/Object.hs:57-61, :172-175);
    center: Vec                 # include/pt.h (PT_RECTANGLE) states how the sides are oriented
    normal: Vec
    height: float
    width: float


Form = Union[InfinitePlane, Disk, Sphere, Rectangle]


@dataclass
class Object:                   # src/Object.hs:30-36
    color: Vec
    emittance: float
    reflective: bool
    form: Form
    name: str = ""

    def to_c(self) -> _lib.pt_object:
        o = _lib.pt_object()
        f = self.form
        if isinstance(f, Disk):
            o.kind, pt, o.radius = _lib.PT_DISK, f.center, float(f.radius)
        elif isinstance(f, Sphere):
            o.kind, pt, o.radius = _lib.PT_SPHERE, f.center, float(f.radius)
        elif isinstance(f, Rectangle):
            o.kind, pt, o.radius = _lib.PT_RECTANGLE, f.center, 0.0
            o.extent[:] = [float(f.height), float(f.width)]
        elif isinstance(f, InfinitePlane):
            o.kind, pt, o.radius = _lib.PT_INFINITE_PLANE, f.point, 0.0
        else:
            raise TypeError(f"unknown Form {type(f).__name__}")
        o.reflective = 1 if self.reflective else 0
        o.point[:] = [float(x) for x in pt]
        o.normal[:] = [float(x) for x in f.normal]
        o.color[:] = [float(x) for x in self.color]
        o.emittance = float(self.emittance)
        return o


def to_c_array(objs: Sequence[Object]):
    arr = (_lib.pt_object * len(objs))()
    for k, o in enumerate(objs):
        arr[k] = o.to_c()
    return arr


# src/Object.hs:82-145, same names
infLight = Object((1, 1, 1), 5, True, InfinitePlane((0, 0, -10), (0, 0, -1)), "infinite light")
disk = Object((0, 1, 0), 0, False, Disk((0, 0, 10), (1, 0, -1), 600), "disk")
light = Object((1, 1, 1), 2, True, Disk((0, 0, -1), (0, 0, 1), 20), "light")
infPlane = Object((1, 1, 1), 0, False, InfinitePlane((20, -20, 100), (0, 0, -1)), "plane 1")
infPlane2 = Object((0.1, 1, 1), 0, False, InfinitePlane((20, -20, 100), (0, 1, 0)), "plane 2")
infPlane3 = Object((1, 1, 0.1), 0, False, InfinitePlane((20, -20, 100), (-1, 0, 0)), "plane 3")

# objects = V.fromList [infPlane, infPlane2, infPlane3, light]  (src/Object.hs:147-148)
objects = [infPlane, infPlane2, infPlane3, light]


def from_c(o) -> Object:
    """A pt_object back as an Object (the inverse of Object.to_c)."""
    pt, nrm = tuple(o.point), tuple(o.normal)
    if o.kind == _lib.PT_DISK:
        form = Disk(pt, nrm, o.radius)
    elif o.kind == _lib.PT_SPHERE:
        form = Sphere(pt, o.radius)
    elif o.kind == _lib.PT_RECTANGLE:
        form = Rectangle(pt, nrm, o.extent[0], o.extent[1])
    else:
        form = InfinitePlane(pt, nrm)
    return Object(tuple(o.color), o.emittance, bool(o.reflective), form)


def synthetic_scene(n: int, seed: int, kind: str = "disks"):
    """The synthetic n-object scene of BASELINE.json configs[3]/[4],
    built by libpt's own generator (pt_synthetic_scene: SplitMix64) so that the
    C++ and Haskell hosts get the same scene from (n, seed).  kind: "disks" (the
    variant the reference can express), "spheres" or "rectangles" (extensions)."""
    code = {"disks": _lib.PT_DISK, "spheres": _lib.PT_SPHERE, "rectangles": _lib.PT_RECTANGLE}[kind]
    arr = (_lib.pt_object * n)()
    _lib.check(_lib.lib.pt_synthetic_scene(arr, n, seed, code))
    objs = [from_c(o) for o in arr]
    for k, o in enumerate(objs):
        o.name = "back plane" if k == 0 else f"{kind[:-1]} {k}"
    return objs


def random_sphere_scene(n: int, seed: int):
    """BASELINE.json configs[3] as written ("random-sphere scene"): the generator
    of random_disk_scene with every disk replaced by a sphere of the same centre
    and radius.  Spheres are an extension with no reference counterpart
    (SURVEY.md F4); include/pt.h (PT_SPHERE) states their definition."""
    objs = random_disk_scene(n, seed)
    return [o if isinstance(o.form, InfinitePlane) else
            Object(o.color, o.emittance, o.reflective, Sphere(o.form.center, o.form.radius), o.name.replace("disk", "sphere"))
            for o in objs]


def random_rectangle_scene(n: int, seed: int):
    """random_disk_scene with every disk of radius r replaced by a Rectangle of the
    same centre and normal, height 2r and width 1.2r.  Rectangles are an extension
    (commented out in the reference, src/Object.hs:57-61); include/pt.h
    (PT_RECTANGLE) states their definition."""
    objs = random_disk_scene(n, seed)
    return [o if isinstance(o.form, InfinitePlane) else
            Object(o.color, o.emittance, o.reflective,
                   Rectangle(o.form.center, o.form.normal, 2.0 * o.form.radius, 1.2 * o.form.radius),
                   o.name.replace("disk", "rectangle"))
            for o in objs]


def random_disk_scene(n: int, seed: int):
    """Random test scenes of the shape of SURVEY.md 8(d)'s (numpy's generator; the
    parity tests' scenes -- the BENCHED synthetic scenes come from synthetic_scene,
    i.e. libpt's SplitMix64 generator): object 0 is a back plane at z=1500, the rest are
    disks with random unit normals in x,y in [-400,400], z in [200,1200];
    radii U(5,40); colour U(0.2,1)^3; half reflective; 2% emissive U(1,5)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    objs = [Object((1, 1, 1), 0, False, InfinitePlane((0, 0, 1500), (0, 0, -1)), "back plane")]
    for k in range(1, n):
        c = (rng.uniform(-400, 400), rng.uniform(-400, 400), rng.uniform(200, 1200))
        nrm = rng.normal(size=3)
        nrm = tuple(float(x) for x in nrm / np.linalg.norm(nrm))
        emit = float(rng.uniform(1, 5)) if rng.uniform() < 0.02 else 0.0
        objs.append(Object(tuple(float(x) for x in rng.uniform(0.2, 1, size=3)), emit,
                           bool(rng.uniform() < 0.5), Disk(tuple(float(x) for x in c), nrm,
                                                            float(rng.uniform(5, 40))), f"disk {k}"))
    return objs
