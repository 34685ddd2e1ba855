"""ctypes binding of oracle/libpt_oracle.so -- the CPU fp64 RESTATEMENT of the
reference (test infrastructure; never imported by the product package)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")
SO = os.path.join(ODIR, "libpt_oracle.so")

STDGEN, PHILOX = 0, 1


class Vec3(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("z", C.c_double)]

    def t(self):
        return (self.x, self.y, self.z)


class OObject(C.Structure):
    _fields_ = [("kind", C.c_int32), ("reflective", C.c_int32), ("point", C.c_double * 3),
                ("normal", C.c_double * 3), ("radius", C.c_double), ("color", C.c_double * 3),
                ("emittance", C.c_double), ("extent", C.c_double * 2)]


class OParams(C.Structure):
    _fields_ = [("height", C.c_int32), ("width", C.c_int32), ("camera_depth", C.c_double),
                ("max_bounces", C.c_int32), ("rng", C.c_int32), ("seed", C.c_uint32)]


class StdGen(C.Structure):
    _fields_ = [("s1", C.c_int32), ("s2", C.c_int32)]


def _load():
    if not os.path.exists(SO):
        subprocess.check_call(["make", "-C", ODIR], stdout=subprocess.DEVNULL)
    L = C.CDLL(SO)
    d, v = C.c_double, Vec3
    sig = {
        "o_dot": (d, [v, v]), "o_cross": (v, [v, v]), "o_norm2": (d, [v]), "o_normalize": (v, [v]),
        "o_arctan2": (d, [d, d]), "o_to_spherical": (None, [v, C.POINTER(d), C.POINTER(d)]),
        "o_from_spherical": (v, [d, d]), "o_rotate_abs": (v, [v, d, d]), "o_rotate_rel": (v, [d, d, v]),
        "o_default_scene": (C.c_int, [C.POINTER(OObject)]),
        "o_distance_from_prime": (C.c_int, [v, v, C.POINTER(OObject), C.POINTER(d)]),
        "o_distance_from": (C.c_int, [v, v, C.POINTER(OObject), C.POINTER(d)]),
        "o_march": (v, [v, v, d]),
        "o_normal_at": (v, [C.POINTER(OObject), v]),
        "o_rectangle_axes": (None, [C.POINTER(OObject), C.POINTER(v), C.POINTER(v)]),
        "o_mk_stdgen": (StdGen, [C.c_int64]), "o_stdgen_next": (C.c_int32, [C.POINTER(StdGen)]),
        "o_stdgen_random_int": (C.c_int64, [C.POINTER(StdGen)]),
        "o_stdgen_random_double": (d, [C.POINTER(StdGen)]),
        "o_stdgen_randomR": (d, [C.POINTER(StdGen), d, d]),
        "o_philox2x32_10": (None, [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]),
        "o_specular_u": (v, [d, d, d, v, v]), "o_diffuse_u": (v, [d, d, v]),
        "o_primary_direction": (v, [C.POINTER(OParams), C.c_int, C.c_int]),
        "o_closest_object_to": (C.c_int, [C.POINTER(OObject), C.c_int, v, v, C.c_int, C.POINTER(d)]),
        "o_primary_hits": (None, [C.POINTER(OParams), C.POINTER(OObject), C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_void_p]),
        "o_starting_gens": (None, [C.POINTER(OParams), C.c_void_p]),
        "o_trace": (None, [C.POINTER(OParams), C.POINTER(OObject), C.c_int, C.c_void_p, C.c_int64, C.c_int64,
                           C.c_int64, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_int]),
        "o_trace_pixels": (None, [C.POINTER(OParams), C.POINTER(OObject), C.c_int, C.c_void_p, C.c_int64, C.c_void_p,
                                  C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64),
                                  C.c_int]),
        "o_trace_pixels_from_primary": (None, [C.POINTER(OParams), C.POINTER(OObject), C.c_int, C.c_void_p, C.c_int64,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                               C.c_void_p, C.POINTER(C.c_uint64), C.c_int]),
        "o_to_word8": (C.c_uint8, [d]),
        "o_repa3_to_image": (None, [C.POINTER(OParams), C.c_void_p, C.c_void_p]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    return L


L = _load()


def vec(a):
    return Vec3(float(a[0]), float(a[1]), float(a[2]))


def default_scene():
    arr = (OObject * 4)()
    L.o_default_scene(arr)
    return arr


def scene_from(objs):
    """Build the oracle's object array from pathtracer_amd.scene.Object values
    (same byte layout as pt_object)."""
    arr = (OObject * len(objs))()
    for k, o in enumerate(objs):
        c = o.to_c()
        C.memmove(C.byref(arr[k]), C.byref(c), C.sizeof(OObject))
    return arr


def params(H, W, depth, bounces, rng=PHILOX, seed=0):
    return OParams(H, W, float(depth), bounces, rng, seed)


def primary_hits(p, objs):
    n = len(objs)
    idx = np.empty((p.height, p.width), np.int32)
    t = np.empty((p.height, p.width), np.float64)
    nrm = np.empty((p.height, p.width, 3), np.float64)
    L.o_primary_hits(C.byref(p), objs, n, idx.ctypes.data, t.ctypes.data, nrm.ctypes.data)
    return idx, t, nrm


class Tracer:
    """Stateful oracle render: carries the per-pixel StdGen like traceCanvas does."""

    def __init__(self, p, objs, threads=8):
        self.p, self.objs, self.n, self.threads = p, objs, len(objs), threads
        self.sum = np.zeros((p.height, p.width, 3), np.float64)
        self.sumsq = np.zeros((p.height, p.width, 3), np.float64)
        self.segments = C.c_uint64(0)
        self.done = 0
        self.gens = None
        if p.rng == STDGEN:
            self.gens = (StdGen * (p.height * p.width))()
            L.o_starting_gens(C.byref(p), self.gens)

    def trace(self, iters, s_begin=None, stride=1):
        sb = self.done if s_begin is None else s_begin
        L.o_trace(C.byref(self.p), self.objs, self.n, self.gens, sb, iters, stride, self.sum.ctypes.data,
                  self.sumsq.ctypes.data, C.byref(self.segments), self.threads)
        self.done += iters
        return self.sum


def trace_pixels(p, objs, pixels, iters, s_begin=0, stride=1, threads=8):
    """Philox-stream sums of the listed pixels (i*W + j) only: (len(pixels), 3)."""
    assert p.rng == PHILOX
    pixels = np.ascontiguousarray(pixels, np.int32)
    out = np.zeros((len(pixels), 3), np.float64)
    segs = C.c_uint64(0)
    L.o_trace_pixels(C.byref(p), objs, len(objs), pixels.ctypes.data, len(pixels), None, s_begin, iters, stride,
                     out.ctypes.data, None, C.byref(segs), threads)
    return out, segs.value


def trace_pixels_from_primary(p, objs, pixels, idx, point, direction, iters, s_begin=0, stride=1, threads=8):
    """Test aid (no reference semantic): trace_pixels with each listed pixel's primary segment supplied -- struck
    object index (or -1), hit point and primary direction -- e.g. the device's own fp32-rounded numbers, so that both
    sides start the first bounce from the same point (the analytic tie lines).  -> (sums, segments from the first
    bounce on)."""
    assert p.rng == PHILOX
    pixels = np.ascontiguousarray(pixels, np.int32)
    idx = np.ascontiguousarray(idx, np.int32)
    point = np.ascontiguousarray(point, np.float64).reshape(-1, 3)
    direction = np.ascontiguousarray(direction, np.float64).reshape(-1, 3)
    assert len(idx) == len(pixels) == len(point) == len(direction)
    out = np.zeros((len(pixels), 3), np.float64)
    segs = C.c_uint64(0)
    L.o_trace_pixels_from_primary(C.byref(p), objs, len(objs), pixels.ctypes.data, len(pixels), idx.ctypes.data,
                                  point.ctypes.data, direction.ctypes.data, s_begin, iters, stride, out.ctypes.data,
                                  C.byref(segs), threads)
    return out, segs.value


def primary_directions(p, pixels):
    """rayFromCamToPixel's direction (src/Lib.hs:41-51) of the listed pixels (i*W + j): (len, 3) float64."""
    out = np.zeros((len(pixels), 3), np.float64)
    for q, px in enumerate(pixels):
        v = L.o_primary_direction(C.byref(p), int(px) // p.width, int(px) % p.width)
        out[q] = (v.x, v.y, v.z)
    return out


def trace_pixels_stdgen(p, objs, pixels, iters, threads=8):
    """The reference's own StdGen stream for the listed pixels (i*W + j) only: each keeps the generator that
    startingGens gives it (src/Lib.hs:34-39: `randoms (mkStdGen 0)` in row-major pixel order) and threads it
    through `iters` applications of traceCanvas (src/Lib.hs:58-65), so the sums equal the full-frame Tracer's at
    those pixels.  -> (sum, sumsq, segments), (len(pixels), 3) each."""
    assert p.rng == STDGEN
    pixels = np.ascontiguousarray(pixels, np.int32)
    every = (StdGen * (p.height * p.width))()
    L.o_starting_gens(C.byref(p), every)
    gens = (StdGen * len(pixels))()
    for q, px in enumerate(pixels):
        gens[q] = every[int(px)]
    out = np.zeros((len(pixels), 3), np.float64)
    sq = np.zeros((len(pixels), 3), np.float64)
    segs = C.c_uint64(0)
    L.o_trace_pixels(C.byref(p), objs, len(objs), pixels.ctypes.data, len(pixels), gens, 0, iters, 1,
                     out.ctypes.data, sq.ctypes.data, C.byref(segs), threads)
    return out, sq, segs.value


def crop_pixels(W, i0, j0, size=128):
    """Pixel indices i*W + j of the size x size window whose top-left pixel is (i0, j0), row-major."""
    ii, jj = np.meshgrid(np.arange(i0, i0 + size), np.arange(j0, j0 + size), indexing="ij")
    return (ii * W + jj).reshape(-1).astype(np.int32)


def repa3_to_image(p, canvas):
    canvas = np.ascontiguousarray(canvas, np.float64)
    out = np.empty((p.width, p.height, 3), np.uint8)
    L.o_repa3_to_image(C.byref(p), canvas.ctypes.data, out.ctypes.data)
    return out
