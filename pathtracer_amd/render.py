"""Host side of the radiance loop: `Renderer` owns one libpt context (one GPU);
`traces` and `repa3ToImage` keep the reference's names and meaning
(src/Lib.hs:53-56, src/Conversion.hs:22-28)."""
import ctypes as C
from typing import Iterator, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import check, lib
from .scene import Object, Params, objects as default_objects, to_c_array


class Renderer:
    """One pt_ctx: Params + an uploaded scene + the fp32 radiance-sum canvas in HBM.
    `devices=[...]` makes it a pt_create_multi context: the same interface over several GPUs of this process,
    every pt_trace sharded by sample index and the reads all-reduced over RCCL inside libpt."""

    def __init__(self, params: Params, objs: Optional[Sequence[Object]] = None, *, seed: int = 0,
                 device: int = 0, sample_offset: int = 0, sample_stride: int = 1, flags: int = 0,
                 devices: Optional[Sequence[int]] = None):
        self.params = params
        p = _lib.pt_params(params.height, params.width, float(params.cameraDepth), params.maxBounces,
                           seed, device, sample_offset, sample_stride, flags)
        self._ctx = C.c_void_p()
        if devices is None:
            check(lib.pt_create(C.byref(self._ctx), C.byref(p)))
        else:
            arr = (C.c_int32 * len(devices))(*devices)
            check(lib.pt_create_multi(C.byref(self._ctx), C.byref(p), arr, len(devices)))
        self.shape = (params.height, params.width, 3)
        if objs is None:
            objs = default_objects
        self.upload(objs)

    # -- lifetime ---------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None):
            lib.pt_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- the path -----------------------------------------------------------
    def upload(self, objs: Sequence[Object]):
        arr = to_c_array(objs)
        check(lib.pt_scene_upload(self._ctx, arr, len(objs)), self._ctx)
        self.n_objects = len(objs)

    def trace(self, iters: int = 1):
        """`iters` applications of traceCanvas (src/Lib.hs:58-65)."""
        check(lib.pt_trace(self._ctx, iters), self._ctx)

    def reset(self):
        check(lib.pt_reset(self._ctx), self._ctx)

    def synchronize(self):
        check(lib.pt_synchronize(self._ctx), self._ctx)

    def read_sum(self) -> np.ndarray:
        """Element `samples_done` of Lib.traces: (H, W, 3) float64 SUM of samples."""
        out = np.empty(self.shape, dtype=np.float64)
        check(lib.pt_read_sum(self._ctx, out.ctypes.data_as(C.c_void_p)), self._ctx)
        return out

    def read_sum_f32(self) -> np.ndarray:
        out = np.empty(self.shape, dtype=np.float32)
        check(lib.pt_read_sum_f32(self._ctx, out.ctypes.data_as(C.c_void_p)), self._ctx)
        return out

    def write_sum_f32(self, canvas: np.ndarray, samples_done: int = 0):
        a = np.ascontiguousarray(canvas, dtype=np.float32)
        assert a.shape == self.shape
        check(lib.pt_write_sum_f32(self._ctx, a.ctypes.data_as(C.c_void_p), samples_done), self._ctx)

    # Checkpoint / resume.  The reference's resumable state is the
    # per-pixel pair (Vec3 sum, StdGen) threaded by `iterate` (src/Lib.hs:56-65); with a
    # counter-based generator it is just (radiance sums, samples done).
    def checkpoint(self) -> dict:
        return {"sum": self.read_sum_f32(), "samples_done": self.samples_done,
                "params": (self.params.height, self.params.width, self.params.cameraDepth, self.params.maxBounces)}

    def restore(self, state: dict):
        want = (self.params.height, self.params.width, self.params.cameraDepth, self.params.maxBounces)
        if tuple(state["params"]) != want:
            raise ValueError(f"checkpoint was taken with Params {tuple(state['params'])}, this renderer has {want}")
        self.write_sum_f32(state["sum"], int(state["samples_done"]))

    def read_rgb8(self, mode: int = _lib.PT_RGB8_REFERENCE) -> np.ndarray:
        """repa3ToImage: array (W, H, 3) uint8 = rows of an image H wide, W tall."""
        out = np.empty((self.params.width, self.params.height, 3), dtype=np.uint8)
        check(lib.pt_read_rgb8(self._ctx, out.ctypes.data_as(C.c_void_p), mode), self._ctx)
        return out

    def read_rgb8_noalloc(self, mode: int = _lib.PT_RGB8_REFERENCE) -> np.ndarray:
        """read_rgb8 through the entry point that allocates nothing on the device (pt_read_rgb8_noalloc)."""
        out = np.empty((self.params.width, self.params.height, 3), dtype=np.uint8)
        check(lib.pt_read_rgb8_noalloc(self._ctx, out.ctypes.data_as(C.c_void_p), mode), self._ctx)
        return out

    def primary_hits(self):
        H, W = self.params.height, self.params.width
        idx = np.empty((H, W), dtype=np.int32)
        t = np.empty((H, W), dtype=np.float64)
        nrm = np.empty((H, W, 3), dtype=np.float64)
        check(lib.pt_primary_hits(self._ctx, idx.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p),
                                  nrm.ctypes.data_as(C.c_void_p)), self._ctx)
        return idx, t, nrm

    @property
    def segments(self) -> int:
        v = C.c_uint64()
        check(lib.pt_segments(self._ctx, C.byref(v)), self._ctx)
        return v.value

    def bvh_counters(self):
        """(16-byte gathers of index nodes, 16-byte gathers of object records) since the last reset, for scenes
        searched through the spatial index."""
        a, b = C.c_uint64(), C.c_uint64()
        check(lib.pt_bvh_counters(self._ctx, C.byref(a), C.byref(b)), self._ctx)
        return a.value, b.value

    def bvh_padding_gathers(self) -> int:
        """The part of bvh_counters()[1] spent on the empty slot of leaves with an odd number of objects."""
        a = C.c_uint64()
        check(lib.pt_bvh_padding_gathers(self._ctx, C.byref(a)), self._ctx)
        return a.value

    @property
    def samples_done(self) -> int:
        v = C.c_int64()
        check(lib.pt_samples_done(self._ctx, C.byref(v)), self._ctx)
        return v.value

    # -- device-side access -------------------------------------------------
    def canvas_device_ptr(self):
        p, n = C.c_void_p(), C.c_size_t()
        check(lib.pt_canvas_device_ptr(self._ctx, C.byref(p), C.byref(n)), self._ctx)
        return p.value, n.value

    def set_stream(self, hip_stream: int):
        check(lib.pt_set_stream(self._ctx, C.c_void_p(hip_stream)), self._ctx)

    def trace_gate(self, after: "Renderer") -> bool:
        """pt_trace_gate: what is queued on this context's stream next waits until `after`'s most recent trace launch
        has dispatched its last workgroup (it only drains from then on).  False where the device cannot do it."""
        s = lib.pt_trace_gate(self._ctx, after._ctx)
        if s == _lib.PT_ERR_UNSUPPORTED:
            return False
        check(s, self._ctx)
        return True

    def rgb8_device(self, d_out: int, mode: int = _lib.PT_RGB8_REFERENCE, samples: int = 0):
        check(lib.pt_rgb8_device(self._ctx, C.c_void_p(d_out), mode, samples), self._ctx)

    def rgb8_device_on(self, d_out: int, hip_stream: int, mode: int = _lib.PT_RGB8_REFERENCE, samples: int = 0):
        """pt_rgb8_device on a caller-named stream (the caller orders it)."""
        check(lib.pt_rgb8_device_on(self._ctx, C.c_void_p(d_out), mode, samples, C.c_void_p(hip_stream)), self._ctx)

    # -- multi-GPU, one process per GPU (include/pt.h: pt_comm_*) -------------
    @staticmethod
    def comm_unique_id() -> bytes:
        """Rank 0's RCCL id, to be handed to every other rank (PT_COMM_ID_BYTES bytes)."""
        buf = C.create_string_buffer(_lib.PT_COMM_ID_BYTES)
        check(lib.pt_comm_unique_id(buf))
        return buf.raw

    def comm_init_rank(self, uid: bytes, rank: int, n_ranks: int):
        if len(uid) != _lib.PT_COMM_ID_BYTES:
            raise ValueError("comm_init_rank: the id is PT_COMM_ID_BYTES bytes")
        check(lib.pt_comm_init_rank(self._ctx, C.c_char_p(uid), rank, n_ranks), self._ctx)

    def comm_share(self, owner: "Renderer"):
        """pt_comm_share: use `owner`'s RCCL communicator (a second canvas of the same rank, frames in flight)."""
        check(lib.pt_comm_share(self._ctx, owner._ctx), self._ctx)

    def allreduce_sum(self):
        """One ncclAllReduce(sum) of the canvas over all ranks, queued on the context's stream; the reads
        return the total until the next trace / reset."""
        check(lib.pt_allreduce_sum(self._ctx), self._ctx)

    def comm_info(self) -> dict:
        a, b, c, d = (C.c_int32() for _ in range(4))
        check(lib.pt_comm_info(self._ctx, C.byref(a), C.byref(b), C.byref(c), C.byref(d)), self._ctx)
        return {"n_ranks": a.value, "rank": b.value, "n_local": c.value, "rccl_version": d.value}

    @staticmethod
    def jit_cache_size() -> int:
        """pt_jit_cache_size: scene kernels the process holds loaded."""
        return lib.pt_jit_cache_size()

    def specialise(self) -> bool:
        """pt_scene_specialise: compile the trace kernel for this scene (seconds; worth it for a scene that is
        traced for a while).  True if the scene kernel is now in use, False if the library reported that the
        step is not available (the general kernel stays in use; `last_error` says why)."""
        s = lib.pt_scene_specialise(self._ctx)
        if s == _lib.PT_ERR_UNSUPPORTED:
            return False
        check(s, self._ctx)
        return True

    def jit_info(self) -> str:
        """Which hiprtc compiled the scene kernel in use ("" = the general kernel)."""
        buf = C.create_string_buffer(512)
        check(lib.pt_jit_info(self._ctx, buf, len(buf)), self._ctx)
        return buf.value.decode(errors="replace")

    def last_error(self) -> str:
        return (lib.pt_last_error(self._ctx) or b"").decode(errors="replace")

    def set_timing(self, on: bool = True):
        check(lib.pt_set_timing(self._ctx, 1 if on else 0), self._ctx)

    def last_trace_ms(self):
        ms, n = C.c_float(), C.c_int32()
        check(lib.pt_last_trace_ms(self._ctx, C.byref(ms), C.byref(n)), self._ctx)
        return ms.value, n.value

    # -- device-function parity hooks --------------------------------------
    def dev_diffuse(self, normal, u):
        normal = np.ascontiguousarray(normal, dtype=np.float32)
        u = np.ascontiguousarray(u, dtype=np.float32)
        out = np.empty_like(normal)
        check(lib.pt_dev_diffuse(self._ctx, len(normal), normal.ctypes.data_as(C.c_void_p),
                                 u.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)), self._ctx)
        return out

    def dev_specular(self, vector, normal, u, noise: float = -1.0):
        vector = np.ascontiguousarray(vector, dtype=np.float32)
        normal = np.ascontiguousarray(normal, dtype=np.float32)
        u = np.ascontiguousarray(u, dtype=np.float32)
        out = np.empty_like(vector)
        check(lib.pt_dev_specular(self._ctx, len(vector), vector.ctypes.data_as(C.c_void_p),
                                  normal.ctypes.data_as(C.c_void_p), u.ctypes.data_as(C.c_void_p),
                                  noise, out.ctypes.data_as(C.c_void_p)), self._ctx)
        return out

    def dev_distance(self, obj: Object, origin, vector):
        origin = np.ascontiguousarray(origin, dtype=np.float32)
        vector = np.ascontiguousarray(vector, dtype=np.float32)
        out = np.empty(len(origin), dtype=np.float32)
        co = obj.to_c()
        check(lib.pt_dev_distance(self._ctx, len(origin), C.byref(co), origin.ctypes.data_as(C.c_void_p),
                                  vector.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)), self._ctx)
        return out

    def dev_box_test(self, boxes, pad: float, origin, vector, t_best):
        """The spatial index's box test on (n, 6) boxes (lo, hi), encoded as the builder encodes them."""
        boxes = np.ascontiguousarray(boxes, dtype=np.float64)
        origin = np.ascontiguousarray(origin, dtype=np.float32)
        vector = np.ascontiguousarray(vector, dtype=np.float32)
        t_best = np.ascontiguousarray(t_best, dtype=np.float32)
        out = np.empty(len(boxes), dtype=np.uint8)
        check(lib.pt_dev_box_test(self._ctx, len(boxes), boxes.ctypes.data_as(C.c_void_p), float(pad),
                                  origin.ctypes.data_as(C.c_void_p), vector.ctypes.data_as(C.c_void_p),
                                  t_best.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)), self._ctx)
        return out

    def dev_philox(self, c0, c1, key: int):
        c0 = np.ascontiguousarray(c0, dtype=np.uint32)
        c1 = np.ascontiguousarray(c1, dtype=np.uint32)
        out = np.empty((len(c0), 2), dtype=np.uint32)
        check(lib.pt_dev_philox(self._ctx, len(c0), c0.ctypes.data_as(C.c_void_p),
                                c1.ctypes.data_as(C.c_void_p), key, out.ctypes.data_as(C.c_void_p)), self._ctx)
        return out


def traces(params: Params, objs: Optional[Sequence[Object]] = None, **kw) -> Iterator[np.ndarray]:
    """Lib.traces (src/Lib.hs:53-56): element k is the (H, W, 3) float64 sum of
    k samples per pixel; element 0 is the black canvas.  Lazy, like the list:
    pulling element k+1 runs one more traceCanvas on the GPU."""
    with Renderer(params, objs, **kw) as r:
        yield r.read_sum()
        while True:
            r.trace(1)
            yield r.read_sum()


def repa3ToImage(renderer: Renderer, mode: int = _lib.PT_RGB8_REFERENCE) -> np.ndarray:
    """Conversion.repa3ToImage (src/Conversion.hs:22-28) of the renderer's
    current canvas: uint8 rows of an image Params.height wide, Params.width tall."""
    return renderer.read_rgb8(mode)
