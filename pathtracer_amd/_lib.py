"""ctypes binding of libpt.so (include/pt.h).

There is no Python or CPU fallback: if the HIP library is missing this module
raises at import, and if no GPU is present `pt_create` returns
PT_ERR_NO_DEVICE, which `check` turns into an exception.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PT_LIBPT_PATH") or os.path.join(_HERE, "libpt.so")    # override: A/B runs of another build

PT_OK = 0
PT_ERR_INVALID, PT_ERR_NO_DEVICE, PT_ERR_HIP, PT_ERR_NO_SCENE, PT_ERR_LIMIT = -1, -2, -3, -4, -5
PT_ERR_UNSUPPORTED = -6
PT_INFINITE_PLANE, PT_DISK, PT_SPHERE, PT_RECTANGLE = 0, 1, 2, 3
PT_RGB8_REFERENCE, PT_RGB8_MEAN_CLAMP, PT_RGB8_MEAN_GAMMA = 0, 1, 2
PT_FLAG_NO_PRIMARY_REUSE = 1
PT_FLAG_LDS_TILES = 2
PT_FLAG_NO_STEAL = 4
PT_FLAG_NO_BVH = 8


class pt_object(C.Structure):
    """Mirror of `pt_object` (include/pt.h); Object + Form of src/Object.hs:30-56."""
    _fields_ = [
        ("kind", C.c_int32),
        ("reflective", C.c_int32),
        ("point", C.c_double * 3),
        ("normal", C.c_double * 3),
        ("radius", C.c_double),
        ("color", C.c_double * 3),
        ("emittance", C.c_double),
        ("extent", C.c_double * 2),
    ]


class pt_params(C.Structure):
    """Mirror of `pt_params` (include/pt.h); src/Params.hs:3-11 at run time."""
    _fields_ = [
        ("height", C.c_int32),
        ("width", C.c_int32),
        ("camera_depth", C.c_double),
        ("max_bounces", C.c_int32),
        ("seed", C.c_uint32),
        ("device", C.c_int32),
        ("sample_offset", C.c_int32),
        ("sample_stride", C.c_int32),
        ("flags", C.c_int32),
    ]


class PtError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"libpt status {status}: {message}")
        self.status = status


# every symbol include/pt.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SIGNATURES = {
    "pt_abi_version": (C.c_int, []),
    "pt_status_string": (C.c_char_p, [C.c_int]),
    "pt_create": (C.c_int, [C.POINTER(_P), C.POINTER(pt_params)]),
    "pt_destroy": (None, [_P]),
    "pt_scene_upload": (C.c_int, [_P, C.POINTER(pt_object), C.c_int32]),
    "pt_synthetic_scene": (C.c_int, [C.POINTER(pt_object), C.c_int32, C.c_uint64, C.c_int32]),
    "pt_trace": (C.c_int, [_P, C.c_int32]),
    "pt_reset": (C.c_int, [_P]),
    "pt_read_sum": (C.c_int, [_P, _P]),
    "pt_read_sum_f32": (C.c_int, [_P, _P]),
    "pt_read_rgb8": (C.c_int, [_P, _P, C.c_int32]),
    "pt_read_rgb8_noalloc": (C.c_int, [_P, _P, C.c_int32]),
    "pt_primary_hits": (C.c_int, [_P, _P, _P, _P]),
    "pt_segments": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "pt_samples_done": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "pt_bvh_counters": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "pt_bvh_padding_gathers": (C.c_int, [_P, _P]),
    "pt_canvas_device_ptr": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "pt_set_stream": (C.c_int, [_P, _P]),
    "pt_synchronize": (C.c_int, [_P]),
    "pt_rgb8_device": (C.c_int, [_P, _P, C.c_int32, C.c_int64]),
    "pt_rgb8_device_on": (C.c_int, [_P, _P, C.c_int32, C.c_int64, _P]),
    "pt_write_sum_f32": (C.c_int, [_P, _P, C.c_int64]),
    "pt_dev_diffuse": (C.c_int, [_P, C.c_int32, _P, _P, _P]),
    "pt_dev_specular": (C.c_int, [_P, C.c_int32, _P, _P, _P, C.c_float, _P]),
    "pt_dev_distance": (C.c_int, [_P, C.c_int32, C.POINTER(pt_object), _P, _P, _P]),
    "pt_dev_box_test": (C.c_int, [_P, C.c_int32, _P, C.c_double, _P, _P, _P, _P]),
    "pt_dev_philox": (C.c_int, [_P, C.c_int32, _P, _P, C.c_uint32, _P]),
    "pt_scene_specialise": (C.c_int, [_P]),
    "pt_jit_info": (C.c_int, [_P, C.c_char_p, C.c_size_t]),
    "pt_jit_cache_size": (C.c_int, []),
    "pt_scene_kernel_ship": (C.c_int, [C.POINTER(pt_object), C.c_int32, C.c_char_p, C.c_char_p, C.c_char_p]),
    "pt_set_timing": (C.c_int, [_P, C.c_int32]),
    "pt_last_trace_ms": (C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
    "pt_create_multi": (C.c_int, [C.POINTER(_P), C.POINTER(pt_params), C.POINTER(C.c_int32), C.c_int32]),
    "pt_comm_unique_id": (C.c_int, [_P]),
    "pt_comm_init_rank": (C.c_int, [_P, _P, C.c_int32, C.c_int32]),
    "pt_comm_share": (C.c_int, [_P, _P]),
    "pt_trace_gate": (C.c_int, [_P, _P]),
    "pt_allreduce_sum": (C.c_int, [_P]),
    "pt_comm_info": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "pt_last_error": (C.c_char_p, [_P]),
}
PT_COMM_ID_BYTES = 128


def _share_torch_hip_runtime():
    """One HIP runtime per process.  The PyTorch-ROCm wheel bundles its own
    libamdhip64 and asks for it by the unversioned name, which the loader does
    not match against a copy that libpt.so pulled in first from /opt/rocm by its
    soname: torch would then start a SECOND runtime and find no device
    ("no ROCm-capable device is detected").  Loading the wheel's copy first
    makes both orders resolve to the same object (libpt.so's NEEDED entry matches
    it by soname, torch's by path).  No torch import happens here."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load(path=LIB_PATH):
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: build it with `make -C pathtracer_amd/csrc` "
            "(or __graft_entry__.build()); pathtracer_amd has no fallback path")
    _share_torch_hip_runtime()
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError if the ABI lost a symbol
        fn.restype, fn.argtypes = res, args
    return lib


lib = load()


def check(status, ctx=None):
    if status != PT_OK:
        msg = lib.pt_last_error(ctx)
        raise PtError(status, (msg or b"").decode() or lib.pt_status_string(status).decode())
