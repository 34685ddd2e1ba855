"""The C-ABI library loads without a GPU and exports every symbol include/pt.h
declares; the Python binding covers every one of them; with no device present
the product fails loudly instead of falling back to anything."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "pt.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"^[A-Za-z_][\w\s\*]*?\b(pt_[a-z0-9_]+)\s*\(", src, flags=re.M)
    return sorted(set(names))


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for must in ("pt_create", "pt_destroy", "pt_scene_upload", "pt_trace", "pt_read_sum", "pt_read_rgb8",
                 "pt_primary_hits", "pt_segments", "pt_last_error"):
        assert must in names
    assert len(names) >= 20


def test_library_exports_every_declared_symbol():
    from pathtracer_amd import _lib
    raw = C.CDLL(_lib.LIB_PATH)
    for name in declared_functions():
        assert hasattr(raw, name), f"libpt.so does not export {name}"
        assert name in _lib.SIGNATURES, f"pathtracer_amd._lib has no signature for {name}"
    assert set(_lib.SIGNATURES) == set(declared_functions())
    assert raw.pt_abi_version() == 3


def test_struct_layouts_match_the_header():
    from pathtracer_amd import _lib
    import oracle as O
    assert C.sizeof(_lib.pt_object) == 4 + 4 + 8 * 3 + 8 * 3 + 8 + 8 * 3 + 8 + 8 * 2 == 112
    assert C.sizeof(_lib.pt_params) == 40
    assert C.sizeof(O.OObject) == C.sizeof(_lib.pt_object)      # tests hand one byte image to both


def test_status_strings():
    from pathtracer_amd._lib import lib
    assert lib.pt_status_string(0) == b"ok"
    assert lib.pt_status_string(-2) == b"no usable HIP device"
    assert lib.pt_status_string(-77) == b"unknown status"


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="checks the no-GPU failure mode")
def test_no_gpu_fails_loudly():
    import pathtracer_amd as pa
    with pytest.raises(pa.PtError) as e:
        pa.Renderer(pa.Params(8, 8, 500.0, 1, 4))
    assert e.value.status == -2 and "no CPU path" in str(e.value)


def test_argument_validation_needs_no_gpu():
    from pathtracer_amd import _lib
    ctx = C.c_void_p()
    bad = _lib.pt_params(0, 8, 500.0, 4, 0, 0, 0, 1, 0)
    assert _lib.lib.pt_create(C.byref(ctx), C.byref(bad)) == _lib.PT_ERR_INVALID
    bad = _lib.pt_params(8, 8, 500.0, 65, 0, 0, 0, 1, 0)
    assert _lib.lib.pt_create(C.byref(ctx), C.byref(bad)) == _lib.PT_ERR_LIMIT
    assert _lib.lib.pt_create(None, None) == _lib.PT_ERR_INVALID
    assert _lib.lib.pt_trace(None, 1) == _lib.PT_ERR_INVALID
    assert b"null" in _lib.lib.pt_last_error(None)
    _lib.lib.pt_destroy(None)                                   # no-op, must not crash
    # multi-GPU entry points: argument checks come before any device or RCCL call
    ok = _lib.pt_params(8, 8, 500.0, 4, 0, 0, 0, 1, 0)
    devs = (C.c_int32 * 1)(0)
    assert _lib.lib.pt_create_multi(C.byref(ctx), C.byref(ok), devs, 0) == _lib.PT_ERR_LIMIT
    assert _lib.lib.pt_create_multi(C.byref(ctx), C.byref(ok), devs, 65) == _lib.PT_ERR_LIMIT
    assert _lib.lib.pt_create_multi(C.byref(ctx), C.byref(ok), None, 1) == _lib.PT_ERR_INVALID
    assert _lib.lib.pt_comm_unique_id(None) == _lib.PT_ERR_INVALID
    assert _lib.lib.pt_allreduce_sum(None) == _lib.PT_ERR_INVALID
    assert _lib.lib.pt_comm_init_rank(None, None, 0, 1) == _lib.PT_ERR_INVALID
    assert _lib.lib.pt_comm_share(None, None) == _lib.PT_ERR_INVALID


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under pathtracer_amd/, host/ or
    include/ may name it."""
    for base in ("pathtracer_amd", "host", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", ".c", "Makefile")):
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    assert "pt_oracle" not in text and "libpt_oracle" not in text, os.path.join(dirpath, f)
