"""haskell/Lib.hs, checked as far as is possible without GHC (the
image has none).  The shim marshals Params.* (src/Params.hs:3-11) and
Object.objects (src/Object.hs:30-56) into pt_params / pt_object with byte
offsets written by hand; they are parsed out of the Haskell text
(tools/shim_offsets.py) and compared with offsetof / sizeof of include/pt.h as
a C compiler sees it.  Every `foreign import` is checked against the header:
the symbol exists and takes that many arguments.  (The shim's call SEQUENCE
runs on the GPU as host/shim_replay.c, tests/test_gpu_host.py.)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

PROBE = r"""
#include <stddef.h>
#include <stdio.h>
#include "pt.h"
int main(void) {
    printf("PARAMS_SIZE %zu\n", sizeof(pt_params));
    printf("PARAMS_HEIGHT %zu\n", offsetof(pt_params, height));
    printf("PARAMS_WIDTH %zu\n", offsetof(pt_params, width));
    printf("PARAMS_CAMERA_DEPTH %zu\n", offsetof(pt_params, camera_depth));
    printf("PARAMS_MAX_BOUNCES %zu\n", offsetof(pt_params, max_bounces));
    printf("PARAMS_SEED %zu\n", offsetof(pt_params, seed));
    printf("PARAMS_DEVICE %zu\n", offsetof(pt_params, device));
    printf("PARAMS_SAMPLE_OFFSET %zu\n", offsetof(pt_params, sample_offset));
    printf("PARAMS_SAMPLE_STRIDE %zu\n", offsetof(pt_params, sample_stride));
    printf("PARAMS_FLAGS %zu\n", offsetof(pt_params, flags));
    printf("OBJECT_SIZE %zu\n", sizeof(pt_object));
    printf("OBJECT_KIND %zu\n", offsetof(pt_object, kind));
    printf("OBJECT_REFLECTIVE %zu\n", offsetof(pt_object, reflective));
    printf("OBJECT_POINT %zu\n", offsetof(pt_object, point));
    printf("OBJECT_NORMAL %zu\n", offsetof(pt_object, normal));
    printf("OBJECT_RADIUS %zu\n", offsetof(pt_object, radius));
    printf("OBJECT_COLOR %zu\n", offsetof(pt_object, color));
    printf("OBJECT_EMITTANCE %zu\n", offsetof(pt_object, emittance));
    printf("OBJECT_EXTENT0 %zu\n", offsetof(pt_object, extent));
    printf("OBJECT_EXTENT1 %zu\n", offsetof(pt_object, extent) + sizeof(double));
    return 0;
}
"""


def test_shim_offsets_match_the_header(tmp_path):
    import shim_offsets
    src = tmp_path / "probe.c"
    src.write_text(PROBE)
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    want = dict((k, int(v)) for k, v in (ln.split() for ln in subprocess.check_output([str(exe)], text=True).splitlines()))
    got = shim_offsets.parse()
    assert got == want
    # the instances' alignment is that of the widest member (double)
    text = open(os.path.join(ROOT, "haskell", "Lib.hs")).read()
    assert text.count("alignment _ = 8") == 2


def _header_arity():
    src = open(os.path.join(ROOT, "include", "pt.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(pt_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", src):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else len(args.split(","))
    return out


def test_every_foreign_import_matches_the_header():
    text = open(os.path.join(ROOT, "haskell", "Lib.hs")).read()
    arity = _header_arity()
    imports = re.findall(r'foreign import ccall safe "(&?)(pt_[a-z0-9_]+)"\s+\w+\s*::\s*(.+)', text)
    assert len(imports) >= 8
    for amp, name, sig in imports:
        assert name in arity, f"{name} is not declared in include/pt.h"
        if amp:                                             # a function pointer (the finalizer): FunPtr (a -> IO ())
            sig = re.search(r"FunPtr \((.*)\)", sig).group(1)
        n_args = len(re.split(r"->", sig)) - 1
        assert n_args == arity[name], (name, sig, arity[name])
    names = {n for _, n, _ in imports}
    # what the shim's `context`, `fetch` and `check` need
    assert {"pt_create", "pt_scene_upload", "pt_trace", "pt_reset", "pt_samples_done", "pt_read_sum",
            "pt_last_error", "pt_destroy"} <= names


def test_elements_hold_no_buffer():
    """`traces` is a CAF that app/Web.hs:43 walks for ever: an element may keep its index, not H*W*3 doubles."""
    text = open(os.path.join(ROOT, "haskell", "Lib.hs")).read()
    assert text.count("mallocForeignPtrArray") == 1          # the one buffer of `slot`
    body = text[text.index("element :: Int"):text.index("traces ::")]
    assert "malloc" not in body and "fetch k" in body
