"""The run-time compiled scene kernel (pt_scene_specialise, pathtracer_amd/csrc/pt_jit.hip)
without a GPU: its source must compile under hiprtc for gfx950 -- with the
compiler of /opt/rocm and with the older one bundled with PyTorch, which a
process that imported torch ends up using --, and libpt.so must carry the
CURRENT text of the three headers it hands to hiprtc (a stale library would
compile yesterday's kernel)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pathtracer_amd", "csrc")
CHECK = os.path.join(ROOT, "tools", "jit_compile_check.py")


def test_library_embeds_the_current_kernel_source():
    lib = open(os.path.join(ROOT, "pathtracer_amd", "libpt.so"), "rb").read()
    for name in ("pt_static_kernel.hpp", "pt_internal.hpp", "pt_device.hpp"):
        text = open(os.path.join(CSRC, name), "rb").read()
        assert text in lib, f"libpt.so was built from another {name}: run make in pathtracer_amd/csrc"


@pytest.mark.parametrize("ns,spec,compiler", [(4, 0, "rocm"), (8, 1, "rocm"), (4, 0, "torch"), (8, 1, "torch")])
def test_scene_kernel_compiles_with_hiprtc(ns, spec, compiler):
    try:
        import ctypes
        ctypes.CDLL("libhiprtc.so")
    except OSError:
        pytest.skip("no libhiprtc.so on this machine")
    args = [sys.executable, CHECK, str(ns), str(spec)] + (["torch"] if compiler == "torch" else [])
    r = subprocess.run(args, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert ".vgpr_spill_count: 0" in r.stdout
    if ns == 4:
        # the default scene's back wall and light are parallel: 3 reciprocals per loop instead of 4, two loops
        assert "v_rcp_f32 in the code object: 6" in r.stdout
        vgprs = int(r.stdout.split(".vgpr_count:")[1].split()[0])
        assert vgprs <= 64, "the scene kernel lost the eighth wave per SIMD"
