"""Scene kernels shipped with the library (pt_scene_kernel_ship, include/pt.h; no GPU): the build (`make scene_kernels`,
__graft_entry__.build()) has stored a kernel for the reference's constant scene (src/Object.hs:147-148) that matches
THIS library's kernel source -- a stale file would simply never be found, and the host would silently fall back to
compiling at run time -- and the source the library hands out for offline compilation is the hiprtc one."""
import ctypes as C
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARCH = b"gfx950:sramecc+:xnack-"          # MI355X as hipDeviceProp_t::gcnArchName spells it


@pytest.fixture(scope="module")
def pa():
    import pathtracer_amd
    return pathtracer_amd


def test_the_default_scene_has_a_shipped_kernel_for_this_build(pa):
    from pathtracer_amd import _lib, scene
    objs = scene.to_c_array(scene.objects)
    assert _lib.lib.pt_scene_kernel_ship(objs, len(scene.objects), ARCH, None, None) == _lib.PT_OK, \
        "run `make -C pathtracer_amd/csrc scene_kernels` (or __graft_entry__.build())"
    files = [f for f in os.listdir(os.path.join(ROOT, "pathtracer_amd", "scene_kernels")) if f.endswith(".co")]
    assert len(files) == 1                                   # and nothing left over from earlier sources
    # another scene, another target: not shipped
    three = scene.to_c_array(scene.objects[:3])
    assert _lib.lib.pt_scene_kernel_ship(three, 3, ARCH, None, None) == _lib.PT_ERR_UNSUPPORTED
    assert _lib.lib.pt_scene_kernel_ship(objs, len(scene.objects), b"gfx942", None, None) == _lib.PT_ERR_UNSUPPORTED


def test_the_source_for_offline_compilation(pa, tmp_path):
    from pathtracer_amd import _lib, scene
    objs = scene.to_c_array(scene.objects)
    out = tmp_path / "k.hip"
    assert _lib.lib.pt_scene_kernel_ship(objs, len(scene.objects), ARCH, str(out).encode(), None) == _lib.PT_OK
    text = out.read_text()
    assert "#define PT_JIT 1" in text and "#define PT_JIT_NS 4" in text and "#define PT_JIT_SPEC 0" in text
    assert text.count("PT_JIT_F(0x") == 4 * 12 and '#include "pt_static_kernel.hpp"' in text
    # argument checks
    assert _lib.lib.pt_scene_kernel_ship(None, 4, ARCH, None, None) == _lib.PT_ERR_INVALID
    assert _lib.lib.pt_scene_kernel_ship(objs, 17, ARCH, None, None) == _lib.PT_ERR_INVALID
    assert _lib.lib.pt_scene_kernel_ship(objs, 4, ARCH, None, str(tmp_path / "missing.co").encode()) == _lib.PT_ERR_INVALID
