"""pt_scene_specialise (include/pt.h): trace_static_kernel compiled at run time
for one scene, zero-coefficient terms left out.  No counterpart in the
reference; the bar is the library's own general kernel, bit for bit -- same
canvas, same segment count -- on the default scene (all four objects along the
axes: every test shortened) and on scenes with general normals, mirrors,
spheres, rectangles and 5..8 objects (little or nothing to leave out)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pa():
    import pathtracer_amd
    return pathtracer_amd


def _render(pa, P, objs, spp, specialise, launches=1, **kw):
    with pa.Renderer(P, objs, seed=3, **kw) as r:
        if specialise:
            assert r.specialise(), r.last_error()
        for _ in range(launches):
            r.trace(spp)
        return r.read_sum_f32(), r.segments


def _mixed(pa, n):
    """n <= 8 objects: axis walls, tilted planes and disks, a mirror, a sphere, a rectangle, a lamp."""
    o = [pa.Object((1, 1, 1), 0, False, pa.InfinitePlane((0, 0, 100), (0, 0, -1)), "back wall"),
         pa.Object((1, 1, 1), 2.5, True, pa.Disk((0, 0, -1), (0, 0, 1), 25), "lamp"),
         pa.Object((0.9, 0.4, 0.3), 0, False, pa.InfinitePlane((0, -30, 0), (0.2, 1, 0.1)), "tilted wall"),
         pa.Object((0.8, 0.8, 1), 0, True, pa.Disk((5, 10, 70), (0.3, -0.4, -1), 30), "mirror disk"),
         pa.Object((0.3, 0.9, 0.4), 0, False, pa.Sphere((-10.0, -5.0, 60.0), 8.0), "ball"),
         pa.Object((0.9, 0.9, 0.2), 0, False, pa.Rectangle((25, 0, 60), (-1, 0, 0), 60, 40), "card"),
         pa.Object((1, 1, 1), 0, False, pa.InfinitePlane((0, 40, 0), (0, -2, 0)), "wall, normal of length 2"),
         pa.Object((0.5, 0.5, 0.5), 0, True, pa.InfinitePlane((-30, 0, 0), (1, 0, 0)), "mirror floor"),
         # 9-16: scene kernels only (the general path for these is the streamed object list)
         pa.Object((0.7, 0.2, 0.6), 0, False, pa.Disk((-12, 18, 55), (0.1, -0.9, -0.3), 9), "plate"),
         pa.Object((0.2, 0.6, 0.9), 0, False, pa.Sphere((12.0, -14.0, 45.0), 5.0), "small ball"),
         pa.Object((0.9, 0.6, 0.2), 0, False, pa.Rectangle((-20, -10, 80), (0.5, 0.2, -1), 30, 20), "tilted card"),
         pa.Object((1, 1, 1), 1.5, True, pa.Disk((20, 20, 30), (-0.4, -0.4, 0.6), 6), "second lamp"),
         pa.Object((0.6, 0.9, 0.9), 0, True, pa.Sphere((0.0, 25.0, 65.0), 7.0), "mirror ball"),
         pa.Object((0.4, 0.4, 0.9), 0, False, pa.Disk((-25, 30, 40), (1, -1, 0), 12), "disk in the corner"),
         pa.Object((0.9, 0.3, 0.9), 0, False, pa.Rectangle((0, -29, 50), (0, 1, 0), 25, 25), "rug on the tilted wall"),
         pa.Object((0.3, 0.3, 0.3), 0, False, pa.Disk((10, 0, 99), (0, 0, -1), 15), "picture on the back wall")]
    return o[:n]


def test_default_scene_bit_for_bit(pa):
    P = pa.Params(200, 136, 500.0, 1, 8)
    a, sa = _render(pa, P, pa.objects, 96, False)
    b, sb = _render(pa, P, pa.objects, 96, True)
    assert sa == sb and sa > 200 * 136 * 96
    assert np.array_equal(a, b)
    assert a.sum() > 0


@pytest.mark.parametrize("n", [2, 3, 4, 5, 6, 7, 8])
def test_mixed_scenes_bit_for_bit(pa, n):
    P = pa.Params(120, 150, 90.0, 1, 6)
    objs = _mixed(pa, n)
    a, sa = _render(pa, P, objs, 48, False)
    b, sb = _render(pa, P, objs, 48, True)
    assert sa == sb
    assert np.array_equal(a, b)
    assert a.sum() > 0


@pytest.mark.parametrize("n", [9, 12, 16])
def test_scene_kernels_for_nine_to_sixteen_objects(pa, n):
    """9-16 objects have no general by-value kernel (their geometry would not fit the scalar registers): the general
    path streams the object list.  The scene kernel carries the geometry in its code, so the search's arithmetic, the
    paths and pt_segments are the same; the canvases are compared to fp32 summation order and, when the two kernels
    happen to add a pixel's samples in the same order, found equal bit for bit."""
    P = pa.Params(120, 150, 90.0, 1, 6)
    objs = _mixed(pa, n)
    a, sa = _render(pa, P, objs, 48, False)
    b, sb = _render(pa, P, objs, 48, True)
    assert sa == sb
    np.testing.assert_allclose(b, a, rtol=4e-6, atol=1e-6)
    assert a.sum() > 0
    print(f"{n} objects: scene kernel vs streamed list bit for bit: {np.array_equal(a, b)}")
    # sliced (>= 512 samples) and repeated launches through the scene kernel
    with pa.Renderer(P, objs, seed=3) as r:
        assert r.specialise(), r.last_error()
        assert "hiprtc" in r.jit_info() or "hipcc at build time" in r.jit_info()
        r.trace(520); r.trace(8)
        c, sc = r.read_sum_f32(), r.segments
    with pa.Renderer(P, objs, seed=3) as r:
        r.trace(520); r.trace(8)
        np.testing.assert_allclose(c, r.read_sum_f32(), rtol=4e-6, atol=1e-6)
        # (the streamed kernel repeats the per-pixel primary search in its second launch; the by-value kernels keep it)
        assert sc + 120 * 150 == r.segments


def test_sliced_and_repeated_launches(pa):
    """>= 512 samples: the launch runs as two sample slices; then a second launch that reads the kept primary hits."""
    P = pa.Params(64, 128, 500.0, 1, 8)
    a, sa = _render(pa, P, pa.objects, 520, False, launches=2)
    b, sb = _render(pa, P, pa.objects, 520, True, launches=2)
    assert sa == sb
    assert np.array_equal(a, b)


def test_a_new_scene_drops_the_kernel(pa):
    P = pa.Params(64, 64, 90.0, 1, 5)
    ref, _ = _render(pa, P, _mixed(pa, 4), 32, False)
    with pa.Renderer(P, pa.objects, seed=3) as r:
        assert r.specialise()
        r.upload(_mixed(pa, 4))              # the default scene's kernel must not trace this one
        r.trace(32)
        assert np.array_equal(r.read_sum_f32(), ref)
        assert r.specialise()                # and the new scene can have its own
        r.reset()
        r.trace(32)
        assert np.array_equal(r.read_sum_f32(), ref)


def test_where_it_does_not_apply(pa):
    P = pa.Params(32, 32, 90.0, 1, 4)
    with pa.Renderer(P, pa.random_disk_scene(40, 1)) as r:      # not a by-value scene
        assert not r.specialise()
        assert "16 objects" in r.last_error()
        r.trace(4)
    with pa.Renderer(P, pa.objects, flags=pa.PT_FLAG_NO_STEAL) as r:
        assert not r.specialise()
        r.trace(4)


def test_scene_kernels_are_shared_kept_and_unloaded(pa):
    """The kernel cache (pt_jit.hip): contexts with one scene share one kernel; a kernel nobody holds stays loaded
    until more than 16 unused ones have piled up, then the least recently used go (hipModuleUnload); a kernel in use
    is never unloaded, however many there are."""
    P = pa.Params(32, 64, 500.0, 2, 4)
    def scene(k):                                           # distinct geometry per k
        return [pa.Object((1, 1, 1), 0.0, False, pa.InfinitePlane((0.0, 0.0, 100.0 + k), (0.0, 0.0, -1.0)), "wall"),
                pa.Object((1, 1, 1), 2.0, True, pa.Disk((0.0, 0.0, -1.0), (0.0, 0.0, 1.0), 20.0), "light")]
    with pa.Renderer(P, scene(0)) as a, pa.Renderer(P, scene(0)) as b:
        if not a.specialise():
            pytest.skip("pt_scene_specialise not available: " + a.last_error())
        base = pa.Renderer.jit_cache_size()
        assert b.specialise() and pa.Renderer.jit_cache_size() == base          # the same kernel
        assert a.specialise() and pa.Renderer.jit_cache_size() == base          # asked twice: still one hold
        a.trace(2); b.trace(2)
        assert np.array_equal(a.read_sum_f32(), b.read_sum_f32())
    held = []
    try:
        for k in range(1, 20):                              # 19 more scenes, all in use at once
            r = pa.Renderer(P, scene(k))
            assert r.specialise()
            held.append(r)
        assert pa.Renderer.jit_cache_size() >= 19           # none of them may go
        for r in held:
            r.trace(1)
            assert r.read_sum_f32().any()
    finally:
        for r in held:
            r.close()
    assert pa.Renderer.jit_cache_size() <= 16               # released: trimmed to what is kept for the next user
    with pa.Renderer(P, scene(19)) as r:                    # the most recent one is still there (no compile: same count)
        n = pa.Renderer.jit_cache_size()
        assert r.specialise() and pa.Renderer.jit_cache_size() == n
        r.trace(2)
        assert r.read_sum_f32().any()


def test_scene_kernels_are_kept_on_disk(pa, tmp_path):
    """The on-disk cache of compiled scene kernels (pt_jit.hip): a second PROCESS with the same scene, library and
    compiler loads the code object from $PT_KERNEL_CACHE instead of compiling it, and renders the same canvas bit for
    bit; another scene is another file; PT_KERNEL_CACHE=off keeps the directory untouched; a damaged file is a miss."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = ("import sys, json, hashlib; sys.path.insert(0, %r)\n"
            "import pathtracer_amd as pa\n"
            "objs = pa.objects if sys.argv[1] == 'default' else pa.objects[:3]\n"
            "with pa.Renderer(pa.Params(96, 80, 500.0, 1, 6), objs, seed=3) as r:\n"
            "    assert r.specialise(), r.last_error()\n"
            "    r.trace(24)\n"
            "    print(json.dumps({'info': r.jit_info(), 'sha': hashlib.sha256(r.read_sum_f32().tobytes()).hexdigest()}))\n" % root)

    def run(scene, cache):
        env = dict(os.environ, PT_KERNEL_CACHE=cache, PT_SHIPPED_KERNELS="off")     # (the default scene's kernel is also shipped with the library: not here)
        env.pop("PT_SPECIALISE", None)
        p = subprocess.run([sys.executable, "-c", prog, scene], env=env, capture_output=True, text=True, timeout=120)
        assert p.returncode == 0, p.stderr[-2000:]
        return json.loads(p.stdout.strip().splitlines()[-1])

    cache = tmp_path / "kernels"
    first = run("default", str(cache))
    assert "compiled in this process" in first["info"]
    files = sorted(os.listdir(cache))
    assert len(files) == 1 and files[0].endswith(".co")
    second = run("default", str(cache))
    assert "from the on-disk cache" in second["info"] and second["sha"] == first["sha"]
    assert sorted(os.listdir(cache)) == files
    other = run("three planes", str(cache))                 # another scene: compiled, a second file
    assert "compiled in this process" in other["info"] and len(os.listdir(cache)) == 2
    off = tmp_path / "unused"
    assert "compiled in this process" in run("default", "off")["info"] and not off.exists()
    # a truncated file is not trusted: compiled again (and replaced)
    path = cache / files[0]
    data = path.read_bytes()
    path.write_bytes(data[: len(data) // 2])
    again = run("default", str(cache))
    assert "compiled in this process" in again["info"] and again["sha"] == first["sha"]
    assert path.read_bytes() == data or len(path.read_bytes()) == len(data)


def test_default_scene_kernel_is_shipped_with_the_library(pa):
    """`make scene_kernels` compiles the reference's constant scene (src/Object.hs:147-148) offline with hipcc and stores
    the code object next to libpt.so (pt_scene_kernel_ship): a process that may not load a compiler (PT_NO_HIPRTC=1)
    and has no kernel cache still gets the scene kernel for it -- the same canvas bit for bit as the general kernel --,
    and for any other scene an honest PT_ERR_UNSUPPORTED; with the shipped directory ignored too, none at all."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = ("import sys, json, hashlib; sys.path.insert(0, %r)\n"
            "import pathtracer_amd as pa\n"
            "objs = pa.objects if sys.argv[1] == 'default' else pa.objects[:3]\n"
            "with pa.Renderer(pa.Params(96, 80, 500.0, 1, 6), objs, seed=3) as r:\n"
            "    r.trace(24); g = hashlib.sha256(r.read_sum_f32().tobytes()).hexdigest(); r.reset()\n"
            "    ok = r.specialise(); err = r.last_error()\n"
            "    r.trace(24)\n"
            "    print(json.dumps({'ok': ok, 'err': err, 'info': r.jit_info(), 'general': g,\n"
            "                      'sha': hashlib.sha256(r.read_sum_f32().tobytes()).hexdigest()}))\n" % root)

    def run(scene, **extra):
        env = dict(os.environ, PT_KERNEL_CACHE="off", PT_NO_HIPRTC="1", **extra)
        env.pop("PT_SPECIALISE", None)
        p = subprocess.run([sys.executable, "-c", prog, scene], env=env, capture_output=True, text=True, timeout=120)
        assert p.returncode == 0, p.stderr[-2000:]
        return json.loads(p.stdout.strip().splitlines()[-1])

    a = run("default")
    assert a["ok"] and "shipped with the library" in a["info"] and "hipcc at build time" in a["info"], a
    assert a["sha"] == a["general"]                          # the same image as the general kernel, bit for bit
    b = run("three planes")
    assert not b["ok"] and b["info"] == "" and "no kernel for this scene was shipped" in b["err"], b
    assert b["sha"] == b["general"]                          # the general kernel stayed in use
    c = run("default", PT_SHIPPED_KERNELS="off")
    assert not c["ok"] and c["info"] == "", c
