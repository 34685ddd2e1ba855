"""pt_read_rgb8_noalloc (include/pt.h): the one entry point that allocates
nothing -- its RGB8 staging buffer is a static device array -- against
pt_read_rgb8, byte for byte, on two calls in a row, the second with a smaller
image (the array keeps the first call's bytes beyond the second image's end).
Collected last (test_zz_*): pytest -x would hide every later test behind a
failure here."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_noalloc_read_equals_read_rgb8_twice_second_smaller():
    import pathtracer_amd as pa
    for (H, W, spp) in ((300, 200, 5), (64, 48, 3)):                 # the second image is smaller
        with pa.Renderer(pa.Params(H, W, 500.0, spp, 6), seed=2) as r:
            r.trace(spp)
            for mode in (pa.PT_RGB8_REFERENCE, pa.PT_RGB8_MEAN_CLAMP, pa.PT_RGB8_MEAN_GAMMA):
                got = r.read_rgb8_noalloc(mode)
                want = r.read_rgb8(mode)
                assert got.shape == (W, H, 3)
                assert np.array_equal(got, want), (H, W, mode)
            assert r.read_rgb8(pa.PT_RGB8_REFERENCE).any()


def test_noalloc_read_at_the_bench_size_and_beyond_the_limit():
    import pathtracer_amd as pa
    with pa.Renderer(pa.Params(1920, 1080, 500.0, 2, 4), seed=1) as r:   # 6,220,800 bytes: fits the 8 MiB array
        r.trace(2)
        assert np.array_equal(r.read_rgb8_noalloc(), r.read_rgb8())
    with pa.Renderer(pa.Params(2000, 1500, 500.0, 1, 4)) as r:          # 9,000,000 bytes: refused, nothing written
        with pytest.raises(pa.PtError) as e:
            r.read_rgb8_noalloc()
        assert e.value.status == pa.PT_ERR_LIMIT and "8388608" in str(e.value)


def test_noalloc_read_of_a_device_group():
    import pathtracer_amd as pa
    with pa.Renderer(pa.Params(96, 64, 500.0, 6, 6), seed=4, devices=[0, 0]) as g:
        g.trace(6)
        assert np.array_equal(g.read_rgb8_noalloc(pa.PT_RGB8_MEAN_CLAMP), g.read_rgb8(pa.PT_RGB8_MEAN_CLAMP))


def _free_bytes():
    import ctypes as C
    hip = C.CDLL("libamdhip64.so.7")                                  # the runtime libpt.so is linked against
    free, total = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def test_noalloc_read_leaves_the_free_device_memory_unchanged():
    """hipMemGetInfo's free bytes before and after the call are equal (the first call of a process may load the
    kernel's code object, so one call warms up); pt_read_rgb8 on a fresh context, by contrast, allocates."""
    import pathtracer_amd as pa
    with pa.Renderer(pa.Params(640, 480, 500.0, 2, 4), seed=3) as r:
        r.trace(2)
        r.read_rgb8_noalloc()
        r.synchronize()
        before = _free_bytes()
        for mode in (pa.PT_RGB8_REFERENCE, pa.PT_RGB8_MEAN_GAMMA):
            got = r.read_rgb8_noalloc(mode)
            assert _free_bytes() == before, mode
        want = r.read_rgb8(pa.PT_RGB8_MEAN_GAMMA)                     # allocates its staging buffer on first use
        assert np.array_equal(got, want)
        assert _free_bytes() <= before


def test_noalloc_reads_of_two_contexts_on_one_device_take_turns():
    """Two contexts on one GPU share that GPU's static array; reads from two threads must each return their own
    image (the call holds a per-GPU lock until its copy out of the array has completed)."""
    import threading
    import pathtracer_amd as pa
    a = pa.Renderer(pa.Params(512, 384, 500.0, 3, 6), seed=5)
    b = pa.Renderer(pa.Params(512, 384, 500.0, 3, 6), seed=6)
    try:
        a.trace(3); b.trace(3)
        want = {id(a): a.read_rgb8(), id(b): b.read_rgb8()}
        assert not np.array_equal(want[id(a)], want[id(b)])
        bad = []

        def reader(r):
            for _ in range(40):
                if not np.array_equal(r.read_rgb8_noalloc(), want[id(r)]):
                    bad.append(id(r))
        ts = [threading.Thread(target=reader, args=(r,)) for r in (a, b)]
        for t in ts: t.start()
        for t in ts: t.join()
        assert not bad
    finally:
        a.close(); b.close()
