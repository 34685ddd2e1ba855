"""Launch shapes around the kernels' internal thresholds (through the C ABI):
canvases that are not a multiple of the 64-pixel tile, sample counts on both
sides of the slicing threshold (512) and of the samples-per-launch limit
(4096), depth caps from 2 to 64, one and several pt_trace calls.  In every case
the default kernel (work sharing, kept primary hits, sample slices) must trace
the same Philox-keyed paths as the strict kernel: equal pt_segments (up to the
primary searches the default kernel does not repeat), sums equal to fp32
summation order."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pa():
    import pathtracer_amd
    return pathtracer_amd


CASES = [
    # H, W, spp chunks, bounces
    (7, 9, (511,), 8), (7, 9, (512,), 8), (7, 9, (513,), 8),
    (31, 33, (1023, 1), 5), (16, 16, (1024,), 2), (16, 16, (1025,), 3),
    (5, 13, (4096,), 4), (5, 13, (4097,), 4), (5, 13, (600, 600, 3), 64),
    (64, 1, (700,), 8), (1, 64, (700,), 8), (1, 1, (2000,), 30),
    (65, 3, (512, 512), 6),
]


@pytest.mark.parametrize("H,W,chunks,bounces", CASES)
def test_default_kernel_equals_strict_kernel(pa, H, W, chunks, bounces):
    spp = sum(chunks)
    P = pa.Params(H, W, 37.0, spp, bounces)
    with pa.Renderer(P, seed=13) as fast, pa.Renderer(P, seed=13, flags=pa.PT_FLAG_NO_STEAL) as strict:
        for n in chunks:
            fast.trace(n)
        strict.trace(spp)
        a, b = fast.read_sum_f32(), strict.read_sum_f32()
        assert fast.samples_done == strict.samples_done == spp
        # the strict kernel searches the primary hits in every launch (4096 samples at most each), the default
        # one once per scene
        assert fast.segments == strict.segments - (-(-spp // 4096) - 1) * H * W
    assert np.all(np.isfinite(a))
    assert np.allclose(a, b, rtol=1e-5, atol=1e-5)
    assert b.sum() > 0


@pytest.mark.parametrize("H,W,spp,bounces,n", [(7, 9, 40, 8, 50), (1, 1, 300, 16, 200), (65, 3, 9, 64, 1025),
                                               (64, 64, 2, 2, 300), (33, 31, 17, 5, 2500)])
def test_indexed_kernel_equals_strict_kernel(pa, H, W, spp, bounces, n):
    """trace_bvh_kernel (resumable searches, walk and shade steps, the wave's sample
    pool) against the strict kernel on the same indexed scene."""
    objs = pa.random_disk_scene(n, 40 + n)
    P = pa.Params(H, W, 60.0, spp, bounces)
    with pa.Renderer(P, objs, seed=2) as fast, pa.Renderer(P, objs, seed=2, flags=pa.PT_FLAG_NO_STEAL) as strict:
        fast.trace(spp)
        strict.trace(spp)
        a, b = fast.read_sum_f32(), strict.read_sum_f32()
        assert fast.segments == strict.segments
    assert np.all(np.isfinite(a))
    assert np.allclose(a, b, rtol=1e-5, atol=1e-5)
