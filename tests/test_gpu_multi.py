"""Multi-GPU behind the C ABI (include/pt.h: pt_create_multi, pt_comm_*,
pt_allreduce_sum; pathtracer_amd/csrc/pt_multi.hip).  The reference has ONE
`traces` (src/Lib.hs:53-56): a context over several devices, or a rank context
after its all-reduce, must return what one GPU returns -- the same Philox-keyed
paths, sums equal up to fp32 summation order, bit for bit with one member.

The box has one GPU, so the device list here names device 0 (once: RCCL with a
world of 1; several times: the group's sample split, partial canvases and
read-after-continue logic, summed on the device before the collective)."""
import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu

H, W, B = 96, 128, 8


def plain(spp, seed=5, **kw):
    import pathtracer_amd as pa
    with pa.Renderer(pa.Params(H, W, 500.0, spp, B), seed=seed, **kw) as r:
        r.trace(spp)
        return r.read_sum_f32(), r.segments, r.read_rgb8(pa.PT_RGB8_REFERENCE), r.read_rgb8(pa.PT_RGB8_MEAN_CLAMP)


def test_one_device_group_equals_the_plain_context_bit_for_bit():
    import pathtracer_amd as pa
    want, segs, rgb_ref, rgb_mean = plain(24)
    with pa.Renderer(pa.Params(H, W, 500.0, 24, B), seed=5, devices=[0]) as g:
        info = g.comm_info()
        assert info["n_ranks"] == 1 and info["n_local"] == 1
        assert info["rccl_version"] > 0, "librccl was not reached through libpt"
        g.trace(24)
        assert g.samples_done == 24
        assert np.array_equal(g.read_sum_f32(), want)
        assert g.segments == segs
        assert np.array_equal(g.read_rgb8(pa.PT_RGB8_REFERENCE), rgb_ref)
        assert np.array_equal(g.read_rgb8(pa.PT_RGB8_MEAN_CLAMP), rgb_mean)
        assert np.array_equal(g.read_sum(), want.astype(np.float64))
        idx, t, _ = g.primary_hits()
    oidx, ot, _ = O.primary_hits(O.params(H, W, 500.0, B, O.PHILOX, 5), O.default_scene())
    assert np.array_equal(idx, oidx) and np.array_equal(t, ot)


@pytest.mark.parametrize("members", [2, 3, 5])
def test_group_renders_the_same_paths_as_one_gpu(members):
    import pathtracer_amd as pa
    spp = 24
    want, segs, _, _ = plain(spp)
    with pa.Renderer(pa.Params(H, W, 500.0, spp, B), seed=5, devices=[0] * members) as g:
        assert g.comm_info()["n_local"] == members
        g.trace(spp)
        got = g.read_sum_f32()
        # every member runs the per-pixel primary search once for itself (pt_segments counts what runs)
        assert g.segments == segs + (members - 1) * H * W
        rgb = g.read_rgb8(pa.PT_RGB8_REFERENCE)
    np.testing.assert_allclose(got, want, rtol=4e-6, atol=1e-6)
    assert np.array_equal(rgb, O.repa3_to_image(O.params(H, W, 500.0, B), got.astype(np.float64)))


def test_group_against_the_oracle_on_the_same_philox_stream():
    import pathtracer_amd as pa
    spp = 8
    with pa.Renderer(pa.Params(H, W, 500.0, spp, B), seed=9, devices=[0, 0, 0]) as g:
        g.trace(spp)
        dev = g.read_sum()
    ora = O.Tracer(O.params(H, W, 500.0, B, O.PHILOX, 9), O.default_scene(), threads=4).trace(spp)
    bad = (np.abs(dev - ora) > 2e-4 * np.abs(ora) + 1e-5).any(axis=2).mean()
    assert bad <= 0.01


def test_reads_leave_the_partial_sums_alone():
    """A read all-reduces out of place: the render goes on afterwards (app/Web.hs:42-65 pulls frame after
    frame) and nothing is counted twice."""
    import pathtracer_amd as pa
    want12, _, _, _ = plain(12)
    want5, _, _, _ = plain(5)
    with pa.Renderer(pa.Params(H, W, 500.0, 12, B), seed=5, devices=[0, 0]) as g:
        g.trace(5)
        np.testing.assert_allclose(g.read_sum_f32(), want5, rtol=4e-6, atol=1e-6)
        np.testing.assert_allclose(g.read_sum_f32(), want5, rtol=4e-6, atol=1e-6)     # a second read: the kept total
        g.trace(7)
        assert g.samples_done == 12
        np.testing.assert_allclose(g.read_sum_f32(), want12, rtol=4e-6, atol=1e-6)
        g.reset()
        assert g.samples_done == 0 and g.segments == 0
        assert not g.read_sum_f32().any()
        g.trace(1)                                      # fewer samples than members: the second member sits it out
        want1, _, _, _ = plain(1)
        np.testing.assert_allclose(g.read_sum_f32(), want1, rtol=4e-6, atol=1e-6)


def test_group_checkpoint_resume():
    import pathtracer_amd as pa
    want, _, _, _ = plain(10)
    P = pa.Params(H, W, 500.0, 10, B)
    with pa.Renderer(P, seed=5, devices=[0, 0]) as g:
        g.trace(4)
        state = g.checkpoint()
    with pa.Renderer(P, seed=5, devices=[0, 0, 0]) as g:
        g.restore(state)
        g.trace(6)
        np.testing.assert_allclose(g.read_sum_f32(), want, rtol=4e-6, atol=1e-6)


def test_group_with_the_scene_kernel():
    import pathtracer_amd as pa
    want, _, _, _ = plain(16)
    with pa.Renderer(pa.Params(H, W, 500.0, 16, B), seed=5, devices=[0, 0]) as g:
        if not g.specialise():
            pytest.skip("pt_scene_specialise not available: " + g.last_error())
        assert "hiprtc" in g.jit_info() or "hipcc at build time" in g.jit_info()
        g.trace(16)
        np.testing.assert_allclose(g.read_sum_f32(), want, rtol=4e-6, atol=1e-6)


def test_group_refuses_what_does_not_apply():
    import torch
    import pathtracer_amd as pa
    with pa.Renderer(pa.Params(32, 32, 500.0, 1, 4), devices=[0, 0]) as g:
        with pytest.raises(pa.PtError) as e:
            g.set_stream(torch.cuda.Stream().cuda_stream)
        assert e.value.status == pa.PT_ERR_UNSUPPORTED
    with pytest.raises(pa.PtError):
        pa.Renderer(pa.Params(32, 32, 500.0, 1, 4), devices=[0] * 9)        # a device at most 8 times
    with pytest.raises(pa.PtError):
        pa.Renderer(pa.Params(32, 32, 500.0, 1, 4), devices=[0, 99])


def test_rank_context_allreduce_world_of_one():
    """The process-per-GPU layout with a world of 1: pt_comm_unique_id, pt_comm_init_rank and one ncclAllReduce
    on the context's stream; the reads return the total until the next trace."""
    import pathtracer_amd as pa
    want8, _, rgb_ref, _ = plain(8)
    want11, _, _, _ = plain(11)
    with pa.Renderer(pa.Params(H, W, 500.0, 8, B), seed=5) as r:
        uid = pa.Renderer.comm_unique_id()
        assert len(uid) == 128
        r.comm_init_rank(uid, 0, 1)
        info = r.comm_info()
        assert info == {"n_ranks": 1, "rank": 0, "n_local": 1, "rccl_version": info["rccl_version"]} and info["rccl_version"] > 0
        r.trace(8)
        r.allreduce_sum()
        assert np.array_equal(r.read_sum_f32(), want8)
        assert np.array_equal(r.read_rgb8(pa.PT_RGB8_REFERENCE), rgb_ref)
        r.trace(3)                                      # the total is stale now: reads are the context's own sums
        own = r.read_sum_f32()                          # (8 + 3 samples against 11 at once: fp32 summation order)
        np.testing.assert_allclose(own, want11, rtol=4e-6, atol=1e-6)
        assert not np.array_equal(own, want8)
        r.allreduce_sum()
        assert np.array_equal(r.read_sum_f32(), own)
        with pytest.raises(pa.PtError):
            r.comm_init_rank(uid, 0, 1)                 # one communicator per context


def test_allreduce_needs_a_communicator():
    import pathtracer_amd as pa
    with pa.Renderer(pa.Params(32, 32, 500.0, 1, 4)) as r:
        with pytest.raises(pa.PtError):
            r.allreduce_sum()


def test_contexts_of_a_rank_share_one_communicator():
    """pt_comm_share: the second canvas of a rank's frame pipeline uses the first's communicator (world of 1 here);
    both reduce correctly, in either order, and the communicator outlives the context that created it."""
    import pathtracer_amd as pa
    want8, _, _, _ = plain(8)
    want5, _, _, _ = plain(5)
    a = pa.Renderer(pa.Params(H, W, 500.0, 8, B), seed=5)
    b = pa.Renderer(pa.Params(H, W, 500.0, 8, B), seed=5)
    c = pa.Renderer(pa.Params(H, W + 8, 500.0, 8, B), seed=5)
    try:
        with pytest.raises(pa.PtError):
            b.comm_share(a)                             # a has no communicator yet
        a.comm_init_rank(pa.Renderer.comm_unique_id(), 0, 1)
        b.comm_share(a)
        assert b.comm_info() == a.comm_info() and b.comm_info()["rccl_version"] > 0
        with pytest.raises(pa.PtError):
            b.comm_share(a)                             # one communicator per context
        with pytest.raises(pa.PtError):
            c.comm_share(a)                             # another canvas size
        a.trace(8); b.trace(5)
        b.allreduce_sum(); a.allreduce_sum()
        assert np.array_equal(a.read_sum_f32(), want8)
        assert np.array_equal(b.read_sum_f32(), want5)
        a.close()                                       # b keeps the communicator alive
        b.trace(3)
        b.allreduce_sum()
        np.testing.assert_allclose(b.read_sum_f32(), want8, rtol=4e-6, atol=1e-6)
    finally:
        a.close(); b.close(); c.close()
