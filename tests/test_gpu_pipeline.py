"""PipelinedFrames (pathtracer_amd/distributed.py): the double-buffered frame
loop of the multi-GPU bench.  Frame k's output runs on a side stream while frame
k+1 is traced into the other canvas; every frame must still equal the in-line
reset -> trace -> output sequence byte for byte, and the oracle's quantiser
(Conversion.repa3ToImage, src/Conversion.hs:22-28) of the same sums."""
import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("depth,gate", [(2, False), (3, False), (2, True), (3, True)])
def test_pipelined_frames_equal_inline_frames(depth, gate):
    import torch
    import pathtracer_amd as pa
    from pathtracer_amd.distributed import PipelinedFrames, canvas_tensor

    H, W, SPP, B = 160, 96, 6, 6
    P = pa.Params(H, W, 500.0, SPP, B)
    with pa.Renderer(P, seed=3) as ref:
        ref.trace(SPP)
        want = ref.read_rgb8(pa.PT_RGB8_MEAN_CLAMP)
        want_ref_mode = ref.read_rgb8(pa.PT_RGB8_REFERENCE)
        sums = ref.read_sum_f32()
    assert np.array_equal(want_ref_mode, O.repa3_to_image(O.params(H, W, 500.0, B), sums.astype(np.float64)))

    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        rs = [pa.Renderer(P, seed=3) for _ in range(depth)]      # frames in flight: canvases and trace streams
        cvs = [canvas_tensor(q, 0) for q in rs]
        rgbs = [torch.zeros(H * W * 3, dtype=torch.uint8, device="cuda") for _ in rs]
        pipe = PipelinedFrames(rs, cvs, rgbs, rank=0, spp_total=SPP, my_spp=SPP, mode=pa.PT_RGB8_MEAN_CLAMP, dist=None,
                               gate=gate)
        assert pipe.depth == depth and len(pipe.mains) == depth
        # the guard against the round-1 hang (profiles/r03_priority_probe.txt): no stream of the loop has a priority
        assert pipe.side.priority == 0 and all(m.priority == 0 for m in pipe.mains)
        seen = []
        for k in range(7):
            b = pipe.step()
            if k >= 1:                                     # read frame k-1 while frame k is in flight
                prev = (b - 1) % depth
                pipe.done[prev].synchronize()
                seen.append(rgbs[prev].cpu().numpy().reshape(W, H, 3).copy())
        pipe.drain()
        seen.append(rgbs[b].cpu().numpy().reshape(W, H, 3).copy())
        assert pipe.gate == gate                           # (pt_trace_gate is available on this device)
        for q in rs:
            q.close()
    assert len(seen) == 7
    for img in seen:
        assert np.array_equal(img, want)


@pytest.mark.parametrize("sharing", ["side_on_trace_stream_1", "side_on_trace_stream_0", "one_stream_for_everything"])
def test_pipeline_drains_when_its_streams_share_a_queue(sharing):
    """HIP maps streams onto a few in-order hardware queues, and a wait between streams becomes a barrier packet that
    blocks its whole queue; a loop whose waits formed a cycle through a shared queue would hang (round 1 saw a hang
    with a high-priority output stream).  PipelinedFrames' graph -- trace stream b waits for the output
    of frame k-2, the output stream waits for the trace (and collective) of frame k -- has no such cycle: forced
    onto shared streams, the worst mapping there is, every frame still arrives and equals the in-line frame."""
    import torch
    import pathtracer_amd as pa
    from pathtracer_amd.distributed import PipelinedFrames, canvas_tensor

    H, W, SPP, B = 160, 96, 6, 6
    P = pa.Params(H, W, 500.0, SPP, B)
    with pa.Renderer(P, seed=3) as ref:
        ref.trace(SPP)
        want = ref.read_rgb8(pa.PT_RGB8_MEAN_CLAMP)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        rs = [pa.Renderer(P, seed=3) for _ in range(2)]
        for q in rs:                                      # libpt's own collective, world of 1, as bench.py ships it
            q.comm_init_rank(pa.Renderer.comm_unique_id(), 0, 1)
        cvs = [canvas_tensor(q, 0) for q in rs]
        rgbs = [torch.zeros(H * W * 3, dtype=torch.uint8, device="cuda") for _ in rs]
        pipe = PipelinedFrames(rs, cvs, rgbs, rank=0, spp_total=SPP, my_spp=SPP, mode=pa.PT_RGB8_MEAN_CLAMP,
                               allreduce=lambda q, cv: q.allreduce_sum())
        if sharing == "one_stream_for_everything":
            pipe.mains[1] = pipe.mains[0]
            rs[1].set_stream(pipe.mains[0].cuda_stream)
        pipe.side = pipe.mains[0] if sharing != "side_on_trace_stream_1" else pipe.mains[1]
        for _ in range(7):
            pipe.step()
        pipe.drain()
        for g in rgbs:
            assert np.array_equal(g.cpu().numpy().reshape(W, H, 3), want)
        for q in rs:
            q.close()


def test_pipeline_refuses_the_null_stream():
    import torch
    import pathtracer_amd as pa
    from pathtracer_amd.distributed import PipelinedFrames
    if torch.cuda.current_stream().cuda_stream != 0:
        pytest.skip("the current stream is not the null stream here")
    P = pa.Params(32, 32, 500.0, 1, 4)
    with pa.Renderer(P) as a, pa.Renderer(P) as b:
        with pytest.raises(ValueError):
            PipelinedFrames([a, b], [None, None], [None, None], rank=0, spp_total=1, my_spp=1, mode=0, dist=None)


def test_rgb8_on_stream_needs_a_traced_canvas():
    import torch
    import pathtracer_amd as pa
    P = pa.Params(32, 32, 500.0, 1, 4)
    s = torch.cuda.Stream()
    out = torch.zeros(32 * 32 * 3, dtype=torch.uint8, device="cuda")
    with pa.Renderer(P) as r:
        r.trace(1)
        r.reset()
        with pytest.raises(pa.PtError):
            r.rgb8_device_on(out.data_ptr(), s.cuda_stream)        # nothing traced since pt_reset: lazy zero-fill pending
        r.trace(1)
        r.synchronize()
        r.rgb8_device_on(out.data_ptr(), s.cuda_stream)
        s.synchronize()
        assert np.array_equal(out.cpu().numpy().reshape(32, 32, 3), r.read_rgb8())


def test_trace_gate_opens_before_the_gating_launch_ends():
    """pt_trace_gate: a kernel queued behind the gate on another stream starts while the gating trace launch is still
    running (its last workgroup has been dispatched, its waves are finishing), not after it."""
    import torch
    import pathtracer_amd as pa
    P = pa.Params(1920, 1080, 500.0, 64, 8)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    with pa.Renderer(P, seed=1) as a, pa.Renderer(pa.Params(64, 64, 500.0, 1, 4), seed=1) as b:
        a.set_stream(sa.cuda_stream); b.set_stream(sb.cuda_stream)
        assert b.trace_gate(a)                             # nothing traced on `a` yet: no wait, still supported
        a.trace(64); a.synchronize()                       # warm-up: the fp64 primary pass, code objects
        b.trace(1); b.synchronize()
        best = None
        for _ in range(5):
            ends, started = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            t0.record(sa)
            a.trace(64)                                    # ~1 ms of trace kernel on stream sa
            ends.record(sa)
            assert b.trace_gate(a)
            started.record(sb)                             # behind the gate on stream sb
            torch.cuda.synchronize()
            lead = t0.elapsed_time(ends) - t0.elapsed_time(started)
            best = lead if best is None else max(best, lead)
        assert best > 0.02, f"the gate opened {best:.3f} ms before the launch ended"
    with pa.Renderer(pa.Params(32, 32, 500.0, 1, 4)) as r:
        with pytest.raises(pa.PtError):
            r.trace_gate(r)
