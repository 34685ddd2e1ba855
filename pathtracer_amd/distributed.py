"""Multi-GPU sample split: one process per GPU, each renders a
disjoint subset of the frame's global sample indices, then ONE all-reduce (sum)
of the fp32 radiance canvas over RCCL/xGMI combines them.

Rank g of G renders global samples {g, g+G, g+2G, ...}; Philox is keyed by the
global sample index, so the union over ranks is the same set of paths for any
G, and the result is independent of G up to fp32 summation order.
"""
from typing import Tuple


def shard_samples(total_spp: int, rank: int, world: int) -> Tuple[int, int, int]:
    """-> (count, sample_offset, sample_stride) for this rank: ranks below
    total % world take one extra sample, so counts sum to total_spp."""
    if world < 1 or not 0 <= rank < world or total_spp < 0:
        raise ValueError("need world >= 1, 0 <= rank < world, total_spp >= 0")
    count = total_spp // world + (1 if rank < total_spp % world else 0)
    return count, rank, world


def global_samples(total_spp: int, rank: int, world: int):
    count, off, stride = shard_samples(total_spp, rank, world)
    return [off + k * stride for k in range(count)]


def block_samples(done: int, iters: int, member: int, members: int) -> Tuple[int, int]:
    """-> (first sample index k, count) of `member`'s share of one pt_trace(iters) call on a pt_create_multi
    context whose group has rendered `done` samples so far: contiguous blocks, sizes differing by at most one,
    their union the call's range [done, done + iters) whatever the number of members.  The host-side mirror of
    group_trace (pathtracer_amd/csrc/pt_multi.hip), for tests and for hosts that schedule devices themselves."""
    if members < 1 or not 0 <= member < members or iters < 0 or done < 0:
        raise ValueError("need members >= 1, 0 <= member < members, iters >= 0, done >= 0")
    base, rem = divmod(iters, members)
    first = done + member * base + min(member, rem)
    return first, base + (1 if member < rem else 0)


def allreduce_canvas(canvas, group=None):
    """In-place sum of the radiance canvas over all ranks (torch tensor on the
    rank's GPU for the nccl(=RCCL) backend, CPU tensor under gloo)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(canvas, op=dist.ReduceOp.SUM, group=group)
    return canvas


def canvas_tensor(renderer, device_index: int):
    """The renderer's device canvas as a flat fp32 torch tensor, no copy."""
    import torch
    ptr, nbytes = renderer.canvas_device_ptr()

    class _Canvas:
        __cuda_array_interface__ = {"shape": (nbytes // 4,), "typestr": "<f4", "data": (ptr, False),
                                    "version": 2, "strides": None}
    return torch.as_tensor(_Canvas(), device=torch.device("cuda", device_index))


def stream_on_another_queue(first, tries=8, spin_cycles=2_000_000, also=()):
    """A new torch stream whose work runs CONCURRENTLY with `first`'s.  HIP maps
    streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default) and two
    streams that share one execute in order, however independent their work:
    measured here, the two trace streams of the frame pipeline landed on one
    queue in a single-process run and on two under RCCL, 14.4 against 13.4 ms
    per 1080p/1024-spp frame.  No API tells which queue a stream got, so this
    probes: a ~1-ms spin on both streams takes one spin's time if they overlap
    and two if they do not; streams that do not are kept alive (so the runtime
    hands out another queue next time) and returned for the caller to hold.
    `also`: further streams the new one must run concurrently with (a third
    trace stream against the first two).  Falls back to the last stream tried."""
    import torch
    spare = []
    s = None
    if not hasattr(torch.cuda, "_sleep"):               # no spin kernel to probe with: take what the runtime gives
        return torch.cuda.Stream(), spare
    def concurrent(a, s):
        t = []
        for both in (False, False, True):               # the first pass warms up (first use of a stream is slow)
            torch.cuda.synchronize()
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record(a)
            s.wait_event(e0)
            with torch.cuda.stream(a):
                torch.cuda._sleep(spin_cycles)
                e1.record(a)
            if both:
                with torch.cuda.stream(s):
                    torch.cuda._sleep(spin_cycles)
            e2.record(s)
            torch.cuda.synchronize()
            t.append(max(e0.elapsed_time(e1), e0.elapsed_time(e2)))
        return t[2] < 1.5 * t[1]

    for _ in range(tries):
        s = torch.cuda.Stream()
        if all(concurrent(a, s) for a in (first, *also)):
            return s, spare
        spare.append(s)
    return s, spare


class PipelinedFrames:
    """Frame pipeline for the multi-GPU bench and for progressive serving
    (the pull-next-frame contract of app/Web.hs:42-65): frame k is traced into
    canvas k % 2 while frame k-1's all-reduce (RCCL's own stream) and RGB8
    output (a side stream) are still in flight, so the collective's time is
    hidden behind the next frame's kernel instead of added to it, and the two
    frames' trace kernels run on two streams so that one's tail overlaps the
    other's head.  With three renderers the frames rotate over three canvases
    and three trace streams: a collective that is slow to get CUs next to a
    trace kernel (it sits on its frame's trace stream, in front of that
    stream's next frame) then delays nothing until it takes two frames' time.
    Every frame
    is still one pt_reset + pt_trace + ONE all-reduce + the output kernel; only
    their placement on streams changes.  `drain()` waits for everything.

    renderers: two (or three) contexts on this rank's GPU with identical
    parameters and scene; canvases: their device canvases as torch tensors
    (canvas_tensor); rgb: as many device uint8 buffers (rank 0 only, else None).
    """

    def __init__(self, renderers, canvases, rgb, *, rank, spp_total, my_spp, mode, dist=None, allreduce=None, gate=False):
        """allreduce(renderer, canvas): queues the frame's ONE collective after the trace on the current
        stream and returns None (libpt's own ncclAllReduce, Renderer.allreduce_sum: it runs on the context's
        stream and the output kernel then reads the context's total) or a work handle to wait on
        (torch.distributed with async_op=True, in place on the canvas).  `dist` alone = the torch form."""
        import torch
        self.torch = torch
        self.r, self.cv, self.rgb = renderers, canvases, rgb
        self.rank, self.spp_total, self.my_spp, self.mode, self.dist = rank, spp_total, my_spp, mode, dist
        if allreduce is None and dist is not None:
            def allreduce(_r, cv):
                return dist.all_reduce(cv, op=dist.ReduceOp.SUM, async_op=True)
        self.allreduce = allreduce
        if torch.cuda.current_stream().cuda_stream == 0:
            raise ValueError("PipelinedFrames needs an explicit torch stream as the current stream: libpt cannot "
                             "be pointed at the legacy null stream (pt_set_stream(NULL) = the context's own stream)")
        # One trace stream per canvas: frames k and k+1 are independent, so the head of
        # frame k+1's kernel fills the CUs that the tail of frame k's kernel leaves idle.
        self.depth = len(renderers)
        self.mains, self._spare = [torch.cuda.current_stream()], []
        while len(self.mains) < self.depth:
            st, spare = stream_on_another_queue(self.mains[0], also=self.mains[1:])
            self.mains.append(st)
            self._spare += spare
        # The output stream has the DEFAULT priority, on purpose.  On a high-priority stream (hipStreamCreateWithPriority)
        # this same loop can hang in drain(): reproduced in round 2 with and without a collective, depending only on
        # which streams the process had created before (profiles/r02_priority_probe.txt).  The
        # stream/event graph below has no cycle -- it drains with its three streams on three hardware queues and
        # with all of them forced onto one (tests/test_gpu_pipeline.py) -- so the wait that never ends is inside the
        # runtime's handling of waits between queues of different priority, which this loop therefore never creates.
        self.side = torch.cuda.Stream()
        for r, m in zip(renderers, self.mains):
            r.set_stream(m.cuda_stream)
        self.done = [None] * self.depth   # event: canvas k % depth is free again
        self.k = 0
        # gate: frame k+1's trace starts when frame k's trace launch has dispatched its last workgroup (pt_trace_gate),
        # i.e. exactly when wave slots begin to fall free, instead of whenever its canvas is free.  Measured
        # (profiles/r03_frame_pipeline.txt): consecutive trace kernels then overlap by 380 instead of 160-190 us, and the
        # step takes the same time -- the frame that waits is the one after, behind the starved collective.  Off by default.
        self.gate = gate

    def step(self):
        torch = self.torch
        b = self.k % self.depth
        r, cv, main = self.r[b], self.cv[b], self.mains[b]
        with torch.cuda.stream(main):
            if self.done[b] is not None:
                main.wait_event(self.done[b])                # frame k-depth's output has left this canvas
            if self.gate and self.k > 0:
                self.gate = r.trace_gate(self.r[(self.k - 1) % self.depth])
            r.reset()
            r.trace(self.my_spp)
            work = None
            if self.allreduce is not None:                   # ordered after the trace on `main` (torch: RCCL's stream
                work = self.allreduce(r, cv)                 # waits on `main` at this call)
            traced = torch.cuda.Event()
            traced.record(main)
        with torch.cuda.stream(self.side):
            self.side.wait_event(traced)
            if work is not None:
                work.wait()                                   # the side stream waits for the collective
            if self.rank == 0:
                r.rgb8_device_on(self.rgb[b].data_ptr(), self.side.cuda_stream, self.mode, self.spp_total)
            ev = torch.cuda.Event()
            ev.record(self.side)
        self.done[b] = ev
        self.k += 1
        return b

    def drain(self):
        self.side.synchronize()
        for m in self.mains:
            m.synchronize()
