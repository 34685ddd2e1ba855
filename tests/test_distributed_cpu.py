"""N > 1 host logic on CPU: world_size-2 (and 3) gloo process groups run the
sample split and the radiance all-reduce exactly as bench.py does on GPUs, with
the oracle standing in for the per-rank renderer (Philox stream keyed by the
GLOBAL sample index).  The all-reduced canvas must equal the single-rank one."""
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, spp, out_dir):
    sys.path[:0] = [ROOT, HERE]
    import torch
    import torch.distributed as dist
    import oracle as O
    from pathtracer_amd.distributed import allreduce_canvas, shard_samples
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    H, W, B = 24, 32, 6
    count, off, stride = shard_samples(spp, rank, world)
    tr = O.Tracer(O.params(H, W, 500.0, B, O.PHILOX, 7), O.default_scene(), threads=1)
    if count:
        tr.trace(count, s_begin=off, stride=stride)
    canvas = torch.from_numpy(tr.sum.reshape(-1).copy())
    allreduce_canvas(canvas)
    segs = torch.tensor([tr.segments.value], dtype=torch.int64)
    dist.all_reduce(segs)
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), canvas.numpy())
    if rank == 0:
        np.save(os.path.join(out_dir, "segs.npy"), segs.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,spp", [(2, 8), (2, 7), (3, 8)])
def test_sample_split_allreduce_gloo(tmp_path, world, spp):
    import torch.multiprocessing as mp
    import oracle as O
    port = _free_port()
    mp.spawn(_worker, args=(world, port, spp, str(tmp_path)), nprocs=world, join=True)
    tr = O.Tracer(O.params(24, 32, 500.0, 6, O.PHILOX, 7), O.default_scene(), threads=1)
    want = tr.trace(spp).reshape(-1)
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npy")
        assert np.allclose(got, want, rtol=1e-12, atol=1e-12)      # same paths, different summation order
    assert int(np.load(tmp_path / "segs.npy")[0]) == tr.segments.value


def test_shard_samples_partition():
    from pathtracer_amd.distributed import global_samples, shard_samples
    for total in (0, 1, 7, 8, 1024, 1023):
        for world in (1, 2, 3, 4, 8):
            all_s = sorted(s for r in range(world) for s in global_samples(total, r, world))
            assert all_s == list(range(total))
            counts = [shard_samples(total, r, world)[0] for r in range(world)]
            assert sum(counts) == total and max(counts) - min(counts) <= 1
    with pytest.raises(ValueError):
        shard_samples(8, 2, 2)


def test_device_group_block_split_renders_the_same_frame():
    """pt_create_multi's split (contiguous blocks of every pt_trace call, pt_multi.hip group_trace; mirrored by
    block_samples): the blocks of a call tile its range for any group size, and the oracle rendering each block on
    the Philox stream sums to the one-device frame -- over several calls, including calls of fewer samples than members."""
    import oracle as O
    from pathtracer_amd.distributed import block_samples
    for done, iters in ((0, 0), (0, 1), (3, 2), (0, 1024), (5, 7), (100, 1023)):
        for members in (1, 2, 3, 5, 8):
            blocks = [block_samples(done, iters, m, members) for m in range(members)]
            assert sum(c for _, c in blocks) == iters
            nxt = done
            for first, count in blocks:
                assert first == nxt and count in (iters // members, iters // members + 1)
                nxt += count
    H, W, B = 20, 28, 6
    p = O.params(H, W, 500.0, B, O.PHILOX, 3)
    whole = O.Tracer(p, O.default_scene(), threads=1)
    members = 3
    parts = [O.Tracer(p, O.default_scene(), threads=1) for _ in range(members)]
    done = 0
    for iters in (5, 1, 7, 2):                           # four pt_trace calls on the group
        whole.trace(iters)
        for m, tr in enumerate(parts):
            first, count = block_samples(done, iters, m, members)
            if count:
                tr.trace(count, s_begin=first)
        done += iters
    total = sum(tr.sum for tr in parts)
    assert np.allclose(total, whole.sum, rtol=1e-12, atol=1e-12)
    assert sum(tr.segments.value for tr in parts) == whole.segments.value
