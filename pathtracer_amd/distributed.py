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
