"""View-sharded data parallelism for the rasterizer (BASELINE.json configs 4-5).

The reference never batches (one view per optimisation step, Edit_core/tetgs_texture/refine.py:54,
257-275).  Frames are independent given the (replicated) Gaussian parameters, so a batch of V views
shards by view: each rank renders its contiguous shard view after view, autograd accumulates the
per-Gaussian gradients of every view into ONE flat fp32 buffer (parameter ``.grad`` tensors are views
into it), and the step ends with ONE all-reduce(sum) of that buffer -- RCCL over xGMI on GPUs
(backend "nccl"), gloo in the CPU tests.  There is no other data-path collective.

Host logic only: nothing here touches the HIP library, so it runs on CPU with any per-view render
function (the tests inject the CPU oracle).
"""
from __future__ import annotations

from typing import Callable, Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist


def shard_views(num_views: int, rank: int, world_size: int) -> range:
    """Contiguous shard of ``range(num_views)`` for ``rank``; sizes differ by at most one."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of {world_size}")
    base, extra = divmod(num_views, world_size)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


class FlatGradients:
    """One contiguous gradient buffer for a list of leaf parameters.

    ``p.grad`` of every parameter becomes a view into ``self.flat``; autograd accumulates in place
    across the views of a shard, and ``all_reduce`` sums the whole step with a single collective
    (236 B per Gaussian on the SH path = 118 MB at 500k Gaussians)."""

    def __init__(self, params: Sequence[torch.Tensor]):
        self.params = list(params)
        if not self.params:
            raise ValueError("no parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        for p in self.params:
            if p.device != dev or p.dtype != dt or not p.is_leaf or not p.requires_grad:
                raise ValueError("parameters must be leaf tensors requiring grad, on one device, of one dtype")
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, dtype=dt, device=dev)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero_(self) -> None:
        self.flat.zero_()

    def all_reduce(self, group=None, async_op: bool = False):
        """Sum over ranks (no-op without an initialised process group or with world size 1)."""
        if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
            return None
        return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def render_batch_sharded(render_view: Callable[[int], torch.Tensor], upstream: Callable[[int, torch.Tensor], torch.Tensor],
                         num_views: int, grads: FlatGradients, group=None, rank: Optional[int] = None,
                         world_size: Optional[int] = None) -> List[int]:
    """One data-parallel step over a batch of ``num_views`` views.

    ``render_view(v)`` returns the image of view ``v`` (built on the parameters held by ``grads``);
    ``upstream(v, image)`` returns dL/d image.  After the call every rank's ``grads.flat`` holds the
    gradient of the whole batch.  Returns the views this rank rendered."""
    if rank is None:
        rank = dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0
    if world_size is None:
        world_size = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    mine = shard_views(num_views, rank, world_size)
    grads.zero_()
    for v in mine:
        img = render_view(v)
        img.backward(upstream(v, img.detach()))
    grads.all_reduce(group)
    return list(mine)
