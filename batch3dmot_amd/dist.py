"""Data-parallel gradient synchronisation: one process per GPU, scene-graph batches sharded over
ranks, ONE flat all-reduce of the gradient-carrying parameters per step (RCCL over xGMI when the
backend is "nccl"; "gloo" on CPU for tests).

The reference has no distributed GNN trainer (its only DDP code trains the image encoder,
training/train_resnet_ae_ddp.py:125-172).  The payload here is tiny -- 84,157 fp32 for PoseGNN,
1.31 M for the camera+LiDAR+radar GNN -- so the exchange is latency bound:
gradients are packed into one contiguous buffer and reduced with a single collective instead of
DDP's per-bucket calls; parameters that never receive a gradient (``knn_conv``: its result is
discarded by the reference) are left out of the buffer.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


class FlatGradSync:
    """``FlatGradSync(params)`` or ``FlatGradSync(params, flat=optim.FlatAdam)``: with a FlatAdam the
    gradients of its parameters already live in one buffer and are reduced in place; everything else
    is packed into a second flat buffer."""

    def __init__(self, params: Iterable[torch.nn.Parameter], group: Optional[dist.ProcessGroup] = None, flat=None):
        self.flat_opt = flat
        skip = {id(p) for p in flat.params} if flat is not None else set()
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad and id(p) not in skip]
        self.group = group
        self._flat: Optional[torch.Tensor] = None
        # The reduction is chosen ONCE, here: RCCL ("nccl") averages inside the collective; every other backend sums
        # and the result is scaled.  (A collective that raised may have run on some ranks and not on others, so
        # nothing is retried with a different op.)
        self._avg = bool(dist.is_initialized() and dist.get_backend(group) == "nccl")

    @property
    def world_size(self) -> int:
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def sync(self, force: bool = False, force_collective: bool = False) -> None:
        """grad <- mean over ranks.  Call between backward() and optimizer.step().  ``force``: reduce the flat buffer
        even if no backward has been seen to write it (the backward ran inside a replayed hipGraph).
        ``force_collective``: issue the collective in a process group of ONE rank too (mean over one rank: the values must
        not change) -- how a 1-GPU box executes RCCL's all-reduce on the launch stream between two hipGraph replays,
        the sequence every rank of an N-GPU run goes through (tests/test_rccl_single_rank.py)."""
        if self.world_size == 1 and not (force_collective and dist.is_initialized()):
            return
        if self.flat_opt is not None and (force or not self.flat_opt.fresh):
            g = self.flat_opt.flat_grad
            if self._avg:
                dist.all_reduce(g, op=dist.ReduceOp.AVG, group=self.group)
            else:
                dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group)
                g.mul_(1.0 / self.world_size)
        live = [p for p in self.params if p.grad is not None]
        if not live:
            return
        n = sum(p.grad.numel() for p in live)
        if self._flat is None or self._flat.numel() != n or self._flat.device != live[0].grad.device:
            self._flat = torch.empty(n, dtype=torch.float32, device=live[0].grad.device)
        flat = self._flat
        views = []
        off = 0
        for p in live:
            k = p.grad.numel()
            views.append(flat[off:off + k].view_as(p.grad))
            off += k
        torch._foreach_copy_(views, [p.grad for p in live])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        flat.mul_(1.0 / self.world_size)
        torch._foreach_copy_([p.grad for p in live], views)


def shard(items: list, rank: int, world: int) -> list:
    """Round-robin shard of independent units (graph windows / scenes): rank r takes r::world."""
    return items[rank::world]
