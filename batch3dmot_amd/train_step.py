"""Training-step harness mirroring ``Batch3DMOT.train`` (reference train.py:124-160).

    out, _ = gnn.forward(data); out = out.squeeze(1)
    loss   = BCELoss(weight=data.edge_weights)(out, data.y.float()) / params.gnn.batch_size
    optimizer.zero_grad(); loss.backward(); optimizer.step()

The loss itself is a handful of element-wise torch ops on an [E] vector (the fused loss kernel is
a "next" row of SURVEY.md section 8f).  ``PoseGNN`` emits logits (pose_gnn.py:45-53; the release
ships no trainer for it), so its step uses the numerically equivalent BCE-with-logits.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F


def edge_loss(out: torch.Tensor, data, batch_size: int, loss_kind: str = "cb", logits: bool = False):
    gt = data.y.float()
    out = out.squeeze(1)
    w = data.edge_weights if loss_kind == "cb" else None        # train.py:136-139
    if logits:
        loss = F.binary_cross_entropy_with_logits(out, gt, weight=w)
    else:
        loss = F.binary_cross_entropy(out, gt, weight=w)
    return loss / batch_size                                       # train.py:141


def train_step(gnn, data, optimizer, batch_size: int = 2, loss_kind: str = "cb", logits: bool = False,
               grad_sync: Optional[object] = None):
    """One optimisation step; ``grad_sync`` (batch3dmot_amd.dist.FlatGradSync) averages gradients over
    the ranks of a data-parallel job between backward and the optimizer step."""
    out, aux = gnn(data)
    loss = edge_loss(out, data, batch_size, loss_kind, logits)
    optimizer.zero_grad(set_to_none=True)
    loss.backward()
    if grad_sync is not None:
        grad_sync.sync()
    optimizer.step()
    return loss.detach(), out.detach(), aux


def make_optimizer(gnn, lr: float = 1e-4, weight_decay: float = 1e-4, betas=(0.9, 0.999)):
    """Adam exactly as train.py:106-109 (``gnn.*`` keys of the YAML config)."""
    params = [p for p in gnn.parameters() if p.requires_grad]
    return torch.optim.Adam(params, lr=lr, weight_decay=weight_decay, betas=betas)
