"""Training / validation metric of the reference loop (train.py:18,143-150 and :188-196): average precision of
the edge scores, on the whole batch and on the edges of every class, from ONE C-ABI call
(``b3d_average_precision``: a device radix sort that orders all the sets at once, two scans, a fixed-order
reduction; float64).  The reference calls ``torchmetrics.functional...average_precision(out, gt, pos_label=1)``
eight times per step, each a sort plus a host synchronisation.

``average_precision(preds, target, pos_label=1)`` keeps that call's shape; ``average_precision_per_class`` is the
loop over ``class_dict_used`` (train.py:145-150) in one call.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import _lib


def _run(preds: torch.Tensor, target: torch.Tensor, edge_classes: Optional[torch.Tensor], num_classes: int):
    lib = _lib.load()
    p = preds.detach().reshape(-1)
    _lib.require_cuda(p, "preds", torch.float32)
    y = target.reshape(-1)
    if y.dtype not in (torch.float32, torch.int64):
        y = y.float()
    y = y.contiguous()
    n = p.numel()
    if y.numel() != n:
        raise ValueError(f"average precision: {n} scores, {y.numel()} labels")
    ec = None
    if edge_classes is not None:
        ec = edge_classes.reshape(-1).float().contiguous()
        if ec.numel() != n:
            raise ValueError(f"average precision: {n} scores, {ec.numel()} edge classes")
    ap = torch.empty(num_classes + 1, dtype=torch.float64, device=p.device)
    cnt = torch.empty(num_classes + 1, dtype=torch.int32, device=p.device)
    nbytes = lib.b3d_average_precision_workspace_bytes(n, num_classes)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=p.device)
    _lib.check(lib.b3d_average_precision(p.contiguous().data_ptr(), y.data_ptr(), int(y.dtype == torch.int64),
                                         ec.data_ptr() if ec is not None else None, n, num_classes, ws.data_ptr(), nbytes,
                                         ap.data_ptr(), cnt.data_ptr(), _lib.current_stream(p.device)), "b3d_average_precision")
    return ap, cnt


def average_precision(preds: torch.Tensor, target: torch.Tensor, pos_label: int = 1) -> torch.Tensor:
    """Binary average precision, a 0-dim float64 tensor on the device of ``preds`` (no host synchronisation)."""
    if pos_label != 1:
        raise ValueError("the reference only ever passes pos_label=1 (train.py:143)")
    return _run(preds, target, None, 0)[0][0]


def average_precision_per_class(preds: torch.Tensor, target: torch.Tensor, edge_classes: torch.Tensor,
                                class_dict: Dict[str, int]):
    """(overall AP, {category: AP}) as train.py:143-150 logs them: a category appears only if the batch holds an edge
    of it (``torch.sum(edge_classes == cls_idx) > 0``); one host read for the whole dictionary."""
    num_classes = max(class_dict.values())
    ap, cnt = _run(preds, target, edge_classes, num_classes)
    ap_h, cnt_h = ap.cpu(), cnt.cpu()
    per_class = {c: float(ap_h[i]) for c, i in class_dict.items() if int(cnt_h[i]) > 0}
    return ap[0], per_class
