"""TEST INFRASTRUCTURE ONLY -- name-keyed deterministic weights.

Fixtures for the camera+LiDAR+radar model would otherwise carry 4.2 M parameters each.  Both
the golden generator (which fills the REFERENCE modules) and the tests (which fill the oracle /
the HIP modules) call ``seeded_fill_`` so that only inputs and outputs need to be stored.
Values depend only on (parameter name, shape, salt) and torch's CPU generator.
"""
from __future__ import annotations

import zlib

import torch


def seeded_fill_(module: torch.nn.Module, salt: int = 0, only_trainable: bool = False) -> None:
    with torch.no_grad():
        for name, t in list(module.named_parameters()) + list(module.named_buffers()):
            if not t.is_floating_point():
                continue
            if only_trainable and not getattr(t, "requires_grad", False):
                continue
            g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + salt) & 0x7FFFFFFF)
            if name.endswith("running_var"):
                v = 0.5 + torch.rand(t.shape, generator=g)
            elif name.endswith("running_mean"):
                v = torch.randn(t.shape, generator=g) * 0.1
            elif t.dim() >= 2:
                fan_in = t[0].numel() if t.dim() > 1 else t.numel()
                v = (torch.rand(t.shape, generator=g) * 2 - 1) * (1.45 / fan_in ** 0.5)
            elif name.endswith("weight"):          # norm scale
                v = 0.8 + 0.4 * torch.rand(t.shape, generator=g)
            else:                                   # biases
                v = (torch.rand(t.shape, generator=g) * 2 - 1) * 0.1
            t.copy_(v.to(t.dtype))


def grad_digest(named_grads) -> dict:
    """Compact, order-independent pin of a gradient set: per tensor (norm, projection on a seeded
    random vector, first 8 values).  ``None`` grads are recorded as ``None``."""
    out = {}
    for name, gr in named_grads.items():
        if gr is None:
            out[name] = None
            continue
        g = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
        r = torch.randn(gr.numel(), generator=g, dtype=torch.float64)
        flat = gr.detach().reshape(-1).double().cpu()
        out[name] = {"norm": flat.norm().item(), "proj": float(flat @ r),
                     "head": flat[:8].clone(), "shape": tuple(gr.shape)}
    return out
