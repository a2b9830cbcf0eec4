"""Optimizer of the training loop (reference train.py:106-109, 157-160) on one contiguous buffer.

``FlatAdam(gnn, lr, betas, eps, weight_decay)`` is ``torch.optim.Adam`` for the parameters the HIP
backward produces gradients for (``PoseGNN`` / ``GNN``'s own Linear stacks):

* their storage is re-pointed into ONE flat fp32 buffer (``state_dict`` keys, shapes and values are
  unchanged -- every parameter becomes a view);
* ``b3d_pose_backward`` / ``b3d_clr_backward`` write their gradients straight into a second flat
  buffer whose views are the parameters' ``.grad`` (no per-parameter allocation, no AccumulateGrad
  kernels); a second backward before ``zero_grad`` accumulates, as autograd would;
* ``step()`` is one ``b3d_adam_step`` launch; ``zero_grad()`` is a flag;
* data-parallel jobs all-reduce the flat gradient buffer in place (``dist.FlatGradSync``).

Every other trainable parameter of the module (the sensor encoders of the camera+LiDAR+radar GNN,
``knn_conv``) stays with an inner ``torch.optim.Adam`` of the same hyper-parameters, so parameters
that never receive a gradient are skipped exactly as torch skips them.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import torch

from . import _lib


class FlatAdam:
    def __init__(self, gnn: torch.nn.Module, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, capturable: bool = False):
        if not hasattr(gnn, "_hip_params"):
            raise TypeError("FlatAdam needs a batch3dmot_amd PoseGNN / GNN (module with _hip_params())")
        hip: List[torch.nn.Parameter] = list(gnn._hip_params())
        if not hip or not all(p.requires_grad for p in hip):
            raise ValueError("FlatAdam: every Linear stack of the GNN must be trainable (requires_grad)")
        dev = hip[0].device
        if dev.type != "cuda":
            raise ValueError("FlatAdam: move the module to the GPU first (the HIP path has no CPU fallback)")
        self.gnn = gnn
        self.params = hip
        self.defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)
        self.param_groups = [dict(self.defaults, params=hip)]          # lr schedulers read/write 'lr' here
        n = sum(p.numel() for p in hip)
        self.numel = n
        self.flat_param = torch.empty(n, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self._scratch: Optional[torch.Tensor] = None
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.step_count = 0
        # capturable: the step counter lives on the device (b3d_adam_step_dev), so a step captured into a hipGraph
        # (torch.cuda.graph) replays correctly; step_count then only counts eager calls
        self.capturable = capturable
        self.step_dev = torch.zeros((), dtype=torch.int64, device=dev) if capturable else None
        self.grad_views: List[torch.Tensor] = []
        off = 0
        with torch.no_grad():
            for p in hip:
                k = p.numel()
                view = self.flat_param[off:off + k].view_as(p)
                view.copy_(p)
                p.data = view
                gv = self.flat_grad[off:off + k].view_as(p)
                self.grad_views.append(gv)
                off += k
        self.fresh = True               # flat_grad holds nothing that must be kept
        self._has_grad = False
        hip_ids = {id(p) for p in hip}
        rest = [p for p in gnn.parameters() if p.requires_grad and id(p) not in hip_ids]
        self.rest = rest
        self.inner = torch.optim.Adam(rest, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay) if rest else None
        gnn._grad_sink = self

    # ---- protocol used by the autograd Functions ---------------------------------------------------
    def targets(self) -> List[torch.Tensor]:
        """Tensors the backward kernel must OVERWRITE with this pass's gradients."""
        if self.fresh:
            return self.grad_views
        if self._scratch is None:
            self._scratch = torch.empty_like(self.flat_grad)
        out, off = [], 0
        for p in self.params:
            k = p.numel()
            out.append(self._scratch[off:off + k].view_as(p))
            off += k
        return out

    def deposited(self) -> None:
        if not self.fresh:
            self.flat_grad.add_(self._scratch)
        self.fresh = False
        if not self._has_grad:
            for p, gv in zip(self.params, self.grad_views):
                p.grad = gv
            self._has_grad = True

    # ---- torch.optim.Optimizer surface the training loop uses --------------------------------------
    def zero_grad(self, set_to_none: Optional[bool] = None) -> None:
        """Default (``None``): lazy -- the flat gradient buffer is marked overwritable and the next
        backward overwrites it; the ``.grad`` views stay attached (holding the previous values until
        then).  ``True`` detaches the views (``p.grad is None``), ``False`` zero-fills the buffer."""
        self.fresh = True
        if self._has_grad and set_to_none is not None:
            if set_to_none:
                for p in self.params:
                    p.grad = None
                self._has_grad = False
            else:
                self.flat_grad.zero_()
        if self.inner is not None:
            self.inner.zero_grad(set_to_none=True if set_to_none is None else set_to_none)

    @torch.no_grad()
    def step(self) -> None:
        if not self.fresh:               # a backward deposited gradients since the last zero_grad
            g = self.param_groups[0]
            self.step_count += 1
            lib = _lib.load()
            if self.capturable:
                _lib.check(lib.b3d_adam_step_dev(self.flat_param.data_ptr(), self.flat_grad.data_ptr(),
                                                 self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self.numel,
                                                 C.c_float(g["lr"]), C.c_float(g["betas"][0]), C.c_float(g["betas"][1]),
                                                 C.c_float(g["eps"]), C.c_float(g["weight_decay"]), self.step_dev.data_ptr(),
                                                 _lib.current_stream(self.flat_param.device)), "b3d_adam_step_dev")
            else:
                _lib.check(lib.b3d_adam_step(self.flat_param.data_ptr(), self.flat_grad.data_ptr(), self.exp_avg.data_ptr(),
                                             self.exp_avg_sq.data_ptr(), self.numel, C.c_float(g["lr"]),
                                             C.c_float(g["betas"][0]), C.c_float(g["betas"][1]), C.c_float(g["eps"]),
                                             C.c_float(g["weight_decay"]), self.step_count,
                                             _lib.current_stream(self.flat_param.device)), "b3d_adam_step")
        if self.inner is not None and any(p.grad is not None for p in self.rest):
            for grp in self.inner.param_groups:
                grp["lr"] = self.param_groups[0]["lr"]
            self.inner.step()

    def state_dict(self) -> dict:
        # capturable: the counter that the bias corrections read lives on the device and advances on graph replays
        step = int(self.step_dev.item()) if self.capturable else self.step_count
        return {"step": step, "exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(),
                "param_groups": [{k: v for k, v in self.param_groups[0].items() if k != "params"}],
                "inner": self.inner.state_dict() if self.inner is not None else None}

    def load_state_dict(self, sd: dict) -> None:
        self.step_count = int(sd["step"])
        if self.capturable:
            self.step_dev.fill_(self.step_count)
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.param_groups[0].update(sd["param_groups"][0])
        if self.inner is not None and sd.get("inner") is not None:
            self.inner.load_state_dict(sd["inner"])
