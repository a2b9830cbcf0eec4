"""Training-step harness mirroring ``Batch3DMOT.train`` (reference train.py:124-160).

    out, _ = gnn.forward(data); out = out.squeeze(1)
    loss   = BCELoss(weight=data.edge_weights)(out, data.y.float()) / params.gnn.batch_size
    optimizer.zero_grad(); loss.backward(); optimizer.step()

``edge_loss`` is that loss as a handful of element-wise torch ops; ``fused_edge_loss`` is the same
value and its gradient from one HIP kernel (``b3d_edge_loss``), which ``train_step`` uses on the
GPU.  ``PoseGNN`` emits logits (pose_gnn.py:45-53; the release ships no trainer for it), so its step
uses the numerically equivalent BCE-with-logits.
"""
from __future__ import annotations

from typing import Optional

import collections

import torch
import torch.nn.functional as F

# which implementation every step took: {"fused_loss" | "torch_loss", "flat_adam" | "torch_optimizer"} -> steps (tests assert on it)
PATHS: "collections.Counter" = collections.Counter()


def edge_loss(out: torch.Tensor, data, batch_size: int, loss_kind: str = "cb", logits: bool = False):
    gt = data.y.float()
    out = out.squeeze(1)
    w = data.edge_weights if loss_kind == "cb" else None        # train.py:136-139
    if logits:
        loss = F.binary_cross_entropy_with_logits(out, gt, weight=w)
    else:
        loss = F.binary_cross_entropy(out, gt, weight=w)
    return loss / batch_size                                       # train.py:141


def fused_edge_loss(out: torch.Tensor, data, batch_size: int, loss_kind: str = "cb", logits: bool = False):
    """(loss, d loss / d out) of ``edge_loss`` from one kernel launch.  ``out`` [E,1] or [E] on the GPU."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    o = out.detach().reshape(-1)
    _lib.require_cuda(o, "out", torch.float32)
    y = data.y.reshape(-1)
    if y.dtype not in (torch.float32, torch.int64):
        y = y.float()
    y = y.contiguous()
    w = data.edge_weights.reshape(-1).float().contiguous() if loss_kind == "cb" else None
    n = o.numel()
    if y.numel() != n or (w is not None and w.numel() != n):
        raise ValueError(f"edge loss: out has {n} rows, y {y.numel()}, weights {None if w is None else w.numel()}")
    loss = torch.empty((), dtype=torch.float32, device=o.device)
    grad = torch.empty_like(o)
    nbytes = lib.b3d_edge_loss_workspace_bytes(n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=o.device)
    _lib.check(lib.b3d_edge_loss(o.data_ptr(), y.data_ptr(), int(y.dtype == torch.int64),
                                 w.data_ptr() if w is not None else None, n, int(bool(logits)),
                                 C.c_float(1.0 / batch_size), ws.data_ptr(), nbytes, loss.data_ptr(), grad.data_ptr(),
                                 _lib.current_stream(o.device)), "b3d_edge_loss")
    return loss, grad.view_as(out)


def forward_backward(gnn, data, optimizer, batch_size: int = 2, loss_kind: str = "cb", logits: bool = False,
                     fused_loss: Optional[bool] = None, forward_kwargs: Optional[dict] = None, after_forward=None):
    """The part of a step in front of the gradient exchange: forward, ``zero_grad``, loss, backward.  Split out so
    that a data-parallel loop can capture it (and ``optimizer.step()``) into hipGraphs and keep only the all-reduce
    eager between the two replays."""
    out, aux = gnn(data, **forward_kwargs) if forward_kwargs else gnn(data)
    if after_forward is not None:
        after_forward()                         # e.g. EncodeAhead.launch(next batch): independent work under the backward sweep
    if fused_loss is None:
        fused_loss = out.is_cuda                # GPU scores: always b3d_edge_loss; the torch form only on request (fused_loss=False)
    PATHS["fused_loss" if fused_loss else "torch_loss"] += 1
    PATHS["flat_adam" if hasattr(optimizer, "flat_grad") else "torch_optimizer"] += 1
    if hasattr(optimizer, "flat_grad"):
        optimizer.zero_grad()                   # optim.FlatAdam: lazy (the backward overwrites the flat buffer)
    else:
        optimizer.zero_grad(set_to_none=True)
    if fused_loss:
        loss, g = fused_edge_loss(out, data, batch_size, loss_kind, logits)
        out.backward(g)
    else:
        loss = edge_loss(out, data, batch_size, loss_kind, logits)
        loss.backward()
    return loss.detach(), out.detach(), aux


def train_step(gnn, data, optimizer, batch_size: int = 2, loss_kind: str = "cb", logits: bool = False,
               grad_sync: Optional[object] = None, fused_loss: Optional[bool] = None, forward_kwargs: Optional[dict] = None,
               after_forward=None):
    """One optimisation step; ``grad_sync`` (batch3dmot_amd.dist.FlatGradSync) averages gradients over
    the ranks of a data-parallel job between backward and the optimizer step.  ``fused_loss``
    (default: on when the model output lives on the GPU) takes loss and d loss/d out from
    ``b3d_edge_loss`` and seeds ``out.backward`` with it."""
    loss, out, aux = forward_backward(gnn, data, optimizer, batch_size, loss_kind, logits, fused_loss, forward_kwargs, after_forward)
    if grad_sync is not None:
        grad_sync.sync()
    optimizer.step()
    return loss, out, aux


def make_optimizer(gnn, lr: float = 1e-4, weight_decay: float = 1e-4, betas=(0.9, 0.999), flat: Optional[bool] = None,
                   capturable: bool = False):
    """Adam exactly as train.py:106-109 (``gnn.*`` keys of the YAML config).  ``flat`` (default: on
    for a GPU-resident PoseGNN / GNN whose Linear stacks are all trainable) returns
    ``optim.FlatAdam``: same update, one kernel launch, gradients written in place by the backward."""
    params = [p for p in gnn.parameters() if p.requires_grad]
    if flat is None:
        hip = list(gnn._hip_params()) if hasattr(gnn, "_hip_params") else []
        flat = bool(hip) and all(p.is_cuda and p.requires_grad for p in hip)
        if hip and not flat and any(p.is_cuda for p in hip):
            # no silent fallback: a GPU-resident HIP model whose Linear stacks are not all trainable cannot use the one-launch
            # Adam (its flat buffer IS the backward's gradient storage)
            frozen = sum(1 for p in hip if not p.requires_grad)
            raise RuntimeError(f"make_optimizer: {frozen} of the {len(hip)} parameters the HIP backward writes are frozen (or not on the "
                               "GPU), so optim.FlatAdam (b3d_adam_step) cannot own their gradients; pass flat=False to use "
                               "torch.optim.Adam deliberately")
    if flat:
        from .optim import FlatAdam
        return FlatAdam(gnn, lr=lr, weight_decay=weight_decay, betas=betas, capturable=capturable)
    return torch.optim.Adam(params, lr=lr, weight_decay=weight_decay, betas=betas)


class EncodeAhead:
    """The frozen encoders of the NEXT batch underneath the current batch's training step.

    ``GNN.forward`` runs ResNetAE / PointNet / RadarNet first and the message passing after them (clr_att_gnn.py:107-141,
    :143-188).  The encoders are frozen (clr_att_gnn.py:26-33): nothing an optimizer step changes feeds them, so batch
    k + 1 can be encoded while batch k is still in its forward / backward / Adam -- the usual prefetch, one stage deeper
    than the loader's H2D copy.  Results are those of the sequential loop bit for bit: the encoders see the batches in
    the same order (train-mode BatchNorm: running statistics; Dropout: the generator's draw order -- the GNN itself draws
    nothing), only earlier.

        ahead = EncodeAhead(gnn)
        ahead.launch(batches[0])
        for k, batch in enumerate(batches):
            encoded = ahead.take(batch)                      # joins the side stream into the current one
            if k + 1 < len(batches):
                ahead.launch(batches[k + 1])                 # runs under the step below
            train_step(gnn, batch, optimizer, forward_kwargs={"encoded": encoded})

    or, placing the parts where the step has room for them (what ``bench.py`` times: the camera encoder under the forward, the two
    point encoders under the backward sweep, whose node-phase kernels leave a quarter of the CUs idle):

            ahead.launch(nxt, parts="img")
            train_step(gnn, batch, optimizer, forward_kwargs={"encoded": encoded},
                       after_forward=lambda: ahead.launch(nxt, parts="points"))

    One side stream, the three encoders one after the other on it (they have a whole step of time).  ``static`` (a tuple of
    preallocated tensors shaped like ``encode_modalities``' result) makes ``launch`` write there -- what a hipGraph-captured
    step needs; without it the outputs are fresh tensors, recorded on the consuming stream by ``take``."""

    def __init__(self, gnn):
        self.gnn = gnn
        self.stream = None
        self.pending = None

    _SLOTS = {"img": (0,), "lidar": (1, 2), "radar": (3, 4), "points": (1, 2, 3, 4), "all": (0, 1, 2, 3, 4)}

    def launch(self, data, rows=None, static=None, parts: str = "all"):
        """``parts``: "all" (default), or several calls for the same batch at different places of the current step -- "img"
        (ResNetAE), "lidar" (PointNet), "radar" (RadarNet), "points" (= lidar + radar), in any order; ``take`` needs all of them.
        When "radar" comes in front of "lidar" with Dropout live, PointNet's mask is drawn first and parked
        (``encoders.predraw_dropout_mask``): the generator is drawn from in the reference's order whatever the launch order."""
        if parts not in self._SLOTS:
            raise ValueError(f"EncodeAhead.launch: parts = {parts!r}")
        if self.pending is not None and (self.pending[0] is not data or parts == "all" or any(k in self.pending[1] for k in self._SLOTS[parts])):
            raise RuntimeError("EncodeAhead.launch: the previous batch was never taken (or this part of the batch was launched already)")
        dev = data.pose_feats.device
        if self.stream is None or self.stream.device != dev:
            self.stream = torch.cuda.Stream(dev)
        cur = torch.cuda.current_stream(dev)
        self.stream.wait_stream(cur)
        if rows is None and parts != "img":
            rows = self.gnn.modality_rows(data)              # (on the caller's side: the counts are shapes)
        keep = self.gnn.encoder_streams
        self.gnn.encoder_streams = False                     # one branch: no forks inside the side stream
        try:
            with torch.cuda.stream(self.stream):
                st = static if static is not None else (None,) * 5       # the big outputs go straight into their static buffers
                if parts == "all":
                    out = [self.gnn._encode_img(data, out=st[0])] + list(self.gnn._encode_lidar(data, rows[0], out=st[1])) \
                        + list(self.gnn._encode_radar(data, rows[1], out=st[3]))
                elif parts == "img":
                    out = [self.gnn._encode_img(data, out=st[0])]
                elif parts == "lidar":
                    out = list(self.gnn._encode_lidar(data, rows[0], out=st[1]))
                elif parts == "radar":
                    if not (self.pending is not None and 1 in self.pending[1]) and rows[0].numel() >= 2:
                        from . import encoders
                        pn = self.gnn.pointnet            # its fc2 Dropout acts on [lidar rows, 256] (pointnet.py:190)
                        encoders.predraw_dropout_mask(pn.dropout, int(rows[0].numel()), pn.fc2.out_features, dev)
                    out = list(self.gnn._encode_radar(data, rows[1], out=st[3]))
                else:
                    out = list(self.gnn._encode_lidar(data, rows[0], out=st[1])) + list(self.gnn._encode_radar(data, rows[1], out=st[3]))
                slots = self._SLOTS[parts]
                if static is not None:
                    for k, src in zip(slots, out):
                        dst = static[k]
                        if dst.shape != src.shape:
                            raise ValueError(f"EncodeAhead: static buffer {tuple(dst.shape)} vs encoder output {tuple(src.shape)}")
                        if src.data_ptr() != dst.data_ptr():
                            dst.copy_(src)
                    out = [static[k] for k in slots]
                if rows is not None and not torch.cuda.is_current_stream_capturing():
                    for t in rows:
                        t.record_stream(self.stream)          # produced elsewhere, read by this stream's gathers
        finally:
            self.gnn.encoder_streams = keep
        have = dict(self.pending[1]) if self.pending is not None else {}
        have.update(zip(slots, out))
        self.pending = (data, have, static is not None)
        return tuple(out)

    def launch_graph(self, data, ws=None):
        """The NEXT batch's graph structure (CSR by destination + CSC by source, ``b3d_graph_build``: four integer kernels) on the
        side stream, under the current step -- SURVEY.md 8f #2 ("pre-build dst-CSR + src-CSC once"); it depends on ``edge_index``
        only.  Left in ``data._b3d_graph``, where ``GNN.forward`` looks for it; ``take`` joins the side stream.  ``ws``: a caller-owned
        ``uint8`` buffer of ``_lib.Graph.workspace_bytes(N, E)`` bytes (fixed addresses, for hipGraph-captured steps)."""
        from . import _lib
        dev = data.pose_feats.device
        if self.stream is None or self.stream.device != dev:
            self.stream = torch.cuda.Stream(dev)
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        ei = data.edge_index
        with torch.cuda.stream(self.stream):
            g = _lib.Graph(ei if ei.is_contiguous() else ei.contiguous(), data.pose_feats.size(0),
                           validated=getattr(data, "_b3d_valid_edge_index", None) is ei, ws=ws)
            if not torch.cuda.is_current_stream_capturing():
                ei.record_stream(self.stream)
        data._b3d_graph = g
        self._graph_for = data
        return g

    def launch_rows(self, data, static_rows, mismatch):
        """The NEXT batch's modality row ids on the side stream, written into ``static_rows`` with the counts checked on the device
        (``GNN.modality_rows_into``): the prologue of a hipGraph-captured step without a host read-back.  Call before the
        ``launch(..., rows=static_rows)`` calls that consume them (same stream: ordered)."""
        dev = data.pose_feats.device
        if self.stream is None or self.stream.device != dev:
            self.stream = torch.cuda.Stream(dev)
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self.stream):
            self.gnn.modality_rows_into(data, static_rows, mismatch)

    def take(self, data):
        if self.pending is None or self.pending[0] is not data:
            raise RuntimeError("EncodeAhead.take: this batch was not the one launched")
        _, have, is_static = self.pending
        if len(have) != 5:
            raise RuntimeError(f"EncodeAhead.take: only parts {sorted(have)} of this batch were launched")
        out = tuple(have[k] for k in range(5))
        self.pending = None
        cur = torch.cuda.current_stream(data.pose_feats.device)
        cur.wait_stream(self.stream)
        if not is_static and not torch.cuda.is_current_stream_capturing():
            for t in out:
                t.record_stream(cur)
        return out
