"""Frozen per-detection modality encoders (ADJACENT to the hot path).

``GNN.__init__`` receives three encoder modules and freezes them (reference
clr_att_gnn.py:17-33); the hot path only calls ``img_encoder.encode``,
``lidar_encoder.forward_feat`` and ``radar_encoder.forward_feat`` (clr_att_gnn.py:125,131,139).
For GPU inputs all three run in the HIP kernels of ``libb3d_hip.so``, in eval mode (inference) AND in the
frozen train mode a training step holds them in (BatchNorm on batch statistics, running statistics
updated, Dropout live):

* point-wise conv-BN-ReLU stacks + max-pool of PointNet (STN and trunk) and RadarNet: ``b3d_point_feat`` /
  ``b3d_point_feat_stats`` + the BatchNorm bookkeeping launches (csrc/b3d_encoders.hip);
* their fully connected heads: one ``b3d_fc_bn_forward`` launch per Linear (csrc/b3d_fc.hip);
* ``ResNetAE.encode``: six implicit-GEMM phase kernels (``b3d_resnet_encode``, csrc/b3d_resnet.hip).

**No silent fallback.**  The PyTorch modules below exist for CPU tensors (checkpoint handling, the restatement
the tests compare with) and for callers that opt out explicitly (``module.use_hip = False``, e.g. to train an
encoder itself: the HIP path has no autograd).  A GPU input that the HIP path cannot take -- autograd through
unfrozen encoder parameters, a crop that is not 3x32x32, a single row in train mode -- raises ``RuntimeError``
with the reason instead of quietly running MIOpen / rocBLAS; ``path_counts()`` reports which branch every call
took (the GPU tests assert on it).

The classes restate the three architectures with the reference's parameter names so that reference checkpoints
load (``resnet.*``, ``pointnet.*``, ``radarnet.*`` keys):

* ``ResNetAE.encode``            models/resnet_fully_conv.py:56-161   3x32x32 -> 96
* ``PointNetClassifier.forward_feat`` models/pointnet.py:9-57,111-192  3x128 -> 256
* ``RadarNetClassifier.forward_feat`` models/radarnet.py:9-64          4x64 -> 256

Only the sub-modules whose weights a checkpoint holds are declared; decoder halves that the
hot path never calls are declared too (unused) so ``load_state_dict(strict=True)`` works.
"""
from __future__ import annotations

import collections

import torch
from torch import nn
import torch.nn.functional as F

# (stage, "hip" | "torch") -> calls since the last reset; see path_counts()
_PATHS: "collections.Counter" = collections.Counter()


def path_counts(reset: bool = False) -> dict:
    """{(stage, "hip" | "torch"): calls} of every encoder stage evaluated since the last reset."""
    out = dict(_PATHS)
    if reset:
        _PATHS.clear()
    return out


def _route(module: nn.Module, stage: str, x: torch.Tensor, blocker) -> bool:
    """True: the HIP kernels take this call.  False: the PyTorch modules do -- only for CPU tensors or an explicit
    ``module.use_hip = False``.  A GPU input the HIP path cannot take (``blocker``: the reason, or None) raises."""
    if not x.is_cuda or not getattr(module, "use_hip", True):
        _PATHS[(stage, "torch")] += 1
        return False
    if blocker is not None:
        raise RuntimeError(f"{type(module).__name__} ({stage}): {blocker}.  The HIP path cannot take this call and there is no "
                           "silent fallback; set `module.use_hip = False` on the encoder to run the PyTorch modules "
                           "(MIOpen / rocBLAS) deliberately.")
    _PATHS[(stage, "hip")] += 1
    return True


def _autograd_blocker(module: nn.Module, x: torch.Tensor):
    if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in module.parameters())):
        return "autograd is on and the input or a parameter requires a gradient (the HIP encoders are forward-only: the GNN holds them frozen, clr_att_gnn.py:26-33)"
    return None


def _fold(layer: nn.Module, bn: nn.Module):
    """(W', b') of eval-mode ``bn(layer(x))`` for a Linear / ConvNd ``layer``: BatchNorm with running statistics is a
    per-channel affine map.  Folding is not only fewer launches: MIOpen's inference BatchNorm takes ~0.5 ms per call
    at these shapes on MI355X (90 % of ``ResNetAE.encode`` before folding: 4.9 ms -> see tools/bench_resnet_encode.py).
    Cached on the BatchNorm module, keyed by the versions of everything it depends on."""
    srcs = (layer.weight, layer.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var)
    key = tuple((t.data_ptr(), t._version) for t in srcs)
    hit = getattr(bn, "_b3d_folded", None)
    if hit is not None and hit[0] == key:
        return hit[1], hit[2]
    with torch.no_grad():
        scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        w = (layer.weight * scale.view(-1, *([1] * (layer.weight.dim() - 1)))).contiguous()
        b = ((layer.bias - bn.running_mean) * scale + bn.bias).contiguous()
    bn._b3d_folded = (key, w, b)
    return w, b


def _fold_on(bn: nn.Module) -> bool:
    """Folding applies in eval mode unless the module opts out (``bn.fold_bn = False``: the CPU oracle does, it keeps
    the reference's operation order)."""
    return (not bn.training) and getattr(bn, "fold_bn", True)


def _conv_bn(conv: nn.Conv2d, bn: nn.BatchNorm2d, x):
    if not _fold_on(bn):
        return bn(conv(x))
    w, b = _fold(conv, bn)
    return F.conv2d(x, w, b, conv.stride, conv.padding)


def _fc_bn(fc: nn.Linear, bn: nn.BatchNorm1d, x):
    if not _fold_on(bn):
        return bn(fc(x))
    w, b = _fold(fc, bn)
    return F.linear(x, w, b)


def _fold_bn(conv: nn.Conv1d, bn: nn.BatchNorm1d):
    """Eval-mode BatchNorm folded into the kernel-1 convolution in front of it: (W', b') with
    bn(conv(x)) = W' x + b'."""
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    w = (conv.weight.squeeze(-1) * scale[:, None]).float().contiguous()
    b = ((conv.bias - bn.running_mean) * scale + bn.bias).float().contiguous()
    return w, b


def point_feat_hip(convs, bns, x: torch.Tensor, trans=None, relu_last: bool = False) -> torch.Tensor:
    """conv-BN-ReLU x2, conv-BN(-ReLU), max over the points: one HIP launch (``b3d_point_feat``), eval mode only.
    ``x`` [B, C, P] on the GPU; returns [B, 1024]."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    if any(bn.training for bn in bns):
        raise RuntimeError("point_feat_hip folds BatchNorm running statistics: eval mode only")
    x = x.float().contiguous()
    _lib.require_cuda(x, "point cloud", torch.float32)
    b, c, p = x.shape
    with torch.no_grad():
        folded = [_fold_bn(cv, bn) for cv, bn in zip(convs, bns)]
    layers = (_lib.b3d_linear * 3)()
    for i, (w, bias) in enumerate(folded):
        layers[i].w, layers[i].b = w.data_ptr(), bias.data_ptr()
    out = torch.empty(b, 1024, dtype=torch.float32, device=x.device)
    nbytes = lib.b3d_point_feat_workspace_bytes()
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    t = trans.float().contiguous() if trans is not None else None
    _lib.check(lib.b3d_point_feat(layers, x.data_ptr(), t.data_ptr() if t is not None else None, b, c, p, int(relu_last),
                                  ws.data_ptr(), nbytes, out.data_ptr(), _lib.current_stream(x.device)), "b3d_point_feat")
    return out


def point_feat_train_hip(convs, bns, x: torch.Tensor, trans=None, relu_last: bool = False) -> torch.Tensor:
    """Train-mode (batch-statistics BatchNorm) form of ``point_feat_hip``, all HIP.  The batch statistics of the two
    narrow layers follow from the moments of their inputs (``b3d_point_moments`` + ``b3d_bn_fold_moments``); the
    128 -> 1024 layer runs in ``b3d_point_feat_stats``, which never stores its [points, 1024] output: per cloud it
    returns max / min / sum / sum of squares per feature, enough for the batch statistics and for
    max_p BN(z) = scale * (max_p z if scale > 0 else min_p z) + shift.  Running statistics are updated as
    ``nn.BatchNorm1d`` would."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    x = x.float().contiguous()
    _lib.require_cuda(x, "point cloud", torch.float32)
    b, c, p = x.shape
    n = b * p
    with torch.no_grad():
        folded = []
        # Batch statistics of the two narrow layers WITHOUT materialising their [B, C, P] pre-activations: the
        # pre-activation of a kernel-1 convolution is affine in its input, z = W h + b, so over all B * P points
        # mean(z) = W mean(h) + b,   var(z)_c = W_c Cov(h) W_c^T   with the (small) second-moment matrix of the INPUT:
        # 3x3 / 4x4 for the first layer, 64x64 for the second (b3d_point_moments, accumulated on the matrix cores).
        def tracked(bn):                                      # (running_mean, running_var, num_batches_tracked, momentum)
            if not bn.track_running_stats or bn.running_mean is None:
                return None, None, None, 0.0
            for t_ in (bn.running_mean, bn.running_var):
                _lib.require_cuda(t_, "BatchNorm running statistic", torch.float32)
            _lib.require_cuda(bn.num_batches_tracked, "num_batches_tracked", torch.int64)
            return (bn.running_mean.data_ptr(), bn.running_var.data_ptr(), bn.num_batches_tracked.data_ptr(),
                    float(bn.momentum) if bn.momentum is not None else -1.0)

        stream = _lib.current_stream(x.device)
        t = trans.float().contiguous() if trans is not None else None
        tp = t.data_ptr() if t is not None else None
        nbm = lib.b3d_point_moments_workspace_bytes()
        wsm = torch.empty(nbm, dtype=torch.uint8, device=x.device)
        keep = []
        prev = None
        for li, (cv, bn) in enumerate(zip(convs[:2], bns[:2])):
            w2d = cv.weight.squeeze(-1).float().contiguous()
            bias, gamma, beta = cv.bias.float().contiguous(), bn.weight.float().contiguous(), bn.bias.float().contiguous()
            o, k = w2d.shape
            mu = torch.empty(k, dtype=torch.float64, device=x.device)
            second = torch.empty(k, k, dtype=torch.float64, device=x.device)
            fold1 = None
            if prev is not None:
                fold1 = _lib.b3d_linear()
                fold1.w, fold1.b = prev[0].data_ptr(), prev[1].data_ptr()
            _lib.check(lib.b3d_point_moments(C.byref(fold1) if fold1 is not None else None, x.data_ptr(), tp, b, c, p,
                                             wsm.data_ptr(), nbm, mu.data_ptr(), second.data_ptr(), stream), "b3d_point_moments")
            wf = torch.empty_like(w2d)
            bf = torch.empty(o, dtype=torch.float32, device=x.device)
            rm, rv, nbt, mom = tracked(bn)
            # mean / variance of the pre-activation, the running-statistics update and the fold: one launch
            _lib.check(lib.b3d_bn_fold_moments(mu.data_ptr(), second.data_ptr(), k, w2d.data_ptr(), bias.data_ptr(), o,
                                               gamma.data_ptr(), beta.data_ptr(), rm, rv, nbt, mom, float(bn.eps), n,
                                               wf.data_ptr(), bf.data_ptr(), stream), "b3d_bn_fold_moments")
            keep.append((w2d, bias, gamma, beta, mu, second))
            prev = (wf, bf)
            folded.append(prev)
        folded.append((convs[2].weight.squeeze(-1).float().contiguous(), convs[2].bias.float().contiguous()))
        layers = (_lib.b3d_linear * 3)()
        for i, (w, bias) in enumerate(folded):
            layers[i].w, layers[i].b = w.data_ptr(), bias.data_ptr()
        outs = torch.empty(4, b, 1024, dtype=torch.float32, device=x.device)
        nbytes = lib.b3d_point_feat_workspace_bytes()
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        _lib.check(lib.b3d_point_feat_stats(layers, x.data_ptr(), tp, b, c, p,
                                            ws.data_ptr(), nbytes, outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(),
                                            outs[3].data_ptr(), stream), "b3d_point_feat_stats")
        # batch statistics of the last layer, its running statistics and y = BN(max | min) (+ ReLU): two launches
        bn3 = bns[2]
        gamma, beta = bn3.weight.float().contiguous(), bn3.bias.float().contiguous()
        rm, rv, nbt, mom = tracked(bn3)
        y = torch.empty(b, 1024, dtype=torch.float32, device=x.device)
        nb2 = lib.b3d_bn_minmax_workspace_bytes(1024)
        ws2 = torch.empty(nb2, dtype=torch.uint8, device=x.device)
        _lib.check(lib.b3d_bn_minmax_apply(outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), outs[3].data_ptr(), b, 1024, n,
                                           gamma.data_ptr(), beta.data_ptr(), rm, rv, nbt, mom, float(bn3.eps), int(relu_last),
                                           ws2.data_ptr(), nb2, y.data_ptr(), stream), "b3d_bn_minmax_apply")
        return y


def _tickets(owner: nn.Module, dev) -> torch.Tensor:
    """64 zero bytes of arrival counters per encoder module (``b3d_point_stack_train`` leaves them zero): created once per
    (module, device), never resized -- hipGraphs captured earlier keep writing to the same words on replay."""
    t = owner.__dict__.get("_b3d_tickets")
    if t is None or t.device != dev:
        t = torch.zeros(16, dtype=torch.int32, device=dev)
        owner.__dict__["_b3d_tickets"] = t
    return t


def point_stack_train_hip(owner: nn.Module, convs, bns, x: torch.Tensor, trans=None):
    """The train-mode point stack (conv-BN-ReLU x2, conv-BN, max over the points; BatchNorm on THIS batch's statistics, running
    statistics updated) in ONE library call (``b3d_point_stack_train``, round 5: seven launches instead of ten composed here).
    Returns ``(ext [B, 1024], scale [1024], shift [1024])``: the stack's output is ``ext * scale + shift`` (+ ReLU for the STN), an
    affine map the next Linear applies while it stages its input (``fc_head_hip(in_affine=...)``) or ``_materialize`` writes out."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    x = x.float().contiguous()
    _lib.require_cuda(x, "point cloud", torch.float32)
    b, c, p = x.shape
    dev = x.device
    keep = []

    def f32(t_):
        t_ = t_.detach().float().contiguous()
        _lib.require_cuda(t_, "point stack parameter", torch.float32)
        keep.append(t_)
        return t_.data_ptr()

    cl = (_lib.b3d_linear * 3)()
    bl = (_lib.b3d_batchnorm * 3)()
    for i, (cv, bn) in enumerate(zip(convs, bns)):
        cl[i].w, cl[i].b = f32(cv.weight.squeeze(-1)), f32(cv.bias)
        bl[i].gamma, bl[i].beta = f32(bn.weight), f32(bn.bias)
        if bn.track_running_stats and bn.running_mean is not None:
            for t_ in (bn.running_mean, bn.running_var):
                _lib.require_cuda(t_, "BatchNorm running statistic", torch.float32)
            _lib.require_cuda(bn.num_batches_tracked, "num_batches_tracked", torch.int64)
            bl[i].running_mean, bl[i].running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
            bl[i].num_batches_tracked = bn.num_batches_tracked.data_ptr()
        bl[i].momentum = float(bn.momentum) if bn.momentum is not None else -1.0
        bl[i].eps = float(bn.eps)
    t = trans.float().contiguous() if trans is not None else None
    with torch.no_grad():
        ext = torch.empty(b, 1024, dtype=torch.float32, device=dev)
        aff = torch.empty(2, 1024, dtype=torch.float32, device=dev)
        nbytes = lib.b3d_point_stack_train_workspace_bytes(b)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _lib.check(lib.b3d_point_stack_train(cl, bl, x.data_ptr(), _lib.ptr(t), b, c, p, _tickets(owner, dev).data_ptr(), ws.data_ptr(),
                                             nbytes, ext.data_ptr(), aff[0].data_ptr(), aff[1].data_ptr(), _lib.current_stream(dev)),
                   "b3d_point_stack_train")
    return ext, aff[0], aff[1]


def _train_stack(owner, convs, bns, x, trans, relu_last):
    """(output or extremes, affine or None) of a train-mode point stack: the one-call form, or -- B3D_OLD_STACK=1, A/B runs of
    tools/ only -- round 4's composition of ten launches (``point_feat_train_hip``)."""
    import os
    if os.environ.get("B3D_OLD_STACK"):
        return point_feat_train_hip(convs, bns, x, trans=trans, relu_last=relu_last), None
    ext, sc, sh = point_stack_train_hip(owner, convs, bns, x, trans=trans)
    return ext, (sc, sh, relu_last)


def _materialize(x: torch.Tensor, affine) -> torch.Tensor:
    """``x`` itself, or -- for a stack output that is still (extremes, scale, shift, relu) -- the activation it stands for."""
    if affine is None:
        return x
    from . import _lib
    sc, sh, relu = affine
    out = torch.empty_like(x)
    _lib.check(_lib.load().b3d_affine(x.data_ptr(), sc.data_ptr(), sh.data_ptr(), x.size(0), x.size(1), int(bool(relu)), out.data_ptr(),
                                      _lib.current_stream(x.device)), "b3d_affine")
    return out


def _bn_struct(bn: nn.BatchNorm1d, train: bool, keep: list):
    from . import _lib
    s = _lib.b3d_batchnorm()
    g, b = bn.weight.detach().float().contiguous(), bn.bias.detach().float().contiguous()
    keep += [g, b]
    s.gamma, s.beta = g.data_ptr(), b.data_ptr()
    tracked = bn.track_running_stats and bn.running_mean is not None
    if tracked:
        _lib.require_cuda(bn.running_mean, "BatchNorm running statistic", torch.float32)
        _lib.require_cuda(bn.running_var, "BatchNorm running statistic", torch.float32)
        s.running_mean, s.running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
        s.num_batches_tracked = bn.num_batches_tracked.data_ptr() if train else None
    elif not train:
        raise ValueError("fc head: eval mode needs running statistics")
    s.momentum = float(bn.momentum) if bn.momentum is not None else -1.0
    s.eps = float(bn.eps)
    return s


def _draw_dropout_mask(b: int, n: int, p: float, dev) -> torch.Tensor:
    """The [b, n] Dropout mask (0 or 1 / (1 - p)) of one fc-head stage: ONE torch call on a tensor of ones, so the draw comes
    from torch's Philox stream exactly as ``nn.Dropout`` on a [b, n] input would take it.  Tests that pin the HIP path against
    masks the reference drew (tests/golden/g11_*) substitute this function."""
    return F.dropout(torch.ones(b, n, dtype=torch.float32, device=dev), p, True)


# Dropout masks drawn AHEAD of the launch that uses them, keyed by the nn.Dropout module: train_step.EncodeAhead enqueues RadarNet in
# front of PointNet when that fills the step better, and the generator must still be drawn from in the reference's order (PointNet's
# mask first, pointnet.py:190, then RadarNet's, radarnet.py:62).
_PREDRAWN: dict = {}


def predraw_dropout_mask(dropout: nn.Dropout, b: int, n: int, dev) -> None:
    """Draw the [b, n] mask ``dropout`` will need NOW (current stream, torch's generator) and park it for the next ``fc_head_hip``
    stage that uses this module; nothing happens in eval mode or for p = 0."""
    if dropout.training and dropout.p > 0:
        _PREDRAWN[id(dropout)] = _draw_dropout_mask(b, n, float(dropout.p), dev)


def discard_predrawn_masks() -> None:
    _PREDRAWN.clear()


def _fc_workspace(owner: nn.Module, nbytes: int, dev) -> torch.Tensor:
    """Workspace of ``b3d_fc_bn_forward`` for ``owner``'s head: zero-filled ONCE per (module, power-of-two size bucket, device) and
    kept for the life of the module -- the launches leave the arrival counters zero, the rest is overwritten by every call.  A
    bucket is never freed or replaced (hipGraphs captured earlier keep writing to it on replay: a buffer that was swapped for a
    larger one used to be freed under them), and never CREATED inside a stream capture (its zero fill would be captured instead
    of executed): there, and for the first call of a new size while capturing, the call gets a fresh zero-filled tensor."""
    bucket = 1 << max(12, int(nbytes - 1).bit_length())
    cache = owner.__dict__.setdefault("_b3d_fc_ws", {})
    t = cache.get((bucket, str(dev)))
    if t is None:
        t = torch.zeros(bucket, dtype=torch.uint8, device=dev)
        if not torch.cuda.is_current_stream_capturing():
            cache[(bucket, str(dev))] = t
    return t


def _identity9(owner: nn.Module, dev) -> torch.Tensor:
    """The flattened 3 x 3 identity STN3d adds to its last Linear (pointnet.py:53-56), built once per (module, device)."""
    t = owner.__dict__.get("_b3d_iden")
    if t is None or t.device != dev:
        t = torch.eye(3, dtype=torch.float32, device=dev).view(9).contiguous()
        if not torch.cuda.is_current_stream_capturing():
            owner.__dict__["_b3d_iden"] = t
    return t


def fc_head_hip(owner: nn.Module, x: torch.Tensor, stages, final_relu: bool = True, in_affine=None, out=None):
    """A chain of Linear (+ BatchNorm1d + ReLU) stages on [B, K] rows, one HIP launch per Linear (``b3d_fc_bn_forward``):
    every launch applies the PREVIOUS stage's BatchNorm + ReLU while it reads its input, multiplies an optional Dropout
    mask into its output and accumulates the batch statistics its own BatchNorm needs -- BatchNorm / ReLU / Dropout
    never run as kernels of their own.  ``stages``: list of ``(fc, bn or None, dropout or None, add or None)``; a stage without ``bn``
    must be the last one.  A Linear may have at most 4,032 outputs (one arrival counter per 64-column tile in the 256-byte
    workspace header; the encoders' widest is 512).  Train mode (``bn.training``): batch statistics, running statistics updated as torch does; the
    Dropout mask is drawn by ONE torch call on a tensor of ones (the Philox stream stays torch's).  No autograd (frozen).
    ``in_affine``: ``(scale [K], shift [K], relu)`` -- the first Linear's input is ``act(x * scale + shift)`` (a train-mode point
    stack hands over its per-cloud extremes and the batch's last BatchNorm this way, ``point_stack_train_hip``)."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    x = x.detach().float().contiguous()
    _lib.require_cuda(x, "fc head input", torch.float32)
    b = x.size(0)
    dev = x.device
    stream = _lib.current_stream(dev)
    # Scratch of the chain (arrival counters + per-row-tile partial statistics).  The counters must be zero on entry and every launch
    # leaves them zero (_fc_workspace): one zero fill per (module, size bucket) for the life of the process, not one per call.
    nmax = max(fc.out_features for fc, _, _, _ in stages)
    nbytes = lib.b3d_fc_bn_workspace_bytes(b, nmax)
    ws = _fc_workspace(owner, nbytes, dev)
    keep = []
    in_scale = in_shift = None
    in_relu = 1
    if in_affine is not None:
        in_scale, in_shift = (t_.detach().float().contiguous() for t_ in in_affine[:2])
        in_relu = int(bool(in_affine[2]))
    cur = x
    with torch.no_grad():
        for i, (fc, bn, dropout, add) in enumerate(stages):
            w = fc.weight.detach().float().contiguous()
            bias = fc.bias.detach().float().contiguous() if fc.bias is not None else None
            n, k = w.shape
            if cur.size(1) != k:
                raise ValueError(f"fc head: stage {i} expects {k} inputs, got {cur.size(1)}")
            train = bool(bn.training) if bn is not None else False
            mask = None
            if dropout is not None and dropout.training and dropout.p > 0:
                mask = _PREDRAWN.pop(id(dropout), None)
                if mask is None or tuple(mask.shape) != (b, n) or mask.device != dev:
                    mask = _draw_dropout_mask(b, n, float(dropout.p), dev)
            y = torch.empty(b, n, dtype=torch.float32, device=dev)
            sc = torch.empty(n, dtype=torch.float32, device=dev) if bn is not None else None
            sh = torch.empty(n, dtype=torch.float32, device=dev) if bn is not None else None
            bs = _bn_struct(bn, train, keep) if bn is not None else None
            addv = add.detach().float().contiguous() if add is not None else None
            _lib.check(lib.b3d_fc_bn_forward(cur.data_ptr(), b, k, w.data_ptr(), _lib.ptr(bias), n, _lib.ptr(in_scale), _lib.ptr(in_shift),
                                             in_relu, _lib.ptr(mask), _lib.ptr(addv), C.byref(bs) if bs is not None else None, int(train),
                                             y.data_ptr(), _lib.ptr(sc), _lib.ptr(sh), ws.data_ptr(), ws.numel(), stream),
                       "b3d_fc_bn_forward")
            keep += [w, bias, mask, addv, cur, in_scale, in_shift]
            cur, in_scale, in_shift, in_relu = y, sc, sh, 1
        if in_scale is not None:                     # the last stage had a BatchNorm: materialise relu(bn(y))
            if not final_relu:
                raise ValueError("fc head: a trailing BatchNorm without ReLU is not a shape of these encoders")
            if out is None:
                out = torch.empty_like(cur)
            elif out.shape != cur.shape or out.dtype != torch.float32 or not out.is_contiguous() or out.device != cur.device:
                raise ValueError(f"fc head: out must be a contiguous float32 {tuple(cur.shape)} tensor on {cur.device}")
            _lib.check(lib.b3d_affine_relu(cur.data_ptr(), in_scale.data_ptr(), in_shift.data_ptr(), b, cur.size(1), out.data_ptr(), stream),
                       "b3d_affine_relu")
            cur = out
    return cur


def _empty(module: nn.Module, x: torch.Tensor, width: int):
    """An empty batch on the GPU (no row of a batch carries the modality: clr_att_gnn.py:131,139 call the encoder on a [0, C, P]
    tensor): the result is an empty [0, width] tensor, nothing is launched and no statistic moves -- None otherwise."""
    if x.is_cuda and x.size(0) == 0 and getattr(module, "use_hip", True):
        return x.new_zeros((0, width), dtype=torch.float32)
    return None


def _stack_blocker(module: nn.Module, x: torch.Tensor):
    """Why the HIP point stack cannot take ``x`` [B, C, P] (None: it can).  Eval mode needs no autograd through it; train mode
    additionally frozen parameters (batch statistics without a backward) and more than one point."""
    if module.training:
        if x.size(0) * x.size(2) <= 1:
            return "train-mode BatchNorm needs more than one value per channel"
        if x.requires_grad or any(p.requires_grad for p in module.parameters()):
            return "train mode with unfrozen parameters (the HIP encoders are forward-only: the GNN holds them frozen, clr_att_gnn.py:26-33)"
        return None
    return _autograd_blocker(module, x)


def _fc_blocker(module: nn.Module, x: torch.Tensor, bns):
    if any(bn.training for bn in bns) and x.size(0) <= 1:
        return "train-mode BatchNorm needs more than one row"
    return _autograd_blocker(module, x)


def resnet_encode_hip(m: "ResNetAE", x: torch.Tensor, out=None) -> torch.Tensor:
    """``ResNetAE.encode`` in six HIP phase kernels (``b3d_resnet_encode``): direct convolutions with the producer's
    BatchNorm / residual add / ReLU applied while the next phase stages its input; in train mode the batch statistics
    are accumulated by the producing phase and the running statistics updated as ``nn.BatchNorm2d`` does.  No autograd
    (the GNN holds the encoder frozen, clr_att_gnn.py:26-33)."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    x = x.float().contiguous()
    _lib.require_cuda(x, "image crops", torch.float32)
    if x.dim() != 4 or tuple(x.shape[1:]) != (3, 32, 32):
        raise ValueError(f"resnet_encode_hip: crops are [N, 3, 32, 32], got {tuple(x.shape)}")
    blocks = (m.res_block1, m.res_block2, m.res_block3)
    for blk in blocks:
        if not (isinstance(blk.downsample, nn.Sequential) and len(blk.downsample) == 2):
            raise ValueError("resnet_encode_hip: every block of ResNetAE has a conv + BatchNorm downsample branch")
    convs = [m.conv] + [c for blk in blocks for c in (blk.conv1, blk.conv2, blk.downsample[0])]
    bns = [b for blk in blocks for b in (blk.bn1, blk.bn2, blk.downsample[1])]
    train = bool(m.training)
    n = x.size(0)
    keep = []

    def f32(t_):
        t_ = t_.detach().float().contiguous()
        _lib.require_cuda(t_, "ResNetAE parameter", torch.float32)
        keep.append(t_)
        return t_.data_ptr()

    cl = (_lib.b3d_linear * 10)()
    for i, cv in enumerate(convs):
        cl[i].w, cl[i].b = f32(cv.weight), f32(cv.bias)
    bl = (_lib.b3d_batchnorm * 9)()
    for i, bn in enumerate(bns):
        bl[i].gamma, bl[i].beta = f32(bn.weight), f32(bn.bias)
        tracked = bn.track_running_stats and bn.running_mean is not None
        if tracked:
            _lib.require_cuda(bn.running_mean, "BatchNorm running statistic", torch.float32)
            _lib.require_cuda(bn.running_var, "BatchNorm running statistic", torch.float32)
            bl[i].running_mean, bl[i].running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
            bl[i].num_batches_tracked = bn.num_batches_tracked.data_ptr() if train else None
        elif not train:
            raise ValueError("resnet_encode_hip: eval mode needs running statistics")
        bl[i].momentum = float(bn.momentum) if bn.momentum is not None else -1.0
        bl[i].eps = float(bn.eps)
    if out is None:
        out = torch.empty(n, 96, dtype=torch.float32, device=x.device)
    elif tuple(out.shape) != (n, 96) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != x.device:
        raise ValueError(f"resnet_encode_hip: out must be a contiguous float32 [{n}, 96] tensor on {x.device}")
    nbytes = lib.b3d_resnet_encode_workspace_bytes(n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    _lib.check(lib.b3d_resnet_encode(cl, bl, x.data_ptr(), n, int(train), ws.data_ptr(), nbytes, out.data_ptr(),
                                     _lib.current_stream(x.device)), "b3d_resnet_encode")
    return out


def reference_order_(module: nn.Module) -> nn.Module:
    """Make every sub-module evaluate operation for operation as the reference does (no HIP kernels, no BatchNorm
    folding): what the CPU oracle runs on."""
    for mod in module.modules():
        mod.use_hip = False
        mod.fold_bn = False
    return module


class _ResidualBlock(nn.Module):
    def __init__(self, cin, cout, k, stride, down):
        super().__init__()
        self.downsample = down
        self.conv1 = nn.Conv2d(cin, cout, k, stride, padding=1)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, k, stride, padding=1)
        self.bn2 = nn.BatchNorm2d(cout)

    def forward(self, x):
        if self.downsample is None:
            skip = x
        elif isinstance(self.downsample, nn.Sequential) and len(self.downsample) == 2:
            skip = _conv_bn(self.downsample[0], self.downsample[1], x)
        else:
            skip = self.downsample(x)
        y = F.relu(_conv_bn(self.conv1, self.bn1, x))
        y = _conv_bn(self.conv2, self.bn2, y)
        return F.relu(y + skip)


def _down(cin, cout, k, s):
    return nn.Sequential(nn.Conv2d(cin, cout, k, s), nn.BatchNorm2d(cout))


class ResNetAE(nn.Module):
    supports_out = True          # encode(x, out=...): the HIP path can write into a caller's buffer (clr_att_gnn.GNN._encode_img)

    def __init__(self):
        super().__init__()
        self.conv = nn.Conv2d(3, 12, kernel_size=4, stride=2, padding=1)
        self.bn = nn.BatchNorm2d(12)          # declared by the reference, not used by encode()
        self.res_block1 = _ResidualBlock(12, 24, 4, 2, _down(12, 24, 5, 3))
        self.res_block2 = _ResidualBlock(24, 48, 3, 1, _down(24, 48, 1, 1))
        self.res_block3 = _ResidualBlock(48, 96, 3, 2, _down(48, 96, 3, 2))
        self.fc_encoder = nn.Sequential(nn.Linear(192, 128), nn.BatchNorm1d(128, momentum=0.01), nn.ReLU(),
                                        nn.Linear(128, 64), nn.BatchNorm1d(64, momentum=0.01), nn.ReLU())
        self.fc_decoder = nn.Sequential(nn.Linear(64, 128), nn.BatchNorm1d(128, momentum=0.01), nn.ReLU(),
                                        nn.Linear(128, 192), nn.BatchNorm1d(192, momentum=0.01), nn.ReLU())
        self.conv_decoder = nn.Sequential(
            nn.ConvTranspose2d(96, 72, 4, stride=2, padding=1), nn.ReLU(),
            nn.ConvTranspose2d(72, 48, 4, stride=2, padding=1), nn.ReLU(),
            nn.ConvTranspose2d(48, 24, 4, stride=2, padding=1), nn.ReLU(),
            nn.ConvTranspose2d(24, 12, 4, stride=2, padding=1), nn.ReLU(),
            nn.ConvTranspose2d(12, 3, 4, stride=2, padding=1), nn.Sigmoid())

    def encode(self, x, out=None):
        """``out`` (HIP path only): a [N, 96] float32 tensor the embedding is written to."""
        why = None
        if x.dim() == 4 and _empty(self, x, 96) is not None:
            return _empty(self, x, 96)
        if x.dim() != 4 or tuple(x.shape[1:]) != (3, 32, 32):
            why = f"crops are [N, 3, 32, 32] (the size the GNN feeds), got {tuple(x.shape)}"
        elif self.training and x.size(0) <= 1:
            why = "train-mode BatchNorm needs more than one crop"
        elif self.training and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            why = "train mode with unfrozen parameters (the HIP encoders are forward-only: the GNN holds them frozen, clr_att_gnn.py:26-33)"
        elif not self.training:
            why = _autograd_blocker(self, x)
        if _route(self, "resnet.encode", x, why):
            return resnet_encode_hip(self, x, out=out)
        out = self.res_block3(self.res_block2(self.res_block1(self.conv(x))))
        return out.view(out.size(0), -1)


class _STN3d(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1, self.conv2, self.conv3 = nn.Conv1d(3, 64, 1), nn.Conv1d(64, 128, 1), nn.Conv1d(128, 1024, 1)
        self.fc1, self.fc2, self.fc3 = nn.Linear(1024, 512), nn.Linear(512, 256), nn.Linear(256, 9)
        self.bn1, self.bn2, self.bn3 = nn.BatchNorm1d(64), nn.BatchNorm1d(128), nn.BatchNorm1d(1024)
        self.bn4, self.bn5 = nn.BatchNorm1d(512), nn.BatchNorm1d(256)

    def forward(self, x):
        b = x.size(0)
        affine = None
        if _empty(self, x, 9) is not None:
            return _empty(self, x, 9).view(0, 3, 3)
        if _route(self, "stn.points", x, _stack_blocker(self, x)):
            cb = ((self.conv1, self.conv2, self.conv3), (self.bn1, self.bn2, self.bn3))
            if self.training:
                x, affine = _train_stack(self, *cb, x, None, True)     # relu(bn3(.)) of pointnet.py:40, applied by the consumer
            else:
                x = point_feat_hip(*cb, x, relu_last=True)
        else:
            x = F.relu(self.bn1(self.conv1(x)))
            x = F.relu(self.bn2(self.conv2(x)))
            x = F.relu(self.bn3(self.conv3(x)))
            x = torch.max(x, 2, keepdim=True)[0].view(-1, 1024)
        if _route(self, "stn.fc", x, _fc_blocker(self, x, (self.bn4, self.bn5))):
            iden = _identity9(self, x.device)
            x = fc_head_hip(self, x, [(self.fc1, self.bn4, None, None), (self.fc2, self.bn5, None, None), (self.fc3, None, None, iden)],
                            in_affine=affine)
            return x.view(-1, 3, 3)
        x = _materialize(x, affine)
        x = F.relu(_fc_bn(self.fc1, self.bn4, x))
        x = F.relu(_fc_bn(self.fc2, self.bn5, x))
        x = self.fc3(x)
        iden = torch.eye(3, dtype=x.dtype, device=x.device).view(1, 9).repeat(b, 1)
        return (x + iden).view(-1, 3, 3)


class _PointNetFeat(nn.Module):
    def __init__(self):
        super().__init__()
        self.stn = _STN3d()
        self.conv1, self.conv2, self.conv3 = nn.Conv1d(3, 64, 1), nn.Conv1d(64, 128, 1), nn.Conv1d(128, 1024, 1)
        self.bn1, self.bn2, self.bn3 = nn.BatchNorm1d(64), nn.BatchNorm1d(128), nn.BatchNorm1d(1024)

    def forward(self, x):
        return _materialize(*self.forward_parts(x))

    def forward_parts(self, x):
        """(features or per-cloud extremes, None or the affine map (scale, shift, relu) that turns the extremes into the features):
        the train-mode HIP stack leaves its last BatchNorm to the consumer (``point_stack_train_hip``)."""
        if _empty(self, x, 1024) is not None:
            return _empty(self, x, 1024), None
        trans = self.stn(x)
        if _route(self, "pointnet.points", x, _stack_blocker(self, x)):          # the bmm is applied while the kernel loads the points
            cb = ((self.conv1, self.conv2, self.conv3), (self.bn1, self.bn2, self.bn3))
            if self.training:
                return _train_stack(self, *cb, x, trans, False)        # bn3 without ReLU (pointnet.py:158)
            return point_feat_hip(*cb, x, trans=trans), None
        x = torch.bmm(x.transpose(2, 1), trans).transpose(2, 1)
        x = F.relu(self.bn1(self.conv1(x)))
        x = F.relu(self.bn2(self.conv2(x)))
        x = self.bn3(self.conv3(x))
        return torch.max(x, 2, keepdim=True)[0].view(-1, 1024), None


class PointNetClassifier(nn.Module):
    supports_out = True          # forward_feat(x, out=...)

    def __init__(self, k=7, feature_transform=False):
        super().__init__()
        if feature_transform:
            raise NotImplementedError("feature_transform=True is not used by the GNN path")
        self.feat = _PointNetFeat()
        self.fc1, self.fc2, self.fc3 = nn.Linear(1024, 512), nn.Linear(512, 256), nn.Linear(256, k)
        self.dropout = nn.Dropout(p=0.3)
        self.bn1, self.bn2 = nn.BatchNorm1d(512), nn.BatchNorm1d(256)

    def forward_feat(self, x, out=None):
        """``out`` (HIP path only): a [B, 256] float32 tensor the result is written to (a static buffer of a captured step)."""
        if _empty(self, x, 256) is not None:
            return _empty(self, x, 256)
        x, affine = self.feat.forward_parts(x)
        if _route(self, "pointnet.fc", x, _fc_blocker(self, x, (self.bn1, self.bn2))):
            return fc_head_hip(self, x, [(self.fc1, self.bn1, None, None), (self.fc2, self.bn2, self.dropout, None)], in_affine=affine,
                               out=out)
        x = _materialize(x, affine)
        x = F.relu(_fc_bn(self.fc1, self.bn1, x))
        if not _fold_on(self.bn2):
            return F.relu(self.bn2(self.dropout(self.fc2(x))))
        return F.relu(_fc_bn(self.fc2, self.bn2, x))           # eval: dropout is the identity


class _RadarNetFeat(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1, self.conv2, self.conv3 = nn.Conv1d(4, 64, 1), nn.Conv1d(64, 128, 1), nn.Conv1d(128, 1024, 1)
        self.bn1, self.bn2, self.bn3 = nn.BatchNorm1d(64), nn.BatchNorm1d(128), nn.BatchNorm1d(1024)

    def forward(self, x):
        return _materialize(*self.forward_parts(x))

    def forward_parts(self, x):
        """As ``_PointNetFeat.forward_parts``."""
        if _empty(self, x, 1024) is not None:
            return _empty(self, x, 1024), None
        if _route(self, "radarnet.points", x, _stack_blocker(self, x)):
            cb = ((self.conv1, self.conv2, self.conv3), (self.bn1, self.bn2, self.bn3))
            if self.training:
                return _train_stack(self, *cb, x, None, False)         # bn3 without ReLU (radarnet.py:35)
            return point_feat_hip(*cb, x), None
        x = F.relu(self.bn1(self.conv1(x)))
        x = F.relu(self.bn2(self.conv2(x)))
        x = self.bn3(self.conv3(x))
        return torch.max(x, 2, keepdim=True)[0].view(-1, 1024), None


class RadarNetClassifier(nn.Module):
    supports_out = True          # forward_feat(x, out=...)

    def __init__(self, k=2, feature_transform=False):
        super().__init__()
        self.feat = _RadarNetFeat()
        self.fc1, self.fc2, self.fc3 = nn.Linear(1024, 512), nn.Linear(512, 256), nn.Linear(256, k)
        self.dropout = nn.Dropout(p=0.3)
        self.bn1, self.bn2 = nn.BatchNorm1d(512), nn.BatchNorm1d(256)

    def forward_feat(self, x, out=None):
        """``out``: as ``PointNetClassifier.forward_feat``."""
        if _empty(self, x, 256) is not None:
            return _empty(self, x, 256)
        x, affine = self.feat.forward_parts(x)
        if _route(self, "radarnet.fc", x, _fc_blocker(self, x, (self.bn1, self.bn2))):
            return fc_head_hip(self, x, [(self.fc1, self.bn1, None, None), (self.fc2, self.bn2, self.dropout, None)], in_affine=affine,
                               out=out)
        x = _materialize(x, affine)
        x = F.relu(_fc_bn(self.fc1, self.bn1, x))
        if not _fold_on(self.bn2):
            return F.relu(self.bn2(self.dropout(self.fc2(x))))
        return F.relu(_fc_bn(self.fc2, self.bn2, x))           # eval: dropout is the identity
