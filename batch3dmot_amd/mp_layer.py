"""``CausalMessagePassing.forward`` as a standalone operator (reference pose_gnn.py:125-252,
clr_att_gnn.py:227-356): ``(x, edge_index, edge_attr, initial_x[, att_edge_attr]) -> (x', edge_attr')``.

The models call the same kernels through their whole-forward entry points; this module serves code
that drives a layer directly (as the reference's ``GNN.forward`` does, pose_gnn.py:83).
Both width sets have forward and backward (``b3d_pose_layer_*`` / ``b3d_clr_layer_*``); the camera+LiDAR+radar layer
also returns the gradient of ``att_edge_attr``.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import torch

from . import _lib
from ._lib import B3D_FLAG_TRAINING

_DIMS = {"p": (48, 32), "clr": (96, 64)}          # (node width, edge width)


def _params(module) -> List[torch.Tensor]:
    out = []
    for seq in (module.edge_update, module.create_past_msgs, module.create_future_msgs, module.combine_future_past):
        for lin in seq:
            if isinstance(lin, torch.nn.Linear):
                out += [lin.weight, lin.bias]
    return out


def _mp_struct(tensors):
    s = _lib.b3d_mp_weights()
    k = 0
    for arr in (s.edge_update, s.create_past_msgs, s.create_future_msgs, s.combine_future_past):
        for i in range(len(arr)):
            arr[i].w = tensors[k].data_ptr()
            arr[i].b = tensors[k + 1].data_ptr()
            k += 2
    assert k == len(tensors)
    return s


class _MPLayerFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, kind, graph, training, x, x0, e, att, *params):
        lib = _lib.load()
        dev = x.device
        N, E = graph.N, graph.E
        dx, de = _DIMS[kind]
        params = [p.detach() for p in params]
        w = _mp_struct(params)
        x_new = torch.empty((N, dx), dtype=torch.float32, device=dev)
        e_new = torch.empty((E, de), dtype=torch.float32, device=dev)
        stream = _lib.current_stream(dev)
        if kind == "p":
            flags = B3D_FLAG_TRAINING if training else 0
            nbytes = lib.b3d_pose_layer_workspace_bytes(N, E, flags)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            _lib.check(lib.b3d_pose_layer_forward(C.byref(w), C.byref(graph.c), x.data_ptr(), x0.data_ptr(), e.data_ptr(),
                                                  flags, ws.data_ptr(), nbytes, x_new.data_ptr(), e_new.data_ptr(), stream),
                       "b3d_pose_layer_forward")
        else:
            flags = B3D_FLAG_TRAINING if training else 0
            nbytes = lib.b3d_clr_layer_workspace_bytes(N, E, flags)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            _lib.check(lib.b3d_clr_layer_forward(C.byref(w), C.byref(graph.c), x.data_ptr(), x0.data_ptr(), e.data_ptr(),
                                                 att.data_ptr(), flags, ws.data_ptr(), nbytes, x_new.data_ptr(), e_new.data_ptr(),
                                                 stream), "b3d_clr_layer_forward")
        ctx.set_materialize_grads(False)
        ctx.kind, ctx.graph, ctx.training, ctx.ws, ctx.nbytes = kind, graph, training, ws, nbytes
        # save_for_backward: no reference cycle through the output e_new, and in-place edits of x / e between forward
        # and backward are detected by autograd's version counters
        ctx.has_att = att is not None
        if att is not None:
            ctx.save_for_backward(x, x0, e, e_new, att, *params)
        else:
            ctx.save_for_backward(x, x0, e, e_new, *params)
        return x_new, e_new

    @staticmethod
    def backward(ctx, d_x_new, d_e_new):
        if not ctx.training:
            raise RuntimeError("backward through a CausalMessagePassing layer whose forward kept no state")
        lib = _lib.load()
        if ctx.has_att:
            x, x0, e, e_new, att, *params = ctx.saved_tensors
        else:
            x, x0, e, e_new, *params = ctx.saved_tensors
            att = None
        params = [p.detach() for p in params]
        dev = x.device
        if d_x_new is not None:
            d_x_new = d_x_new.contiguous().float()
        if d_e_new is not None:
            d_e_new = d_e_new.contiguous().float()
        d_x, d_x0, d_e = torch.empty_like(x), torch.empty_like(x0), torch.empty_like(e)
        grads = [torch.empty_like(p) for p in params]
        w = _mp_struct(params)
        g = _mp_struct(grads)
        if ctx.kind == "p":
            _lib.check(lib.b3d_pose_layer_backward(C.byref(w), C.byref(ctx.graph.c), x.data_ptr(), x0.data_ptr(), e.data_ptr(),
                                                   e_new.data_ptr(), ctx.ws.data_ptr(), ctx.nbytes, _lib.ptr(d_x_new),
                                                   _lib.ptr(d_e_new), d_x.data_ptr(), d_x0.data_ptr(), d_e.data_ptr(),
                                                   C.byref(g), _lib.current_stream(dev)), "b3d_pose_layer_backward")
            return (None, None, None, d_x, d_x0, d_e, None) + tuple(grads)
        d_att = torch.empty_like(att)
        _lib.check(lib.b3d_clr_layer_backward(C.byref(w), C.byref(ctx.graph.c), x.data_ptr(), x0.data_ptr(), e.data_ptr(),
                                              att.data_ptr(), e_new.data_ptr(), ctx.ws.data_ptr(), ctx.nbytes, _lib.ptr(d_x_new),
                                              _lib.ptr(d_e_new), d_x.data_ptr(), d_x0.data_ptr(), d_e.data_ptr(), d_att.data_ptr(),
                                              C.byref(g), _lib.current_stream(dev)), "b3d_clr_layer_backward")
        return (None, None, None, d_x, d_x0, d_e, d_att) + tuple(grads)


def mp_layer_forward(module, kind: str, x: torch.Tensor, edge_index: torch.Tensor, edge_attr: torch.Tensor,
                     initial_x: torch.Tensor, att_edge_attr: Optional[torch.Tensor]):
    dx, de = _DIMS[kind]
    n = x.size(0)
    for t, name, width, rows in ((x, "x", dx, n), (initial_x, "initial_x", dx, n),
                                 (edge_attr, "edge_attr", de, edge_index.size(1))):
        _lib.require_cuda(t, name, torch.float32)
        if t.dim() != 2 or t.size(1) != width or t.size(0) != rows:
            raise ValueError(f"{name} must be [{rows}, {width}], got {tuple(t.shape)}")
    if kind == "clr":
        if att_edge_attr is None:
            raise ValueError("att_edge_attr is required for the camera+LiDAR+radar layer (clr_att_gnn.py:186)")
        _lib.require_cuda(att_edge_attr, "att_edge_attr", torch.float32)
        if tuple(att_edge_attr.shape) != (edge_index.size(1), 64):
            raise ValueError(f"att_edge_attr must be [{edge_index.size(1)}, 64], got {tuple(att_edge_attr.shape)}")
    if n == 0 or edge_index.size(1) == 0:
        raise ValueError("empty graph: the reference's callers skip these (predict.py:179-180)")
    params = _params(module)
    for p in params:
        _lib.require_cuda(p, "parameter", torch.float32)
    inputs = [x, initial_x, edge_attr] + ([att_edge_attr] if att_edge_attr is not None else [])
    training = torch.is_grad_enabled() and any(t.requires_grad for t in inputs + params)
    graph = _graph_for(module, edge_index, n)
    return _MPLayerFunction.apply(kind, graph, training, x, initial_x, edge_attr, att_edge_attr, *params)


# id(module) -> (key, weakref to the edge_index tensor, _lib.Graph, build event, build stream).  Kept OFF the module: a
# _lib.Graph holds a ctypes struct with pointers, which copy.deepcopy / torch.save of the module cannot pickle; the entry goes
# away with the module (weakref.finalize).
_GRAPH_CACHE: dict = {}


def _graph_for(module, edge_index: torch.Tensor, n: int):
    """CSR / CSC structure of ``edge_index``, built (and its endpoints validated: one blocking 4-byte read-back) once per
    edge_index TENSOR: a model that applies the layer ``gnn_depth`` times to the same graph pays for one build.  The entry
    is keyed by the tensor object, its storage, shape and in-place version counter.  A structure built on one stream and
    reused from another is ordered by the build's event.  Not used inside a stream capture (a structure built there exists
    only once that graph has been replayed)."""
    import weakref
    if torch.cuda.is_current_stream_capturing():
        return _lib.Graph(edge_index.contiguous(), n)
    key = (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape), tuple(edge_index.stride()), n)
    cur = torch.cuda.current_stream(edge_index.device)
    hit = _GRAPH_CACHE.get(id(module))
    if hit is not None and hit[0] == key and hit[1]() is edge_index:
        if hit[4] != cur:
            cur.wait_event(hit[3])
            # the structure's storage belongs to the BUILD stream's pool: tell the caching allocator that this stream reads it
            # too, or a replaced / dropped entry could be handed out again on the build stream while kernels enqueued here
            # still walk it
            hit[2].ws.record_stream(cur)
        return hit[2]
    graph = _lib.Graph(edge_index.contiguous(), n)
    ev = torch.cuda.Event()
    ev.record(cur)
    if id(module) not in _GRAPH_CACHE:
        weakref.finalize(module, _GRAPH_CACHE.pop, id(module), None)
    _GRAPH_CACHE[id(module)] = (key, weakref.ref(edge_index), graph, ev, cur)
    return graph
