"""Post-processing of per-window edge scores (reference predict.py:199-259), vectorised.

    scene_edges[(gid_out, gid_in)].append(score)            predict.py:221   (over overlapping windows)
    avg = mean per global edge                              predict.py:227
    keep avg > threshold[class of the source node]          predict.py:231-233
    per node: best incoming / best outgoing kept edge       predict.py:92-117 (greedy_filter_node_flux)

The reference keys everything by ``str(meta)`` dictionaries in Python loops; here global node ids
are integers and the three steps are segmented mean / compare / segmented argmax on tensors (any
device).  Tie rule of the reference: ``max(d, key=d.get)`` returns the first maximal entry in
insertion order, i.e. the edge whose first appearance over the windows is earliest -- reproduced
exactly so that the kept-edge set and the argmax indices are identical.
"""
from __future__ import annotations

from typing import Dict, Sequence

import torch

# predict.py:231
EDGE_SCORE_THRESHOLDS = {"bicycle": 0.1, "bus": 0.005, "car": 0.02, "motorcycle": 0.03,
                         "pedestrian": 0.025, "trailer": 0.04, "truck": 0.005}


def average_window_scores(pairs: torch.Tensor, scores: torch.Tensor, num_nodes: int):
    """pairs [M,2] int64 (global source, global destination) of every scored edge of every window in
    processing order, scores [M].  Returns (unique pairs [U,2] in first-appearance order, mean [U]
    float64, first-appearance position [U])."""
    key = pairs[:, 0] * num_nodes + pairs[:, 1]
    uniq, inv = torch.unique(key, return_inverse=True)
    u = uniq.numel()
    pos = torch.arange(key.numel(), device=key.device)
    first = torch.full((u,), key.numel(), dtype=torch.long, device=key.device).scatter_reduce(0, inv, pos, "amin")
    ssum = torch.zeros(u, dtype=torch.float64, device=key.device).index_add_(0, inv, scores.double())
    cnt = torch.zeros(u, dtype=torch.float64, device=key.device).index_add_(0, inv, torch.ones_like(scores, dtype=torch.float64))
    order = torch.argsort(first)
    up = torch.stack([uniq // num_nodes, uniq % num_nodes], 1)
    return up[order], (ssum / cnt)[order], first[order]


def _segmented_first_argmax(seg: torch.Tensor, other: torch.Tensor, score: torch.Tensor, num_nodes: int):
    """For every node n: `other` of the highest-scoring edge with seg == n (first in order on ties)."""
    out = torch.full((num_nodes,), -1, dtype=torch.long, device=seg.device)
    if seg.numel() == 0:
        return out
    best = torch.full((num_nodes,), float("-inf"), dtype=score.dtype, device=seg.device).scatter_reduce(0, seg, score, "amax")
    is_best = score == best[seg]
    pos = torch.arange(seg.numel(), device=seg.device)
    big = seg.numel()
    first = torch.full((num_nodes,), big, dtype=torch.long, device=seg.device).scatter_reduce(
        0, seg[is_best], pos[is_best], "amin")
    has = first < big
    out[has] = other[first[has]]
    return out


def greedy_edges(pairs: torch.Tensor, scores: torch.Tensor, node_class: torch.Tensor,
                 class_names: Sequence[str], thresholds: Dict[str, float] = EDGE_SCORE_THRESHOLDS):
    """Full H2 step.  node_class [num_nodes] indexes ``class_names``.  Returns a dict with
    ``kept_pairs`` [K,2] (first-appearance order), ``kept_scores`` [K] float64, ``pred`` [num_nodes]
    (best predecessor or -1) and ``succ`` [num_nodes]."""
    n = node_class.numel()
    up, mean, _first = average_window_scores(pairs, scores, n)
    thr = torch.tensor([thresholds[c] for c in class_names], dtype=torch.float64, device=pairs.device)
    keep = mean > thr[node_class[up[:, 0]]]
    kp, ks = up[keep], mean[keep]
    pred = _segmented_first_argmax(kp[:, 1], kp[:, 0], ks, n)       # incoming: keyed by destination
    succ = _segmented_first_argmax(kp[:, 0], kp[:, 1], ks, n)       # outgoing: keyed by source
    return {"kept_pairs": kp, "kept_scores": ks, "pred": pred, "succ": succ}


def greedy_edges_hip(pairs: torch.Tensor, scores: torch.Tensor, node_class: torch.Tensor,
                     class_names: Sequence[str], thresholds: Dict[str, float] = EDGE_SCORE_THRESHOLDS):
    """``greedy_edges`` on the GPU in one C-ABI call (``b3d_post_greedy``: stable device sort by global edge id,
    means in order of appearance, threshold, segmented first-arg-max).  Same outputs, same order, same ties."""
    from . import _lib
    lib = _lib.load()
    _lib.require_cuda(pairs, "pairs", torch.int64)
    dev = pairs.device
    m, n = int(pairs.size(0)), int(node_class.numel())
    sc = scores.reshape(-1).float().contiguous()
    nc = node_class.to(device=dev, dtype=torch.int64).contiguous()
    thr = torch.tensor([thresholds[c] for c in class_names], dtype=torch.float64, device=dev)
    nbytes = lib.b3d_post_workspace_bytes(m, n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    kept_pairs = torch.empty((max(m, 1), 2), dtype=torch.int64, device=dev)
    kept_scores = torch.empty(max(m, 1), dtype=torch.float64, device=dev)
    pred = torch.empty(n, dtype=torch.int64, device=dev)
    succ = torch.empty(n, dtype=torch.int64, device=dev)
    counts = torch.empty(3, dtype=torch.int32, device=dev)
    _lib.check(lib.b3d_post_greedy(pairs.contiguous().data_ptr(), sc.data_ptr(), m, nc.data_ptr(), n, thr.data_ptr(),
                                   len(class_names), ws.data_ptr(), nbytes, kept_pairs.data_ptr(), kept_scores.data_ptr(), pred.data_ptr(),
                                   succ.data_ptr(), counts.data_ptr(), _lib.current_stream(dev)), "b3d_post_greedy")
    _, k, invalid = counts.tolist()                         # the only host read: the size of the kept-edge list
    if invalid:
        raise ValueError(f"{invalid} entries of pairs / node_class lie outside [0, {n}) / [0, {len(class_names)}) "
                         "(the reference's dictionaries raise a KeyError for them, predict.py:92-117)")
    return {"kept_pairs": kept_pairs[:k], "kept_scores": kept_scores[:k], "pred": pred, "succ": succ}


# predict.py:302
TRACK_JOIN_SCORES = dict(EDGE_SCORE_THRESHOLDS)


def create_trajectories(pred_edges, scene_nodes, join_score: Dict[str, float] = TRACK_JOIN_SCORES):
    """``create_trajectories(pred_edges, scene_nodes)`` of the reference (predict.py:262-375, mode "hier"): the greedy
    edges ``[((j, i), score), ...]`` of a scene, taken by descending score, grow clusters at their ends and join two
    clusters tail-to-head where the edge's score exceeds the threshold of the destination's class.  ``scene_nodes``
    maps node id -> {"category_name": ...}.  Returns the tracks (lists of node ids) in the reference's order.

    The merge is sequential by construction; it runs in the library's host code (``b3d_tracks_from_edges``: linked
    clusters, O(M log M)) instead of Python lists and dictionaries."""
    import ctypes as C
    import numpy as np
    from . import _lib
    lib = _lib.load()
    ids = sorted(scene_nodes.keys())
    remap = {gid: k for k, gid in enumerate(ids)}
    names = sorted(join_score.keys())
    cls = np.asarray([names.index(scene_nodes[g]["category_name"]) for g in ids], dtype=np.int64)
    thr = np.asarray([join_score[c] for c in names], dtype=np.float64)
    m = len(pred_edges)
    pairs = np.asarray([[remap[e[0][0]], remap[e[0][1]]] for e in pred_edges], dtype=np.int64).reshape(m, 2)
    scores = np.asarray([float(e[1]) for e in pred_edges], dtype=np.float64)
    nodes = np.empty(max(2 * m, 1), dtype=np.int64)
    ptr = np.empty(m + 2, dtype=np.int64)
    nt = C.c_int64()
    _lib.check(lib.b3d_tracks_from_edges(pairs.ctypes.data, scores.ctypes.data, m, cls.ctypes.data, len(ids), thr.ctypes.data,
                                         len(names), nodes.ctypes.data, ptr.ctypes.data, C.byref(nt)), "b3d_tracks_from_edges")
    return [[ids[k] for k in nodes[ptr[t]:ptr[t + 1]]] for t in range(nt.value)]
