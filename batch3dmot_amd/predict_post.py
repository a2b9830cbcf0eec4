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


def tracks_from_arrays(pairs, scores, node_class, class_names: Sequence[str], join_score: Dict[str, float] = TRACK_JOIN_SCORES):
    """``create_trajectories`` on arrays (no dictionaries): greedy edges ``pairs`` [M,2] int64 / ``scores`` [M] float64 in the
    reference's list order, node ids dense in [0, len(node_class)), ``node_class`` indexing ``class_names``.  Returns the
    tracks as a list of int64 arrays (views of one buffer) in the reference's order."""
    import ctypes as C
    import numpy as np
    from . import _lib
    lib = _lib.load()
    pairs = np.ascontiguousarray(pairs, dtype=np.int64).reshape(-1, 2)
    scores = np.ascontiguousarray(scores, dtype=np.float64)
    cls = np.ascontiguousarray(node_class, dtype=np.int64)
    thr = np.asarray([join_score[c] for c in class_names], dtype=np.float64)
    m = int(pairs.shape[0])
    if m == 0:
        return []
    nodes = np.empty(2 * m, dtype=np.int64)
    ptr = np.empty(m + 2, dtype=np.int64)
    nt = C.c_int64()
    _lib.check(lib.b3d_tracks_from_edges(pairs.ctypes.data, scores.ctypes.data, m, cls.ctypes.data, int(cls.shape[0]), thr.ctypes.data,
                                         len(class_names), nodes.ctypes.data, ptr.ctypes.data, C.byref(nt)), "b3d_tracks_from_edges")
    return np.split(nodes[:ptr[nt.value]], ptr[1:nt.value])


def _union_graph(graphs, with_sensor_feats: bool):
    """Disjoint union of window graphs for one forward: node / edge tensors back to back, edge_index offset per window, the
    timestamps of window j shifted by 100000 j (frames of different windows stay apart for the discarded k-NN block).  The raw
    camera / LiDAR / radar tensors are only carried when the encoders run inside the forward."""
    from .data import Data
    offs, o = [], 0
    for g in graphs:
        offs.append(o)
        o += g.pose_feats.size(0)
    u = Data(pose_feats=torch.cat([g.pose_feats for g in graphs]),
             edge_index=torch.cat([g.edge_index + o_ for g, o_ in zip(graphs, offs)], 1).contiguous(),
             edge_attr=torch.cat([g.edge_attr for g in graphs]),
             node_timestamps=torch.cat([g.node_timestamps.to(torch.int64) + 100000 * j for j, g in enumerate(graphs)]))
    if with_sensor_feats:
        for k in ("img_feats", "lidar_feats", "radar_feats"):
            if getattr(graphs[0], k, None) is not None:
                setattr(u, k, torch.cat([getattr(g, k) for g in graphs]))
    return u


def predict_scene(model, windows, node_class: torch.Tensor, class_names: Sequence[str],
                  thresholds: Dict[str, float] = EDGE_SCORE_THRESHOLDS, cache=True, tracks: bool = True,
                  join_score: Dict[str, float] = TRACK_JOIN_SCORES, ids_of=None, windows_per_forward: int = 8):
    """Scene-level inference: the counterpart of the reference's ``combine_batches_to_scene`` + ``create_trajectories``
    (predict.py:143-259, 262-375) for one scene.

    ``windows``: the scene's overlapping graph windows in processing order (stride 1 over the frames, predict.py:172), each a
    ``Data`` on the device with the model's inputs and ``global_ids`` [n] (the scene-level id of every detection: the
    reference's ``meta2gid``, predict.py:199-207; ids must lie in [0, node_class.numel())).  Windows without nodes or
    without edges are skipped as the reference skips them (predict.py:179-180).  ``node_class`` [num scene nodes] indexes
    ``class_names``.

    Per window the model scores its edges under ``no_grad``; the scores of an edge that several windows contain are averaged
    (predict.py:221,227), thresholded by the class of its source (:231-233) and reduced to the best incoming / outgoing edge
    of every node (``b3d_post_greedy``: :92-117); ``tracks=True`` also clusters the greedy edges (``pred_edge_pairs`` /
    ``pred_edge_scores``: the reference's ``pred_edges`` list as arrays, in its order) into trajectories
    (``b3d_tracks_from_edges``: :262-375; ``tracks``: a list of node-id arrays).

    ``cache``: True (default) -- the camera+LiDAR+radar model encodes every DETECTION once per scene (``EmbeddingCache``: a
    sorted-id device table filled from the windows in one pass) instead of once per window it appears in; an
    ``EmbeddingCache`` instance to reuse / inspect; False -- the encoders run inside every window's forward as in the
    reference.  ``windows_per_forward``: windows scored by one forward as a disjoint-union graph (1: one forward per window, as
    the reference runs them).  Host synchronisations: 3 per scene for the cache (sizes), 1 for the kept-edge count, 1 for the
    tracks."""
    from .clr_att_gnn import GNN, EmbeddingCache, window_node_ids
    ids_of = ids_of or window_node_ids
    wins = [w for w in windows if w.pose_feats.size(0) > 0 and w.edge_index.size(1) > 0]
    if not wins:
        raise ValueError("no window of this scene has nodes and edges")
    dev = wins[0].pose_feats.device
    gids = [ids_of(w).to(dev) for w in wins]
    tables = None
    cache_obj = None
    if isinstance(model, GNN) and cache is not False:
        cache_obj = cache if isinstance(cache, EmbeddingCache) else EmbeddingCache()
        # every module's own flag is restored: `model.train(was_training)` would put a sub-module the caller (or the sticky
        # `.eval()` switch of clr_att_gnn.py:128-139) had left in eval mode back into train mode
        modes = [(mod, mod.training) for mod in model.modules()]
        model.eval()
        try:
            cache_obj.add_windows(model, wins, ids_of=ids_of)
            tables = cache_obj.scene_tables(gids)
        finally:
            for mod, flag in modes:
                mod.training = flag
    pairs, scores = [], []
    with torch.no_grad():
        for b0 in range(0, len(wins), max(1, windows_per_forward)):
            grp = list(range(b0, min(b0 + max(1, windows_per_forward), len(wins))))
            enc = EmbeddingCache.tables_slice(tables, grp[0], grp[-1] + 1) if tables is not None else None
            if len(grp) == 1:
                w = wins[grp[0]]
            else:
                # several windows as ONE disjoint-union graph per forward (what train.py's loader does with its batches,
                # train.py:88-96): windows do not interact -- every kernel is per node / per edge / per node's own edge list --
                # so the scores are those of one forward per window, at a fraction of the launches.  Frames of different windows
                # are kept apart for the (discarded) frame-wise k-NN block by an offset on the timestamps.
                w = _union_graph([wins[k] for k in grp], with_sensor_feats=enc is None)
            out = model(w, encoded=enc)[0] if enc is not None else model(w)[0]
            scores.append(out.reshape(-1).float())
            for k in grp:
                g, wk = gids[k], wins[k]
                pairs.append(torch.stack([g[wk.edge_index[0]], g[wk.edge_index[1]]], 1))
    res = greedy_edges_hip(torch.cat(pairs), torch.cat(scores), node_class, class_names, thresholds)
    res["windows"] = len(wins)
    res["cache"] = cache_obj
    if tracks:
        # greedy edges in the reference's order (predict.py:246-259: node by node, its outgoing then its incoming edge; a dict,
        # so an edge that is both keeps its first position) -- vectorised: an edge (a, b) enters at min(2 a | succ[a] == b,
        # 2 b + 1 | pred[b] == a)
        import numpy as np
        kp, ks = res["kept_pairs"].cpu().numpy(), res["kept_scores"].cpu().numpy()
        pred, succ = res["pred"].cpu().numpy(), res["succ"].cpu().numpy()
        big = np.iinfo(np.int64).max
        t_out = np.where(succ[kp[:, 0]] == kp[:, 1], 2 * kp[:, 0], big)
        t_in = np.where(pred[kp[:, 1]] == kp[:, 0], 2 * kp[:, 1] + 1, big)
        t = np.minimum(t_out, t_in)
        sel = np.nonzero(t < big)[0]
        sel = sel[np.argsort(t[sel], kind="stable")]
        res["pred_edge_pairs"], res["pred_edge_scores"] = kp[sel], ks[sel]       # (arrays: a scene has tens of thousands)
        res["tracks"] = tracks_from_arrays(kp[sel], ks[sel], node_class.cpu().numpy(), class_names, join_score)
    return res
