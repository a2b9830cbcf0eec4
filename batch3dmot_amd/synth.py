"""Seeded synthetic "nuScenes-shaped" tracking graphs (there is no dataset on the GPU box).

Follows the reference's graph construction
(preprocessing/construct_detection_graph_disjoint_parallel_only_poses.py:109-290 and
utils/graph_utils.py:7-89):

* a window of ``T`` frames, detections of 7 classes drawn from ``rel_freq_train``
  (utils/graph_data.py:61-68), ego-frame centres in the 1-50 m annulus (pose_config.yaml:49-50);
* ``pose_feats [N,19] = [x,y,z, w,l,h, yaw, vx,vy,vz, onehot(7), score, rel_frame]``
  (...only_poses.py:159-186);
* every detection of frame t>=1 is linked to its ``min(K, #same-class past detections)`` nearest
  past detections under ``1/2 d/max + 1/4 |dyaw|/max + 1/4 |dv|/max`` (graph_utils.py:67-83),
  direction past -> current, emitted destination-ascending then distance-ascending
  (...only_poses.py:206-224);
* ``edge_attr [E,4] float64 = [centre distance, |dyaw|, log volume ratio, frame delta]``
  (graph_utils.py:19-30, ...only_poses.py:263-267);
* ``y`` marks the temporally closest edge between two detections of one synthetic track;
* ``edge_weights`` are the class-balanced factors of utils/graph_data.py:126-138.

Camera / LiDAR / radar tensors for the CLR model: ``img_feats [N,3,32,32]`` in [0,1),
``lidar_feats [N,128,3]`` (storage of a [3,128] cloud, 30 % all-zero rows) and
``radar_feats [N,64,4]`` (75 % all-zero rows).
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from .data import Data, collate

CLASSES = ("car", "truck", "bus", "trailer", "pedestrian", "motorcycle", "bicycle")  # ids 1..7
REL_FREQ_TRAIN = {"bicycle": 0.07455396870915335, "bus": 0.013947840246335299,
                  "car": 0.44736907722651076, "motorcycle": 0.055813302136334404,
                  "pedestrian": 0.1980141158741746, "trailer": 0.06407160593555014,
                  "truck": 0.14623008987194142}
_WLH = {"car": (1.95, 4.6, 1.7), "truck": (2.5, 6.9, 2.8), "bus": (2.9, 11.0, 3.4),
        "trailer": (2.9, 12.3, 3.9), "pedestrian": (0.67, 0.73, 1.77),
        "motorcycle": (0.77, 2.1, 1.5), "bicycle": (0.6, 1.7, 1.3)}
SEED_BASE = 5621  # gnn.manual_seed, pose_config.yaml:96


def cb_scaling_factor(cls_name: str) -> float:
    """utils/graph_data.py:126-138."""
    num_edges = 5
    beta = (num_edges - 1) / num_edges
    return (1 - beta) / (1 - beta ** (num_edges * REL_FREQ_TRAIN[cls_name]))


def make_graph(num_nodes: int = 1500, target_edges: Optional[int] = None, k: int = 40,
               frames: int = 5, graph_idx: int = 0, window_start: int = 0,
               modalities: bool = False, lidar_frac: float = 0.7, radar_frac: float = 0.25,
               max_frame_gap: Optional[int] = None) -> Data:
    """``max_frame_gap``: candidate edges reach at most this many frames back (a whole SCENE built in one call, whose 5-frame
    windows -- ``scene_windows`` -- then hold the edges a per-window construction would give them)."""
    g = torch.Generator().manual_seed(SEED_BASE + graph_idx)
    n_t = num_nodes // frames
    # ---- synthetic tracks: objects observed in most frames, constant velocity + noise ----------
    n_obj = max(1, int(round(n_t / 0.85)))
    prior = torch.tensor([REL_FREQ_TRAIN[c] for c in CLASSES])
    obj_cls = torch.multinomial(prior, n_obj, replacement=True, generator=g)          # 0..6
    rad = torch.sqrt(torch.rand(n_obj, generator=g) * (50.0 ** 2 - 1.0) + 1.0)
    ang = torch.rand(n_obj, generator=g) * 2 * math.pi
    pos0 = torch.stack([rad * torch.cos(ang), rad * torch.sin(ang), torch.randn(n_obj, generator=g)], 1)
    vel = torch.cat([torch.randn(n_obj, 2, generator=g) * 3.0, torch.zeros(n_obj, 1)], 1)
    wlh0 = torch.tensor([_WLH[CLASSES[c]] for c in obj_cls.tolist()])
    wlh0 = wlh0 * (0.85 + 0.3 * torch.rand(n_obj, 3, generator=g))
    yaw0 = (torch.rand(n_obj, generator=g) * 2 - 1) * math.pi

    rows, ts, oid = [], [], []
    for t in range(frames):
        order = torch.randperm(n_obj, generator=g)[:n_t]
        p = pos0[order] + vel[order] * (0.5 * t) + torch.randn(n_t, 3, generator=g) * 0.15
        yaw = yaw0[order] + torch.randn(n_t, generator=g) * 0.05
        v = vel[order] + torch.randn(n_t, 3, generator=g) * torch.tensor([0.3, 0.3, 0.0])
        score = 0.05 + 0.95 * torch.rand(n_t, generator=g)
        onehot = torch.nn.functional.one_hot(obj_cls[order], 7).float()
        rows.append(torch.cat([p, wlh0[order], yaw[:, None], v, onehot, score[:, None],
                               torch.full((n_t, 1), float(t))], 1))
        ts.append(torch.full((n_t,), window_start + t, dtype=torch.long))
        oid.append(order)
    pose = torch.cat(rows, 0).float()
    node_ts = torch.cat(ts)
    obj = torch.cat(oid)
    cls = obj_cls[obj]
    n = pose.size(0)
    rel = (node_ts - window_start)

    # ---- candidate lists: same class, strictly earlier frame --------------------------------
    pos, yaw, velv = pose[:, :3].double(), pose[:, 6].double(), pose[:, 7:10].double()
    vol = pose[:, 3:6].double().prod(1)
    cand_d, cand_src, cand_dst = [], [], []
    for t in range(1, frames):
        for c in range(7):
            cur = torch.nonzero((rel == t) & (cls == c)).squeeze(1)
            past = torch.nonzero((rel < t) & (cls == c) & ((rel >= t - max_frame_gap) if max_frame_gap else (rel < t))).squeeze(1)
            if cur.numel() == 0 or past.numel() == 0:
                continue
            d3 = torch.cdist(pos[cur], pos[past])
            dyaw = (yaw[cur][:, None] - yaw[past][None]).abs()
            dyaw = torch.minimum(dyaw % (2 * math.pi), 2 * math.pi - dyaw % (2 * math.pi))
            dv = torch.cdist(velv[cur], velv[past])
            nrm = lambda m: m / m.max(1, keepdim=True).values.clamp_min(1e-12)  # noqa: E731
            md = 0.5 * nrm(d3) + 0.25 * nrm(dyaw) + 0.25 * nrm(dv)
            cand_d.append(md.reshape(-1))
            cand_src.append(past[None].expand(cur.numel(), -1).reshape(-1))
            cand_dst.append(cur[:, None].expand(-1, past.numel()).reshape(-1))
    if not cand_d:
        raise ValueError("graph too small: no candidate edges")
    cd, cs, cdst = torch.cat(cand_d), torch.cat(cand_src), torch.cat(cand_dst)
    # rank of every candidate inside its destination's list (ascending distance)
    key = cdst.double() * 4.0 + cd  # md <= 1 < 4 keeps destinations separated
    perm = torch.argsort(key, stable=True)
    cd, cs, cdst = cd[perm], cs[perm], cdst[perm]
    counts = torch.bincount(cdst, minlength=n)
    start = torch.cumsum(counts, 0) - counts
    rank = torch.arange(cdst.numel()) - start[cdst]
    if target_edges is not None:
        # the K whose edge count is closest to the target
        best = None
        for kk in range(1, int(counts.max()) + 1):
            n_e = int(torch.minimum(counts, torch.tensor(kk)).sum())
            if best is None or abs(n_e - target_edges) < best[0]:
                best = (abs(n_e - target_edges), kk)
            if n_e >= target_edges:
                break
        k = best[1]
    keep = rank < k
    src, dst = cs[keep], cdst[keep]
    edge_index = torch.stack([src, dst]).long().contiguous()

    # ---- edge features (float64), labels, class-balanced weights ----------------------------
    d_c = (pos[src] - pos[dst]).norm(dim=1)
    dy = (yaw[src] - yaw[dst]).abs()
    dy = torch.minimum(dy % (2 * math.pi), 2 * math.pi - dy % (2 * math.pi))
    lv = torch.log(vol[src] / vol[dst])
    dt = (rel[dst] - rel[src]).double()
    edge_attr = torch.stack([d_c, dy, lv, dt], 1)
    same = obj[src] == obj[dst]
    # temporally closest same-track edge per destination
    best = torch.full((n,), 10 ** 6, dtype=torch.long)
    best.scatter_reduce_(0, dst[same], (rel[dst] - rel[src])[same], reduce="amin")
    y = (same & ((rel[dst] - rel[src]) == best[dst])).long()
    w_cls = torch.tensor([cb_scaling_factor(c) for c in CLASSES], dtype=torch.float32)
    data = Data(pose_feats=pose, edge_index=edge_index, edge_attr=edge_attr, y=y,
                node_timestamps=node_ts, edge_weights=w_cls[cls[dst]],
                edge_classes=(cls[dst] + 1).float(), node_classes=(cls + 1).float())
    data.track_id = obj
    if modalities:
        data.img_feats = torch.rand(n, 3, 32, 32, generator=g)
        lid = torch.zeros(n, 3, 128)
        has_l = torch.rand(n, generator=g) < lidar_frac
        npts = torch.randint(6, 129, (n,), generator=g)
        cloud = torch.randn(n, 3, 128, generator=g)
        cloud = cloud / cloud.norm(dim=1, keepdim=True).max(dim=2, keepdim=True).values
        mask = (torch.arange(128)[None] < npts[:, None]) & has_l[:, None]
        lid = cloud * mask[:, None, :]
        data.lidar_feats = lid.reshape(n, 128, 3).contiguous()   # storage of a [3,128] cloud
        has_r = torch.rand(n, generator=g) < radar_frac
        rpts = torch.randint(2, 65, (n,), generator=g)
        rc = torch.randn(n, 4, 64, generator=g)
        rmask = (torch.arange(64)[None] < rpts[:, None]) & has_r[:, None]
        data.radar_feats = (rc * rmask[:, None, :]).reshape(n, 64, 4).contiguous()
    return data


def make_batch(num_graphs: int = 2, nodes_per_graph: int = 1500, edges_per_graph: Optional[int] = 15000,
               first_graph_idx: int = 0, modalities: bool = False, frames: int = 5, k: int = 40) -> Data:
    """A collated batch (reference train.py:86-90 uses batch_size 2)."""
    graphs = [make_graph(nodes_per_graph, edges_per_graph, k=k, frames=frames,
                         graph_idx=first_graph_idx + i, modalities=modalities)
              for i in range(num_graphs)]
    return collate(graphs)


def scene_windows(scene: Data, frames: int, per_frame: int, size: int = 5):
    """Overlapping windows of `size` frames with stride 1 (predict.py:172): nodes are ordered by frame, a window is
    a contiguous node range; edges with both ends inside, re-indexed; rel_frame (pose_feats[:, 18]) restarts at 0."""
    out = []
    for b in range(frames - size + 1):
        lo, hi = b * per_frame, (b + size) * per_frame
        keep = (scene.edge_index[0] >= lo) & (scene.edge_index[1] >= lo) & (scene.edge_index[0] < hi) & (scene.edge_index[1] < hi)
        pose = scene.pose_feats[lo:hi].clone()
        pose[:, 18] -= b
        w = Data(pose_feats=pose, edge_index=(scene.edge_index[:, keep] - lo).contiguous(), edge_attr=scene.edge_attr[keep].clone(),
                 node_timestamps=scene.node_timestamps[lo:hi].clone())
        for k in ("img_feats", "lidar_feats", "radar_feats"):
            if getattr(scene, k, None) is not None:
                setattr(w, k, getattr(scene, k)[lo:hi].clone())
        w.global_ids = torch.arange(lo, hi)
        out.append(w)
    return out


def make_scene(frames: int = 40, per_frame: int = 400, k: int = 14, scene_idx: int = 0, modalities: bool = True, window: int = 5):
    """A synthetic scene of ``frames`` frames x ``per_frame`` detections whose ``window``-frame windows (stride 1, predict.py:172)
    have ~``window * per_frame`` nodes and ~10 edges per node: returns (scene Data, list of window Data with ``global_ids``)."""
    scene = make_graph(frames * per_frame, None, k=k, frames=frames, graph_idx=7000 + scene_idx, modalities=modalities,
                       max_frame_gap=window - 1)
    return scene, scene_windows(scene, frames, per_frame, size=window)


def write_window_files(stem, seed, n_per_frame=12, global_offset=1000):
    """One synthetic window in the reference's on-disk layout (construct_detection_graphs_parallel.py:623-650), plus one
    isolated node, which no edge touches.  Returns the number of nodes."""
    import json
    from .graph_data import CLASS_DICT
    INV_CLASS = {v: k for k, v in CLASS_DICT.items()}
    g = make_graph(5 * n_per_frame, 40 * n_per_frame, graph_idx=seed, modalities=True)
    n = g.pose_feats.size(0)
    cls = g.node_classes.long().clone()
    # append an isolated node: its class never reaches node_classes in the reference's loop
    pose = torch.cat([g.pose_feats, g.pose_feats[:1]])
    ts = torch.cat([g.node_timestamps, g.node_timestamps[:1]])
    torch.save(pose, stem + "_pose_features.pth")
    torch.save(torch.cat([g.img_feats, g.img_feats[:1]]), stem + "_img_features.pth")
    torch.save(torch.cat([g.lidar_feats, g.lidar_feats[:1]]), stem + "_lidar_features.pth")
    torch.save(torch.cat([g.radar_feats, g.radar_feats[:1]]), stem + "_radar_features.pth")
    torch.save(ts, stem + "_node_timestamps.pth")
    torch.save(g.edge_attr, stem + "_edge_features.pth")
    torch.save(g.edge_index.t().contiguous(), stem + "_edges.pth")
    torch.save(g.y.reshape(1, -1), stem + "_gt.pth")
    torch.save(torch.arange((n + 1) * 7, dtype=torch.float32).reshape(n + 1, 7), stem + "_node_boxes.pth")
    gen = torch.Generator().manual_seed(seed)
    gids = (torch.randperm(5 * (n + 1), generator=gen)[: n + 1] + global_offset).tolist()
    meta = {}
    for i in range(n + 1):
        c = int(cls[i]) if i < n and int(cls[i]) > 0 else 1
        meta[str(i)] = {"category_name": INV_CLASS[c], "global_node_id": gids[i], "score": 0.5, "token": f"t{i}"}
    with open(stem + "_node_metadata.json", "w") as fh:
        json.dump(meta, fh)
    return n + 1
