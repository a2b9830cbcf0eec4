"""Light attribute-bag graph container + collate.

The reference hands ``torch_geometric.data.Data`` objects to ``forward(data)``
(utils/graph_data.py:230-242, predict.py:182-190) and batches them with
``torch_geometric.loader.DataLoader`` (train.py:88-96).  torch_geometric is not a dependency
of this package: any object exposing the attributes works (a real PyG ``Data`` included);
this module supplies a minimal equivalent for tests, the benchmark and PyG-free callers.
"""
from __future__ import annotations

from typing import Iterable, List

import torch

_NODE_KEYS = ("pose_feats", "img_feats", "lidar_feats", "radar_feats", "node_timestamps",
              "node_classes", "boxes")
_EDGE_KEYS = ("edge_attr", "y", "edge_weights", "edge_classes")


class Data:
    """Attribute bag; every tensor attribute moves with ``.to(device)``."""

    def __init__(self, **kwargs):
        self.batch = None
        for k, v in kwargs.items():
            setattr(self, k, v)

    def keys(self):
        return [k for k, v in self.__dict__.items() if v is not None]

    def to(self, device, non_blocking: bool = False) -> "Data":
        out = Data()
        for k, v in self.__dict__.items():
            setattr(out, k, v.to(device, non_blocking=non_blocking) if torch.is_tensor(v) else v)
        ei = self.__dict__.get("edge_index")
        pf = self.__dict__.get("pose_feats")
        if torch.is_tensor(ei) and torch.is_tensor(pf) and not ei.is_cuda and ei.numel() and torch.device(device).type == "cuda":
            # the reference fails with an index error on an edge whose endpoint is not a node (pose_gnn.py:180); checking
            # the CPU copy here costs microseconds and spares the GPU graph build its read-back of the error counter
            lo, hi = int(ei.min()), int(ei.max())
            if lo < 0 or hi >= pf.size(0):
                raise ValueError(f"edge_index has endpoints outside [0, {pf.size(0)}): min {lo}, max {hi}")
            out._b3d_valid_edge_index = out.edge_index      # identity-checked by the models: valid for THIS tensor only
        return out

    @property
    def num_nodes(self) -> int:
        n = self.__dict__.get("_num_nodes")
        return int(n) if n is not None else int(self.pose_feats.size(0))

    @num_nodes.setter
    def num_nodes(self, n):
        self.__dict__["_num_nodes"] = n

    @property
    def num_edges(self) -> int:
        return int(self.edge_index.size(1))

    def __repr__(self):
        parts = [f"{k}={list(v.shape)}" for k, v in self.__dict__.items() if torch.is_tensor(v)]
        return "Data(" + ", ".join(parts) + ")"


def collate(graphs: Iterable[Data]) -> Data:
    """Concatenate graphs the way PyG's ``Batch.from_data_list`` does: node/edge tensors are
    concatenated along dim 0, ``edge_index`` gets the node offset of its graph added, and a
    ``batch`` vector maps nodes to graphs.  ``node_timestamps`` are NOT offset -- the reference's
    frame-wise k-NN therefore mixes graphs of a batch that share a timestamp value
    (pose_gnn.py:76-78 pass no ``batch`` vector)."""
    graphs = list(graphs)
    out = Data()
    off = 0
    ei: List[torch.Tensor] = []
    bvec: List[torch.Tensor] = []
    for g_idx, g in enumerate(graphs):
        n = g.pose_feats.size(0)
        ei.append(g.edge_index + off)
        bvec.append(torch.full((n,), g_idx, dtype=torch.long, device=g.pose_feats.device))
        off += n
    out.edge_index = torch.cat(ei, dim=1).contiguous()
    out.batch = torch.cat(bvec)
    for k in _NODE_KEYS + _EDGE_KEYS:
        vals = [getattr(g, k, None) for g in graphs]
        if all(v is not None for v in vals):
            setattr(out, k, torch.cat(vals, dim=0))
    out.num_graphs = len(graphs)
    return out


def class_balanced_edge_weights(edge_index: torch.Tensor, node_class: torch.Tensor, class_freq: torch.Tensor,
                                num_edges: int = 5):
    """Vectorised form of the loader's per-edge Python loop (reference utils/graph_data.py:126-138,
    194-228): ``edge_weights``, ``edge_classes`` and ``node_classes`` of a window in three tensor ops.

    ``edge_index`` [2,E] (row 0 past / source, row 1 current / destination), ``node_class`` [N] integer
    class id of every node (the reference's ``class_dict`` value, ids >= 1), ``class_freq`` [C+1] the
    relative training frequency per class id (``rel_freq_train``; index 0 unused).  Works on any device.

    weight_e = (1 - beta) / (1 - beta ** (num_edges * freq[class])), beta = (num_edges - 1) / num_edges,
    for an edge whose two nodes share a class.  The reference's branch for edges between different
    classes reads an attribute it never defines (graph_data.py:223) -- its graphs only link detections
    of one class -- so such an edge raises here as it does there.
    """
    src, dst = edge_index[0], edge_index[1]
    ca, cb = node_class[src], node_class[dst]
    if bool((ca != cb).any()):
        raise ValueError("edge between nodes of different classes: undefined in the reference (graph_data.py:223)")
    beta = (num_edges - 1) / num_edges
    factor = (1.0 - beta) / (1.0 - torch.pow(torch.full_like(class_freq, beta, dtype=torch.float64),
                                              num_edges * class_freq.double()))
    weights = factor[ca.long()].float()
    edge_classes = ca.float()
    node_classes = torch.zeros(node_class.numel(), dtype=torch.float32, device=node_class.device)
    node_classes[src] = ca.float()
    node_classes[dst] = cb.float()
    return weights, edge_classes, node_classes
