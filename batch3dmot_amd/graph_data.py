"""Window loader: the reference's ``GraphDataset`` (utils/graph_data.py:22-257) without its per-edge
Python loops, plus a prefetching batch iterator (SURVEY.md section 8f #2 -- the input side of the path).

On-disk format, one window = ten files written by the graph construction tools
(preprocessing/construct_detection_graphs_parallel.py:623-650)::

    {dir}{scene_token}_len{L}_{i}_pose_features.pth   [N,19] f32      _edges.pth          [E,2] int64
    ..._img_features.pth   [N,3,32,32]                                 _gt.pth             [1,E]
    ..._lidar_features.pth [N,128,3]                                   _edge_features.pth  [E,4] f64
    ..._radar_features.pth [N,64,4]                                    _node_timestamps.pth [N]
    ..._node_boxes.pth (inference only)                                _node_metadata.json  {"0": {...}, ...}

What ``__getitem__`` returns is what the reference returns (graph_data.py:230-257): a ``Data`` with
``pose_feats, img_feats, lidar_feats, radar_feats, edge_index [2,E], edge_attr, y, node_timestamps,
edge_weights, edge_classes, node_classes, num_nodes, batch_idx`` and, for inference,
``global_edge_index [2,E], global_node_timestamps [N,2], boxes`` together with ``str(global_node_metadata)``.

What differs is how: the reference walks every edge in Python, looking two category names up in the JSON
and calling ``cb_scaling_factor`` (graph_data.py:194-228, ~30 us per edge -- 1 s per 30 k-edge window);
here the JSON is read once into a class-id vector and a global-id vector and the three per-edge loops
are three gathers (``data.class_balanced_edge_weights``).  Modality files a poses-only model never
reads can be skipped (``modalities=``), which the reference cannot do.
"""
from __future__ import annotations

import json
import os
import queue
import threading
from typing import Iterable, Iterator, List, Optional, Sequence

import torch

from .data import Data, class_balanced_edge_weights, collate

# graph_data.py:60-67 (scripts/statistics.py over the training split)
REL_FREQ_TRAIN = {"bicycle": 0.07455396870915335, "bus": 0.013947840246335299, "car": 0.44736907722651076,
                  "motorcycle": 0.055813302136334404, "pedestrian": 0.1980141158741746,
                  "trailer": 0.06407160593555014, "truck": 0.14623008987194142}
# pose_config.yaml:122-129 (classes.nuscenes_tracking_eval)
CLASS_DICT = {"car": 1, "truck": 2, "bus": 3, "trailer": 4, "pedestrian": 5, "motorcycle": 6, "bicycle": 7}

_MODALITY_FILES = {"img": "_img_features.pth", "lidar": "_lidar_features.pth", "radar": "_radar_features.pth"}


def _get(params, path: str, default):
    """params.main.slice_factor-style lookup on a ParamLib-like object or a nested dict."""
    cur = params
    for key in path.split("."):
        if cur is None:
            return default
        cur = cur.get(key) if isinstance(cur, dict) else getattr(cur, key, None)
    return default if cur is None else cur


class GraphDataset:
    """Same constructor arguments and file naming as the reference (graph_data.py:25-58).

    ``params`` may be the reference's ``ParamLib``, any object / nested dict with ``main.slice_factor``,
    ``main.class_dict``, ``gnn.batch_size_graph`` and ``classes.<name>``, or ``None`` (reference defaults).
    """

    def __init__(self, params, scenes: Sequence[dict], graph_data_dir: str, batch_size_graph: int, inference: bool,
                 modalities: Iterable[str] = ("img", "lidar", "radar"), edge_weighting: bool = True):
        self.params = params
        self.scenes = scenes
        self.inference = inference
        self.batch_size_graph = batch_size_graph
        self.edge_weighting = edge_weighting
        self.modalities = tuple(modalities)
        self.rel_freq_train = dict(REL_FREQ_TRAIN)
        name = _get(params, "main.class_dict", "nuscenes_tracking_eval")
        cd = _get(params, "classes." + name, None)
        self.class_dict = dict(vars(cd)) if cd is not None and not isinstance(cd, dict) else dict(cd or CLASS_DICT)
        slice_factor = int(_get(params, "main.slice_factor", 1))
        len_tag = str(_get(params, "gnn.batch_size_graph", batch_size_graph))
        self.batches: List[str] = []
        for scene in scenes[0::slice_factor]:
            num_batches = int(scene["nbr_samples"]) - self.batch_size_graph          # graph_data.py:49
            for batch_no in range(0, num_batches):
                self.batches.append(graph_data_dir + str(scene["token"]) + "_len" + len_tag + "_" + str(batch_no))
        n_cls = max(self.class_dict.values()) + 1
        self._freq = torch.zeros(n_cls, dtype=torch.float64)
        for cls_name, idx in self.class_dict.items():
            self._freq[idx] = self.rel_freq_train[cls_name]

    def get_metadata(self):
        return self.batches, self.scenes

    def cb_scaling_factor(self, edge_class: str) -> float:
        """graph_data.py:126-138."""
        num_edges = 5
        beta = (num_edges - 1) / num_edges
        return (1 - beta) / (1 - beta ** (num_edges * self.rel_freq_train[edge_class]))

    def __len__(self):
        return len(self.batches)

    def __getitem__(self, idx):
        stem = self.batches[idx]
        pose_features = torch.load(stem + "_pose_features.pth")
        feats = {}
        for m, suffix in _MODALITY_FILES.items():
            feats[m] = torch.load(stem + suffix) if m in self.modalities else None
        node_timestamps = torch.load(stem + "_node_timestamps.pth")
        edge_features = torch.load(stem + "_edge_features.pth")
        edges = torch.load(stem + "_edges.pth")
        gt = torch.load(stem + "_gt.pth")
        with open(stem + "_node_metadata.json", "r") as fh:
            node_metadata = json.load(fh)
        n = pose_features.shape[0]
        edge_index = edges.t().contiguous()

        if self.edge_weighting:
            # graph_data.py:194-228 as gathers; nodes no edge touches keep class 0, like the reference's zeros()
            names = [node_metadata[str(i)]["category_name"] for i in range(n)]
            node_class = torch.tensor([self.class_dict[c] for c in names], dtype=torch.long)
            weights, edge_classes, node_classes = class_balanced_edge_weights(edge_index, node_class, self._freq)
        else:
            # the reference leaves edge_classes / node_classes undefined on this branch (NameError at :237-238)
            weights = torch.ones(edges.shape[0])
            edge_classes = torch.zeros(edges.shape[0])
            node_classes = torch.zeros(n)

        data = Data(pose_feats=pose_features, img_feats=feats["img"], lidar_feats=feats["lidar"],
                    radar_feats=feats["radar"], edge_index=edge_index, edge_attr=edge_features,
                    y=gt.t().contiguous(), node_timestamps=node_timestamps, edge_weights=weights,
                    edge_classes=edge_classes, node_classes=node_classes, batch_idx=idx)
        data.num_nodes = n
        if not self.inference:
            return data

        # graph_data.py:177-192, 244-255
        gid = torch.tensor([node_metadata[str(i)]["global_node_id"] for i in range(n)], dtype=edges.dtype)
        data.global_edge_index = gid[edges].t().contiguous()
        data.global_node_timestamps = torch.stack([gid.to(torch.float32), node_timestamps.to(torch.float32)], dim=1)
        data.boxes = torch.load(stem + "_node_boxes.pth")
        global_node_metadata = {node_metadata[str(i)]["global_node_id"]: node_metadata[str(i)] for i in range(n)}
        return data, str(global_node_metadata)


def iterate_batches(dataset, batch_size: int, shuffle: bool = False, generator: Optional[torch.Generator] = None,
                    device=None, prefetch: int = 2, drop_last: bool = False) -> Iterator[Data]:
    """``DataLoader(dataset, batch_size, shuffle)`` of train.py:88-96 for training windows: yields one collated
    ``Data`` per ``batch_size`` windows.  A worker thread loads and collates ``prefetch`` batches ahead; with a
    CUDA ``device`` the batch is pinned and copied with ``non_blocking=True`` on a copy stream, and the consumer's
    stream waits for that copy only -- the H2D transfer of batch k+1 overlaps the step on batch k."""
    order = torch.randperm(len(dataset), generator=generator).tolist() if shuffle else list(range(len(dataset)))
    chunks = [order[i:i + batch_size] for i in range(0, len(order), batch_size)]
    if drop_last and chunks and len(chunks[-1]) < batch_size:
        chunks.pop()
    dev = torch.device(device) if device is not None else None
    on_gpu = dev is not None and dev.type == "cuda"
    copy_stream = torch.cuda.Stream(dev) if on_gpu else None
    q: "queue.Queue" = queue.Queue(maxsize=max(1, prefetch))
    stop = threading.Event()

    def work():
        try:
            for chunk in chunks:
                if stop.is_set():
                    return
                items = [dataset[i] for i in chunk]
                batch = collate([it[0] if isinstance(it, tuple) else it for it in items])
                if on_gpu:
                    for k, v in list(batch.__dict__.items()):
                        if torch.is_tensor(v):
                            setattr(batch, k, v.pin_memory())
                    with torch.cuda.stream(copy_stream):
                        moved = batch.to(dev, non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(copy_stream)
                    q.put((moved, ev, batch))              # `batch` keeps the pinned source alive until consumed
                else:
                    q.put((batch.to(dev) if dev is not None else batch, None, None))
            q.put(None)
        except BaseException as exc:                      # surface loader errors in the consumer
            q.put(exc)

    t = threading.Thread(target=work, daemon=True)
    t.start()
    try:
        while True:
            item = q.get()
            if item is None:
                return
            if isinstance(item, BaseException):
                raise item
            batch, ev, _keep = item
            if ev is not None:
                cur = torch.cuda.current_stream(dev)
                cur.wait_event(ev)
                batch._b3d_ready_event = ev      # a side stream that reads the batch (GNN.mask_stream) waits for the copy too
                # The batch was allocated on the copy stream: without this the caching allocator would hand its blocks
                # back to the copy stream the moment the consumer drops the batch, and the worker's next H2D copy could
                # overwrite them while kernels of the consumer's (asynchronously enqueued) step are still reading.
                for v in batch.__dict__.values():
                    if torch.is_tensor(v) and v.is_cuda:
                        v.record_stream(cur)
            yield batch
    finally:
        stop.set()
        while t.is_alive():
            try:
                q.get_nowait()
            except queue.Empty:
                t.join(timeout=0.05)
