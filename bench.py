#!/usr/bin/env python3
"""Headline benchmark: edges/sec (fwd+bwd) on nuScenes-shaped detection graphs (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (default, BASELINE.json `configs[3]`, the configuration the 1/2/4/8-GPU metric is quoted on): the
camera+LiDAR+radar GNN (clr_att_gnn.py:16-188, depth 6) training step of train.py:124-160 -- modality masks,
the three frozen encoders in train mode (BatchNorm with batch statistics, clr_att_gnn.py:26-33), graph structure
build, forward, class-balanced BCE, backward, gradient all-reduce (N > 1), Adam -- on a collated batch of 2 synthetic
graphs (1,500 nodes / ~15,500 edges each, T = 5 frames): ~3,000 nodes / ~31,000 edges per GPU.  Inputs are resident
in HBM before the timed region.  Weak scaling: every rank processes its own batches; the only collective is the flat
all-reduce of the 1,310,193 gradient-carrying parameters (5.24 MB).

    --model pose        `configs[1]`: pose_config.yaml poses-only PoseGNN training step (round-1 headline)
    --encoders precomputed   the camera+LiDAR+radar step with the encoder outputs given
    --modalities cl     `configs[2]`: radar rows all zero ("camera+LiDAR")
    --scaling strong    SURVEY.md 8d(4): the global batch is fixed at 16 graphs, every rank steps 16 / N of them
    --mode infer        `configs[4]`: forward only under no_grad, 64 windows of 2,000 nodes / ~20,000 edges in flight
                        per GPU (predict.py:172-196; replicas only, no collective) + the stand-alone k-NN + GAT block
                        at n_t = 400, D = 96, k = 20

At N = 1 the default run also reports both of those as `secondary` figures (shorter timed regions).

Prints ONE JSON line (rank 0).  `roofline` describes the kernel family with the largest summed device time, measured
with HIP events on the launch stream (b3d_prof_*); `cpu_baseline` times the CPU oracle (the restated reference
path) on the host cores, rank 0, N = 1 only.

Timed region: hipGraph replays (one captured step per pool batch; what cannot be captured -- the modality masks' row
compaction, whose counts are shapes -- runs eagerly in front of each replay and feeds it).  At N > 1 a step is two
replays around the eager gradient all-reduce: graph A = forward + backward (writes the flat gradient buffer), the flat
all-reduce, graph B = the optimizer step (host enqueue of the eager step, ~10 ms, would otherwise bound the N > 1
figure, not the GPU).  Event records cannot be captured, so the kernel families are timed in an eager pass of the same
K steps right after the timed region.  `--no-graph` enqueues every step eagerly.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md, v_mfma_f32_16x16x4_f32 (= fp32 vector peak)
PEAK_BF16_MFMA_TFLOPS = 2500.0    # dense bf16 MFMA (MI355X_MICROARCH.md); a bf16x6 product costs six bf16 MFMA MACs
PEAK_BF16X6_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0     # = 416.7 fp32-equivalent TFLOP/s
PEAK_HBM_GBS = 8000.0
BOUND = {"mp_edge_fwd": "mfma", "mp_edge_bwd": "mfma", "wgrad_edge": "hbm", "mp_node_fwd": "hbm", "mp_node_bwd": "hbm",
         "att_fwd": "mfma", "att_bwd": "mfma", "point_feat": "mfma"}
# the kernel families of the message-passing path itself; `roofline` names the one with the most device
# time.  The encoder family (8f #1, next to the path) is reported as a labelled secondary (`roofline_encoders`).
PATH_FAMILIES = ("mp_edge_fwd", "mp_edge_bwd", "wgrad_edge", "mp_node_fwd", "mp_node_bwd", "att_fwd", "att_bwd")
DTYPE = "f32 (bf16x6 split products, fp32 accumulate)"     # what the wide layers compute in: b3d_dev.hpp


# ---- work models -------------------------------------------------------------------------------------------------
class PoseWork:
    """pose_gnn.py:94-120 widths.  MACs per row of one CausalMessagePassing layer."""
    name = "pose"
    MAC_EU, MAC_MSG, MAC_NODE = 128 * 96 + 96 * 64 + 64 * 32, 2 * (128 * 96 + 96 * 64), 128 * 96 + 96 * 64 + 64 * 48
    # executed: the node columns of the three first layers are evaluated per node (csrc/b3d_hoist.hpp)
    X_EU, X_MSG = 32 * 96 + 96 * 64 + 64 * 32, 2 * (32 * 96 + 96 * 64)
    X_NODE_TAB, X_NODE_GP = 48 * 384, 384 * 96

    @staticmethod
    def step_bytes(n, e):       # fwd = 1,572 E + 9,868 N; fwd+bwd = 3x
        return 3.0 * (1572.0 * e + 9868.0 * n)

    @staticmethod
    def step_flops(n, e, **_):  # fwd = E * 690,824 + N * 264,144 (2 * MAC); bwd = 2 * fwd
        return 3.0 * (690824.0 * e + 264144.0 * n)

    @classmethod
    def families(cls, n, e, depth, training=True, **_):
        """Per STEP (see ClrWork.families); every layer of this model runs on the exact fp32 MFMA.  training=False: the forward
        stores no hidden activations."""
        # mp_edge_bwd: depth launches -- the last layer's message stacks carry no gradient (edge_update only there)
        alg = {"mp_edge_fwd": 2.0 * (cls.MAC_EU + cls.MAC_MSG) * e * depth, "mp_edge_bwd": 2.0 * (cls.MAC_EU * depth + cls.MAC_MSG * (depth - 1)) * e,
               "wgrad_edge": 2.0 * (e * (cls.MAC_EU * depth + cls.MAC_MSG * (depth - 1)) + n * cls.MAC_NODE * (depth - 1)),
               "mp_node_fwd": 2.0 * cls.MAC_NODE * n * depth, "mp_node_bwd": 2.0 * cls.MAC_NODE * n * (depth - 1)}
        exe = {"mp_edge_fwd": 2.0 * (cls.X_EU + cls.X_MSG) * e * depth, "mp_edge_bwd": 2.0 * (cls.X_EU * depth + cls.X_MSG * (depth - 1)) * e,
               "wgrad_edge": 2.0 * (e * (cls.X_EU * depth + cls.X_MSG * (depth - 1))
                                    + n * (cls.MAC_NODE * (depth - 1) + 2 * 96 * 48 * depth + 4 * 96 * 48 * (depth - 1))),
               "mp_node_fwd": 2.0 * (cls.MAC_NODE * depth + cls.X_NODE_TAB * (depth - 1)) * n,
               "mp_node_bwd": 2.0 * (cls.MAC_NODE + cls.X_NODE_GP) * n * (depth - 1)}
        byts = {"mp_edge_fwd": depth * e * (8 + 4 * (32 + 32 + 64 + 64 + (352 if training else 0))),      # idx, e in/out, fut, past, saved hidden
                "mp_edge_bwd": (depth - 1) * e * (8 + 4 * (32 + 32 + 352 + 192 + 384))    # de out/in, saved, per-edge node grads, G
                               + e * (8 + 4 * (32 + 32 + 160 + 192)),                      # last layer: edge_update only
                "wgrad_edge": e * 4 * ((192 + 160 + 64) * depth + (192 + 128 + 192 + 32) * (depth - 1)),
                "mp_node_fwd": depth * (e * 4 * 128 + n * 4 * (128 + 48 + 160)),
                "mp_node_bwd": (depth - 1) * (e * 4 * 192 + n * 4 * (128 + 48 + 48 + 160 + 208))}
        return alg, exe, byts, {}


class ClrWork:
    """clr_att_gnn.py:196-222 widths (+ att_edge_encoder :81-91)."""
    name = "clr"
    MAC_EU, MAC_MSG = 320 * 256 + 256 * 128 + 128 * 64, 2 * (256 * 192 + 192 * 128)            # 122,880 + 147,456
    MAC_NODE = 256 * 192 + 192 * 128 + 128 * 96                                                  # 86,016
    MAC_ATT = 640 * 512 + 512 * 384 + 384 * 256 + 256 * 128 + 128 * 64                           # 663,552
    # executed once the node columns of a first layer are evaluated per node: edge_update.0 keeps e | att (128
    # columns), the message stacks keep e' (64), att_edge_encoder.0 keeps e (64)
    X_EU, X_MSG = 128 * 256 + 256 * 128 + 128 * 64, 2 * (64 * 192 + 192 * 128)
    X_ATT = 64 * 512 + 512 * 384 + 384 * 256 + 256 * 128 + 128 * 64
    X_NODE_TAB = 96 * (2 * 256 + 2 * 192)          # per-node table of the next layer's first layers
    X_NODE_GP = (2 * 256 + 2 * 192) * 96 + 2 * 192 * 96
    X_ATT_NODE = 2 * 288 * 512                     # per-node parts of att_edge_encoder.0
    MAC_POINT_L = 128 * (3 * 64 + 64 * 128 + 128 * 1024)      # one PointNet stack (STN or trunk) per LiDAR cloud
    MAC_POINT_R = 64 * (4 * 64 + 64 * 128 + 128 * 1024)

    @staticmethod
    def step_bytes(n, e):       # fwd ~ 4,900 E + 20,500 N; x3
        return 3.0 * (4900.0 * e + 20500.0 * n)

    @staticmethod
    def step_flops(n, e, f_l=0.7, f_r=0.25, **_):   # SURVEY.md 8d (attention hoisted, encoders excluded)
        return 3.0 * (4581776.0 * e + n * 2.0 * (521616 + 59392 + 73728 * f_l + 81920 * f_r))

    @classmethod
    def families(cls, n, e, depth, hoist_mp=False, hoist_att=False, nl=0, nr=0, training=True, **_):
        """Per STEP: algorithmic FLOPs (the reference's per-edge arithmetic), executed FLOPs (what the
        kernels multiply: node columns of hoisted first layers are per-node work), algorithmic bytes, and the fraction of
        the executed MACs that run as bf16x6 (layers with 64..256 inputs in multiples of 32)."""
        eu, msg = (cls.X_EU, cls.X_MSG) if hoist_mp else (cls.MAC_EU, cls.MAC_MSG)
        att = cls.X_ATT if hoist_att else cls.MAC_ATT
        tab = cls.X_NODE_TAB if hoist_mp else 0
        gp = cls.X_NODE_GP if hoist_mp else 0
        att_node = cls.X_ATT_NODE if hoist_att else 0
        alg = {"mp_edge_fwd": 2.0 * (cls.MAC_EU + cls.MAC_MSG) * e * depth,
               "mp_edge_bwd": 2.0 * (cls.MAC_EU * depth + cls.MAC_MSG * (depth - 1)) * e,  # depth launches: the last layer's message stacks carry no gradient
               "wgrad_edge": 2.0 * (e * (cls.MAC_EU * depth + cls.MAC_MSG * (depth - 1) + cls.MAC_ATT) + n * cls.MAC_NODE * (depth - 1)),
               "mp_node_fwd": 2.0 * cls.MAC_NODE * n * depth, "mp_node_bwd": 2.0 * cls.MAC_NODE * n * (depth - 1),
               "att_fwd": 2.0 * cls.MAC_ATT * e, "att_bwd": 2.0 * cls.MAC_ATT * e,
               "point_feat": 2.0 * (2 * cls.MAC_POINT_L * nl + cls.MAC_POINT_R * nr)}
        exe = {"mp_edge_fwd": 2.0 * (eu + msg) * e * depth, "mp_edge_bwd": 2.0 * (eu * depth + msg * (depth - 1)) * e,
               "wgrad_edge": 2.0 * (e * (eu * depth + msg * (depth - 1) + att) + n * ((cls.MAC_NODE + tab) * (depth - 1) + att_node)),
               "mp_node_fwd": 2.0 * (cls.MAC_NODE * depth + tab * (depth - 1)) * n, "mp_node_bwd": 2.0 * (cls.MAC_NODE * (depth - 1) + gp * depth) * n,
               "att_fwd": 2.0 * (att * e + att_node * n), "att_bwd": 2.0 * (att * e + att_node * n),
               "point_feat": alg["point_feat"]}
        sav = (256 + 128 + 192 + 192) if training else 0       # hidden activations kept for the backward
        byts = {"mp_edge_fwd": depth * e * (8 + 4 * (64 + 64 + 64 + 2 * 128 + sav)),
                "mp_edge_bwd": (depth - 1) * e * (8 + 4 * (64 + 64 + 2 * 64 + sav + (256 + 128 + 64 + 192 + 192) + (0 if hoist_mp else 384)))
                               + e * (8 + 4 * (64 + 64 + 2 * 64 + (256 + 128) + (256 + 128 + 64))),
                "wgrad_edge": e * 4 * ((256 + 128 + 64 + 256 + 128 + 128) * depth + (2 * 192 + 2 * 128 + 2 * 192 + 64) * (depth - 1)
                                       + 2 * (512 + 384 + 256 + 128) + 64 + 64),
                "mp_node_fwd": depth * (e * 4 * 256 + n * 4 * (256 + 96 + 320)),
                "mp_node_bwd": (depth - 1) * (e * 4 * 384 + n * 4 * (256 + 96 + 96 + 320 + 416)),
                "att_fwd": e * 4 * (64 + 2 * (512 + 384 + 256 + 128) + 64),
                "att_bwd": e * 4 * (64 + 3 * (512 + 384 + 256 + 128) + 640),
                "point_feat": 4.0 * (2 * nl * (3 * 128 + 4 * 1024) + nr * (4 * 64 + 4 * 1024))}
        bf = {"mp_edge_fwd": 1.0 if hoist_mp else 0.0, "mp_edge_bwd": 1.0 if hoist_mp else 0.0, "point_feat": 0.999,
              "wgrad_edge": 1.0 if hoist_mp else 0.0,       # the cooperative kernel of the hoisted plan (csrc/b3d_wgemm.hpp)
              # every per-edge layer of att_edge_encoder runs as bf16x6 (384 / 512-input layers included); the per-node
              # parts of its first layer (att_node_linear) are exact-fp32 MFMA
              "att_fwd": cls.X_ATT * e / (cls.X_ATT * e + att_node * n) if hoist_att else 0.06,
              "att_bwd": cls.X_ATT * e / (cls.X_ATT * e + att_node * n) if hoist_att else 0.2}
        return alg, exe, byts, bf


# ---- workloads -----------------------------------------------------------------------------------------------------
class Workload:
    """One training step over a pool of HBM-resident batches.  `step(i)`: the whole step, eagerly.  `pre(i)` +
    `captured(i)`: the same step split into the part that cannot be captured into a hipGraph (runs eagerly in front of
    every replay and feeds it) and the part that can."""

    def __init__(self, kind, dev, rank, world, args, encoders="frozen", graphs=2, modalities="clr", ap_metrics=False):
        from batch3dmot_amd import encoders as enc_mod, synth
        # train.py:143-150 computes the average precision of the scores (overall + one per class: 8 torchmetrics calls, each a
        # sort and a host read) inside every iteration; with ap_metrics the step runs the one-call device form of it
        # (b3d_average_precision; the values stay on the device -- the reference reads them for its progress bar only)
        self.ap_metrics = ap_metrics
        self.ap_ret = {}
        from batch3dmot_amd.dist import FlatGradSync
        from batch3dmot_amd.train_step import make_optimizer
        self.kind, self.dev, self.encoders = kind, dev, encoders
        self.graphs, self.modalities = graphs, modalities       # graphs per step and GPU (train.py:86-90: batch_size 2)
        self.cap_ret = {}                                        # pool batch -> (loss, out, aux) tensors of its captured step
        torch.manual_seed(5621)                      # gnn.manual_seed, pose_config.yaml:96
        if kind == "pose":
            from batch3dmot_amd.pose_gnn import PoseGNN
            self.work = PoseWork
            self.model = PoseGNN().to(dev)
            self.logits = True
        else:
            from batch3dmot_amd.clr_att_gnn import GNN
            self.work = ClrWork
            self.model = GNN(enc_mod.ResNetAE(), enc_mod.PointNetClassifier(k=7), enc_mod.RadarNetClassifier(k=7)).to(dev)
            self.model.mask_stream = torch.cuda.Stream(dev)     # inputs are resident: the masks need not queue behind the previous step
            # (a high-priority stream for the masks was measured: 4.75 -> 7.9 ms per step -- stream priorities slow the whole replay here)
            self.logits = False
        self.model.run_dead_knn = not args.no_dead_knn
        if getattr(args, "no_dead_last_messages", False) and kind == "clr":
            self.model.run_dead_last_messages = False           # (the last layer's message stacks + node update: their x is never read)
        self.model.single_stream = True
        # The discarded k-NN + GAT block on the library's side stream (forked where x[l] exists, joined at the end of forward):
        # with the encoders inside forward this only adds jitter (A/B round 5: 4.48 vs 4.50 ms; PoseGNN 0.946 vs 0.963), but in the
        # encode-ahead step -- where the forward shares the GPU with ResNetAE only -- taking its 190 us off the launch stream's chain
        # is worth 2.9 % (4.18 -> 4.06 ms, two rounds on one box); postponing the join to the end of backward loses most of it again
        # (4.13).  B3D_KNN_SIDE=0 / 1 / 2 overrides.
        knn_side = os.environ.get("B3D_KNN_SIDE", "1" if (kind == "clr" and encoders == "frozen" and getattr(args, "encode_ahead", False)) else "0")
        if knn_side != "0":
            self.model.single_stream = False
            self.model.defer_knn_join = knn_side == "2"
        self.model.train()
        self.opt = make_optimizer(self.model, capturable=True)   # Adam(lr 1e-4, wd 1e-4, betas .9/.999): train.py:106-109
        self.force_collective = bool(getattr(args, "force_collective", False))
        self.sync = (FlatGradSync(self.model.parameters(), flat=self.opt if hasattr(self.opt, "flat_grad") else None)
                     if (world > 1 or self.force_collective) else None)
        self.pool_cpu = [synth.make_batch(graphs, 1500, 15000, first_graph_idx=rank * 1000 + graphs * i, modalities=(kind == "clr"))
                         for i in range(4)]
        if kind == "clr" and modalities == "cl":                 # BASELINE.json configs[2]: no radar return anywhere
            for b in self.pool_cpu:
                b.radar_feats = torch.zeros_like(b.radar_feats)
        self.pool = [b.to(dev) for b in self.pool_cpu]
        self.n_nodes = self.pool[0].pose_feats.size(0)
        self.edges = [b.edge_index.size(1) for b in self.pool]
        self.enc = None
        self.rows_static = None
        self.ahead = None
        self.enc_static = None
        self.graph_ws = None
        self.row_mismatch = None                                 # device counter: a replayed batch whose modality counts differ from the captured ones
        self._pending = None
        if kind == "clr":
            rows = [self.model.modality_rows(b) for b in self.pool]
            self.nl = sum(int(r[0].numel()) for r in rows) / len(rows)
            self.nr = sum(int(r[1].numel()) for r in rows) / len(rows)
            if encoders == "precomputed":
                self.enc = [self.model.encode_modalities(b, rows=r) for b, r in zip(self.pool, rows)]
            else:
                self.rows_static = [(r[0].clone(), r[1].clone()) for r in rows]
                if getattr(args, "encode_ahead", False):
                    # train_step.EncodeAhead: the frozen encoders of pool batch k + 1 run on a side stream under the step of
                    # batch k and leave their outputs in static buffers (a captured step needs fixed addresses); batch 0 is
                    # encoded here, once, in front of everything.  Same encoder passes, same order, same bits as encoding
                    # inside forward (tests/test_timed_config.py).
                    from batch3dmot_amd.train_step import EncodeAhead
                    self.ahead = EncodeAhead(self.model)
                    f32, i32 = dict(dtype=torch.float32, device=dev), dict(dtype=torch.int32, device=dev)
                    self.enc_static = [(torch.zeros(self.n_nodes, 96, **f32), torch.zeros(r[0].numel(), 256, **f32), torch.zeros(r[0].numel(), **i32),
                                        torch.zeros(r[1].numel(), 256, **f32), torch.zeros(r[1].numel(), **i32)) for r in rows]
                    # ... and the CSR / CSC structure of batch k + 1 (round 6: EncodeAhead.launch_graph, one build per step as before,
                    # a step earlier), into per-batch static buffers
                    # ... and its modality row ids (round 6: masks + compaction inside the captured step, on the same side stream, the
                    # counts -- shapes of the captured graph -- checked on the DEVICE: no eager prologue, no host read-back per step)
                    if os.environ.get("B3D_ROWS_IN_GRAPH", "1") != "0":               # A/B switch
                        self.row_mismatch = torch.zeros(1, dtype=torch.int32, device=dev)
                    from batch3dmot_amd import _lib
                    if os.environ.get("B3D_GRAPH_AHEAD", "0") != "0":                 # A/B switch, default OFF: measured neutral (3.974 / 3.981 vs 3.963 / 3.958 ms)
                      self.graph_ws = [torch.empty(_lib.Graph.workspace_bytes(self.n_nodes, b.edge_index.size(1)), dtype=torch.uint8, device=dev)
                                     for b in self.pool]
                      for kb, b in enumerate(self.pool):         # every pool batch holds a structure in ITS static buffer from here on
                        self.ahead.launch_graph(b, ws=self.graph_ws[kb])
                    self.ahead.launch(self.pool[0], rows=self.rows_static[0], static=self.enc_static[0])
                    self.ahead.take(self.pool[0])

    def _run(self, i, kwargs, after_forward=None):
        from batch3dmot_amd.train_step import train_step
        b = self.pool[i % len(self.pool)]
        keep_graph = None
        if hasattr(b, "_b3d_graph"):
            if self.graph_ws is not None and kwargs is not None and "encoded" in kwargs and self.enc is None:
                pass                             # encode-ahead step: the structure was built under the previous step (launch_graph)
            else:
                keep_graph = b._b3d_graph if self.graph_ws is not None else None
                del b._b3d_graph                 # the CSR/CSC build is part of every step
        ret = train_step(self.model, b, self.opt, batch_size=self.graphs, loss_kind="cb", logits=self.logits, grad_sync=self.sync,
                         forward_kwargs=kwargs, after_forward=after_forward)
        if keep_graph is not None:
            b._b3d_graph = keep_graph            # (a serial pass in between: the encode-ahead steps keep finding the static structure)
        if self.ap_metrics:
            from batch3dmot_amd import metrics
            scores = torch.sigmoid(ret[1]) if self.logits else ret[1]
            self.ap_ret[i % len(self.pool)] = metrics._run(scores, b.y, b.edge_classes, 7)
        return ret

    def _run_fb(self, i, kwargs, after_forward=None):
        from batch3dmot_amd.train_step import forward_backward
        b = self.pool[i % len(self.pool)]
        if hasattr(b, "_b3d_graph") and not (self.graph_ws is not None and kwargs is not None and "encoded" in kwargs and self.enc is None):
            del b._b3d_graph
        return forward_backward(self.model, b, self.opt, batch_size=self.graphs, loss_kind="cb", logits=self.logits, forward_kwargs=kwargs,
                                after_forward=after_forward)

    def captured_fb(self, i):
        """The part of `captured` in front of the gradient exchange (N > 1: graph A; the all-reduce runs eagerly between
        it and the optimizer graph)."""
        k = i % len(self.pool)
        if self.enc is not None:
            ret = self._run_fb(i, {"encoded": self.enc[k]})
        elif self.ahead is not None:
            ret = self._ahead_step(i, self._run_fb, self.rows_static[(k + 1) % len(self.pool)])
        elif self.rows_static is not None:
            ret = self._run_fb(i, {"rows": self.rows_static[k]})
        else:
            ret = self._run_fb(i, None)
        self.cap_ret[k] = ret
        return ret

    def opt_step(self):
        self.opt.step()

    def _ahead_step(self, i, run, rows_next):
        """Step i with the encoders of the NEXT pool batch underneath it: fork (side stream) -> encoders(k + 1) into their static
        buffers | step(k) on the outputs the previous step left for batch k -> join."""
        k, kn = i % len(self.pool), (i + 1) % len(self.pool)
        if self.graph_ws is not None:
            self.ahead.launch_graph(self.pool[kn], ws=self.graph_ws[kn])     # one CSR / CSC build per step: the next batch's
        if self.row_mismatch is not None and rows_next is not None:
            self.ahead.launch_rows(self.pool[kn], rows_next, self.row_mismatch)   # (captured steps: rows_next IS rows_static[kn])
        launch = lambda parts="all": self.ahead.launch(self.pool[kn], rows=rows_next, static=self.enc_static[kn], parts=parts)       # noqa: E731
        # Where the next batch's encoders enter the current step (same work, same bits; A/B on one box, three rounds each, round 5):
        # everything in front of the forward 4.21 / 4.24 / 4.22 ms, everything behind the forward 4.17 / 4.23 / 4.23, ResNetAE in
        # front of the forward and the two point encoders behind it ("split") 4.16 / 4.21 / 4.17 -- the backward sweep has the
        # thinner kernels (node phase: 188 workgroups on 256 CUs) for the fat point stacks to run beside.  B3D_AHEAD_AT overrides.
        where = os.environ.get("B3D_AHEAD_AT", "split")
        if where == "backward":
            ret = run(i, {"encoded": self.enc_static[k]}, launch)
        elif where == "split":
            launch("img")
            ret = run(i, {"encoded": self.enc_static[k]}, lambda: launch("points"))
        elif where == "split5":          # experiment: ResNetAE + RadarNet under the forward, PointNet under the backward sweep
            launch("img"); launch("radar")
            ret = run(i, {"encoded": self.enc_static[k]}, lambda: launch("lidar"))
        elif where == "split3":          # experiment: ResNetAE + PointNet under the forward, RadarNet under the backward sweep
            launch("img"); launch("lidar")
            ret = run(i, {"encoded": self.enc_static[k]}, lambda: launch("radar"))
        elif where == "split4":          # experiment: PointNet under the forward, ResNetAE + RadarNet under the backward sweep
            launch("lidar")
            ret = run(i, {"encoded": self.enc_static[k]}, lambda: (launch("img"), launch("radar")))
        else:
            launch()
            ret = run(i, {"encoded": self.enc_static[k]})
        self.ahead.take(self.pool[kn])
        return ret

    def step(self, i):
        if self.enc is not None:
            return self._run(i, {"encoded": self.enc[i % len(self.pool)]})
        if self.ahead is not None:
            return self._ahead_step(i, self._run, None)       # rows of the next batch: masks + compaction inside launch()
        return self._run(i, None)

    def step_serial(self, i):
        """The same step with the encoders inside forward, whatever the timed mode: the pass that times kernel families with event
        pairs uses it, so that a launch's duration is the kernel's own (nothing else shares the GPU).  The encode-ahead buffers are
        left alone; run only after the timed region and its state checks."""
        if self.enc is not None:
            return self._run(i, {"encoded": self.enc[i % len(self.pool)]})
        return self._run(i, None)

    def pre(self, i):
        """The eager prologue of step i: masks + compaction of ONE batch, every step.  Round 5: begun one step ahead -- the
        launches of the batch step i + 1 needs are enqueued here, in front of replay i, and the counts of the batch THIS step needs
        (begun in front of replay i - 1) are read and handed to the replay: the host never waits behind a running step
        (GNN.modality_rows_begin / _end)."""
        if self.rows_static is None or self.row_mismatch is not None:
            return                                                # (round 6, encode-ahead: masks + compaction are inside the captured step)
        shift = 1 if self.ahead is not None else 0                # the batch whose encoders run in this step
        k, kn = (i + shift) % len(self.pool), (i + 1 + shift) % len(self.pool)
        if self._pending is None or self._pending[0] != k:
            self._pending = (k, self.model.modality_rows_begin(self.pool[k]))
        nxt = (kn, self.model.modality_rows_begin(self.pool[kn]))
        li, ri = self.model.modality_rows_end(self._pending[1])
        self._pending = nxt
        sl, sr = self.rows_static[k]
        if li.numel() != sl.numel() or ri.numel() != sr.numel():
            raise RuntimeError("modality row counts changed under a captured step")
        sl.copy_(li)                                              # the replay consumes THIS step's rows
        sr.copy_(ri)

    def captured(self, i):
        k = i % len(self.pool)
        if self.enc is not None:
            ret = self._run(i, {"encoded": self.enc[k]})
        elif self.ahead is not None:
            ret = self._ahead_step(i, self._run, self.rows_static[(k + 1) % len(self.pool)])
        elif self.rows_static is not None:
            ret = self._run(i, {"rows": self.rows_static[k]})
        else:
            ret = self._run(i, None)
        self.cap_ret[k] = ret              # static tensors of the graph: they hold the last replay's loss / scores
        return ret

    def describe(self, world):
        if self.kind == "pose":
            return ("pose_config.yaml poses-only PoseGNN depth 6, training step (CSR/CSC build + fwd + cb-BCE + bwd + Adam"
                    + (" + flat RCCL grad all-reduce" if world > 1 else "") + ")")
        enc = ("frozen ResNetAE / PointNet / RadarNet encoders in train mode inside the step" if self.encoders == "frozen"
               else "encoder outputs precomputed")
        if self.ahead is not None:
            enc += (" (one encoder pass per step, enqueued for the NEXT pool batch on a side stream under this batch's step -- ResNetAE under "
                    "its forward, PointNet / RadarNet under its backward sweep: train_step.EncodeAhead"
                    + ("; the next batch's CSR/CSC build on the same side stream" if self.graph_ws is not None else "") + ")")
        name = "camera+LiDAR+radar" if self.modalities == "clr" else "camera+LiDAR (radar rows all zero)"
        return (name + " GNN (clr_att_gnn) depth 6, training step (modality masks + " + enc
                + " + CSR/CSC build + fwd + cb-BCE + bwd + Adam" + (" + flat RCCL grad all-reduce of 5.24 MB" if world > 1 else "")
                + (" + average precision of the scores, overall and per class, as train.py:143-150 (b3d_average_precision)" if self.ap_metrics else "") + ")")


class StubWorkload:
    """--stub-cpu (testing aid, tests/test_bench_control_flow.py): a tiny torch model on the CPU in place of the HIP
    workload, so that the N > 1 control flow of this file -- process group, sharded pool, gradient all-reduce inside
    the step, barrier / ramp / timed loop, MAX / SUM reductions, one JSON line from rank 0 -- runs under gloo without
    a GPU.  Its numbers mean nothing."""
    kind, encoders, rows_static, logits = "stub", "na", None, True

    def __init__(self, dev, rank, world):
        from batch3dmot_amd.dist import FlatGradSync
        self.dev = dev
        torch.manual_seed(5621)
        self.model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 1))
        self.model.depth, self.model.run_dead_knn = 1, False
        self.opt = torch.optim.SGD(self.model.parameters(), lr=1e-3)
        self.sync = FlatGradSync(self.model.parameters()) if world > 1 else None
        g = torch.Generator().manual_seed(rank)
        self.pool = [torch.randn(100 + 10 * i, 8, generator=g) for i in range(4)]
        self.edges = [b.size(0) for b in self.pool]
        self.n_nodes = 10

    def step(self, i):
        b = self.pool[i % len(self.pool)]
        self.opt.zero_grad()
        self.model(b).sum().backward()
        if self.sync is not None:
            self.sync.sync()
        self.opt.step()

    pre = captured = step_serial = step

    def describe(self, world):
        return "stub (CPU control-flow test)"


def dev_sync(dev):
    if dev.type == "cuda":
        torch.cuda.synchronize()


def trace(msg):
    if os.environ.get("B3D_BENCH_TRACE"):
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def capture(wl, split):
    """One hipGraph per pool batch, captured on a side stream as `torch.cuda.graph` wants it.  split (N > 1): graph A =
    forward + backward per batch, plus one graph B = the optimizer step.  Returns (graphs, opt_graph); raises if the
    stack cannot capture."""
    graphs, opt_graph = [], None
    torch.cuda.synchronize()
    cap_stream = torch.cuda.Stream()
    cap_stream.wait_stream(torch.cuda.current_stream())
    for i in range(len(wl.pool)):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=cap_stream, capture_error_mode="thread_local"):
            if split:
                wl.captured_fb(i)
            else:
                wl.captured(i)
        graphs.append(g)
        trace(f"captured batch {i}")
    if split:
        opt_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(opt_graph, stream=cap_stream, capture_error_mode="thread_local"):
            wl.opt_step()
        trace("captured the optimizer step")
    torch.cuda.current_stream().wait_stream(cap_stream)
    torch.cuda.synchronize()
    return graphs, opt_graph


_SKIP_PRE = os.environ.get("B3D_SKIP_PRE", "0") == "1"      # diagnostic only: the replay without its eager prologue (tools/)


def run_step(wl, graphs, opt_graph, split, i, stamps=None):
    """Step i of the timed region: the eager part in front of the replay (`pre`), the replay(s), and at N > 1 the eager
    flat all-reduce between them; or the eager step when nothing was captured."""
    if graphs is not None:
        t0 = time.perf_counter()
        if not _SKIP_PRE:
            wl.pre(i)
        t1 = time.perf_counter()
        graphs[i % len(wl.pool)].replay()
        if stamps is not None:
            stamps.append((t1 - t0, time.perf_counter() - t1))
        if split:
            # the flat gradient buffer was written by the replay
            wl.sync.sync(force=True, force_collective=getattr(wl, "force_collective", False))
            opt_graph.replay()
    else:
        wl.step(i)


def snapshot(wl):
    """Everything a step changes: parameters and buffers (BatchNorm running statistics, counters), Adam state, the
    device RNG state (Dropout of the frozen encoders' heads)."""
    o = wl.opt
    sd = {k: v.detach().clone() for k, v in wl.model.state_dict().items()}
    st = {"exp_avg": o.exp_avg.clone(), "exp_avg_sq": o.exp_avg_sq.clone(),
          "step_dev": o.step_dev.clone() if getattr(o, "step_dev", None) is not None else None,
          "step_count": o.step_count, "fresh": o.fresh}
    enc = [tuple(t.clone() for t in slot) for slot in wl.enc_static] if getattr(wl, "enc_static", None) else None
    return sd, st, torch.cuda.get_rng_state(wl.dev), enc


def restore(wl, snap):
    sd, st, rng, enc = snap
    o = wl.opt
    torch.cuda.synchronize()
    with torch.no_grad():
        for k, v in wl.model.state_dict().items():
            v.copy_(sd[k])
        o.exp_avg.copy_(st["exp_avg"])
        o.exp_avg_sq.copy_(st["exp_avg_sq"])
        if st["step_dev"] is not None:
            o.step_dev.copy_(st["step_dev"])
    o.step_count, o.fresh = st["step_count"], st["fresh"]
    if enc is not None:                                          # encoder outputs waiting for their step (EncodeAhead)
        with torch.no_grad():
            for slot, saved in zip(wl.enc_static, enc):
                for dst, src in zip(slot, saved):
                    dst.copy_(src)
    torch.cuda.set_rng_state(rng, wl.dev)
    torch.cuda.synchronize()


def state_digest(wl):
    """Flat copies of what `snapshot` covers, for bitwise comparisons (tests/test_timed_config.py)."""
    sd = wl.model.state_dict()
    enc = {f"encode_ahead.{i}.{j}": t.clone() for i, slot in enumerate(getattr(wl, "enc_static", None) or []) for j, t in enumerate(slot)}
    return {**enc, **{"model." + k: v.detach().clone() for k, v in sd.items()},
            "adam.exp_avg": wl.opt.exp_avg.clone(), "adam.exp_avg_sq": wl.opt.exp_avg_sq.clone(),
            **({"adam.step": wl.opt.step_dev.clone()} if getattr(wl.opt, "step_dev", None) is not None else {})}


def loss_check(wl, graphs, i):
    """What the timed region computes, checked against the eager step: from the same state, replay step i (with its eager
    prologue) and read the loss out of the graph's static tensor; restore; run the eager step on the same batch.  The two
    losses must agree (they are the same kernels on the same inputs: bitwise on this stack).  State is left as after the
    eager step."""
    k = i % len(wl.pool)
    snap = snapshot(wl)
    wl.pre(i)
    graphs[k].replay()
    torch.cuda.synchronize()
    loss_replay = float(wl.cap_ret[k][0])
    restore(wl, snap)
    loss_eager = float(wl.step(i)[0])
    torch.cuda.synchronize()
    return {"batch": k, "replayed_loss": loss_replay, "eager_loss": loss_eager, "equal": loss_replay == loss_eager,
            "abs_diff": abs(loss_replay - loss_eager)}


def collective_check(wl, graphs, opt_graph, i):
    """--force-collective (one rank): graph A | RCCL all-reduce(AVG) of the flat gradient buffer on the launch stream | graph B
    must leave exactly the state of the eager step without a collective (a mean over one rank changes no bit)."""
    snap = snapshot(wl)
    run_step(wl, graphs, opt_graph, True, i)
    torch.cuda.synchronize()
    a = state_digest(wl)
    restore(wl, snap)
    wl.step(i)                                                   # eager; FlatGradSync.sync() skips the collective at one rank
    torch.cuda.synchronize()
    b = state_digest(wl)
    bad = [k for k in a if not torch.equal(a[k], b[k])]
    return {"tensors": len(a), "equal": not bad, "differing": bad[:5]}


def measure(wl: Workload, args, world, dist, steps, warmup, ramp_ms, use_graph):
    """W instrumented warm-up steps, optional hipGraph capture, untimed clock ramp, K timed steps (barrier +
    synchronize on both sides), eager instrumented pass.  Returns a dict of raw measurements."""
    dev = wl.dev
    pool_n = len(wl.pool)
    stub = wl.kind == "stub"
    if stub:
        for i in range(warmup):
            wl.step(i)
        ramp_steps = 8 if (ramp_ms > 0 and world > 1) else 0
        for i in range(ramp_steps):
            wl.step(i)
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            wl.step(warmup + i)
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        return {"dt": dt, "edges": sum(wl.edges[(warmup + i) % pool_n] for i in range(steps)), "t_enqueue": dt, "fam_all": None,
                "fam": {}, "dom": None, "graphs": False, "graph_note": "stub", "ramp_steps": ramp_steps, "pair_us": 0.0,
                "step_ms": None, "loss_check": None}
    from batch3dmot_amd import _lib
    fam_names = list(_lib.KERNEL_FAMILIES)
    # Warm-up doubles as the instrumented pass: every kernel family is timed with HIP event pairs (diagnostic
    # table) and the family with the largest device time is picked; the TIMED region carries no event pairs.
    trace(f"{wl.kind}/{wl.encoders}: warm-up")
    # The instrumented warm-up steps run SERIALLY (encoders inside forward): with encode-ahead two streams share the GPU and a
    # launch's duration is no longer the kernel's own -- the family with the most device time must be picked from undisturbed
    # durations (the first split-placement run picked the node backward, inflated by the point stacks running beside it).  Two
    # un-instrumented steps in the timed form follow, so that everything the capture replays has run eagerly once.
    _lib.prof_enable(True)
    for i in range(warmup):
        wl.step_serial(i)
    torch.cuda.synchronize()
    fam_all = _lib.prof_read() if warmup > 0 else None
    _lib.prof_enable(False)
    if getattr(wl, "ahead", None) is not None:
        for i in range(2):
            wl.step(i)
        torch.cuda.synchronize()
    trace("warm-up done")
    graphs, graph_note, opt_graph = None, None, None
    # N > 1: graph A (forward + backward) | eager all-reduce | graph B (optimizer); --force-collective: the same at N = 1
    split = world > 1 or bool(getattr(wl, "force_collective", False))
    if use_graph:
        try:
            graphs, opt_graph = capture(wl, split)
        except Exception as exc:                                   # capture unsupported here: eager timed region
            graphs, opt_graph = None, None
            graph_note = f"hipGraph capture failed ({type(exc).__name__}: {str(exc)[:200]}); eager timed region"
            torch.cuda.synchronize()
        if world > 1:                                              # every rank takes the same path (each step holds a collective)
            ok = torch.tensor([1.0 if graphs is not None else 0.0], device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if float(ok.item()) == 0.0 and graphs is not None:
                graphs, opt_graph = None, None
                graph_note = "hipGraph capture failed on another rank; eager timed region"

    stamps = []

    def timed_step(i):
        run_step(wl, graphs, opt_graph, split, i, stamps)

    # Clock ramp (untimed): a fresh box idles at ~550 MHz; keep the GPU busy until ramp_ms have passed, the same
    # number of steps on every rank (each holds a collective).
    ramp_steps = 0
    if ramp_ms > 0 and world > 1:
        ramp_steps = max(8, int(ramp_ms / 10.0) // 8 * 8)
        for i in range(ramp_steps):
            timed_step(i)
        torch.cuda.synchronize()
    elif ramp_ms > 0:
        t_r = time.perf_counter()
        while (time.perf_counter() - t_r) * 1e3 < ramp_ms and ramp_steps < 2000:
            for i in range(4):
                timed_step(i)
            torch.cuda.synchronize()
            ramp_steps += 4
    trace(f"ramp done ({ramp_steps} steps)")
    # one event per step boundary (K + 1 records on the launch stream, no synchronisation): per-step durations for the
    # median beside the contract's mean
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks[0].record()
    del stamps[:]
    for i in range(steps):
        timed_step(warmup + i)
        marks[i + 1].record()
    # Host time to enqueue the K steps (diagnostic): the GPU's rate, not the host's cost -- the eager prologue reads the modality
    # counts back (`.item()`: it returns when the mask kernels have run, and those queue behind the previous replay), and a bare
    # replay loop blocks in hipGraphLaunch once 40-80 ms of work are queued (tools/host_graph_node_cost.py).  What a step COSTS the
    # host is the launch call itself: the median over the timed steps is reported beside the loop mean.
    t_enqueue = time.perf_counter() - t0
    t_first = sorted(s[1] for s in stamps)[len(stamps) // 2] if stamps else 0.0
    t_pre = sorted(s[0] for s in stamps)[len(stamps) // 2] if stamps else 0.0
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if getattr(wl, "row_mismatch", None) is not None and int(wl.row_mismatch.item()) != 0:
        raise RuntimeError("a replayed step saw modality row counts that differ from the captured graph's (device-side check of "
                           "GNN.modality_rows_into): the timed region is invalid")
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
    trace(f"timed region done: {1e3 * dt / steps:.3f} ms/step")
    # the replayed step against the eager step, from the same state (N = 1: at N > 1 a step holds a collective)
    lc = None
    if graphs is not None and not split:
        try:
            lc = loss_check(wl, graphs, warmup + steps)
        except Exception as exc:
            lc = {"error": f"{type(exc).__name__}: {str(exc)[:200]}"}
    elif graphs is not None and world == 1:
        try:
            lc = {"collective_vs_eager_state": collective_check(wl, graphs, opt_graph, warmup + steps)}
        except Exception as exc:
            lc = {"error": f"{type(exc).__name__}: {str(exc)[:200]}"}
    # the same K steps again, eagerly, with event pairs: on the dominant path family only (undisturbed), then on all
    dom = None
    if fam_all:
        cand = [k for k in fam_names if k in PATH_FAMILIES and fam_all[k][1] > 0]
        dom = max(cand, key=lambda k: fam_all[k][0]) if cand else None
    fam = {}
    for sel in ([dom] if dom else [None]) + (["point_feat"] if fam_all and fam_all.get("point_feat", (0, 0))[1] > 0 else []):
        _lib.prof_enable(True, families=[sel] if sel else None)
        for i in range(steps):
            wl.step_serial(warmup + i)
        torch.cuda.synchronize()
        got = _lib.prof_read()
        _lib.prof_enable(False)
        if sel is None:
            fam = got
        else:
            fam[sel] = got[sel]
    # what an event pair adds to the kernel it brackets: half of a pair around an empty kernel (the other half is that
    # kernel's own dispatch-to-completion; calibrated against rocprofv3 durations of the same launches: 29.5 us by events
    # vs 24.1 us by rocprofv3 with an 11.3 us empty pair, profiles/r02_a_*)
    pair_us = 0.5 * _lib.prof_pair_overhead_us(torch.cuda.current_stream(dev).cuda_stream)
    my_edges = sum(wl.edges[(warmup + i) % pool_n] for i in range(steps))
    return {"dt": dt, "edges": my_edges, "t_enqueue": t_enqueue, "t_first": t_first, "t_pre": t_pre, "fam_all": fam_all, "fam": fam, "dom": dom,
            "graphs": graphs is not None, "graph_note": graph_note, "ramp_steps": ramp_steps, "pair_us": pair_us,
            "step_ms": step_ms, "loss_check": lc}


def mfma_peak(bf_frac):
    """fp32-equivalent MFMA peak of a kernel that runs the fraction bf_frac of its MACs as bf16x6 and the rest on the
    exact fp32 MFMA: time adds, so the peaks combine harmonically."""
    return 1.0 / (bf_frac / PEAK_BF16X6_TFLOPS + (1.0 - bf_frac) / PEAK_FP32_MFMA_TFLOPS)


def family_table(famd, steps, alg, exe, byts, bf, pair_us=0.0):
    out = {}
    for name, (ms, n) in famd.items():
        if n == 0 or steps == 0:
            continue
        avg_ev = 1e3 * ms / n
        avg_us = max(avg_ev - pair_us, 1e-3)
        us_step = avg_us * n / steps
        k = {"launches_per_step": round(n / steps, 2), "avg_us": round(avg_us, 2), "us_per_step": round(us_step, 1)}
        if name in alg:
            k["bound"] = BOUND[name]
            k["executed_tflops"] = round(exe[name] / (us_step * 1e-6) / 1e12, 2)        # fp32-equivalent
            k["algorithmic_tflops"] = round(alg[name] / (us_step * 1e-6) / 1e12, 2)
            k["algorithmic_gbs"] = round(byts[name] / (us_step * 1e-6) / 1e9, 1)
            k["bf16x6_fraction_of_macs"] = round(bf.get(name, 0.0), 3)
            k["mfma_peak_tflops"] = round(mfma_peak(bf.get(name, 0.0)), 1)
        out[name] = k
    return out


def lib_sha16():
    import hashlib
    from batch3dmot_amd import _lib
    try:
        with open(_lib.LIB_PATH, "rb") as fh:
            return hashlib.sha256(fh.read()).hexdigest()[:16]
    except OSError:
        return None


def load_traffic(workload_key):
    """HBM bytes per launch from the rocprofv3 --pmc passes of this command (FETCH_SIZE / WRITE_SIZE in separate
    passes, read bytes doubled as MI355X_MICROARCH.md prescribes for gfx950), summarised by tools/pmc_traffic.py into
    profiles/traffic_pmc.json, keyed by workload.  Counters cannot be collected from inside this process: the figure
    comes from the builder's last PMC session, and `_source` says which library build and day that was."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic_pmc.json")) as fh:
            d = json.load(fh)
        return d.get(workload_key), d.get("_source")
    except (OSError, ValueError):
        return None, None


def roofline_of(dom, kd, alg, exe, byts, pair_us, traffic, traffic_src):
    lps = kd["launches_per_step"]
    if BOUND[dom] == "mfma":
        achieved, peak, unit = kd["executed_tflops"], kd["mfma_peak_tflops"], "TFLOP/s"
    else:
        achieved, peak, unit = kd["algorithmic_gbs"], PEAK_HBM_GBS, "GB/s"
    src = None
    if traffic_src:
        src = dict(traffic_src)
        src["matches_this_library"] = (traffic_src.get("lib_sha16") == lib_sha16())
    tr = (traffic or {}).get(dom)
    us = kd["avg_us"]
    both = {"mfma": {"achieved_tflops": kd["executed_tflops"], "peak_tflops": kd["mfma_peak_tflops"],
                     "frac": round(kd["executed_tflops"] / kd["mfma_peak_tflops"], 4),
                     "numerator": "executed fp32-equivalent FLOPs of the family per step",
                     "denominator": "MFMA peak of the instructions it runs (157.3 exact-fp32, 2500 / 6 = 416.7 bf16x6, combined harmonically)"},
            "hbm": {"achieved_gbs": kd["algorithmic_gbs"], "peak_gbs": PEAK_HBM_GBS, "frac": round(kd["algorithmic_gbs"] / PEAK_HBM_GBS, 4),
                    "numerator": "bytes of the operands this launch is handed, each counted once (DESIGN.md section 4 kernel table)",
                    "denominator": "8 TB/s HBM3E peak",
                    "measured_traffic_gbs": round(tr / (us * 1e-6) / 1e9, 1) if tr else None,
                    "measured_traffic_frac": round(tr / (us * 1e-6) / 1e9 / PEAK_HBM_GBS, 4) if tr else None}}
    return {"kernel": dom, "bound": BOUND[dom], "achieved": achieved, "peak": peak, "unit": unit,
            "frac": round(achieved / peak, 4), "traffic": tr, "traffic_source": src, "both_roofs": both,
            "avg_launch_us": kd["avg_us"], "event_pair_overhead_us_subtracted": round(pair_us, 2),
            "launches_per_step": lps,
            "executed_flops_per_launch": exe[dom] / lps, "algorithmic_flops_per_launch": alg[dom] / lps,
            "algorithmic_bytes_per_launch": byts[dom] / lps,
            "bf16x6_fraction_of_macs": kd["bf16x6_fraction_of_macs"],
            "algorithmic_tflops": kd["algorithmic_tflops"],
            "executed_frac_of_fp32_mfma_peak": round(kd["executed_tflops"] / PEAK_FP32_MFMA_TFLOPS, 4)}


ROOFLINE_NOTE = ("roofline.kernel = the SURVEY.md 8a kernel family (message passing, attention encoder, their weight gradient) "
                 "with the most device time per step; the frozen encoders' point stacks (8f #1, next to the path) are "
                 "`roofline_encoders`.  mfma-bound kernels: achieved = EXECUTED fp32-equivalent FLOPs of the family per step / its "
                 "device time per step (node columns of a Linear over a concatenation that are evaluated per node are not "
                 "edge-kernel work); peak = the MFMA peak of the instructions it runs: 157.3 TFLOP/s for exact-fp32 MFMA layers, "
                 "2500 / 6 = 416.7 fp32-equivalent TFLOP/s for bf16x6 layers (six bf16 MFMA MACs per fp32 product), combined "
                 "harmonically by the kernel's share of each.  algorithmic_* count the reference's per-edge FLOPs (SURVEY.md 8d) "
                 "over the same time.  Durations: HIP event pairs on the launch stream around each launch of the family in an "
                 "eager pass of the same K steps, minus half the cost of a pair around an empty kernel (calibrated against "
                 "rocprofv3 durations, profiles/).  traffic: HBM bytes per launch from rocprofv3 PMC passes (see traffic_source).  "
                 "`bound` names the NEARER roof; `both_roofs` gives the fraction of the matrix pipes and of HBM for the same launch, "
                 "each with its numerator and denominator.  The cooperative weight gradient (wgrad_edge) saturates neither: its "
                 "time is the staging chain of a 32-row step (split to bf16 pieces, LDS writes, barrier, transposed fragment reads, "
                 "MFMAs, barrier -- DESIGN.md section 4), HBM being the nearer roof by measured traffic.")


def workload_roofline(wl, m, args, steps):
    """(roofline of the path family with the most device time, the family table, (alg, exe, byts, traffic, traffic_src)) of one
    measured workload."""
    from batch3dmot_amd import _lib
    e_avg = m["edges"] / steps
    hoist = _lib.features()
    kw = dict(n=wl.n_nodes, e=e_avg, depth=wl.model.depth)
    if wl.kind == "clr":
        kw.update(hoist_mp=bool(hoist.get("clr_hoist_mp")), hoist_att=bool(hoist.get("clr_hoist_att")), nl=wl.nl, nr=wl.nr)
    alg, exe, byts, bf = wl.work.families(**kw)
    kernels = family_table(m["fam"], steps, alg, exe, byts, bf, m["pair_us"])
    dom = m["dom"] or max((k for k in kernels if k in PATH_FAMILIES), key=lambda k: kernels[k]["us_per_step"])
    workload_key = f"{wl.kind}:{wl.encoders if wl.kind == 'clr' else 'na'}:knn{int(bool(wl.model.run_dead_knn))}"
    traffic, traffic_src = load_traffic(workload_key)
    roofline = roofline_of(dom, kernels[dom], alg, exe, byts, m["pair_us"], traffic, traffic_src)
    return roofline, kernels, (alg, exe, byts, bf, traffic, traffic_src)


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start `torch.distributed.run` with N ranks of this same command as a
    CHILD process (train_resnet_ae_ddp.py:288-290 self-spawns with mp.spawn) and exit with its code; rank 0 of the child
    prints the JSON line on the stdout this process was given.  Called before anything initialises the GPU in this
    process -- a process that has touched the GPU must never be replaced by another program -- and the parent makes no
    GPU call at all (torch.cuda.device_count() does not initialise the runtime on this image)."""
    import socket
    import subprocess
    if "--stub-cpu" not in sys.argv and "--all-ranks-on-device-0" not in sys.argv:
        have = torch.cuda.device_count()
        if have < n:
            raise SystemExit(f"--gpus {n}: only {have} GPU(s) visible")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL across processes needs it on this stack
    env.setdefault("OMP_NUM_THREADS", "8")
    rc = subprocess.call(cmd, env=env)
    if rc:
        raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--mode", choices=["train", "infer"], default="train",
                    help="train: the training step (default).  infer: BASELINE.json configs[4], forward only")
    ap.add_argument("--model", choices=["clr", "pose"], default="clr")
    ap.add_argument("--encoders", choices=["frozen", "precomputed"], default="frozen",
                    help="camera+LiDAR+radar step: frozen encoders in train mode inside the step (default) or their outputs given")
    ap.add_argument("--modalities", choices=["clr", "cl"], default="clr", help="cl: radar rows all zero (camera+LiDAR, configs[2])")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: 2 graphs per GPU and step.  strong: global batch of 16 graphs, 16 / N per GPU (SURVEY.md 8d(4))")
    ap.add_argument("--no-dead-knn", action="store_true",
                    help="skip the k-NN + GAT block whose result the reference discards (secondary figure)")
    ap.add_argument("--no-dead-last-messages", action="store_true",
                    help="do not execute the last layer's message stacks and node update (their result is never read: "
                         "clr_att_gnn.py:188); outputs and gradients are identical.  Reported as secondary.clr_without_dead_work")
    ap.add_argument("--no-graph", action="store_true", help="enqueue every step eagerly")
    ap.add_argument("--serial-encoders", action="store_true",
                    help="encode every batch inside its own forward (rounds 1-4's timed step).  Default since round 5: "
                         "train_step.EncodeAhead -- the frozen encoders of the NEXT pool batch run on a side stream under the current "
                         "batch's step (one encoder pass and one message-passing step per timed step either way; parameters, BatchNorm "
                         "statistics and Adam state bit-equal to the serial loop, tests/test_timed_config.py).  With two kernels sharing "
                         "the GPU a per-launch duration is no longer a property of the kernel, so the kernel families behind `roofline` "
                         "are timed in a serial pass of the same steps; the default run reports the serial step as a secondary")
    ap.add_argument("--encode-ahead", dest="encode_ahead_flag", action="store_true",
                    help="training: the default (kept for round-4 command lines).  --mode infer: the encoders of the NEXT window on a side "
                         "stream under this window's forward (off by default there)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary figures of the default N = 1 run")
    ap.add_argument("--ramp-ms", type=float, default=250.0, help="untimed clock ramp in front of the timed region (0 = none)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to exercise the N > 1 path on one GPU)")
    ap.add_argument("--all-ranks-on-device-0", action="store_true", help="testing aid for the N > 1 path on a 1-GPU box (with --backend gloo)")
    ap.add_argument("--stub-cpu", action="store_true", help="testing aid: run this file's control flow with a stub workload on the CPU (gloo)")
    ap.add_argument("--scene", action="store_true",
                    help="--mode infer: scene-level inference (predict_post.predict_scene: embedding cache, window mean, greedy flux, "
                         "tracks) on a synthetic 24-frame scene, cached against uncached")
    ap.add_argument("--force-collective", action="store_true",
                    help="N = 1 only: initialise the process group with ONE rank and run the N > 1 timed region (forward + backward "
                         "graph | flat all-reduce on the launch stream | optimizer graph) -- executes RCCL next to the captured graphs "
                         "on a 1-GPU box")
    args = ap.parse_args()
    # training: encode-ahead unless --serial-encoders; inference: only on request
    args.encode_ahead = args.encode_ahead_flag if args.mode == "infer" else not args.serial_encoders

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus)            # nothing of this process has touched the GPU yet
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    import torch.distributed as dist
    if args.stub_cpu:
        return main_stub(args, rank, world, dist)
    dev = torch.device("cuda", 0 if args.all_ranks_on_device_0 else local_rank)
    torch.cuda.set_device(dev)
    # Nothing of this process runs on the legacy NULL stream: eager work enqueued there between the replays of a
    # captured step (the modality-row compaction) ended in a GPU memory fault at the next replay on this stack
    # (tools/debug_clr_capture.py: cases nm_rows_nocopy vs nm_curstream), and a non-blocking stream is what a
    # training loop with a prefetching loader runs on anyway.
    torch.cuda.set_stream(torch.cuda.Stream(dev))
    if world > 1 or args.force_collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend=args.backend, rank=rank, world_size=world)
    if args.mode == "infer":
        return main_infer(args, dev, rank, world, dist)

    from batch3dmot_amd import _lib
    graphs_per_gpu = 2
    if args.scaling == "strong":
        if 16 % world:
            raise SystemExit("--scaling strong: the global batch of 16 graphs must divide over the ranks")
        graphs_per_gpu = 16 // world
    wl = Workload(args.model, dev, rank, world, args, encoders=args.encoders, graphs=graphs_per_gpu, modalities=args.modalities)
    m = measure(wl, args, world, dist, args.steps, args.warmup, args.ramp_ms, use_graph=not args.no_graph)

    dt, total_edges = reduce_over_ranks(m, dev, world, dist)

    secondary = None
    if (world == 1 and rank == 0 and not args.no_secondary and args.model == "clr" and args.encoders == "frozen"
            and args.modalities == "clr" and args.scaling == "weak"):
        secondary = {}
        for key, kind, enc, mod in (("clr_serial_encoders", "clr", "frozen", "clr"), ("clr_with_ap_metrics", "clr", "frozen", "clr"),
                                    ("clr_without_dead_work", "clr", "frozen", "clr"),
                                    ("clr_encoders_precomputed", "clr", "precomputed", "clr"),
                                    ("camera_lidar", "clr", "frozen", "cl"),
                                    ("camera_lidar_encoders_precomputed", "clr", "precomputed", "cl"), ("pose_gnn", "pose", "frozen", "clr")):
            try:
                a2 = argparse.Namespace(**{**vars(args), "encode_ahead": args.encode_ahead and key != "clr_serial_encoders", "model": kind})
                if key == "clr_without_dead_work":
                    # what the reference computes and never reads, left out: the frame-wise k-NN + GAT block (pose_gnn.py:80) and the
                    # last layer's message stacks + node update (clr_att_gnn.py:188).  Same outputs, same gradients; NOT the headline.
                    a2.no_dead_knn = True
                    a2.no_dead_last_messages = True
                w2 = Workload(kind, dev, rank, world, a2, encoders=enc, modalities=mod, ap_metrics=(key == "clr_with_ap_metrics"))
                k2 = max(10, args.steps // 2)
                m2 = measure(w2, args, world, dist, k2, max(3, args.warmup // 2), 60.0, use_graph=not args.no_graph)
                secondary[key] = {"workload": w2.describe(world), "value": round(m2["edges"] / m2["dt"], 1), "unit": "edges/s",
                                  "ms_per_step": round(1e3 * m2["dt"] / k2, 4),
                                  "ms_per_step_median": round(m2["step_ms"][len(m2["step_ms"]) // 2], 4),
                                  "timed_region": "hipGraph replay" if m2["graphs"] else "eager",
                                  "replay_vs_eager_loss": m2["loss_check"]}
                if key == "pose_gnn":                              # BASELINE.md section 4 quotes this model's ceilings: its own roofline
                    secondary[key]["roofline"], secondary[key]["kernels"] = workload_roofline(w2, m2, a2, k2)[:2]
                del w2
            except Exception as exc:                                # a secondary figure must never cost the headline
                secondary[key] = {"error": f"{type(exc).__name__}: {str(exc)[:200]}"}
            torch.cuda.empty_cache()
        # BASELINE.json configs[4] in the default line: 64 windows of 2,000 / ~20,000 per step, forward only, with its own roofline
        # and the CPU oracle's forward beside it
        try:
            a3 = argparse.Namespace(**{**vars(args), "model": "clr", "encode_ahead": False, "mode": "infer"})
            secondary["infer_clr"] = infer_measure(a3, dev, rank, world, dist, max(3, args.steps // 8), 4, cpu=not args.no_cpu_baseline)
        except Exception as exc:
            secondary["infer_clr"] = {"error": f"{type(exc).__name__}: {str(exc)[:200]}"}
        torch.cuda.empty_cache()
        try:
            secondary["infer_scene_clr"] = infer_scene_measure(a3, dev, rank, 3)
        except Exception as exc:
            secondary["infer_scene_clr"] = {"error": f"{type(exc).__name__}: {str(exc)[:200]}"}
        torch.cuda.empty_cache()
        if not args.no_cpu_baseline:
            try:
                secondary["config0_mini_pose_cpu"] = config0_mini_pose(dev)
            except Exception as exc:
                secondary["config0_mini_pose_cpu"] = {"error": f"{type(exc).__name__}: {str(exc)[:200]}"}

    if rank == 0:
        e_avg = m["edges"] / args.steps
        roofline, kernels, (alg, exe, byts, bf, traffic, traffic_src) = workload_roofline(wl, m, args, args.steps)
        kernels_warmup = family_table(m["fam_all"], args.warmup, alg, exe, byts, bf, m["pair_us"]) if m["fam_all"] else {}
        roofline["note"] = ROOFLINE_NOTE
        roofline_enc = None
        if "point_feat" in kernels and "point_feat" in alg:
            roofline_enc = roofline_of("point_feat", kernels["point_feat"], alg, exe, byts, m["pair_us"], traffic, traffic_src)
            roofline_enc["note"] = "frozen encoders' point stacks (SURVEY.md 8f #1): next to the path, not the path"
        ms_step = 1e3 * dt / args.steps
        sf_kw = dict(f_l=wl.nl / wl.n_nodes, f_r=wl.nr / wl.n_nodes) if args.model == "clr" else {}
        step_bytes = wl.work.step_bytes(wl.n_nodes, e_avg)
        step_flops = wl.work.step_flops(wl.n_nodes, e_avg, **sf_kw)
        whole = {"algorithmic_bytes_per_step": step_bytes, "hbm_gbs": round(step_bytes / (ms_step * 1e-3) / 1e9, 1),
                 "hbm_frac": round(step_bytes / (ms_step * 1e-3) / 1e9 / PEAK_HBM_GBS, 5),
                 "algorithmic_flops_per_step": step_flops,
                 "fp32_tflops": round(step_flops / (ms_step * 1e-3) / 1e12, 2),
                 "fp32_frac": round(step_flops / (ms_step * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                 "note": "message-passing path only (SURVEY.md 8d): the frozen encoders' FLOPs are not counted"}
        # SURVEY.md 8d fractions NEXT to the operand fraction (round 6): `frac` above counts the bytes / FLOPs this design hands the
        # launch (saved activations + G tensors exist because the step stores forward and re-reads backward); these two count what
        # the reference's algorithm needs -- compulsory bytes of the WHOLE step over the step time, and the dominant family's
        # algorithmic (unhoisted, per-edge) FLOPs over its launch time against the exact-fp32 MFMA peak (above 1 where hoisting to
        # per-node work and bf16x6 products beat an fp32 evaluation of the reference's formula).
        roofline["sec8d_bytes_frac"] = whole["hbm_frac"]
        roofline["sec8d_flops_frac"] = round(roofline["algorithmic_tflops"] / PEAK_FP32_MFMA_TFLOPS, 4)
        roofline["sec8d_note"] = ("sec8d_bytes_frac = SURVEY.md 8d compulsory bytes of the whole step / step time / 8 TB/s; sec8d_flops_frac = the "
                                  "family's algorithmic FLOPs per launch / its launch time / 157.3 TFLOP/s (exact-fp32 MFMA peak); `frac` is the "
                                  "operand-byte (or executed-FLOP) fraction of the launch as designed, not a section-8d fraction")
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(wl)
        line = {"metric": "edges/sec (fwd+bwd) on nuScenes-shaped detection graphs",
                "value": round(total_edges / dt, 1), "unit": "edges/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(ms_step, 4), "higher_is_better": True,
                "scaling": args.scaling, "vs_baseline": None, "dtype": DTYPE if args.model == "clr" else "f32", "data": "synthetic",
                "config": {"workload": wl.describe(world), "graphs_per_gpu": graphs_per_gpu, "nodes_per_gpu": wl.n_nodes,
                           "edges_per_gpu": round(e_avg, 1), "frames": 5,
                           "lidar_rows_per_gpu": getattr(wl, "nl", None), "radar_rows_per_gpu": getattr(wl, "nr", None),
                           "dead_knn_gat_block_executed": bool(wl.model.run_dead_knn),
                           "parallelism": f"graph-batch sharding x{world}"},
                "roofline": roofline, "roofline_encoders": roofline_enc, "cpu_baseline": cpu, "whole_step": whole,
                "replay_vs_eager_loss": m["loss_check"],
                "ms_per_step_median": round(m["step_ms"][len(m["step_ms"]) // 2], 4) if m["step_ms"] else None,
                "secondary": secondary,
                "kernels": kernels, "kernels_instrumented_warmup": kernels_warmup,
                "library_sha16": lib_sha16(),
                "untimed_clock_ramp_steps": m["ramp_steps"], "host_enqueue_ms_per_step": round(1e3 * m["t_enqueue"] / args.steps, 4),
                "host_graph_launch_ms_median": round(1e3 * m.get("t_first", 0.0), 4),
                "host_prologue_ms_median": round(1e3 * m.get("t_pre", 0.0), 4),
                "timed_region": ("hipGraph replay (one captured training step per pool batch"
                                 + (": forward + backward graph, eager flat all-reduce, optimizer graph" if (world > 1 or args.force_collective) else "")
                                 + ("; the modality masks + row compaction of the next batch are part of the captured step, their counts checked on the device"
                                    if wl.row_mismatch is not None else
                                    ("; the modality masks + row compaction run eagerly in front of each replay and feed it" if wl.rows_static is not None else ""))
                                 + "); kernel families timed with HIP events in an eager pass of the same K steps right after it")
                                if m["graphs"] else ("eager" + (f" ({m['graph_note']})" if m["graph_note"] else ""))}
        print(json.dumps(line))
    if world > 1 or args.force_collective:
        dist.destroy_process_group()


def infer_scene_measure(args, dev, rank, steps, frames=24, per_frame=400):
    """Scene-level inference (predict.py:143-259 for one scene): `predict_post.predict_scene` over the stride-1 windows of a
    synthetic scene of `frames` x `per_frame` detections (5-frame windows of 2,000 nodes / ~20,000 edges) -- every detection
    encoded once per scene through the device-table EmbeddingCache, window forwards, window mean, thresholds, greedy flux and
    track clustering -- against the same pass with the encoders inside every window's forward (the reference's way).  Eager
    launches (window shapes differ), host synchronisations included; a step = one scene."""
    from batch3dmot_amd import encoders as enc_mod, synth
    from batch3dmot_amd.clr_att_gnn import GNN
    from batch3dmot_amd.predict_post import predict_scene
    torch.manual_seed(5621)
    model = GNN(enc_mod.ResNetAE(), enc_mod.PointNetClassifier(k=7), enc_mod.RadarNetClassifier(k=7)).to(dev).eval()
    model.run_dead_knn = not args.no_dead_knn
    scene, wins = synth.make_scene(frames=frames, per_frame=per_frame, k=20, scene_idx=rank, modalities=True)
    wins = [w.to(dev) for w in wins]
    node_cls = (scene.node_classes.long() - 1).to(dev)
    names = list(synth.CLASSES)
    edges = float(sum(w.edge_index.size(1) for w in wins))
    rows_windows = sum(w.pose_feats.size(0) for w in wins)
    out = {}
    for key, use_cache, wpf in (("cached", True, 8), ("uncached", False, 8), ("cached_one_window_per_forward", True, 1),
                                ("uncached_one_window_per_forward", False, 1)):
        r = None
        for _ in range(2):
            r = predict_scene(model, wins, node_cls, names, cache=use_cache, windows_per_forward=wpf)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            r = predict_scene(model, wins, node_cls, names, cache=use_cache, windows_per_forward=wpf)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        out[key] = {"ms_per_scene": round(1e3 * dt, 3), "ms_per_window": round(1e3 * dt / len(wins), 4), "edges_per_s": round(edges / dt, 1),
                    "kept_edges": int(r["kept_pairs"].size(0)), "tracks": len(r["tracks"]),
                    "encoder_rows": (dict(r["cache"].encoder_rows) if r["cache"] is not None else {"img": rows_windows})}
    return {"metric": "edges/sec (forward only), scene-level inference with post-processing", "unit": "edges/s",
            "value": out["cached"]["edges_per_s"], "value_uncached": out["uncached"]["edges_per_s"],
            "speedup_from_cache": round(out["uncached"]["ms_per_scene"] / out["cached"]["ms_per_scene"], 3),
            "config": {"workload": f"predict_post.predict_scene: scene of {frames} frames x {per_frame} detections, {len(wins)} stride-1 windows of "
                                   f"{wins[0].pose_feats.size(0)} nodes / ~{int(edges / len(wins))} edges (predict.py:143-259,262-375), eager launches",
                       "detections": int(scene.pose_feats.size(0)), "window_rows_total": rows_windows, "windows": len(wins),
                       "dead_knn_gat_block_executed": bool(model.run_dead_knn)},
            "cached": out["cached"], "uncached": out["uncached"],
            "one_window_per_forward": {"cached": out["cached_one_window_per_forward"], "uncached": out["uncached_one_window_per_forward"]},
            "windows_per_forward": 8, "steps": steps, "library_sha16": lib_sha16()}


def main_infer(args, dev, rank, world, dist):
    if getattr(args, "scene", False):
        # scenes shard over the ranks (predict.py:595-611 walks the scenes one after the other; they share nothing): every rank
        # runs its own scene, no collective on the data path; the line sums the ranks' rates over the slowest rank's time
        if world > 1:
            dist.barrier()
        line = infer_scene_measure(args, dev, rank, max(2, args.steps // 10))
        if world > 1:
            t = torch.tensor([line["cached"]["ms_per_scene"], line["uncached"]["ms_per_scene"]], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            edges = torch.tensor([line["value"] * line["cached"]["ms_per_scene"] * 1e-3], dtype=torch.float64, device=dev)   # edges of this rank's scene
            dist.all_reduce(edges, op=dist.ReduceOp.SUM)
            line["value"] = round(float(edges) / (float(t[0]) * 1e-3), 1)
            line["value_uncached"] = round(float(edges) / (float(t[1]) * 1e-3), 1)
            line["n_gpus"] = world
            line["config"]["parallelism"] = f"scenes sharded over {world} ranks (no collective on the data path)"
        if rank == 0:
            print(json.dumps(line))
        if world > 1:
            dist.destroy_process_group()
        return
    line = infer_measure(args, dev, rank, world, dist, args.steps, args.warmup, cpu=(world == 1 and not args.no_cpu_baseline))
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


def infer_measure(args, dev, rank, world, dist, steps, warmup, cpu=True):
    """BASELINE.json configs[4] (predict.py:172-196): forward only under no_grad, eval-mode encoders inside, windows of
    2,000 detections / ~20,000 edges; a step = 64 windows enqueued back to back on the launch stream (no host
    synchronisation inside: the modality row counts of the pool windows are read once, in front of the timed region).
    Replicas only: no collective; at N > 1 every rank runs its own windows and the edges are summed.  Returns the JSON
    line (rank 0; None elsewhere) with `roofline` (the path family with the most device time per window, HIP event pairs in
    an eager pass over the pool) and `cpu_baseline` (the oracle's forward on the host, bounded sample)."""
    from batch3dmot_amd import _lib, encoders as enc_mod, synth
    from batch3dmot_amd.clr_att_gnn import GNN
    from batch3dmot_amd.pose_gnn import GATConvParams, PoseGNN
    torch.manual_seed(5621)
    windows = 64
    clr = args.model == "clr"
    if clr:
        model = GNN(enc_mod.ResNetAE(), enc_mod.PointNetClassifier(k=7), enc_mod.RadarNetClassifier(k=7)).to(dev).eval()
    else:
        model = PoseGNN().to(dev).eval()
    model.run_dead_knn = not args.no_dead_knn
    pool_cpu = [synth.make_graph(2000, 20000, graph_idx=rank * 1000 + 300 + i, modalities=clr) for i in range(8)]
    pool = [b.to(dev) for b in pool_cpu]
    rows = [model.modality_rows(b) for b in pool] if clr else None
    edges = [b.edge_index.size(1) for b in pool]

    ahead = None
    if clr and getattr(args, "encode_ahead", False):
        # train_step.EncodeAhead in a serving loop: the encoders of the NEXT window on a side stream under this window's GNN
        # forward (one encoder pass and one forward per window, as in the sequential loop; static output buffers per pool window)
        from batch3dmot_amd.train_step import EncodeAhead
        ahead = EncodeAhead(model)
        f32, i32 = dict(dtype=torch.float32, device=dev), dict(dtype=torch.int32, device=dev)
        enc_static = [(torch.zeros(2000, 96, **f32), torch.zeros(r[0].numel(), 256, **f32), torch.zeros(r[0].numel(), **i32),
                       torch.zeros(r[1].numel(), 256, **f32), torch.zeros(r[1].numel(), **i32)) for r in rows]
        with torch.no_grad():
            ahead.launch(pool[0], rows=rows[0], static=enc_static[0])
            ahead.take(pool[0])

    def window(k):
        b = pool[k]
        if hasattr(b, "_b3d_graph"):
            del b._b3d_graph                     # the CSR/CSC build is part of every window
        if ahead is not None:
            kn = (k + 1) % len(pool)
            ahead.launch(pool[kn], rows=rows[kn], static=enc_static[kn])
            out = model(b, encoded=enc_static[k])
            ahead.take(pool[kn])
            return out
        return model(b, rows=rows[k]) if clr else model(b)

    with torch.no_grad():
        for k in range(len(pool)):
            window(k)
        torch.cuda.synchronize()
        # one hipGraph per pool window (eager enqueue of ~150 launches costs as much host time as the window takes on
        # the GPU); the outputs are the graphs' static tensors, as a serving loop would read them
        graphs, keep, note = None, [], None
        if not args.no_graph:
            try:
                graphs = []
                cap_stream = torch.cuda.Stream()
                cap_stream.wait_stream(torch.cuda.current_stream())
                for k in range(len(pool)):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=cap_stream, capture_error_mode="thread_local"):
                        keep.append(window(k))
                    graphs.append(g)
                torch.cuda.current_stream().wait_stream(cap_stream)
                torch.cuda.synchronize()
            except Exception as exc:
                graphs, note = None, f"hipGraph capture failed ({type(exc).__name__}: {str(exc)[:160]})"
                torch.cuda.synchronize()

        def step(i):
            for w in range(windows):
                k = (i * windows + w) % len(pool)
                if graphs is not None:
                    graphs[k].replay()
                else:
                    window(k)

        for i in range(max(1, warmup // 4)):
            step(i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        my_edges = float(sum(edges[(i * windows + w) % len(pool)] for i in range(steps) for w in range(windows)))
        # the same 64 windows per step, EIGHT per forward as one disjoint-union graph (predict_post.predict_scene scores a scene's
        # windows this way: every kernel is per node / per edge / per node's own edge list, so the scores are those of one forward
        # per window -- tests/test_scene_e2e.py); one captured forward over the pool, 8 replays per step
        batched = None
        if clr and ahead is None and world == 1:
            try:
                from batch3dmot_amd.predict_post import _union_graph
                big = _union_graph(pool, with_sensor_feats=True)
                big_rows = model.modality_rows(big)
                ref_scores = torch.cat([model(b, rows=r)[0].reshape(-1) for b, r in zip(pool, rows)])
                got = model(big, rows=big_rows)[0].reshape(-1)
                same = bool(torch.allclose(got, ref_scores, rtol=0, atol=1e-6))
                gb = torch.cuda.CUDAGraph()
                cs = torch.cuda.Stream()
                cs.wait_stream(torch.cuda.current_stream())
                if hasattr(big, "_b3d_graph"):
                    del big._b3d_graph
                with torch.cuda.graph(gb, stream=cs, capture_error_mode="thread_local"):
                    if hasattr(big, "_b3d_graph"):
                        del big._b3d_graph
                    keep.append(model(big, rows=big_rows))
                torch.cuda.current_stream().wait_stream(cs)
                for _ in range(3):
                    gb.replay()
                torch.cuda.synchronize()
                tb = time.perf_counter()
                for _ in range(steps * (windows // len(pool))):
                    gb.replay()
                torch.cuda.synchronize()
                dtb = time.perf_counter() - tb
                batched = {"windows_per_forward": len(pool), "edges_per_s": round(my_edges / dtb, 1),
                           "ms_per_window": round(1e3 * dtb / steps / windows, 4), "scores_equal_to_one_window_per_forward": same}
            except Exception as exc:
                batched = {"error": f"{type(exc).__name__}: {str(exc)[:200]}"}
        # kernel families: HIP event pairs around every launch in an eager pass over the pool (4 x 8 windows), on the launch stream
        fam_passes = 4
        _lib.prof_enable(True)
        for _ in range(fam_passes):
            for k in range(len(pool)):
                window(k)
        torch.cuda.synchronize()
        fam = _lib.prof_read()
        _lib.prof_enable(False)
        pair_us = 0.5 * _lib.prof_pair_overhead_us(torch.cuda.current_stream(dev).cuda_stream)
        # the k-NN + GAT block alone at n_t = 400, D = 96, k = 20 (5 frames of 400 detections)
        x = torch.randn(2000, 96, device=dev)
        ts = torch.arange(5, device=dev).repeat_interleave(400)
        conv = GATConvParams(96).to(dev)
        for _ in range(5):
            _lib.knn_gat(x, ts, conv, k=20)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(50):
            _lib.knn_gat(x, ts, conv, k=20)
        torch.cuda.synchronize()
        knn_ms = 1e3 * (time.perf_counter() - t1) / 50
    dtm, tot = reduce_over_ranks({"dt": dt, "edges": my_edges}, dev, world, dist)
    if rank != 0:
        return None
    name = "camera+LiDAR+radar GNN (clr_att_gnn), eval-mode encoders inside" if clr else "poses-only PoseGNN"
    if ahead is not None:
        name += " (the encoders of window k + 1 on a side stream under the forward of window k: train_step.EncodeAhead)"
    # roofline of the forward-only path: a "step" of the family table is ONE window
    e_avg = sum(edges) / len(edges)
    work = ClrWork if clr else PoseWork
    kw = dict(n=2000, e=e_avg, depth=model.depth, training=False)
    if clr:
        hoist = _lib.features()
        kw.update(hoist_mp=bool(hoist.get("clr_hoist_mp")), hoist_att=bool(hoist.get("clr_hoist_att")),
                  nl=sum(int(r[0].numel()) for r in rows) / len(rows), nr=sum(int(r[1].numel()) for r in rows) / len(rows))
    alg, exe, byts, bf = work.families(**kw)
    kernels = family_table(fam, fam_passes * len(pool), alg, exe, byts, bf, pair_us)
    fwd_fams = [k for k in kernels if k in ("mp_edge_fwd", "mp_node_fwd", "att_fwd")]
    roofline = None
    if fwd_fams:
        dom = max(fwd_fams, key=lambda k: kernels[k]["us_per_step"])
        traffic, traffic_src = load_traffic(f"{args.model}:infer:knn{int(not args.no_dead_knn)}")
        roofline = roofline_of(dom, kernels[dom], alg, exe, byts, pair_us, traffic, traffic_src)
        roofline["launches_per_window"] = roofline.pop("launches_per_step")
        roofline["note"] = ("the SURVEY.md 8a forward family with the most device time per window; accounting as in the training line's "
                            "`roofline` (executed fp32-equivalent FLOPs / event-pair device time / the MFMA peak of the instructions it runs)")
    cpu_b = infer_cpu_baseline(args, pool_cpu, clr) if cpu else None
    return {
        "metric": "edges/sec (forward only) on nuScenes-shaped detection graphs", "value": round(tot / dtm, 1), "unit": "edges/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(1e3 * dtm / steps, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE if clr else "f32", "data": "synthetic",
        "config": {"workload": f"large-batch inference (predict.py:172-196): {name}, depth 6, forward under no_grad, "
                               f"{windows} windows of 2,000 nodes / ~20,000 edges in flight per GPU and step, CSR/CSC build per window",
                   "windows_per_step": windows, "nodes_per_window": 2000, "edges_per_window": round(e_avg, 1),
                   "dead_knn_gat_block_executed": bool(model.run_dead_knn), "parallelism": f"replicas x{world} (no collective)"},
        "roofline": roofline, "cpu_baseline": cpu_b, "eight_windows_per_forward": batched,
        "ms_per_window": round(1e3 * dtm / steps / windows, 4),
        "host_enqueue_ms_per_window": round(1e3 * t_enq / steps / windows, 4),
        "kernels_per_window": kernels,
        "knn_gat_block_nt400_d96_k20_ms": round(knn_ms, 4), "library_sha16": lib_sha16(),
        "timed_region": "hipGraph replay (one captured forward per pool window)" if graphs is not None else "eager" + (f" ({note})" if note else "")}


def infer_cpu_baseline(args, pool_cpu, clr):
    """The oracle's forward (oracle/ref_torch.py; eval mode, no_grad, encoders inside, the discarded k-NN + GAT block included)
    on the host cores over the same pool windows: bounded sample, median per-window time."""
    from oracle import ref_encoders, ref_torch   # checker / baseline only
    from batch3dmot_amd import encoders as enc_mod
    model_name, cores = host_info()
    threads = min(16, cores)
    torch.set_num_threads(threads)
    torch.manual_seed(5621)
    if clr:
        mm = ref_torch.GNN(ref_encoders.ResNetAE(), ref_encoders.PointNetClassifier(k=7), ref_encoders.RadarNetClassifier(k=7),
                           run_dead_knn=not args.no_dead_knn, loop_masks=False).eval()
    else:
        mm = ref_torch.PoseGNN(run_dead_knn=not args.no_dead_knn).eval()
    per_edge, t_all = [], time.perf_counter()
    with torch.no_grad():
        for i in range(2):
            mm(pool_cpu[i % len(pool_cpu)])
        n_timed = 8 if clr else 24
        for i in range(n_timed):
            b = pool_cpu[(2 + i) % len(pool_cpu)]
            t0 = time.perf_counter()
            mm(b)
            per_edge.append((time.perf_counter() - t0) / b.edge_index.size(1))
    per_edge.sort()
    return {"value": round(1.0 / per_edge[len(per_edge) // 2], 1), "unit": "edges/s", "cores": threads, "kind": "port",
            "sample": f"median of {n_timed} forward passes over the same 2,000-node / ~20,000-edge windows (2 warm-up), "
                      f"{time.perf_counter() - t_all:.1f} s, torch {torch.__version__} CPU",
            "host_cpu": model_name, "host_logical_cores": cores}


def reduce_over_ranks(m, dev, world, dist):
    """(max over ranks of the timed region, sum over ranks of the processed edges)."""
    if world == 1:
        return m["dt"], float(m["edges"])
    tot = torch.tensor([m["dt"], float(m["edges"])], dtype=torch.float64, device=dev)
    tmax = tot[:1].clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    esum = tot[1:].clone()
    dist.all_reduce(esum, op=dist.ReduceOp.SUM)
    return float(tmax), float(esum)


def main_stub(args, rank, world, dist):
    dev = torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    wl = StubWorkload(dev, rank, world)
    m = measure(wl, args, world, dist, args.steps, args.warmup, args.ramp_ms, use_graph=False)
    dt, total_edges = reduce_over_ranks(m, dev, world, dist)
    if rank == 0:
        print(json.dumps({"metric": "edges/sec (fwd+bwd) on nuScenes-shaped detection graphs", "value": round(total_edges / dt, 1),
                          "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": {"workload": wl.describe(world)},
                          "edges_summed_over_ranks": total_edges, "untimed_clock_ramp_steps": m["ramp_steps"]}))
    if world > 1:
        dist.destroy_process_group()


def host_info():
    model, cores = "unknown", os.cpu_count() or 1
    try:
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                if ln.startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return model, cores


def cpu_baseline(wl: Workload):
    """The oracle (oracle/ref_torch.py, pinned against the reference sources) timed on the host: the same batches, the
    same step (masks, encoders in train mode, forward incl. the discarded k-NN + GAT block, loss, backward, Adam).
    Bounded sample: a few steps at the fastest thread count, plus one 1-thread figure on a smaller sample."""
    from oracle import ref_encoders, ref_torch   # checker / baseline only
    from batch3dmot_amd import encoders as enc_mod, synth
    model_name, cores = host_info()
    threads = min(16, cores)                     # replaced by the fastest setting of the sweep below

    def build():
        torch.manual_seed(5621)
        if wl.kind == "pose":
            mm = ref_torch.PoseGNN(run_dead_knn=True)
        else:
            mm = ref_torch.GNN(ref_encoders.ResNetAE(), ref_encoders.PointNetClassifier(k=7), ref_encoders.RadarNetClassifier(k=7),
                               run_dead_knn=True, loop_masks=False)
        oo = torch.optim.Adam([p for p in mm.parameters() if p.requires_grad], lr=1e-4, weight_decay=1e-4, betas=(0.9, 0.999))
        return mm, oo

    def run(nthreads, batches, warm, steps):
        """(edges/s from the MEDIAN step time, seconds spent)"""
        torch.set_num_threads(nthreads)
        mm, oo = build()
        t_all = time.perf_counter()
        for i in range(warm):
            ref_torch.train_step(mm, batches[i % len(batches)], oo, batch_size=wl.graphs, loss_kind="cb", logits=wl.logits)
        per_edge = []
        for i in range(steps):
            b = batches[(warm + i) % len(batches)]
            t0 = time.perf_counter()
            ref_torch.train_step(mm, b, oo, batch_size=wl.graphs, loss_kind="cb", logits=wl.logits)
            per_edge.append((time.perf_counter() - t0) / b.edge_index.size(1))
        per_edge.sort()
        return 1.0 / per_edge[len(per_edge) // 2], time.perf_counter() - t_all

    # SURVEY.md 8d asks for os.cpu_count() threads and for 1 thread.  A short sweep (1 warm-up + 2 steps each) shows where this
    # graph size stops scaling on the box's cores; the sample proper runs at the fastest setting of the sweep.
    sweep, t_sweep = {}, time.perf_counter()
    # (all 256 logical cores of the GPU box were measured once, profiles/r04_a_bench_default.json: 112 edges/s, 95 x slower than
    # 16 threads -- ~280 s per step of oversubscribed small GEMMs -- so the sweep stops at 64)
    for t in sorted({min(16, cores), min(32, cores), min(64, cores)}):
        if time.perf_counter() - t_sweep > 40.0:              # bounded: the line must not wait minutes for an oversubscribed point
            sweep[t] = None
            continue
        try:
            sweep[t] = round(run(t, wl.pool_cpu, 1, 2)[0], 1)
        except Exception:                                     # a thread count the host refuses must not cost the line
            pass
    if any(v_ for v_ in sweep.values()):
        threads = max((t for t in sweep if sweep[t]), key=lambda t: sweep[t])
    # SURVEY.md 8d protocol, bounded to ~1 minute of CPU work: >= 2 warm-up steps, median of >= 10 (40 for the small model)
    if wl.kind == "pose":
        v, dt = run(threads, wl.pool_cpu, 5, 40)
        sample = f"median of 40 training steps of the same batches (5 warm-up), {dt:.1f} s"
        one = [synth.make_batch(2, 1500, 15000, first_graph_idx=0)]
        v1, dt1 = run(1, one, 2, 10)
        sample1 = f"median of 10 training steps of one batch (2 warm-up), {dt1:.1f} s"
    else:
        v, dt = run(threads, wl.pool_cpu, 2, 10)
        sample = f"median of 10 training steps of the same batches (2 warm-up), {dt:.1f} s"
        one = [synth.make_batch(1, 750, 7500, first_graph_idx=0, modalities=True)]
        if wl.modalities == "cl":
            one[0].radar_feats = torch.zeros_like(one[0].radar_feats)
        v1, dt1 = run(1, one, 2, 10)
        sample1 = f"median of 10 training steps of a 750-node / {one[0].edge_index.size(1)}-edge graph (2 warm-up), {dt1:.1f} s"
    return {"value": round(v, 1), "unit": "edges/s", "cores": threads, "kind": "port", "sample": sample + f", torch {torch.__version__} CPU",
            "one_thread": {"value": round(v1, 1), "unit": "edges/s", "cores": 1, "sample": sample1},
            "thread_sweep_edges_per_s": {str(t): v_ for t, v_ in sweep.items()},
            "host_cpu": model_name, "host_logical_cores": cores}


def config0_mini_pose(dev):
    """BASELINE.json configs[0] (`mini_config.yaml` poses-only: ONE nuScenes-mini scene graph, the reference's PyTorch forward on CPU --
    plumbing): the oracle's PoseGNN forward (`pose_gnn.py:58-86` restated, dead k-NN block executed) on a mini-sized synthetic window
    (5 frames x ~60 detections) on the host cores, the HIP forward of the same graph beside it, and their difference."""
    from batch3dmot_amd import synth
    from batch3dmot_amd.pose_gnn import PoseGNN
    from oracle import ref_torch            # checker / CPU baseline only
    from oracle.seeded import seeded_fill_
    data = synth.make_graph(300, 3000, graph_idx=77)
    ora = ref_torch.PoseGNN(run_dead_knn=True)
    seeded_fill_(ora, 5621)
    threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    with torch.no_grad():
        for _ in range(2):
            o_ref, _x = ora(data)
        ts = []
        for _ in range(10):
            t0 = time.perf_counter()
            ora(data)
            ts.append(time.perf_counter() - t0)
    ts.sort()
    cpu_ms = 1e3 * ts[len(ts) // 2]
    m = PoseGNN().to(dev).eval()
    m.load_state_dict(ora.state_dict())
    m.run_dead_knn = True
    dd = data.to(dev)
    with torch.no_grad():
        for _ in range(3):
            o_hip, _x = m(dd)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            m(dd)
        e1.record()
        torch.cuda.synchronize()
    hip_ms = e0.elapsed_time(e1) / 20
    e = int(data.edge_index.size(1))
    err = float((o_hip.cpu().double() - o_ref.double()).abs().max() / o_ref.double().abs().max().clamp_min(1e-30))
    return {"workload": "BASELINE.json configs[0]: poses-only PoseGNN forward on one mini-sized scene window (5 frames), CPU plumbing case",
            "nodes": int(data.pose_feats.size(0)), "edges": e,
            "cpu_forward_ms_median": round(cpu_ms, 3), "cpu_edges_per_s": round(e / (cpu_ms * 1e-3), 1), "cores": threads, "kind": "port",
            "hip_forward_ms_eager": round(hip_ms, 4), "hip_edges_per_s": round(e / (hip_ms * 1e-3), 1),
            "hip_vs_cpu_max_rel_err": err}


if __name__ == "__main__":
    main()
