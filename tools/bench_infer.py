"""BASELINE.json config 5 on ONE GPU (the driver's 8-GPU figure is 8 independent replicas of this: inference has no
collective): forward-only edges/s over windows of 2,000 detections / ~20,000 edges, poses-only and
camera+LiDAR+radar models, plus the stand-alone k-NN + GAT block at n_t = 400, D = 96, k = 20.
python tools/bench_infer.py [windows]"""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from batch3dmot_amd import _lib, encoders, synth
from batch3dmot_amd.clr_att_gnn import GNN, EmbeddingCache
from batch3dmot_amd.pose_gnn import GATConvParams, PoseGNN

windows = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
torch.manual_seed(5621)


def run(fn, pool, reps):
    for i in range(8):
        fn(pool[i % len(pool)])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(reps):
        fn(pool[i % len(pool)])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    e = sum(pool[i % len(pool)].edge_index.size(1) for i in range(reps))
    return {"ms_per_window": round(1e3 * dt / reps, 3), "edges_per_s": round(e / dt, 1)}


res = {"window": {"nodes": 2000, "edges_target": 20000, "frames": 5}}
with torch.no_grad():
    pool = [synth.make_graph(2000, 20000, graph_idx=100 + i).to(dev) for i in range(8)]
    m = PoseGNN().to(dev).eval()
    def fwd_pose(b):
        if hasattr(b, "_b3d_graph"): del b._b3d_graph
        return m(b)
    res["pose_gnn_forward"] = run(fwd_pose, pool, windows)
    m.run_dead_knn = False
    res["pose_gnn_forward_without_dead_knn"] = run(fwd_pose, pool, windows)

    poolc = [synth.make_graph(2000, 20000, graph_idx=200 + i, modalities=True).to(dev) for i in range(4)]
    for i, b in enumerate(poolc):                                  # global ids: window i holds detections [400 i, 400 i + 2000)
        gid = torch.arange(b.pose_feats.size(0), device=dev) + 400 * i
        b.global_node_timestamps = torch.stack([gid.float(), b.node_timestamps.float()], 1)
    g = GNN(encoders.ResNetAE(), encoders.PointNetClassifier(k=7), encoders.RadarNetClassifier(k=7)).to(dev).eval()
    def fwd_clr(b):
        if hasattr(b, "_b3d_graph"): del b._b3d_graph
        return g(b)
    res["clr_gnn_forward_with_encoders"] = run(fwd_clr, poolc, max(8, windows // 4))
    for mod in list(g.pointnet.modules()) + list(g.radarnet.modules()):
        mod.use_hip = False
    res["clr_gnn_forward_with_encoders_pytorch_point_stacks"] = run(fwd_clr, poolc, max(8, windows // 4))
    for mod in list(g.pointnet.modules()) + list(g.radarnet.modules()):
        mod.use_hip = True
    enc = [g.encode_modalities(b) for b in poolc]
    k = {"i": 0}
    def fwd_clr_pre(b):
        if hasattr(b, "_b3d_graph"): del b._b3d_graph
        i = k["i"]; k["i"] += 1
        return g(b, encoded=enc[i % len(poolc)])
    res["clr_gnn_forward_encoders_precomputed"] = run(fwd_clr_pre, poolc, max(8, windows // 4))
    cache = EmbeddingCache()
    def fwd_clr_cached(b):
        if hasattr(b, "_b3d_graph"): del b._b3d_graph
        return g(b, encoded=g.encode_modalities(b, cache=cache))
    res["clr_gnn_forward_embedding_cache_warm"] = run(fwd_clr_cached, poolc, max(8, windows // 4))

    x = torch.randn(2000, 96, device=dev)
    ts = torch.arange(5, device=dev).repeat_interleave(400)
    conv = GATConvParams(96).to(dev)
    for _ in range(5): _lib.knn_gat(x, ts, conv, k=20)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): _lib.knn_gat(x, ts, conv, k=20)
    torch.cuda.synchronize()
    res["knn_gat_block_nt400_d96_k20"] = {"ms": round(1e3 * (time.perf_counter() - t0) / 50, 4)}
print(json.dumps(res))
