"""Phase times of predict_post.predict_scene on the synthetic 24-frame scene (synchronising between phases)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from batch3dmot_amd import encoders, synth, predict_post
from batch3dmot_amd.clr_att_gnn import GNN, EmbeddingCache

dev = torch.device("cuda:0")
torch.manual_seed(5621)
m = GNN(encoders.ResNetAE(), encoders.PointNetClassifier(k=7), encoders.RadarNetClassifier(k=7)).to(dev).eval()
scene, wins = synth.make_scene(frames=24, per_frame=400, k=20)
wins = [w.to(dev) for w in wins]
node_cls = (scene.node_classes.long() - 1).to(dev)
names = list(synth.CLASSES)


def T():
    torch.cuda.synchronize()
    return time.perf_counter()


for rep in range(3):
    t0 = T()
    c = EmbeddingCache()
    c.add_windows(m, wins)
    t1 = T()
    enc = c.windows([w.global_ids for w in wins])
    t2 = T()
    r = predict_post.predict_scene(m, wins, node_cls, names, cache=c, tracks=False)
    t3 = T()
    r2 = predict_post.predict_scene(m, wins, node_cls, names, cache=c, tracks=True)
    t4 = T()
    pairs = torch.cat([torch.stack([w.global_ids[w.edge_index[0]], w.global_ids[w.edge_index[1]]], 1) for w in wins])
    sc = torch.rand(pairs.size(0), device=dev)
    t5 = T()
    predict_post.greedy_edges_hip(pairs, sc, node_cls, names)
    t6 = T()
    print(f"cache fill {1e3*(t1-t0):.1f} ms | window tuples {1e3*(t2-t1):.1f} | scene without tracks (cache warm) {1e3*(t3-t2):.1f} | "
          f"with tracks {1e3*(t4-t3):.1f} | greedy_edges_hip alone {1e3*(t6-t5):.1f}")
