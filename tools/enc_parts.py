import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from batch3dmot_amd import encoders, synth
from batch3dmot_amd.clr_att_gnn import GNN
dev = torch.device("cuda:0"); torch.manual_seed(0)
g = GNN(encoders.ResNetAE(), encoders.PointNetClassifier(k=7), encoders.RadarNetClassifier(k=7)).to(dev).eval()
b = synth.make_graph(2000, 20000, graph_idx=200, modalities=True).to(dev)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return round(1e3 * (time.perf_counter() - t0) / n, 3)
with torch.no_grad():
    lid = b.lidar_feats[b.lidar_feats.reshape(2000, -1).abs().sum(1) > 0].view(-1, 3, 128)
    rad = b.radar_feats[b.radar_feats.reshape(2000, -1).abs().sum(1) > 0].view(-1, 4, 64)
    print("resnet.encode", t(lambda: g.resnet.encode(b.img_feats)), "ms;", "pointnet", lid.shape[0], t(lambda: g.pointnet.forward_feat(lid)), "ms;",
          "radarnet", rad.shape[0], t(lambda: g.radarnet.forward_feat(rad)), "ms;", "encode_modalities", t(lambda: g.encode_modalities(b)), "ms")
