"""Secondary measurement (BASELINE.json configs 3/4): GNN (camera+LiDAR+radar) training step on one MI355X,
3,000 nodes / ~30,000 edges, with the frozen encoders (PyTorch-ROCm) and with their outputs precomputed."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from batch3dmot_amd import _lib, encoders, synth
from batch3dmot_amd.clr_att_gnn import GNN
from batch3dmot_amd.train_step import make_optimizer, train_step

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
torch.manual_seed(5621)
m = GNN(encoders.ResNetAE(), encoders.PointNetClassifier(k=7), encoders.RadarNetClassifier(k=7)).to(dev).train()
opt = make_optimizer(m)
pool = [synth.make_batch(2, 1500, 15000, first_graph_idx=2 * i, modalities=True).to(dev) for i in range(2)]
res = {}
for mode in ("with_encoders", "encoders_precomputed"):
    enc = None
    if mode == "encoders_precomputed":
        enc = [m.encode_modalities(b) for b in pool]
    def step(i):
        b = pool[i % len(pool)]
        if hasattr(b, "_b3d_graph"):
            del b._b3d_graph
        gt = b.y.float()
        out, _ = m(b, encoded=None if enc is None else enc[i % len(pool)])
        loss = torch.nn.functional.binary_cross_entropy(out.squeeze(1), gt, weight=b.edge_weights) / 2
        opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
    for i in range(5): step(i)
    torch.cuda.synchronize(); _lib.prof_enable(True); t0 = time.perf_counter()
    for i in range(steps): step(i)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    fam = _lib.prof_read(); _lib.prof_enable(False)
    E = pool[0].edge_index.size(1)
    res[mode] = {"ms_per_step": round(1e3 * dt / steps, 3), "edges_per_s": round(E * steps / dt, 1),
                 "kernels_us_per_step": {k: round(1e3 * v[0] / steps, 1) for k, v in fam.items() if v[1]}}
print(json.dumps({"model": "GNN (clr_att_gnn) training step", "nodes": 3000, "edges": E, **res}))
