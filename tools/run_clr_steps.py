"""A few forward + backward passes of the camera+LiDAR+radar model on the benchmark-sized batch (for rocprofv3 --pmc runs):
python tools/run_clr_steps.py [steps]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from batch3dmot_amd import synth, encoders
from batch3dmot_amd.clr_att_gnn import GNN

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
torch.manual_seed(5621)
m = GNN(encoders.ResNetAE(), encoders.PointNetClassifier(k=7), encoders.RadarNetClassifier(k=7)).to(dev)
m.run_dead_knn = False
m.train()
big = synth.make_batch(2, 1500, 15000, modalities=True).to(dev)
lw = torch.randn(big.edge_index.size(1), 1, device=dev)
for it in range(steps):
    m.zero_grad(set_to_none=True)
    out, _ = m(big)
    (out * lw).sum().backward()
torch.cuda.synchronize()
print("done")
