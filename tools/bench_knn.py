"""k-NN + GAT block alone at the bench shape (run under rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from batch3dmot_amd import _lib, synth
from batch3dmot_amd.pose_gnn import GATConvParams
dev = torch.device("cuda:0")
d = synth.make_batch(2, 1500, 15000).to(dev)
x = torch.randn(d.pose_feats.size(0), 48, device=dev)
conv = GATConvParams(48).to(dev)
for k in (20, 1):
    for _ in range(12):
        _lib.knn_gat(x, d.node_timestamps, conv, k=k)
torch.cuda.synchronize()
print("ok")
