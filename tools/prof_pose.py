"""Run the bench workload a few times (for rocprofv3)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from batch3dmot_amd import synth
from batch3dmot_amd.pose_gnn import PoseGNN

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dead = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
torch.manual_seed(5621)
m = PoseGNN().to(dev)
m.run_dead_knn = bool(dead)
big = synth.make_batch(2, 1500, 15000).to(dev)
E = big.edge_index.size(1)
lw = torch.randn(E, 1, device=dev)
for it in range(iters):
    torch.cuda.synchronize(); t0 = time.time()
    if hasattr(big, "_b3d_graph"):
        del big._b3d_graph
    m.zero_grad(set_to_none=True)
    out, x_enc = m(big)
    (out * lw).sum().backward()
    torch.cuda.synchronize(); t1 = time.time()
    if it >= iters - 3:
        print(f"iter {it}: {1e3*(t1-t0):.3f} ms -> {E/(t1-t0)/1e6:.2f} M edges/s")
