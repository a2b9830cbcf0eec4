"""Host-side enqueue cost of one training step, by component (no device syncs inside the loop)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from batch3dmot_amd import _lib, synth
from batch3dmot_amd.pose_gnn import PoseGNN
from batch3dmot_amd.train_step import make_optimizer, fused_edge_loss

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = PoseGNN().to(dev); model.train()
model.run_dead_knn = "--no-knn" not in sys.argv
opt = make_optimizer(model, flat="--torch-adam" not in sys.argv)
pool = [synth.make_batch(2, 1500, 15000, first_graph_idx=2 * i).to(dev) for i in range(4)]
prof = "--prof" in sys.argv
_lib.prof_enable(prof)
acc = {}
def tick(name, t0):
    t1 = time.perf_counter(); acc[name] = acc.get(name, 0.0) + (t1 - t0); return t1
for it in range(60):
    if it == 20:
        torch.cuda.synchronize(); acc = {}; T0 = time.perf_counter()
    b = pool[it % 4]
    if hasattr(b, "_b3d_graph"): del b._b3d_graph
    t = time.perf_counter()
    out, aux = model(b); t = tick("forward (graph build + b3d_pose_forward)", t)
    opt.zero_grad(); t = tick("zero_grad", t)
    loss, g = fused_edge_loss(out, b, 2, "cb", True); t = tick("loss", t)
    out.backward(g); t = tick("backward", t)
    opt.step(); t = tick("adam", t)
enq = time.perf_counter() - T0
torch.cuda.synchronize()
tot = time.perf_counter() - T0
print(f"prof={prof} knn={model.run_dead_knn}: enqueue {enq/40*1e3:.3f} ms/step, total {tot/40*1e3:.3f} ms/step")
for k, v in acc.items():
    print(f"  {k:45s} {v/40*1e6:8.1f} us")
