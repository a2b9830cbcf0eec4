"""Phase time stamps of the node kernels (library built with -DB3D_EXP_STAMPS): python tools/phase_stamps.py"""
import ctypes as C, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from batch3dmot_amd import synth, _lib
from batch3dmot_amd.pose_gnn import PoseGNN

dev = torch.device("cuda:0")
torch.manual_seed(5621)
m = PoseGNN().to(dev)
m.run_dead_knn = False
big = synth.make_batch(2, 1500, 15000).to(dev)
E = big.edge_index.size(1)
lw = torch.randn(E, 1, device=dev)
for it in range(5):
    m.zero_grad(set_to_none=True)
    out, _ = m(big)
    (out * lw).sum().backward()
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros((4, 512, 32), dtype=np.int64)
assert lib.b3d_debug_stamps(buf.ctypes.data_as(C.c_void_p)) == 0
names = {0: ["start", "segsum", "L0 128>96", "L1 96>64", "L2 64>48", "proj 48>432"],
         1: ["start", "act loads", "list sums", "4 products", "L4 48>64", "L5 64>96", "L6 96>128"],
         2: ["start", "gathers", "L0 32>96", "L1 96>64", "L2 64>32", "L3 32>96", "L4 96>64", "L5 32>96", "L6 96>64", "last store"],
         3: ["start", "loads", "L0 64>96", "L1 96>32", "L2 64>96", "L3 96>32", "L4 32>64", "L5 64>96", "L6 96>32+"]}
E_tiles = (E + 127) // 128
for k in (0, 1, 2, 3):
    nwg = ((big.pose_feats.size(0) + 15) // 16) if k < 2 else E_tiles
    s = buf[k, :nwg, :len(names[k])].astype(np.float64) * 0.01          # 100 MHz -> us
    t0 = s[:, 0].min()
    print(["node_fwd_h", "node_bwd_h", "edge_fwd_h", "edge_bwd_h"][k], f"workgroups {nwg}: start spread {s[:,0].max()-t0:.2f} us, end: mean {(s[:,-1]-t0).mean():.2f} max {(s[:,-1]-t0).max():.2f} us")
    d = np.diff(s, axis=1)
    for i, n in enumerate(names[k][1:]):
        print(f"   {n:14s} mean {d[:, i].mean():6.2f}  p50 {np.median(d[:, i]):6.2f}  max {d[:, i].max():6.2f} us")

# inside the layers of edge_fwd_h: [previous layer done] -> acquire (vmcnt + barrier) -> hook -> MFMAs
s = buf[2, :E_tiles].astype(np.float64) * 0.01
print("edge_fwd_h, layers 1..6: acquire wait | hook issue | MFMAs (us, mean over workgroups)")
for li, (h0, h1) in enumerate([(10, 11), (12, 13), (14, 15), (16, 17), (18, 19), (20, 21)], start=1):
    prev, nxt = s[:, 1 + li], s[:, 2 + li]
    print(f"   L{li}: {np.mean(s[:, h0] - prev):5.2f} | {np.mean(s[:, h1] - s[:, h0]):5.2f} | {np.mean(nxt - s[:, h1]):5.2f}")
