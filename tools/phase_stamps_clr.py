"""Phase time stamps of the hoisted edge kernels of the camera+LiDAR+radar model (library built with
EXTRA=-DB3D_EXP_STAMPS): python tools/phase_stamps_clr.py"""
import ctypes as C, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from batch3dmot_amd import synth, _lib, encoders
from batch3dmot_amd.clr_att_gnn import GNN

dev = torch.device("cuda:0")
torch.manual_seed(5621)
m = GNN(encoders.ResNetAE(), encoders.PointNetClassifier(k=7), encoders.RadarNetClassifier(k=7)).to(dev)
m.run_dead_knn = False
m.train()
big = synth.make_batch(2, 1500, 15000, modalities=True).to(dev)
E = big.edge_index.size(1)
lw = torch.randn(E, 1, device=dev)
for it in range(4):
    m.zero_grad(set_to_none=True)
    out, _ = m(big)
    (out * lw).sum().backward()
torch.cuda.synchronize()
lib = _lib.load()
lib = C.CDLL(_lib.LIB_PATH)
buf = np.zeros((4, 512, 32), dtype=np.int64)
assert lib.b3d_debug_stamps_clr(buf.ctypes.data_as(C.c_void_p)) == 0
names = {2: ["start", "gathers", "L0 128>256", "L1 256>128", "L2 128>64", "L3 64>192", "L4 192>128", "L5 64>192", "L6 192>128", "last store"],
         3: ["start", "loads", "L0 128>192", "L1 192>64", "L2 128>192", "L3 192>64", "L4 64>128", "L5 128>256", "L6 256>128"]}
E_tiles = (E + 127) // 128
for k in (2, 3):
    nwg = min(E_tiles, 512)
    s = buf[k, :nwg, :len(names[k])].astype(np.float64) * 0.01          # 100 MHz -> us
    t0 = s[:, 0].min()
    print(["", "", "edge_fwd_h", "edge_bwd_h"][k], f"workgroups {nwg}: start spread {s[:,0].max()-t0:.2f} us, end: mean {(s[:,-1]-t0).mean():.2f} max {(s[:,-1]-t0).max():.2f} us")
    d = np.diff(s, axis=1)
    for i, n in enumerate(names[k][1:]):
        print(f"   {n:14s} mean {d[:, i].mean():6.2f}  p50 {np.median(d[:, i]):6.2f}  max {d[:, i].max():6.2f} us")
s = buf[2, :min(E_tiles, 512)].astype(np.float64) * 0.01
print("edge_fwd_h, layers 1..6: acquire wait | hook issue | MFMAs (us, mean over workgroups)")
for li, (h0, h1) in enumerate([(10, 11), (12, 13), (14, 15), (16, 17), (18, 19), (20, 21)], start=1):
    prev, nxt = s[:, 1 + li], s[:, 2 + li]
    print(f"   L{li}: {np.mean(s[:, h0] - prev):5.2f} | {np.mean(s[:, h1] - s[:, h0]):5.2f} | {np.mean(nxt - s[:, h1]):5.2f}")
