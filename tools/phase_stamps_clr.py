"""Phase time stamps of the camera+LiDAR+radar node kernels (library built with -DB3D_EXP_STAMPS, e.g. make VARIANT=stamps
EXTRA=-DB3D_EXP_STAMPS; B3D_LIB=stamps python tools/phase_stamps_clr.py): where a 16-row tile of mp_node_fwd_split_h /
node_bwd_g<MLP> spends its time, and how much of it wavefront 0 sits in the weight ring's acquire (counted wait + barrier)."""
import argparse, ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from batch3dmot_amd import _lib

dev = torch.device("cuda:0")
args = argparse.Namespace(no_dead_knn=False, encode_ahead=False, force_collective=False)
w = bench.Workload("clr", dev, 0, 1, args, encoders="precomputed", graphs=2)
for i in range(6):
    w.step(i)
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros((4, 512, 32), dtype=np.int64)
lib.b3d_debug_stamps_clr.argtypes = [C.c_void_p]
assert lib.b3d_debug_stamps_clr(buf.ctypes.data_as(C.c_void_p)) == 0
names = {0: ["start", "segment sums", "L0 (M -> H1)", "L1 (H1 -> H2)", "L2 (H2 -> x')", "table of the next layer"],
         1: ["start", "loads", "product 0", "products 1-3", "L4 (dx' -> dH2)", "L5 (dH2 -> dH1)", "L6 (dH1 -> dM)"]}
N = 3000
nwg = (N + 15) // 16
for k in (0, 1):
    s = buf[k, :nwg, :len(names[k])].astype(np.float64) * 0.01          # 100 MHz -> us
    acq = buf[k, :nwg, 31].astype(np.float64) * 0.01
    t0 = s[:, 0].min()
    tot = s[:, -1] - s[:, 0]
    print(["mp_node_fwd_split_h", "node_bwd_g<MLP>"][k], f"workgroups {nwg}: start spread {s[:,0].max()-t0:.2f} us; a workgroup lasts mean {tot.mean():.2f} "
          f"max {tot.max():.2f} us; launch end (last workgroup - first start) {(s[:,-1]-t0).max():.2f} us")
    d = np.diff(s, axis=1)
    for i, n in enumerate(names[k][1:]):
        print(f"   {n:26s} mean {d[:, i].mean():6.2f}  p50 {np.median(d[:, i]):6.2f}  max {d[:, i].max():6.2f} us")
    print(f"   inside the ring's acquire (wave 0, all chunks): mean {acq.mean():6.2f}  p50 {np.median(acq):6.2f}  max {acq.max():6.2f} us = {100*acq.mean()/tot.mean():.0f} % of the workgroup")
