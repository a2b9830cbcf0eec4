"""GPU diagnostic: GNN (CLR) HIP path vs the golden fixtures / oracle."""
import ctypes as C, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from batch3dmot_amd import _lib, encoders
from batch3dmot_amd.data import Data
from batch3dmot_amd.clr_att_gnn import GNN
from oracle import ref_torch
from oracle.seeded import seeded_fill_, grad_digest


def rel(a, b):
    return ((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


def lw(t, salt):
    g = torch.Generator().manual_seed(1234 + salt); return torch.randn(t.shape, generator=g)


dev = torch.device("cuda:0")
for name in ("g2_clr.pt", "g2b_clr_one_lidar.pt"):
    g = torch.load(os.path.join(ROOT, "tests/golden", name), weights_only=False)
    data = Data(**g["data"])
    m = GNN(encoders.ResNetAE(), encoders.PointNetClassifier(k=7), encoders.RadarNetClassifier(k=7))
    seeded_fill_(m, g["salt"])
    m = m.to(dev).eval()
    m.keep_workspace = True
    ora = ref_torch.GNN(encoders.ResNetAE(), encoders.PointNetClassifier(k=7), encoders.RadarNetClassifier(k=7), loop_masks=False)
    seeded_fill_(ora, g["salt"]); ora.eval()
    d = data.to(dev)
    out, x_sens = m(d)
    torch.cuda.synchronize()
    print(name, "prob rel", rel(out, g["out"]), "x_sens rel", rel(x_sens, g["x_sens"]))
    ws, nbytes, flags, N, E, nl, nr = m._last_workspace
    lib = _lib.load()
    for l in range(m.depth + 1):
        px, pe, pa = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _lib.check(lib.b3d_clr_debug_ptrs(ws.data_ptr(), nbytes, N, E, nl, nr, m.depth, flags, l, C.byref(px), C.byref(pe), C.byref(pa)), "dbg")
        x = ws[px.value - ws.data_ptr():][:N * 96 * 4].view(torch.float32).view(N, 96)
        e = ws[pe.value - ws.data_ptr():][:E * 64 * 4].view(torch.float32).view(E, 64)
        if l >= 1:
            gx, ge = g["layers"][l - 1]
            print(f"  layer {l-1}: x rel {rel(x, gx):.3e} e rel {rel(e, ge):.3e}")
    cap = []
    o2, xs2 = ora(data, capture=cap)
    att = ws[pa.value - ws.data_ptr():][:E * 64 * 4].view(torch.float32).view(E, 64)
    print("  att_edge_attr rel", rel(att, cap[0][1]))
    loss = (out * lw(out, 0).to(dev)).sum() + (x_sens * lw(x_sens, 1).to(dev)).sum() * 0.1
    loss.backward()
    torch.cuda.synchronize()
    l2 = (o2 * lw(o2, 0)).sum() + (xs2 * lw(xs2, 1)).sum() * 0.1
    l2.backward()
    worst = 0
    for (n, p), (_, q) in zip(m.named_parameters(), ora.named_parameters()):
        if not q.requires_grad:
            continue
        if q.grad is None:
            print("   ", n, "ref None ours", None if p.grad is None else "tensor"); continue
        if n.endswith("in_proj_weight") or n.endswith("in_proj_bias"):
            dd = q.grad.shape[0] // 3
            r = rel(p.grad[2 * dd:], q.grad[2 * dd:]); qk = float(p.grad[:2 * dd].abs().max()); rq = float(q.grad[:2 * dd].abs().max())
            print(f"    {n:45s} v-rows rel {r:.3e}; q/k rows ours max {qk:.1e} ref max {rq:.1e}")
        else:
            r = rel(p.grad, q.grad)
            if r > 1e-5: print(f"    {n:45s} rel {r:.3e} |ref|max {q.grad.abs().max():.3e}")
        worst = max(worst, r)
    print("  worst grad rel", worst)
