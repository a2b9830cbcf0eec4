"""Diagnostic (not a test): per-parameter gradient error of the HIP camera+LiDAR+radar path and of the fp32 CPU oracle
against a float64 evaluation, at the benchmark size, for a few seeds.  Usage: python tests/tools/clr_grad_error_seeds.py [seeds]"""
import copy, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from batch3dmot_amd import encoders, synth
from batch3dmot_amd.clr_att_gnn import GNN
from batch3dmot_amd.data import Data
from oracle import ref_torch
from oracle.seeded import seeded_fill_

def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
def l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))

dev = torch.device("cuda:0")
nodes = int(os.environ.get("NODES", "1500"))
for seed in [int(s) for s in sys.argv[1:]] or [23]:
    ora = ref_torch.GNN(encoders.ResNetAE(), encoders.PointNetClassifier(k=7), encoders.RadarNetClassifier(k=7), run_dead_knn=False, loop_masks=False)
    seeded_fill_(ora, seed); ora.eval()
    m = GNN(encoders.ResNetAE(), encoders.PointNetClassifier(k=7), encoders.RadarNetClassifier(k=7)).to(dev).eval()
    m.load_state_dict(ora.state_dict())
    big = synth.make_batch(2, nodes, 10 * nodes, first_graph_idx=40 + seed, modalities=True)
    g = torch.Generator().manual_seed(seed)
    ro, rs = ora(big)
    c0, c1 = torch.randn(ro.shape, generator=g), torch.randn(rs.shape, generator=g)
    ((ro * c0).sum() + 0.1 * (rs * c1).sum()).backward()
    go, gs = m(big.to(dev))
    ((go * c0.to(dev)).sum() + 0.1 * (gs * c1.to(dev)).sum()).backward()
    o64 = copy.deepcopy(ora).double(); o64.zero_grad()
    big64 = Data(**{k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in big.__dict__.items()})
    do, ds = o64(big64)
    ((do * c0.double()).sum() + 0.1 * (ds * c1.double()).sum()).backward()
    print(f"seed {seed}: out hip {rel(go, do):.2e} cpu {rel(ro, do):.2e}")
    for (name, p), (_, q), (_, r) in zip(m.named_parameters(), ora.named_parameters(), o64.named_parameters()):
        if not q.requires_grad or r.grad is None or name.startswith("knn_conv") or float(r.grad.abs().max()) == 0: continue
        pg, qg, rg = p.grad, q.grad, r.grad
        if "in_proj" in name:
            t = rg.shape[0] // 3; pg, qg, rg = pg[2*t:], qg[2*t:], rg[2*t:]
        print(f"  {name:45s} l2 hip {l2(pg, rg):.2e} cpu {l2(qg, rg):.2e} | max hip {rel(pg, rg):.2e} cpu {rel(qg, rg):.2e}")
