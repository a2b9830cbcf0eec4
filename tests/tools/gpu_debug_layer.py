import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import ref_torch
from oracle.seeded import seeded_fill_
from batch3dmot_amd import synth
from batch3dmot_amd.pose_gnn import CausalMessagePassing
dev = torch.device("cuda:0")
n, k = int(sys.argv[1]), int(sys.argv[2])
d = synth.make_graph(n, None, k=k, graph_idx=31)
ora = ref_torch.CausalMessagePassing("p"); seeded_fill_(ora, 5)
m = CausalMessagePassing(); m.load_state_dict(ora.state_dict()); m.to(dev)
g = torch.Generator().manual_seed(1)
N, E = d.pose_feats.size(0), d.edge_index.size(1)
x = torch.randn(N, 48, generator=g); x0 = torch.randn(N, 48, generator=g); e = torch.randn(E, 32, generator=g)
cx, ce = torch.randn(N, 48, generator=g), torch.randn(E, 32, generator=g)
def run(mod, dev_):
    xs = [t.clone().to(dev_).requires_grad_(True) for t in (x, x0, e)]
    xn, en = mod(xs[0], d.edge_index.to(dev_), xs[2], xs[1])
    ((xn * cx.to(dev_)).sum() + (en * ce.to(dev_)).sum()).backward()
    return xn.detach().cpu(), en.detach().cpu(), [t.grad.cpu() for t in xs], {k_: p.grad.cpu() for k_, p in mod.named_parameters()}
ref = run(ora, torch.device("cpu")); got = run(m, dev)
print("N", N, "E", E)
src, dst = d.edge_index
indeg = torch.bincount(dst, minlength=N); outdeg = torch.bincount(src, minlength=N)
print("max indeg", indeg.max().item(), "max outdeg", outdeg.max().item())
for a, b, name in zip(got[2], ref[2], ("d x", "d x0", "d e")):
    diff = (a - b).abs()
    rows = torch.nonzero(diff.max(1).values > 1e-4 * b.abs().max()).flatten()
    print(name, "max err", diff.max().item(), "scale", b.abs().max().item(), "bad rows", rows.numel(), rows[:20].tolist())
    if name != "d e" and rows.numel():
        print("   indeg", indeg[rows[:20]].tolist(), "outdeg", outdeg[rows[:20]].tolist())
        r = rows[0].item(); print("   cols", torch.nonzero(diff[r] > 1e-4).flatten().tolist()[:48])
for k_ in ref[3]:
    print(k_, ((got[3][k_] - ref[3][k_]).abs().max() / ref[3][k_].abs().max()).item())
