"""Randomised parity sweep (GPU box; the oracle is the checker): whole-model forward + backward of both models on graphs of random
size and degree -- edge counts that are not multiples of the 64-row tiles, a handful of edges, isolated nodes, missing modalities --
against the CPU oracle, plus bitwise repeatability of the HIP side.  Prints one line per case and the worst errors; exit status 1
on a violation.  ``run_cases`` is what tests/test_fuzz_parity_hip.py runs on a 12-case seeded subset (round 6).

    python tools/fuzz_parity.py [--cases 24] [--seed 0]
"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def rel(x, y):
    x, y = x.detach().double().cpu(), y.detach().double().cpu()
    return float((x - y).abs().max() / y.abs().max().clamp_min(1e-30))


def l2(x, y):
    x, y = x.detach().double().cpu(), y.detach().double().cpu()
    return float((x - y).norm() / y.norm().clamp_min(1e-30))


def grads(model, data, w):
    model.zero_grad(set_to_none=True)
    out, xs = model(data)
    (out * w.to(out.device)).sum().backward()
    return out.detach(), xs.detach(), {n: (p.grad.detach().clone() if p.grad is not None else None) for n, p in model.named_parameters()}


def run_cases(cases=24, seed=0, max_nodes=700, out=print):
    """Returns (failures, worst output error, worst gradient L2 error)."""
    from oracle import ref_encoders, ref_torch            # checker only
    from oracle.seeded import seeded_fill_
    from batch3dmot_amd import encoders, synth
    from batch3dmot_amd.clr_att_gnn import GNN
    from batch3dmot_amd.pose_gnn import PoseGNN
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(seed)
    bad = 0
    worst_out = worst_l2 = 0.0
    for case in range(cases):
        kind = "clr" if case % 2 == 0 else "pose"
        n = int(torch.randint(12, max_nodes, (1,), generator=gen))
        k = int(torch.randint(2, 14, (1,), generator=gen))
        data = synth.make_graph(n, None, k=k, graph_idx=7000 + case, modalities=(kind == "clr"))
        if kind == "clr" and case % 6 == 2:
            data.radar_feats.zero_()                      # camera + LiDAR only
        if kind == "clr" and case % 6 == 4:
            data.lidar_feats.zero_(); data.radar_feats.zero_()
        salt = 300 + case
        if kind == "clr":
            ora = ref_torch.GNN(ref_encoders.ResNetAE(), ref_encoders.PointNetClassifier(k=7), ref_encoders.RadarNetClassifier(k=7),
                                run_dead_knn=False, loop_masks=False)
            seeded_fill_(ora, salt)
            ora.eval()
            m = GNN(encoders.ResNetAE(), encoders.PointNetClassifier(k=7), encoders.RadarNetClassifier(k=7))
            m.load_state_dict(ora.state_dict())
            m = m.to(dev).eval()
        else:
            ora = ref_torch.PoseGNN(run_dead_knn=False)
            seeded_fill_(ora, salt)
            m = PoseGNN().to(dev)
            m.load_state_dict(ora.state_dict())
        E = data.edge_index.size(1)
        with torch.no_grad():
            shape = ora(data)[0].shape
        w = torch.randn(shape, generator=gen)
        o_ref, x_ref, g_ref = grads(ora, data, w)
        dd = data.to(dev)
        o1, x1, g1 = grads(m, dd, w)
        o2, x2, g2 = grads(m, dd, w)
        torch.cuda.synchronize()
        same = torch.equal(o1, o2) and all((g1[n] is None and g2[n] is None) or torch.equal(g1[n], g2[n]) for n in g1)
        e_out, e_x = rel(o1, o_ref), rel(x1, x_ref)
        gl2 = 0.0
        gname = ""
        for nme, gr in g_ref.items():
            if gr is None:
                if g1[nme] is not None and float(g1[nme].abs().max()) != 0.0:
                    bad += 1; out("  unexpected gradient " + nme)
                continue
            if nme.endswith("in_proj_weight") or nme.endswith("in_proj_bias"):
                kk = 2 * gr.shape[0] // 3
                v = l2(g1[nme][kk:], gr[kk:])
            else:
                v = l2(g1[nme], gr)
            if v > gl2:
                gl2, gname = v, nme
        ok = same and e_out < 1e-4 and e_x < 1e-4 and gl2 < 2e-3           # L2 over a tensor: a ReLU flip on a small graph moves single rows
        note = ""
        if same and e_out < 1e-4 and e_x < 1e-4 and not ok:
            # a gradient beyond the bound: is it a ReLU pre-activation within rounding of zero?  The float64 oracle decides -- if the fp32
            # ORACLE is as far from it as the kernels are from the fp32 oracle, the unit sits on the fence and both answers are fp32-correct
            import copy
            o64 = copy.deepcopy(ora).double()
            o64.zero_grad(set_to_none=True)
            pre_min = [float("inf")]
            hooks = []
            mods = list(o64.modules())
            for seq in mods:
                if isinstance(seq, torch.nn.Sequential):
                    ch = list(seq)
                    for a_, b_ in zip(ch, ch[1:]):
                        if isinstance(a_, torch.nn.Linear) and isinstance(b_, torch.nn.ReLU):
                            hooks.append(a_.register_forward_hook(
                                lambda _m, _i, out_: pre_min.__setitem__(0, min(pre_min[0], float(out_.detach().abs().min())) if out_.numel() else pre_min[0])))
            if kind == "clr":
                d64 = copy.copy(data)
                for f in ("pose_feats", "edge_attr", "img_feats", "lidar_feats", "radar_feats"):
                    setattr(d64, f, getattr(data, f).double())
                out64 = o64(d64)[0]
            else:                                                     # (ref_torch.PoseGNN.forward casts edge_attr with .float(): by hand)
                e64 = o64.edge_encoder(data.edge_attr.float().double())
                x64 = o64.node_encoder(data.pose_feats.double())
                x0 = x64
                for _ in range(6):
                    x64, e64 = o64.message_passing(x64, data.edge_index, e64, x0)
                out64 = o64.edge_classifier(e64)
            (out64 * w.double()).sum().backward()
            g64 = {n_: (p_.grad.detach().clone() if p_.grad is not None else None) for n_, p_ in o64.named_parameters()}
            own = l2(g_ref[gname], g64[gname])
            mine = l2(g1[gname].cpu().double(), g64[gname])
            for h_ in hooks:
                h_.remove()
            # the smallest |ReLU pre-activation| of the float64 run: below ~1e-6 a correct fp32 evaluation may land on either side of zero
            note = (f"  [float64 oracle: fp32 oracle off by {own:.1e}, kernels off by {mine:.1e}; smallest |ReLU pre-activation| "
                    f"{pre_min[0]:.1e}]")
            ok = mine < 3.0 * max(own, 1e-6) or mine < 2e-3 or (pre_min[0] < 2e-6 and mine < 2e-2)
        bad += 0 if ok else 1
        worst_out, worst_l2 = max(worst_out, e_out, e_x), max(worst_l2, gl2)
        out(f"{kind:4s} N {n:4d} E {E:5d} (E % 64 = {E % 64:2d})  out {e_out:.1e}  x {e_x:.1e}  worst grad L2 {gl2:.1e} ({gname})  "
              f"repeatable {same}  {'ok' if ok else 'FAIL'}{note}")
    return bad, worst_out, worst_l2


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=24)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    bad, worst_out, worst_l2 = run_cases(a.cases, a.seed, out=lambda s_: print(s_, flush=True))
    print(f"cases {a.cases}  failures {bad}  worst output error {worst_out:.2e}  worst gradient L2 error {worst_l2:.2e}")
    sys.exit(1 if bad else 0)
