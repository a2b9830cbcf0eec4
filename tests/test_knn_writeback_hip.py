"""`knn_writeback=True` (non-reference: the frame-wise k-NN + GAT block's result is USED) and the block's
backward (`b3d_knn_gat_backward`) against the oracle's autograd.  The neighbour lists carry no gradient; the oracle is handed the
lists the HIP forward chose, so a near-tie between two distances cannot make the two sides select different graphs."""
import pytest
import torch

from oracle import ref_encoders, ref_torch
from oracle.seeded import seeded_fill_

pytestmark = pytest.mark.gpu
TOL = 1e-4


def rel(a, b):
    b = b.detach().double().cpu()
    return float((a.detach().double().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-30))


class no_torch_linear:
    """Inside: ``torch.nn.functional.linear`` (every ``nn.Linear`` / ``nn.MultiheadAttention`` forward goes through it) raises --
    the product's writeback path must run its dense stacks through the library's operators (b3d_mlp_*, b3d_xattn_node_affine_*)."""

    def __enter__(self):
        import torch.nn.functional as F
        self.F, self.keep = F, F.linear

        def banned(*a, **k):
            raise AssertionError("torch.nn.functional.linear called inside the HIP product path")
        F.linear = banned
        return self

    def __exit__(self, *exc):
        self.F.linear = self.keep
        return False


@pytest.mark.parametrize("d,frames,per_frame", [(48, 5, 120), (96, 4, 300), (48, 3, 7)])
def test_knn_gat_block_backward_matches_oracle_autograd(d, frames, per_frame):
    from batch3dmot_amd import _lib
    from batch3dmot_amd.pose_gnn import GATConvParams
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(100 + d + per_frame)
    n = frames * per_frame
    x = torch.randn(n, d, generator=g)
    ts = torch.arange(frames).repeat_interleave(per_frame)
    conv = GATConvParams(d)
    with torch.no_grad():
        conv.bias.copy_(0.1 * torch.randn(d, generator=g))
    ora = ref_torch.GATConv(d)
    ora.load_state_dict(conv.state_dict())
    conv = conv.to(dev)
    xd = x.to(dev).requires_grad_(True)
    y, nbr, cnt = _lib.knn_gat_conv(xd, ts.to(dev), conv, 20, return_graph=True)
    w = torch.randn(n, d, generator=g)
    (y * w.to(dev)).sum().backward()
    assert int(cnt.max()) == min(20, per_frame - 1)
    xo = x.clone().requires_grad_(True)
    yo = ref_torch._knn_block_writeback(xo, ts, ora, graph=(nbr.cpu(), cnt.cpu()))
    (yo * w).sum().backward()
    assert rel(y, yo) < TOL
    assert rel(xd.grad, xo.grad) < TOL
    for name in ("lin_src.weight", "att_src", "att_dst", "bias"):
        a = dict(conv.named_parameters())[name].grad
        b = dict(ora.named_parameters())[name].grad
        assert a.shape == b.shape and rel(a, b) < TOL, name
    # fixed summation order: a second backward gives the same bits
    xd2 = x.to(dev).requires_grad_(True)
    conv.zero_grad()
    y2 = _lib.knn_gat_conv(xd2, ts.to(dev), conv, 20)
    (y2 * w.to(dev)).sum().backward()
    assert torch.equal(xd2.grad, xd.grad)


def test_pose_gnn_with_knn_writeback_matches_the_oracle():
    from batch3dmot_amd import synth
    from batch3dmot_amd.pose_gnn import PoseGNN
    dev = torch.device("cuda:0")
    data = synth.make_graph(400, None, k=7, graph_idx=31)
    ora = ref_torch.PoseGNN(knn_writeback=True)
    seeded_fill_(ora, 77)
    m = PoseGNN().to(dev)
    m.load_state_dict(ora.state_dict(), strict=True)
    m.knn_writeback = True
    w = torch.randn((data.edge_index.size(1), 1), generator=torch.Generator().manual_seed(5))
    with no_torch_linear():
        out, x_enc = m(data.to(dev))
        (out * w.to(dev)).sum().backward()
    assert len(m._last_knn) == 3                                   # layers 0, 2, 4
    ora.knn_graphs = [(nbr.cpu(), cnt.cpu()) for nbr, cnt in m._last_knn]
    o_ref, x_ref = ora(data)
    (o_ref * w).sum().backward()
    assert rel(out, o_ref) < TOL and rel(x_enc, x_ref) < TOL
    plain = ref_torch.PoseGNN(run_dead_knn=False)
    plain.load_state_dict(ora.state_dict())
    assert rel(plain(data)[0], o_ref) > 1e-3                        # the written-back block does change the scores
    got, want = dict(m.named_parameters()), dict(ora.named_parameters())
    for name, q in want.items():
        assert q.grad is not None, name                            # knn_conv trains in this mode
        assert rel(got[name].grad, q.grad) < 2e-4, (name, rel(got[name].grad, q.grad))


def test_clr_gnn_with_knn_writeback_matches_the_oracle():
    """Whole camera+LiDAR+radar model with natural weights: two correct fp32 evaluations differ at the 1e-3 .. 2e-2 level in EVERY
    gradient when a single ReLU unit sits within rounding of zero and takes the other branch on one side (DESIGN.md section 2;
    which seed has such a unit changes with the summation order of the kernels).  Three (weights, graph) pairs are evaluated: the
    outputs must agree to 1e-4 on all of them, and every parameter's gradient to 2e-4 on at least one of them -- a wrong backward
    fails all three; the block and the layer are held entry by entry in the operator tests."""
    from batch3dmot_amd import encoders, synth
    from batch3dmot_amd.clr_att_gnn import GNN
    dev = torch.device("cuda:0")
    best = {}
    clean_runs = 0
    for salt, gi in ((79, 32), (78, 33), (80, 34)):
        data = synth.make_graph(300, None, k=6, graph_idx=gi, modalities=True)
        ora = ref_torch.GNN(ref_encoders.ResNetAE(), ref_encoders.PointNetClassifier(k=7), ref_encoders.RadarNetClassifier(k=7), loop_masks=False,
                            knn_writeback=True)
        seeded_fill_(ora, salt)
        ora.eval()
        m = GNN(encoders.ResNetAE(), encoders.PointNetClassifier(k=7), encoders.RadarNetClassifier(k=7)).to(dev).eval()
        m.load_state_dict(ora.state_dict(), strict=True)
        m.knn_writeback = True
        w = torch.randn((data.edge_index.size(1), 1), generator=torch.Generator().manual_seed(6))
        with no_torch_linear():
            out, x_sens = m(data.to(dev))
            (out * w.to(dev)).sum().backward()
        ora.knn_graphs = [(nbr.cpu(), cnt.cpu()) for nbr, cnt in m._last_knn]
        o_ref, s_ref = ora(data)
        (o_ref * w).sum().backward()
        assert rel(out, o_ref) < TOL and rel(x_sens, s_ref) < TOL
        got, want = dict(m.named_parameters()), dict(ora.named_parameters())
        worst = 0.0
        for name, q in want.items():
            if q.grad is None:
                continue
            if float(q.grad.abs().max()) == 0.0:                      # q / k thirds of in_proj: exactly zero on both sides
                assert got[name].grad is None or float(got[name].grad.abs().max()) == 0.0, name
                continue
            r = rel(got[name].grad, q.grad)
            best[name] = min(best.get(name, 1.0), r)
            worst = max(worst, r)
        clean_runs += worst < 2e-4
        assert want["knn_conv.att_src"].grad is not None
    assert len(best) > 40
    for name, r in best.items():
        assert r < 2e-4, (name, r)
    assert clean_runs >= 1                                          # at least one pair without any flipped unit
