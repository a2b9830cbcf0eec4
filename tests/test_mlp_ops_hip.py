"""The standalone MLP-stack and one-key attention operators of the C ABI (`b3d_mlp_forward / _backward`,
`b3d_xattn_node_affine_*`) against plain PyTorch modules evaluated in float64 on the CPU: every stack the
reference declares (pose_gnn.py:29-53, clr_att_gnn.py:35-91), outputs, input gradients and every parameter gradient entry by
entry; ragged row counts, one row, no rows; bitwise repeatability (fixed-order weight-gradient sums)."""
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu
TOL = 1e-4          # north_star's bar; the operator is an exact-fp32 fmaf chain, measured ~1e-6


def rel(a, b):
    b = b.detach().double().cpu()
    return float((a.detach().double().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-30))


def stack(widths, sigmoid=False, relu_last=False):
    mods = []
    for i in range(len(widths) - 1):
        mods.append(nn.Linear(widths[i], widths[i + 1]))
        if i + 2 < len(widths) or relu_last:
            mods.append(nn.ReLU())
    if sigmoid:
        mods.append(nn.Sigmoid())
    return nn.Sequential(*mods)


STACKS = [
    ([4, 8, 16, 32], False, 3001),          # PoseGNN edge_encoder
    ([19, 24, 36, 48], False, 777),         # PoseGNN node_encoder (19-float rows: scalar loads)
    ([32, 16, 8, 4, 1], False, 2050),       # PoseGNN edge_classifier (logits)
    ([4, 16, 32, 64], False, 1000),         # GNN edge_encoder
    ([19, 48, 96], False, 300),             # GNN node_encoder
    ([64, 32, 16, 8, 1], True, 5000),       # GNN edge_classifier + Sigmoid
    ([256, 192, 128], False, 211),          # fc_lidar_encoder
    ([256, 192, 128, 64], False, 75),       # fc_radar_encoder
    ([640, 512, 384, 256, 128, 64], False, 1500),   # att_edge_encoder
    ([48, 40], False, 65),                  # one layer
    ([7, 5, 3], False, 1),                  # one row, odd widths
]


@pytest.mark.parametrize("widths,sigmoid,rows", STACKS)
def test_mlp_operator_matches_float64_modules(widths, sigmoid, rows):
    from batch3dmot_amd import _lib
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(sum(widths) + rows)
    ref = stack(widths, sigmoid)
    with torch.no_grad():
        for p in ref.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (1.5 / max(p.shape[-1], 1) ** 0.5))
    x = torch.randn(rows, widths[0], generator=g)
    w = torch.randn(rows, widths[-1], generator=g)
    import copy
    m = copy.deepcopy(ref).to(dev)
    xd = x.to(dev).requires_grad_(True)
    y = _lib.mlp(m, xd)
    (y * w.to(dev)).sum().backward()
    r64 = copy.deepcopy(ref).double()
    x64 = x.double().requires_grad_(True)
    y64 = r64(x64)
    (y64 * w.double()).sum().backward()
    assert y.shape == y64.shape
    assert rel(y, y64) < TOL, rel(y, y64)
    assert rel(xd.grad, x64.grad) < TOL, rel(xd.grad, x64.grad)
    for (n, p), (_, q) in zip(m.named_parameters(), r64.named_parameters()):
        assert p.grad is not None and rel(p.grad, q.grad) < TOL, (n, rel(p.grad, q.grad))
    # the same call again: bit-equal (no float atomics, fixed chunk order)
    g1 = [p.grad.clone() for p in m.parameters()]
    for p in m.parameters():
        p.grad = None
    xd2 = x.to(dev).requires_grad_(True)
    y2 = _lib.mlp(m, xd2)
    (y2 * w.to(dev)).sum().backward()
    assert torch.equal(y, y2) and torch.equal(xd.grad, xd2.grad)
    for a, p in zip(g1, m.parameters()):
        assert torch.equal(a, p.grad)


def test_mlp_operator_inference_and_empty_batch():
    from batch3dmot_amd import _lib
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    m = stack([64, 32, 16, 8, 1], sigmoid=True).to(dev)
    x = torch.randn(999, 64, device=dev)
    with torch.no_grad():
        y = _lib.mlp(m, x)                                   # ping-pong workspace, nothing saved
        want = m.double()(x.double())
    m.float()
    assert rel(y, want) < TOL
    # no rows: empty output, zero gradients
    x0 = torch.zeros(0, 64, device=dev, requires_grad=True)
    y0 = _lib.mlp(m, x0)
    assert y0.shape == (0, 1)
    y0.sum().backward()
    for p in m.parameters():
        assert p.grad is not None and float(p.grad.abs().max()) == 0.0
    with pytest.raises(ValueError):
        _lib.mlp(m, torch.randn(5, 63, device=dev))
    with pytest.raises(ValueError, match="GPU"):
        _lib.mlp(m, torch.randn(5, 64))


@pytest.mark.parametrize("d,rows", [(96, 3000), (128, 517), (64, 9)])
def test_xattn_node_affine_equals_multihead_attention_with_one_key(d, rows):
    """nn.MultiheadAttention called as clr_att_gnn.py:144-155 calls it (one query, one key per edge) against the per-node operator:
    same output for ANY query / key, zero gradients in the query / key thirds of in_proj."""
    from batch3dmot_amd import _lib
    dev = torch.device("cuda:0")
    torch.manual_seed(d + rows)
    att = nn.MultiheadAttention(embed_dim=d, num_heads=2, kdim=d, vdim=d, batch_first=True)
    with torch.no_grad():
        att.in_proj_bias.copy_(torch.randn(3 * d) * 0.1)
        att.out_proj.bias.copy_(torch.randn(d) * 0.1)
    v = torch.randn(rows, d)
    q = torch.randn(rows, d)
    w = torch.randn(rows, d)
    import copy
    a64 = copy.deepcopy(att).double()
    v64 = v.double().requires_grad_(True)
    y64, _ = a64(q.double().unsqueeze(1), v64.unsqueeze(1), v64.unsqueeze(1))
    (y64.squeeze(1) * w.double()).sum().backward()
    ad = copy.deepcopy(att).to(dev)
    vd = v.to(dev).requires_grad_(True)
    y = _lib.xattn_node_affine(ad, vd)
    (y * w.to(dev)).sum().backward()
    assert rel(y, y64.squeeze(1)) < TOL
    # the reference's value gradient also flows through the key projection of the SAME tensor -- which is dead (softmax over one
    # key), so d value is the v_proj path alone
    assert rel(vd.grad, v64.grad) < TOL
    assert rel(ad.out_proj.weight.grad, a64.out_proj.weight.grad) < TOL
    assert rel(ad.out_proj.bias.grad, a64.out_proj.bias.grad) < TOL
    assert rel(ad.in_proj_weight.grad[2 * d:], a64.in_proj_weight.grad[2 * d:]) < TOL
    assert rel(ad.in_proj_bias.grad[2 * d:], a64.in_proj_bias.grad[2 * d:]) < TOL
    assert float(ad.in_proj_weight.grad[:2 * d].abs().max()) == 0.0 and float(ad.in_proj_bias.grad[:2 * d].abs().max()) == 0.0
    assert float(a64.in_proj_weight.grad[:2 * d].abs().max()) < 1e-12
