"""CPU: the oracle (oracle/ref_torch.py) against the golden vectors produced by executing the
reference's own sources (oracle/make_golden.py).  Pins the oracle."""
import pytest
import torch

from conftest import load_golden, data_from
from oracle import ref_encoders, ref_torch
from oracle.seeded import seeded_fill_, grad_digest
from batch3dmot_amd import encoders


def _loss_weights(t, salt):
    g = torch.Generator().manual_seed(1234 + salt)
    return torch.randn(t.shape, generator=g)


@pytest.mark.parametrize("name", ["g1_pose.pt", "g1b_pose_batch2.pt", "g5_pose_tiny.pt"])
def test_pose_oracle_matches_reference(name):
    g = load_golden(name)
    data = data_from(g["data"])
    m = ref_torch.PoseGNN()
    m.load_state_dict(g["state_dict"], strict=True)
    m2 = ref_torch.PoseGNN()
    seeded_fill_(m2, g["salt"])
    for k, v in m.state_dict().items():      # the seeded formula reproduces the stored weights
        assert torch.equal(v, m2.state_dict()[k]), k
    cap = []
    out, x_enc = m(data, capture=cap)
    assert torch.equal(out, g["out"]) and torch.equal(x_enc, g["x_enc"])
    assert len(cap) == 6 == len(g["layers"])
    for (x, e), (gx, ge) in zip(cap, g["layers"]):
        assert torch.equal(x, gx) and torch.equal(e, ge)
    loss = (out * _loss_weights(out, 0)).sum() + (x_enc * _loss_weights(x_enc, 1)).sum()
    loss.backward()
    for n, p in m.named_parameters():
        gg = g["grads"][n]
        if gg is None:
            assert p.grad is None and n.startswith("knn_conv")
        else:
            torch.testing.assert_close(p.grad, gg, rtol=1e-5, atol=2e-6 * float(gg.abs().max()))


def test_pose_dead_knn_has_no_effect():
    g = load_golden("g1_pose.pt")
    data = data_from(g["data"])
    m = ref_torch.PoseGNN(run_dead_knn=False)
    m.load_state_dict(g["state_dict"])
    out, _ = m(data)
    assert torch.equal(out, g["out"])


def _clr(salt):
    m = ref_torch.GNN(ref_encoders.ResNetAE(), ref_encoders.PointNetClassifier(k=7), ref_encoders.RadarNetClassifier(k=7))
    seeded_fill_(m, salt)
    return m.eval()


def _digest_close(d_have, d_want, rtol=2e-5):
    assert set(d_have) == set(d_want)
    for n, w in d_want.items():
        h = d_have[n]
        if w is None:
            assert h is None, n
            continue
        tol = rtol * max(w["norm"], 1e-6)
        assert abs(h["norm"] - w["norm"]) <= tol, n
        assert abs(h["proj"] - w["proj"]) <= tol * (w["head"].numel() and (torch.tensor(w["shape"]).prod().item() ** 0.5)), n
        torch.testing.assert_close(h["head"], w["head"], rtol=1e-4, atol=tol)


@pytest.mark.parametrize("name", ["g2_clr.pt", "g2b_clr_one_lidar.pt"])
def test_clr_oracle_matches_reference(name):
    g = load_golden(name)
    data = data_from(g["data"])
    m = _clr(g["salt"])
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == g["state_keys"]
    eo = g["encoder_out"]
    n = data.pose_feats.size(0)
    with torch.no_grad():
        torch.testing.assert_close(m.resnet.encode(data.img_feats), eo["x_img"], rtol=1e-5, atol=1e-6)
        if eo["has_lidar"].any():
            torch.testing.assert_close(
                m.pointnet.forward_feat(data.lidar_feats[eo["has_lidar"]].view(-1, 3, 128)),
                eo["pointnet_out"][eo["has_lidar"]], rtol=1e-5, atol=1e-6)
    cap = []
    out, x_sens = m(data, capture=cap)
    torch.testing.assert_close(out, g["out"], rtol=0, atol=1e-6)
    torch.testing.assert_close(x_sens, g["x_sens"], rtol=1e-5, atol=1e-6)
    for (x, e), (gx, ge) in zip(cap[1:], g["layers"]):
        torch.testing.assert_close(x, gx, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(e, ge, rtol=1e-5, atol=1e-6)
    loss = (out * _loss_weights(out, 0)).sum() + (x_sens * _loss_weights(x_sens, 1)).sum() * 0.1
    loss.backward()
    grads = {n_: p.grad for n_, p in m.named_parameters() if p.requires_grad}
    assert all(v is None for k, v in grads.items() if k.startswith("knn_conv"))
    _digest_close(grad_digest(grads), g["grad_digest"])


def test_train_step_oracle_matches_reference():
    g = load_golden("g3_train_step.pt")
    data = data_from(g["data"])
    m = _clr(g["salt"])
    opt = torch.optim.Adam(m.parameters(), lr=1e-4, weight_decay=1e-4, betas=(0.9, 0.999))
    loss, out, _ = ref_torch.train_step(m, data, opt, batch_size=2, loss_kind="cb")
    torch.testing.assert_close(out, g["out"], rtol=0, atol=1e-6)
    torch.testing.assert_close(loss, g["loss"], rtol=1e-5, atol=0)
    if "grads" in g:      # round 6: the fixture holds every gradient tensor of the reference's step in full
        from conftest import assert_grads_entrywise
        assert_grads_entrywise({n: p.grad for n, p in m.named_parameters() if p.requires_grad},
                               {n: w for n, w in g["grads"].items() if w is not None}, tol=1e-5)
    after = {n: p.detach() for n, p in m.named_parameters() if p.requires_grad}
    _digest_close(grad_digest(after), g["after_digest"], rtol=1e-6)


@pytest.mark.parametrize("name", ["g9_train_mode_step.pt", "g9b_train_mode_one_radar_row.pt"])
def test_train_mode_step_oracle_matches_reference(name):
    """The step as train.py runs it -- model in .train(): batch-statistics BatchNorm in the frozen encoders, running
    statistics updated, the < 2 rows switch to eval (g9b) -- from the reference itself, Dropout neutralised."""
    g = load_golden(name)
    data = data_from(g["data"])
    m = _clr(g["salt"])
    m.train()
    m.pointnet.dropout.p = 0.0
    m.radarnet.dropout.p = 0.0
    opt = torch.optim.Adam(m.parameters(), lr=1e-4, weight_decay=1e-4, betas=(0.9, 0.999))
    loss, out, x_sens = ref_torch.train_step(m, data, opt, batch_size=2, loss_kind="cb")
    torch.testing.assert_close(out, g["out"], rtol=0, atol=2e-6)
    torch.testing.assert_close(x_sens, g["x_sens"], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(loss, g["loss"], rtol=1e-5, atol=0)
    assert {"pointnet": m.pointnet.training, "radarnet": m.radarnet.training, "resnet": m.resnet.training,
            "fc_lidar_encoder": m.fc_lidar_encoder.training, "fc_radar_encoder": m.fc_radar_encoder.training} == g["modes"]
    bufs = dict(m.named_buffers())
    for n, v in g["running_stats"].items():
        torch.testing.assert_close(bufs[n].double(), v.double(), rtol=1e-5, atol=1e-6)
    if "grads" in g:      # round 6: the fixture holds every gradient tensor of the reference's step in full
        from conftest import assert_grads_entrywise
        assert_grads_entrywise({n: p.grad for n, p in m.named_parameters() if p.requires_grad},
                               {n: w for n, w in g["grads"].items() if w is not None}, tol=1e-5)
    after = {n: p.detach() for n, p in m.named_parameters() if p.requires_grad}
    _digest_close(grad_digest(after), g["after_digest"], rtol=1e-6)


def test_train_mode_step_with_live_dropout_oracle_matches_reference():
    """g11: the same step with Dropout LIVE (p = 0.3, pointnet.py:190 / radarnet.py:62, as train.py runs it).  The fixture holds
    the masks the reference's run drew (oracle/make_golden.py: DropoutTape); the restatement is fed the same masks."""
    g = load_golden("g11_train_mode_dropout_live.pt")
    data = data_from(g["data"])
    masks = [m_.clone() for m_ in g["dropout_masks"]]
    assert len(masks) == 2 and masks[0].shape == (g["lidar_rows"], 256) and masks[1].shape == (g["radar_rows"], 256)
    keep = 1.0 / (1.0 - g["dropout_p"])
    for m_ in masks:
        assert bool(((m_ == 0) | ((m_ - keep).abs() < 1e-6)).all()) and 0.2 < float((m_ == 0).float().mean()) < 0.4
    m = _clr(g["salt"])
    m.train()
    import torch.nn.functional as F
    real = F.dropout

    def replay(input, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return input
        mk = masks.pop(0)
        assert mk.shape == input.shape
        return input * mk
    F.dropout = replay
    try:
        opt = torch.optim.Adam(m.parameters(), lr=1e-4, weight_decay=1e-4, betas=(0.9, 0.999))
        loss, out, x_sens = ref_torch.train_step(m, data, opt, batch_size=2, loss_kind="cb")
    finally:
        F.dropout = real
    assert not masks                                             # both Dropout layers ran
    torch.testing.assert_close(out, g["out"], rtol=0, atol=2e-6)
    torch.testing.assert_close(x_sens, g["x_sens"], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(loss, g["loss"], rtol=1e-5, atol=0)
    bufs = dict(m.named_buffers())
    for n, v in g["running_stats"].items():
        torch.testing.assert_close(bufs[n].double(), v.double(), rtol=1e-5, atol=1e-6)
    if "grads" in g:      # round 6: the fixture holds every gradient tensor of the reference's step in full
        from conftest import assert_grads_entrywise
        assert_grads_entrywise({n: p.grad for n, p in m.named_parameters() if p.requires_grad},
                               {n: w for n, w in g["grads"].items() if w is not None}, tol=1e-5)
    after = {n: p.detach() for n, p in m.named_parameters() if p.requires_grad}
    _digest_close(grad_digest(after), g["after_digest"], rtol=1e-6)


def test_predict_post_oracle_matches_reference():
    g = load_golden("g4_predict_post.pt")
    pairs, present, scores = g["pairs"], g["present"], g["scores"]
    cnt = present.sum(0)
    avg = ((scores.double() * present).sum(0) / cnt)
    thr = torch.tensor([g["thresholds"][g["class_names"][int(c)]] for c in g["node_cls"][pairs[:, 0]]],
                       dtype=torch.float64)
    kept, pred, succ = ref_torch.greedy_flux(g["node_cls"].numel(), pairs.tolist(), avg.tolist(), thr.tolist())
    assert torch.equal(pairs[kept], g["kept_pairs"])
    assert pred == g["pred"].tolist() and succ == g["succ"].tolist()


def test_post_adam_tolerance_accepts_the_golden_and_rejects_a_missing_or_reversed_update():
    """conftest.assert_adam_heads_close (the check the GPU training-step tests apply to the weights after Adam): the golden
    against itself passes; weights that never moved, and weights that moved the wrong way, fail."""
    from conftest import assert_adam_heads_close
    from batch3dmot_amd import encoders
    from batch3dmot_amd.clr_att_gnn import GNN
    g = load_golden("g9_train_mode_step.pt")
    m = GNN(encoders.ResNetAE(), encoders.PointNetClassifier(k=7), encoders.RadarNetClassifier(k=7))
    seeded_fill_(m, g["salt"])
    before = {n: p.detach().reshape(-1)[:8].double().clone() for n, p in m.named_parameters() if p.requires_grad}
    gold = g["after_digest"]
    assert_adam_heads_close(before, gold, gold, lr=1e-4)
    stuck = {n: {"head": before[n]} for n in gold}
    reversed_ = {n: {"head": 2 * before[n] - gold[n]["head"]} for n in gold}
    for bad in (stuck, reversed_):
        with pytest.raises(AssertionError):
            assert_adam_heads_close(before, bad, gold, lr=1e-4)
