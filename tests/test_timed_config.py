"""GPU: the configuration bench.py TIMES, checked as a whole.

* the hipGraph-captured camera+LiDAR+radar training step -- train-mode frozen encoders, ``mask_stream``, ``pre()`` +
  ``rows_static`` feeding the replay, a non-null launch stream: bench.Workload / bench.capture / bench.run_step themselves --
  replayed over the 4-batch pool must leave bitwise the same parameters, BatchNorm statistics and Adam state as the same
  number of eager steps from the same state;
* the train-mode training step (encoders in train mode: batch-statistics BatchNorm, running-statistics updates; Dropout
  neutralised on both sides) against the CPU oracle's ``train_step`` at 300 nodes and at the benchmark size.
"""
import argparse

import pytest
import torch

from oracle.seeded import grad_digest, seeded_fill_

pytestmark = pytest.mark.gpu
TOL = 1e-4


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _bench_workload(kind, encoders="frozen"):
    import bench
    dev = torch.device("cuda:0")
    torch.cuda.set_stream(torch.cuda.Stream(dev))            # bench.main(): nothing runs on the legacy NULL stream
    ahead = encoders == "frozen+ahead"                        # bench.py --encode-ahead (its `clr_encode_ahead` secondary)
    args = argparse.Namespace(no_dead_knn=False, encode_ahead=ahead)
    return bench, bench.Workload(kind, dev, 0, 1, args, encoders="frozen" if ahead else encoders)


@pytest.mark.parametrize("kind,encoders", [("clr", "frozen"), ("clr", "frozen+ahead"), ("clr", "precomputed"), ("pose", "frozen")])
def test_captured_replay_equals_eager_bitwise(kind, encoders):
    bench, wl = _bench_workload(kind, encoders)
    K = 4
    for i in range(2):                                        # warm-up, as bench.measure does before it captures
        wl.step(i)
    torch.cuda.synchronize()
    graphs, opt_graph = bench.capture(wl, split=False)
    start = bench.snapshot(wl)
    for i in range(K):                                        # the timed region's step, verbatim
        bench.run_step(wl, graphs, opt_graph, False, 2 + i)
    torch.cuda.synchronize()
    replayed = bench.state_digest(wl)
    losses_replayed = [float(wl.cap_ret[(2 + i) % len(wl.pool)][0]) for i in range(K)]   # each pool batch once: its last replay
    bench.restore(wl, start)
    losses_eager = []
    for i in range(K):
        losses_eager.append(float(wl.step(2 + i)[0]))
    torch.cuda.synchronize()
    eager = bench.state_digest(wl)
    assert losses_replayed == losses_eager
    assert replayed.keys() == eager.keys()
    changed = 0
    for k in eager:
        assert torch.equal(replayed[k], eager[k]), k
        changed += int(not torch.equal(eager[k], start[0][k[6:]])) if k.startswith("model.") else 0
    assert changed >= 30                                      # the steps did train (weights, running statistics, counters)
    # and twice the same eager steps from the same state: run-to-run reproducibility of the whole step
    bench.restore(wl, start)
    for i in range(K):
        wl.step(2 + i)
    torch.cuda.synchronize()
    again = bench.state_digest(wl)
    for k in eager:
        assert torch.equal(again[k], eager[k]), k


def test_bench_loss_check_field():
    bench, wl = _bench_workload("clr")
    for i in range(2):
        wl.step(i)
    torch.cuda.synchronize()
    graphs, _ = bench.capture(wl, split=False)
    lc = bench.loss_check(wl, graphs, 5)
    assert lc["equal"] and lc["replayed_loss"] > 0, lc


def _train_pair(salt, dev):
    """Oracle and HIP model with the same weights, both in TRAIN mode (clr_att_gnn.py:26-33 freezes the encoders'
    parameters but leaves them in train mode: batch-statistics BatchNorm, running-statistics updates); Dropout
    (pointnet.py:190, radarnet.py:62) neutralised on both sides, its mask is not part of the contract."""
    from batch3dmot_amd import encoders
    from batch3dmot_amd.clr_att_gnn import GNN
    from oracle import ref_encoders, ref_torch
    ora = ref_torch.GNN(ref_encoders.ResNetAE(), ref_encoders.PointNetClassifier(k=7), ref_encoders.RadarNetClassifier(k=7),
                        run_dead_knn=False, loop_masks=False)
    seeded_fill_(ora, salt)
    m = GNN(encoders.ResNetAE(), encoders.PointNetClassifier(k=7), encoders.RadarNetClassifier(k=7))
    m.load_state_dict(ora.state_dict())
    m = m.to(dev)
    for mod in (ora, m):
        mod.train()
        mod.pointnet.dropout.p = 0.0
        mod.radarnet.dropout.p = 0.0
    return ora, m


@pytest.mark.parametrize("size", ["n300", "bench_size"])
def test_train_mode_step_matches_oracle(size):
    from batch3dmot_amd import synth
    from batch3dmot_amd.train_step import make_optimizer, train_step
    from oracle import ref_torch
    dev = torch.device("cuda:0")
    if size == "n300":
        data = synth.make_graph(300, None, k=6, graph_idx=733, modalities=True)
        bs = 1
    else:
        data = synth.make_batch(2, 1500, 15000, first_graph_idx=52, modalities=True)
        bs = 2
    ora, m = _train_pair(31, dev)
    o_opt = torch.optim.Adam([p for p in ora.parameters() if p.requires_grad], lr=1e-4, weight_decay=1e-4, betas=(0.9, 0.999))
    r_loss, r_out, r_sens = ref_torch.train_step(ora, data, o_opt, batch_size=bs, loss_kind="cb", logits=False)
    opt = make_optimizer(m)
    loss, out, x_sens = train_step(m, data.to(dev), opt, batch_size=bs, loss_kind="cb", logits=False)
    torch.cuda.synchronize()
    assert m.pointnet.training and m.radarnet.training and m.resnet.training        # no sticky eval switch here
    assert abs(float(loss) - float(r_loss)) <= 1e-5 * abs(float(r_loss)), (float(loss), float(r_loss))
    assert rel(out.reshape(-1), r_out.reshape(-1)) < TOL
    assert rel(x_sens, r_sens) < TOL
    # running statistics of every BatchNorm of the three encoders (the train-mode side effect)
    ob = dict(ora.named_buffers())
    n_stats = 0
    for name, buf in m.named_buffers():
        if name.endswith("running_mean") or name.endswith("running_var"):
            if not any(name.startswith(p) for p in ("resnet.bn.", "resnet.fc_", "pointnet.fc3", "radarnet.fc3")):
                assert rel(buf, ob[name]) < 1e-5, (name, rel(buf, ob[name]))
                n_stats += 1
        elif name.endswith("num_batches_tracked"):
            assert int(buf) == int(ob[name]), name
    assert n_stats >= 40
    # weights after the Adam step
    have = grad_digest({n: p.detach() for n, p in m.named_parameters() if p.requires_grad})
    want = grad_digest({n: p.detach() for n, p in ora.named_parameters() if p.requires_grad})
    for n, w in want.items():
        # (an element whose gradient is a few 1e-8 moves by a fraction of lr with the summation order -- conftest.assert_adam_heads_close
        # holds the elements; the norm may differ by that much per element)
        numel = 1
        for d_ in w["shape"]:
            numel *= d_
        assert abs(have[n]["norm"] - w["norm"]) <= 2e-6 * max(w["norm"], 1e-6) + 0.02 * 1e-4 * numel ** 0.5, n
        torch.testing.assert_close(have[n]["head"], w["head"], rtol=1e-5, atol=2e-7)


def test_sticky_eval_switch_inside_a_training_step():
    """clr_att_gnn.py:128-130,136-138: fewer than two rows of a modality switch that encoder (and its head) to eval for
    good; the step still trains everything else."""
    from batch3dmot_amd import synth
    from batch3dmot_amd.train_step import make_optimizer, train_step
    from oracle import ref_torch
    dev = torch.device("cuda:0")
    data = synth.make_graph(120, None, k=6, graph_idx=741, modalities=True)
    data.radar_feats[1:] = 0.0                                 # one radar row at most
    data.radar_feats[0, 0, 0] = 1.0
    ora, m = _train_pair(37, dev)
    o_opt = torch.optim.Adam([p for p in ora.parameters() if p.requires_grad], lr=1e-4, weight_decay=1e-4, betas=(0.9, 0.999))
    r_loss, r_out, _ = ref_torch.train_step(ora, data, o_opt, batch_size=1, loss_kind="cb", logits=False)
    opt = make_optimizer(m)
    loss, out, _ = train_step(m, data.to(dev), opt, batch_size=1, loss_kind="cb", logits=False)
    assert not m.radarnet.training and not m.fc_radar_encoder.training and m.pointnet.training
    assert not ora.radarnet.training
    assert abs(float(loss) - float(r_loss)) <= 1e-5 * abs(float(r_loss))
    assert rel(out.reshape(-1), r_out.reshape(-1)) < TOL


@pytest.mark.parametrize("name", ["g9_train_mode_step.pt", "g9b_train_mode_one_radar_row.pt", "g11_train_mode_dropout_live.pt"])
def test_train_mode_step_matches_reference_golden(name, monkeypatch):
    """The same step against the REFERENCE's own train-mode run (oracle/make_golden.py:golden_train_mode_step executes
    clr_att_gnn.py / pointnet.py / radarnet.py / resnet_fully_conv.py in .train()): scores, loss, x_sens,
    the encoders' BatchNorm running statistics, which sub-modules ended in eval mode (g9b: one radar row), and every
    trainable weight after Adam.  g9 / g9b: Dropout p = 0.  g11: Dropout LIVE (p = 0.3, pointnet.py:190 / radarnet.py:62) --
    the fixture holds the masks the reference's run drew and the HIP fc heads (b3d_fc_bn_forward's `mask`) are fed the same."""
    from conftest import assert_adam_heads_close, data_from, load_golden
    from batch3dmot_amd import encoders
    from batch3dmot_amd.clr_att_gnn import GNN
    from batch3dmot_amd.train_step import make_optimizer, train_step
    dev = torch.device("cuda:0")
    g = load_golden(name)
    data = data_from(g["data"]).to(dev)
    m = GNN(encoders.ResNetAE(), encoders.PointNetClassifier(k=7), encoders.RadarNetClassifier(k=7))
    seeded_fill_(m, g["salt"])
    m = m.to(dev).train()
    masks = None
    if "dropout_masks" in g:
        masks = [mk.to(dev) for mk in g["dropout_masks"]]

        def recorded(b, n, p, device):
            mk = masks.pop(0)
            assert tuple(mk.shape) == (b, n) and abs(p - g["dropout_p"]) < 1e-12
            return mk.contiguous()
        monkeypatch.setattr(encoders, "_draw_dropout_mask", recorded)
    else:
        m.pointnet.dropout.p = 0.0
        m.radarnet.dropout.p = 0.0
    opt = make_optimizer(m)
    before = {n: p.detach().reshape(-1)[:8].double().cpu().clone() for n, p in m.named_parameters() if p.requires_grad}
    encoders.path_counts(reset=True)
    loss, out, x_sens = train_step(m, data, opt, batch_size=2, loss_kind="cb", logits=False)
    torch.cuda.synchronize()
    assert masks is None or not masks                         # g11: both Dropout layers consumed their recorded mask
    took = encoders.path_counts()
    assert took and all(kind == "hip" for (_, kind) in took), took    # every encoder stage ran in the HIP kernels
    assert rel(out.reshape(-1), g["out"].reshape(-1)) < TOL and rel(x_sens, g["x_sens"]) < TOL
    assert abs(float(loss) - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    assert {"pointnet": m.pointnet.training, "radarnet": m.radarnet.training, "resnet": m.resnet.training,
            "fc_lidar_encoder": m.fc_lidar_encoder.training, "fc_radar_encoder": m.fc_radar_encoder.training} == g["modes"]
    bufs = dict(m.named_buffers())
    for n, v in g["running_stats"].items():
        if n.endswith("num_batches_tracked"):
            assert int(bufs[n]) == int(v), n
        else:
            assert rel(bufs[n], v) < 1e-5, (n, rel(bufs[n], v))
    # the gradients the step wrote (FlatAdam's flat buffer: b3d_adam_step reads it, never writes it) against the reference's
    from conftest import assert_grad_digest_close
    assert_grad_digest_close(grad_digest({n: p.grad for n, p in m.named_parameters() if p.requires_grad and p.grad is not None}),
                             {n: w for n, w in g["grad_digest"].items() if w is not None})
    if "grads" in g:     # g9 / g11 (round 6): every entry of every gradient tensor of the train-mode step against the reference's run
        from conftest import assert_grads_entrywise
        dead = [n for n, w in g["grads"].items() if w is None]                 # knn_conv: no gradient in the reference (pose_gnn.py:80)
        have_g = {n: p.grad for n, p in m.named_parameters() if p.requires_grad}
        assert all(have_g[n] is None or float(have_g[n].abs().max()) == 0.0 for n in dead), dead
        assert_grads_entrywise(have_g, {n: w for n, w in g["grads"].items() if w is not None}, tol=TOL, ref_dev=g.get("grads_f64_dev"))
    have = grad_digest({n: p.detach() for n, p in m.named_parameters() if p.requires_grad})
    for n, w in g["after_digest"].items():
        # (an element whose gradient is a few 1e-8 moves by a fraction of lr with the summation order -- conftest.assert_adam_heads_close
        # holds the elements; the norm may differ by that much per element)
        numel = 1
        for d_ in w["shape"]:
            numel *= d_
        assert abs(have[n]["norm"] - w["norm"]) <= 2e-6 * max(w["norm"], 1e-6) + 0.02 * 1e-4 * numel ** 0.5, n
    assert_adam_heads_close(before, have, g["after_digest"], lr=1e-4)


def test_encode_ahead_equals_the_sequential_loop_bitwise():
    """train_step.EncodeAhead: the frozen encoders of batch k + 1 enqueued on a side stream under the step of batch k.  Train
    mode, Dropout ACTIVE: losses, parameters, BatchNorm statistics and counters after four steps are bit-equal to the loop
    that encodes inside forward (same order of batches through the encoders, same order of generator draws)."""
    from batch3dmot_amd import encoders, synth
    from batch3dmot_amd.clr_att_gnn import GNN
    from batch3dmot_amd.train_step import EncodeAhead, make_optimizer, train_step
    dev = torch.device("cuda:0")
    batches = [synth.make_batch(2, 150, 900, first_graph_idx=700 + 2 * i, modalities=True).to(dev) for i in range(4)]

    def run(pipelined):
        m = GNN(encoders.ResNetAE(), encoders.PointNetClassifier(k=7), encoders.RadarNetClassifier(k=7))
        seeded_fill_(m, 31)
        m = m.to(dev).train()
        torch.manual_seed(7)
        torch.cuda.manual_seed(7)
        opt = make_optimizer(m)
        losses = []
        ahead = EncodeAhead(m) if pipelined else None
        if pipelined:
            ahead.launch(batches[0])
        for k, b in enumerate(batches):
            kw, hook = None, None
            if pipelined:
                kw = {"encoded": ahead.take(b)}
                if k + 1 < len(batches):
                    nxt = batches[k + 1]
                    if pipelined == "split":             # bench.py's placement: camera under the forward, point encoders under the backward
                        ahead.launch(nxt, parts="img")
                        hook = lambda nxt=nxt: ahead.launch(nxt, parts="points")      # noqa: E731
                    elif pipelined == "radar_first":     # RadarNet launched in front of PointNet: PointNet's Dropout mask is pre-drawn
                        ahead.launch(nxt, parts="img")
                        ahead.launch(nxt, parts="radar")
                        hook = lambda nxt=nxt: ahead.launch(nxt, parts="lidar")       # noqa: E731
                    else:
                        ahead.launch(nxt)
            loss, _, _ = train_step(m, b, opt, batch_size=2, loss_kind="cb", logits=False, forward_kwargs=kw, after_forward=hook)
            losses.append(float(loss))
        torch.cuda.synchronize()
        return losses, {k: v.detach().clone() for k, v in m.state_dict().items()}

    l_seq, s_seq = run(False)
    for form in (True, "split", "radar_first"):
        l_pipe, s_pipe = run(form)
        assert l_seq == l_pipe, form
        for k in s_seq:
            assert torch.equal(s_seq[k], s_pipe[k]), (form, k)
    assert len({round(x, 6) for x in l_seq}) == 4             # four different batches, four losses


def test_encode_ahead_parts_are_checked():
    """EncodeAhead.launch(parts=...): a batch cannot be taken before all of its parts were launched, a part cannot be launched twice,
    and the next batch cannot start before the previous one was taken."""
    from batch3dmot_amd import encoders, synth
    from batch3dmot_amd.clr_att_gnn import GNN
    from batch3dmot_amd.train_step import EncodeAhead
    dev = torch.device("cuda:0")
    a, b = (synth.make_batch(2, 60, 300, first_graph_idx=900 + 2 * i, modalities=True).to(dev) for i in range(2))
    m = GNN(encoders.ResNetAE(), encoders.PointNetClassifier(k=7), encoders.RadarNetClassifier(k=7)).to(dev).eval()
    ahead = EncodeAhead(m)
    with pytest.raises(ValueError):
        ahead.launch(a, parts="everything")
    ahead.launch(a, parts="img")
    with pytest.raises(RuntimeError, match="only parts"):
        ahead.take(a)
    with pytest.raises(RuntimeError):
        ahead.launch(a, parts="img")                      # twice
    with pytest.raises(RuntimeError):
        ahead.launch(b, parts="img")                      # the previous batch was never taken
    ahead.launch(a, parts="radar")
    ahead.launch(a, parts="lidar")
    got = ahead.take(a)
    with torch.no_grad():
        want = m.encode_modalities(a)
    torch.cuda.synchronize()
    assert len(got) == 5
    for x, y in zip(got, want):
        assert torch.equal(x, y)
    with pytest.raises(RuntimeError):
        ahead.take(a)                                     # nothing pending
    ahead.launch(b)                                       # ... and the next batch goes through as a whole
    assert len(ahead.take(b)) == 5
