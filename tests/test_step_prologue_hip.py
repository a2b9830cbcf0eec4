"""GPU: the parts of a step's prologue that round 6 moved off the launch stream / off the host:
  * ``GNN.modality_rows_into`` (``b3d_modality_rows_expect``): row ids of clr_att_gnn.py:107-121 without a host read-back, the
    counts checked on the device -- the capturable form of ``modality_rows``;
  * ``_lib.Graph(..., ws=...)`` / ``EncodeAhead.launch_graph``: the CSR / CSC structure in a caller-owned buffer, built for the NEXT
    batch on the side stream;
  * roctx markers (``b3d_prof_markers``) do not disturb a forward."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(dev):
    from batch3dmot_amd import encoders
    from batch3dmot_amd.clr_att_gnn import GNN
    from oracle.seeded import seeded_fill_
    m = GNN(encoders.ResNetAE(), encoders.PointNetClassifier(k=7), encoders.RadarNetClassifier(k=7))
    seeded_fill_(m, 77)
    return m.to(dev)


def test_modality_rows_into_matches_the_read_back_form_and_flags_a_mismatch():
    from batch3dmot_amd import synth
    dev = torch.device("cuda:0")
    m = _model(dev)
    data = synth.make_batch(2, 150, 900, first_graph_idx=810, modalities=True).to(dev)
    li, ri = m.modality_rows(data)
    n = data.pose_feats.size(0)
    # reference semantics: torch.nonzero of the row sums
    assert torch.equal(li, torch.nonzero(data.lidar_feats.reshape(n, -1).sum(1) != 0).squeeze(1))
    assert torch.equal(ri, torch.nonzero(data.radar_feats.reshape(n, -1).sum(1) != 0).squeeze(1))
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    sl, sr = torch.full_like(li, -1), torch.full_like(ri, -1)
    m.modality_rows_into(data, (sl, sr), flag)
    torch.cuda.synchronize()
    assert torch.equal(sl, li) and torch.equal(sr, ri) and int(flag) == 0
    # a buffer one entry short (a batch with MORE rows than the captured graph expects): nothing is written past it, the flag says so
    short = torch.full((li.numel() - 1,), -1, dtype=torch.int64, device=dev)
    guard = torch.full((ri.numel() + 3,), -7, dtype=torch.int64, device=dev)
    m.modality_rows_into(data, (short, guard[: ri.numel()]), flag)
    torch.cuda.synchronize()
    assert torch.equal(short, li[:-1]) and torch.equal(guard[: ri.numel()], ri) and bool((guard[ri.numel():] == -7).all())
    assert int(flag) == 1
    # ... and one entry long (FEWER rows than expected)
    longer = torch.full((li.numel() + 1,), -1, dtype=torch.int64, device=dev)
    m.modality_rows_into(data, (longer, sr), flag)
    torch.cuda.synchronize()
    assert torch.equal(longer[:-1], li) and int(longer[-1]) == -1 and int(flag) == 2
    with pytest.raises(ValueError):
        m.modality_rows_into(data, (sl.int(), sr), flag)


def test_modality_rows_into_is_capturable():
    from batch3dmot_amd import synth
    dev = torch.device("cuda:0")
    m = _model(dev)
    data = synth.make_batch(1, 200, 1200, first_graph_idx=820, modalities=True).to(dev)
    li, ri = m.modality_rows(data)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    sl, sr = torch.zeros_like(li), torch.zeros_like(ri)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
        m.modality_rows_into(data, (sl, sr), flag)
    torch.cuda.current_stream().wait_stream(s)
    for _ in range(3):
        sl.fill_(-1); sr.fill_(-1)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(sl, li) and torch.equal(sr, ri)
    assert int(flag) == 0
    data.radar_feats.zero_()                  # the next replay sees a batch without radar rows: the device-side check reports it
    g.replay()
    torch.cuda.synchronize()
    assert int(flag) == (1 if ri.numel() else 0)


def test_graph_in_a_caller_owned_buffer_and_built_ahead():
    from batch3dmot_amd import _lib, synth
    from batch3dmot_amd.train_step import EncodeAhead
    dev = torch.device("cuda:0")
    data = synth.make_batch(2, 120, 800, first_graph_idx=830, modalities=True).to(dev)
    n, e = data.pose_feats.size(0), data.edge_index.size(1)
    ref = _lib.Graph(data.edge_index, n).arrays()
    ws = torch.empty(_lib.Graph.workspace_bytes(n, e) + 64, dtype=torch.uint8, device=dev)
    own = _lib.Graph(data.edge_index, n, ws=ws)
    assert own.ws is ws
    for k, v in own.arrays().items():
        assert torch.equal(v, ref[k]), k
    with pytest.raises(ValueError):
        _lib.Graph(data.edge_index, n, ws=ws[:100])
    m = _model(dev).eval()
    ahead = EncodeAhead(m)
    if hasattr(data, "_b3d_graph"):
        del data._b3d_graph
    g2 = ahead.launch_graph(data, ws=ws)
    torch.cuda.current_stream().wait_stream(ahead.stream)
    assert data._b3d_graph is g2 and g2.ws is ws
    for k, v in g2.arrays().items():
        assert torch.equal(v, ref[k]), k
    with torch.no_grad():
        a = m(data)[0]                                   # the forward finds the structure and uses it
        assert data._b3d_graph is g2
        del data._b3d_graph
        b = m(data)[0]
    assert torch.equal(a, b)


def test_roctx_markers_switch_on_and_off_around_a_forward():
    from batch3dmot_amd import _lib, synth
    from batch3dmot_amd.pose_gnn import PoseGNN
    dev = torch.device("cuda:0")
    data = synth.make_graph(120, None, k=5, graph_idx=840).to(dev)
    m = PoseGNN().to(dev)
    with torch.no_grad():
        a = m(data)[0]
        on = _lib.prof_markers(True)                     # False only where no roctx library exists
        b = m(data)[0]
        assert _lib.prof_markers(False) is False
        c = m(data)[0]
    torch.cuda.synchronize()
    assert isinstance(on, bool) and torch.equal(a, b) and torch.equal(a, c)


def test_past_run_sums_follow_the_edge_order():
    """Round 6: with the edges grouped by destination (how detection graphs list them) the camera+LiDAR+radar edge kernel adds the
    `past` messages of a destination inside the wavefront (DPP row shifts) and the node kernel sums one row per (destination, 16-edge
    block) run; any other edge order keeps one row per edge.  Both orders of the SAME graph must give the same model output (fp32
    summation order aside) -- forward and every gradient -- and the graph build must say which form it chose."""
    from batch3dmot_amd import _lib, synth
    dev = torch.device("cuda:0")
    m = _model(dev).eval()
    data = synth.make_graph(260, None, k=7, graph_idx=850, modalities=True)
    e = data.edge_index.size(1)
    assert bool((data.edge_index[1][1:] >= data.edge_index[1][:-1]).all())          # synthetic graphs are grouped by destination
    g_sorted = _lib.Graph(data.edge_index.to(dev), 260)
    flag = lambda g: int(g._view(g.c.dst_unsorted, 1).item())                        # noqa: E731
    assert flag(g_sorted) == 0
    n_tail = int(g_sorted._view(g_sorted.c.past_ptr, 261)[-1])
    assert 0 < n_tail < e and n_tail <= e // 16 + 260 + 1                            # run tails: far fewer rows than edges
    perm = torch.randperm(e, generator=torch.Generator().manual_seed(3))
    shuffled = synth.make_graph(260, None, k=7, graph_idx=850, modalities=True)
    shuffled.edge_index = data.edge_index[:, perm].contiguous()
    shuffled.edge_attr = data.edge_attr[perm].contiguous()
    g_shuf = _lib.Graph(shuffled.edge_index.to(dev), 260)
    assert flag(g_shuf) == 1 and int(g_shuf._view(g_shuf.c.past_ptr, 261)[-1]) == e  # one row per edge
    outs = []
    for d, p in ((data, None), (shuffled, perm)):
        m.zero_grad(set_to_none=True)
        out, xs = m(d.to(dev))
        w = torch.linspace(-1.0, 1.0, e, device=dev).reshape(-1, 1)
        if p is not None:
            w = w[p.to(dev)]
        ((out * w).sum() + 0.1 * xs.sum()).backward()
        o = out.detach().reshape(-1)
        if p is not None:
            o = torch.empty_like(o).index_copy_(0, p.to(dev), o)                     # back to the sorted edge order
        outs.append((o, xs.detach(), {n: q.grad.detach().clone() for n, q in m.named_parameters() if q.grad is not None}))
    (o1, x1, g1), (o2, x2, g2) = outs
    rel = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))   # noqa: E731
    assert rel(o1, o2) < 1e-5 and rel(x1, x2) < 1e-6
    # two fp32 evaluations with different summation orders everywhere (every per-edge tensor is permuted): 5e-4 of a tensor's largest
    # entry, with conftest's clause for a ReLU unit that sits within rounding of zero on this 260-node graph
    from conftest import assert_grad_close
    for n in g2:
        assert_grad_close(g1[n], g2[n], "edge order / " + n)


def test_skipping_the_dead_last_layer_messages_changes_nothing():
    """`GNN.run_dead_last_messages = False` (B3D_FLAG_SKIP_DEAD_LAST_MESSAGES): the last layer's create_future_msgs / create_past_msgs /
    combine_future_past feed nothing (clr_att_gnn.py:188 returns edge_classifier(edge_attr)) -- scores, x_sens and EVERY gradient are
    bit-identical with and without them, in training and in inference."""
    from batch3dmot_amd import synth
    dev = torch.device("cuda:0")
    m = _model(dev).train()
    for sub in (m.pointnet, m.radarnet):
        sub.dropout.p = 0.0
    data = synth.make_batch(2, 140, 900, first_graph_idx=860, modalities=True).to(dev)
    rows = m.modality_rows(data)
    m.eval()                                                  # (frozen encoders in eval: the two runs must see the same BatchNorm statistics)
    res = []
    for run_dead in (True, False):
        m.run_dead_last_messages = run_dead
        m.zero_grad(set_to_none=True)
        out, xs = m(data, rows=rows)
        w = torch.linspace(-1.0, 1.0, out.numel(), device=dev).reshape(out.shape)
        ((out * w).sum() + 0.1 * xs.sum()).backward()
        res.append((out.detach().clone(), xs.detach().clone(), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}))
        with torch.no_grad():
            res[-1] += (m(data, rows=rows)[0].clone(),)
    (o1, x1, g1, i1), (o2, x2, g2, i2) = res
    assert torch.equal(o1, o2) and torch.equal(x1, x2) and torch.equal(i1, i2)
    assert set(g1) == set(g2)
    for n in g1:
        assert torch.equal(g1[n], g2[n]), n


def test_a_graph_struct_without_run_lists_keeps_one_row_per_edge():
    """`b3d_graph.past_ptr / past_rows / dst_unsorted` are optional (include/b3d.h): a hand-made struct that leaves them NULL gets the
    one-row-per-edge path -- same scores as the run sums up to fp32 summation order."""
    from batch3dmot_amd import _lib, synth
    dev = torch.device("cuda:0")
    m = _model(dev).eval()
    data = synth.make_graph(220, None, k=6, graph_idx=870, modalities=True).to(dev)
    rows = m.modality_rows(data)
    with torch.no_grad():
        a = m(data, rows=rows)[0].clone()
        g = data._b3d_graph
        assert int(g._view(g.c.dst_unsorted, 1).item()) == 0          # grouped by destination: the run sums were used
        keep = (g.c.past_ptr, g.c.past_rows, g.c.dst_unsorted)
        g.c.past_ptr, g.c.past_rows, g.c.dst_unsorted = None, None, None
        b = m(data, rows=rows)[0].clone()
        g.c.past_ptr, g.c.past_rows, g.c.dst_unsorted = keep
        c = m(data, rows=rows)[0].clone()
    assert torch.equal(a, c)                                           # bitwise repeatable
    assert float((a - b).abs().max()) <= 2e-6 * float(a.abs().max())


def test_run_tail_lists_are_exact():
    """Index work is bit-exact (north_star): `b3d_graph.past_ptr / past_rows` against a plain-Python restatement -- for a graph grouped by
    destination the last edge of every (destination, aligned 16-edge block) run; for any other order the destination's CSR list."""
    from batch3dmot_amd import _lib, synth
    dev = torch.device("cuda:0")
    for n, k, idx in ((37, 3, 880), (260, 7, 881), (700, 13, 882)):
        data = synth.make_graph(n, None, k=k, graph_idx=idx)
        ei = data.edge_index
        e = ei.size(1)
        g = _lib.Graph(ei.to(dev), n)
        ptr = g._view(g.c.past_ptr, n + 1).cpu().tolist()
        rows = g._view(g.c.past_rows, ptr[-1]).cpu().tolist()
        dst = ei[1].tolist()
        want_ptr, want_rows = [0], []
        for node in range(n):
            edges = [i for i in range(e) if dst[i] == node]            # contiguous: the graph is grouped by destination
            blocks = sorted({i // 16 for i in edges})
            want_rows += [max(i for i in edges if i // 16 == b) for b in blocks]
            want_ptr.append(len(want_rows))
        assert int(g._view(g.c.dst_unsorted, 1).item()) == 0 and ptr == want_ptr and rows == want_rows, (n, k)
        # the same graph with its edges reversed: not grouped by ascending destination -> the CSR lists (edge ids ascending per node)
        rev = torch.flip(ei, dims=[1]).contiguous()
        g2 = _lib.Graph(rev.to(dev), n)
        a = g2.arrays()
        ptr2 = g2._view(g2.c.past_ptr, n + 1).cpu()
        assert int(g2._view(g2.c.dst_unsorted, 1).item()) == 1
        assert torch.equal(ptr2, a["dst_ptr"].cpu()) and torch.equal(g2._view(g2.c.past_rows, e).cpu(), a["dst_perm"].cpu())
