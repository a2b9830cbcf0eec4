"""GPU: the HIP PoseGNN path (through the C ABI) against the golden vectors of the reference and
against the CPU oracle.  Tolerances: 1e-4 relative to the tensor's max-abs for floating-point
node / edge features (BASELINE.json north_star), bit-exact for index results."""
import ctypes as C

import pytest
import torch

from conftest import data_from, load_golden

pytestmark = pytest.mark.gpu

TOL = 1e-4


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _loss_weights(t, salt):
    g = torch.Generator().manual_seed(1234 + salt)
    return torch.randn(t.shape, generator=g)


def _model(state_dict, dev):
    from batch3dmot_amd.pose_gnn import PoseGNN
    m = PoseGNN().to(dev)
    m.load_state_dict(state_dict, strict=True)
    return m


def _layer_tensors(m, N, E):
    from batch3dmot_amd import _lib
    ws, nbytes, flags, _, _ = m._last_workspace
    lib = _lib.load()
    out = []
    for l in range(m.depth + 1):
        px, pe = C.c_void_p(), C.c_void_p()
        _lib.check(lib.b3d_pose_debug_layer_ptrs(ws.data_ptr(), nbytes, N, E, m.depth, flags, l,
                                                 C.byref(px), C.byref(pe)), "layer ptrs")
        ox, oe = px.value - ws.data_ptr(), pe.value - ws.data_ptr()
        out.append((ws[ox:ox + N * 48 * 4].view(torch.float32).view(N, 48).clone(),
                    ws[oe:oe + E * 32 * 4].view(torch.float32).view(E, 32).clone()))
    return out


@pytest.mark.parametrize("name", ["g1_pose.pt", "g1b_pose_batch2.pt", "g5_pose_tiny.pt"])
@pytest.mark.parametrize("dead_knn", [False, True])
def test_forward_backward_match_reference_golden(name, dead_knn):
    dev = torch.device("cuda:0")
    g = load_golden(name)
    data = data_from(g["data"]).to(dev)
    m = _model(g["state_dict"], dev)
    m.run_dead_knn = dead_knn
    m.keep_workspace = True
    out, x_enc = m(data)
    assert out.shape == g["out"].shape and x_enc.shape == g["x_enc"].shape
    assert rel(out, g["out"]) < TOL and rel(x_enc, g["x_enc"]) < TOL
    N, E = data.pose_feats.size(0), data.edge_index.size(1)
    for (x, e), (gx, ge) in zip(_layer_tensors(m, N, E)[1:], g["layers"]):
        assert rel(x, gx) < TOL and rel(e, ge) < TOL
    loss = (out * _loss_weights(out, 0).to(dev)).sum() + (x_enc * _loss_weights(x_enc, 1).to(dev)).sum()
    loss.backward()
    for n, p in m.named_parameters():
        gg = g["grads"][n]
        if gg is None:
            assert p.grad is None, n          # knn_conv: the reference discards its result
        else:
            assert rel(p.grad, gg) < TOL, n


def test_state_dict_keys_match_reference():
    from batch3dmot_amd.pose_gnn import PoseGNN
    g = load_golden("g1_pose.pt")
    m = PoseGNN()
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == \
           {k: tuple(v.shape) for k, v in g["state_dict"].items()}


def test_graph_structure_is_exact():
    from batch3dmot_amd import _lib, synth
    dev = torch.device("cuda:0")
    d = synth.make_graph(400, None, k=12, graph_idx=7)
    ei = d.edge_index
    perm = torch.randperm(ei.size(1), generator=torch.Generator().manual_seed(1))
    for edges in (ei, ei[:, perm].contiguous()):          # destination-sorted and shuffled
        gr = _lib.Graph(edges.to(dev), 400)
        a = {k: v.cpu().long() for k, v in gr.arrays().items()}
        assert torch.equal(a["src"], edges[0]) and torch.equal(a["dst"], edges[1])
        for key, row in (("dst", 1), ("src", 0)):
            order = torch.argsort(edges[row], stable=True)
            assert torch.equal(a[key + "_perm"], order)        # grouped by node, ascending edge id
            cnt = torch.bincount(edges[row], minlength=400)
            assert torch.equal(a[key + "_ptr"], torch.cat([torch.zeros(1, dtype=torch.long), cnt.cumsum(0)]))


def test_edge_order_does_not_matter():
    """The reference emits destination-sorted edges; the kernels must not rely on it."""
    from batch3dmot_amd import synth
    dev = torch.device("cuda:0")
    g = load_golden("g1_pose.pt")
    d = data_from(g["data"])
    perm = torch.randperm(d.edge_index.size(1), generator=torch.Generator().manual_seed(3))
    d2 = data_from({**g["data"], "edge_index": d.edge_index[:, perm].contiguous(), "edge_attr": d.edge_attr[perm]})
    m = _model(g["state_dict"], dev)
    out, _ = m(d2.to(dev))
    assert rel(out, g["out"][perm]) < TOL


def test_full_size_against_oracle_and_float64():
    """BASELINE.json config[1] size (3,000 nodes / ~30,000 edges): outputs within 1e-4 of the CPU
    oracle and of a float64 evaluation; gradients as close to the float64 evaluation as fp32 gets.

    Through six ReLU layers the fp32 gradient of this model is chaotic at the 1e-4 .. 1e-3 level: a
    pre-activation that rounds to the other side of zero flips a unit.  Against float64, over eight
    synthetic batches (tests/tools/grad_error_seeds.py, MI355X) the l2-relative gradient error per parameter is
    1e-5 .. 1.0e-3 for the fp32 CPU oracle and 3e-5 .. 1.1e-3 for the HIP path, neither consistently
    ahead -- while the forward outputs of both sit at 1.5e-6.  GRAD_TOL bounds that range; the small-graph
    tests hold the gradients to 1e-4 against the oracle where no unit sits near a tie."""
    GRAD_TOL = 3e-3
    import copy
    from batch3dmot_amd import synth
    from batch3dmot_amd.data import Data
    from oracle import ref_torch
    from oracle.seeded import seeded_fill_
    dev = torch.device("cuda:0")
    big = synth.make_batch(2, 1500, 15000, first_graph_idx=50)
    ora = ref_torch.PoseGNN(run_dead_knn=False)
    seeded_fill_(ora, 5)
    m = _model(ora.state_dict(), dev)
    lw = _loss_weights(torch.empty(big.edge_index.size(1), 1), 3)
    out, x_enc = m(big.to(dev))
    (out * lw.to(dev)).sum().backward()
    o32, x32 = ora(big)
    (o32 * lw).sum().backward()
    assert rel(out, o32) < TOL and rel(x_enc, x32) < TOL
    ora64 = copy.deepcopy(ora).double()
    ora64.zero_grad()
    big64 = Data(**{k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v)
                    for k, v in big.__dict__.items()})
    big64.edge_attr = big.edge_attr.float().double()
    e64 = ora64.edge_encoder(big64.edge_attr)
    x64 = ora64.node_encoder(big64.pose_feats)
    x0 = x64
    for _ in range(6):
        x64, e64 = ora64.message_passing(x64, big64.edge_index, e64, x0)
    o64 = ora64.edge_classifier(e64)
    (o64 * lw.double()).sum().backward()
    assert rel(out, o64) < TOL
    for (n, p), (_, q), (_, r) in zip(m.named_parameters(), ora.named_parameters(), ora64.named_parameters()):
        if r.grad is None:
            continue
        err_hip, err_cpu = rel(p.grad, r.grad), rel(q.grad, r.grad)
        assert err_hip < max(3.0 * err_cpu, GRAD_TOL), (n, err_hip, err_cpu)


def test_full_size_gradients_entrywise_with_tie_free_weights():
    """3,000 nodes / ~30,000 edges with weights under which no ReLU is within 0.2 of zero on this input
    (tests/tiefree.py; margin measured in the float64 run): every ENTRY of every gradient within 1e-4 of the tensor's
    largest entry against float64."""
    import copy
    from batch3dmot_amd import synth
    from batch3dmot_amd.data import Data
    from oracle import ref_torch
    from oracle.seeded import seeded_fill_
    from tiefree import make_tie_free, measure_margin
    dev = torch.device("cuda:0")
    big = synth.make_batch(2, 1500, 15000, first_graph_idx=50)
    ora = ref_torch.PoseGNN(run_dead_knn=False)
    seeded_fill_(ora, 9)
    ora.to(dev)
    big_dev = copy.deepcopy(big).to(dev)
    make_tie_free(ora, lambda: ora(big_dev), seed=6)
    ora.cpu()
    m = _model(ora.state_dict(), dev)
    lw = _loss_weights(torch.empty(big.edge_index.size(1), 1), 3)
    out, x_enc = m(big.to(dev))
    (out * lw.to(dev)).sum().backward()
    ora64 = copy.deepcopy(ora).double()
    ea64 = big.edge_attr.float().double()
    res = {}

    def fwd64():
        e64 = ora64.edge_encoder(ea64)
        x64 = ora64.node_encoder(big.pose_feats.double())
        x0 = x64
        for _ in range(6):
            x64, e64 = ora64.message_passing(x64, big.edge_index, e64, x0)
        res["out"] = ora64.edge_classifier(e64)
    margin = measure_margin(ora64, fwd64)
    assert margin >= 0.2, margin
    o64 = res["out"]
    assert float(o64.detach().std()) > 1e-3
    (o64 * lw.double()).sum().backward()
    assert rel(out, o64) < TOL
    checked = 0
    for (n, p), (_, r) in zip(m.named_parameters(), ora64.named_parameters()):
        if r.grad is None or float(r.grad.abs().max()) == 0.0:
            continue
        assert rel(p.grad, r.grad) < TOL, (n, rel(p.grad, r.grad))
        checked += 1
    assert checked >= 30


def test_results_are_bitwise_reproducible():
    """No float atomics anywhere: two runs give identical bits (outputs and gradients)."""
    from batch3dmot_amd import synth
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    from batch3dmot_amd.pose_gnn import PoseGNN
    m = PoseGNN().to(dev)
    d = synth.make_graph(600, None, k=10, graph_idx=11).to(dev)
    res = []
    for _ in range(2):
        m.zero_grad(set_to_none=True)
        if hasattr(d, "_b3d_graph"):
            del d._b3d_graph
        out, x_enc = m(d)
        out.square().sum().backward()
        res.append([out.clone(), x_enc.clone()] + [p.grad.clone() for p in m.parameters() if p.grad is not None])
    for a, b in zip(*res):
        assert torch.equal(a, b)


def test_inference_mode_matches_training_forward():
    dev = torch.device("cuda:0")
    g = load_golden("g1_pose.pt")
    data = data_from(g["data"]).to(dev)
    m = _model(g["state_dict"], dev)
    with torch.no_grad():
        out, x_enc = m(data)
    assert rel(out, g["out"]) < TOL and not out.requires_grad


def test_invalid_inputs_raise():
    from batch3dmot_amd import synth
    from batch3dmot_amd.pose_gnn import PoseGNN
    dev = torch.device("cuda:0")
    m = PoseGNN().to(dev)
    d = synth.make_graph(50, None, k=3, graph_idx=1).to(dev)
    bad = data_from({**d.__dict__, "pose_feats": d.pose_feats[:, :18].contiguous()})
    with pytest.raises(ValueError):
        m(bad)
    empty = data_from({**d.__dict__, "edge_index": d.edge_index[:, :0].contiguous(), "edge_attr": d.edge_attr[:0]})
    with pytest.raises(ValueError, match="empty graph"):
        m(empty)
    oob = d.edge_index.clone()
    oob[0, 0] = 10 ** 6
    # an endpoint that is not a node: the reference raises an index error (pose_gnn.py:180); here a ValueError, from the
    # graph build's counter for tensors that are already on the GPU ...
    with pytest.raises(ValueError, match="outside"):
        m(data_from({**d.__dict__, "edge_index": oob}))
    torch.cuda.synchronize()
    # ... the build itself stays consistent (the edge becomes the self loop (0, 0), nothing indexes out of bounds)
    from batch3dmot_amd import _lib
    gr = _lib.Graph(oob, 50, validated=True)
    a = {k: v.cpu().long() for k, v in gr.arrays().items()}
    assert gr.invalid_edges() == 1 and int(a["src"][0]) == 0 and int(a["dst"][0]) == 0
    assert int(a["dst_ptr"][-1]) == oob.size(1) and int(a["src_ptr"][-1]) == oob.size(1)
    assert sorted(a["dst_perm"].tolist()) == list(range(oob.size(1))) and sorted(a["src_perm"].tolist()) == list(range(oob.size(1)))
    # ... and from Data.to() for a batch that comes from the host (no GPU read-back)
    cpu = data_from({**{k: (v.cpu() if torch.is_tensor(v) else v) for k, v in d.__dict__.items()}, "edge_index": oob.cpu()})
    with pytest.raises(ValueError, match="outside"):
        cpu.to(dev)
    neg = d.edge_index.clone()
    neg[1, 3] = -1
    with pytest.raises(ValueError, match="outside"):
        m(data_from({**d.__dict__, "edge_index": neg}))


def _tiny_batch(dev, idx=0):
    from batch3dmot_amd import synth
    return synth.make_batch(2, 60, None, first_graph_idx=700 + idx, k=6).to(dev)


def test_flat_adam_matches_torch_adam():
    """optim.FlatAdam (one b3d_adam_step launch, gradients written in place by backward) == torch.optim.Adam
    (train.py:106-109) on the same model, data and hyper-parameters, step for step."""
    import copy
    from batch3dmot_amd.pose_gnn import PoseGNN
    from batch3dmot_amd.optim import FlatAdam
    from batch3dmot_amd.train_step import train_step
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    a = PoseGNN().to(dev)
    b = copy.deepcopy(a)
    hp = dict(lr=1e-3, weight_decay=1e-4, betas=(0.9, 0.999))
    opt_a = torch.optim.Adam([p for p in a.parameters() if p.requires_grad], **hp)
    opt_b = FlatAdam(b, **hp)
    keys = list(a.state_dict().keys())
    assert list(b.state_dict().keys()) == keys                      # re-pointing keeps names and shapes
    for i in range(4):
        data = _tiny_batch(dev, i)
        la, _, _ = train_step(a, data, opt_a, logits=True, fused_loss=False)
        lb, _, _ = train_step(b, data, opt_b, logits=True)
        torch.testing.assert_close(lb, la, rtol=1e-5, atol=1e-7)
    sa, sb = a.state_dict(), b.state_dict()
    for k in keys:
        torch.testing.assert_close(sb[k], sa[k], rtol=2e-5, atol=2e-7, msg=k)
    assert opt_b.step_count == 4
    # knn_conv never receives a gradient: untouched by both optimizers
    assert torch.equal(sb["knn_conv.bias"], sa["knn_conv.bias"])


def test_flat_adam_capturable_state_round_trip():
    """capturable=True keeps the step counter on the device (it advances on hipGraph replays): state_dict / load_state_dict
    must carry it, or the bias corrections restart at t = 1 against warm moments."""
    import copy
    from batch3dmot_amd.pose_gnn import PoseGNN
    from batch3dmot_amd.optim import FlatAdam
    from batch3dmot_amd.train_step import train_step
    dev = torch.device("cuda:0")
    torch.manual_seed(8)
    a = PoseGNN().to(dev)
    b = copy.deepcopy(a)
    hp = dict(lr=1e-3, weight_decay=1e-4, betas=(0.9, 0.999))
    opt_a = FlatAdam(a, capturable=True, **hp)
    for i in range(3):
        train_step(a, _tiny_batch(dev, 20 + i), opt_a, logits=True)
    sd = opt_a.state_dict()
    assert sd["step"] == 3
    b.load_state_dict(a.state_dict())
    opt_b = FlatAdam(b, capturable=True, **hp)
    opt_b.load_state_dict(copy.deepcopy(sd))
    assert int(opt_b.step_dev) == 3 and opt_b.step_count == 3
    d = _tiny_batch(dev, 30)
    train_step(a, d, opt_a, logits=True)
    train_step(b, d, opt_b, logits=True)
    assert torch.equal(opt_a.flat_param, opt_b.flat_param)          # the resumed run takes the identical 4th step


def test_flat_adam_gradient_accumulation_and_zero_grad():
    from batch3dmot_amd.pose_gnn import PoseGNN
    from batch3dmot_amd.optim import FlatAdam
    dev = torch.device("cuda:0")
    torch.manual_seed(4)
    m = PoseGNN().to(dev)
    opt = FlatAdam(m, lr=1e-3)
    d0, d1 = _tiny_batch(dev, 10), _tiny_batch(dev, 11)
    m(d0)[0].sum().backward()
    g0 = opt.flat_grad.clone()
    assert m.edge_encoder[0].weight.grad.data_ptr() == opt.flat_grad.data_ptr()    # .grad is a view of the buffer
    m(d1)[0].sum().backward()                                                     # no zero_grad: accumulates
    g01 = opt.flat_grad.clone()
    opt.zero_grad()
    m(d1)[0].sum().backward()                                                     # lazy zero_grad: overwritten
    g1 = opt.flat_grad.clone()
    torch.testing.assert_close(g01, g0 + g1, rtol=1e-6, atol=1e-6)
    before = opt.flat_param.clone()
    opt.zero_grad()
    opt.step()                                                                    # nothing deposited: no update
    assert torch.equal(opt.flat_param, before)
    opt.zero_grad(set_to_none=True)
    assert m.edge_encoder[0].weight.grad is None
    sd = opt.state_dict()
    opt.load_state_dict(sd)
    assert opt.step_count == sd["step"]


@pytest.mark.parametrize("side_stream", [False, True])
def test_dead_knn_block_inside_the_model_matches_oracle(side_stream):
    """The reference discards this block's result (pose_gnn.py:80), so nothing downstream can notice a wrong
    one: check the last executed block (layer 4, input x[4]) inside the workspace against the oracle's GATConv
    on the oracle's k-NN graph -- both the path that reads GATConv.lin(x) from the per-node table and the
    side-stream path that computes it itself."""
    from batch3dmot_amd import _lib, synth
    from oracle import ref_torch
    from oracle.seeded import seeded_fill_
    dev = torch.device("cuda:0")
    d = synth.make_graph(240, None, k=6, graph_idx=77)
    ora = ref_torch.PoseGNN(run_dead_knn=False)
    seeded_fill_(ora, 21)
    with torch.no_grad():
        for p in ora.knn_conv.parameters():
            p.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(p.numel())) * 0.3)
    m = _model(ora.state_dict(), dev)
    m.run_dead_knn, m.single_stream, m.keep_workspace = True, not side_stream, True
    m(d.to(dev))
    torch.cuda.synchronize()
    N, E = d.pose_feats.size(0), d.edge_index.size(1)
    x4 = _layer_tensors(m, N, E)[4][0].cpu()
    ws, nbytes, flags, _, _ = m._last_workspace
    py, pn, pc = C.c_void_p(), C.c_void_p(), C.c_void_p()
    _lib.check(_lib.load().b3d_pose_debug_knn_ptrs(ws.data_ptr(), nbytes, N, E, m.depth, flags, C.byref(py), C.byref(pn),
                                                   C.byref(pc)), "knn ptrs")
    y = ws[py.value - ws.data_ptr():][:N * 48 * 4].view(torch.float32).view(N, 48).cpu()
    ts = d.node_timestamps
    for t in torch.unique(ts).tolist():
        idx = torch.nonzero(ts == t).squeeze(1)
        ei = ref_torch.knn_graph(x4[idx], 20)
        torch.testing.assert_close(y[idx], ora.knn_conv(x4[idx], ei), rtol=1e-4, atol=1e-5)


def test_training_step_captured_into_a_hip_graph_replays_exactly():
    """bench.py's timed region replays hipGraph-captured training steps.  A captured step (graph build, forward,
    fused loss, backward, Adam with the step counter on the device) replayed N times must leave the same
    parameters as N eager steps."""
    import copy
    from batch3dmot_amd.pose_gnn import PoseGNN
    from batch3dmot_amd.train_step import make_optimizer, train_step
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    a = PoseGNN().to(dev)
    b = copy.deepcopy(a)
    opt_a = make_optimizer(a, lr=1e-3, capturable=True)
    opt_b = make_optimizer(b, lr=1e-3, capturable=True)
    data = _tiny_batch(dev, 30)

    def step(m, opt):
        if hasattr(data, "_b3d_graph"):
            del data._b3d_graph
        return train_step(m, data, opt, logits=True)

    for _ in range(4):
        step(a, opt_a)
    step(b, opt_b)                                   # eager warm-up, then capture one step and replay it 3 times
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.graph(g, stream=s):
        step(b, opt_b)                               # capture does not execute
    torch.cuda.current_stream().wait_stream(s)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert int(opt_a.step_dev) == 4 and int(opt_b.step_dev) == 4
    for (n, p), (_, q) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(p, q), n


def test_split_capture_for_data_parallel_steps_replays_exactly():
    """bench.py at N > 1: forward..backward of every pool batch is one hipGraph, Adam another, the gradient all-reduce
    stays eager between the two replays.  Two batches, captured back to back, replayed alternately with an eager
    in-place operation on the flat gradient buffer in between (the all-reduce's place): same parameters as eager."""
    import copy
    from batch3dmot_amd.pose_gnn import PoseGNN
    from batch3dmot_amd.train_step import forward_backward, make_optimizer, train_step
    dev = torch.device("cuda:0")
    torch.manual_seed(12)
    a = PoseGNN().to(dev)
    b = copy.deepcopy(a)
    opt_a = make_optimizer(a, lr=1e-3, capturable=True)
    opt_b = make_optimizer(b, lr=1e-3, capturable=True)
    batches = [_tiny_batch(dev, 30), _tiny_batch(dev, 31)]

    class Halve:                                        # stands in for FlatGradSync: an eager op on the flat gradients
        def __init__(self, opt):
            self.opt = opt
        def sync(self):
            self.opt.flat_grad.mul_(0.5)

    def fresh(d):
        if hasattr(d, "_b3d_graph"):
            del d._b3d_graph

    for k in range(5):                                  # eager reference: 1 warm-up + 4 steps
        d = batches[k % 2]
        fresh(d)
        train_step(a, d, opt_a, logits=True, grad_sync=Halve(opt_a))
    fresh(batches[0])
    train_step(b, batches[0], opt_b, logits=True, grad_sync=Halve(opt_b))      # warm-up
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    graphs = []
    for d in batches:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
            fresh(d)
            forward_backward(b, d, opt_b, logits=True)
        graphs.append(g)
    g_opt = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g_opt, stream=s, capture_error_mode="thread_local"):
        opt_b.step()
    torch.cuda.current_stream().wait_stream(s)
    sync_b = Halve(opt_b)
    for k in range(1, 5):
        graphs[k % 2].replay()
        sync_b.sync()
        g_opt.replay()
    torch.cuda.synchronize()
    assert int(opt_a.step_dev) == 5 and int(opt_b.step_dev) == 5
    for (n, p), (_, q) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(p, q), n
