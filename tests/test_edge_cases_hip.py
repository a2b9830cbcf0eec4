"""GPU: ragged and degenerate graphs through the HIP PoseGNN path against the CPU oracle -- the inputs
the reference's callers can produce (predict.py:172-259 runs every window of a scene, however small):
isolated nodes, sources-only / sinks-only nodes, duplicate edges, self loops, a hub with hundreds of
incident edges, graphs smaller than one 16-row tile, every gnn_depth down to 1."""
import pytest
import torch

from conftest import assert_grad_close
from oracle import ref_torch
from oracle.seeded import seeded_fill_

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _data(n, edge_index, seed, frames=3):
    from batch3dmot_amd.data import Data
    g = torch.Generator().manual_seed(seed)
    e = edge_index.size(1)
    pose = torch.randn(n, 19, generator=g)
    attr = torch.randn(e, 4, generator=g, dtype=torch.float64)
    ts = torch.randint(0, frames, (n,), generator=g)
    return Data(pose_feats=pose, edge_index=edge_index.contiguous(), edge_attr=attr, node_timestamps=ts,
                y=torch.zeros(e), edge_weights=torch.ones(e), batch=None)


def _check(data, depth=6, seed=3, dead_knn=True):
    from batch3dmot_amd.pose_gnn import PoseGNN
    dev = torch.device("cuda:0")
    ora = ref_torch.PoseGNN(gnn_depth=depth, run_dead_knn=False)
    seeded_fill_(ora, seed)
    m = PoseGNN(gnn_depth=depth).to(dev)
    m.load_state_dict(ora.state_dict())
    m.run_dead_knn = dead_knn
    g = torch.Generator().manual_seed(99)
    e, n = data.edge_index.size(1), data.pose_feats.size(0)
    c_out, c_x = torch.randn(e, 1, generator=g), torch.randn(n, 48, generator=g)
    ro, rx = ora(data)
    ((ro * c_out).sum() + (rx * c_x).sum()).backward()
    go, gx = m(data.to(dev))
    ((go * c_out.to(dev)).sum() + (gx * c_x.to(dev)).sum()).backward()
    assert _rel(go, ro) < TOL and _rel(gx, rx) < TOL
    for (name, p), (_, q) in zip(m.named_parameters(), ora.named_parameters()):
        if q.grad is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        assert_grad_close(p.grad, q.grad, name, tol=5 * TOL)


def test_isolated_source_only_and_sink_only_nodes():
    # node 0: isolated; 1, 2: sources only; 7: sink only; duplicates of edge (1 -> 7); 23 nodes (< 2 tiles)
    src = torch.tensor([1, 1, 1, 2, 3, 4, 5, 5, 6, 9, 10, 12, 20, 21, 22])
    dst = torch.tensor([7, 7, 7, 7, 4, 5, 6, 8, 8, 11, 11, 13, 21, 22, 20])
    order = torch.argsort(dst, stable=True)
    _check(_data(23, torch.stack([src[order], dst[order]]), 1))


def test_self_loops_and_unsorted_edges():
    src = torch.tensor([4, 0, 3, 3, 2, 1, 4, 0])
    dst = torch.tensor([4, 0, 1, 3, 0, 2, 1, 4])        # two self loops, destination order shuffled
    _check(_data(5, torch.stack([src, dst]), 2))


@pytest.mark.parametrize("n,e", [(1, 1), (2, 1), (3, 17), (17, 16), (16, 129)])
def test_tiny_graphs_around_tile_boundaries(n, e):
    g = torch.Generator().manual_seed(n * 1000 + e)
    ei = torch.stack([torch.randint(0, n, (e,), generator=g), torch.sort(torch.randint(0, n, (e,), generator=g)).values])
    _check(_data(n, ei, 3))


def test_hub_node_with_hundreds_of_incident_edges():
    n = 400
    g = torch.Generator().manual_seed(5)
    hub_in = torch.stack([torch.arange(1, 301), torch.zeros(300, dtype=torch.long)])          # 300 edges into node 0
    hub_out = torch.stack([torch.zeros(250, dtype=torch.long), torch.arange(100, 350)])        # 250 edges out of node 0
    rest = torch.stack([torch.randint(1, n, (500,), generator=g), torch.randint(1, n, (500,), generator=g)])
    ei = torch.cat([hub_in, hub_out, rest], 1)
    ei = ei[:, torch.argsort(ei[1], stable=True)]
    _check(_data(n, ei, 4))


@pytest.mark.parametrize("depth", [1, 2, 3, 9])
def test_every_depth(depth):
    from batch3dmot_amd import synth
    d = synth.make_graph(100, None, k=5, graph_idx=900 + depth)
    _check(d, depth=depth, seed=10 + depth)


def test_single_frame_and_frames_smaller_than_k():
    """k-NN block (executed, discarded) on frames of 1, 2 and 21 nodes next to a larger one."""
    from batch3dmot_amd import synth
    d = synth.make_graph(200, None, k=5, graph_idx=950)
    n = d.pose_feats.size(0)
    ts = torch.zeros(n, dtype=torch.long)
    ts[0] = 5; ts[1:3] = 6; ts[3:24] = 7
    d.node_timestamps = ts
    _check(d, dead_knn=True)


@pytest.mark.gpu
def test_several_tiles_per_workgroup_give_the_same_bits():
    """Row-tiled kernels grid-stride once an input exceeds 2,048 tiles (262,144 edges); the weights of the hoisted edge
    kernels then stay resident in LDS across the tiles of a workgroup.  B3D_GRID_CAP forces that path at test size:
    outputs and gradients must be bitwise those of the one-tile-per-workgroup launch."""
    import copy, os
    from batch3dmot_amd import synth
    from batch3dmot_amd.pose_gnn import PoseGNN
    dev = torch.device("cuda:0")
    torch.manual_seed(21)
    a = PoseGNN().to(dev)
    b = copy.deepcopy(a)
    d = synth.make_graph(900, None, k=12, graph_idx=321).to(dev)
    assert d.edge_index.size(1) > 5 * 3 * 128                    # > 5 tiles per workgroup at cap 3
    lw = torch.randn(d.edge_index.size(1), 1, device=dev)

    def run(m):
        if hasattr(d, "_b3d_graph"):
            del d._b3d_graph
        out, x = m(d)
        (out * lw).sum().backward()
        return out.detach(), x.detach()

    ref = run(a)
    os.environ["B3D_GRID_CAP"] = "3"
    try:
        got = run(b)
    finally:
        del os.environ["B3D_GRID_CAP"]
    assert torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1])
    for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        if p.grad is not None:
            assert torch.equal(p.grad, q.grad), n
