"""CPU, world_size 2 over gloo: flat gradient all-reduce == single-process gradient on the
concatenated batch, with the oracle standing in for the model."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))



def _free_port():
    """A TCP port nobody holds right now (bound to port 0, read back, released): two suites on one machine cannot collide the
    way a pid-derived port can."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]

def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from batch3dmot_amd import synth
    from batch3dmot_amd.dist import FlatGradSync, shard
    from batch3dmot_amd.train_step import edge_loss
    from oracle import ref_torch
    from oracle.seeded import seeded_fill_
    torch.set_num_threads(2)
    m = ref_torch.PoseGNN(run_dead_knn=False)
    seeded_fill_(m, 9)
    graphs = [synth.make_graph(80, None, k=5, graph_idx=400 + i) for i in range(4)]
    mine = shard(graphs, rank, world)
    assert len(mine) == 2
    from batch3dmot_amd.data import collate
    b = collate(mine)
    out, _ = m(b)
    loss = edge_loss(out, b, batch_size=len(mine), logits=True)
    loss.backward()
    sync = FlatGradSync(m.parameters())
    sync.sync()
    ret[rank] = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    # flat-optimizer fast path (optim.FlatAdam keeps its gradients in one buffer): reduced in place,
    # its parameters left out of the packed exchange
    import types
    flat_params = list(m.message_passing.parameters())
    buf = torch.full((7,), float(rank + 1))
    stub = types.SimpleNamespace(params=flat_params, fresh=False, flat_grad=buf)
    before = {id(p): p.grad.clone() for p in flat_params}
    FlatGradSync(m.parameters(), flat=stub).sync()
    assert torch.equal(buf, torch.full((7,), 1.5))
    assert all(torch.equal(p.grad, before[id(p)]) for p in flat_params)
    dist.destroy_process_group()


def test_flat_allreduce_equals_single_process_gradient():
    sys.path.insert(0, ROOT)
    from batch3dmot_amd import synth
    from batch3dmot_amd.data import collate
    from batch3dmot_amd.train_step import edge_loss
    from oracle import ref_torch
    from oracle.seeded import seeded_fill_
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert set(ret.keys()) == {0, 1}
    for n in ret[0]:
        assert torch.equal(ret[0][n], ret[1][n])            # every rank holds the same averaged gradient
    assert not any(k.startswith("knn_conv") for k in ret[0])
    # single process: mean over ranks of per-rank (mean-over-edges / local batch size) losses
    m = ref_torch.PoseGNN(run_dead_knn=False)
    seeded_fill_(m, 9)
    graphs = [synth.make_graph(80, None, k=5, graph_idx=400 + i) for i in range(4)]
    total = 0
    for r in range(2):
        b = collate(graphs[r::2])
        out, _ = m(b)
        total = total + edge_loss(out, b, batch_size=2, logits=True) / 2
    total.backward()
    for n, p in m.named_parameters():
        if p.grad is None:
            continue
        torch.testing.assert_close(ret[0][n], p.grad, rtol=1e-5, atol=1e-7)
