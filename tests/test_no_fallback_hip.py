"""No silent fallback (GPU): every stage of the frozen encoders, the loss and the optimizer of a product step run in the HIP
library -- asserted through the routing counters of `batch3dmot_amd.encoders` / `train_step` and the library's own launch
counters (`b3d_prof_read`) -- and a GPU input the HIP path cannot take raises instead of quietly running MIOpen / rocBLAS."""
import pytest
import torch

from oracle.seeded import seeded_fill_

pytestmark = pytest.mark.gpu

STAGES = {"stn.points", "stn.fc", "pointnet.points", "pointnet.fc", "radarnet.points", "radarnet.fc", "resnet.encode"}


def _model(dev, salt=3):
    from batch3dmot_amd import encoders
    from batch3dmot_amd.clr_att_gnn import GNN
    m = GNN(encoders.ResNetAE(), encoders.PointNetClassifier(k=7), encoders.RadarNetClassifier(k=7))
    seeded_fill_(m, salt)
    return m.to(dev)


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_every_encoder_stage_loss_and_optimizer_run_in_the_library(mode):
    from batch3dmot_amd import _lib, encoders, synth, train_step as ts
    dev = torch.device("cuda:0")
    data = synth.make_graph(200, None, k=6, graph_idx=77, modalities=True).to(dev)
    m = _model(dev)
    encoders.path_counts(reset=True)
    ts.PATHS.clear()
    _lib.prof_enable(True)
    try:
        if mode == "eval":
            m.eval()
            with torch.no_grad():
                m(data)
        else:
            m.train()
            opt = ts.make_optimizer(m)
            assert hasattr(opt, "flat_grad")                     # optim.FlatAdam (b3d_adam_step), not torch.optim.Adam
            ts.train_step(m, data, opt, batch_size=1, loss_kind="cb", logits=False)
        torch.cuda.synchronize()
        fam = _lib.prof_read()
    finally:
        _lib.prof_enable(False)
    took = encoders.path_counts()
    assert {s for (s, kind) in took if kind == "hip"} == STAGES, took
    assert not [k for k in took if k[1] == "torch"], took
    assert fam["point_feat"][1] >= 3 and fam["mp_edge_fwd"][1] == 6 and fam["att_fwd"][1] >= 1, fam
    if mode == "train":
        assert fam["mp_edge_bwd"][1] == 6 and fam["wgrad_edge"][1] >= 1, fam
        assert ts.PATHS == {"fused_loss": 1, "flat_adam": 1}, dict(ts.PATHS)


def test_gpu_inputs_the_hip_encoders_cannot_take_raise():
    from batch3dmot_amd import encoders
    dev = torch.device("cuda:0")
    rn = encoders.ResNetAE().to(dev).eval()
    with pytest.raises(RuntimeError, match="3, 32, 32"):
        rn.encode(torch.rand(4, 3, 16, 16, device=dev))           # not the GNN's crop size
    pn = encoders.PointNetClassifier(k=7).to(dev).eval()
    x = torch.randn(5, 3, 128, device=dev)
    with pytest.raises(RuntimeError, match="autograd"):
        pn.forward_feat(x)                                        # unfrozen parameters, grad mode on
    with torch.no_grad():
        pn.forward_feat(x)                                        # ... fine without autograd
    pn.train()
    with pytest.raises(RuntimeError, match="unfrozen"):
        with torch.no_grad():
            pn.forward_feat(x)                                    # train-mode statistics: frozen encoders only
    # the deliberate opt-out runs the PyTorch modules and is counted as such
    for mod in pn.modules():
        mod.use_hip = False
    encoders.path_counts(reset=True)
    y = pn.forward_feat(x)
    assert y.requires_grad and all(kind == "torch" for (_, kind) in encoders.path_counts()), encoders.path_counts()


def test_make_optimizer_refuses_a_silent_torch_adam():
    from batch3dmot_amd import train_step as ts
    dev = torch.device("cuda:0")
    m = _model(dev)
    m.edge_encoder[0].weight.requires_grad = False                # one Linear of the HIP backward frozen
    with pytest.raises(RuntimeError, match="flat=False"):
        ts.make_optimizer(m)
    assert isinstance(ts.make_optimizer(m, flat=False), torch.optim.Adam)
