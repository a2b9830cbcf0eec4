"""Checkpoint paths of the drop-in modules, as the reference's callers use them (no GPU): `load_state_dict(..., strict=False)`
with a partial checkpoint (train.py:73-74), the newer PyG name of the GATConv projection (`knn_conv.lin.weight`),
reference-ordered full checkpoints, and the encoders' loud routing (encoders.path_counts)."""
import pytest
import torch

from oracle import ref_encoders, ref_torch
from oracle.seeded import seeded_fill_


def _clr():
    from batch3dmot_amd import encoders
    from batch3dmot_amd.clr_att_gnn import GNN
    return GNN(encoders.ResNetAE(), encoders.PointNetClassifier(k=7), encoders.RadarNetClassifier(k=7))


def test_partial_checkpoint_loads_non_strict_as_train_py_does():
    """train.py:73-74: `gnn.load_state_dict(torch.load(...), strict=False)` -- a checkpoint that holds only some sub-modules
    (here: the message passing and the edge encoder, plus a key the model does not know) loads those and leaves the rest."""
    src = ref_torch.GNN(ref_encoders.ResNetAE(), ref_encoders.PointNetClassifier(k=7), ref_encoders.RadarNetClassifier(k=7))
    seeded_fill_(src, 5)
    full = src.state_dict()
    part = {k: v.clone() for k, v in full.items() if k.startswith("message_passing.") or k.startswith("edge_encoder.")}
    part["some_head_of_a_later_release.weight"] = torch.zeros(3)
    m = _clr()
    before = {k: v.clone() for k, v in m.state_dict().items()}
    res = m.load_state_dict(part, strict=False)
    assert res.unexpected_keys == ["some_head_of_a_later_release.weight"]
    assert "node_encoder.0.weight" in res.missing_keys and "resnet.conv.weight" in res.missing_keys
    assert not any(k.startswith("message_passing.") or k.startswith("edge_encoder.") for k in res.missing_keys)
    after = m.state_dict()
    for k in after:
        want = part[k] if k in part else before[k]
        assert torch.equal(after[k], want), k
    with pytest.raises(RuntimeError):
        _clr().load_state_dict(part, strict=True)


@pytest.mark.parametrize("kind", ["pose", "clr"])
def test_newer_pyg_lin_weight_name_is_accepted(kind):
    """PyG >= 2.3 stores GATConv's shared projection as `lin.weight`; 2.0.x (the reference's era) as `lin_src.weight` aliased by
    `lin_dst.weight`.  Both load, into the same parameter, strict."""
    if kind == "pose":
        from batch3dmot_amd.pose_gnn import PoseGNN
        make, d = PoseGNN, 48
    else:
        make, d = _clr, 96
    m = make()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    assert "knn_conv.lin_src.weight" in sd and "knn_conv.lin_dst.weight" in sd and tuple(sd["knn_conv.att_src"].shape) == (1, 1, d)
    w = torch.randn(d, d)
    new = {k: v for k, v in sd.items() if k not in ("knn_conv.lin_src.weight", "knn_conv.lin_dst.weight")}
    new["knn_conv.lin.weight"] = w
    m2 = make()
    res = m2.load_state_dict(new, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert torch.equal(m2.knn_conv.lin_src.weight, w) and m2.knn_conv.lin_dst is m2.knn_conv.lin_src
    old = dict(sd)
    old["knn_conv.lin_src.weight"] = old["knn_conv.lin_dst.weight"] = w
    m3 = make()
    m3.load_state_dict(old, strict=True)
    assert torch.equal(m3.knn_conv.lin_src.weight, w)
    # the alias is ONE parameter: it appears once among the parameters an optimizer would receive
    assert sum(1 for p in m3.parameters() if p is m3.knn_conv.lin_src.weight) == 1


def test_state_dict_round_trip_through_a_file(tmp_path):
    """predict.py:404 / train.py:73: torch.save(state_dict) -> torch.load -> load_state_dict, reference key set and order."""
    from batch3dmot_amd.pose_gnn import PoseGNN
    ora = ref_torch.PoseGNN()
    seeded_fill_(ora, 9)
    f = tmp_path / "gnn.pth"
    torch.save(ora.state_dict(), f)
    m = PoseGNN()
    m.load_state_dict(torch.load(f), strict=True)
    for (ka, va), (kb, vb) in zip(m.state_dict().items(), ora.state_dict().items()):
        assert ka == kb and torch.equal(va, vb)


def test_encoder_routing_is_counted_and_cpu_inputs_take_the_pytorch_modules():
    from batch3dmot_amd import encoders
    encoders.path_counts(reset=True)
    pn = encoders.PointNetClassifier(k=7).eval()
    pn.forward_feat(torch.randn(3, 3, 128))
    rn = encoders.RadarNetClassifier(k=7).eval()
    rn.forward_feat(torch.randn(2, 4, 64))
    encoders.ResNetAE().eval().encode(torch.rand(2, 3, 32, 32))
    took = encoders.path_counts(reset=True)
    assert took == {("stn.points", "torch"): 1, ("stn.fc", "torch"): 1, ("pointnet.points", "torch"): 1, ("pointnet.fc", "torch"): 1,
                    ("radarnet.points", "torch"): 1, ("radarnet.fc", "torch"): 1, ("resnet.encode", "torch"): 1}
    assert encoders.path_counts() == {}


class PyGLikeBatch:
    """What `torch_geometric.loader.DataLoader` hands to `forward(data)` (train.py:88-96, utils/graph_data.py:230-242), reduced to
    the behaviour the models rely on: tensors live in a private mapping and are served through `__getattr__` (as PyG's
    `Data._store` does), unknown attributes raise AttributeError, `.to(device)` returns a moved copy, `batch` / `ptr` /
    `num_graphs` exist.  torch_geometric itself is not installed here."""

    def __init__(self, **kw):
        object.__setattr__(self, "_store", dict(kw))

    def __getattr__(self, k):
        try:
            return object.__getattribute__(self, "_store")[k]
        except KeyError:
            raise AttributeError(k) from None

    def __setattr__(self, k, v):
        self._store[k] = v

    def to(self, device):
        return PyGLikeBatch(**{k: (v.to(device) if torch.is_tensor(v) else v) for k, v in self._store.items()})

    @property
    def num_graphs(self):
        return int(self._store["batch"].max()) + 1


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["pose", "clr"])
def test_a_pyg_like_batch_object_works_as_data(kind):
    from batch3dmot_amd import synth
    from batch3dmot_amd.pose_gnn import PoseGNN
    dev = torch.device("cuda:0")
    d = synth.make_batch(2, 60, 300, first_graph_idx=11, modalities=(kind == "clr"))
    fields = {k: getattr(d, k) for k in ("pose_feats", "edge_index", "edge_attr", "node_timestamps", "batch", "y", "edge_weights")}
    if kind == "clr":
        fields.update({k: getattr(d, k) for k in ("img_feats", "lidar_feats", "radar_feats")})
    pyg = PyGLikeBatch(**fields, ptr=torch.tensor([0, 60, 120]))
    m = (PoseGNN() if kind == "pose" else _clr()).to(dev).eval()
    with torch.no_grad():
        a = m(pyg.to(dev))
        b = m(d.to(dev))
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    # ... and through the training step the callers run (train.py:124-160)
    from batch3dmot_amd.train_step import make_optimizer, train_step
    m.train()
    loss, out, _ = train_step(m, pyg.to(dev), make_optimizer(m), batch_size=2, loss_kind="cb", logits=(kind == "pose"))
    assert torch.isfinite(loss).item() and out.shape == (d.edge_index.size(1), 1)
