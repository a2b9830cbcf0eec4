"""Loader-side class-balanced edge weights: the vectorised form against the
oracle's loop-for-loop restatement of utils/graph_data.py:194-228, and against the constants the
synthetic generator uses."""
import pytest
import torch

from batch3dmot_amd import synth
from batch3dmot_amd.data import class_balanced_edge_weights
from oracle.ref_torch import edge_weights_loop


def _window(seed):
    g = torch.Generator().manual_seed(seed)
    n = 60
    cls = torch.randint(1, 8, (n,), generator=g)
    src, dst = [], []
    for _ in range(400):
        a, b = torch.randint(0, n, (2,), generator=g).tolist()
        if cls[a] == cls[b] and a != b:
            src.append(a); dst.append(b)
    return n, cls, torch.tensor([src, dst])


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_vectorised_weights_equal_the_reference_loop(seed):
    n, cls, ei = _window(seed)
    class_dict = {c: i + 1 for i, c in enumerate(synth.CLASSES)}
    names = [synth.CLASSES[int(c) - 1] for c in cls]
    w_ref, ec_ref, nc_ref = edge_weights_loop(ei.t(), names, synth.REL_FREQ_TRAIN, class_dict, n)
    freq = torch.zeros(8, dtype=torch.float64)
    for c, i in class_dict.items():
        freq[i] = synth.REL_FREQ_TRAIN[c]
    w, ec, nc = class_balanced_edge_weights(ei, cls, freq)
    torch.testing.assert_close(w, w_ref, rtol=1e-6, atol=0)
    assert torch.equal(ec, ec_ref) and torch.equal(nc, nc_ref)
    # the generator's per-class constants are the same factors
    for e in range(0, ei.size(1), 37):
        assert abs(float(w[e]) - synth.cb_scaling_factor(names[int(ei[0, e])])) < 1e-6


def test_mixed_class_edge_is_rejected_like_the_reference():
    cls = torch.tensor([1, 2, 2])
    with pytest.raises(ValueError):
        class_balanced_edge_weights(torch.tensor([[0], [1]]), cls, torch.ones(8, dtype=torch.float64))
    with pytest.raises(ValueError):
        edge_weights_loop(torch.tensor([[0, 1]]), ["car", "bus", "bus"], synth.REL_FREQ_TRAIN, {"car": 1, "bus": 3}, 3)
