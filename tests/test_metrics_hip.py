"""``b3d_average_precision`` against the oracle's numpy restatement of torchmetrics' binary average precision and
against sklearn (train.py:143-150): overall and per edge class, ties, degenerate sets."""
import math

import numpy as np
import pytest
import torch

from oracle.ref_torch import average_precision_np

CLASS_DICT = {"car": 1, "truck": 2, "bus": 3, "trailer": 4, "pedestrian": 5, "motorcycle": 6, "bicycle": 7}


def _case(n, seed, ties, pos_frac=0.05):
    g = torch.Generator().manual_seed(seed)
    s = torch.rand(n, generator=g)
    if ties:
        s = (s * 50).round() / 50                  # many equal scores: one curve point per distinct value
    y = (torch.rand(n, generator=g) < pos_frac).long()
    c = torch.randint(1, 8, (n,), generator=g).float()
    return s, y, c


def test_the_restatement_equals_sklearn():
    from sklearn.metrics import average_precision_score
    for seed, ties in [(0, False), (1, True), (2, True)]:
        s, y, _ = _case(5000, seed, ties)
        assert abs(average_precision_np(s.numpy(), y.numpy()) - average_precision_score(y.numpy(), s.numpy())) < 1e-12
    assert math.isnan(average_precision_np(np.array([0.3, 0.2]), np.array([0, 0])))


@pytest.mark.gpu
@pytest.mark.parametrize("n,seed,ties,ydtype", [(31000, 3, False, torch.float32), (31000, 4, True, torch.int64),
                                                (257, 5, True, torch.float32), (1, 6, False, torch.int64)])
def test_overall_and_per_class_match_the_oracle(n, seed, ties, ydtype):
    from batch3dmot_amd.metrics import average_precision, average_precision_per_class
    dev = torch.device("cuda:0")
    s, y, c = _case(n, seed, ties, pos_frac=0.3 if n < 1000 else 0.05)
    if n == 1:
        y[:] = 1
    ap = average_precision(s.to(dev), y.to(ydtype).to(dev))
    assert ap.dtype == torch.float64 and ap.is_cuda
    assert abs(float(ap) - average_precision_np(s.numpy(), y.numpy())) < 1e-12
    ap_all, per_class = average_precision_per_class(s.to(dev).unsqueeze(1), y.to(ydtype).to(dev), c.to(dev), CLASS_DICT)
    assert float(ap_all) == float(ap)                                   # bitwise: same sort, same sums
    for name, idx in CLASS_DICT.items():
        m = c == idx
        if int(m.sum()) == 0:
            assert name not in per_class                                # train.py:147 skips empty classes
            continue
        want = average_precision_np(s[m].numpy(), y[m].numpy())
        got = per_class[name]
        assert (math.isnan(want) and math.isnan(got)) or abs(got - want) < 1e-12, (name, got, want)


@pytest.mark.gpu
def test_degenerate_sets_and_argument_checks():
    from batch3dmot_amd.metrics import average_precision, average_precision_per_class
    dev = torch.device("cuda:0")
    s = torch.tensor([0.9, 0.9, 0.1, 0.5], device=dev)
    assert math.isnan(float(average_precision(s, torch.zeros(4, device=dev))))            # no positive: 0 / 0
    assert float(average_precision(s, torch.ones(4, device=dev))) == 1.0
    # all scores tied: one curve point, precision = prevalence
    assert abs(float(average_precision(torch.full((8,), 0.5, device=dev), torch.tensor([1, 0, 0, 0, 1, 0, 0, 0], device=dev))) - 0.25) < 1e-15
    ap_all, per_class = average_precision_per_class(s, torch.tensor([1, 0, 1, 0], device=dev),
                                                    torch.tensor([1.0, 1.0, 3.0, 0.0], device=dev), CLASS_DICT)
    assert set(per_class) == {"car", "bus"} and per_class["car"] == 0.5 and per_class["bus"] == 1.0
    with pytest.raises(ValueError):
        average_precision(s, torch.ones(3, device=dev))
    with pytest.raises(Exception):
        average_precision(s.cpu(), torch.ones(4))
    # bitwise reproducible
    big = torch.rand(100000, device=dev); yb = (torch.rand(100000, device=dev) < 0.1).float()
    assert float(average_precision(big, yb)) == float(average_precision(big, yb))
