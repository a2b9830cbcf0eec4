import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    return torch.load(os.path.join(GOLDEN, name), weights_only=False)


def data_from(d):
    from batch3dmot_amd.data import Data
    return Data(**d)


def assert_grad_close(have, want, name, tol=5e-4, flip_l2=4.0, flip_max=3e-2):
    """Gradient of one parameter against the fp32 CPU oracle on a SMALL graph.

    Two correct fp32 evaluations of these networks differ by more than rounding in isolated places: a ReLU
    pre-activation within a few ulps of zero takes the other branch (the models evaluate millions of units per
    forward; the window is ~1e-7 of the unit's scale).  One flipped unit changes ONE edge's (or node's) contribution
    to the gradients of its own layer and of everything upstream -- about 1/(number of edges) of a gradient entry,
    i.e. ~1e-3 on the few-hundred-edge graphs of these tests.  So: max-norm error below `tol` -- or, where a flip
    shows, an L2-relative error below `flip_l2 * tol` and a max-norm error below `flip_max` (an error of the kernels
    moves every row by more than that; the reference-generated golden fixtures and the float64 comparisons at the
    benchmark size hold the tight bounds)."""
    import torch
    a, b = have.detach().double().cpu(), want.detach().double().cpu()
    scale = float(b.abs().max().clamp_min(1e-30))
    mx = float((a - b).abs().max()) / scale
    if mx < tol:
        return
    l2 = float((a - b).norm() / b.norm().clamp_min(1e-30))
    assert l2 < flip_l2 * tol and mx < flip_max, (name, mx, l2)
