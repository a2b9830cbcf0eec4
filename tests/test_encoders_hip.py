"""``b3d_point_feat``: the point-wise conv-BN-ReLU stacks + max-pool of PointNet / RadarNet in
eval mode against the same modules on the PyTorch path (the restatement of models/pointnet.py / radarnet.py that the
golden CLR vectors pin, tests/test_oracle_golden.py)."""
import pytest
import torch

from oracle.seeded import seeded_fill_

pytestmark = pytest.mark.gpu
TOL = 1e-4


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _randomise_bn(m, seed):
    g = torch.Generator().manual_seed(seed)
    for mod in m.modules():
        if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm):
            mod.running_mean.copy_(torch.randn(mod.num_features, generator=g) * 0.1)
            mod.running_var.copy_(torch.rand(mod.num_features, generator=g) + 0.5)


def _clouds(b, c, p, seed, dev):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(b, c, p, generator=g)
    npts = torch.randint(6, p + 1, (b,), generator=g)
    x = x * (torch.arange(p)[None, None, :] < npts[:, None, None])       # zero-padded clouds, as the dataset stores them
    return x.to(dev)


@pytest.mark.parametrize("b", [1, 2, 7, 300, 2100])
def test_pointnet_forward_feat(b):
    from batch3dmot_amd import encoders
    dev = torch.device("cuda:0")
    m = encoders.PointNetClassifier(k=7)
    seeded_fill_(m, 3)
    _randomise_bn(m, 4)
    m = m.to(dev).eval()
    x = _clouds(b, 3, 128, 10 + b, dev)
    with torch.no_grad():
        got = m.forward_feat(x)
        for mod in m.modules():
            mod.use_hip = False
        want = m.forward_feat(x)
    assert got.shape == want.shape == (b, 256)
    assert rel(got, want) < TOL
    # the trunk alone (STN transform applied inside the kernel; no ReLU after the last layer)
    with torch.no_grad():
        want_feat = m.feat(x)
        for mod in m.modules():
            mod.use_hip = True
        got_feat = m.feat(x)
    assert rel(got_feat, want_feat) < TOL and float(got_feat.min()) < 0.0


@pytest.mark.parametrize("b", [1, 3, 500])
def test_radarnet_forward_feat(b):
    from batch3dmot_amd import encoders
    dev = torch.device("cuda:0")
    m = encoders.RadarNetClassifier(k=7)
    seeded_fill_(m, 5)
    _randomise_bn(m, 6)
    m = m.to(dev).eval()
    x = _clouds(b, 4, 64, 20 + b, dev)
    with torch.no_grad():
        got = m.forward_feat(x)
        for mod in m.modules():
            mod.use_hip = False
        want = m.forward_feat(x)
    assert rel(got, want) < TOL


def test_train_mode_keeps_the_batch_statistics_path_and_bad_shapes_are_rejected():
    from batch3dmot_amd import encoders
    dev = torch.device("cuda:0")
    m = encoders.RadarNetClassifier(k=7).to(dev).train()
    x = _clouds(4, 4, 64, 1, dev)
    with pytest.raises(RuntimeError, match="unfrozen"):                  # trainable parameters in train mode: no silent PyTorch path
        m.forward_feat(x)
    for mod in m.modules():
        mod.use_hip = False                                              # ... the deliberate one
    out = m.forward_feat(x)
    assert out.shape == (4, 256) and out.requires_grad
    with pytest.raises(RuntimeError):
        encoders.point_feat_hip((m.feat.conv1, m.feat.conv2, m.feat.conv3), (m.feat.bn1, m.feat.bn2, m.feat.bn3), x)
    m.eval()
    with pytest.raises(Exception):
        encoders.point_feat_hip((m.feat.conv1, m.feat.conv2, m.feat.conv3), (m.feat.bn1, m.feat.bn2, m.feat.bn3),
                                torch.zeros(2, 4, 100, device=dev))


def test_folded_batchnorm_paths_equal_the_reference_operation_order():
    """Eval mode folds BatchNorm into the conv / linear in front of it (MIOpen's inference BatchNorm is the slow part
    of the PyTorch path); ``reference_order_`` turns folding and the HIP kernels off -- what the CPU oracle runs."""
    import copy
    from batch3dmot_amd import encoders
    dev = torch.device("cuda:0")
    for ctor, shape in ((encoders.ResNetAE, (300, 3, 32, 32)), (lambda: encoders.PointNetClassifier(k=7), (64, 3, 128)),
                        (lambda: encoders.RadarNetClassifier(k=7), (64, 4, 64))):
        m = ctor()
        seeded_fill_(m, 8)
        _randomise_bn(m, 9)
        m = m.to(dev).eval()
        ref = encoders.reference_order_(copy.deepcopy(m))
        x = torch.rand(*shape, device=dev)
        with torch.no_grad():
            got = m.encode(x) if hasattr(m, "encode") else m.forward_feat(x)
            want = ref.encode(x) if hasattr(ref, "encode") else ref.forward_feat(x)
        assert rel(got, want) < 1e-5
    # the fold cache follows in-place updates of the statistics
    bn = m.bn1
    with torch.no_grad():
        a = m.forward_feat(x)
        bn.running_var.mul_(4.0)
        b = m.forward_feat(x)
        want = encoders.reference_order_(copy.deepcopy(m)).forward_feat(x)
    assert rel(b, want) < 1e-5 and rel(a, want) > 1e-3


@pytest.mark.parametrize("kind,b", [("pointnet", 300), ("pointnet", 3), ("radarnet", 200), ("radarnet", 5)])
def test_train_mode_with_frozen_parameters_uses_batch_statistics(kind, b):
    """How the GNN holds its encoders during training (clr_att_gnn.py:26-33: frozen, but in train mode): BatchNorm
    normalises with the statistics of the batch and updates its running statistics.  HIP path (last layer's
    statistics from per-cloud sums, max / min selected by the sign of the scale) against the PyTorch modules."""
    import copy
    from batch3dmot_amd import encoders
    dev = torch.device("cuda:0")
    m = encoders.PointNetClassifier(k=7) if kind == "pointnet" else encoders.RadarNetClassifier(k=7)
    seeded_fill_(m, 13)
    _randomise_bn(m, 14)
    with torch.no_grad():                                   # negative BatchNorm scales exercise the min branch
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.weight.mul_(torch.where(torch.arange(mod.num_features) % 3 == 0, -1.0, 1.0))
    for p in m.parameters():
        p.requires_grad = False
    m = m.to(dev).train()
    ref = copy.deepcopy(m)
    for mod in ref.modules():
        mod.use_hip = False
    x = _clouds(b, 3 if kind == "pointnet" else 4, 128 if kind == "pointnet" else 64, 40 + b, dev)
    if b < 16:
        # BatchNorm1d over a handful of rows in the fc heads is ill-conditioned (it amplifies 1e-6 input differences
        # by 1 / std of 3 samples): compare the point stack alone
        got, want = m.feat(x), ref.feat(x)
    else:
        torch.manual_seed(0); got = m.forward_feat(x)       # same dropout mask on both paths
        torch.manual_seed(0); want = ref.forward_feat(x)
    assert rel(got, want) < TOL
    for (n1, b1), (_, b2) in zip(m.feat.named_buffers(), ref.feat.named_buffers()):
        if b1.dtype.is_floating_point:
            assert rel(b1, b2) < TOL, n1
        else:
            assert torch.equal(b1, b2), n1                  # num_batches_tracked


@pytest.mark.parametrize("n", [1500, 37, 5])
def test_resnet_encode_train_mode_matches_the_pytorch_modules(n):
    """``ResNetAE.encode`` as the GNN runs it during training (frozen, train mode: batch-statistics BatchNorm, running
    statistics updated): the six HIP phase kernels against the PyTorch modules (MIOpen convolutions + BatchNorm)."""
    import copy
    from batch3dmot_amd import encoders
    dev = torch.device("cuda:0")
    m = encoders.ResNetAE()
    seeded_fill_(m, 21)
    _randomise_bn(m, 22)
    with torch.no_grad():                                   # negative scales too
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.mul_(torch.where(torch.arange(mod.num_features) % 4 == 0, -1.0, 1.0))
    for p in m.parameters():
        p.requires_grad = False
    m = m.to(dev).train()
    ref = copy.deepcopy(m)
    for mod in ref.modules():
        mod.use_hip = False
    g = torch.Generator().manual_seed(5)
    for step in range(2):                                   # two steps: the running statistics compound
        x = torch.rand(n, 3, 32, 32, generator=g).to(dev)
        with torch.no_grad():
            got, want = m.encode(x), ref.encode(x)
        assert got.shape == (n, 96)
        # n = 5: the last BatchNorms see 5 values per channel, their normalised output amplifies rounding differences
        assert rel(got, want) < (2e-5 if n > 5 else 1e-3), (step, rel(got, want))
    for (k, a), (_, b) in zip(m.state_dict().items(), ref.state_dict().items()):
        if "running" in k:
            assert rel(a, b) < 1e-5, k
        elif "num_batches_tracked" in k:
            assert int(a) == int(b), k
    # block 1/2/3 BatchNorms were exercised, the unused top-level `bn` was not
    assert int(m.res_block3.bn2.num_batches_tracked) == 2 and int(m.bn.num_batches_tracked) == 0


def test_resnet_encode_eval_mode_and_argument_checks():
    import copy
    from batch3dmot_amd import encoders, _lib
    dev = torch.device("cuda:0")
    m = encoders.ResNetAE()
    seeded_fill_(m, 23)
    _randomise_bn(m, 24)
    m = m.to(dev).eval()
    ref = encoders.reference_order_(copy.deepcopy(m))
    for n in (1, 65, 1000):
        x = torch.rand(n, 3, 32, 32, device=dev)
        with torch.no_grad():
            got, want = m.encode(x), ref.encode(x)
        assert rel(got, want) < 1e-5
    before = copy.deepcopy(m.state_dict())
    with torch.no_grad():
        m.encode(torch.rand(4, 3, 32, 32, device=dev))
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k]), k                 # eval mode leaves the statistics alone
    with pytest.raises(ValueError):
        encoders.resnet_encode_hip(m, torch.rand(4, 3, 16, 16, device=dev))
    lib = _lib.load()
    assert lib.b3d_resnet_encode(None, None, None, 4, 0, None, 0, None, None) != 0


@pytest.mark.parametrize("c,p,b,with_trans", [(3, 128, 257, True), (3, 128, 5, False), (4, 64, 130, False)])
def test_point_moments_and_bn_fold_match_float64(c, p, b, with_trans):
    """``b3d_point_moments`` (mean / second moments of a point stack's layer inputs, the 64 x 64 case accumulated by
    fp32 MFMA) and ``b3d_bn_fold_moments`` (batch statistics of the affine pre-activation from those moments, the
    running-statistics update and the fold) against the same quantities formed in float64 by PyTorch."""
    import ctypes as C
    from batch3dmot_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(31)
    x = (torch.randn(b, c, p, generator=g) * 2.0 + 0.5).to(dev)
    trans = (torch.eye(3).repeat(b, 1, 1) + 0.1 * torch.randn(b, 3, 3, generator=g)).to(dev) if with_trans else None
    w1 = (torch.randn(64, c, generator=g) * 0.5).to(dev)
    b1 = (torch.randn(64, generator=g) * 0.1).to(dev)
    xin = torch.bmm(x.transpose(2, 1), trans).transpose(2, 1) if with_trans else x
    pts = xin.permute(0, 2, 1).reshape(-1, c).double()                     # [n, c]
    n = pts.size(0)
    nbytes = lib.b3d_point_moments_workspace_bytes()
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    stream = _lib.current_stream(dev)
    tp = trans.contiguous().data_ptr() if with_trans else None

    def moments(fold):
        k = 64 if fold is not None else c
        mu = torch.empty(k, dtype=torch.float64, device=dev)
        sec = torch.empty(k, k, dtype=torch.float64, device=dev)
        _lib.check(lib.b3d_point_moments(C.byref(fold) if fold is not None else None, x.data_ptr(), tp, b, c, p, ws.data_ptr(), nbytes,
                                         mu.data_ptr(), sec.data_ptr(), stream), "b3d_point_moments")
        return mu, sec

    mu, sec = moments(None)
    assert rel(mu, pts.mean(0)) < 1e-6 and rel(sec, pts.t() @ pts / n) < 1e-6
    fold = _lib.b3d_linear()
    fold.w, fold.b = w1.data_ptr(), b1.data_ptr()
    mu2, sec2 = moments(fold)
    h1 = torch.relu(pts @ w1.double().t() + b1.double())
    assert rel(mu2, h1.mean(0)) < 2e-6 and rel(sec2, h1.t() @ h1 / n) < 2e-6
    # fold: a 64 -> 128 layer behind h1, BatchNorm with negative scales and tracked statistics
    w2 = (torch.randn(128, 64, generator=g) * 0.2).to(dev)
    b2 = (torch.randn(128, generator=g) * 0.1).to(dev)
    bn = torch.nn.BatchNorm1d(128).to(dev)
    with torch.no_grad():
        bn.weight.copy_(torch.randn(128, generator=g))
        bn.bias.copy_(torch.randn(128, generator=g))
    ref = torch.nn.BatchNorm1d(128).to(dev).double()
    ref.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in bn.state_dict().items()})
    z = h1 @ w2.double().t() + b2.double()
    ref.train()
    want = ref(z)                                                            # normalised pre-activation, float64
    wf = torch.empty_like(w2)
    bf = torch.empty_like(b2)
    _lib.check(lib.b3d_bn_fold_moments(mu2.data_ptr(), sec2.data_ptr(), 64, w2.data_ptr(), b2.data_ptr(), 128, bn.weight.data_ptr(),
                                       bn.bias.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                                       bn.num_batches_tracked.data_ptr(), 0.1, float(bn.eps), n, wf.data_ptr(), bf.data_ptr(), stream),
               "b3d_bn_fold_moments")
    got = h1 @ wf.double().t() + bf.double()
    assert rel(got, want) < 2e-5
    assert rel(bn.running_mean.double(), ref.running_mean) < 1e-5 and rel(bn.running_var.double(), ref.running_var) < 1e-5
    assert int(bn.num_batches_tracked) == 1
