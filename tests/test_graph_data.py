"""Window loader: files in the reference's on-disk format
(construct_detection_graphs_parallel.py:623-650) are read by ``graph_data.GraphDataset`` and by the oracle's
loop-for-loop restatement of ``GraphDataset.__getitem__`` (utils/graph_data.py:152-257); every field must match.
Then the batch iterator against a plain collate."""
import json
import types

import pytest
import torch

from batch3dmot_amd import synth
from batch3dmot_amd.data import collate
from batch3dmot_amd.graph_data import CLASS_DICT, REL_FREQ_TRAIN, GraphDataset, iterate_batches
from oracle.ref_torch import window_getitem_loop

INV_CLASS = {v: k for k, v in CLASS_DICT.items()}


def _write_window(stem, seed, n_per_frame=12, global_offset=1000):
    return synth.write_window_files(stem, seed, n_per_frame=n_per_frame, global_offset=global_offset)


@pytest.fixture()
def window_dir(tmp_path):
    d = str(tmp_path) + "/"
    scenes = [{"token": "sceneA", "nbr_samples": 8}, {"token": "sceneB", "nbr_samples": 7}]
    k = 0
    for sc in scenes:
        for b in range(int(sc["nbr_samples"]) - 5):
            _write_window(d + f"{sc['token']}_len5_{b}", seed=10 + k)
            k += 1
    return d, scenes


def _same(a, b):
    if a.dtype.is_floating_point:
        torch.testing.assert_close(a, b, rtol=1e-6, atol=0)
    else:
        assert torch.equal(a, b)
    assert a.dtype == b.dtype and a.shape == b.shape


@pytest.mark.parametrize("inference", [False, True])
def test_every_field_equals_the_reference_loops(window_dir, inference):
    d, scenes = window_dir
    params = types.SimpleNamespace(main=types.SimpleNamespace(slice_factor=1, class_dict="nuscenes_tracking_eval"),
                                   gnn=types.SimpleNamespace(batch_size_graph=5),
                                   classes=types.SimpleNamespace(nuscenes_tracking_eval=CLASS_DICT))
    ds = GraphDataset(params, scenes, d, 5, inference)
    assert len(ds) == 3 + 2 and ds.batches[0] == d + "sceneA_len5_0" and ds.batches[-1] == d + "sceneB_len5_1"
    assert ds.get_metadata() == (ds.batches, scenes)
    for idx in range(len(ds)):
        got = ds[idx]
        ref = window_getitem_loop(ds.batches[idx], inference, REL_FREQ_TRAIN, CLASS_DICT)
        if inference:
            (got, got_meta), (ref, ref_meta) = got, ref
            assert got_meta == ref_meta
        for key, val in ref.items():
            if torch.is_tensor(val):
                _same(getattr(got, key), val)
            else:
                assert getattr(got, key) == val, key
        assert got.batch_idx == idx
        assert float(got.node_classes[-1]) == 0.0                   # the isolated node keeps class 0, as in the loop
        for e in range(0, got.edge_index.size(1), 41):
            name = INV_CLASS[int(got.edge_classes[e])]
            assert abs(float(got.edge_weights[e]) - ds.cb_scaling_factor(name)) < 1e-6


def test_params_may_be_a_dict_or_absent_and_modalities_can_be_skipped(window_dir):
    d, scenes = window_dir
    a = GraphDataset(None, scenes, d, 5, False, modalities=())
    b = GraphDataset({"main": {"slice_factor": 2}, "gnn": {"batch_size_graph": 5}}, scenes, d, 5, False)
    assert len(a) == 5 and len(b) == 3                            # every second scene
    w = a[1]
    assert w.img_feats is None and w.lidar_feats is None and w.radar_feats is None and w.pose_feats.size(1) == 19


@pytest.mark.parametrize("shuffle", [False, True])
def test_batch_iterator_yields_the_collated_windows(window_dir, shuffle):
    d, scenes = window_dir
    ds = GraphDataset(None, scenes, d, 5, False)
    gen = torch.Generator().manual_seed(3)
    order = torch.randperm(len(ds), generator=torch.Generator().manual_seed(3)).tolist() if shuffle else list(range(len(ds)))
    batches = list(iterate_batches(ds, 2, shuffle=shuffle, generator=gen, prefetch=2))
    assert len(batches) == 3
    for k, b in enumerate(batches):
        want = collate([ds[i] for i in order[2 * k: 2 * k + 2]])
        for key in ("pose_feats", "edge_index", "edge_attr", "y", "edge_weights", "node_timestamps", "batch"):
            assert torch.equal(getattr(b, key), getattr(want, key)), key
    assert len(list(iterate_batches(ds, 2, drop_last=True))) == 2


def test_loader_errors_surface_in_the_consumer(window_dir):
    d, scenes = window_dir
    ds = GraphDataset(None, [{"token": "missing", "nbr_samples": 6}], d, 5, False)
    with pytest.raises(FileNotFoundError):
        list(iterate_batches(ds, 1))


@pytest.mark.gpu
def test_prefetched_batches_on_the_gpu_feed_the_model(window_dir):
    """Pinned + copy-stream delivery: tensors arrive on the device bit-identical, and the path consumes them."""
    from batch3dmot_amd.pose_gnn import PoseGNN
    d, scenes = window_dir
    ds = GraphDataset(None, scenes, d, 5, False, modalities=())
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = PoseGNN().to(dev).eval()
    outs = []
    for b in iterate_batches(ds, 2, device=dev, prefetch=2):
        assert b.pose_feats.is_cuda and b.edge_index.is_cuda
        with torch.no_grad():
            outs.append(m(b)[0].cpu())
    for k, o in enumerate(outs):
        want = collate([ds[i] for i in range(2 * k, min(2 * k + 2, len(ds)))]).to(dev)
        with torch.no_grad():
            assert torch.equal(m(want)[0].cpu(), o)


@pytest.mark.gpu
def test_prefetched_batches_stay_intact_under_a_long_running_consumer(window_dir):
    """The worker allocates batch k+1 on its copy stream while the consumer's kernels on batch k may still be queued:
    the consumer's stream is recorded on every tensor, so the allocator cannot hand batch k's blocks to the next
    copy early.  A long-running kernel between taking a batch and reading it makes the race window wide."""
    d, scenes = window_dir
    ds = GraphDataset(None, scenes, d, 5, False, modalities=())
    dev = torch.device("cuda:0")
    want = [collate([ds[i] for i in range(2 * k, min(2 * k + 2, len(ds)))]) for k in range(3)]
    spin = torch.randn(4096, 4096, device=dev)
    sums = []
    for rep in range(3):
        for b in iterate_batches(ds, 2, device=dev, prefetch=2):
            for _ in range(6):
                spin = (spin @ spin).clamp_(-1.0, 1.0)            # ~tens of ms of queued device work
            sums.append((b.pose_feats.clone(), b.edge_attr.clone(), b.edge_index.clone()))   # read BEHIND the queued work
            del b                                                  # blocks go back to the allocator while work is queued
    torch.cuda.synchronize()
    for i, (pf, ea, ei) in enumerate(sums):
        w = want[i % 3]
        assert torch.equal(pf.cpu(), w.pose_feats) and torch.equal(ea.cpu(), w.edge_attr) and torch.equal(ei.cpu(), w.edge_index), i


def _restore_windows(g, d):
    """Write the fixture's window files back to disk in the reference's layout."""
    for b, files in enumerate(g["windows"]):
        stem = d + f"{g['scenes'][0]['token']}_len5_{b}"
        for sfx, v in files.items():
            if sfx.endswith(".json"):
                with open(stem + sfx, "w") as fh:
                    json.dump(v, fh)
            else:
                torch.save(v, stem + sfx)


@pytest.mark.parametrize("inference", [False, True])
def test_loader_matches_the_reference_class_fixture(tmp_path, inference):
    """g7: what the REFERENCE's own GraphDataset.__getitem__ returned for these window files (oracle/make_golden.py,
    golden_loader: the class is taken from utils/graph_data.py by ast and executed).  Both the product loader and the
    oracle's loop-for-loop restatement must return exactly that."""
    from conftest import load_golden
    g = load_golden("g7_loader.pt")
    d = str(tmp_path) + "/"
    _restore_windows(g, d)
    ds = GraphDataset(None, g["scenes"], d, 5, inference)
    assert len(ds) == len(g["windows"])
    for idx in range(len(ds)):
        want = g["inference" if inference else "train"][idx]
        got = ds[idx]
        ref = window_getitem_loop(ds.batches[idx], inference, REL_FREQ_TRAIN, CLASS_DICT)
        if inference:
            (got, got_meta), (ref, ref_meta) = got, ref
            assert got_meta == want["global_node_metadata_str"] == ref_meta
        for k, v in want.items():
            if torch.is_tensor(v):
                _same(getattr(got, k), v)
                _same(ref[k], v)
            elif k in ("num_nodes", "batch_idx"):
                assert getattr(got, k) == v
        for k, sfx in want["passed_through"].items():
            assert torch.equal(getattr(got, k), g["windows"][idx][sfx]), k
