"""EmbeddingCache (clr_att_gnn.py) as a sorted-id table: the table logic on the CPU -- lookups, merging
of new ids, first-occurrence gathering over a scene's windows, per-window row compaction -- with the encoders' PyTorch mode
and a torch stand-in for the one HIP call (`modality_present`).  The GPU tests hold the same properties on the HIP path."""
import torch

from batch3dmot_amd import clr_att_gnn, encoders, synth


def _torch_present(feats):
    return feats.reshape(feats.size(0), -1).sum(1) != 0


def test_cache_tables_match_direct_encoding(monkeypatch):
    monkeypatch.setattr(clr_att_gnn, "modality_present", _torch_present)
    torch.manual_seed(3)
    m = clr_att_gnn.GNN(encoders.ResNetAE(), encoders.PointNetClassifier(k=7), encoders.RadarNetClassifier(k=7)).eval()
    scene, wins = synth.make_scene(frames=7, per_frame=12, k=6, scene_idx=1)
    for w in wins:
        w.global_ids = w.global_ids * 3 + 11                      # sparse, non-contiguous ids
    cache = clr_att_gnn.EmbeddingCache()
    cache.add_windows(m, wins)
    n = scene.pose_feats.size(0)
    assert len(cache) == n and cache.misses == n and cache.encoder_rows["img"] == n
    assert torch.equal(cache.ids, torch.arange(n) * 3 + 11)
    enc = cache.windows([w.global_ids for w in wins])
    with torch.no_grad():
        for w, e in zip(wins, enc):
            has_l, has_r = _torch_present(w.lidar_feats), _torch_present(w.radar_feats)
            li, ri = torch.nonzero(has_l).squeeze(1), torch.nonzero(has_r).squeeze(1)
            assert torch.equal(e[2].long(), li) and torch.equal(e[4].long(), ri)
            torch.testing.assert_close(e[0], m.resnet.encode(w.img_feats).float(), rtol=1e-5, atol=1e-6)
            torch.testing.assert_close(e[1], m.pointnet.forward_feat(w.lidar_feats[li].view(-1, 3, 128)).float(), rtol=1e-4, atol=1e-5)
            torch.testing.assert_close(e[3], m.radarnet.forward_feat(w.radar_feats[ri].view(-1, 4, 64)).float(), rtol=1e-4, atol=1e-5)
    # incremental use: window by window, nothing is encoded twice; an unknown id is an error
    c2 = clr_att_gnn.EmbeddingCache()
    for w in wins:
        got = m.encode_modalities(w, cache=c2)
        ref = cache.window(w.global_ids)
        for a, b in zip(got, ref):
            torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5)
    assert len(c2) == n and c2.misses == n and c2.hits == sum(w.pose_feats.size(0) for w in wins) - n
    try:
        cache.window(torch.tensor([5]))
        raise AssertionError("expected KeyError")
    except KeyError:
        pass
