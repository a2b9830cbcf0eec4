"""H2 end to end (predict.py:172-259): overlapping windows of one scene -> model scores per window
-> mean over windows -> per-class threshold -> best predecessor / successor per node.  The fixtures were produced by
the REFERENCE model and the reference's own aggregate_node_flux / greedy_filter_node_flux (oracle/make_golden.py,
golden_scene); mean scores keep >= 1e-4 distance from their thresholds and from competing arg-maxima, so the kept-edge
set and the arg-max indices must be IDENTICAL."""
import pytest
import torch

from conftest import data_from, load_golden
from oracle.seeded import seeded_fill_


def _windows(g):
    from batch3dmot_amd import synth
    return synth.scene_windows(data_from(g["scene"]), g["frames"], g["per_frame"])


def _calibrate(model, g):
    with torch.no_grad():
        last = model.edge_classifier[6]
        last.weight *= g["gain"]
        last.bias *= g["gain"]
        last.bias += g["bias_shift"]
    return model


def _flatten(wins, scores):
    pairs = torch.cat([torch.stack([w.global_ids[w.edge_index[0].cpu()], w.global_ids[w.edge_index[1].cpu()]], 1) for w in wins])
    return pairs, torch.cat([s.reshape(-1).float().cpu() for s in scores])


def _same_indices(r, g):
    kp = r["kept_pairs"].cpu()
    order = torch.argsort(kp[:, 0] * 10 ** 6 + kp[:, 1])
    assert torch.equal(kp[order], g["kept_pairs"])                                 # identical thresholded edge set
    torch.testing.assert_close(r["kept_scores"].cpu()[order], g["kept_scores"], rtol=0, atol=2e-5)
    assert torch.equal(r["pred"].cpu(), g["pred"]) and torch.equal(r["succ"].cpu(), g["succ"])   # identical arg-max indices


@pytest.mark.parametrize("name", ["g6_scene_pose.pt", "g6_scene_clr.pt"])
def test_oracle_and_torch_post_processing_reproduce_the_reference_indices(name):
    from batch3dmot_amd import encoders
    from batch3dmot_amd.predict_post import greedy_edges
    from oracle import ref_torch
    g = load_golden(name)
    assert min(g["margins"]) > 1e-4
    if g["kind"] == "clr":
        m = ref_torch.GNN(encoders.ResNetAE(), encoders.PointNetClassifier(k=7), encoders.RadarNetClassifier(k=7), run_dead_knn=False)
    else:
        m = ref_torch.PoseGNN(run_dead_knn=False)
    seeded_fill_(m, g["salt"])
    _calibrate(m.eval(), g)
    wins = _windows(g)
    with torch.no_grad():
        scores = [m(w)[0] for w in wins]
    for s, ref in zip(scores, g["scores"]):
        assert float((s.reshape(-1) - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    pairs, sc = _flatten(wins, scores)
    _same_indices(greedy_edges(pairs, sc, g["node_cls"], g["class_names"], g["thresholds"]), g)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["g6_scene_pose.pt", "g6_scene_clr.pt"])
def test_hip_model_and_hip_post_processing_reproduce_the_reference_indices(name):
    from batch3dmot_amd import encoders
    from batch3dmot_amd.clr_att_gnn import GNN
    from batch3dmot_amd.pose_gnn import PoseGNN
    from batch3dmot_amd.predict_post import greedy_edges_hip
    dev = torch.device("cuda:0")
    g = load_golden(name)
    if g["kind"] == "clr":
        m = GNN(encoders.ResNetAE(), encoders.PointNetClassifier(k=7), encoders.RadarNetClassifier(k=7))
    else:
        m = PoseGNN()
    seeded_fill_(m, g["salt"])
    m = _calibrate(m, g).to(dev).eval()
    wins = _windows(g)
    scores = []
    with torch.no_grad():
        for w, ref in zip(wins, g["scores"]):
            s = m(w.to(dev))[0]
            assert float((s.reshape(-1).cpu() - ref).abs().max()) <= 1e-4 * float(ref.abs().max())    # north_star: 1e-4 on features
            scores.append(s)
    pairs, sc = _flatten(wins, scores)
    _same_indices(greedy_edges_hip(pairs.to(dev), sc.to(dev), g["node_cls"].to(dev), g["class_names"], g["thresholds"]), g)
