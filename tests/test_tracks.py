"""hierarchical track clustering (predict.py:262-375).  The fixture holds greedy edges of
synthetic scenes and the tracks the REFERENCE's own create_trajectories returned for them (oracle/make_golden.py,
golden_tracks); the library's host implementation (b3d_tracks_from_edges) must return the identical list of lists."""
import pytest
import torch

from conftest import load_golden


def test_tracks_equal_the_reference():
    from batch3dmot_amd.predict_post import create_trajectories
    g = load_golden("g8_tracks.pt")
    names = g["class_names"]
    assert len(g["cases"]) >= 4
    for case in g["cases"]:
        scene_nodes = {i: {"category_name": names[int(c)]} for i, c in enumerate(case["node_cls"])}
        pred_edges = [((int(p[0]), int(p[1])), float(s)) for p, s in zip(case["pred_pairs"], case["pred_scores"])]
        got = create_trajectories(pred_edges, scene_nodes)
        assert got == [list(map(int, t)) for t in case["tracks"]]
        seen = [n for t in got for n in t]
        assert len(seen) == len(set(seen))                          # a detection belongs to at most one track


def test_tracks_argument_checks():
    from batch3dmot_amd.predict_post import create_trajectories
    nodes = {0: {"category_name": "car"}, 1: {"category_name": "car"}}
    assert create_trajectories([], nodes) == []
    assert create_trajectories([((0, 1), 0.5)], nodes) == [[0, 1]]
    with pytest.raises(RuntimeError, match="self loop"):
        create_trajectories([((1, 1), 0.5)], nodes)
