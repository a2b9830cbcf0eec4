"""TEST INFRASTRUCTURE ONLY.

CPU restatement of the Batch3DMOT GNN message-passing hot path (reference:
batch_3dmot/models/pose_gnn.py, batch_3dmot/models/clr_att_gnn.py).  Nothing in
``batch3dmot_amd`` may import this package: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do, and
only as the checker / the timed CPU baseline.
"""
