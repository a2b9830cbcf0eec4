"""batch3dmot_amd -- MI355X (gfx950) native GNN message-passing path for Batch3DMOT graphs.

Drop-in ``nn.Module``s for the reference's ``batch_3dmot.models.pose_gnn.PoseGNN`` and
``batch_3dmot.models.clr_att_gnn.GNN`` whose arithmetic runs in hand-written HIP kernels
reached through the C-ABI library ``libb3d_hip.so`` (``include/b3d.h``).
"""
from .data import Data, collate  # noqa: F401

__all__ = ["Data", "collate", "PoseGNN", "GNN", "CausalMessagePassing"]


def __getattr__(name):
    # model modules import the HIP library lazily so that ``import batch3dmot_amd`` (data
    # containers, synthetic graphs) works on a box that has not built the extension yet.
    if name == "PoseGNN":
        from .pose_gnn import PoseGNN
        return PoseGNN
    if name == "GNN":
        from .clr_att_gnn import GNN
        return GNN
    if name == "CausalMessagePassing":
        # the poses-only widths (pose_gnn.py:89-252); the camera+LiDAR+radar widths live in clr_att_gnn.CausalMessagePassing
        from .pose_gnn import CausalMessagePassing
        return CausalMessagePassing
    raise AttributeError(name)
