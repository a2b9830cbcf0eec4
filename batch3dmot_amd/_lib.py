"""ctypes binding of ``libb3d_hip.so`` (C ABI declared in ``include/b3d.h``).

The product path has no CPU fallback: if the HIP library is missing or a call fails this module
raises -- it never routes through PyTorch ops or the oracle.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# B3D_LIB=name selects an experiment build (make VARIANT=name: libb3d_hip_name.so); profiling / A-B tools only
LIB_PATH = os.path.join(_HERE, "libb3d_hip_%s.so" % os.environ["B3D_LIB"] if os.environ.get("B3D_LIB") else "libb3d_hip.so")

B3D_FLAG_TRAINING = 1
B3D_FLAG_RUN_DEAD_KNN = 2
B3D_FLAG_SINGLE_STREAM = 4
B3D_FLAG_SKIP_DEAD_LAST_MESSAGES = 16
B3D_FLAG_DEFER_SIDE_JOIN = 8

c_float_p = C.POINTER(C.c_float)


class b3d_graph(C.Structure):
    _fields_ = [("N", C.c_int32), ("E", C.c_int32), ("src", C.c_void_p), ("dst", C.c_void_p),
                ("dst_ptr", C.c_void_p), ("dst_perm", C.c_void_p), ("src_ptr", C.c_void_p),
                ("src_perm", C.c_void_p), ("invalid_edges", C.c_void_p), ("dst_unsorted", C.c_void_p),
                ("past_ptr", C.c_void_p), ("past_rows", C.c_void_p)]


class b3d_linear(C.Structure):
    _fields_ = [("w", C.c_void_p), ("b", C.c_void_p)]


class b3d_gat(C.Structure):
    _fields_ = [("lin", C.c_void_p), ("att_src", C.c_void_p), ("att_dst", C.c_void_p), ("bias", C.c_void_p)]


class b3d_gat_grad(C.Structure):
    _fields_ = [("lin", C.c_void_p), ("att_src", C.c_void_p), ("att_dst", C.c_void_p), ("bias", C.c_void_p)]


class b3d_batchnorm(C.Structure):
    _fields_ = [("gamma", C.c_void_p), ("beta", C.c_void_p), ("running_mean", C.c_void_p), ("running_var", C.c_void_p),
                ("num_batches_tracked", C.c_void_p), ("momentum", C.c_float), ("eps", C.c_float)]


class b3d_mp_weights(C.Structure):
    _fields_ = [("edge_update", b3d_linear * 3), ("create_past_msgs", b3d_linear * 2),
                ("create_future_msgs", b3d_linear * 2), ("combine_future_past", b3d_linear * 3)]


class b3d_pose_weights(C.Structure):
    _fields_ = [("edge_encoder", b3d_linear * 3), ("node_encoder", b3d_linear * 3),
                ("edge_classifier", b3d_linear * 4), ("mp", b3d_mp_weights), ("knn_conv", b3d_gat)]


class b3d_pose_grads(C.Structure):       # same layout as the weights minus knn_conv
    _fields_ = [("edge_encoder", b3d_linear * 3), ("node_encoder", b3d_linear * 3),
                ("edge_classifier", b3d_linear * 4), ("mp", b3d_mp_weights)]


class b3d_mha(C.Structure):
    _fields_ = [("in_proj_weight", C.c_void_p), ("in_proj_bias", C.c_void_p), ("out_proj_weight", C.c_void_p),
                ("out_proj_bias", C.c_void_p)]


class b3d_mlp_desc(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("widths", C.c_int32 * 6), ("relu_mask", C.c_uint32), ("final_sigmoid", C.c_int32)]


class b3d_clr_weights(C.Structure):
    _fields_ = [("edge_encoder", b3d_linear * 3), ("node_encoder", b3d_linear * 2), ("edge_classifier", b3d_linear * 4),
                ("fc_lidar_encoder", b3d_linear * 2), ("fc_radar_encoder", b3d_linear * 3),
                ("c2c_att", b3d_mha), ("l2l_att", b3d_mha), ("r2r_att", b3d_mha),
                ("att_edge_encoder", b3d_linear * 5), ("mp", b3d_mp_weights), ("knn_conv", b3d_gat)]


class b3d_clr_grads(C.Structure):        # same layout minus knn_conv
    _fields_ = [("edge_encoder", b3d_linear * 3), ("node_encoder", b3d_linear * 2), ("edge_classifier", b3d_linear * 4),
                ("fc_lidar_encoder", b3d_linear * 2), ("fc_radar_encoder", b3d_linear * 3),
                ("c2c_att", b3d_mha), ("l2l_att", b3d_mha), ("r2r_att", b3d_mha),
                ("att_edge_encoder", b3d_linear * 5), ("mp", b3d_mp_weights)]


class b3d_clr_inputs(C.Structure):
    _fields_ = [("pose_feats", C.c_void_p), ("edge_attr", C.c_void_p), ("node_timestamps", C.c_void_p),
                ("x_img", C.c_void_p), ("pointnet_out", C.c_void_p), ("lidar_nodes", C.c_void_p), ("n_lidar", C.c_int32),
                ("radarnet_out", C.c_void_p), ("radar_nodes", C.c_void_p), ("n_radar", C.c_int32),
                ("encoders_ready", C.c_void_p)]


_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load the HIP library or fail loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `make -C batch3dmot_amd/csrc` (or "
            "`python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    lib.b3d_version.restype = C.c_int
    lib.b3d_last_error.restype = C.c_char_p
    lib.b3d_graph_workspace_bytes.restype = C.c_size_t
    lib.b3d_graph_workspace_bytes.argtypes = [C.c_int32, C.c_int32]
    lib.b3d_graph_build.restype = C.c_int
    lib.b3d_graph_build.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t,
                                    C.POINTER(b3d_graph), C.c_void_p]
    lib.b3d_pose_workspace_bytes.restype = C.c_size_t
    lib.b3d_pose_workspace_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_uint32]
    lib.b3d_pose_forward.restype = C.c_int
    lib.b3d_pose_forward.argtypes = [C.POINTER(b3d_pose_weights), C.POINTER(b3d_graph), C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_int32, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p,
                                     C.c_void_p, C.c_void_p]
    lib.b3d_pose_backward.restype = C.c_int
    lib.b3d_pose_backward.argtypes = [C.POINTER(b3d_pose_weights), C.POINTER(b3d_graph), C.c_void_p, C.c_void_p,
                                      C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                      C.POINTER(b3d_pose_grads), C.c_void_p]
    lib.b3d_pose_debug_layer_ptrs.restype = C.c_int
    lib.b3d_pose_debug_layer_ptrs.argtypes = [C.c_void_p, C.c_size_t, C.c_int32, C.c_int32, C.c_int32, C.c_uint32,
                                              C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    lib.b3d_clr_workspace_bytes.restype = C.c_size_t
    lib.b3d_clr_workspace_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint32]
    lib.b3d_clr_forward.restype = C.c_int
    lib.b3d_clr_forward.argtypes = [C.POINTER(b3d_clr_weights), C.POINTER(b3d_graph), C.POINTER(b3d_clr_inputs), C.c_int32,
                                    C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.b3d_clr_backward.restype = C.c_int
    lib.b3d_clr_backward.argtypes = [C.POINTER(b3d_clr_weights), C.POINTER(b3d_graph), C.POINTER(b3d_clr_inputs), C.c_int32,
                                     C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.POINTER(b3d_clr_grads), C.c_void_p]
    lib.b3d_clr_debug_ptrs.restype = C.c_int
    lib.b3d_clr_debug_ptrs.argtypes = [C.c_void_p, C.c_size_t, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_uint32, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                       C.POINTER(C.c_void_p)]
    lib.b3d_modality_mask.restype = C.c_int
    lib.b3d_modality_mask.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.b3d_knn_gat_workspace_bytes.restype = C.c_size_t
    lib.b3d_knn_gat_workspace_bytes.argtypes = [C.c_int32, C.c_int32]
    lib.b3d_knn_gat_forward.restype = C.c_int
    lib.b3d_knn_gat_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(b3d_gat),
                                        C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.b3d_knn_gat_backward_workspace_bytes.restype = C.c_size_t
    lib.b3d_knn_gat_backward_workspace_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    lib.b3d_knn_gat_backward.restype = C.c_int
    lib.b3d_knn_gat_backward.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(b3d_gat), C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(b3d_gat_grad), C.c_void_p]
    lib.b3d_edge_loss_workspace_bytes.restype = C.c_size_t
    lib.b3d_edge_loss_workspace_bytes.argtypes = [C.c_int32]
    lib.b3d_edge_loss.restype = C.c_int
    lib.b3d_edge_loss.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int32, C.c_int, C.c_float,
                                  C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.b3d_adam_step.restype = C.c_int
    lib.b3d_adam_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float,
                                  C.c_float, C.c_float, C.c_float, C.c_int64, C.c_void_p]
    lib.b3d_pose_layer_workspace_bytes.restype = C.c_size_t
    lib.b3d_pose_layer_workspace_bytes.argtypes = [C.c_int32, C.c_int32, C.c_uint32]
    lib.b3d_pose_layer_forward.restype = C.c_int
    lib.b3d_pose_layer_forward.argtypes = [C.POINTER(b3d_mp_weights), C.POINTER(b3d_graph), C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                           C.c_void_p]
    lib.b3d_pose_layer_backward.restype = C.c_int
    lib.b3d_pose_layer_backward.argtypes = [C.POINTER(b3d_mp_weights), C.POINTER(b3d_graph), C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(b3d_mp_weights), C.c_void_p]
    lib.b3d_clr_layer_workspace_bytes.restype = C.c_size_t
    lib.b3d_clr_layer_workspace_bytes.argtypes = [C.c_int32, C.c_int32, C.c_uint32]
    lib.b3d_clr_layer_forward.restype = C.c_int
    lib.b3d_clr_layer_forward.argtypes = [C.POINTER(b3d_mp_weights), C.POINTER(b3d_graph), C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                          C.c_void_p]
    lib.b3d_clr_layer_backward.restype = C.c_int
    lib.b3d_clr_layer_backward.argtypes = [C.POINTER(b3d_mp_weights), C.POINTER(b3d_graph), C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(b3d_mp_weights), C.c_void_p]
    lib.b3d_side_join.restype = C.c_int
    lib.b3d_side_join.argtypes = [C.c_void_p]
    lib.b3d_pose_debug_knn_ptrs.restype = C.c_int
    lib.b3d_pose_debug_knn_ptrs.argtypes = [C.c_void_p, C.c_size_t, C.c_int32, C.c_int32, C.c_int32, C.c_uint32,
                                            C.c_void_p, C.c_void_p, C.c_void_p]
    lib.b3d_adam_step_dev.restype = C.c_int
    lib.b3d_adam_step_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float,
                                      C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    lib.b3d_average_precision_workspace_bytes.restype = C.c_size_t
    lib.b3d_average_precision_workspace_bytes.argtypes = [C.c_int64, C.c_int32]
    lib.b3d_average_precision.restype = C.c_int
    lib.b3d_average_precision.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p,
                                          C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.b3d_point_feat_workspace_bytes.restype = C.c_size_t
    lib.b3d_point_feat_workspace_bytes.argtypes = []
    lib.b3d_point_feat_stats.restype = C.c_int
    lib.b3d_point_feat_stats.argtypes = [C.POINTER(b3d_linear), C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                         C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.b3d_point_feat.restype = C.c_int
    lib.b3d_point_feat.argtypes = [C.POINTER(b3d_linear), C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                   C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    lib.b3d_resnet_encode_workspace_bytes.restype = C.c_size_t
    lib.b3d_resnet_encode_workspace_bytes.argtypes = [C.c_int32]
    lib.b3d_modality_rows.restype = C.c_int
    lib.b3d_modality_rows.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.b3d_modality_rows_expect.restype = C.c_int
    lib.b3d_modality_rows_expect.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                             C.c_void_p]
    lib.b3d_fc_bn_workspace_bytes.restype = C.c_size_t
    lib.b3d_fc_bn_workspace_bytes.argtypes = [C.c_int32, C.c_int32]
    lib.b3d_fc_bn_forward.restype = C.c_int
    lib.b3d_fc_bn_forward.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                      C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_size_t, C.c_void_p]
    lib.b3d_affine.restype = C.c_int
    lib.b3d_affine.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.b3d_point_stack_train_workspace_bytes.restype = C.c_size_t
    lib.b3d_point_stack_train_workspace_bytes.argtypes = [C.c_int32]
    lib.b3d_point_stack_train.restype = C.c_int
    lib.b3d_point_stack_train.argtypes = [C.POINTER(b3d_linear), C.POINTER(b3d_batchnorm), C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                          C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.b3d_fc_ticket_init.restype = C.c_int
    lib.b3d_fc_ticket_init.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    lib.b3d_affine_relu.restype = C.c_int
    lib.b3d_affine_relu.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.b3d_resnet_encode.restype = C.c_int
    lib.b3d_resnet_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p,
                                      C.c_void_p]
    lib.b3d_point_moments_workspace_bytes.restype = C.c_size_t
    lib.b3d_point_moments_workspace_bytes.argtypes = []
    lib.b3d_point_moments.restype = C.c_int
    lib.b3d_point_moments.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t,
                                      C.c_void_p, C.c_void_p, C.c_void_p]
    lib.b3d_bn_fold_moments.restype = C.c_int
    lib.b3d_bn_fold_moments.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int64, C.c_void_p, C.c_void_p,
                                        C.c_void_p]
    lib.b3d_bn_minmax_workspace_bytes.restype = C.c_size_t
    lib.b3d_bn_minmax_workspace_bytes.argtypes = [C.c_int32]
    lib.b3d_bn_minmax_apply.restype = C.c_int
    lib.b3d_bn_minmax_apply.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int32, C.c_void_p,
                                        C.c_size_t, C.c_void_p, C.c_void_p]
    lib.b3d_post_workspace_bytes.restype = C.c_size_t
    lib.b3d_post_workspace_bytes.argtypes = [C.c_int64, C.c_int64]
    lib.b3d_post_greedy.restype = C.c_int
    lib.b3d_post_greedy.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32,
                                    C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.b3d_tracks_from_edges.restype = C.c_int
    lib.b3d_tracks_from_edges.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32,
                                          C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
    lib.b3d_prof_enable.argtypes = [C.c_int]
    lib.b3d_prof_markers.argtypes = [C.c_int]
    lib.b3d_prof_markers.restype = C.c_int
    lib.b3d_prof_select.argtypes = [C.c_uint32]
    lib.b3d_prof_select.restype = C.c_int
    lib.b3d_prof_read.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    lib.b3d_prof_pair_overhead_us.restype = C.c_int
    lib.b3d_prof_pair_overhead_us.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]
    lib.b3d_features.restype = C.c_uint32
    lib.b3d_features.argtypes = []
    lib.b3d_mlp_workspace_bytes.restype = C.c_size_t
    lib.b3d_mlp_workspace_bytes.argtypes = [C.POINTER(b3d_mlp_desc), C.c_int64, C.c_uint32]
    lib.b3d_mlp_forward.restype = C.c_int
    lib.b3d_mlp_forward.argtypes = [C.POINTER(b3d_mlp_desc), C.POINTER(b3d_linear), C.c_void_p, C.c_int64, C.c_uint32, C.c_void_p,
                                    C.c_size_t, C.c_void_p, C.c_void_p]
    lib.b3d_mlp_backward_scratch_bytes.restype = C.c_size_t
    lib.b3d_mlp_backward_scratch_bytes.argtypes = [C.POINTER(b3d_mlp_desc), C.c_int64]
    lib.b3d_mlp_backward.restype = C.c_int
    lib.b3d_mlp_backward.argtypes = [C.POINTER(b3d_mlp_desc), C.POINTER(b3d_linear), C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                     C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.POINTER(b3d_linear), C.c_void_p]
    lib.b3d_xattn_node_affine_workspace_bytes.restype = C.c_size_t
    lib.b3d_xattn_node_affine_workspace_bytes.argtypes = [C.c_int64, C.c_int32, C.c_uint32]
    lib.b3d_xattn_node_affine_forward.restype = C.c_int
    lib.b3d_xattn_node_affine_forward.argtypes = [C.POINTER(b3d_mha), C.c_int32, C.c_void_p, C.c_int64, C.c_uint32, C.c_void_p,
                                                  C.c_size_t, C.c_void_p, C.c_void_p]
    lib.b3d_xattn_node_affine_scratch_bytes.restype = C.c_size_t
    lib.b3d_xattn_node_affine_scratch_bytes.argtypes = [C.c_int64, C.c_int32]
    lib.b3d_xattn_node_affine_backward.restype = C.c_int
    lib.b3d_xattn_node_affine_backward.argtypes = [C.POINTER(b3d_mha), C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                                   C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.POINTER(b3d_mha),
                                                   C.c_void_p]
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load().b3d_last_error()
        raise RuntimeError(f"{what} failed ({status}): {msg.decode() if msg else '?'}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def current_stream(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def require_cuda(t: torch.Tensor, name: str, dtype=None) -> None:
    if not t.is_cuda:
        raise ValueError(f"{name} must live on the GPU (got {t.device}); the HIP path has no CPU fallback")
    if dtype is not None and t.dtype != dtype:
        raise ValueError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")


class Graph:
    """Device-side graph structure built once per batch and reused by all layers and by backward."""

    def __init__(self, edge_index: torch.Tensor, num_nodes: int, validated: bool = False, ws: Optional[torch.Tensor] = None):
        """``validated``: the caller has already checked that every endpoint lies in [0, num_nodes) (``Data.to`` does
        so on the CPU copy).  Otherwise the build's counter of out-of-range edges is read back (one 4-byte copy, a
        host synchronisation) and a ``ValueError`` is raised, where the reference raises an index error
        (pose_gnn.py:180); during stream capture the read is impossible and skipped -- the build has rewritten such
        edges to the self loop (0, 0), so nothing indexes out of bounds, and ``invalid_edges()`` reports them later."""
        require_cuda(edge_index, "edge_index", torch.int64)
        if edge_index.dim() != 2 or edge_index.size(0) != 2:
            raise ValueError(f"edge_index must be [2, E], got {tuple(edge_index.shape)}")
        lib = load()
        self.N, self.E = int(num_nodes), int(edge_index.size(1))
        nbytes = lib.b3d_graph_workspace_bytes(self.N, self.E)
        if ws is not None:
            # a caller-owned buffer (``Graph.workspace_bytes``): the structure lands at fixed addresses -- what a hipGraph-captured
            # step needs when the NEXT batch's structure is built under the current step (train_step.EncodeAhead.launch_graph)
            if ws.dtype != torch.uint8 or ws.device != edge_index.device or ws.numel() < nbytes or not ws.is_contiguous():
                raise ValueError(f"Graph: workspace must be a contiguous uint8 tensor of >= {nbytes} bytes on {edge_index.device}")
            self.ws = ws
        else:
            self.ws = torch.empty(nbytes, dtype=torch.uint8, device=edge_index.device)
        self.c = b3d_graph()
        check(lib.b3d_graph_build(edge_index.data_ptr(), self.N, self.E, self.ws.data_ptr(), nbytes,
                                  C.byref(self.c), current_stream(edge_index.device)), "b3d_graph_build")
        self._keep = edge_index
        if not validated and not torch.cuda.is_current_stream_capturing():
            bad = self.invalid_edges()
            if bad:
                raise ValueError(f"edge_index has {bad} edge(s) with an endpoint outside [0, {self.N})")

    @staticmethod
    def workspace_bytes(num_nodes: int, num_edges: int) -> int:
        return int(load().b3d_graph_workspace_bytes(int(num_nodes), int(num_edges)))

    def invalid_edges(self) -> int:
        """Number of edges with an endpoint outside [0, N) (synchronises)."""
        return int(self._view(self.c.invalid_edges, 1).item())

    def _view(self, p, n):
        off = (p - self.ws.data_ptr())
        return self.ws[off:off + 4 * n].view(torch.int32)

    def arrays(self):
        """int32 views (for tests): src, dst, dst_ptr, dst_perm, src_ptr, src_perm."""
        g = self.c
        return {"src": self._view(g.src, self.E), "dst": self._view(g.dst, self.E),
                "dst_ptr": self._view(g.dst_ptr, self.N + 1), "dst_perm": self._view(g.dst_perm, self.E),
                "src_ptr": self._view(g.src_ptr, self.N + 1), "src_perm": self._view(g.src_perm, self.E)}


KERNEL_FAMILIES = {"mp_edge_fwd": 0, "mp_edge_bwd": 1, "mp_node_fwd": 2, "mp_node_bwd": 3, "wgrad_edge": 4,
                   "wgrad_other": 5, "other": 6, "att_fwd": 7, "att_bwd": 8, "knn_gat": 9, "point_feat": 10}


def prof_enable(on: bool, families=None) -> None:
    """Time kernel families with HIP events; ``families`` (names) restricts the event pairs."""
    mask = 0xFFFFFFFF
    if families is not None:
        mask = 0
        for f in families:
            mask |= 1 << KERNEL_FAMILIES[f]
    check(load().b3d_prof_select(C.c_uint32(mask)), "b3d_prof_select")
    check(load().b3d_prof_enable(1 if on else 0), "b3d_prof_enable")
    check(load().b3d_prof_reset(), "b3d_prof_reset")


def prof_markers(on: bool) -> bool:
    """roctx ranges named after the kernel family around every launch (rocprofv3 --marker-trace); False if no marker library."""
    return bool(load().b3d_prof_markers(1 if on else 0))


def prof_read() -> dict:
    """{family: (total_ms, launches)} measured with HIP events on the launch stream."""
    out = {}
    for name, fam in KERNEL_FAMILIES.items():
        ms, n = C.c_double(), C.c_int()
        check(load().b3d_prof_read(fam, C.byref(ms), C.byref(n)), "b3d_prof_read")
        out[name] = (ms.value, n.value)
    return out


def prof_pair_overhead_us(stream: int, reps: int = 256) -> float:
    """Average elapsed time of a HIP event pair around an EMPTY kernel on ``stream`` (us)."""
    out = C.c_double()
    check(load().b3d_prof_pair_overhead_us(stream, reps, C.byref(out)), "b3d_prof_pair_overhead_us")
    return out.value


def features() -> dict:
    """Execution plans of this build (which first layers are evaluated per node): for FLOP accounting."""
    f = int(load().b3d_features())
    return {"pose_hoist": bool(f & 1), "clr_hoist_mp": bool(f & 2), "clr_hoist_att": bool(f & 4)}


def knn_gat(x: torch.Tensor, node_timestamps: torch.Tensor, conv, k: int = 20):
    """Frame-wise k-NN + GATConv as a standalone operator (reference pose_gnn.py:74-80).  ``conv`` is a
    ``GATConvParams``.  Returns (nbr [N,32] int32, cnt [N] int32, y [N,D])."""
    require_cuda(x, "x", torch.float32)
    lib = load()
    n, d = x.shape
    ts = node_timestamps.to(torch.int64).contiguous()
    require_cuda(ts, "node_timestamps", torch.int64)
    if ts.device != x.device or ts.numel() != n:
        raise ValueError(f"node_timestamps must hold one int64 per row of x on {x.device}, got {tuple(ts.shape)} on {ts.device}")
    g = b3d_gat()
    keep = [conv.lin_src.weight.detach().contiguous(), conv.att_src.detach().reshape(-1).contiguous(),
            conv.att_dst.detach().reshape(-1).contiguous(), conv.bias.detach().contiguous()]
    g.lin, g.att_src, g.att_dst, g.bias = (t.data_ptr() for t in keep)
    nbytes = lib.b3d_knn_gat_workspace_bytes(n, d)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    nbr = torch.empty((n, 32), dtype=torch.int32, device=x.device)
    cnt = torch.empty(n, dtype=torch.int32, device=x.device)
    y = torch.empty((n, d), dtype=torch.float32, device=x.device)
    check(lib.b3d_knn_gat_forward(x.data_ptr(), ts.data_ptr(), n, d, k, C.byref(g), ws.data_ptr(), nbytes,
                                  nbr.data_ptr(), cnt.data_ptr(), y.data_ptr(), current_stream(x.device)), "b3d_knn_gat_forward")
    return nbr, cnt, y


class _KnnGatFunction(torch.autograd.Function):
    """y = GATConv(x, knn_graph(x) per frame) with gradients (``b3d_knn_gat_forward`` / ``b3d_knn_gat_backward``); the neighbour
    selection carries no gradient."""

    @staticmethod
    def forward(ctx, x, ts, k, lin, att_src, att_dst, bias):
        lib = load()
        n, d = x.shape
        keep = [lin.detach().contiguous(), att_src.detach().reshape(-1).contiguous(), att_dst.detach().reshape(-1).contiguous(),
                bias.detach().contiguous()]
        g = b3d_gat()
        g.lin, g.att_src, g.att_dst, g.bias = (t.data_ptr() for t in keep)
        xc = x.detach().contiguous()
        nbytes = lib.b3d_knn_gat_workspace_bytes(n, d)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        nbr = torch.empty((n, 32), dtype=torch.int32, device=x.device)
        cnt = torch.empty(n, dtype=torch.int32, device=x.device)
        y = torch.empty((n, d), dtype=torch.float32, device=x.device)
        check(lib.b3d_knn_gat_forward(xc.data_ptr(), ts.data_ptr(), n, d, k, C.byref(g), ws.data_ptr(), nbytes,
                                      nbr.data_ptr(), cnt.data_ptr(), y.data_ptr(), current_stream(x.device)), "b3d_knn_gat_forward")
        ctx.k = k
        ctx.shapes = (att_src.shape, att_dst.shape)
        ctx.save_for_backward(xc, nbr, cnt, *keep)
        ctx.mark_non_differentiable(nbr, cnt)
        return y, nbr, cnt

    @staticmethod
    def backward(ctx, d_y, _d_nbr, _d_cnt):
        lib = load()
        x, nbr, cnt, lin, att_src, att_dst, bias = ctx.saved_tensors
        n, d = x.shape
        g = b3d_gat()
        g.lin, g.att_src, g.att_dst, g.bias = lin.data_ptr(), att_src.data_ptr(), att_dst.data_ptr(), bias.data_ptr()
        d_y = d_y.contiguous().float()
        if d_y.data_ptr() % 16:                      # the kernels read d_y with 16-byte loads
            d_y = d_y.clone()
        d_x = torch.empty_like(x)
        grads = [torch.empty_like(lin), torch.empty_like(att_src), torch.empty_like(att_dst), torch.empty_like(bias)]
        gg = b3d_gat_grad()
        gg.lin, gg.att_src, gg.att_dst, gg.bias = (t.data_ptr() for t in grads)
        nbytes = lib.b3d_knn_gat_backward_workspace_bytes(n, d, ctx.k)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        check(lib.b3d_knn_gat_backward(x.data_ptr(), n, d, ctx.k, C.byref(g), nbr.data_ptr(), cnt.data_ptr(), d_y.data_ptr(),
                                       ws.data_ptr(), nbytes, d_x.data_ptr(), C.byref(gg), current_stream(x.device)),
              "b3d_knn_gat_backward")
        return d_x, None, None, grads[0], grads[1].reshape(ctx.shapes[0]), grads[2].reshape(ctx.shapes[1]), grads[3]


def knn_gat_conv(x: torch.Tensor, node_timestamps: torch.Tensor, conv, k: int = 20, return_graph: bool = False):
    """Differentiable frame-wise k-NN + GATConv (reference pose_gnn.py:74-80 with the result USED: ``knn_writeback``): ``y`` [N,D]
    with gradients to ``x`` and to ``conv``'s parameters (``GATConvParams``).  D = 48 or 96."""
    require_cuda(x, "x", torch.float32)
    ts = node_timestamps.to(torch.int64).contiguous()
    require_cuda(ts, "node_timestamps", torch.int64)
    if ts.device != x.device or ts.numel() != x.size(0):
        raise ValueError(f"node_timestamps must hold one int64 per row of x on {x.device}, got {tuple(ts.shape)} on {ts.device}")
    y, nbr, cnt = _KnnGatFunction.apply(x, ts, int(k), conv.lin_src.weight, conv.att_src, conv.att_dst, conv.bias)
    return (y, nbr, cnt) if return_graph else y


def _mlp_desc(widths, relu_mask: int, final_sigmoid: bool) -> b3d_mlp_desc:
    d = b3d_mlp_desc()
    d.n_layers = len(widths) - 1
    for i, w in enumerate(widths):
        d.widths[i] = int(w)
    d.relu_mask = int(relu_mask)
    d.final_sigmoid = 1 if final_sigmoid else 0
    return d


class _MlpFunction(torch.autograd.Function):
    """``nn.Sequential(Linear, ReLU, ..., [Sigmoid])`` as ``b3d_mlp_forward`` / ``b3d_mlp_backward``."""

    @staticmethod
    def forward(ctx, x, relu_mask, final_sigmoid, training, *wb):
        lib = load()
        n_layers = len(wb) // 2
        ws_ = wb[0::2]
        widths = [int(ws_[0].size(1))] + [int(w.size(0)) for w in ws_]
        if x.dim() != 2 or x.size(1) != widths[0]:
            raise ValueError(f"mlp: input must be [rows, {widths[0]}], got {tuple(x.shape)}")
        xc = x.detach().contiguous()
        keep = [t.detach().contiguous() for t in wb]
        for t in keep:
            require_cuda(t, "mlp parameter", torch.float32)
        rows = int(xc.size(0))
        d = _mlp_desc(widths, relu_mask, final_sigmoid)
        layers = (b3d_linear * n_layers)()
        for l in range(n_layers):
            layers[l].w, layers[l].b = keep[2 * l].data_ptr(), keep[2 * l + 1].data_ptr()
        flags = B3D_FLAG_TRAINING if training else 0           # (decided by the caller: grad mode is off inside Function.forward)
        nbytes = lib.b3d_mlp_workspace_bytes(C.byref(d), rows, flags)
        if nbytes == 0:
            raise ValueError(f"mlp: unsupported stack {widths}")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        y = torch.empty((rows, widths[-1]), dtype=torch.float32, device=x.device)
        check(lib.b3d_mlp_forward(C.byref(d), layers, xc.data_ptr(), rows, flags, ws.data_ptr(), nbytes, y.data_ptr(),
                                  current_stream(x.device)), "b3d_mlp_forward")
        ctx.d, ctx.ws, ctx.nbytes, ctx.training = d, ws, nbytes, training
        ctx.save_for_backward(xc, y, *keep)
        return y

    @staticmethod
    def backward(ctx, d_y):
        lib = load()
        if not ctx.training:
            raise RuntimeError("backward through an mlp forward that ran without gradient tracking")
        xc, y, *keep = ctx.saved_tensors
        n_layers = len(keep) // 2
        rows = int(xc.size(0))
        layers = (b3d_linear * n_layers)()
        grads = (b3d_linear * n_layers)()
        out = [torch.empty_like(t) for t in keep]
        for l in range(n_layers):
            layers[l].w, layers[l].b = keep[2 * l].data_ptr(), keep[2 * l + 1].data_ptr()
            grads[l].w, grads[l].b = out[2 * l].data_ptr(), out[2 * l + 1].data_ptr()
        d_y = d_y.contiguous().float()
        d_x = torch.empty_like(xc) if ctx.needs_input_grad[0] else None
        sbytes = lib.b3d_mlp_backward_scratch_bytes(C.byref(ctx.d), rows)
        scratch = torch.empty(sbytes, dtype=torch.uint8, device=xc.device)
        check(lib.b3d_mlp_backward(C.byref(ctx.d), layers, xc.data_ptr(), y.data_ptr(), rows, ctx.ws.data_ptr(), ctx.nbytes,
                                   scratch.data_ptr(), sbytes, d_y.data_ptr(), ptr(d_x), grads, current_stream(xc.device)),
              "b3d_mlp_backward")
        return (d_x, None, None, None, *out)


def mlp(seq, x: torch.Tensor) -> torch.Tensor:
    """``seq(x)`` for an ``nn.Sequential`` of Linear / ReLU (/ trailing Sigmoid) modules through the library's MLP operator
    (``b3d_mlp_forward`` / ``_backward``), differentiable in ``x`` and in the Linear parameters."""
    from torch import nn
    require_cuda(x, "x", torch.float32)
    lins, relu_mask, sigmoid = [], 0, False
    for m in seq:
        if isinstance(m, nn.Linear):
            if sigmoid:
                raise ValueError("mlp: a Linear behind the Sigmoid")
            if m.bias is None:
                raise ValueError("mlp: Linear layers without bias are not supported")
            lins.append(m)
        elif isinstance(m, nn.ReLU):
            if not lins:
                raise ValueError("mlp: ReLU in front of the first Linear")
            relu_mask |= 1 << (len(lins) - 1)
        elif isinstance(m, nn.Sigmoid):
            sigmoid = True
        else:
            raise ValueError(f"mlp: unsupported module {type(m).__name__}")
    if not 1 <= len(lins) <= 5:
        raise ValueError(f"mlp: {len(lins)} Linear layers (1..5 supported)")
    wb = [t for m in lins for t in (m.weight, m.bias)]
    training = torch.is_grad_enabled() and (x.requires_grad or any(t.requires_grad for t in wb))
    return _MlpFunction.apply(x, relu_mask, sigmoid, training, *wb)


class _XattnFunction(torch.autograd.Function):
    """``nn.MultiheadAttention`` with one query and one key per edge == ``out_proj(v_proj(value))`` per node
    (``b3d_xattn_node_affine_*``; clr_att_gnn.py:143-159)."""

    @staticmethod
    def forward(ctx, x, training, in_w, in_b, out_w, out_b):
        lib = load()
        d = int(out_w.size(0))
        if x.dim() != 2 or x.size(1) != d or tuple(in_w.shape) != (3 * d, d):
            raise ValueError(f"xattn: x {tuple(x.shape)}, in_proj_weight {tuple(in_w.shape)}, embed dim {d}")
        xc = x.detach().contiguous()
        keep = [t.detach().contiguous() for t in (in_w, in_b, out_w, out_b)]
        for t in keep:
            require_cuda(t, "attention parameter", torch.float32)
        att = b3d_mha()
        att.in_proj_weight, att.in_proj_bias, att.out_proj_weight, att.out_proj_bias = (t.data_ptr() for t in keep)
        n = int(xc.size(0))
        flags = B3D_FLAG_TRAINING if training else 0
        nbytes = lib.b3d_xattn_node_affine_workspace_bytes(n, d, flags)
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=x.device)
        y = torch.empty((n, d), dtype=torch.float32, device=x.device)
        check(lib.b3d_xattn_node_affine_forward(C.byref(att), d, xc.data_ptr(), n, flags, ws.data_ptr(), nbytes, y.data_ptr(),
                                                current_stream(x.device)), "b3d_xattn_node_affine_forward")
        ctx.dim, ctx.ws, ctx.nbytes, ctx.training = d, ws, nbytes, training
        ctx.save_for_backward(xc, y, *keep)
        return y

    @staticmethod
    def backward(ctx, d_y):
        lib = load()
        if not ctx.training:
            raise RuntimeError("backward through an xattn forward that ran without gradient tracking")
        xc, y, *keep = ctx.saved_tensors
        d, n = ctx.dim, int(xc.size(0))
        att = b3d_mha()
        att.in_proj_weight, att.in_proj_bias, att.out_proj_weight, att.out_proj_bias = (t.data_ptr() for t in keep)
        out = [torch.empty_like(t) for t in keep]
        g = b3d_mha()
        g.in_proj_weight, g.in_proj_bias, g.out_proj_weight, g.out_proj_bias = (t.data_ptr() for t in out)
        d_y = d_y.contiguous().float()
        d_x = torch.empty_like(xc) if ctx.needs_input_grad[0] else None
        sbytes = lib.b3d_xattn_node_affine_scratch_bytes(n, d)
        scratch = torch.empty(sbytes, dtype=torch.uint8, device=xc.device)
        check(lib.b3d_xattn_node_affine_backward(C.byref(att), d, xc.data_ptr(), y.data_ptr(), n, ctx.ws.data_ptr(), ctx.nbytes,
                                                 scratch.data_ptr(), sbytes, d_y.data_ptr(), ptr(d_x), C.byref(g),
                                                 current_stream(xc.device)), "b3d_xattn_node_affine_backward")
        return (d_x, None, *out)


def xattn_node_affine(att, x: torch.Tensor) -> torch.Tensor:
    """What ``att(query, key, value)`` returns when every query has ONE key (clr_att_gnn.py:143-159), for the value rows
    ``x`` [N, D]: ``att.out_proj(v_proj(x))``, differentiable; the query / key thirds of ``in_proj`` get zero gradients."""
    require_cuda(x, "x", torch.float32)
    ps = (att.in_proj_weight, att.in_proj_bias, att.out_proj.weight, att.out_proj.bias)
    training = torch.is_grad_enabled() and (x.requires_grad or any(t.requires_grad for t in ps))
    return _XattnFunction.apply(x, training, *ps)


class Workspace:
    """Owner of a forward's workspace tensor.  A training forward may return while the discarded k-NN
    block is still running on the library's side stream (B3D_FLAG_DEFER_SIDE_JOIN); backward joins
    it.  If backward never runs, the join happens here, BEFORE the tensor goes back to the caching
    allocator, so the block can never write into memory that has been handed to someone else."""

    def __init__(self, tensor: torch.Tensor, pending: bool):
        self.tensor = tensor
        self.pending = pending

    def joined(self) -> None:
        self.pending = False

    def __del__(self):
        if getattr(self, "pending", False):
            try:
                check(load().b3d_side_join(current_stream(self.tensor.device)), "b3d_side_join")
            except Exception:
                torch.cuda.synchronize(self.tensor.device)
