/* libb3d_hip.so -- C ABI of the MI355X (gfx950) GNN message-passing path for Batch3DMOT graphs.
 *
 * What this boundary replaces.  The reference is pure Python: its hot path is reached through
 * two nn.Module.forward calls,
 *     PoseGNN.forward(data) -> (edge_logits [E,1], x_enc [N,48])      batch_3dmot/models/pose_gnn.py:58-86
 *     GNN.forward(data)     -> (edge_prob  [E,1], x_sens [N,288])     batch_3dmot/models/clr_att_gnn.py:95-188
 * called from Batch3DMOT.train (train.py:133,174) and combine_batches_to_scene (predict.py:194),
 * and all of its arithmetic lives in torch / torch_geometric / torch_scatter / torch_cluster
 * kernels.  The entry points below are what a ctypes binding of those two forwards (and of the
 * autograd backward that train.py:159 triggers) binds; batch3dmot_amd/_lib.py is that binding
 * and INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless a comment says "host"; tensors are dense,
 *    row-major, fp32 unless stated; Linear weights are [out, in] as torch stores them.
 *  - the library allocates nothing, frees nothing and keeps no pointer after a call returns:
 *    scratch comes from the caller as a workspace whose size the *_workspace_* queries report.
 *  - every call is asynchronous on the given hipStream_t and never synchronises the device.
 *  - return value: B3D_OK (0) or a negative b3d_status; b3d_last_error() gives the message of
 *    the calling thread's last failure.  Nothing throws or aborts.
 *  - re-entrant; the only global state is the thread-local error string.
 */
#ifndef B3D_H_
#define B3D_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* b3d_stream;   /* = hipStream_t */

typedef enum b3d_status {
  B3D_OK = 0,
  B3D_ERR_ARG = -1,        /* bad shape / null pointer / unsupported option */
  B3D_ERR_HIP = -2,        /* a HIP runtime call or kernel launch failed */
  B3D_ERR_WORKSPACE = -3   /* workspace too small */
} b3d_status;

int b3d_version(void);                 /* 10000*major + 100*minor + patch */
const char* b3d_last_error(void);      /* host string, valid until the thread's next failing call */

/* ---- graph structure (replaces MessagePassing.__collect__ index plumbing, pose_gnn.py:180,
 *      and torch_scatter's index handling, pose_gnn.py:240) ---------------------------------- */
typedef struct b3d_graph {
  int32_t N, E;
  const int32_t* src;        /* [E] edge_index[0]: past / source node j          */
  const int32_t* dst;        /* [E] edge_index[1]: current / destination node i  */
  const int32_t* dst_ptr;    /* [N+1] CSR by destination                         */
  const int32_t* dst_perm;   /* [E] edge ids grouped by destination, ascending   */
  const int32_t* src_ptr;    /* [N+1] CSC by source                              */
  const int32_t* src_perm;   /* [E] edge ids grouped by source, ascending        */
  const int32_t* invalid_edges; /* device int32[1]: edges with an endpoint outside [0, N).  The reference raises an
                                   index error for them (pose_gnn.py:180); the build rewrites each to the self loop
                                   (0, 0) so that no kernel can index out of bounds, and the caller reads this counter
                                   (one 4-byte copy) to raise the error -- batch3dmot_amd._lib.Graph does            */
  /* Round 6 -- per-destination sums INSIDE the edge kernel (pose_gnn.py:187-191,228-240 / clr_att_gnn.py:293-294,336-344: the
   * `past` messages are scatter-added at their destination).  Detection graphs list their edges grouped by destination
   * (dst non-decreasing); then a wavefront's 16 consecutive edges hold whole runs of one destination, the camera+LiDAR+radar
   * edge kernel adds each run with DPP row shifts and stores ONE row per (destination, 16-edge block) -- at the run's last
   * edge -- and the node kernel sums those few rows instead of one row per edge.  dst_unsorted: device int32[1], non-zero
   * when the edge list is not grouped like that (or holds an invalid edge): the kernels then keep one row per edge.
   * past_ptr [N+1] / past_rows: the rows of `past` the node kernel sums for node n -- the run tails, or (unsorted) exactly
   * dst_ptr / dst_perm.  Written by b3d_graph_build; NULL in a hand-made struct means "one row per edge". */
  const int32_t* dst_unsorted;
  const int32_t* past_ptr;
  const int32_t* past_rows;
} b3d_graph;

size_t b3d_graph_workspace_bytes(int32_t N, int32_t E);
/* edge_index: [2,E] int64 contiguous (row 0 = source, row 1 = destination), any order. */
int b3d_graph_build(const int64_t* edge_index, int32_t N, int32_t E, void* workspace,
                    size_t workspace_bytes, b3d_graph* out /* host */, b3d_stream stream);

/* ---- parameters ------------------------------------------------------------------------------ */
typedef struct b3d_linear { const float* w; const float* b; } b3d_linear;   /* w [out,in], b [out] */
typedef struct b3d_linear_grad { float* w; float* b; } b3d_linear_grad;

typedef struct b3d_gat {           /* torch_geometric GATConv(D, D, heads=1, add_self_loops=False) */
  const float* lin;                /* [D,D]  (lin_src == lin_dst) */
  const float* att_src;            /* [D] */
  const float* att_dst;            /* [D] */
  const float* bias;               /* [D] */
} b3d_gat;

/* CausalMessagePassing parameters (pose_gnn.py:91-120 / clr_att_gnn.py:193-222) */
typedef struct b3d_mp_weights {
  b3d_linear edge_update[3];
  b3d_linear create_past_msgs[2];
  b3d_linear create_future_msgs[2];
  b3d_linear combine_future_past[3];
} b3d_mp_weights;
typedef struct b3d_mp_grads {
  b3d_linear_grad edge_update[3];
  b3d_linear_grad create_past_msgs[2];
  b3d_linear_grad create_future_msgs[2];
  b3d_linear_grad combine_future_past[3];
} b3d_mp_grads;

/* PoseGNN parameters (pose_gnn.py:29-56) */
typedef struct b3d_pose_weights {
  b3d_linear edge_encoder[3];      /* 4-8-16-32   */
  b3d_linear node_encoder[3];      /* 19-24-36-48 */
  b3d_linear edge_classifier[4];   /* 32-16-8-4-1 */
  b3d_mp_weights mp;
  b3d_gat knn_conv;                /* D = 48 */
} b3d_pose_weights;
typedef struct b3d_pose_grads {
  b3d_linear_grad edge_encoder[3];
  b3d_linear_grad node_encoder[3];
  b3d_linear_grad edge_classifier[4];
  b3d_mp_grads mp;
} b3d_pose_grads;

/* flags */
#define B3D_FLAG_TRAINING      1u   /* keep what backward needs in the workspace            */
#define B3D_FLAG_RUN_DEAD_KNN  2u   /* execute the frame-wise k-NN + GAT block whose result the
                                       reference discards (pose_gnn.py:74-80)                 */
#define B3D_FLAG_DEFER_SIDE_JOIN 8u  /* training forwards only: return while the discarded k-NN + GAT block may still be
                                       running on the library's side stream, so that it overlaps the loss and the
                                       backward sweep.  The caller MUST keep `workspace` alive and untouched until
                                       b3d_pose_backward / b3d_clr_backward on it has been enqueued (it joins the side
                                       stream into its `stream`), or call b3d_side_join(stream) before releasing it */
#define B3D_FLAG_SKIP_DEAD_LAST_MESSAGES 16u  /* b3d_clr_forward: do NOT execute the last layer's create_future_msgs / create_past_msgs /
                                       combine_future_past -- forward returns edge_classifier(edge_attr) (clr_att_gnn.py:188), the `x` of
                                       the last layer is never read and autograd never visits those stacks.  The reference executes them;
                                       so does this library by default.  Outputs and gradients are bit-identical either way */
#define B3D_FLAG_SINGLE_STREAM 4u   /* enqueue every kernel on `stream` itself.  By default work with
                                       no consumer until the end of the call (the discarded k-NN +
                                       GAT block) runs on a library-owned side stream that is forked
                                       from and joined back into `stream` inside the call; results
                                       and ordering as seen from `stream` are the same either way  */

/* ---- PoseGNN.forward / backward ------------------------------------------------------------ */
size_t b3d_pose_workspace_bytes(int32_t N, int32_t E, int32_t depth, uint32_t flags);

/* pose_feats [N,19] f32; edge_attr [E,4] f64 (cast to f32 inside, pose_gnn.py:67);
 * node_timestamps [N] int64 (only read with B3D_FLAG_RUN_DEAD_KNN);
 * out_logits [E,1]; out_x_enc [N,48] (the pre-message-passing node encoding, pose_gnn.py:71,86). */
int b3d_pose_forward(const b3d_pose_weights* w /* host struct of device pointers */,
                     const b3d_graph* g /* host */, const float* pose_feats, const double* edge_attr,
                     const int64_t* node_timestamps, int32_t depth, uint32_t flags, void* workspace,
                     size_t workspace_bytes, float* out_logits, float* out_x_enc, b3d_stream stream);

/* Gradients of every parameter for upstream d_logits [E,1] and d_x_enc [N,48] (either may be
 * NULL = zero).  `workspace` is the one a B3D_FLAG_TRAINING forward filled.  knn_conv receives no
 * gradient (the reference's result is discarded), so there is no field for it. */
int b3d_pose_backward(const b3d_pose_weights* w, const b3d_graph* g, const float* pose_feats,
                      const double* edge_attr, int32_t depth, void* workspace, size_t workspace_bytes,
                      const float* d_logits, const float* d_x_enc, const b3d_pose_grads* grads /* host */,
                      b3d_stream stream);

/* ---- GNN (camera + LiDAR + radar) forward / backward -- clr_att_gnn.py:16-188 -------------------
 * The three frozen per-detection encoders (resnet.encode, pointnet.forward_feat,
 * radarnet.forward_feat; clr_att_gnn.py:125,131,139) are ADJACENT to the path and stay with the
 * caller: their outputs are inputs here.  Everything from the modality heads onwards runs in this
 * library: fc_lidar / fc_radar heads on the rows that have the modality (:127-141), the cross-edge
 * modality attention (:143-159; nn.MultiheadAttention with one query and one key is exactly
 * out_proj(v_proj(value)), hoisted from per edge to per node), att_edge_encoder (:161-164),
 * node / edge encoders, 6 CausalMessagePassing layers with att_edge_attr (:178-186), the sigmoid
 * classifier (:49-58,188). */
typedef struct b3d_mha {            /* nn.MultiheadAttention(D, 2 heads, batch_first) parameters */
  const float* in_proj_weight;      /* [3D, D]  (q | k | v rows) */
  const float* in_proj_bias;        /* [3D] */
  const float* out_proj_weight;     /* [D, D] */
  const float* out_proj_bias;       /* [D] */
} b3d_mha;
typedef struct b3d_mha_grad { float* in_proj_weight; float* in_proj_bias; float* out_proj_weight; float* out_proj_bias; } b3d_mha_grad;

typedef struct b3d_clr_weights {
  b3d_linear edge_encoder[3];       /* 4-16-32-64            clr_att_gnn.py:35-41 */
  b3d_linear node_encoder[2];       /* 19-48-96              :43-47 */
  b3d_linear edge_classifier[4];    /* 64-32-16-8-1 +Sigmoid :49-58 */
  b3d_linear fc_lidar_encoder[2];   /* 256-192-128           :60-64 */
  b3d_linear fc_radar_encoder[3];   /* 256-192-128-64        :66-72 */
  b3d_mha c2c_att, l2l_att, r2r_att;/* D = 96, 128, 64       :77-79 */
  b3d_linear att_edge_encoder[5];   /* 640-512-384-256-128-64 :81-91 */
  b3d_mp_weights mp;                /* widths of clr_att_gnn.py:196-222 */
  b3d_gat knn_conv;                 /* D = 96 */
} b3d_clr_weights;
typedef struct b3d_clr_grads {
  b3d_linear_grad edge_encoder[3];
  b3d_linear_grad node_encoder[2];
  b3d_linear_grad edge_classifier[4];
  b3d_linear_grad fc_lidar_encoder[2];
  b3d_linear_grad fc_radar_encoder[3];
  b3d_mha_grad c2c_att, l2l_att, r2r_att;   /* q / k thirds of in_proj receive exact zeros */
  b3d_linear_grad att_edge_encoder[5];
  b3d_mp_grads mp;
} b3d_clr_grads;

typedef struct b3d_clr_inputs {
  const float* pose_feats;          /* [N,19] */
  const double* edge_attr;          /* [E,4] float64 */
  const int64_t* node_timestamps;   /* [N] (only with B3D_FLAG_RUN_DEAD_KNN) */
  const float* x_img;               /* [N,96]   resnet.encode(img_feats) */
  const float* pointnet_out;        /* [n_lidar,256] pointnet.forward_feat of the rows that have LiDAR */
  const int32_t* lidar_nodes;       /* [n_lidar] their node ids, ascending */
  int32_t n_lidar;
  const float* radarnet_out;        /* [n_radar,256] */
  const int32_t* radar_nodes;       /* [n_radar] */
  int32_t n_radar;
  void* encoders_ready;             /* optional hipEvent_t, recorded by the caller once x_img / pointnet_out / radarnet_out are
                                       complete (on whatever streams produced them).  b3d_clr_forward waits for it on `stream`
                                       right before its first read of the three -- AFTER the part of the forward that does not
                                       need them (weight images, edge / node encoder, layer 0's per-node table, the first k-NN
                                       block), so the frozen encoders can still be running on other streams when it is called.
                                       NULL: the three are complete in `stream` order.  Ignored by b3d_clr_backward. */
} b3d_clr_inputs;

size_t b3d_clr_workspace_bytes(int32_t N, int32_t E, int32_t n_lidar, int32_t n_radar, int32_t depth, uint32_t flags);
/* out_prob [E,1] (after the sigmoid); out_x_sens [N,288] = x_img | x_lidar | x_radar (:172). */
int b3d_clr_forward(const b3d_clr_weights* w, const b3d_graph* g, const b3d_clr_inputs* in, int32_t depth,
                    uint32_t flags, void* workspace, size_t workspace_bytes, float* out_prob,
                    float* out_x_sens, b3d_stream stream);
/* d_prob [E,1], d_x_sens [N,288] (either may be NULL). */
int b3d_clr_backward(const b3d_clr_weights* w, const b3d_graph* g, const b3d_clr_inputs* in, int32_t depth,
                     void* workspace, size_t workspace_bytes, const float* d_prob, const float* d_x_sens,
                     const b3d_clr_grads* grads, b3d_stream stream);
/* Modality presence (clr_att_gnn.py:107-121): has[n] = (sum of row n) != 0, rows of `width` floats. */
int b3d_modality_mask(const float* feats, int32_t N, int32_t width, uint8_t* has /* [N] */, b3d_stream stream);
/* The same masks as ascending row ids (torch.nonzero(mask).squeeze(1), clr_att_gnn.py:131,139): has [N] (scratch, also returned),
 * rows [N] int64 (the first *count entries are valid), count [1] int32 -- two launches, no library scan. */
int b3d_modality_rows(const float* feats, int32_t N, int32_t width, uint8_t* has, int64_t* rows, int32_t* count,
                      b3d_stream stream);
/* The same for a caller that has FIXED the count beforehand (a hipGraph-captured step: the count is a shape of everything behind it):
 * at most `expected` ids are written to rows [expected]; *count receives the real count and *mismatch (int32, device) is incremented
 * when it differs -- no host read-back; the caller inspects the flag whenever it next synchronises (clr_att_gnn.py:107-121). */
int b3d_modality_rows_expect(const float* feats, int32_t N, int32_t width, uint8_t* has, int64_t* rows, int32_t expected,
                             int32_t* count, int32_t* mismatch, b3d_stream stream);

/* ---- one CausalMessagePassing layer as a standalone operator ------------------------------------------
 * Replaces `CausalMessagePassing.forward(x, edge_index, edge_attr, initial_x[, att_edge_attr])`
 * (pose_gnn.py:125-252 / clr_att_gnn.py:227-356): (x [N,DX], e [E,DE], x0 [N,DX]) -> (x' [N,DX], e' [E,DE]).
 * Poses-only widths (DX 48, DE 32): forward and backward.  `flags`: B3D_FLAG_TRAINING keeps what
 * backward needs in `workspace` (pass the same workspace, untouched, to backward).  Backward takes
 * d x' / d e' (NULL = zero), the forward's inputs and its e' output, and fills d x, d x0, d e (NULL =
 * not wanted) and all ten Linear gradients of `grads` (overwritten). */
size_t b3d_pose_layer_workspace_bytes(int32_t N, int32_t E, uint32_t flags);
int b3d_pose_layer_forward(const b3d_mp_weights* weights /* host */, const b3d_graph* g, const float* x, const float* x0,
                           const float* e, uint32_t flags, void* workspace, size_t workspace_bytes, float* x_new,
                           float* e_new, b3d_stream stream);
int b3d_pose_layer_backward(const b3d_mp_weights* weights, const b3d_graph* g, const float* x, const float* x0,
                            const float* e, const float* e_new, void* workspace, size_t workspace_bytes,
                            const float* d_x_new, const float* d_e_new, float* d_x, float* d_x0, float* d_e,
                            const b3d_mp_grads* grads /* host struct of device pointers */, b3d_stream stream);
/* Camera+LiDAR+radar widths (DX 96, DE 64, att_edge_attr [E,64]): the same pair; backward also returns d att_edge_attr
 * (NULL = not wanted).  The model entry points (b3d_clr_forward / _backward) run the same arithmetic over all layers with
 * the first layers' node columns evaluated per node. */
size_t b3d_clr_layer_workspace_bytes(int32_t N, int32_t E, uint32_t flags);
int b3d_clr_layer_forward(const b3d_mp_weights* weights, const b3d_graph* g, const float* x, const float* x0,
                          const float* e, const float* att_edge_attr, uint32_t flags, void* workspace, size_t workspace_bytes,
                          float* x_new, float* e_new, b3d_stream stream);
int b3d_clr_layer_backward(const b3d_mp_weights* weights, const b3d_graph* g, const float* x, const float* x0,
                           const float* e, const float* att_edge_attr, const float* e_new, void* workspace,
                           size_t workspace_bytes, const float* d_x_new, const float* d_e_new, float* d_x, float* d_x0,
                           float* d_e, float* d_att_edge_attr, const b3d_mp_grads* grads, b3d_stream stream);

/* ---- row-wise MLP stacks as a standalone operator (b3d_mlp_fwd/bwd) -----------------------------------
 * Replaces nn.Sequential(Linear, ReLU, Linear, ...[, Sigmoid]).forward and its autograd for the reference's small stacks:
 * edge_encoder / node_encoder / edge_classifier (pose_gnn.py:29-53, clr_att_gnn.py:35-58), fc_lidar_encoder /
 * fc_radar_encoder (clr_att_gnn.py:60-72), att_edge_encoder (clr_att_gnn.py:81-91).  The whole-model entry points run the
 * same stacks inside their fused kernels; these are for callers that drive the layers themselves (GNN.knn_writeback).
 * x [rows, widths[0]] -> y [rows, widths[n_layers]], dense row-major fp32; layers[l] = {w [widths[l+1], widths[l]], b or NULL}.
 * Exact-fp32 matrix instruction (v_mfma_f32_16x16x4_f32, bitwise an fmaf chain); weight gradients summed over fixed row
 * chunks in chunk order (bitwise reproducible, no float atomics).
 * forward: with B3D_FLAG_TRAINING the hidden activations stay in `workspace` for backward (pass it on untouched).
 * backward: x and y as the forward saw / produced them, d_y [rows, widths[n_layers]]; d_x may be NULL; grads[l].w / .b are
 * OVERWRITTEN (each may be NULL); `scratch` (b3d_mlp_backward_scratch_bytes) is free after the call.  rows == 0: zero gradients. */
typedef struct b3d_mlp_desc {
  int32_t n_layers;          /* 1..5 Linear layers */
  int32_t widths[6];         /* widths[0] = input features, widths[l+1] = outputs of layer l; each 1..1024 */
  uint32_t relu_mask;        /* bit l: a ReLU follows layer l */
  int32_t final_sigmoid;     /* != 0: a Sigmoid follows the last layer (clr_att_gnn.py:57) */
} b3d_mlp_desc;
size_t b3d_mlp_workspace_bytes(const b3d_mlp_desc* d /* host */, int64_t rows, uint32_t flags);
int b3d_mlp_forward(const b3d_mlp_desc* d /* host */, const b3d_linear* layers /* host array of device pointers */,
                    const float* x, int64_t rows, uint32_t flags, void* workspace, size_t workspace_bytes, float* y,
                    b3d_stream stream);
size_t b3d_mlp_backward_scratch_bytes(const b3d_mlp_desc* d /* host */, int64_t rows);
int b3d_mlp_backward(const b3d_mlp_desc* d, const b3d_linear* layers, const float* x, const float* y, int64_t rows,
                     void* workspace, size_t workspace_bytes, void* scratch, size_t scratch_bytes, const float* d_y,
                     float* d_x, const b3d_linear_grad* grads /* host */, b3d_stream stream);

/* ---- the one-key cross-edge attention per NODE (b3d_xattn_node_affine) -------------------------------
 * clr_att_gnn.py:143-159 calls nn.MultiheadAttention(D, 2 heads) with ONE query and ONE key per edge: the softmax over a
 * single key is 1, so the module's output is out_proj(v_proj(value)) and depends on the value's node only.
 * y [N,D] = out_proj(in_proj[2D:3D] x + in_proj_bias[2D:3D]).  Backward fills d_x (may be NULL) and `grads` (every field may
 * be NULL): the query / key thirds of in_proj_weight / in_proj_bias receive exact zeros, as autograd gives them.
 * Workspace / scratch as for b3d_mlp_*. */
size_t b3d_xattn_node_affine_workspace_bytes(int64_t N, int32_t D, uint32_t flags);
int b3d_xattn_node_affine_forward(const b3d_mha* att /* host */, int32_t D, const float* x, int64_t N, uint32_t flags,
                                  void* workspace, size_t workspace_bytes, float* y, b3d_stream stream);
size_t b3d_xattn_node_affine_scratch_bytes(int64_t N, int32_t D);
int b3d_xattn_node_affine_backward(const b3d_mha* att, int32_t D, const float* x, const float* y, int64_t N, void* workspace,
                                   size_t workspace_bytes, void* scratch, size_t scratch_bytes, const float* d_y, float* d_x,
                                   const b3d_mha_grad* grads /* host */, b3d_stream stream);

/* Same update with the step counter on the device (`*step_dev` = number of steps taken so far, incremented by
 * the call): for training steps captured into a hipGraph, where a by-value `step` would be replayed. */
int b3d_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                      float beta1, float beta2, float eps, float weight_decay, int64_t* step_dev, b3d_stream stream);

/* ---- frame-wise k-NN + GATConv (pose_gnn.py:74-80, clr_att_gnn.py:178-184) as a standalone operator --
 * For every distinct timestamp value: k nearest neighbours (Euclidean, feature space, no self
 * loops, fewer than k in frames of <= k nodes) among the nodes of that frame, then
 * GATConv(D, D, heads=1, add_self_loops=False) on that graph.  D = 48 or 96, k <= 32.
 * out_nbr [N,32] int32 (neighbour node ids by ascending distance, unspecified beyond out_cnt),
 * out_cnt [N] int32, out_y [N,D].  The model forwards run exactly this when B3D_FLAG_RUN_DEAD_KNN
 * is set and discard the result, as the reference does. */
size_t b3d_knn_gat_workspace_bytes(int32_t N, int32_t D);
int b3d_knn_gat_forward(const float* x, const int64_t* node_timestamps, int32_t N, int32_t D, int32_t k,
                        const b3d_gat* gat /* host */, void* workspace, size_t workspace_bytes,
                        int32_t* out_nbr, int32_t* out_cnt, float* out_y, b3d_stream stream);

/* Backward of the block for `knn_writeback=True` (the mode in which its result is used; the reference drops it, pose_gnn.py:80:
 * SURVEY.md Appendix A.3, section 8b's b3d_gat_bwd).  x [N,D] and `gat` as given to the forward, nbr / cnt as the forward returned
 * them (no gradient flows through the neighbour selection), d_y [N,D].  Outputs: d_x [N,D] (may be NULL) and the parameter
 * gradients (each may be NULL): lin [D,D], att_src / att_dst / bias [D].  Fixed summation order (CSC lists of the k-NN graph by
 * b3d_graph_build, column sums in node order): bitwise reproducible.  D = 48 or 96, 1 <= k <= 32 (the forward's k). */
typedef struct b3d_gat_grad { float* lin; float* att_src; float* att_dst; float* bias; } b3d_gat_grad;
size_t b3d_knn_gat_backward_workspace_bytes(int32_t N, int32_t D, int32_t k);
int b3d_knn_gat_backward(const float* x, int32_t N, int32_t D, int32_t k, const b3d_gat* gat /* host */, const int32_t* nbr,
                         const int32_t* cnt, const float* d_y, void* workspace, size_t workspace_bytes, float* d_x,
                         const b3d_gat_grad* grads /* host */, b3d_stream stream);

/* ---- fused edge loss of the training loop (train.py:136-141) ----------------------------------------
 *   loss = scale * mean_i( w_i * BCE(out_i, y_i) ),   d_out_i = d loss / d out_i,   scale = 1/batch_size
 * out [E] float32: probabilities (torch.nn.BCELoss semantics incl. the -100 log clamp) or, with
 * from_logits != 0, logits (BCEWithLogitsLoss).  y [E]: float32, or int64 when y_is_int64 != 0.
 * weight [E] float32 or NULL (train.py:136-139, the class-balanced factors `data.edge_weights`).
 * loss_out: device float[1].  d_out [E] or NULL.  E == 0 is an error.  Deterministic reduction. */
size_t b3d_edge_loss_workspace_bytes(int32_t E);
int b3d_edge_loss(const float* out, const void* y, int y_is_int64, const float* weight, int32_t E, int from_logits,
                  float scale, void* workspace, size_t workspace_bytes, float* loss_out, float* d_out,
                  b3d_stream stream);

/* ---- Adam step of the training loop (train.py:106-109, 160) over one contiguous fp32 buffer --------
 * torch.optim.Adam(lr, betas, eps, weight_decay) semantics (L2 decay added to the gradient, bias
 * corrections from `step` = 1, 2, ...; amsgrad off).  All four arrays [n] on the device. */
int b3d_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                  float beta1, float beta2, float eps, float weight_decay, int64_t step, b3d_stream stream);

/* Make `stream` wait for everything the library has enqueued on its side streams of the current device
 * (pending work of a B3D_FLAG_DEFER_SIDE_JOIN forward).  Cheap when nothing is pending. */
int b3d_side_join(b3d_stream stream);

/* ---- post-processing of per-window edge scores (predict.py:199-233, 92-124) -----------------------------
 * pairs [M,2] int64: (global source id, global destination id) of every scored edge of every window, in
 * processing order; scores [M] float32.  Mean score per distinct pair (float64 sum in order of appearance /
 * count), kept iff mean > class_threshold[node_class[source]], kept edges in first-appearance order; per node
 * the best incoming edge's source (pred) and the best outgoing edge's destination (succ), ties to the edge that
 * appeared first, -1 if none.  node_class [N] int64 (index into class_threshold [C] float64).
 * Outputs: kept_pairs [M,2] int64 / kept_scores [M] float64 (first counts[1] rows valid), pred / succ [N] int64,
 * counts: device int32[3] = {distinct edges, kept edges, entries of pairs / node_class outside [0,N) / [0,C)}.
 * Out-of-range entries are clamped (nothing indexes out of bounds) and counted; a caller treats counts[2] != 0 as
 * the KeyError / IndexError the reference's dictionaries raise.  Scores of any sign.  Deterministic; no host
 * synchronisation. */
size_t b3d_post_workspace_bytes(int64_t M, int64_t N);
int b3d_post_greedy(const int64_t* pairs, const float* scores, int64_t M, const int64_t* node_class, int64_t N,
                    const double* class_threshold, int32_t num_classes, void* workspace, size_t workspace_bytes,
                    int64_t* kept_pairs,
                    double* kept_scores, int64_t* pred, int64_t* succ, int32_t* counts, b3d_stream stream);

/* ---- hierarchical track clustering of the greedy edges (predict.py:262-375, mode "hier") -- HOST arrays ------------
 * A sequential greedy merge over at most two edges per detection of a scene (each decision depends on the clusters
 * the higher-scoring edges formed), so it runs on the host.  pairs [M,2] int64 (source j, destination i), scores [M]
 * float64 in list order (a repeated pair keeps its first position and takes its last score, as the reference's dict
 * does); edges are taken by descending score, ties in list order; node_class [N] int64 indexes join_threshold [C]
 * float64 (the class of the DESTINATION decides whether two clusters join).  Outputs: track_nodes [<= 2 M] int64 =
 * the tracks' node ids back to back, track_ptr [<= M + 1] int64, *n_tracks; tracks in the reference's order. */
int b3d_tracks_from_edges(const int64_t* pairs /* host */, const double* scores /* host */, int64_t M,
                          const int64_t* node_class /* host */, int64_t N, const double* join_threshold /* host */,
                          int32_t num_classes, int64_t* track_nodes /* host */, int64_t* track_ptr /* host */,
                          int64_t* n_tracks /* host */);

/* ---- point-cloud feature stacks of the frozen LiDAR / radar encoders, eval mode ------------------------------
 * (models/pointnet.py:9-57 STN3d and :111-165 PointNetfeat, models/radarnet.py:9-37 RadarNetfeat)
 * out[b, :] = max over the P points of  L3(relu(L2(relu(L1(x'[b, :, p])))))  (+ ReLU if relu_last), where L1..L3 are
 * the three conv1d(kernel 1) layers WITH their eval-mode BatchNorm folded in by the caller: conv[i].w [out, in]
 * (64 x C, 128 x 64, 1024 x 128), conv[i].b [out]; x [B, C, P] float32 (C <= 4, P = 128 or 64);
 * trans [B, 3, 3] or NULL: x'[b, :, p] = x[b, :, p]^T . trans[b]  (the bmm of pointnet.py:137).  out [B, 1024]. */
size_t b3d_point_feat_workspace_bytes(void);
/* Train-mode form (BatchNorm with the statistics of this batch): conv[0], conv[1] carry their BatchNorm folded with
 * the batch statistics the caller has computed for them, conv[2] is the raw last conv; per cloud and feature the
 * maximum, minimum, sum and sum of squares of conv[2]'s output over the points are returned ([B, 1024] each), from
 * which the caller forms the last BatchNorm's statistics and applies it to the maximum (scale > 0) or minimum. */
int b3d_point_feat_stats(const b3d_linear* conv, const float* x, const float* trans, int32_t B, int32_t C, int32_t P,
                         void* workspace, size_t workspace_bytes, float* out_max, float* out_min, float* out_sum,
                         float* out_sq, b3d_stream stream);
int b3d_point_feat(const b3d_linear* conv, const float* x, const float* trans, int32_t B, int32_t C, int32_t P,
                   int32_t relu_last, void* workspace, size_t workspace_bytes, float* out, b3d_stream stream);

/* ResNetAE.encode (reference batch_3dmot/models/resnet_fully_conv.py:42-82 ResidualBlock, :84-161 ResNetAE.encode: conv(3,12,4,2,1), ResidualBlock(12,24,k4,s2), ResidualBlock(24,48,k3,s1),
 * ResidualBlock(48,96,k3,s2)) on x [N,3,32,32] -> out [N,96], BatchNorm in train mode (`train` != 0: batch statistics, running
 * statistics and num_batches_tracked updated as nn.BatchNorm2d does) or eval mode (running statistics).
 * conv[10]: conv, block1.{conv1,conv2,downsample.0}, block2.{...}, block3.{...}, weights [out,in,k,k];
 * bn[9]: block1.{bn1,bn2,downsample.1}, block2.{...}, block3.{...}.  momentum < 0: cumulative average (momentum=None). */
typedef struct b3d_batchnorm {
  const float* gamma;
  const float* beta;
  float* running_mean;             /* NULL (with running_var): statistics not tracked; train mode only */
  float* running_var;
  int64_t* num_batches_tracked;    /* may be NULL */
  float momentum, eps;
} b3d_batchnorm;
size_t b3d_resnet_encode_workspace_bytes(int32_t N);
int b3d_resnet_encode(const b3d_linear* conv, const b3d_batchnorm* bn, const float* x, int32_t N, int32_t train,
                      void* workspace, size_t workspace_bytes, float* out, b3d_stream stream);

/* Fully connected heads of the frozen point encoders (reference batch_3dmot/models/pointnet.py:46-48 STN3d fc1-bn4-relu, fc2-bn5-relu,
 * fc3 + identity; pointnet.py:188-192 and radarnet.py:60-64 forward_feat: fc1-bn1-relu, fc2-dropout-bn2-relu), one launch per Linear:
 *     y [B,N] = mask * ( act(x * in_scale + in_shift) . w^T + bias ) + add          act = ReLU (in_relu != 0) or the identity
 * x [B,K] (K a multiple of 4), w [N,K] row-major; in_scale / in_shift [K] (both or neither: the PRODUCER's BatchNorm affine, applied
 * -- with its ReLU if in_relu -- while x is read; in_relu == 0 is the point stacks' last BatchNorm in front of fc1 of
 * forward_feat, pointnet.py:188 / radarnet.py:60, which has no ReLU), mask [B,N] (Dropout: 0 or 1 / (1 - p), drawn by the caller so that the random stream stays the
 * caller's), add [N] (STN3d's flattened identity); each may be NULL.  With `bn` the launch also produces THIS layer's BatchNorm
 * affine out_scale / out_shift [N]: from the batch statistics of y in train mode (`train` != 0, B > 1; running statistics and
 * num_batches_tracked updated as nn.BatchNorm1d does; summed in a fixed order, bitwise reproducible), from the running statistics
 * in eval mode.  The first 256 bytes of the (256-byte aligned) workspace are arrival counters (one per 64-column tile of y, plus one
 * for the launch: N <= 4032) that must be ZERO on entry; the launch leaves them zero (b3d_fc_ticket_init zeroes them for a fresh
 * workspace).  b3d_affine_relu: out = relu(y * scale + shift),
 * the last activation of a chain.  Products are bf16x6 (exact three-way bf16 split of both operands, six of the nine piece
 * products on v_mfma_f32_16x16x32_bf16, fp32 accumulation): fp32-class accuracy (~3e-7 relative), not bitwise an fp32 fmaf chain;
 * a +-inf input yields NaN (inf - inf in the split) where torch's Linear yields +-inf.  Batch variance: per-tile sums of squared
 * deviations from the tile mean, combined in float64 (no E[y^2] - E[y]^2 cancellation). */
size_t b3d_fc_bn_workspace_bytes(int32_t B, int32_t N);
int b3d_fc_ticket_init(void* workspace, size_t workspace_bytes, b3d_stream stream);
int b3d_fc_bn_forward(const float* x, int32_t B, int32_t K, const float* w, const float* bias, int32_t N,
                      const float* in_scale, const float* in_shift, int32_t in_relu, const float* mask, const float* add,
                      const b3d_batchnorm* bn, int32_t train, float* y, float* out_scale, float* out_shift,
                      void* workspace, size_t workspace_bytes, b3d_stream stream);
int b3d_affine_relu(const float* y, const float* scale, const float* shift, int32_t B, int32_t N, float* out, b3d_stream stream);
/* out = y * scale + shift (+ ReLU if relu != 0), per column. */
int b3d_affine(const float* y, const float* scale, const float* shift, int32_t B, int32_t N, int32_t relu, float* out, b3d_stream stream);

/* Mean mu [K] and second moments second [K,K] = E[h h^T] (float64) over all B * P points of the input of a point stack's
 * first layer (fold1 == NULL: h = the point, transformed by `trans` if given, K = C) or of its second layer (fold1 = the first
 * layer with ITS BatchNorm folded in, [64,C] / [64]: h = relu(fold1(point)), K = 64). */
size_t b3d_point_moments_workspace_bytes(void);
int b3d_point_moments(const b3d_linear* fold1, const float* x, const float* trans, int32_t B, int32_t C, int32_t P,
                      void* workspace, size_t workspace_bytes, double* mu, double* second, b3d_stream stream);

/* Train-mode BatchNorm bookkeeping of the point stacks in one or two launches each (the PyTorch form is ~25 tiny launches per
 * layer).  b3d_bn_fold_moments: the pre-activation of a kernel-1 convolution is affine in its input, z = W h + b, so over all
 * `count` points mean(z) = W mu + b and var(z)_o = W_o (second - mu mu^T) W_o^T with the input's mean `mu` [C] and second
 * moments `second` [C,C] (float64, C <= 64); the batch's BatchNorm is folded into the layer (wf [O,C] = W * scale, bf [O] =
 * b * scale + shift) and the running statistics / num_batches_tracked are updated as nn.BatchNorm1d does (momentum < 0: the
 * cumulative average of momentum=None; NULL running statistics: not tracked).
 * b3d_bn_minmax_apply: from the per-cloud max / min / sum / sum of squares [B,F] that b3d_point_feat_stats returns, the batch
 * statistics over `count` points, the running-statistics update, and y [B,F] = BN(max) where the scale is positive, BN(min)
 * where it is negative (+ ReLU). */
int b3d_bn_fold_moments(const double* mu, const double* second, int32_t C, const float* W, const float* b, int32_t O,
                        const float* gamma, const float* beta, float* running_mean, float* running_var,
                        int64_t* num_batches_tracked, float momentum, float eps, int64_t count, float* wf, float* bf,
                        b3d_stream stream);
size_t b3d_bn_minmax_workspace_bytes(int32_t F);
int b3d_bn_minmax_apply(const float* vmax, const float* vmin, const float* vsum, const float* vsq, int32_t B, int32_t F,
                        int64_t count, const float* gamma, const float* beta, float* running_mean, float* running_var,
                        int64_t* num_batches_tracked, float momentum, float eps, int32_t relu, void* workspace,
                        size_t workspace_bytes, float* y, b3d_stream stream);

/* ---- a whole train-mode point stack in ONE call (round 5) ---------------------------------------------------------------------
 * conv1d(C,64,1)+BN+ReLU -> conv1d(64,128,1)+BN+ReLU -> conv1d(128,1024,1)+BN -> max over the points, BatchNorm on the statistics of
 * THIS batch (pointnet.py:23-42 STN3d, :128-160 PointNetfeat, radarnet.py:17-36 RadarNetfeat in .train()); running statistics and
 * num_batches_tracked of the three BatchNorms updated as nn.BatchNorm1d does.  conv[3]: the RAW convolutions (w [out,in], b);
 * bn[3]: their BatchNorms.  Seven launches (b3d_point_moments / b3d_bn_fold_moments / b3d_point_feat_stats / b3d_bn_minmax_apply
 * composed by the caller were ten): the first layer's statistics are finished by the last workgroup of the input-moments launch, the
 * 64 x 64 moments of the second layer's input accumulate as bf16x6, and sign(gamma) of the LAST BatchNorm is folded into the last
 * convolution so that only the maximum is tracked.  Outputs: ext [B,1024] = max over the points of sign(gamma3) * conv3(h2), and
 * the batch's last BatchNorm as an affine map of it:  BN3(.) maximised over the points = ext * out_scale + out_shift (out_scale >= 0;
 * [1024] each) -- apply it with b3d_affine, or hand the pair to b3d_fc_bn_forward as in_scale / in_shift of the next Linear.
 * `tickets`: 64 bytes of arrival counters that must be ZERO on entry; the call leaves them zero (stream order), so one zero-filled
 * buffer per encoder serves every call. */
size_t b3d_point_stack_train_workspace_bytes(int32_t B);
int b3d_point_stack_train(const b3d_linear* conv /* host [3] */, const b3d_batchnorm* bn /* host [3] */, const float* x,
                          const float* trans, int32_t B, int32_t C, int32_t P, void* tickets, void* workspace,
                          size_t workspace_bytes, float* ext, float* out_scale, float* out_shift, b3d_stream stream);

/* ---- average precision of the edge scores (train.py:18,143-150,188-196) -----------------------------------
 * torchmetrics.functional average_precision(out, gt, pos_label=1) of the whole batch (ap[0]) and of the edges of
 * every class c in 1..num_classes (ap[c], the reference's out[edge_classes == c]); count[s] = edges in the set
 * (the reference skips classes with none).  scores [E] float32; y [E] float32 or int64 (positive iff == 1);
 * edge_classes [E] float32 class ids or NULL (overall only).  ap: device float64[num_classes + 1] (NaN for a set
 * without positives, as torchmetrics' 0/0), count: device int32[num_classes + 1].  One curve point per distinct
 * score; float64; deterministic; no host synchronisation. */
size_t b3d_average_precision_workspace_bytes(int64_t E, int32_t num_classes);
int b3d_average_precision(const float* scores, const void* y, int32_t y_is_int64, const float* edge_classes, int64_t E,
                          int32_t num_classes, void* workspace, size_t workspace_bytes, double* ap, int32_t* count,
                          b3d_stream stream);

/* ---- test hooks: addresses of intermediate tensors inside a workspace a forward has filled ---------------- */
int b3d_pose_debug_layer_ptrs(void* workspace, size_t workspace_bytes, int32_t N, int32_t E, int32_t depth,
                              uint32_t flags, int32_t layer, float** x /* [N,48] */, float** e /* [E,32] */);
/* outputs of the LAST executed k-NN + GAT block (layer 2*floor((depth-1)/2)): y [N,48], nbr [N,32], cnt [N] */
int b3d_pose_debug_knn_ptrs(void* workspace, size_t workspace_bytes, int32_t N, int32_t E, int32_t depth, uint32_t flags,
                            float** y, int32_t** nbr, int32_t** cnt);

/* ---- kernel-family timers (measurement aid for bench.py; off by default) --------------------
 * When enabled, every launch of the listed kernel families is bracketed by hipEventRecord on the
 * launch stream.  b3d_prof_read synchronises on the recorded events and returns the summed device
 * time and launch count of one family since the last reset.  Process-global, mutex-protected. */
typedef enum b3d_kernel_family {
  B3D_K_EDGE_FWD = 0,   /* mp_edge_fwd  : fused edge phase, forward              */
  B3D_K_EDGE_BWD = 1,   /* mp_edge_bwd  : fused edge phase, data gradient        */
  B3D_K_NODE_FWD = 2,   /* mp_node_fwd  : segment sums + node MLP                */
  B3D_K_NODE_BWD = 3,   /* mp_node_bwd                                           */
  B3D_K_WGRAD_EDGE = 4, /* wgrad launch over the edge stacks of one layer        */
  B3D_K_WGRAD_OTHER = 5,
  B3D_K_OTHER = 6,
  B3D_K_ATT_FWD = 7,    /* att_edge_encoder (clr_att_gnn.py:161-164), forward                       */
  B3D_K_ATT_BWD = 8,    /* att_edge_encoder data gradient                                            */
  B3D_K_KNN = 9,        /* frame-wise k-NN + GAT block whose result the reference discards           */
  B3D_K_POINT_FEAT = 10,/* point-cloud stacks of the frozen LiDAR / radar encoders                   */
  B3D_K_COUNT = 11
} b3d_kernel_family;
int b3d_prof_enable(int on);
int b3d_prof_select(uint32_t family_mask);   /* bit f set: family f is timed while enabled (default: all).  Event
                                               pairs cost device time; time one family to measure it undisturbed */
int b3d_prof_reset(void);
/* roctx ranges ("b3d:<family>", the names above) around every kernel-family launch, for rocprofv3 --marker-trace timelines
 * (SURVEY.md section 5).  The marker library (librocprofiler-sdk-roctx / libroctx64) is looked up at run time; returns 1 when
 * ranges are being emitted, 0 when switched off or no marker library is present.  B3D_ROCTX=1 in the environment switches
 * them on without a call.  Ranges are host-side: under hipGraph capture they bracket the capture, not the replays. */
int b3d_prof_markers(int on);
int b3d_prof_read(int family, double* total_ms /* host */, int* launches /* host */);
/* Average elapsed time of an event pair around an EMPTY kernel on `stream`, in us (pair cost = this - that kernel). */
int b3d_prof_pair_overhead_us(b3d_stream stream, int reps, double* out_us /* host */);

/* ---- which execution plans this build runs (for callers that account executed FLOPs, e.g. bench.py) ----------
 * bit 0: PoseGNN first layers evaluated per node (hoisted); bit 1: the same for the camera+LiDAR+radar
 * message-passing widths; bit 2: att_edge_encoder.0's node columns evaluated per node. */
#define B3D_FEATURE_POSE_HOIST 1u
#define B3D_FEATURE_CLR_HOIST_MP 2u
#define B3D_FEATURE_CLR_HOIST_ATT 4u
uint32_t b3d_features(void);

#ifdef __cplusplus
}
#endif
#endif /* B3D_H_ */
