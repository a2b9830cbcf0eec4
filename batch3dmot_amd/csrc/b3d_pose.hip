// PoseGNN.forward / backward (reference batch_3dmot/models/pose_gnn.py:24-86) as a sequence of
// gfx950 kernel launches on one stream.  All scratch lives in the caller's workspace.
#include "b3d_launch.hpp"
#include "b3d_knn.hpp"
#include "b3d_wstream.hpp"
#include "b3d_wstream2.hpp"
#include "b3d_hoist.hpp"

namespace b3d {

using D = DimsP;
using HP = Hoist<DimsP>;
#ifndef B3D_NW_EDGE_H
#define B3D_NW_EDGE_H 8
#endif
constexpr int kNWEdgeH = B3D_NW_EDGE_H;   // wavefronts per workgroup of the hoisted edge kernels (4 = two independent workgroups per CU: measured the same 26 / 30 us per launch)

// The model runs with its first layers hoisted to per-node tables (b3d_hoist.hpp); the single-layer operator, whose x / x0
// come from the caller, runs the unsplit kernels (PoseWs::hoist = false).
// encoders / classifier, widths padded to multiples of 16
using SeqEdgeEnc = LayerSeq<L<16, 16>, L<16, 16>, L<16, 32>>;            // 4-8-16-32    pose_gnn.py:29-35
using SeqNodeEnc = LayerSeq<L<32, 32>, L<32, 48>, L<48, 48>>;            // 19-24-36-48  :37-43
using SeqNodeEncH = LayerSeq<L<32, 32>, L<32, 48>, L<48, 48>, L<48, 192>, L<48, 432>>;   // layers 3, 4: x0 terms + layer-0 table (node_proj0_split_kernel)
using SeqCls = LayerSeq<L<32, 16>, L<16, 16>, L<16, 16>, L<16, 16>>;     // 32-16-8-4-1  :45-53
using SeqClsT = LayerSeq<L<16, 16>, L<16, 16>, L<16, 16>, L<16, 32>>;    // W4^T, W3^T, W2^T, W1^T
using SeqEdgeEncT = LayerSeq<L<32, 16>, L<16, 16>>;                      // W3^T, W2^T
using SeqNodeEncT = LayerSeq<L<48, 48>, L<48, 32>>;                      // W3^T, W2^T

enum { LIN_EE0, LIN_EE1, LIN_EE2, LIN_NE0, LIN_NE1, LIN_NE2, LIN_C0, LIN_C1, LIN_C2, LIN_C3,
       LIN_EU0, LIN_EU1, LIN_EU2, LIN_PA0, LIN_PA1, LIN_FU0, LIN_FU1, LIN_CF0, LIN_CF1, LIN_CF2, LIN_COUNT };

struct LinDim { int N, K; };
static const LinDim kLinDims[LIN_COUNT] = {
    {8, 4}, {16, 8}, {32, 16}, {24, 19}, {36, 24}, {48, 36}, {16, 32}, {8, 16}, {4, 8}, {1, 4},
    {96, 128}, {64, 96}, {32, 64}, {96, 128}, {64, 96}, {96, 128}, {64, 96}, {96, 128}, {64, 96}, {48, 64}};
static const bool kLinOnEdges[LIN_COUNT] = {1, 1, 1, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0};

// Task plan of the streaming weight gradient: 14 jobs over the 10 message-passing Linear layers (a
// 96x128 matrix is two 64-column jobs sharing a slab), a work-proportional share of ~2048 wavefront
// tasks each (two resident wavefronts per SIMD).
enum { WJ_EU0A, WJ_EU0B, WJ_EU1, WJ_EU2, WJ_PA0A, WJ_PA0B, WJ_PA1, WJ_FU0A, WJ_FU0B, WJ_FU1,
       WJ_CF0A, WJ_CF0B, WJ_CF1, WJ_CF2,
       // narrow stacks (one application each): classifier, node encoder, edge encoder
       WJ_C3, WJ_C2, WJ_C1, WJ_C0, WJ_NE2, WJ_NE1, WJ_NE0, WJ_EE2, WJ_EE1, WJ_EE0,
       // hoisted first layers (replace WJ_*0A / WJ_*0B): edge columns contract over edges ...
       WJ_HEU0E, WJ_HFU0E, WJ_HPA0E,
       // ... node columns over nodes (G = column blocks of dT)
       WJ_HEU0XI, WJ_HEU0XJ, WJ_HFU0X, WJ_HFU0X0, WJ_HPA0X, WJ_HPA0X0, WJ_COUNT };
constexpr int kWjBig = WJ_C3;              // jobs [0, kWjBig) and the WJ_H*E jobs share ~2048 tasks in proportion to their work
constexpr int kHoistNodeRowsPerTask = 64;
// Column blocks of edge_update.0 / create_*_msgs.0 gradients that the hoisted jobs produce separately
// (own slab each: node- and edge-level jobs have different task counts)
enum { VL_EU0E, VL_FU0E, VL_PA0E, VL_EU0XI, VL_EU0XJ, VL_FU0X, VL_FU0X0, VL_PA0X, VL_PA0X0, VL_COUNT };
constexpr int kNarrowRowsPerTask = 512;
constexpr int kWsTaskCap = 65536;          // entries of the task -> job table in the workspace
struct WsPlan {
  int shape[WJ_COUNT], lin[WJ_COUNT], rows[WJ_COUNT], nvar[WJ_COUNT], rows_per_task[WJ_COUNT], ntasks[WJ_COUNT];
};
static WsPlan ws_plan(int N, int E, int depth, bool layer_mode, bool hoist);

static WsPlan ws_plan(int N, int E, int depth, bool layer_mode, bool hoist) {
  WsPlan p;
  const int shp[WJ_COUNT] = {WS_96_48_16, WS_96_32_32, WS_64_96, WS_32_64, WS_96_48_16, WS_96_48_16, WS_64_96,
                             WS_96_48_16, WS_96_48_16, WS_64_96, WS_96_64, WS_96_64, WS_64_96, WS_48_64,
                             WS_16_16, WS_16_16, WS_16_16, WS_16_32, WS_48_48, WS_48_32, WS_32_32, WS_32_16, WS_16_16, WS_16_16,
                             WS_96_32, WS_96_32, WS_96_32, WS_96_48, WS_96_48, WS_96_48, WS_96_48, WS_96_48, WS_96_48};
  // lin: index into PoseWs::lin for [0, WJ_HEU0E), into PoseWs::vlin for the hoisted jobs
  const int lin[WJ_COUNT] = {LIN_EU0, LIN_EU0, LIN_EU1, LIN_EU2, LIN_PA0, LIN_PA0, LIN_PA1,
                             LIN_FU0, LIN_FU0, LIN_FU1, LIN_CF0, LIN_CF0, LIN_CF1, LIN_CF2,
                             LIN_C3, LIN_C2, LIN_C1, LIN_C0, LIN_NE2, LIN_NE1, LIN_NE0, LIN_EE2, LIN_EE1, LIN_EE0,
                             VL_EU0E, VL_FU0E, VL_PA0E, VL_EU0XI, VL_EU0XJ, VL_FU0X, VL_FU0X0, VL_PA0X, VL_PA0X0};
  auto proportional = [](int i) { return i < kWjBig || (i >= WJ_HEU0E && i <= WJ_HPA0E); };
  double total = 0;
  for (int i = 0; i < WJ_COUNT; ++i) {
    p.shape[i] = shp[i];
    p.lin[i] = lin[i];
    const bool on_nodes = (i >= WJ_CF0A && i <= WJ_CF2) || (i >= WJ_NE2 && i <= WJ_NE0) || i >= WJ_HEU0XI;
    p.rows[i] = on_nodes ? N : E;
    const bool first_layer_unsplit = i == WJ_EU0A || i == WJ_EU0B || i == WJ_PA0A || i == WJ_PA0B || i == WJ_FU0A || i == WJ_FU0B;
    if (i >= WJ_HEU0E) p.nvar[i] = !hoist ? 0 : (i == WJ_HEU0E || i == WJ_HEU0XI || i == WJ_HEU0XJ) ? depth : depth - 1;
    else if (i >= kWjBig) p.nvar[i] = layer_mode ? 0 : 1;
    else if (hoist && first_layer_unsplit) p.nvar[i] = 0;
    else p.nvar[i] = layer_mode ? 1 : (i <= WJ_EU2) ? depth : depth - 1;   // standalone layer: every stack once
    if (proportional(i)) total += (double)p.rows[i] * ws_shape_blocks(shp[i]) * (p.nvar[i] > 0 ? p.nvar[i] : 0);
  }
  for (int i = 0; i < WJ_COUNT; ++i) {
    long rpt;
    if (proportional(i)) {
      const double work = (double)p.rows[i] * ws_shape_blocks(shp[i]) * (p.nvar[i] > 0 ? p.nvar[i] : 0);
      long t = (long)(2048.0 * work / (total > 0 ? total : 1) + 0.5);   // measured flat between 1,536 and 3,072 tasks
      const long maxt = (p.rows[i] + 15) / 16;
      if (t > maxt) t = maxt;
      if (t < 1) t = 1;
      rpt = ((p.rows[i] + t - 1) / t + 3) / 4 * 4;
      if (rpt < 4) rpt = 4;
    } else if (i >= WJ_HEU0XI) {
      rpt = kHoistNodeRowsPerTask;           // N rows x depth layers per task
    } else {
      rpt = kNarrowRowsPerTask;              // load-latency bound tasks: a fixed, short row range each
    }
    p.rows_per_task[i] = (int)rpt;
    p.ntasks[i] = (int)((p.rows[i] + rpt - 1) / rpt);
    if (p.ntasks[i] < 1) p.ntasks[i] = 1;
  }
  return p;
}

constexpr uint32_t kFlagLayerMode = 0x80000000u;   // internal: workspace of the standalone layer operator

struct PoseWs {
  // packed weight images
  float *wp_ee, *wp_ne, *wp_cls, *wp_efwd, *wp_nfwd, *wp_ebwd, *wp_ebwd_nm, *wp_nbwd, *wp_clsT, *wp_eeT, *wp_neT;
  // hoisted first layers: projection images, per-node tables
  bool hoist;
  float *wp_ne_h, *wp_nfwd_h, *wp_efwd_h, *wp_ebwd_h, *wp_ebwd_nm_h, *wp_gproj, *wp_nbwd_h;
  float *dT, *gx;       // [depth][N, TW] gradient of T per layer (kept for the weight gradient), [N, 2 DX] scratch
  float *T, *T0;        // [N, TW] (current layer), [N, 2 MH] (x0 terms, whole forward)
  // encoder / classifier activations
  float *ea_pad, *ee_a1, *ee_a2, *pose_pad, *ne_a1, *ne_a2, *c_a1, *c_a2, *c_a3;
  float* x[16];      // x[0] = x_enc ... x[depth]
  float* e[16];      // e[0] = encoded edge_attr ... e[depth]
  float *sH1[16], *sH2[16], *sF1[16], *sP1[16], *M[16], *nH1[16], *nH2[16];
  float* rmask[16];       // hoisted plan: ReLU masks of sH1 | sH2 | sF1 | sP1, 64 bytes per edge and layer (b3d_hoist.hpp)
  float *fut, *past;
  // backward scratch
  float *de[2], *gdst, *gsrc, *dx0_acc;
  // per-layer G tensors (kept for the single weight-gradient launch after the sweep); index 0 is
  // layer 0 and consecutive layers are `*_ls` floats apart
  float *dM, *GdH1, *GdH2, *Gde, *GdF1, *GdP1, *Gdx, *GnH2, *GnH1;
  WsPlan plan;
  float* zrow;          // 256 zero floats
  WsJob* ws_table;      // device job table of the streaming weight gradient
  int* ws_task_job;
  int* iota;            // 0, 1, 2, ... (identity gather for the streaming weight gradient)
  int iota_n;
  float *gc_top, *gc3, *gc2, *gc1, *ge2, *ge1, *gn_top, *gn2, *gn1;
  LinSlab lin[LIN_COUNT];
  LinSlab vlin[VL_COUNT];   // hoisted first layers: column blocks of edge_update.0 / create_*_msgs.0
  KnnWs knn;
  size_t bytes;
  bool ok;
};

static void carve(PoseWs& w, void* ws, size_t ws_bytes, int N, int E, int depth, uint32_t flags) {
  Carver c(ws, ws_bytes);
  const bool tr = flags & B3D_FLAG_TRAINING;
  const size_t e_ = (size_t)(E > 0 ? E : 1), n_ = (size_t)(N > 0 ? N : 1);
  memset(&w, 0, sizeof(w));
  w.wp_ee = c.take<float>(SeqEdgeEnc::TOTAL_FLOATS);
  w.wp_ne = c.take<float>(SeqNodeEnc::TOTAL_FLOATS);
  w.wp_cls = c.take<float>(SeqCls::TOTAL_FLOATS);
  w.wp_efwd = c.take<float>(D::EdgeFwdSeq::TOTAL_FLOATS);
  w.wp_nfwd = c.take<float>(D::NodeFwdSeq::TOTAL_FLOATS);
  w.hoist = !(flags & kFlagLayerMode);
  if (w.hoist) {
    w.wp_ne_h = c.take<float>(SeqNodeEncH::TOTAL_FLOATS);
    w.wp_nfwd_h = c.take<float>(NodeFwdHSeq<D>::TOTAL_FLOATS);
    w.wp_efwd_h = c.take<float>(HP::EdgeFwdSeq::TOTAL_FLOATS);
    w.T = c.take<float>(n_ * HP::TW);
    w.T0 = c.take<float>(n_ * 2 * D::MH);
  }
  if (!tr) {
    w.x[0] = c.take<float>(n_ * D::DX);
    w.e[0] = c.take<float>(e_ * D::DE);
  }
  w.fut = c.take<float>(e_ * D::DM);
  w.past = c.take<float>(e_ * D::DM);
  if (!tr) {
    // inference: ping-pong x / e, nothing saved
    w.x[1] = c.take<float>(n_ * D::DX);
    w.x[2] = c.take<float>(n_ * D::DX);
    w.e[1] = c.take<float>(e_ * D::DE);
    w.e[2] = c.take<float>(e_ * D::DE);
    for (int l = 3; l <= depth; ++l) { w.x[l] = w.x[1 + (l - 1) % 2]; w.e[l] = w.e[1 + (l - 1) % 2]; }
  } else {
    w.wp_ebwd = c.take<float>(D::EdgeBwdSeq::TOTAL_FLOATS);
    w.wp_ebwd_nm = c.take<float>(D::EdgeBwdSeqNoMsg::TOTAL_FLOATS);
    w.wp_nbwd = c.take<float>(D::NodeBwdSeq::TOTAL_FLOATS);
    if (w.hoist) {
      w.wp_ebwd_h = c.take<float>(HP::EdgeBwdSeq::TOTAL_FLOATS);
      w.wp_ebwd_nm_h = c.take<float>(HP::EdgeBwdSeqNoMsg::TOTAL_FLOATS);
      w.wp_gproj = c.take<float>(HP::GradProjSeq::TOTAL_FLOATS);
      w.wp_nbwd_h = c.take<float>(NodeBwdHSeq<D>::TOTAL_FLOATS);
      w.dT = c.take<float>((size_t)depth * n_ * HP::GW);
      w.gx = c.take<float>(n_ * 2 * D::DX);
    }
    w.wp_clsT = c.take<float>(SeqClsT::TOTAL_FLOATS);
    w.wp_eeT = c.take<float>(SeqEdgeEncT::TOTAL_FLOATS);
    w.wp_neT = c.take<float>(SeqNodeEncT::TOTAL_FLOATS);
    w.ea_pad = c.take<float>(e_ * 16);
    w.ee_a1 = c.take<float>(e_ * 16);
    w.ee_a2 = c.take<float>(e_ * 16);
    w.pose_pad = c.take<float>(n_ * 32);
    w.ne_a1 = c.take<float>(n_ * 32);
    w.ne_a2 = c.take<float>(n_ * 48);
    w.c_a1 = c.take<float>(e_ * 16);
    w.c_a2 = c.take<float>(e_ * 16);
    w.c_a3 = c.take<float>(e_ * 16);
    // x[0..depth] / e[0..depth]: contiguous, uniform layer stride (the streaming weight gradient
    // walks layers by pointer stride)
    {
      float* xb = c.take<float>((size_t)(depth + 1) * n_ * D::DX);
      float* eb = c.take<float>((size_t)(depth + 1) * e_ * D::DE);
      for (int l = 0; l <= depth; ++l) {
        w.x[l] = xb ? xb + (size_t)l * n_ * D::DX : nullptr;
        w.e[l] = eb ? eb + (size_t)l * e_ * D::DE : nullptr;
      }
    }
    auto per_layer = [&](float** arr, size_t per) {
      float* b = c.take<float>((size_t)depth * per);
      for (int l = 0; l < depth; ++l) arr[l] = b ? b + (size_t)l * per : nullptr;
    };
    per_layer(w.sH1, e_ * D::EH1);
    per_layer(w.sH2, e_ * D::EH2);
    per_layer(w.sF1, e_ * D::MH);
    per_layer(w.sP1, e_ * D::MH);
    per_layer(w.rmask, e_ * 16);
    per_layer(w.M, n_ * D::NIN);
    per_layer(w.nH1, n_ * D::NH1);
    per_layer(w.nH2, n_ * D::NH2);
    w.de[0] = c.take<float>(e_ * D::DE);
    w.de[1] = c.take<float>(e_ * D::DE);
    w.gdst = c.take<float>(e_ * 2 * D::DX);
    w.gsrc = c.take<float>(e_ * 2 * D::DX);
    w.dx0_acc = c.take<float>(n_ * D::DX);
    w.dM = c.take<float>((size_t)depth * n_ * D::NIN);
    w.GdH1 = c.take<float>((size_t)depth * e_ * D::EH1);
    w.GdH2 = c.take<float>((size_t)depth * e_ * D::EH2);
    w.Gde = c.take<float>((size_t)depth * e_ * D::DE);
    w.GdF1 = c.take<float>((size_t)depth * e_ * D::MH);
    w.GdP1 = c.take<float>((size_t)depth * e_ * D::MH);
    w.Gdx = c.take<float>((size_t)depth * n_ * D::DX);
    w.GnH2 = c.take<float>((size_t)depth * n_ * D::NH2);
    w.GnH1 = c.take<float>((size_t)depth * n_ * D::NH1);
    w.gc_top = c.take<float>(e_ * 16);
    w.gc3 = c.take<float>(e_ * 16);
    w.gc2 = c.take<float>(e_ * 16);
    w.gc1 = c.take<float>(e_ * 16);
    w.ge2 = c.take<float>(e_ * 16);
    w.ge1 = c.take<float>(e_ * 16);
    w.gn_top = c.take<float>(n_ * 48);
    w.gn2 = c.take<float>(n_ * 48);
    w.gn1 = c.take<float>(n_ * 32);
    w.iota_n = (E > N ? E : N) + 64;
    w.iota = c.take<int>((size_t)w.iota_n);
    w.zrow = c.take<float>(256);
    w.ws_table = c.take<WsJob>(48);
    w.ws_task_job = c.take<int>(kWsTaskCap);
    // weight-gradient slabs; chunk / task counts are a pure function of (N, E, depth)
    w.plan = ws_plan(N, E, depth, (flags & kFlagLayerMode) != 0, w.hoist);
    for (int i = 0; i < LIN_COUNT; ++i) {
      LinSlab& ls = w.lin[i];
      ls.N = kLinDims[i].N; ls.K = kLinDims[i].K;
      ls.NP = pad16(ls.N); ls.KP = pad16(ls.K);
      const long rows = kLinOnEdges[i] ? E : N;
      ls.nchunks = wg_nchunks(rows, ls.NP, ls.KP, 0);
      for (int jx = WJ_COUNT - 1; jx >= 0; --jx)
        if (w.plan.lin[jx] == i) ls.nchunks = w.plan.ntasks[jx];
      ls.slab = c.take<float>(wg_slab_floats(ls.nchunks, ls.NP, ls.KP));
      ls.used = false;
    }
    for (int v = 0; v < VL_COUNT; ++v) {
      LinSlab& ls = w.vlin[v];
      ls.N = D::EH1; ls.K = (v <= VL_PA0E) ? D::DE : D::DX;
      ls.NP = pad16(ls.N); ls.KP = pad16(ls.K);
      ls.nchunks = w.plan.ntasks[WJ_HEU0E + v];
      ls.slab = w.hoist ? c.take<float>(wg_slab_floats(ls.nchunks, ls.NP, ls.KP)) : nullptr;
      ls.used = false;
    }
  }
  if (flags & B3D_FLAG_RUN_DEAD_KNN) knn_carve(w.knn, c, N, D::DX);
  w.bytes = c.off + 256;
  w.ok = c.ok();
}

// ---- forward ------------------------------------------------------------------------------------
static int pack_forward(const b3d_pose_weights* pw, PoseWs& w, bool training, bool knn, hipStream_t stream) {
  PackDesc d[160];
  int n = 0;
  const b3d_linear* ee = pw->edge_encoder;
  const b3d_linear* ne = pw->node_encoder;
  const b3d_linear* cl = pw->edge_classifier;
  const b3d_mp_weights& mp = pw->mp;
  for (int i = 0; i < 3; ++i) d[n++] = pack_desc<SeqEdgeEnc>(i, w.wp_ee, ee[i].w, ee[i].b, kLinDims[LIN_EE0 + i].N, kLinDims[LIN_EE0 + i].K, false);
  for (int i = 0; i < 3; ++i) d[n++] = pack_desc<SeqNodeEnc>(i, w.wp_ne, ne[i].w, ne[i].b, kLinDims[LIN_NE0 + i].N, kLinDims[LIN_NE0 + i].K, false);
  for (int i = 0; i < 4; ++i) d[n++] = pack_desc<SeqCls>(i, w.wp_cls, cl[i].w, cl[i].b, kLinDims[LIN_C0 + i].N, kLinDims[LIN_C0 + i].K, false);
  using EF = D::EdgeFwdSeq;
  for (int i = 0; i < 3; ++i) d[n++] = pack_desc<EF>(i, w.wp_efwd, mp.edge_update[i].w, mp.edge_update[i].b, kLinDims[LIN_EU0 + i].N, kLinDims[LIN_EU0 + i].K, false);
  for (int i = 0; i < 2; ++i) d[n++] = pack_desc<EF>(3 + i, w.wp_efwd, mp.create_future_msgs[i].w, mp.create_future_msgs[i].b, kLinDims[LIN_FU0 + i].N, kLinDims[LIN_FU0 + i].K, false);
  for (int i = 0; i < 2; ++i) d[n++] = pack_desc<EF>(5 + i, w.wp_efwd, mp.create_past_msgs[i].w, mp.create_past_msgs[i].b, kLinDims[LIN_PA0 + i].N, kLinDims[LIN_PA0 + i].K, false);
  for (int i = 0; i < 3; ++i) d[n++] = pack_desc<D::NodeFwdSeq>(i, w.wp_nfwd, mp.combine_future_past[i].w, mp.combine_future_past[i].b, kLinDims[LIN_CF0 + i].N, kLinDims[LIN_CF0 + i].K, false);
  if (w.hoist) {
    constexpr int DX = D::DX, DE = D::DE, EIN = D::EIN, MIN = D::MIN, H1 = D::EH1, MH = D::MH;
    const b3d_linear &eu0 = mp.edge_update[0], &fu0 = mp.create_future_msgs[0], &pa0 = mp.create_past_msgs[0];
    // the projection image sits behind the node encoder (layer 0) and behind the node update (layers 1..)
    auto proj = [&](auto tag, int li, float* base) {
      using S = decltype(tag);
      d[n++] = pack_slice<S>(li, base, eu0.w, eu0.b, H1, DX, EIN, HP::OA, H1, false);             // x[dst] columns + bias
      d[n++] = pack_slice<S>(li, base, eu0.w + DX, nullptr, H1, DX, EIN, HP::OB, H1, false);     // x[src] columns
      d[n++] = pack_slice<S>(li, base, fu0.w, fu0.b, MH, DX, MIN, HP::OF, MH, false);             // future: x[dst]
      d[n++] = pack_slice<S>(li, base, pa0.w, pa0.b, MH, DX, MIN, HP::OP, MH, false);             // past:   x[src]
      d[n++] = pack_slice<S>(li, base, knn ? pw->knn_conv.lin : nullptr, nullptr, DX, DX, DX, HP::OG, DX, false);   // GATConv.lin
    };
    d[n++] = pack_slice<SeqNodeEncH>(3, w.wp_ne_h, fu0.w + DX + DE, nullptr, MH, DX, MIN, 0, MH, false);    // x0 columns
    d[n++] = pack_slice<SeqNodeEncH>(3, w.wp_ne_h, pa0.w + DX + DE, nullptr, MH, DX, MIN, MH, MH, false);
    proj(SeqNodeEncH{}, 4, w.wp_ne_h);
    for (int i = 0; i < 3; ++i) d[n++] = pack_desc<NodeFwdHSeq<D>>(i, w.wp_nfwd_h, mp.combine_future_past[i].w, mp.combine_future_past[i].b, kLinDims[LIN_CF0 + i].N, kLinDims[LIN_CF0 + i].K, false);
    proj(NodeFwdHSeq<D>{}, 3, w.wp_nfwd_h);
    using EH = HP::EdgeFwdSeq;
    d[n++] = pack_slice<EH>(0, w.wp_efwd_h, eu0.w + 2 * DX, nullptr, H1, DE, EIN, 0, H1, false);    // edge columns of edge_update.0
    for (int i = 1; i < 3; ++i) d[n++] = pack_desc<EH>(i, w.wp_efwd_h, mp.edge_update[i].w, mp.edge_update[i].b, kLinDims[LIN_EU0 + i].N, kLinDims[LIN_EU0 + i].K, false);
    d[n++] = pack_slice<EH>(3, w.wp_efwd_h, fu0.w + DX, nullptr, MH, DE, MIN, 0, MH, false);         // e' columns
    d[n++] = pack_desc<EH>(4, w.wp_efwd_h, mp.create_future_msgs[1].w, mp.create_future_msgs[1].b, kLinDims[LIN_FU1].N, kLinDims[LIN_FU1].K, false);
    d[n++] = pack_slice<EH>(5, w.wp_efwd_h, pa0.w + DX, nullptr, MH, DE, MIN, 0, MH, false);
    d[n++] = pack_desc<EH>(6, w.wp_efwd_h, mp.create_past_msgs[1].w, mp.create_past_msgs[1].b, kLinDims[LIN_PA1].N, kLinDims[LIN_PA1].K, false);
  }
  if (knn) {                              // GATConv lin of the discarded k-NN block (consumed on the side stream)
    d[n++] = pack_desc<LayerSeq<L<D::DX, D::DX>>>(0, w.knn.wp, pw->knn_conv.lin, nullptr, D::DX, D::DX, false);
    w.knn.packed = true;
  }
  if (training) {
    d[n++] = fill_desc(w.iota, w.iota_n, true);       // identity gather + zero row of the streaming weight gradient
    d[n++] = fill_desc(w.zrow, 256, false);
    // transposed images: image rows = forward inputs (K), image cols = forward outputs (N)
    auto T = [&](auto seq_tag, int li, float* base, const b3d_linear& l, int lin) {
      using S = decltype(seq_tag);
      d[n++] = pack_desc<S>(li, base, l.w, nullptr, kLinDims[lin].K, kLinDims[lin].N, true);
    };
    using EB = D::EdgeBwdSeq;
    T(EB{}, 0, w.wp_ebwd, mp.create_past_msgs[1], LIN_PA1);
    T(EB{}, 1, w.wp_ebwd, mp.create_past_msgs[0], LIN_PA0);
    T(EB{}, 2, w.wp_ebwd, mp.create_future_msgs[1], LIN_FU1);
    T(EB{}, 3, w.wp_ebwd, mp.create_future_msgs[0], LIN_FU0);
    T(EB{}, 4, w.wp_ebwd, mp.edge_update[2], LIN_EU2);
    T(EB{}, 5, w.wp_ebwd, mp.edge_update[1], LIN_EU1);
    T(EB{}, 6, w.wp_ebwd, mp.edge_update[0], LIN_EU0);
    using EN = D::EdgeBwdSeqNoMsg;
    T(EN{}, 0, w.wp_ebwd_nm, mp.edge_update[2], LIN_EU2);
    T(EN{}, 1, w.wp_ebwd_nm, mp.edge_update[1], LIN_EU1);
    T(EN{}, 2, w.wp_ebwd_nm, mp.edge_update[0], LIN_EU0);
    using NB = D::NodeBwdSeq;
    T(NB{}, 0, w.wp_nbwd, mp.combine_future_past[2], LIN_CF2);
    T(NB{}, 1, w.wp_nbwd, mp.combine_future_past[1], LIN_CF1);
    T(NB{}, 2, w.wp_nbwd, mp.combine_future_past[0], LIN_CF0);
    T(SeqClsT{}, 0, w.wp_clsT, cl[3], LIN_C3);
    T(SeqClsT{}, 1, w.wp_clsT, cl[2], LIN_C2);
    T(SeqClsT{}, 2, w.wp_clsT, cl[1], LIN_C1);
    T(SeqClsT{}, 3, w.wp_clsT, cl[0], LIN_C0);
    T(SeqEdgeEncT{}, 0, w.wp_eeT, ee[2], LIN_EE2);
    T(SeqEdgeEncT{}, 1, w.wp_eeT, ee[1], LIN_EE1);
    T(SeqNodeEncT{}, 0, w.wp_neT, ne[2], LIN_NE2);
    T(SeqNodeEncT{}, 1, w.wp_neT, ne[1], LIN_NE1);
    if (w.hoist) {
      constexpr int DX = D::DX, DE = D::DE, EIN = D::EIN, MIN = D::MIN, H1 = D::EH1;
      const b3d_linear &eu0 = mp.edge_update[0], &fu0 = mp.create_future_msgs[0], &pa0 = mp.create_past_msgs[0];
      // data-gradient images: full transposes, except the .0 layers, which keep their edge columns
      auto TS = [&](auto tag, int li, float* base, const float* wcol, int ld) {      // [DE rows (inputs), 96 cols (outputs)]
        using S = decltype(tag);
        d[n++] = pack_slice<S>(li, base, wcol, nullptr, DE, H1, ld, 0, S::np(li), true);
      };
      using EB2 = HP::EdgeBwdSeq;
      T(EB2{}, 0, w.wp_ebwd_h, mp.create_past_msgs[1], LIN_PA1);
      TS(EB2{}, 1, w.wp_ebwd_h, pa0.w + DX, MIN);
      T(EB2{}, 2, w.wp_ebwd_h, mp.create_future_msgs[1], LIN_FU1);
      TS(EB2{}, 3, w.wp_ebwd_h, fu0.w + DX, MIN);
      T(EB2{}, 4, w.wp_ebwd_h, mp.edge_update[2], LIN_EU2);
      T(EB2{}, 5, w.wp_ebwd_h, mp.edge_update[1], LIN_EU1);
      TS(EB2{}, 6, w.wp_ebwd_h, eu0.w + 2 * DX, EIN);
      using EN2 = HP::EdgeBwdSeqNoMsg;
      T(EN2{}, 0, w.wp_ebwd_nm_h, mp.edge_update[2], LIN_EU2);
      T(EN2{}, 1, w.wp_ebwd_nm_h, mp.edge_update[1], LIN_EU1);
      TS(EN2{}, 2, w.wp_ebwd_nm_h, eu0.w + 2 * DX, EIN);
      // (dx | dx0) = sum over the four lists of (node columns)^T . dT_list; rows 0:DX from the x columns,
      // rows DX:2DX (future / past only) from the x0 columns.  Standalone (layer 0) and in front of the node MLP.
      auto gp = [&](auto tag, int li0, float* base) {
        using S = decltype(tag);
        d[n++] = pack_slice<S>(li0 + 0, base, eu0.w, nullptr, DX, H1, EIN, 0, DX, true);
        d[n++] = pack_slice<S>(li0 + 1, base, eu0.w + DX, nullptr, DX, H1, EIN, 0, DX, true);
        d[n++] = pack_slice<S>(li0 + 2, base, fu0.w, nullptr, DX, H1, MIN, 0, DX, true);
        d[n++] = pack_slice<S>(li0 + 2, base, fu0.w + DX + DE, nullptr, DX, H1, MIN, DX, DX, true);
        d[n++] = pack_slice<S>(li0 + 3, base, pa0.w, nullptr, DX, H1, MIN, 0, DX, true);
        d[n++] = pack_slice<S>(li0 + 3, base, pa0.w + DX + DE, nullptr, DX, H1, MIN, DX, DX, true);
      };
      gp(HP::GradProjSeq{}, 0, w.wp_gproj);
      gp(NodeBwdHSeq<D>{}, 0, w.wp_nbwd_h);
      T(NodeBwdHSeq<D>{}, 4, w.wp_nbwd_h, mp.combine_future_past[2], LIN_CF2);
      T(NodeBwdHSeq<D>{}, 5, w.wp_nbwd_h, mp.combine_future_past[1], LIN_CF1);
      T(NodeBwdHSeq<D>{}, 6, w.wp_nbwd_h, mp.combine_future_past[0], LIN_CF0);
    }
  }
  return pack_images(d, n, stream);
}

// Activation / gradient sources of the message-passing weight gradient.  In the whole-model backward
// every pointer is layer 0 of a per-layer array (consecutive layers `*s` floats apart, see carve());
// the standalone layer operator passes the caller's tensors (one variant, strides unused).
struct MpGradSrc {
  const float *x, *x0, *e_in, *e_out;
  long xs, es;
  const float *GdH1, *GdH2, *Gde, *GdP1, *GdF1, *dM, *GnH1, *GnH2, *Gdx;
  const float *sH1, *sH2, *sP1, *sF1, *M, *nH1, *nH2;
  const float* dT;          // hoisted first layers: [depth][N, TW] per-node gradient of T
  const float* e_last;      // e[depth]: input of the classifier (whole model only)
  const float* de0;         // gradient of e[0] = G of edge_encoder.4 (whole model only)
  bool have_logit_grad;
};

static int mp_weight_grads(PoseWs& w, const MpGradSrc& ms, int N, int E, const int* src, const int* dst, hipStream_t stream) {
  const size_t eL1 = (size_t)E * D::EH1, eL2 = (size_t)E * D::EH2, eLe = (size_t)E * D::DE, eLm = (size_t)E * D::MH;
  const size_t nLm = (size_t)N * D::NIN, nLx = (size_t)N * D::DX, nL1 = (size_t)N * D::NH1, nL2 = (size_t)N * D::NH2;
  (void)eLe; (void)nLx;
  WsLauncher wl;
  wl.begin(w.ws_table, 48, w.ws_task_job, kWsTaskCap, stream);
  // w.iota / w.zrow were filled by the forward's pack launch
  const int* iota = w.iota;
  auto sg = [iota](const float* p, const int* idx, long vstride, int stride, int col0) {
    WsSeg s; s.ptr = p; s.idx = idx ? idx : iota; s.vstride = vstride; s.stride = stride; s.col0 = col0; return s;
  };
  const WsSeg none = sg(nullptr, nullptr, 0, 0, 0);
  auto add = [&](int wj, const WsSeg& gseg, const WsSeg& a0, int c0, const WsSeg& a1, int c1, bool bias) {
    if (w.plan.nvar[wj] <= 0) return;
    LinSlab& ls = (wj >= WJ_HEU0E) ? w.vlin[w.plan.lin[wj]] : w.lin[w.plan.lin[wj]];
    WsJob jb;
    memset(&jb, 0, sizeof(jb));
    jb.g = gseg; jb.act[0] = a0; jb.act[1] = a1; jb.act[2] = a1;
    jb.wcol[0] = c0; jb.wcol[1] = c1; jb.wcol[2] = 0; jb.wrow = 0; jb.write_bias = bias ? 1 : 0;
    jb.shape = w.plan.shape[wj]; jb.rows = w.plan.rows[wj]; jb.nvar = w.plan.nvar[wj];
    jb.rows_per_task = w.plan.rows_per_task[wj]; jb.ntasks = w.plan.ntasks[wj];
    jb.NP = ls.NP; jb.KP = ls.KP; jb.slab = ls.slab;
    wl.add(jb);
    ls.used = true;
  };
  const float* x0 = ms.x0;
  auto xrow = [&](const int* idx, int c0) { return sg(ms.x, idx, ms.xs, D::DX, c0); };      // x[l][idx], columns c0..
  auto x0row = [&](const int* idx) { return sg(x0, idx, 0, D::DX, 0); };
  auto erow = [&](int l0, int c0) { return sg(l0 ? ms.e_out : ms.e_in, nullptr, ms.es, D::DE, c0); };         // e[l + l0], columns c0..
  // edge_update.0 (every layer): dW columns [x[dst] 0:48 | x[src] 48:96 | e 96:128]
  const WsSeg gH1 = sg(ms.GdH1, nullptr, eL1, D::EH1, 0);
  add(WJ_EU0A, gH1, xrow(dst, 0), 0, xrow(src, 0), 48, true);         // [x[dst] | x[src][0:16]]
  add(WJ_EU0B, gH1, xrow(src, 16), 64, erow(0, 0), 96, false);        // [x[src][16:48] | e]
  // message stacks .0 (layers 0 .. depth-2): columns [x[.] 0:48 | e' 48:80 | x0[.] 80:128]
  const WsSeg gP1 = sg(ms.GdP1, nullptr, eLm, D::MH, 0), gF1 = sg(ms.GdF1, nullptr, eLm, D::MH, 0);
  add(WJ_PA0A, gP1, xrow(src, 0), 0, erow(1, 0), 48, true);           // [x[src] | e'[0:16]]
  add(WJ_PA0B, gP1, x0row(src), 80, erow(1, 16), 64, false);          // [x0[src] | e'[16:32]]
  add(WJ_FU0A, gF1, xrow(dst, 0), 0, erow(1, 0), 48, true);
  add(WJ_FU0B, gF1, x0row(dst), 80, erow(1, 16), 64, false);
  // node update .0 (layers 0 .. depth-2)
  const WsSeg gN1 = sg(ms.GnH1, nullptr, nL1, D::NH1, 0);
  add(WJ_CF0A, gN1, sg(ms.M, nullptr, nLm, D::NIN, 0), 0, none, 0, true);
  add(WJ_CF0B, gN1, sg(ms.M, nullptr, nLm, D::NIN, D::DM), D::DM, none, 0, false);
  // single-job matrices
  add(WJ_EU1, sg(ms.GdH2, nullptr, eL2, D::EH2, 0), sg(ms.sH1, nullptr, eL1, D::EH1, 0), 0, none, 0, true);
  add(WJ_EU2, sg(ms.Gde, nullptr, eLe, D::DE, 0), sg(ms.sH2, nullptr, eL2, D::EH2, 0), 0, none, 0, true);
  add(WJ_PA1, sg(ms.dM, dst, nLm, D::NIN, 0), sg(ms.sP1, nullptr, eLm, D::MH, 0), 0, none, 0, true);
  add(WJ_FU1, sg(ms.dM, src, nLm, D::NIN, D::DM), sg(ms.sF1, nullptr, eLm, D::MH, 0), 0, none, 0, true);
  add(WJ_CF1, sg(ms.GnH2, nullptr, nL2, D::NH2, 0), sg(ms.nH1, nullptr, nL1, D::NH1, 0), 0, none, 0, true);
  add(WJ_CF2, sg(ms.Gdx, nullptr, nLx, D::DX, 0), sg(ms.nH2, nullptr, nL2, D::NH2, 0), 0, none, 0, true);
  // hoisted first layers (nvar == 0 otherwise): edge columns over edges, node columns over nodes
  {
    const long tLs = (long)N * HP::GW;
    auto tcol = [&](int off) { return sg(ms.dT, nullptr, tLs, HP::GW, off); };
    add(WJ_HEU0E, gH1, erow(0, 0), 0, none, 0, true);
    add(WJ_HFU0E, gF1, erow(1, 0), 0, none, 0, true);
    add(WJ_HPA0E, gP1, erow(1, 0), 0, none, 0, true);
    add(WJ_HEU0XI, tcol(HP::OA), xrow(nullptr, 0), 0, none, 0, false);
    add(WJ_HEU0XJ, tcol(HP::OB), xrow(nullptr, 0), 0, none, 0, false);
    add(WJ_HFU0X, tcol(HP::OF), xrow(nullptr, 0), 0, none, 0, false);
    add(WJ_HFU0X0, tcol(HP::OF), x0row(nullptr), 0, none, 0, false);
    add(WJ_HPA0X, tcol(HP::OP), xrow(nullptr, 0), 0, none, 0, false);
    add(WJ_HPA0X0, tcol(HP::OP), x0row(nullptr), 0, none, 0, false);
  }
  // narrow stacks (whole model only; nvar == 0 in layer mode).  Row strides pad every width to 16, the
  // padding columns only reach slab entries outside [N, K], which the reduce never reads.
  auto narrow = [&](int wj, const float* gp, int gstride, const float* ap, int astride) {
    add(wj, sg(gp, nullptr, 0, gstride, 0), sg(ap, nullptr, 0, astride, 0), 0, none, 0, true);
  };
  if (ms.have_logit_grad) narrow(WJ_C3, w.gc_top, 16, w.c_a3, 16);       // edge_classifier.6  [1,4]
  narrow(WJ_C2, w.gc3, 16, w.c_a2, 16);                                  // .4  [4,8]
  narrow(WJ_C1, w.gc2, 16, w.c_a1, 16);                                  // .2  [8,16]
  narrow(WJ_C0, w.gc1, 16, ms.e_last, D::DE);                            // .0  [16,32]
  narrow(WJ_NE2, w.gn_top, 48, w.ne_a2, 48);                             // node_encoder.4  [48,36]
  narrow(WJ_NE1, w.gn2, 48, w.ne_a1, 32);                                // .2  [36,24]
  narrow(WJ_NE0, w.gn1, 32, w.pose_pad, 32);                             // .0  [24,19]
  narrow(WJ_EE2, ms.de0, D::DE, w.ee_a2, 16);                            // edge_encoder.4  [32,16]
  narrow(WJ_EE1, w.ge2, 16, w.ee_a1, 16);                                // .2  [16,8]
  narrow(WJ_EE0, w.ge1, 16, w.ea_pad, 16);                               // .0  [8,4]
  if (w.hoist) {             // every job of the hoisted plan has one activation segment: LDS-DMA ring form
    B3D_REQUIRE(wl.launch2(wstream2_kernel, kWs2LdsBytes, w.zrow, w.iota, B3D_K_WGRAD_EDGE) == 0, "wstream2: LDS attribute");
  } else {
    wl.launch(w.zrow, B3D_K_WGRAD_EDGE);
  }
  B3D_REQUIRE(wl.status == 0, "streaming weight gradient: job table overflow");
  B3D_TRY(launch_check("wstream_kernel"));
  return B3D_OK;
}

static int check_weights(const b3d_pose_weights* pw) {
  B3D_REQUIRE(pw != nullptr, "weights struct is null");
  const b3d_linear* all[] = {pw->edge_encoder, pw->node_encoder, pw->edge_classifier, pw->mp.edge_update,
                             pw->mp.create_past_msgs, pw->mp.create_future_msgs, pw->mp.combine_future_past};
  const int cnt[] = {3, 3, 4, 3, 2, 2, 3};
  for (int g = 0; g < 7; ++g)
    for (int i = 0; i < cnt[g]; ++i)
      B3D_REQUIRE(all[g][i].w != nullptr && all[g][i].b != nullptr, "null weight/bias pointer (group %d layer %d)", g, i);
  return B3D_OK;
}

}  // namespace b3d

using namespace b3d;

extern "C" size_t b3d_pose_workspace_bytes(int32_t N, int32_t E, int32_t depth, uint32_t flags) {
  if (depth < 1 || depth > 15) return 0;
  PoseWs w;
  carve(w, nullptr, 0, N, E, depth, flags);
  return w.bytes;
}

extern "C" int b3d_pose_debug_layer_ptrs(void* workspace, size_t workspace_bytes, int32_t N, int32_t E,
                                         int32_t depth, uint32_t flags, int32_t layer, float** x, float** e) {
  B3D_REQUIRE(depth >= 1 && depth <= 15 && layer >= 0 && layer <= depth, "bad layer");
  PoseWs w;
  carve(w, workspace, workspace_bytes, N, E, depth, flags);
  if (!w.ok) return fail(B3D_ERR_WORKSPACE, "workspace too small");
  *x = w.x[layer];
  *e = w.e[layer];
  return B3D_OK;
}

#ifdef B3D_EXP_STAMPS
extern "C" int b3d_debug_stamps(long long* host_dst) {
  return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(b3d::g_stamps), sizeof(long long) * 4 * 512 * 32) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int b3d_pose_debug_knn_ptrs(void* workspace, size_t workspace_bytes, int32_t N, int32_t E, int32_t depth,
                                       uint32_t flags, float** y, int32_t** nbr, int32_t** cnt) {
  B3D_REQUIRE(depth >= 1 && depth <= 15 && (flags & B3D_FLAG_RUN_DEAD_KNN), "no k-NN block in this workspace");
  PoseWs w;
  carve(w, workspace, workspace_bytes, N, E, depth, flags);
  if (!w.ok) return fail(B3D_ERR_WORKSPACE, "workspace too small");
  *y = w.knn.y; *nbr = w.knn.nbr; *cnt = w.knn.cnt;
  return B3D_OK;
}

extern "C" int b3d_pose_forward(const b3d_pose_weights* pw, const b3d_graph* g, const float* pose_feats,
                                const double* edge_attr, const int64_t* node_timestamps, int32_t depth,
                                uint32_t flags, void* workspace, size_t workspace_bytes, float* out_logits,
                                float* out_x_enc, b3d_stream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  B3D_TRY(check_weights(pw));
  B3D_REQUIRE(g && pose_feats && edge_attr && workspace && out_logits && out_x_enc, "b3d_pose_forward: null argument");
  B3D_REQUIRE(depth >= 1 && depth <= 15, "b3d_pose_forward: depth %d outside [1,15]", depth);
  const int N = g->N, E = g->E;
  B3D_REQUIRE(N > 0 && E > 0, "b3d_pose_forward: empty graph (N=%d, E=%d); callers skip these (predict.py:179)", N, E);
  const bool tr = flags & B3D_FLAG_TRAINING;
  PoseWs w;
  carve(w, workspace, workspace_bytes, N, E, depth, flags);
  if (!w.ok) return fail(B3D_ERR_WORKSPACE, "b3d_pose_forward: workspace %zu < %zu bytes", workspace_bytes, w.bytes);
  if (flags & B3D_FLAG_RUN_DEAD_KNN)
    B3D_REQUIRE(pw->knn_conv.lin && pw->knn_conv.att_src && pw->knn_conv.att_dst && pw->knn_conv.bias,
                "b3d_pose_forward: knn_conv pointers are required with B3D_FLAG_RUN_DEAD_KNN");
  B3D_TRY(pack_forward(pw, w, tr, (flags & B3D_FLAG_RUN_DEAD_KNN) != 0, stream));

  {  // edge encoder: edge_attr.float() -> 4-8-16-32                      pose_gnn.py:67
    ChainFwdArgs<LoadEdgeAttrF64, StoreAligned<2>> a;
    memset(&a, 0, sizeof(a));
    a.rows = E; a.in.ptr = edge_attr;
    a.out = StoreAligned<2>{w.e[0], nullptr, D::DE, 0};
    a.save_in = w.ea_pad; a.save[0] = w.ee_a1; a.save[1] = w.ee_a2;
    a.wpack = w.wp_ee;
    B3D_TRY(launch_rows<kNWEdge>(chain_fwd_kernel<SeqEdgeEnc, 0x3u, LoadEdgeAttrF64, StoreAligned<2>, kNWEdge>, "edge_encoder", a, E, stream, B3D_K_OTHER, chain_lds<SeqEdgeEnc>()));
  }
  {  // node encoder 19-24-36-48 (initial_x == x == x_enc)                          pose_gnn.py:68-71
    ChainFwdArgs<LoadUnaligned<19>, StoreTwo<3>> a;
    memset(&a, 0, sizeof(a));
    a.rows = N; a.in.ptr = pose_feats;
    a.out = StoreTwo<3>{w.x[0], out_x_enc, D::DX};       // layer-0 input and the returned x_enc (pose_gnn.py:86)
    a.save_in = w.pose_pad; a.save[0] = w.ne_a1; a.save[1] = w.ne_a2;
    a.wpack = w.wp_ne;
    B3D_TRY(launch_rows<kNWNode>(chain_fwd_kernel<SeqNodeEnc, 0x3u, LoadUnaligned<19>, StoreTwo<3>, kNWNode>, "node_encoder", a, N, stream, B3D_K_OTHER, chain_lds<SeqNodeEnc>()));
  }
  if (w.hoist) {  // x0 terms of the future / past columns (once per forward) + the per-node table of layer 0
    NodeProj0Args a;
    a.N = N; a.x0 = w.x[0]; a.T0 = w.T0; a.T = w.T;
    a.wpack = w.wp_ne_h + SeqNodeEncH::layer_off(3);
    B3D_TRY(launch_node_split<D>(node_proj0_split_kernel<D>, "node_proj0", a, N, stream, B3D_K_OTHER));
  }
  Side* knn_side = nullptr;
  for (int l = 0; l < depth; ++l) {
    if ((flags & B3D_FLAG_RUN_DEAD_KNN) && (l % 2 == 0)) {
      // frame-wise k-NN + GAT whose result the reference discards          pose_gnn.py:74-80
      B3D_REQUIRE(node_timestamps != nullptr, "node_timestamps required with B3D_FLAG_RUN_DEAD_KNN");
      hipStream_t ks = stream;
      if (!(flags & B3D_FLAG_SINGLE_STREAM)) {
        if (!knn_side) B3D_TRY(side_get(0, &knn_side));
        B3D_TRY(side_fork(stream, knn_side));              // x[l] is complete on `stream` here
        ks = knn_side->s;
      }
      // on the launch stream the block reads GATConv.lin(x[l]) from the per-node table (T is rewritten by the
      // next node update, so a side stream computes its own copy)
      const bool pre = w.hoist && ks == stream;
      B3D_TRY(knn_gat_block<D::DX>(w.knn, w.x[l], node_timestamps, N, pw->knn_conv, 20, ks, pre ? w.T + HP::OG : nullptr, HP::TW));
    }
    if (w.hoist) {
      EdgeFwdHArgs ea;
      memset(&ea, 0, sizeof(ea));
      ea.E = E; ea.src = g->src; ea.dst = g->dst;
      ea.T = w.T; ea.e_in = w.e[l]; ea.a_in = nullptr;
      ea.e_out = w.e[l + 1]; ea.fut = w.fut; ea.past = w.past;
      ea.sH1 = w.sH1[l]; ea.sH2 = w.sH2[l]; ea.sF1 = w.sF1[l]; ea.sP1 = w.sP1[l];
      ea.rmask = reinterpret_cast<unsigned*>(w.rmask[l]);
      ea.wpack = w.wp_efwd_h;
            B3D_TRY(launch_rows<kNWEdgeH>(mp_edge_fwd_h_kernel<D, kNWEdgeH>, "mp_edge_fwd", ea, E, stream, B3D_K_EDGE_FWD, stream_lds_bytes<HP::EdgeFwdSeq>()));
    } else {
    EdgeFwdArgs ea;
    memset(&ea, 0, sizeof(ea));
    ea.E = E; ea.src = g->src; ea.dst = g->dst;
    ea.x = w.x[l]; ea.x0 = w.x[0]; ea.e_in = w.e[l]; ea.a_in = nullptr;
    ea.e_out = w.e[l + 1]; ea.fut = w.fut; ea.past = w.past;
    ea.sH1 = w.sH1[l]; ea.sH2 = w.sH2[l]; ea.sF1 = w.sF1[l]; ea.sP1 = w.sP1[l];
    ea.wpack = w.wp_efwd;
    B3D_TRY(launch_rows<kNWEdge>(mp_edge_fwd_kernel<D, kNWEdge>, "mp_edge_fwd", ea, E, stream, B3D_K_EDGE_FWD));
    }
    NodeFwdArgs na;
    memset(&na, 0, sizeof(na));
    na.N = N; na.dst_ptr = g->dst_ptr; na.dst_perm = g->dst_perm; na.src_ptr = g->src_ptr; na.src_perm = g->src_perm;
    na.past = w.past; na.fut = w.fut; na.M = w.M[l]; na.x_out = w.x[l + 1]; na.sH1 = w.nH1[l]; na.sH2 = w.nH2[l];
    if (w.hoist && l + 1 < depth) {            // + the per-node table the next layer's edge phase gathers
      na.wpack = w.wp_nfwd_h; na.T = w.T; na.T0 = w.T0;
      B3D_TRY((launch_node_split<D, kNodeWavesWide>(mp_node_fwd_split_h_kernel<D>, "mp_node_fwd", na, N, stream, B3D_K_NODE_FWD)));
    } else {
      na.wpack = w.wp_nfwd;
      B3D_TRY((launch_node_split<D, kNodeWavesWide>(mp_node_fwd_split_kernel<D, kNodeWavesWide>, "mp_node_fwd", na, N, stream, B3D_K_NODE_FWD)));
    }
  }
  {  // edge classifier 32-16-8-4-1 -> logits                                pose_gnn.py:86
    ChainFwdArgs<LoadAligned<2>, StoreScalar> a;
    memset(&a, 0, sizeof(a));
    a.rows = E; a.in = LoadAligned<2>{w.e[depth], nullptr, D::DE, 0};
    a.out = StoreScalar{out_logits, 0};
    a.save[0] = w.c_a1; a.save[1] = w.c_a2; a.save[2] = w.c_a3;
    a.wpack = w.wp_cls;
    B3D_TRY(launch_rows<kNWEdge>(chain_fwd_kernel<SeqCls, 0x7u, LoadAligned<2>, StoreScalar, kNWEdge>, "edge_classifier", a, E, stream, B3D_K_OTHER, chain_lds<SeqCls>()));
  }
  if (knn_side && !((flags & B3D_FLAG_DEFER_SIDE_JOIN) && (flags & B3D_FLAG_TRAINING))) B3D_TRY(side_join(knn_side, stream));
  return B3D_OK;
}

extern "C" int b3d_pose_backward(const b3d_pose_weights* pw, const b3d_graph* g, const float* pose_feats,
                                 const double* edge_attr, int32_t depth, void* workspace, size_t workspace_bytes,
                                 const float* d_logits, const float* d_x_enc, const b3d_pose_grads* gr,
                                 b3d_stream stream_) {
  (void)pose_feats; (void)edge_attr;
  hipStream_t stream = (hipStream_t)stream_;
  B3D_TRY(check_weights(pw));
  B3D_REQUIRE(g && workspace && gr, "b3d_pose_backward: null argument");
  B3D_REQUIRE(depth >= 1 && depth <= 15, "b3d_pose_backward: depth %d outside [1,15]", depth);
  const int N = g->N, E = g->E;
  B3D_REQUIRE(N > 0 && E > 0, "b3d_pose_backward: empty graph");
  PoseWs w;
  carve(w, workspace, workspace_bytes, N, E, depth, B3D_FLAG_TRAINING);
  if (!w.ok) return fail(B3D_ERR_WORKSPACE, "b3d_pose_backward: workspace %zu < %zu bytes", workspace_bytes, w.bytes);
  const int* src = g->src;
  const int* dst = g->dst;

  const size_t eL1 = (size_t)E * D::EH1, eL2 = (size_t)E * D::EH2, eLe = (size_t)E * D::DE, eLm = (size_t)E * D::MH;
  const size_t nLm = (size_t)N * D::NIN, nLx = (size_t)N * D::DX, nL1 = (size_t)N * D::NH1, nL2 = (size_t)N * D::NH2;

  // ---- classifier: d_logits -> d e[depth] -------------------------------------------------------
  int cur = 0;
  {
    ChainBwdArgs<LoadScalar, StoreAligned<2>> a;
    memset(&a, 0, sizeof(a));
    a.rows = E; a.in.ptr = d_logits;
    a.out = StoreAligned<2>{w.de[cur], nullptr, D::DE, 0};
    a.act[0] = w.c_a3; a.act[1] = w.c_a2; a.act[2] = w.c_a1; a.act[3] = nullptr;
    a.gsave[0] = w.gc3; a.gsave[1] = w.gc2; a.gsave[2] = w.gc1; a.gsave[3] = nullptr;
    a.gtop = w.gc_top;                       // d_logits padded to 16 columns: G of edge_classifier.6
    a.wpack = w.wp_clsT;
    B3D_TRY(launch_rows<kNWEdge>(chain_bwd_kernel<SeqClsT, LoadScalar, StoreAligned<2>, kNWEdge>, "edge_classifier_bwd", a, E, stream, B3D_K_OTHER, chain_lds<SeqClsT>()));
  }

  // ---- message-passing layers, last to first: data gradients only; the G tensors of every layer
  //      are kept for ONE streaming weight-gradient launch after the sweep -------------------------
  bool dx0_first = true;
  // hoisted first layers: per-node gradient of T (kept per layer) and of (x | x0) from layer `lay`'s G tensors
  auto gradproj = [&](int lay) -> int {
    NodeGradProjArgs ga;
    memset(&ga, 0, sizeof(ga));
    ga.N = N; ga.dst_ptr = g->dst_ptr; ga.dst_perm = g->dst_perm; ga.src_ptr = g->src_ptr; ga.src_perm = g->src_perm;
    ga.GdH1 = w.GdH1 + lay * eL1;
    const bool lay_msgs = lay < depth - 1;
    ga.GdF1 = lay_msgs ? w.GdF1 + lay * eLm : nullptr;
    ga.GdP1 = lay_msgs ? w.GdP1 + lay * eLm : nullptr;
    ga.dT = w.dT + (size_t)lay * N * HP::GW; ga.gx = w.gx;
    ga.wpack = w.wp_gproj;
    B3D_TRY(set_lds(node_gradproj_kernel<D>, GradProjLds<D>::BYTES));
    ProfScope ps(B3D_K_OTHER, stream);          // layer 0 only: a different kernel from the mp_node_bwd family
    hipLaunchKernelGGL(node_gradproj_kernel<D>, dim3((N + 15) / 16), dim3(kGradProjWaves * 64), GradProjLds<D>::BYTES, stream, ga);
    return launch_check("node_gradproj_kernel");
  };
  for (int l = depth - 1; l >= 0; --l) {
    const bool msgs = (l < depth - 1);   // the last layer's node update feeds nothing (pose_gnn.py:86)
    if (msgs) {
      // node backward of layer l consumes the node gradients of layer l+1's edge phase
      NodeBwdArgs nb;
      memset(&nb, 0, sizeof(nb));
      nb.N = N; nb.dst_ptr = g->dst_ptr; nb.dst_perm = g->dst_perm; nb.src_ptr = g->src_ptr; nb.src_perm = g->src_perm;
      if (w.hoist) {
        // per-node gradient of T from layer l+1's first-layer gradients, (dx | dx0), node MLP: one launch
        NodeBwdHArgs hb;
        memset(&hb, 0, sizeof(hb));
        const int lay = l + 1;
        hb.gp.N = N; hb.gp.dst_ptr = g->dst_ptr; hb.gp.dst_perm = g->dst_perm; hb.gp.src_ptr = g->src_ptr; hb.gp.src_perm = g->src_perm;
        hb.gp.GdH1 = w.GdH1 + lay * eL1;
        hb.gp.GdF1 = (lay < depth - 1) ? w.GdF1 + lay * eLm : nullptr;
        hb.gp.GdP1 = (lay < depth - 1) ? w.GdP1 + lay * eLm : nullptr;
        hb.gp.dT = w.dT + (size_t)lay * N * HP::GW;
        hb.dx0_acc = w.dx0_acc; hb.dx0_first = dx0_first ? 1 : 0;
        hb.sH1 = w.nH1[l]; hb.sH2 = w.nH2[l];
        hb.dM = w.dM + l * nLm; hb.Gdx = w.Gdx + l * nLx; hb.GdH2 = w.GnH2 + l * nL2; hb.GdH1 = w.GnH1 + l * nL1;
        hb.wpack = w.wp_nbwd_h;
        B3D_TRY(set_lds(node_bwd_h_kernel<D>, NodeBwdHLds<D>::BYTES));
        {
          ProfScope ps(B3D_K_NODE_BWD, stream);
          hipLaunchKernelGGL(node_bwd_h_kernel<D>, dim3((N + 15) / 16), dim3(kGradProjWaves * 64), NodeBwdHLds<D>::BYTES, stream, hb);
        }
        B3D_TRY(launch_check("node_bwd_h_kernel"));
      } else {
        nb.gdst = w.gdst; nb.gsrc = w.gsrc;
        nb.dx0_acc = w.dx0_acc; nb.dx0_first = dx0_first ? 1 : 0;
        nb.sH1 = w.nH1[l]; nb.sH2 = w.nH2[l];
        nb.dM = w.dM + l * nLm; nb.Gdx = w.Gdx + l * nLx; nb.GdH2 = w.GnH2 + l * nL2; nb.GdH1 = w.GnH1 + l * nL1;
        nb.wpack = w.wp_nbwd;
        B3D_TRY(launch_node_split<D>(mp_node_bwd_split_kernel<D>, "mp_node_bwd", nb, N, stream, B3D_K_NODE_BWD));
      }
      dx0_first = false;
    }
    if (w.hoist) {
      EdgeBwdHArgs eb;
      memset(&eb, 0, sizeof(eb));
      eb.E = E; eb.src = src; eb.dst = dst;
      eb.dM = msgs ? w.dM + l * nLm : nullptr;
      eb.de_out = w.de[cur]; eb.de_in = w.de[cur ^ 1];
      eb.sH1 = w.sH1[l]; eb.sH2 = w.sH2[l]; eb.sF1 = w.sF1[l]; eb.sP1 = w.sP1[l];
      eb.rmask = reinterpret_cast<const unsigned*>(w.rmask[l]);
      eb.GdH1 = w.GdH1 + l * eL1; eb.GdH2 = w.GdH2 + l * eL2; eb.Gde = w.Gde + l * eLe;
      eb.GdF1 = w.GdF1 + l * eLm; eb.GdP1 = w.GdP1 + l * eLm;
      if (msgs) {
        eb.wpack = w.wp_ebwd_h;
        B3D_TRY(launch_rows<kNWEdgeH>(mp_edge_bwd_h_kernel<D, true, kNWEdgeH>, "mp_edge_bwd", eb, E, stream, B3D_K_EDGE_BWD, stream_lds_bytes<HP::EdgeBwdSeq>()));
      } else {
        eb.wpack = w.wp_ebwd_nm_h;
        B3D_TRY(launch_rows<kNWEdgeH>(mp_edge_bwd_h_kernel<D, false, kNWEdgeH>, "mp_edge_bwd_last", eb, E, stream, B3D_K_EDGE_BWD, stream_lds_bytes<HP::EdgeBwdSeqNoMsg>()));
      }
    } else {
    EdgeBwdArgs eb;
    memset(&eb, 0, sizeof(eb));
    eb.E = E; eb.src = src; eb.dst = dst;
    eb.dM = msgs ? w.dM + l * nLm : nullptr;
    eb.de_out = w.de[cur]; eb.de_in = w.de[cur ^ 1];
    eb.sH1 = w.sH1[l]; eb.sH2 = w.sH2[l]; eb.sF1 = w.sF1[l]; eb.sP1 = w.sP1[l];
    eb.da_acc = nullptr; eb.da_first = 0;
    eb.gdst = w.gdst; eb.gsrc = w.gsrc;
    eb.GdH1 = w.GdH1 + l * eL1; eb.GdH2 = w.GdH2 + l * eL2; eb.Gde = w.Gde + l * eLe;
    eb.GdF1 = w.GdF1 + l * eLm; eb.GdP1 = w.GdP1 + l * eLm;
    if (msgs) {
      eb.wpack = w.wp_ebwd;
      B3D_TRY(launch_rows<kNWEdge>(mp_edge_bwd_kernel<D, true, kNWEdge>, "mp_edge_bwd", eb, E, stream, B3D_K_EDGE_BWD));
    } else {
      eb.wpack = w.wp_ebwd_nm;
      B3D_TRY(launch_rows<kNWEdge>(mp_edge_bwd_kernel<D, false, kNWEdge>, "mp_edge_bwd_last", eb, E, stream, B3D_K_EDGE_BWD));
    }
    }
    cur ^= 1;
  }

  // ---- encoders ---------------------------------------------------------------------------------
  if (w.hoist) {  // node encoder: gradient at x_enc = upstream + running d initial_x + layer 0's (dx | dx0)
    B3D_TRY(gradproj(0));
    using In = LoadNodeEncGradH<3>;
    ChainBwdArgs<In, StoreNone> a;
    memset(&a, 0, sizeof(a));
    a.rows = N;
    a.in = In{d_x_enc, dx0_first ? nullptr : w.dx0_acc, w.gx};
    a.gtop = w.gn_top;
    a.act[0] = w.ne_a2; a.act[1] = w.ne_a1;
    a.gsave[0] = w.gn2; a.gsave[1] = w.gn1;
    a.wpack = w.wp_neT;
    B3D_TRY(launch_rows<kNWNode>(chain_bwd_kernel<SeqNodeEncT, In, StoreNone, kNWNode>, "node_encoder_bwd", a, N, stream, B3D_K_OTHER, chain_lds<SeqNodeEncT>()));
  } else {  // node encoder: gradient at x_enc = upstream + running d initial_x + layer-0 scatter transposes
    using In = LoadNodeEncGrad<3>;
    ChainBwdArgs<In, StoreNone> a;
    memset(&a, 0, sizeof(a));
    a.rows = N;
    a.in = In{d_x_enc, dx0_first ? nullptr : w.dx0_acc, w.gdst, w.gsrc, g->dst_ptr, g->dst_perm, g->src_ptr, g->src_perm};
    a.gtop = w.gn_top;
    a.act[0] = w.ne_a2; a.act[1] = w.ne_a1;
    a.gsave[0] = w.gn2; a.gsave[1] = w.gn1;
    a.wpack = w.wp_neT;
    B3D_TRY(launch_rows<kNWNode>(chain_bwd_kernel<SeqNodeEncT, In, StoreNone, kNWNode>, "node_encoder_bwd", a, N, stream, B3D_K_OTHER, chain_lds<SeqNodeEncT>()));
  }
  {  // edge encoder: G_3 = d e[0]
    ChainBwdArgs<LoadAligned<2>, StoreNone> a;
    memset(&a, 0, sizeof(a));
    a.rows = E;
    a.in = LoadAligned<2>{w.de[cur], nullptr, D::DE, 0};
    a.act[0] = w.ee_a2; a.act[1] = w.ee_a1;
    a.gsave[0] = w.ge2; a.gsave[1] = w.ge1;
    a.wpack = w.wp_eeT;
    B3D_TRY(launch_rows<kNWEdge>(chain_bwd_kernel<SeqEdgeEncT, LoadAligned<2>, StoreNone, kNWEdge>, "edge_encoder_bwd", a, E, stream, B3D_K_OTHER, chain_lds<SeqEdgeEncT>()));
  }

  // ---- message-passing weight gradients: all layers, one streaming launch ---------------------
  {
    MpGradSrc ms;
    ms.x = w.x[0]; ms.xs = (long)nLx; ms.x0 = w.x[0]; ms.e_in = w.e[0]; ms.e_out = w.e[1]; ms.es = (long)eLe;
    ms.GdH1 = w.GdH1; ms.GdH2 = w.GdH2; ms.Gde = w.Gde; ms.GdP1 = w.GdP1; ms.GdF1 = w.GdF1; ms.dM = w.dM;
    ms.GnH1 = w.GnH1; ms.GnH2 = w.GnH2; ms.Gdx = w.Gdx;
    ms.sH1 = w.sH1[0]; ms.sH2 = w.sH2[0]; ms.sP1 = w.sP1[0]; ms.sF1 = w.sF1[0]; ms.M = w.M[0]; ms.nH1 = w.nH1[0]; ms.nH2 = w.nH2[0];
    ms.e_last = w.e[depth]; ms.de0 = w.de[cur]; ms.have_logit_grad = d_logits != nullptr; ms.dT = w.dT;
    B3D_TRY(mp_weight_grads(w, ms, N, E, src, dst, stream));
  }

  // ---- slabs -> parameter gradients -------------------------------------------------------------
  {
    RedArgs ra;
    ra.nentries = 0;
    const b3d_linear_grad* groups[] = {gr->edge_encoder, gr->node_encoder, gr->edge_classifier, gr->mp.edge_update,
                                       gr->mp.create_past_msgs, gr->mp.create_future_msgs, gr->mp.combine_future_past};
    const int first[] = {LIN_EE0, LIN_NE0, LIN_C0, LIN_EU0, LIN_PA0, LIN_FU0, LIN_CF0};
    const int cnt[] = {3, 3, 4, 3, 2, 2, 3};
    for (int gi = 0; gi < 7; ++gi)
      for (int i = 0; i < cnt[gi]; ++i) {
        LinSlab& ls = w.lin[first[gi] + i];
        float* dw = groups[gi][i].w;
        float* db = groups[gi][i].b;
        const int li = first[gi] + i;
        if (w.hoist && (li == LIN_EU0 || li == LIN_PA0 || li == LIN_FU0)) {
          // the gradient of a hoisted first layer arrives as three column blocks with slabs of their own
          struct Part { int vl, col; bool bias; };
          const Part eu[3] = {{VL_EU0XI, 0, false}, {VL_EU0XJ, D::DX, false}, {VL_EU0E, 2 * D::DX, true}};
          const Part fu[3] = {{VL_FU0X, 0, false}, {VL_FU0E, D::DX, true}, {VL_FU0X0, D::DX + D::DE, false}};
          const Part pa[3] = {{VL_PA0X, 0, false}, {VL_PA0E, D::DX, true}, {VL_PA0X0, D::DX + D::DE, false}};
          const Part* parts = (li == LIN_EU0) ? eu : (li == LIN_FU0) ? fu : pa;
          if (!w.vlin[parts[0].vl].used) {       // depth == 1: the message stacks receive no gradient
            if (dw) B3D_HIP_CHECK(hipMemsetAsync(dw, 0, (size_t)ls.N * ls.K * sizeof(float), stream));
            if (db) B3D_HIP_CHECK(hipMemsetAsync(db, 0, (size_t)ls.N * sizeof(float), stream));
            continue;
          }
          for (int k = 0; k < 3; ++k) {
            RedEntry e = red_entry(w.vlin[parts[k].vl], dw ? dw + parts[k].col : nullptr, parts[k].bias ? db : nullptr);
            e.ld = ls.K;
            ra.e[ra.nentries++] = e;
          }
          continue;
        }
        if (!ls.used) {
          // no gradient reached this layer (e.g. d_logits == NULL, or depth == 1 for the message stacks)
          if (dw) B3D_HIP_CHECK(hipMemsetAsync(dw, 0, (size_t)ls.N * ls.K * sizeof(float), stream));
          if (db) B3D_HIP_CHECK(hipMemsetAsync(db, 0, (size_t)ls.N * sizeof(float), stream));
          continue;
        }
        ra.e[ra.nentries++] = red_entry(ls, dw, db);
      }
    B3D_TRY(launch_reduce(ra, stream));
  }
  B3D_TRY(b3d_side_join(stream_));      // a B3D_FLAG_DEFER_SIDE_JOIN forward left the k-NN block running under this sweep
  return B3D_OK;
}

// ---- standalone k-NN + GAT operator ---------------------------------------------------------------
extern "C" size_t b3d_knn_gat_workspace_bytes(int32_t N, int32_t Dm) {
  Carver c(nullptr, 0);
  KnnWs k;
  knn_carve(k, c, N, Dm);
  return c.off + 256;
}

extern "C" int b3d_knn_gat_forward(const float* x, const int64_t* ts, int32_t N, int32_t Dm, int32_t k, const b3d_gat* gat,
                                   void* workspace, size_t workspace_bytes, int32_t* out_nbr, int32_t* out_cnt,
                                   float* out_y, b3d_stream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  B3D_REQUIRE(x && ts && gat && workspace && out_nbr && out_cnt && out_y, "b3d_knn_gat_forward: null argument");
  B3D_REQUIRE(Dm == 48 || Dm == 96, "b3d_knn_gat_forward: D must be 48 or 96, got %d", Dm);
  B3D_REQUIRE(N > 0, "b3d_knn_gat_forward: empty input");
  Carver c(workspace, workspace_bytes);
  KnnWs w;
  knn_carve(w, c, N, Dm);
  if (!c.ok()) return fail(B3D_ERR_WORKSPACE, "b3d_knn_gat_forward: workspace too small");
  if (Dm == 48) B3D_TRY(knn_gat_block<48>(w, x, ts, N, *gat, k, stream));
  else B3D_TRY(knn_gat_block<96>(w, x, ts, N, *gat, k, stream));
  B3D_HIP_CHECK(hipMemcpyAsync(out_nbr, w.nbr, (size_t)N * kKnnMaxK * sizeof(int), hipMemcpyDeviceToDevice, stream));
  B3D_HIP_CHECK(hipMemcpyAsync(out_cnt, w.cnt, (size_t)N * sizeof(int), hipMemcpyDeviceToDevice, stream));
  B3D_HIP_CHECK(hipMemcpyAsync(out_y, w.y, (size_t)N * Dm * sizeof(float), hipMemcpyDeviceToDevice, stream));
  return B3D_OK;
}

// ---- standalone CausalMessagePassing layer (pose_gnn.py:125-252) -----------------------------------------
namespace b3d {

static int check_mp_weights(const b3d_mp_weights* mw) {
  B3D_REQUIRE(mw != nullptr, "message-passing weights struct is null");
  const b3d_linear* all[] = {mw->edge_update, mw->create_past_msgs, mw->create_future_msgs, mw->combine_future_past};
  const int cnt[] = {3, 2, 2, 3};
  for (int g = 0; g < 4; ++g)
    for (int i = 0; i < cnt[g]; ++i)
      B3D_REQUIRE(all[g][i].w != nullptr && all[g][i].b != nullptr, "null weight/bias pointer (stack %d layer %d)", g, i);
  return B3D_OK;
}

static int pack_layer(const b3d_mp_weights& mp, PoseWs& w, bool training, hipStream_t stream) {
  PackDesc d[32];
  int n = 0;
  using EF = D::EdgeFwdSeq;
  for (int i = 0; i < 3; ++i) d[n++] = pack_desc<EF>(i, w.wp_efwd, mp.edge_update[i].w, mp.edge_update[i].b, kLinDims[LIN_EU0 + i].N, kLinDims[LIN_EU0 + i].K, false);
  for (int i = 0; i < 2; ++i) d[n++] = pack_desc<EF>(3 + i, w.wp_efwd, mp.create_future_msgs[i].w, mp.create_future_msgs[i].b, kLinDims[LIN_FU0 + i].N, kLinDims[LIN_FU0 + i].K, false);
  for (int i = 0; i < 2; ++i) d[n++] = pack_desc<EF>(5 + i, w.wp_efwd, mp.create_past_msgs[i].w, mp.create_past_msgs[i].b, kLinDims[LIN_PA0 + i].N, kLinDims[LIN_PA0 + i].K, false);
  for (int i = 0; i < 3; ++i) d[n++] = pack_desc<D::NodeFwdSeq>(i, w.wp_nfwd, mp.combine_future_past[i].w, mp.combine_future_past[i].b, kLinDims[LIN_CF0 + i].N, kLinDims[LIN_CF0 + i].K, false);
  if (training) {
    d[n++] = fill_desc(w.iota, w.iota_n, true);
    d[n++] = fill_desc(w.zrow, 256, false);
    auto T = [&](auto seq_tag, int li, float* base, const b3d_linear& l, int lin) {
      using S = decltype(seq_tag);
      d[n++] = pack_desc<S>(li, base, l.w, nullptr, kLinDims[lin].K, kLinDims[lin].N, true);
    };
    using EB = D::EdgeBwdSeq;
    T(EB{}, 0, w.wp_ebwd, mp.create_past_msgs[1], LIN_PA1);
    T(EB{}, 1, w.wp_ebwd, mp.create_past_msgs[0], LIN_PA0);
    T(EB{}, 2, w.wp_ebwd, mp.create_future_msgs[1], LIN_FU1);
    T(EB{}, 3, w.wp_ebwd, mp.create_future_msgs[0], LIN_FU0);
    T(EB{}, 4, w.wp_ebwd, mp.edge_update[2], LIN_EU2);
    T(EB{}, 5, w.wp_ebwd, mp.edge_update[1], LIN_EU1);
    T(EB{}, 6, w.wp_ebwd, mp.edge_update[0], LIN_EU0);
    using NB = D::NodeBwdSeq;
    T(NB{}, 0, w.wp_nbwd, mp.combine_future_past[2], LIN_CF2);
    T(NB{}, 1, w.wp_nbwd, mp.combine_future_past[1], LIN_CF1);
    T(NB{}, 2, w.wp_nbwd, mp.combine_future_past[0], LIN_CF0);
  }
  return pack_images(d, n, stream);
}

// d x[n] = sum over edges with dst == n of gdst[.,0:DX] + sum over edges with src == n of gsrc[.,0:DX];
// d x0[n] likewise from columns DX:2DX -- the transpose of the four node-row gathers of the edge phase.
struct NodeGradArgs {
  int N;
  const int *dst_ptr, *dst_perm, *src_ptr, *src_perm;
  const float *gdst, *gsrc;
  float *d_x, *d_x0;
};
__global__ __launch_bounds__(256) void node_grad_gather_kernel(const NodeGradArgs a) {
  constexpr int XB = D::DX / 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long row = ((long)blockIdx.x * 4 + wave) * 16 + (lane & 15);
  const bool valid = row < a.N;
  v4f g[2 * XB];
#pragma unroll
  for (int b = 0; b < 2 * XB; ++b) g[b] = v4f{0.f, 0.f, 0.f, 0.f};
  if (valid) {
    segment_sum<2 * XB>(a.gdst, 2 * D::DX, 0, a.dst_perm, a.dst_ptr[row], a.dst_ptr[row + 1], g);
    segment_sum<2 * XB>(a.gsrc, 2 * D::DX, 0, a.src_perm, a.src_ptr[row], a.src_ptr[row + 1], g);
  }
  if (a.d_x) store_row<XB>(a.d_x, row, D::DX, 0, valid, g);
  if (a.d_x0) store_row<XB>(a.d_x0, row, D::DX, 0, valid, g + XB);
}

}  // namespace b3d

extern "C" size_t b3d_pose_layer_workspace_bytes(int32_t N, int32_t E, uint32_t flags) {
  PoseWs w;
  carve(w, nullptr, 0, N, E, 1, (flags & B3D_FLAG_TRAINING) | kFlagLayerMode);
  return w.bytes;
}

extern "C" int b3d_pose_layer_forward(const b3d_mp_weights* mw, const b3d_graph* g, const float* x, const float* x0,
                                      const float* e, uint32_t flags, void* workspace, size_t workspace_bytes,
                                      float* x_new, float* e_new, b3d_stream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  B3D_TRY(check_mp_weights(mw));
  B3D_REQUIRE(g && x && x0 && e && workspace && x_new && e_new, "b3d_pose_layer_forward: null argument");
  const int N = g->N, E = g->E;
  B3D_REQUIRE(N > 0 && E > 0, "b3d_pose_layer_forward: empty graph (N=%d, E=%d)", N, E);
  const bool tr = flags & B3D_FLAG_TRAINING;
  PoseWs w;
  carve(w, workspace, workspace_bytes, N, E, 1, (flags & B3D_FLAG_TRAINING) | kFlagLayerMode);
  if (!w.ok) return fail(B3D_ERR_WORKSPACE, "b3d_pose_layer_forward: workspace %zu < %zu bytes", workspace_bytes, w.bytes);
  B3D_TRY(pack_layer(*mw, w, tr, stream));
  EdgeFwdArgs ea;
  memset(&ea, 0, sizeof(ea));
  ea.E = E; ea.src = g->src; ea.dst = g->dst;
  ea.x = x; ea.x0 = x0; ea.e_in = e; ea.a_in = nullptr;
  ea.e_out = e_new; ea.fut = w.fut; ea.past = w.past;
  if (tr) { ea.sH1 = w.sH1[0]; ea.sH2 = w.sH2[0]; ea.sF1 = w.sF1[0]; ea.sP1 = w.sP1[0]; }
  ea.wpack = w.wp_efwd;
  B3D_TRY(launch_rows<kNWEdge>(mp_edge_fwd_kernel<D, kNWEdge>, "mp_edge_fwd", ea, E, stream, B3D_K_EDGE_FWD));
  NodeFwdArgs na;
  memset(&na, 0, sizeof(na));
  na.N = N; na.dst_ptr = g->dst_ptr; na.dst_perm = g->dst_perm; na.src_ptr = g->src_ptr; na.src_perm = g->src_perm;
  na.past = w.past; na.fut = w.fut; na.x_out = x_new;
  if (tr) { na.M = w.M[0]; na.sH1 = w.nH1[0]; na.sH2 = w.nH2[0]; }
  na.wpack = w.wp_nfwd;
  B3D_TRY((launch_node_split<D, kNodeWavesWide>(mp_node_fwd_split_kernel<D, kNodeWavesWide>, "mp_node_fwd", na, N, stream, B3D_K_NODE_FWD)));
  return B3D_OK;
}

extern "C" int b3d_pose_layer_backward(const b3d_mp_weights* mw, const b3d_graph* g, const float* x, const float* x0,
                                       const float* e, const float* e_new, void* workspace, size_t workspace_bytes,
                                       const float* d_x_new, const float* d_e_new, float* d_x, float* d_x0, float* d_e,
                                       const b3d_mp_grads* gr, b3d_stream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  B3D_TRY(check_mp_weights(mw));
  B3D_REQUIRE(g && x && x0 && e && e_new && workspace && gr, "b3d_pose_layer_backward: null argument");
  const int N = g->N, E = g->E;
  B3D_REQUIRE(N > 0 && E > 0, "b3d_pose_layer_backward: empty graph");
  PoseWs w;
  carve(w, workspace, workspace_bytes, N, E, 1, B3D_FLAG_TRAINING | kFlagLayerMode);
  if (!w.ok) return fail(B3D_ERR_WORKSPACE, "b3d_pose_layer_backward: workspace %zu < %zu bytes", workspace_bytes, w.bytes);
  // missing upstream gradients are zeros
  if (!d_x_new) { B3D_HIP_CHECK(hipMemsetAsync(w.dx0_acc, 0, (size_t)N * D::DX * sizeof(float), stream)); d_x_new = w.dx0_acc; }
  if (!d_e_new) { B3D_HIP_CHECK(hipMemsetAsync(w.de[0], 0, (size_t)E * D::DE * sizeof(float), stream)); d_e_new = w.de[0]; }
  NodeBwdArgs nb;
  memset(&nb, 0, sizeof(nb));
  nb.N = N; nb.dst_ptr = g->dst_ptr; nb.dst_perm = g->dst_perm; nb.src_ptr = g->src_ptr; nb.src_perm = g->src_perm;
  nb.g_direct = d_x_new;
  nb.sH1 = w.nH1[0]; nb.sH2 = w.nH2[0];
  nb.dM = w.dM; nb.GdH2 = w.GnH2; nb.GdH1 = w.GnH1;
  nb.wpack = w.wp_nbwd;
  B3D_TRY(launch_node_split<D>(mp_node_bwd_split_kernel<D>, "mp_node_bwd", nb, N, stream, B3D_K_NODE_BWD));
  EdgeBwdArgs eb;
  memset(&eb, 0, sizeof(eb));
  eb.E = E; eb.src = g->src; eb.dst = g->dst;
  eb.dM = w.dM;
  eb.de_out = d_e_new; eb.de_in = d_e ? d_e : w.de[1];
  eb.sH1 = w.sH1[0]; eb.sH2 = w.sH2[0]; eb.sF1 = w.sF1[0]; eb.sP1 = w.sP1[0];
  eb.gdst = w.gdst; eb.gsrc = w.gsrc;
  eb.GdH1 = w.GdH1; eb.GdH2 = w.GdH2; eb.Gde = w.Gde; eb.GdF1 = w.GdF1; eb.GdP1 = w.GdP1;
  eb.wpack = w.wp_ebwd;
  B3D_TRY(launch_rows<kNWEdge>(mp_edge_bwd_kernel<D, true, kNWEdge>, "mp_edge_bwd", eb, E, stream, B3D_K_EDGE_BWD));
  if (d_x || d_x0) {
    NodeGradArgs na{N, g->dst_ptr, g->dst_perm, g->src_ptr, g->src_perm, w.gdst, w.gsrc, d_x, d_x0};
    hipLaunchKernelGGL(node_grad_gather_kernel, dim3((N + 63) / 64), dim3(256), 0, stream, na);
    B3D_TRY(launch_check("node_grad_gather_kernel"));
  }
  MpGradSrc ms;
  ms.x = x; ms.xs = 0; ms.x0 = x0; ms.e_in = e; ms.e_out = e_new; ms.es = 0;
  ms.GdH1 = w.GdH1; ms.GdH2 = w.GdH2; ms.Gde = w.Gde; ms.GdP1 = w.GdP1; ms.GdF1 = w.GdF1; ms.dM = w.dM;
  ms.GnH1 = w.GnH1; ms.GnH2 = w.GnH2; ms.Gdx = d_x_new;
  ms.sH1 = w.sH1[0]; ms.sH2 = w.sH2[0]; ms.sP1 = w.sP1[0]; ms.sF1 = w.sF1[0]; ms.M = w.M[0]; ms.nH1 = w.nH1[0]; ms.nH2 = w.nH2[0];
  ms.e_last = nullptr; ms.de0 = nullptr; ms.have_logit_grad = false; ms.dT = nullptr;
  B3D_TRY(mp_weight_grads(w, ms, N, E, g->src, g->dst, stream));
  RedArgs ra;
  ra.nentries = 0;
  const b3d_linear_grad* groups[] = {gr->edge_update, gr->create_past_msgs, gr->create_future_msgs, gr->combine_future_past};
  const int first[] = {LIN_EU0, LIN_PA0, LIN_FU0, LIN_CF0};
  const int cnt[] = {3, 2, 2, 3};
  for (int gi = 0; gi < 4; ++gi)
    for (int i = 0; i < cnt[gi]; ++i) ra.e[ra.nentries++] = red_entry(w.lin[first[gi] + i], groups[gi][i].w, groups[gi][i].b);
  B3D_TRY(launch_reduce(ra, stream));
  return B3D_OK;
}
