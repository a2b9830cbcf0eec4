// CausalMessagePassing on gfx950: fused edge phase and node phase, forward and backward.
//
// Reference semantics (pose_gnn.py:125-252, clr_att_gnn.py:227-356):
//   e'_k   = MLP_edge  ( x[dst_k] | x[src_k] | e_k (| a_k) )
//   fut_k  = MLP_future( x[dst_k] | e'_k | x0[dst_k] )      summed at the SOURCE  node
//   past_k = MLP_past  ( x[src_k] | e'_k | x0[src_k] )      summed at the DESTINATION node
//   x'_n   = MLP_node  ( sum past | sum fut )
//
// Kernel split per layer:
//   mp_edge_fwd : gather 4 node rows + edge rows, the three MLP stacks in registers, writes e',
//                 per-edge fut/past rows (+ hidden activations when training).
//   mp_node_fwd : deterministic CSR (by dst) / CSC (by src) segment sums of the per-edge message
//                 rows (no float atomics), node MLP (b3d_node.hpp).
//   mp_node_bwd : segment sums of the per-edge node gradients of the NEXT layer's edge backward
//                 (the transpose of its gathers), node MLP data-gradient -> dM.
//   mp_edge_bwd : gathers dM, data-gradient through the three stacks, writes per-edge gradient
//                 rows and the G tensors the weight-gradient kernel contracts over edges.
#pragma once
#include "b3d_dev.hpp"

namespace b3d {

// BF_: weight image format of this model's message-passing layers (b3d_dev.hpp: -1 by width, 0 fp32)
template <int DX_, int DE_, int DA_, int EH1_, int EH2_, int MH_, int DM_, int NH1_, int NH2_, int BF_ = -1>
struct MPDims {
  template <int K, int N> using LL = L<K, N, BF_>;
  // node-sized kernels deal a layer's output blocks to 4-16 wavefronts chunk by chunk: they keep the fp32 format,
  // whose chunks hold more blocks (and their few thousand rows are latency bound, not MFMA bound)
  template <int K, int N> using NL = LF<K, N>;
  static constexpr int DX = DX_, DE = DE_, DA = DA_, EH1 = EH1_, EH2 = EH2_, MH = MH_, DM = DM_,
                       NH1 = NH1_, NH2 = NH2_;
  static constexpr int EIN = 2 * DX + DE + DA;   // edge_update input
  static constexpr int MIN = 2 * DX + DE;        // create_*_msgs input
  static constexpr int NIN = 2 * DM;             // combine_future_past input
  // weight consumption order of each kernel (one packed image per entry)
  using EdgeFwdSeq = LayerSeq<LL<EIN, EH1>, LL<EH1, EH2>, LL<EH2, DE>,      // edge_update.0/.2/.4
                              LL<MIN, MH>, LL<MH, DM>,                     // create_future_msgs.0/.2
                              LL<MIN, MH>, LL<MH, DM>>;                    // create_past_msgs.0/.2
  using NodeFwdSeq = LayerSeq<NL<NIN, NH1>, NL<NH1, NH2>, NL<NH2, DX>>;     // combine_future_past
  // transposed images, data-gradient order
  using EdgeBwdSeq = LayerSeq<LL<DM, MH>, LL<MH, MIN>,                     // past.2^T, past.0^T
                              LL<DM, MH>, LL<MH, MIN>,                     // future.2^T, future.0^T
                              LL<DE, EH2>, LL<EH2, EH1>, LL<EH1, EIN>>;     // edge_update.4^T/.2^T/.0^T
  using EdgeBwdSeqNoMsg = LayerSeq<LL<DE, EH2>, LL<EH2, EH1>, LL<EH1, EIN>>;
  using NodeBwdSeq = LayerSeq<NL<DX, NH2>, NL<NH2, NH1>, NL<NH1, NIN>>;
};

using DimsP = MPDims<48, 32, 0, 96, 64, 96, 64, 96, 64, 0>;        // fp32 images (exact fmaf chain): its hoisted stacks stay resident in LDS           // pose_gnn.py:94-120
using DimsC = MPDims<96, 64, 64, 256, 128, 192, 128, 192, 128, 0>;    // clr_att_gnn.py:196-222, unsplit kernels (fp32 images)
using DimsCB = MPDims<96, 64, 64, 256, 128, 192, 128, 192, 128, -1>;  // the same widths, bf16x3 images: hoisted kernels

struct EdgeFwdArgs {
  int E;
  const int* src;
  const int* dst;
  const float* x;      // [N, DX]
  const float* x0;     // [N, DX]
  const float* e_in;   // [E, DE]
  const float* a_in;   // [E, DA] (CLR) or nullptr
  float* e_out;        // [E, DE]
  float* fut;          // [E, DM]
  float* past;         // [E, DM]
  float* sH1;          // saved hidden activations (training) or nullptr
  float* sH2;
  float* sF1;
  float* sP1;
  const float* wpack;  // EdgeFwdSeq images
};

struct NodeFwdArgs {
  int N;
  const int* dst_ptr;
  const int* dst_perm;  // nullptr = identity (destination-sorted edges)
  const int* src_ptr;
  const int* src_perm;
  const float* past;    // [E, DM]
  const float* fut;     // [E, DM]
  float* M;             // [N, 2 DM] aggregated messages (saved for the weight gradient) or nullptr
  float* x_out;         // [N, DX]
  float* sH1;           // [N, NH1] or nullptr
  float* sH2;           // [N, NH2] or nullptr
  const float* wpack;   // NodeFwdSeq images (+ the projection image for the hoisted variant)
  float* T;             // hoisted variant: [N, TW] per-node table of the next layer
  const float* T0;      //                  [N, 2 MH] x0 terms
};

struct NodeBwdArgs {
  int N;
  const int* dst_ptr;
  const int* dst_perm;
  const int* src_ptr;
  const int* src_perm;
  const float* gdst;    // [E, 2 DX]  (d x[dst] | d x0[dst]) per edge, from the next layer
  const float* gsrc;    // [E, 2 DX]  (d x[src] | d x0[src]) per edge
  const float* g_direct; // gradient given per NODE instead of the two per-edge lists: [N, DX] = d x' (standalone
                         // layer operator; dx0_acc / Gdx are not touched) or, with g_direct_wide, [N, 2 DX] =
                         // (d x' | d x0 contribution) from node_gradproj_kernel (hoisted first layers)
  int g_direct_wide;
  float* dx0_acc;       // [N, DX] running gradient of initial_x
  int dx0_first;        // 1: overwrite dx0_acc, 0: accumulate
  const float* sH1;     // saved activations of THIS layer's node MLP
  const float* sH2;
  float* dM;            // [N, 2 DM]
  float* Gdx;           // [N, DX]   G tensors for the weight gradient
  float* GdH2;          // [N, NH2]
  float* GdH1;          // [N, NH1]
  const float* wpack;   // NodeBwdSeq images
};

struct EdgeBwdArgs {
  int E;
  const int* src;
  const int* dst;
  const float* dM;      // [N, 2 DM] (past | future) or nullptr when MSGS == false
  const float* de_out;  // [E, DE] gradient of this layer's e'
  const float* sH1;
  const float* sH2;
  const float* sF1;
  const float* sP1;
  float* de_in;         // [E, DE] gradient of this layer's input e
  float* da_acc;        // [E, DA] running gradient of att_edge_attr (CLR) or nullptr
  int da_first;
  float* gdst;          // [E, 2 DX]
  float* gsrc;          // [E, 2 DX]
  float* GdH1;          // [E, EH1]
  float* GdH2;          // [E, EH2]
  float* Gde;           // [E, DE]   total gradient of e'
  float* GdF1;          // [E, MH]
  float* GdP1;          // [E, MH]
  const float* wpack;   // EdgeBwdSeq / EdgeBwdSeqNoMsg images
};

// ------------------------------------------------------------------------------------------
template <class D, int NW>
__global__ __launch_bounds__(NW * 64, NW >= 8 ? 2 : 1) void mp_edge_fwd_kernel(const EdgeFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  using Seq = typename D::EdgeFwdSeq;
  constexpr int XB = D::DX / 16, EB = D::DE / 16, AB = D::DA / 16;
  constexpr int H1B = D::EH1 / 16, H2B = D::EH2 / 16, MHB = D::MH / 16, DMB = D::DM / 16;
  WStreamT<NW * 64> ws;
  ws.init(a.wpack, smem);
  ws.template start<Seq>();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ntiles = (a.E + NW * 16 - 1) / (NW * 16);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const bool more = tile + (int)gridDim.x < ntiles;
    const long row = (long)tile * (NW * 16) + wave * kRowsPerWave + (lane & 15);
    const bool valid = row < a.E;
    int s = 0, d = 0;
    if (valid) { s = a.src[row]; d = a.dst[row]; }

    v4f in1[2 * XB + EB + AB];                 // x_i | x_j | e (| a)
    load_row<XB>(a.x, d, D::DX, 0, valid, in1);
    load_row<XB>(a.x, s, D::DX, 0, valid, in1 + XB);
    load_row<EB>(a.e_in, row, D::DE, 0, valid, in1 + 2 * XB);
    if constexpr (AB > 0) load_row<AB>(a.a_in, row, D::DA, 0, valid, in1 + 2 * XB + EB);
    v4f x0i[XB], x0j[XB];
    load_row<XB>(a.x0, d, D::DX, 0, valid, x0i);
    load_row<XB>(a.x0, s, D::DX, 0, valid, x0j);

    // Every store is issued right AFTER the next layer's weight-chunk barrier (as that layer's hook):
    // the barrier drains vmcnt, so a store placed in front of it would be waited for at once.
    v4f h1[H1B], h2[H2B], en[EB];
    linear<Seq, 0, true>(ws, more, in1, h1);
    linear<Seq, 1, true>(ws, more, h1, h2, [&]() { if (a.sH1) store_row<H1B>(a.sH1, row, D::EH1, 0, valid, h1); });
    linear<Seq, 2, false>(ws, more, h2, en, [&]() { if (a.sH2) store_row<H2B>(a.sH2, row, D::EH2, 0, valid, h2); });

    v4f inm[2 * XB + EB], m1[MHB], mo[DMB], mo2[DMB];
    copy_blocks<XB>(inm, in1);                 // x_i | e' | x0_i
    copy_blocks<EB>(inm + XB, en);
    copy_blocks<XB>(inm + XB + EB, x0i);
    linear<Seq, 3, true>(ws, more, inm, m1, [&]() { store_row<EB>(a.e_out, row, D::DE, 0, valid, en); });
    linear<Seq, 4, false>(ws, more, m1, mo, [&]() { if (a.sF1) store_row<MHB>(a.sF1, row, D::MH, 0, valid, m1); });

    copy_blocks<XB>(inm, in1 + XB);            // x_j | e' | x0_j
    copy_blocks<XB>(inm + XB + EB, x0j);
    linear<Seq, 5, true>(ws, more, inm, m1, [&]() { store_row<DMB>(a.fut, row, D::DM, 0, valid, mo); });
    linear<Seq, 6, false>(ws, more, m1, mo2, [&]() { if (a.sP1) store_row<MHB>(a.sP1, row, D::MH, 0, valid, m1); });
    store_row<DMB>(a.past, row, D::DM, 0, valid, mo2);
  }
}

// (the node phase lives in b3d_node.hpp: four wavefronts per 16-row tile)

// ------------------------------------------------------------------------------------------
// MSGS == false: the last layer, whose node update (and therefore both message stacks) receives
// no gradient because the final x is not an output (pose_gnn.py:86, clr_att_gnn.py:188).
template <class D, bool MSGS, int NW>
__global__ __launch_bounds__(NW * 64, NW >= 8 ? 2 : 1) void mp_edge_bwd_kernel(const EdgeBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  using Seq = typename std::conditional<MSGS, typename D::EdgeBwdSeq, typename D::EdgeBwdSeqNoMsg>::type;
  constexpr int L0 = MSGS ? 4 : 0;             // index of edge_update.4^T in Seq
  constexpr int XB = D::DX / 16, EB = D::DE / 16, AB = D::DA / 16;
  constexpr int H1B = D::EH1 / 16, H2B = D::EH2 / 16, MHB = D::MH / 16, DMB = D::DM / 16;
  WStreamT<NW * 64> ws;
  ws.init(a.wpack, smem);
  ws.template start<Seq>();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ntiles = (a.E + NW * 16 - 1) / (NW * 16);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const bool more = tile + (int)gridDim.x < ntiles;
    const long row = (long)tile * (NW * 16) + wave * kRowsPerWave + (lane & 15);
    const bool valid = row < a.E;
    int s = 0, d = 0;
    if (valid) { s = a.src[row]; d = a.dst[row]; }

    v4f de[EB];                                 // total gradient of e'
    load_row<EB>(a.de_out, row, D::DE, 0, valid, de);
    v4f gd[2 * XB], gs[2 * XB];                 // (d x[dst] | d x0[dst]), (d x[src] | d x0[src])
#pragma unroll
    for (int b = 0; b < 2 * XB; ++b) { gd[b] = v4f{0.f, 0.f, 0.f, 0.f}; gs[b] = v4f{0.f, 0.f, 0.f, 0.f}; }

    if constexpr (MSGS) {
      v4f dmsg[DMB], act[MHB], dh[MHB], dxin[2 * XB + EB];
      // past stack: input was (x_j | e' | x0_j); its sum landed at dst -> gradient dM[dst][0:DM]
      load_row<DMB>(a.dM, d, 2 * D::DM, 0, valid, dmsg);
      load_row<MHB>(a.sP1, row, D::MH, 0, valid, act);
      linear<Seq, 0, false, false>(ws, more, dmsg, dh);
      relu_bwd<MHB>(dh, act);
      store_row<MHB>(a.GdP1, row, D::MH, 0, valid, dh);
      linear<Seq, 1, false, false>(ws, more, dh, dxin);
      copy_blocks<XB>(gs, dxin);
      add_blocks<EB>(de, dxin + XB);
      copy_blocks<XB>(gs + XB, dxin + XB + EB);
      // future stack: input was (x_i | e' | x0_i); summed at src -> gradient dM[src][DM:2DM]
      load_row<DMB>(a.dM, s, 2 * D::DM, D::DM, valid, dmsg);
      load_row<MHB>(a.sF1, row, D::MH, 0, valid, act);
      linear<Seq, 2, false, false>(ws, more, dmsg, dh);
      relu_bwd<MHB>(dh, act);
      store_row<MHB>(a.GdF1, row, D::MH, 0, valid, dh);
      linear<Seq, 3, false, false>(ws, more, dh, dxin);
      copy_blocks<XB>(gd, dxin);
      add_blocks<EB>(de, dxin + XB);
      copy_blocks<XB>(gd + XB, dxin + XB + EB);
    }
    store_row<EB>(a.Gde, row, D::DE, 0, valid, de);

    v4f act2[H2B], d2[H2B];
    load_row<H2B>(a.sH2, row, D::EH2, 0, valid, act2);
    linear<Seq, L0 + 0, false, false>(ws, more, de, d2);
    relu_bwd<H2B>(d2, act2);
    store_row<H2B>(a.GdH2, row, D::EH2, 0, valid, d2);
    v4f act1[H1B], d1[H1B];
    load_row<H1B>(a.sH1, row, D::EH1, 0, valid, act1);
    linear<Seq, L0 + 1, false, false>(ws, more, d2, d1);
    relu_bwd<H1B>(d1, act1);
    store_row<H1B>(a.GdH1, row, D::EH1, 0, valid, d1);
    v4f dx1[2 * XB + EB + AB];                  // d(x_i | x_j | e | a)
    linear<Seq, L0 + 2, false, false>(ws, more, d1, dx1);
    add_blocks<XB>(gd, dx1);
    add_blocks<XB>(gs, dx1 + XB);
    store_row<EB>(a.de_in, row, D::DE, 0, valid, dx1 + 2 * XB);
    if constexpr (AB > 0) {
      if (!a.da_first) {
        v4f prev[AB];
        load_row<AB>(a.da_acc, row, D::DA, 0, valid, prev);
        add_blocks<AB>(dx1 + 2 * XB + EB, prev);
      }
      store_row<AB>(a.da_acc, row, D::DA, 0, valid, dx1 + 2 * XB + EB);
    }
    store_row<2 * XB>(a.gdst, row, 2 * D::DX, 0, valid, gd);
    store_row<2 * XB>(a.gsrc, row, 2 * D::DX, 0, valid, gs);
  }
}

}  // namespace b3d
