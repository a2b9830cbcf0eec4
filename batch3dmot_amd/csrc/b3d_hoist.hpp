// Hoisted first layers of the three edge stacks.
//
// Every stack of CausalMessagePassing starts with a Linear over a CONCATENATION of gathered node rows
// and the edge row (pose_gnn.py:205-226):
//     edge_update.0        ( x[dst] | x[src] | e  )
//     create_future_msgs.0 ( x[dst] | e' | x0[dst] )
//     create_past_msgs.0   ( x[src] | e' | x0[src] )
// W . (u | v | w) = W_u u + W_v v + W_w w, and the node parts depend on the NODE only.  They are
// evaluated once per node (N = 3,000 rows) instead of once per edge (E = 30,000 rows):
//     T[n] = ( W_eu[:, x_i] x[n] + b_eu | W_eu[:, x_j] x[n] | W_fu[:, x] x[n] + W_fu[:, x0] x0[n] + b_fu |
//              W_pa[:, x] x[n] + W_pa[:, x0] x0[n] + b_pa )                       [N, 2 EH1 + 2 MH]
// and the edge kernel starts its accumulators from the gathered rows T[dst] / T[src] and multiplies
// only the 32 edge columns.  Per edge and layer 29,696 MAC instead of 57,344; the same split runs
// through the data gradient (per-node segment sums of the first-layer gradients, then one per-node
// product with the transposed node columns) and the weight gradient (node columns contract over
// nodes).  The x0 terms are layer-invariant and computed once per forward.  Summation order differs
// from the unsplit Linear by fp32 rounding only.
#pragma once
#include "b3d_mp.hpp"
#include "b3d_node.hpp"
#include "b3d_chain.hpp"

// wavefronts per by-source list half in the per-node gradient kernels (1..3) and gather depth
#ifndef B3D_SRC_PARTS
#define B3D_SRC_PARTS 3
#endif
#ifndef B3D_SEG_U
#define B3D_SEG_U (B3D_SRC_PARTS == 3 ? 6 : 8)
#endif
namespace b3d {

template <class D>
struct Hoist {
  static constexpr int GW = 2 * D::EH1 + 2 * D::MH;        // per-node parts of the three first layers (and width of dT)
  static constexpr int OG = GW;                             // + GATConv.lin(x) of the discarded k-NN block (pose_gnn.py:79)
  static constexpr int TW = GW + D::DX;                     // width of the per-node table
  static constexpr int OA = 0, OB = D::EH1, OF = 2 * D::EH1, OP = 2 * D::EH1 + D::MH;
  static constexpr int KE = D::DE + D::DA;                  // per-edge input columns of edge_update.0
  using ProjSeq = LayerSeq<typename D::template NL<D::DX, TW>>;                   // x[l]  -> T (without the x0 terms)
  using Proj0Seq = LayerSeq<typename D::template NL<D::DX, 2 * D::MH>>;           // x0    -> x0 terms of the F | P columns
  using EdgeFwdSeq = LayerSeq<typename D::template LL<KE, D::EH1>, typename D::template LL<D::EH1, D::EH2>, typename D::template LL<D::EH2, D::DE>,     // edge_update (.0: edge columns)
                              typename D::template LL<D::DE, D::MH>, typename D::template LL<D::MH, D::DM>,                       // create_future_msgs
                              typename D::template LL<D::DE, D::MH>, typename D::template LL<D::MH, D::DM>>;                      // create_past_msgs
  // transposed images, data-gradient order; the .0 layers keep their edge columns only
  using EdgeBwdSeq = LayerSeq<typename D::template LL<D::DM, D::MH>, typename D::template LL<D::MH, D::DE>,                       // past.2^T, past.0[e']^T
                              typename D::template LL<D::DM, D::MH>, typename D::template LL<D::MH, D::DE>,                       // future.2^T, future.0[e']^T
                              typename D::template LL<D::DE, D::EH2>, typename D::template LL<D::EH2, D::EH1>, typename D::template LL<D::EH1, KE>>;    // edge_update.4^T/.2^T/.0[e]^T
  using EdgeBwdSeqNoMsg = LayerSeq<typename D::template LL<D::DE, D::EH2>, typename D::template LL<D::EH2, D::EH1>, typename D::template LL<D::EH1, KE>>;
  // per-node gradient of (x | x0) from the gradient of T: [W_eu_i^T | W_eu_j^T | W_fu_x^T | W_pa_x^T ; 0 | 0 | W_fu_x0^T | W_pa_x0^T]
  // ... as four accumulating products, one per list (all output blocks of a product sit in one weight chunk,
  // so every owning wavefront works at once): dH1-by-dst and dH1-by-src reach dx only
  using GradProjSeq = LayerSeq<typename D::template NL<D::EH1, D::DX>, typename D::template NL<D::EH1, D::DX>, typename D::template NL<D::MH, 2 * D::DX>, typename D::template NL<D::MH, 2 * D::DX>>;
};

// Storer of the projection chain: the F | P columns receive the layer-invariant x0 terms.
template <class D>
struct StoreProj {
  using H = Hoist<D>;
  static constexpr int NB = H::TW / 16;
  float* T;            // [rows, TW]
  const float* T0;     // [rows, 2 MH]
  __device__ __forceinline__ void operator()(long row, bool valid, const v4f* src) const {
    constexpr int FB = H::OF / 16, T0B = 2 * D::MH / 16, GB = H::OG / 16;
    v4f t0[T0B];
    load_row<T0B>(T0, row, 2 * D::MH, 0, valid, t0);
    store_row<FB>(T, row, H::TW, 0, valid, src);
    v4f out[T0B];
#pragma unroll
    for (int b = 0; b < T0B; ++b) out[b] = src[FB + b] + t0[b];
    store_row<T0B>(T, row, H::TW, H::OF, valid, out);
    store_row<NB - GB>(T, row, H::TW, H::OG, valid, src + GB);
  }
};

struct EdgeFwdHArgs {
  int E;
  const int* src;
  const int* dst;
  const float* T;      // [N, TW] per-node parts of the three first layers (this layer's x)
  const float* e_in;   // [E, DE]
  const float* a_in;   // [E, DA] or nullptr
  float* e_out;
  float* fut;
  float* past;
  float* sH1;
  float* sH2;
  float* sF1;
  float* sP1;
  const float* wpack;  // Hoist::EdgeFwdSeq images
  unsigned* rmask;     // ReLU masks (b3d_dev.hpp).  es::edge_fwd_kernel<D, true>: plane A (sH1 | sH2), 64 bytes per edge;
  unsigned* rmask2;    //   plane B (sF1 | sP1).  mp_edge_fwd_h_kernel: rmask = one [E, 16]-float plane (a word per tensor), or nullptr
  // es::edge_fwd_kernel (round 6): per-destination sums of `past` inside the wavefront when *dst_unsorted == 0 (b3d.h: b3d_graph) --
  // rows that are not the last of their (destination, 16-row block) run go to the dump rows [past_dump0, past_dump0 + 1024) of `past`
  const int* dst_unsorted;   // nullptr: one row per edge
  unsigned past_dump0;
};

template <class D, int NW>
__global__ __launch_bounds__(NW * 64, 2) void mp_edge_fwd_h_kernel(const EdgeFwdHArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  using H = Hoist<D>;
  using Seq = typename H::EdgeFwdSeq;
  constexpr int EB = D::DE / 16, AB = D::DA / 16;
  constexpr int H1B = D::EH1 / 16, H2B = D::EH2 / 16, MHB = D::MH / 16, DMB = D::DM / 16;
  B3D_STAMP(2, 0);
  WStreamG<NW * 64, Seq> ws;                  // hoisted stacks: 126 KB of weights, resident in LDS
  ws.init(a.wpack, smem);
  ws.template start<Seq>();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ntiles = (a.E + NW * 16 - 1) / (NW * 16);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const bool more = tile + (int)gridDim.x < ntiles;
    const long row = (long)tile * (NW * 16) + wave * kRowsPerWave + (lane & 15);
    const bool valid = row < a.E;
    const long rc = valid ? row : (long)a.E - 1;             // rows past the end read the last edge (never stored)
    const int s = a.src[rc], d = a.dst[rc];

    // gathers: unconditional loads straight into their registers, all in flight together; the future / past
    // rows are fetched a layer ahead of their use (as hooks) to keep the prologue light.  The weights are resident
    // in LDS (WStreamG): one barrier in front of the first layer of the first tile, none afterwards.
    v4f ein[EB + AB];
    load_row_u<EB>(a.e_in, rc, D::DE, 0, ein);
    if constexpr (AB > 0) load_row_u<AB>(a.a_in, rc, D::DA, 0, ein + EB);
    v4f h1[H1B], tb[H1B], fi[MHB], pi[MHB];
    load_row_u<H1B>(a.T, d, H::TW, H::OA, h1);
    load_row_u<H1B>(a.T, s, H::TW, H::OB, tb);
    wait_for(ein); wait_for(h1); wait_for(tb);             // every prologue load has landed before the first acquire
    add_blocks<H1B>(h1, tb);
    B3D_STAMP(2, 1);

    // wide stacks (camera+LiDAR+radar: 256-wide hidden layer) fetch the past-stack rows two layers later: a layer
    // there is several microseconds long and the register file is full
    constexpr bool LATE_PI = D::EH1 > 128;
    v4f h2[H2B], en[EB];
    // ReLU masks of sH1 | sH2 | sF1 | sP1 (b3d_dev.hpp): one word each at these widths, the lane's 16 bytes of a [E, 16]-float plane
    static_assert(H1B <= 8 && H2B <= 8 && MHB <= 8, "one mask word per tensor (the wide camera+LiDAR+radar stacks use b3d_edge2.hpp)");
    unsigned mk1[1] = {0u}, mk2[1] = {0u}, mkf[1] = {0u}, mkp[1] = {0u};
    linear_init<Seq, 0, true, false>(ws, more, ein, h1, h1);
    B3D_STAMP(2, 2);
    linear<Seq, 1, true>(ws, more, h1, h2, [&]() {
      B3D_STAMP(2, 10);
      if (a.sH1) store_row<H1B>(a.sH1, row, D::EH1, 0, valid, h1);
      if (a.rmask) relu_mask_words<H1B>(h1, mk1);
      load_row_u<MHB>(a.T, d, H::TW, H::OF, fi);
      B3D_STAMP(2, 11);
    });
    B3D_STAMP(2, 3);
    linear<Seq, 2, false>(ws, more, h2, en, [&]() {
      B3D_STAMP(2, 12);
      if (a.sH2) store_row<H2B>(a.sH2, row, D::EH2, 0, valid, h2);
      if (a.rmask) relu_mask_words<H2B>(h2, mk2);
      if constexpr (!LATE_PI) load_row_u<MHB>(a.T, s, H::TW, H::OP, pi);
      B3D_STAMP(2, 13);
    });
    B3D_STAMP(2, 4);

    v4f mo[DMB], mo2[DMB];
    wait_for(fi);
    linear_init<Seq, 3, true, false>(ws, more, en, fi, fi, [&]() {
      B3D_STAMP(2, 14);
      store_row<EB>(a.e_out, row, D::DE, 0, valid, en);
      if constexpr (LATE_PI) load_row_u<MHB>(a.T, s, H::TW, H::OP, pi);
      B3D_STAMP(2, 15);
    });
    B3D_STAMP(2, 5);
    linear<Seq, 4, false>(ws, more, fi, mo, [&]() {
      B3D_STAMP(2, 16);
      if (a.sF1) store_row<MHB>(a.sF1, row, D::MH, 0, valid, fi);
      if (a.rmask) relu_mask_words<MHB>(fi, mkf);
      B3D_STAMP(2, 17);
    });
    B3D_STAMP(2, 6);
    wait_for(pi);
    linear_init<Seq, 5, true, false>(ws, more, en, pi, pi, [&]() { B3D_STAMP(2, 18); store_row<DMB>(a.fut, row, D::DM, 0, valid, mo); B3D_STAMP(2, 19); });
    B3D_STAMP(2, 7);
    linear<Seq, 6, false>(ws, more, pi, mo2, [&]() {
      B3D_STAMP(2, 20);
      if (a.sP1) store_row<MHB>(a.sP1, row, D::MH, 0, valid, pi);
      if (a.rmask) {
        relu_mask_words<MHB>(pi, mkp);
        if (valid) *reinterpret_cast<u4v*>(reinterpret_cast<char*>(a.rmask) + ((size_t)row * 4 + (lane >> 4)) * 16) = u4v{mk1[0], mk2[0], mkf[0], mkp[0]};
      }
      B3D_STAMP(2, 21);
    });
    B3D_STAMP(2, 8);
    store_row<DMB>(a.past, row, D::DM, 0, valid, mo2);
    B3D_STAMP(2, 9);
  }
}

// Node update + the per-node table of the NEXT layer in one launch (b3d_node.hpp, PROJ stage).
template <class D>
using NodeFwdHSeq = LayerSeq<typename D::template NL<D::NIN, D::NH1>, typename D::template NL<D::NH1, D::NH2>, typename D::template NL<D::NH2, D::DX>,
                             typename D::template NL<D::DX, Hoist<D>::TW>>;

template <class D>
__global__ __launch_bounds__(kNodeWavesWide * 64, 1) void mp_node_fwd_split_h_kernel(const NodeFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  node_fwd_split_body<D, NodeFwdHSeq<D>, true, kNodeWavesWide>(a, smem);
}

// x0 terms + the table of layer 0 from the node encoder's output, four wavefronts per 16-row tile (the
// 48 -> 192 and 48 -> 432 products are 468 MFMAs: too long a chain for one wavefront).
template <class D>
using Proj0Seq2 = LayerSeq<typename D::template NL<D::DX, 2 * D::MH>, typename D::template NL<D::DX, Hoist<D>::TW>>;
struct NodeProj0Args {
  int N;
  const float* x = nullptr;   // [N, DX] the layer's x (nullptr: x0 -- layer 0 of a model, where x == initial_x)
  const float* x0;      // [N, DX]
  float* T0;            // [N, 2 MH]
  float* T;             // [N, TW]
  const float* wpack;   // Proj0Seq2 images
};
// NWS: four wavefronts per tile; eight (a 128-row weight chunk is one block per wavefront instead of two) is an experiment of round 6: slower
template <class D, int NWS = kNodeWaves>
__global__ __launch_bounds__(NWS * 64, 1) void node_proj0_split_kernel(const NodeProj0Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  using H = Hoist<D>;
  using Seq = Proj0Seq2<D>;
  constexpr int XB = D::DX / 16, FB = H::OF / 16, T0B = 2 * D::MH / 16;
  static_assert(FB % NWS == 0 && T0B % NWS == 0, "table columns must split over the wavefronts");
  NodeRing<NWS * 64> ws;      // (ring form:) one barrier per weight chunk: it is also what publishes the previous layer's LDS activations
  ws.init(a.wpack, smem);
  ws.template start<Seq>();
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 16 + (lane & 15);
  const bool valid = row < a.N;
  v4f x[XB];
  load_row<XB>(a.x0, row, D::DX, 0, valid, x);
  v4f t0[T0B / NWS];                          // this wavefront's blocks of the x0 terms (block j <-> table block FB + j)
  linear_split<Seq, 0, false, false, NWS>(
      ws, false, x, [&]() {},
      [&](int mb, v4f v, int slot) {
        t0[slot] = v;
        store_row<1>(a.T0, row, 2 * D::MH, 16 * mb, valid, &v);
      });
  if (a.x) load_row<XB>(a.x, row, D::DX, 0, valid, x);      // (its own loads: nothing of the first layer's reads is pending)
  linear_split<Seq, 1, false, true, NWS>(
      ws, false, x, [&]() {},
      [&](int mb, v4f v, int slot) {
        if (mb >= FB && mb < FB + T0B) v += t0[slot - FB / NWS];
        store_row<1>(a.T, row, H::TW, 16 * mb, valid, &v);
      });
}

// ---- backward -------------------------------------------------------------------------------------
struct EdgeBwdHArgs {
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
  float* da_acc;        // [E, DA] running gradient of att_edge_attr or nullptr
  int da_first;
  float* GdH1;          // [E, EH1]  G tensors: weight gradient AND per-node sums of the first layers
  float* GdH2;
  float* Gde;
  float* GdF1;
  float* GdP1;
  const float* wpack;   // Hoist::EdgeBwdSeq / EdgeBwdSeqNoMsg images
  const unsigned* rmask;   // the forward's ReLU masks (instead of reading sH1 .. sP1 back): es::edge_bwd_kernel planes A and B,
  const unsigned* rmask2;  //   mp_edge_bwd_h_kernel one plane
};

// Data gradient of the edge phase without the node columns of the three first layers: those are
// contracted per NODE from the segment sums of GdH1 / GdF1 / GdP1 (node_gradproj_kernel).
template <class D, bool MSGS, int NW>
__global__ __launch_bounds__(NW * 64, 2) void mp_edge_bwd_h_kernel(const EdgeBwdHArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  using H = Hoist<D>;
  using Seq = typename std::conditional<MSGS, typename H::EdgeBwdSeq, typename H::EdgeBwdSeqNoMsg>::type;
  if constexpr (MSGS) B3D_STAMP(3, 0);
  constexpr int L0 = MSGS ? 4 : 0;
  constexpr int EB = D::DE / 16, AB = D::DA / 16;
  constexpr int H1B = D::EH1 / 16, H2B = D::EH2 / 16, MHB = D::MH / 16, DMB = D::DM / 16;
  WStreamG<NW * 64, Seq> ws;                  // resident weights (128 KB) or a two-slot ring
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
    // Loads are issued a whole layer before their use and stores as the next layer's hook: both have a layer of
    // MFMAs to complete under.  (With a weight RING every chunk acquire drained vmcnt -- the LDS-DMA shares the
    // counter with loads and stores; with the weights resident only the first acquire of the first tile does.)
    v4f de[EB];
    load_row<EB>(a.de_out, row, D::DE, 0, valid, de);
    v4f d2[H2B], d1[H1B];
    // the forward's ReLU masks (words: sH1, sH2, sF1, sP1) instead of the saved activations themselves: 16 bytes per lane for 1.4 KB per edge
    static_assert(H1B <= 8 && H2B <= 8 && MHB <= 8, "one mask word per tensor");
    u4v mk = {0u, 0u, 0u, 0u};
    if (valid) mk = *reinterpret_cast<const u4v*>(reinterpret_cast<const char*>(a.rmask) + ((size_t)row * 4 + (lane >> 4)) * 16);
    if constexpr (MSGS) {
      v4f dmp[DMB], dmf[DMB], dh[MHB], dh2[MHB], dee[EB];
      load_row<DMB>(a.dM, d, 2 * D::DM, 0, valid, dmp);            // past messages were summed at dst
      load_row<DMB>(a.dM, s, 2 * D::DM, D::DM, valid, dmf);        // future messages were summed at src
      wait_for(de); wait_for(dmp); wait_for(dmf);                   // the prologue's loads have landed
      B3D_STAMP(3, 1);
      linear<Seq, 0, false, false>(ws, more, dmp, dh);
      B3D_STAMP(3, 2);
      { const unsigned wd[1] = {mk.w}; relu_bwd_words<MHB>(dh, wd); }
      linear<Seq, 1, false, false>(ws, more, dh, dee, [&]() { store_row<MHB>(a.GdP1, row, D::MH, 0, valid, dh); });
      B3D_STAMP(3, 3);
      add_blocks<EB>(de, dee);
      linear<Seq, 2, false, false>(ws, more, dmf, dh2);
      B3D_STAMP(3, 4);
      { const unsigned wd[1] = {mk.z}; relu_bwd_words<MHB>(dh2, wd); }
      linear<Seq, 3, false, false>(ws, more, dh2, dee, [&]() { store_row<MHB>(a.GdF1, row, D::MH, 0, valid, dh2); });
      B3D_STAMP(3, 5);
      add_blocks<EB>(de, dee);
    } else {
      wait_for(de);
    }
    linear<Seq, L0 + 0, false, false>(ws, more, de, d2, [&]() { store_row<EB>(a.Gde, row, D::DE, 0, valid, de); });
    if constexpr (MSGS) B3D_STAMP(3, 6);
    { const unsigned wd[1] = {mk.y}; relu_bwd_words<H2B>(d2, wd); }
    linear<Seq, L0 + 1, false, false>(ws, more, d2, d1, [&]() { store_row<H2B>(a.GdH2, row, D::EH2, 0, valid, d2); });
    if constexpr (MSGS) B3D_STAMP(3, 7);
    { const unsigned wd[1] = {mk.x}; relu_bwd_words<H1B>(d1, wd); }
    v4f dein[EB + AB];
    linear<Seq, L0 + 2, false, false>(ws, more, d1, dein, [&]() { store_row<H1B>(a.GdH1, row, D::EH1, 0, valid, d1); });
    if constexpr (MSGS) B3D_STAMP(3, 8);
    store_row<EB>(a.de_in, row, D::DE, 0, valid, dein);
    if constexpr (AB > 0) {
      if (!a.da_first) {
        v4f prev[AB];
        load_row<AB>(a.da_acc, row, D::DA, 0, valid, prev);
        add_blocks<AB>(dein + EB, prev);
      }
      store_row<AB>(a.da_acc, row, D::DA, 0, valid, dein + EB);
    }
  }
}

// Per-node gradient of the table T and of (x | x0), 4 wavefronts per 16-row tile:
//   dT[n] = ( sum_{dst=n} dH1 | sum_{src=n} dH1 | sum_{dst=n} dF1 | sum_{src=n} dP1 )      (kept: weight gradient)
//   gx[n] = ( dx | dx0 contribution ) = GradProj . dT[n]
struct NodeGradProjArgs {
  int N;
  const int* dst_ptr;
  const int* dst_perm;
  const int* src_ptr;
  const int* src_perm;
  const float* GdH1;    // [E, EH1]
  const float* GdF1;    // [E, MH] or nullptr (last layer: the message stacks carry no gradient)
  const float* GdP1;
  float* dT;            // [N, GW]
  float* gx;            // [N, 2 DX]
  const float* wpack;   // Hoist::GradProjSeq image
};

template <class D>
struct GradProjLds {
  static constexpr int TB = Hoist<D>::GW / 16;
  static constexpr int PARTB = (B3D_SRC_PARTS - 1) * 2 * (D::EH1 / 16);       // partial sums of the by-source lists (kSrcParts - 1 each)
  static constexpr int BYTES = kLdsBytes + (TB + PARTB) * 64 * 16;
};

// (dx | dx0) = sum over the four lists of (node columns)^T . dT_list: layers LI0 .. LI0+3 of Seq.  The wavefront
// that owns output block mb (mb % NWS == wave, the same in all four products) keeps its partial in `keep`.
template <class Seq, int LI0, int NWS, int LB, class WS, class Hook>
__device__ __forceinline__ v4f gradproj_products(WS& ws, const v4f* __restrict__ xt, int lane, Hook first) {
  v4f keep = {0.f, 0.f, 0.f, 0.f};
  v4f dt[LB];
  auto acc = [&](int, v4f v) { keep += v; };
  linear_split<Seq, LI0 + 0, false, false, NWS>(ws, false, dt, [&]() {
    first();                                   // work that must sit BEHIND the first barrier (stores of the list sums)
#pragma unroll
    for (int b = 0; b < LB; ++b) dt[b] = xt[(0 * LB + b) * 64 + lane];
  }, acc);
  linear_split<Seq, LI0 + 1, false, false, NWS>(ws, false, dt, [&]() {
#pragma unroll
    for (int b = 0; b < LB; ++b) dt[b] = xt[(1 * LB + b) * 64 + lane];
  }, acc);
  linear_split<Seq, LI0 + 2, false, false, NWS>(ws, false, dt, [&]() {
#pragma unroll
    for (int b = 0; b < LB; ++b) dt[b] = xt[(2 * LB + b) * 64 + lane];
  }, acc);
  linear_split<Seq, LI0 + 3, false, false, NWS>(ws, false, dt, [&]() {
#pragma unroll
    for (int b = 0; b < LB; ++b) dt[b] = xt[(3 * LB + b) * 64 + lane];
  }, acc);
  return keep;
}

constexpr int kSrcParts = B3D_SRC_PARTS;
constexpr int kGradProjWaves = 4 + 4 * kSrcParts;
// Sixteen wavefronts sum the four lists (dH1 by dst, dH1 by src, dF1 by dst, dP1 by src) of a 16-row tile.  Fewer
// than 256 tiles make a batch, so the launch lasts as long as its WORST tile: the by-destination lists are short
// (in-degree: the few frames behind a detection), the by-source lists of a tracking graph reach 40+ entries.
//   waves 0..3  : by-destination lists, one wavefront per half of the feature blocks;
//   waves 4..15 : by-source lists, kSrcParts wavefronts per half, each a contiguous third of the list; the
//                 partial sums meet in LDS and are added in a fixed order by the wavefront of the first third.
// The owner wavefronts leave the list sums in xt[(list * LB + block)] and in `part`.
struct ListRole { int list, half, third; bool owner; };

template <class D>
__device__ __forceinline__ ListRole gradproj_list_sums(const NodeGradProjArgs& gp, v4f* __restrict__ xt, v4f* __restrict__ xp,
                                                       v4f* __restrict__ part) {
  static_assert(kGradProjWaves == 4 + 4 * kSrcParts && D::EH1 == D::MH && (D::EH1 / 16) % 2 == 0, "two halves per list, equal widths");
  constexpr int LB = D::EH1 / 16, HB = LB / 2;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // gathers run in layout Q (lane = 4 row + q: b3d_dev.hpp); the sums reach layout L through their LDS slot
  const long row = (long)blockIdx.x * 16 + q_row(lane);
  const bool valid = row < gp.N;
  const int slot = q_slot(lane);
  ListRole r;
  if (wave < 4) { r.list = (wave & 1) * 2; r.half = wave >> 1; r.third = 0; }
  else { const int j = wave - 4; r.list = 1 + (j & 1) * 2; r.half = (j >> 1) & 1; r.third = j >> 2; }
  r.owner = r.third == 0;
#pragma unroll
  for (int b = 0; b < HB; ++b) part[b] = v4f{0.f, 0.f, 0.f, 0.f};
  const float* base = (r.list < 2) ? gp.GdH1 : (r.list == 2 ? gp.GdF1 : gp.GdP1);
  const bool by_dst = !(r.list & 1);
  if (valid && base) {
    constexpr int U = B3D_SEG_U;
    if (by_dst) {
      segment_sum_deep_q<HB, U>(base, 16 * LB, 16 * HB * r.half, gp.dst_perm, gp.dst_ptr[row], gp.dst_ptr[row + 1], part, q_piece(lane));
    } else {
      const int beg = gp.src_ptr[row], len = gp.src_ptr[row + 1] - beg;
      segment_sum_deep_q<HB, U>(base, 16 * LB, 16 * HB * r.half, gp.src_perm, beg + len * r.third / kSrcParts,
                                beg + len * (r.third + 1) / kSrcParts, part, q_piece(lane));
    }
  }
  v4f* own = xt + (r.list * LB + r.half * HB) * 64 + slot;
  {
    v4f* dstp = r.owner ? own : xp + (((r.third - 1) * 2 + (r.list >> 1)) * LB + r.half * HB) * 64 + slot;
#pragma unroll
    for (int b = 0; b < HB; ++b) dstp[b * 64] = part[b];
  }
  if constexpr (kSrcParts > 1) {
    const v4f* others = xp + ((r.list >> 1) * LB + r.half * HB) * 64 + slot;     // + (third - 1) * 2 LB blocks
    __syncthreads();
    if (!by_dst && r.owner) {
#pragma unroll
      for (int b = 0; b < HB; ++b) {
#pragma unroll
        for (int t = 1; t < kSrcParts; ++t) part[b] += others[((t - 1) * 2 * LB + b) * 64];
        own[b * 64] = part[b];
      }
    }
  }
  return r;
}
// the owner's store of its list sums (held in layout Q) into dT
template <class D>
__device__ __forceinline__ void gradproj_store_sums(const ListRole& r, float* __restrict__ dT, int N, const v4f* __restrict__ part) {
  constexpr int LB = D::EH1 / 16, HB = LB / 2;
  const long row = (long)blockIdx.x * 16 + q_row(threadIdx.x & 63);
  if (r.owner) store_row_q<HB>(dT, row, Hoist<D>::GW, 16 * (LB * r.list + HB * r.half), row < N, part);
}

template <class D>
__global__ __launch_bounds__(kGradProjWaves * 64, 1) void node_gradproj_kernel(const NodeGradProjArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  using H = Hoist<D>;
  using Seq = typename H::GradProjSeq;
  constexpr int NWS = kGradProjWaves;
  constexpr int LB = D::EH1 / 16, HB = LB / 2, TB = H::GW / 16;
  NodeRing<NWS * 64> ws;      // (ring form:) one barrier per weight chunk: it is also what publishes the previous layer's LDS activations
  ws.init(a.wpack, smem);
  ws.template start<Seq>();
  v4f* xb = reinterpret_cast<v4f*>(smem + 2 * kWBufFloats);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 16 + (lane & 15);
  const bool valid = row < a.N;
  v4f part[HB];
  const ListRole r = gradproj_list_sums<D>(a, xb, xb + TB * 64, part);
  const v4f keep = gradproj_products<Seq, 0, NWS, LB>(ws, xb, lane, [&]() { gradproj_store_sums<D>(r, a.dT, a.N, part); });
  if (wave < 2 * D::DX / 16) store_row<1>(a.gx, row, 2 * D::DX, 16 * wave, valid, &keep);
  (void)TB;
}

// node_gradproj + the node update's data gradient in ONE launch (layers 0 .. depth-2): the per-node
// (dx | dx0) never leaves the CU.  8 wavefronts per 16-row tile throughout.
template <class D>
using NodeBwdHSeq = LayerSeq<typename D::template NL<D::EH1, D::DX>, typename D::template NL<D::EH1, D::DX>,
                             typename D::template NL<D::MH, 2 * D::DX>, typename D::template NL<D::MH, 2 * D::DX>,   // GradProj
                             typename D::template NL<D::DX, D::NH2>, typename D::template NL<D::NH2, D::NH1>,
                             typename D::template NL<D::NH1, D::NIN>>;                          // combine_future_past^T
struct NodeBwdHArgs {
  NodeGradProjArgs gp;  // lists of layer l+1, dT of layer l+1 (gp.gx unused)
  float* dx0_acc;       // [N, DX] running gradient of initial_x
  int dx0_first;
  const float* sH1;     // saved activations of layer l's node MLP
  const float* sH2;
  float* dM;            // [N, 2 DM]
  float* Gdx;           // [N, DX]  G tensors of layer l's node MLP
  float* GdH2;
  float* GdH1;
  const float* wpack;   // NodeBwdHSeq images
};

template <class D>
struct NodeBwdHLds {
  static constexpr int TB = Hoist<D>::GW / 16, PB = (D::NH1 > 2 * D::DX ? D::NH1 : 2 * D::DX) / 16;
  static constexpr int PARTB = GradProjLds<D>::PARTB;
  static constexpr int XBB = 2 * PB > PARTB ? 2 * PB : PARTB;     // the ping-pong buffers start life as the partial-sum area
  static constexpr int BYTES = kLdsBytes + (TB + XBB) * 64 * 16;
  static_assert(BYTES <= 160 * 1024, "LDS of one CU");
};

template <class D>
__global__ __launch_bounds__(kGradProjWaves * 64, 1) void node_bwd_h_kernel(const NodeBwdHArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  using H = Hoist<D>;
  using Seq = NodeBwdHSeq<D>;
  B3D_STAMP(1, 0);
  constexpr int NWS = kGradProjWaves;
  constexpr int LB = D::EH1 / 16, HB = LB / 2, TB = H::GW / 16;
  constexpr int XB = D::DX / 16, GB = 2 * XB, H1B = D::NH1 / 16, H2B = D::NH2 / 16;
  static_assert(GB <= NWS && H1B <= NWS && H2B <= NWS, "at most one output block per wavefront and layer");
  constexpr int PB = NodeBwdHLds<D>::PB;
  NodeRing<NWS * 64> ws;      // (ring form:) one barrier per weight chunk: it is also what publishes the previous layer's LDS activations
  ws.init(a.wpack, smem);
  ws.template start<Seq>();
  v4f* xt = reinterpret_cast<v4f*>(smem + 2 * kWBufFloats);   // dT tile
  v4f* xb0 = xt + TB * 64;                                     // ping-pong for the stages behind it
  v4f* xb1 = xb0 + PB * 64;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 16 + (lane & 15);
  const bool valid = row < a.gp.N;
  // saved activations: only the block this wavefront owns in the masked layer (the owner masks what it emits);
  // loaded now, in the shadow of the list sums: an acquire in front of their use would expose them
  v4f act2 = {0.f, 0.f, 0.f, 0.f}, act1 = {0.f, 0.f, 0.f, 0.f};
  if (wave < H2B) load_row<1>(a.sH2, row, D::NH2, 16 * wave, valid, &act2);
  if (wave < H1B) load_row<1>(a.sH1, row, D::NH1, 16 * wave, valid, &act1);
  // running d initial_x of the block this wavefront will own
  v4f prev0 = {0.f, 0.f, 0.f, 0.f};
  if (!a.dx0_first && wave >= XB && wave < GB) load_row<1>(a.dx0_acc, row, D::DX, 16 * (wave - XB), valid, &prev0);
  v4f part[HB];
  B3D_STAMP(1, 1);
  const ListRole r = gradproj_list_sums<D>(a.gp, xt, xb0, part);
  B3D_STAMP(1, 2);
  // this wavefront's block of (dx | dx0): block `wave` (< 2 XB); stored behind the next barrier
  v4f own = gradproj_products<Seq, 0, NWS, LB>(ws, xt, lane, [&]() { gradproj_store_sums<D>(r, a.gp.dT, a.gp.N, part); });
  B3D_STAMP(1, 3);
  if (wave < XB) xb0[wave * 64 + lane] = own;                 // d x': input of the node MLP's data gradient
  else if (wave < GB) own += prev0;
  v4f g[XB], d2[H2B], d1[H1B];
  auto mask = [](v4f v, v4f act) {
    return v4f{act.x > 0.f ? v.x : 0.f, act.y > 0.f ? v.y : 0.f, act.z > 0.f ? v.z : 0.f, act.w > 0.f ? v.w : 0.f};
  };
  linear_split<Seq, 4, false, false, NWS>(
      ws, false, g,
      [&]() {
        if (wave < XB) store_row<1>(a.Gdx, row, D::DX, 16 * wave, valid, &own);            // G of combine_future_past.4
        else if (wave < GB) store_row<1>(a.dx0_acc, row, D::DX, 16 * (wave - XB), valid, &own);
#pragma unroll
        for (int b = 0; b < XB; ++b) g[b] = xb0[b * 64 + lane];
      },
      [&](int mb, v4f v) { act2 = mask(v, act2); xb1[mb * 64 + lane] = act2; });          // mb == wave: act2 now holds d H2
  B3D_STAMP(1, 4);
  linear_split<Seq, 5, false, false, NWS>(
      ws, false, d2,
      [&]() {
        if (wave < H2B) store_row<1>(a.GdH2, row, D::NH2, 16 * wave, valid, &act2);
#pragma unroll
        for (int b = 0; b < H2B; ++b) d2[b] = xb1[b * 64 + lane];
      },
      [&](int mb, v4f v) { act1 = mask(v, act1); xb0[mb * 64 + lane] = act1; });
  B3D_STAMP(1, 5);
  linear_split<Seq, 6, false, false, NWS>(
      ws, false, d1,
      [&]() {
        if (wave < H1B) store_row<1>(a.GdH1, row, D::NH1, 16 * wave, valid, &act1);
#pragma unroll
        for (int b = 0; b < H1B; ++b) d1[b] = xb0[b * 64 + lane];
      },
      [&](int mb, v4f v) { store_row<1>(a.dM, row, 2 * D::DM, 16 * mb, valid, &v); });
  B3D_STAMP(1, 6);
}

// ---- the same per-node backward as TWO launches, for stacks whose lists have different widths ------------------
// (camera+LiDAR+radar: dH1 is 256 wide, dF1 / dP1 192; the fused kernel above would need 216 KB of LDS there)
//   node_listsum_kernel : dT[n] = ( sum_{dst=n} dH1 | sum_{src=n} dH1 | sum_{dst=n} dF1 | sum_{src=n} dP1 ), one
//                         wavefront per (16-node tile, list, 64-column group); no LDS, no weights
//   node_bwd_g_kernel   : (dx | dx0) = GradProj . dT[n]  (+ the node MLP's data gradient), dT rows read back from L2
template <class D>
struct ListSumGeom {
  static constexpr int BPT = 4;                                  // blocks (64 columns) per task
  static_assert(D::EH1 % 64 == 0 && D::MH % 64 == 0, "list widths in 64-column groups");
  static constexpr int TA = D::EH1 / 64, TM = D::MH / 64;
  static constexpr int TASKS = 2 * TA + 2 * TM;                  // per 16-node tile
};
template <class D>
__global__ __launch_bounds__(256) void node_listsum_kernel(const NodeGradProjArgs a) {
  using G = ListSumGeom<D>;
  using H = Hoist<D>;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long task = (long)blockIdx.x * 4 + wave;
  const long tile = task / G::TASKS;
  int t = (int)(task - tile * G::TASKS);
  if (tile * 16 >= a.N) return;
  int list, grp;
  if (t < G::TA) { list = 0; grp = t; }
  else if (t < 2 * G::TA) { list = 1; grp = t - G::TA; }
  else if (t < 2 * G::TA + G::TM) { list = 2; grp = t - 2 * G::TA; }
  else { list = 3; grp = t - 2 * G::TA - G::TM; }
  const float* base = (list < 2) ? a.GdH1 : (list == 2 ? a.GdF1 : a.GdP1);
  const int width = (list < 2) ? D::EH1 : D::MH;
  const int tcol = (list == 0 ? H::OA : list == 1 ? H::OB : list == 2 ? H::OF : H::OP) + 64 * grp;
  const long row = tile * 16 + q_row(lane);
  const bool valid = row < a.N;
  v4f part[G::BPT];
#pragma unroll
  for (int b = 0; b < G::BPT; ++b) part[b] = v4f{0.f, 0.f, 0.f, 0.f};
  if (valid && base) {
    const bool by_dst = !(list & 1);
    const int* ptr = by_dst ? a.dst_ptr : a.src_ptr;
    segment_sum_deep_q<G::BPT, 6>(base, width, 64 * grp, by_dst ? a.dst_perm : a.src_perm, ptr[row], ptr[row + 1], part, q_piece(lane));
  }
  store_row_q<G::BPT>(a.dT, row, H::GW, tcol, valid, part);
}

struct NodeBwdGArgs {
  int N;
  const float* dT;      // [N, GW] of layer l+1 (node_listsum_kernel)
  float* gx;            // MLP == false: [N, 2 DX] = (dx | dx0 contribution)
  float* dx0_acc;       // MLP: [N, DX] running gradient of initial_x
  int dx0_first;
  const float* sH1;     // MLP: saved activations of layer l's node MLP
  const float* sH2;
  float* dM;            // MLP: [N, 2 DM]
  float* Gdx;           // MLP: [N, DX]  G tensors of layer l's node MLP
  float* GdH2;
  float* GdH1;
  const float* wpack;   // MLP: NodeBwdHSeq images; else Hoist::GradProjSeq images
};
template <class D>
struct NodeBwdGLds {
  static constexpr int PB = (D::NH1 > 2 * D::DX ? D::NH1 : 2 * D::DX) / 16;
  static constexpr int LMAX = (D::EH1 > D::MH ? D::EH1 : D::MH) / 16;          // widest dT tile of the four products (blocks)
  static constexpr int XBLOCKS = B3D_NODE_TILE_LDS && 2 * LMAX > 2 * PB ? 2 * LMAX : 2 * PB;
  static constexpr int BYTES = kLdsBytes + XBLOCKS * 64 * 16;
  static_assert(BYTES <= 160 * 1024, "LDS of one CU");
};
constexpr int kNodeBwdGWaves = 16;

template <class D, bool MLP>
__global__ __launch_bounds__(kNodeBwdGWaves * 64, 1) void node_bwd_g_kernel(const NodeBwdGArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  using H = Hoist<D>;
  using Seq = typename std::conditional<MLP, NodeBwdHSeq<D>, typename H::GradProjSeq>::type;   // the same four images first
  constexpr int NWS = kNodeBwdGWaves;
  constexpr int LA = D::EH1 / 16, LM = D::MH / 16;
  constexpr int XB = D::DX / 16, GB = 2 * XB, H1B = D::NH1 / 16, H2B = D::NH2 / 16;
  static_assert(GB <= NWS && H1B <= NWS && H2B <= NWS, "at most one output block per wavefront and layer");
  constexpr int PB = NodeBwdGLds<D>::PB;
  if constexpr (MLP) { B3D_STAMP(1, 0); B3D_ACQ_ZERO(); }
  NodeRing<NWS * 64> ws;      // (ring form:) one barrier per weight chunk: it is also what publishes the previous layer's LDS activations
  ws.init(a.wpack, smem);
  ws.template start<Seq>();
  v4f* xb0 = reinterpret_cast<v4f*>(smem + 2 * kWBufFloats);
  v4f* xb1 = xb0 + PB * 64;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 16 + (lane & 15);
  const bool valid = row < a.N;
  v4f act2 = {0.f, 0.f, 0.f, 0.f}, act1 = {0.f, 0.f, 0.f, 0.f}, prev0 = {0.f, 0.f, 0.f, 0.f};
  if constexpr (MLP) {
    if (wave < H2B) load_row<1>(a.sH2, row, D::NH2, 16 * wave, valid, &act2);
    if (wave < H1B) load_row<1>(a.sH1, row, D::NH1, 16 * wave, valid, &act1);
    if (!a.dx0_first && wave >= XB && wave < GB) load_row<1>(a.dx0_acc, row, D::DX, 16 * (wave - XB), valid, &prev0);
  }
  // (dx | dx0) = sum over the four lists of (node columns)^T . dT_list; this wavefront owns output block `wave`
  v4f own = {0.f, 0.f, 0.f, 0.f};
  v4f dt[LA > LM ? LA : LM];
  auto acc = [&](int, v4f v) { own += v; };
#if B3D_NODE_TILE_LDS
  // The operand of a product is the 16-row tile of one list's dT columns.  Every wavefront needs all of it, and a load issued in the
  // layer's hook queues behind the weight chunk the acquire has just put in flight: 16 loads per lane, ~2 us exposed per product
  // (tools/phase_stamps_clr.py, profiles/r06_e_node_phase_stamps.txt).  Round 6: the tile goes through LDS -- wavefront w fetches
  // block w (ONE 16-byte load per lane) a whole product ahead, the hook of product p publishes the tile of product p + 1 into the
  // other of two LDS buffers and takes its own from the one the previous hook filled (a weight-chunk barrier lies between a buffer's
  // write and its reads, and between its reads and the next write).  Same operands, same MFMA order: bit-identical gradients.
  constexpr int LMAX = NodeBwdGLds<D>::LMAX;
  static_assert(LMAX <= NWS, "one tile block per wavefront");
  v4f* tile0 = xb0;                       // both buffers are dead before the stages behind the products write xb0 / xb1
  v4f* tile1 = xb0 + LMAX * 64;
  v4f pre = {0.f, 0.f, 0.f, 0.f};
  auto fetch = [&](int off, int nb) { if (wave < nb) load_row<1>(a.dT, row, H::GW, off + 16 * wave, valid, &pre); };
  auto publish = [&](v4f* t, int nb) { if (wave < nb) t[wave * 64 + lane] = pre; };
  auto take = [&](const v4f* t, auto nb) {
#pragma unroll
    for (int b = 0; b < decltype(nb)::value; ++b) dt[b] = t[b * 64 + lane];
  };
  using NA = std::integral_constant<int, LA>;
  using NM = std::integral_constant<int, LM>;
  fetch(H::OA, LA);
  publish(tile0, LA);
  fetch(H::OB, LA);
  if constexpr (MLP) B3D_STAMP(1, 1);
  linear_split<Seq, 0, false, false, NWS>(ws, false, dt, [&]() { take(tile0, NA{}); publish(tile1, LA); fetch(H::OF, LM); }, acc);
  if constexpr (MLP) B3D_STAMP(1, 2);
  linear_split<Seq, 1, false, false, NWS>(ws, false, dt, [&]() { take(tile1, NA{}); publish(tile0, LM); fetch(H::OP, LM); }, acc);
  linear_split<Seq, 2, false, false, NWS>(ws, false, dt, [&]() { take(tile0, NM{}); publish(tile1, LM); }, acc);
  linear_split<Seq, 3, false, false, NWS>(ws, false, dt, [&]() { take(tile1, NM{}); }, acc);
#else
  load_row<LA>(a.dT, row, H::GW, H::OA, valid, dt);
  if constexpr (MLP) B3D_STAMP(1, 1);
  linear_split<Seq, 0, false, false, NWS>(ws, false, dt, [&]() {}, acc);
  if constexpr (MLP) B3D_STAMP(1, 2);
  linear_split<Seq, 1, false, false, NWS>(ws, false, dt, [&]() { load_row<LA>(a.dT, row, H::GW, H::OB, valid, dt); }, acc);
  linear_split<Seq, 2, false, false, NWS>(ws, false, dt, [&]() { load_row<LM>(a.dT, row, H::GW, H::OF, valid, dt); }, acc);
  linear_split<Seq, 3, false, false, NWS>(ws, false, dt, [&]() { load_row<LM>(a.dT, row, H::GW, H::OP, valid, dt); }, acc);
#endif
  if constexpr (MLP) B3D_STAMP(1, 3);
  if constexpr (!MLP) {
    if (wave < GB) store_row<1>(a.gx, row, 2 * D::DX, 16 * wave, valid, &own);
    return;
  } else {
    if (wave < XB) xb0[wave * 64 + lane] = own;                 // d x': input of the node MLP's data gradient
    else if (wave < GB) own += prev0;
    v4f g[XB], d2[H2B], d1[H1B];
    auto mask = [](v4f v, v4f act) {
      return v4f{act.x > 0.f ? v.x : 0.f, act.y > 0.f ? v.y : 0.f, act.z > 0.f ? v.z : 0.f, act.w > 0.f ? v.w : 0.f};
    };
    linear_split<Seq, 4, false, false, NWS>(
        ws, false, g,
        [&]() {
          if (wave < XB) store_row<1>(a.Gdx, row, D::DX, 16 * wave, valid, &own);            // G of combine_future_past.4
          else if (wave < GB) store_row<1>(a.dx0_acc, row, D::DX, 16 * (wave - XB), valid, &own);
#pragma unroll
          for (int b = 0; b < XB; ++b) g[b] = xb0[b * 64 + lane];
        },
        [&](int mb, v4f v) { act2 = mask(v, act2); xb1[mb * 64 + lane] = act2; });          // mb == wave: act2 now holds d H2
    B3D_STAMP(1, 4);
    linear_split<Seq, 5, false, false, NWS>(
        ws, false, d2,
        [&]() {
          if (wave < H2B) store_row<1>(a.GdH2, row, D::NH2, 16 * wave, valid, &act2);
#pragma unroll
          for (int b = 0; b < H2B; ++b) d2[b] = xb1[b * 64 + lane];
        },
        [&](int mb, v4f v) { act1 = mask(v, act1); xb0[mb * 64 + lane] = act1; });
    B3D_STAMP(1, 5);
    linear_split<Seq, 6, false, false, NWS>(
        ws, false, d1,
        [&]() {
          if (wave < H1B) store_row<1>(a.GdH1, row, D::NH1, 16 * wave, valid, &act1);
#pragma unroll
          for (int b = 0; b < H1B; ++b) d1[b] = xb0[b * 64 + lane];
        },
        [&](int mb, v4f v) { store_row<1>(a.dM, row, 2 * D::DM, 16 * mb, valid, &v); });
    B3D_STAMP(1, 6);
    B3D_ACQ_SAVE(1, 31);
  }
}

// Loader of the node-encoder backward: gradient at x_enc = upstream + running d initial_x + layer 0's (dx | dx0).
template <int XB>
struct LoadNodeEncGradH {
  static constexpr int NB = XB;
  const float* d_x_enc;   // [N, DX] or nullptr
  const float* dx0_acc;   // [N, DX] or nullptr
  const float* gx;        // [N, 2 DX]
  __device__ __forceinline__ void operator()(long row, bool valid, v4f* dst) const {
    v4f g[2 * XB];
    load_row<2 * XB>(gx, row, 32 * XB, 0, valid, g);
    v4f t[XB];
    if (d_x_enc) { load_row<XB>(d_x_enc, row, 16 * XB, 0, valid, t); add_blocks<XB>(g, t); }
    if (dx0_acc) { load_row<XB>(dx0_acc, row, 16 * XB, 0, valid, t); add_blocks<XB>(g, t); }
#pragma unroll
    for (int b = 0; b < XB; ++b) dst[b] = g[b] + g[XB + b];
  }
};

}  // namespace b3d
