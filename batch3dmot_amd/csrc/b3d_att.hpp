// att_edge_encoder.0 with its node columns hoisted (clr_att_gnn.py:161-164).
//
// The first layer of att_edge_encoder is a Linear over cat(s[dst] | s[src] | e) -- 576 of its 640 input columns are
// node rows gathered per edge.  W (u | v | w) = W_u u + W_v v + W_w w, so they are evaluated once per NODE:
//     U[n] = ( W0[:, 0:288] s[n] + b0 | W0[:, 288:576] s[n] )                      [N, 1024]   (att_node_linear_kernel)
//     A0[k] = relu( U[dst_k][0:512] + U[src_k][512:1024] + W0[:, 576:640] e0_k )    [E, 512]    (att0_fwd_kernel)
// 663,552 -> 368,640 MAC per edge for the whole encoder.  The data gradient mirrors it: the per-edge gradient of the
// pre-activation is summed per node over the CSR / CSC lists (att_listsum_kernel), one per-node product gives d s,
// one narrow per-edge product d e0; the weight gradient of the node columns contracts over nodes.
#pragma once
#include "b3d_node.hpp"

namespace b3d {

// ---- per-node Linear with a wide input OR a wide output: out[n] = sum over K-slices of W_slice . in[n, slice] (+ b) -----
// 16 rows per workgroup, 8 wavefronts; the output blocks are dealt to the wavefronts (block mb -> wavefront mb % 8) and stay in
// their registers across the K-slices, so no partial ever leaves the CU.  Seq = NS images L<KS, NOUT> (fp32 format).
struct NodeLinArgs {
  int N;
  const float* in;      // [N, in_stride], slices at columns in_col0 + KS * k
  int in_stride, in_col0;
  float* out;           // [N, out_stride] at column out_col0
  int out_stride, out_col0;
  const float* wpack;
};
constexpr int kNodeLinWaves = 8;
// the ring + (B3D_NODE_TILE_LDS, b3d_node.hpp) two buffers for the operand tile of a K-slice (at most 8 blocks of 1 KB each)
constexpr int kNodeLinLds = kLdsBytes + (B3D_NODE_TILE_LDS ? 2 * kNodeLinWaves * 64 * 16 : 0);

#if B3D_NODE_TILE_LDS
// The operand of K-slice k is a 16-row x 16 KB-column tile that EVERY wavefront needs.  Loaded in the slice's hook it was KB loads per
// lane queued behind the weight chunk just put in flight -- exposed once per slice, sixteen times in the transposed launch.  As in
// node_bwd_g (b3d_hoist.hpp): wavefront w fetches block w two slices ahead, the hook of slice k publishes the tile of slice k + 1 into
// the other LDS buffer and takes its own (one weight-chunk barrier between a buffer's write and its reads, and between its reads and
// the next write).  Same operands and MFMA order: bit-identical results.
template <class Seq, int K, int LAST, int SLOTS, int KB>
__device__ __forceinline__ void node_lin_slices(NodeRing<kNodeLinWaves * 64>& ws, const NodeLinArgs& a, long row, bool valid, v4f* acc,
                                                v4f* tiles, v4f& pre) {
  static_assert(KB <= kNodeLinWaves, "one tile block per wavefront");
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  v4f in[KB];
  linear_split<Seq, K, false, K == 0, kNodeLinWaves>(
      ws, false, in,
      [&]() {
        const v4f* mine = tiles + (K & 1) * kNodeLinWaves * 64;
#pragma unroll
        for (int b = 0; b < KB; ++b) in[b] = mine[b * 64 + lane];
        if constexpr (K + 1 <= LAST) {
          if (wave < KB) tiles[((K + 1) & 1) * kNodeLinWaves * 64 + wave * 64 + lane] = pre;
        }
        if constexpr (K + 2 <= LAST) {
          if (wave < KB) load_row<1>(a.in, row, a.in_stride, a.in_col0 + 16 * KB * (K + 2) + 16 * wave, valid, &pre);
        }
      },
      [&](int mb, v4f v) {
        // block mb belongs to wavefront mb % 8 and is its (mb / 8)-th: a compile-time slot only where a chunk starts on a multiple of
        // eight blocks (the 128-row chunks of the 64- and 96-wide slices); the 96-row chunks of a 128-wide slice do not, so the slot is
        // picked by a wave-uniform compare
        const int s = mb / kNodeLinWaves;
#pragma unroll
        for (int t = 0; t < SLOTS; ++t)
          if (s == t) { if (K == 0) acc[t] = v; else acc[t] += v; }
      });
  if constexpr (K < LAST) node_lin_slices<Seq, K + 1, LAST, SLOTS, KB>(ws, a, row, valid, acc, tiles, pre);
}
#else
template <class Seq, int K, int LAST, int SLOTS, int KB>
__device__ __forceinline__ void node_lin_slices(NodeRing<kNodeLinWaves * 64>& ws, const NodeLinArgs& a, long row, bool valid, v4f* acc) {
  v4f in[KB];
  linear_split<Seq, K, false, K == 0, kNodeLinWaves>(
      ws, false, in,
      [&]() { load_row<KB>(a.in, row, a.in_stride, a.in_col0 + 16 * KB * K, valid, in); },
      [&](int, v4f v, int slot) { if (K == 0) acc[slot] = v; else acc[slot] += v; });
  if constexpr (K < LAST) node_lin_slices<Seq, K + 1, LAST, SLOTS, KB>(ws, a, row, valid, acc);
}
#endif

template <class Seq>
__global__ __launch_bounds__(kNodeLinWaves * 64, 1) void att_node_linear_kernel(const NodeLinArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NS = Seq::NL, KB = Seq::kp(0) / 16, NB = Seq::np(0) / 16;
  constexpr int SLOTS = (NB + kNodeLinWaves - 1) / kNodeLinWaves;
  NodeRing<kNodeLinWaves * 64> ws;       // (b3d_node.hpp: the LDS ring, or -- experiment -- weights straight from global memory)
  ws.init(a.wpack, smem);
  ws.template start<Seq>();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 16 + (lane & 15);
  const bool valid = row < a.N;
  v4f acc[SLOTS];
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) acc[s] = v4f{0.f, 0.f, 0.f, 0.f};
#if B3D_NODE_TILE_LDS
  v4f* tiles = reinterpret_cast<v4f*>(smem + 2 * kWBufFloats);
  v4f pre = {0.f, 0.f, 0.f, 0.f};
  if (wave < KB) {
    load_row<1>(a.in, row, a.in_stride, a.in_col0 + 16 * wave, valid, &pre);
    tiles[wave * 64 + lane] = pre;                              // tile of slice 0: visible behind the first chunk's barrier
    if constexpr (NS > 1) load_row<1>(a.in, row, a.in_stride, a.in_col0 + 16 * KB + 16 * wave, valid, &pre);
  }
  node_lin_slices<Seq, 0, NS - 1, SLOTS, KB>(ws, a, row, valid, acc, tiles, pre);
#else
  node_lin_slices<Seq, 0, NS - 1, SLOTS, KB>(ws, a, row, valid, acc);
#endif
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int mb = wave + kNodeLinWaves * s;
    if (mb < NB) store_row<1>(a.out, row, a.out_stride, a.out_col0 + 16 * mb, valid, &acc[s]);
  }
}

// ---- att_edge_encoder.0 on the edges: gathered per-node parts + the 64 edge columns ---------------------------------------
// The 512 outputs are produced as four 128-wide "layers" over the same 64-wide input, each starting from the gathered rows
// U[dst][128 c ..] + U[src][512 + 128 c ..], which are fetched one layer ahead.
struct Att0FwdArgs {
  int E;
  const int* src;
  const int* dst;
  const float* U;       // [N, 1024]
  const float* e0;      // [E, 64]
  float* A0;            // [E, 512] = relu(pre-activation)
  const float* wpack;   // Att0Seq images
  unsigned* rmask;      // [E, 16] words or nullptr: ReLU mask of A0 (b3d_dev.hpp; word c = the 128 outputs of step c)
};
using Att0Seq = LayerSeq<L<64, 128>, L<64, 128>, L<64, 128>, L<64, 128>>;     // W0[128 c .. 128 c + 128, 576:640]

template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void att0_fwd_kernel(const Att0FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  using Seq = Att0Seq;
  WStreamG<NW * 64, Seq> ws;
  ws.init(a.wpack, smem);
  ws.template start<Seq>();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ntiles = (a.E + NW * 16 - 1) / (NW * 16);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const bool more = tile + (int)gridDim.x < ntiles;
    const long row = (long)tile * (NW * 16) + wave * kRowsPerWave + (lane & 15);
    const bool valid = row < a.E;
    const long rc = valid ? row : (long)a.E - 1;
    const int s = a.src[rc], d = a.dst[rc];
    v4f ein[4], cur[8], ui[8], uj[8];
    load_row_u<4>(a.e0, rc, 64, 0, ein);
    load_row_u<8>(a.U, d, 1024, 0, cur);
    load_row_u<8>(a.U, s, 1024, 512, uj);
    wait_for(ein); wait_for(cur); wait_for(uj);
    add_blocks<8>(cur, uj);
    auto step = [&](auto tag) {
      constexpr int C = decltype(tag)::value;
      linear_init<Seq, C, true, false>(ws, more, ein, cur, cur, [&]() {
        if constexpr (C < 3) {                        // the next 128 outputs' gathered parts, a layer ahead
          load_row_u<8>(a.U, d, 1024, 128 * (C + 1), ui);
          load_row_u<8>(a.U, s, 1024, 512 + 128 * (C + 1), uj);
        }
      });
      store_row<8>(a.A0, row, 512, 128 * C, valid, cur);
      if (a.rmask) {
        unsigned w1[1];
        relu_mask_words<8>(cur, w1);
        if (valid) a.rmask[((size_t)row * 4 + (lane >> 4)) * 4 + C] = w1[0];
      }
      if constexpr (C < 3) {
        wait_for(ui); wait_for(uj);
#pragma unroll
        for (int b = 0; b < 8; ++b) cur[b] = ui[b] + uj[b];
      }
    };
    step(std::integral_constant<int, 0>{});
    step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{});
    step(std::integral_constant<int, 3>{});
  }
}

// ---- per-node sums of a per-edge gradient over the CSR (by destination) and CSC (by source) lists -------------------------
// dU[n] = ( sum_{dst = n} G | sum_{src = n} G ), G [E, W]; one wavefront per (16-node tile, list, 64-column group).
struct AttListSumArgs {
  int N, W;
  const int *dst_ptr, *dst_perm, *src_ptr, *src_perm;
  const float* G;       // [E, W]
  float* dU;            // [N, 2 W]
};
__global__ __launch_bounds__(256) void att_listsum_kernel(const AttListSumArgs a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int groups = a.W / 64;
  const long task = (long)blockIdx.x * 4 + wave;
  const long tile = task / (2 * groups);
  const int t = (int)(task - tile * 2 * groups);
  if (tile * 16 >= a.N) return;
  const int list = t / groups, grp = t - list * groups;
  const long row = tile * 16 + q_row(lane);
  const bool valid = row < a.N;
  v4f part[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) part[b] = v4f{0.f, 0.f, 0.f, 0.f};
  if (valid) {
    const int* ptr = list == 0 ? a.dst_ptr : a.src_ptr;
    segment_sum_deep_q<4, 6>(a.G, a.W, 64 * grp, list == 0 ? a.dst_perm : a.src_perm, ptr[row], ptr[row + 1], part, q_piece(lane));
  }
  store_row_q<4>(a.dU, row, 2 * a.W, list * a.W + 64 * grp, valid, part);
}

}  // namespace b3d
