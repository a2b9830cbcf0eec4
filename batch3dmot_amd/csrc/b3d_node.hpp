// Node phase of CausalMessagePassing for node-sized inputs (a few thousand rows): FOUR wavefronts
// share one 16-row tile.
//
// With one wavefront per tile (mp_node_*_kernel<D, 1>) a tile is a serial chain -- two gathered
// segment sums, then three Linear layers whose MFMAs all sit on one SIMD -- and ~190 such chains
// are all the parallelism a 3,000-node batch offers: the launch is latency bound at ~5 TFLOP/s.
// Here the chain is cut across the 4 SIMDs of a CU:
//   * segment sums: each wavefront sums a slice of the feature blocks of ONE of the two lists;
//   * every Linear: wavefront w computes the output blocks mb with mb % 4 == w (same fmaf chain
//     per output as the single-wave kernel, so results are bitwise those of mp_node_*_kernel for
//     the forward; the backward adds the two list partials in a different, fixed, order);
//   * activations travel between layers through a ping-pong LDS buffer in register ("layout L")
//     image form: block b, lane l -> float4 at [b][l]; conflict-free b128 accesses.
// Weights stream through the same two-slot LDS ring as everywhere else.
#pragma once
#include "b3d_mp.hpp"

namespace b3d {

constexpr int kNodeWaves = 4;
// Ring of the node-sized kernels: the two-slot LDS ring (WStreamT, shipped) or -- EXPERIMENT, round 6, -DB3D_NODE_DIRECT=1 -- none:
// weights read straight from global memory by the wavefront that owns the block (WDirectT, b3d_dev.hpp).  Measured (gpurun_out/
// ab_node_direct.txt, parity green): mp_node_fwd 203 -> 597 us per step, mp_node_bwd 413 -> 1,434.  The images are row-major with
// 400-1,056-byte rows, so a fragment load (lane = row m, 16 bytes) touches 16 half-used cache lines per wave-instruction where an
// LDS-DMA piece moves one contiguous KB; fragment-major images (as the edge kernels' fragment streams) would be the way, not built.
#ifndef B3D_NODE_DIRECT
#define B3D_NODE_DIRECT 0
#endif
// node_bwd_g (b3d_hoist.hpp): the dT tiles of the four gradient products through LDS, fetched a product ahead (1) or 16 loads per lane
// in the layer's hook (0, the round-5 form, kept for the A/B)
#ifndef B3D_NODE_TILE_LDS
#define B3D_NODE_TILE_LDS 1
#endif
// EXPERIMENT (round 6, off): the LDS-DMA pieces of the next weight chunk issued only by the wavefronts that have no block in the current
// one (1) instead of by all (0).  Measured (profiles/r06_e_ab_node_idle_issue.txt, phase stamps): node_bwd_g 24.5 -> 26.2 us per
// workgroup, time inside acquire 4.4 -> 8.4 us -- 12 issuers with 4.3 pieces each deliver the chunk later than 16 with 3.25, and the
// three or four busy wavefronts gain less from starting their MFMAs early than everyone loses at the next barrier.
#ifndef B3D_NODE_IDLE_ISSUE
#define B3D_NODE_IDLE_ISSUE 0
#endif
template <int NT>
using NodeRing = std::conditional_t<B3D_NODE_DIRECT != 0, WDirectT<NT>, WStreamT<NT>>;

template <class D>
struct NodeSplit {
  static constexpr int XB = D::DX / 16, DMB = D::DM / 16, H1B = D::NH1 / 16, H2B = D::NH2 / 16;
  // exchange buffer: the widest thing that crosses waves (backward: two partial gradient rows)
  static constexpr int XBUF_BLOCKS = 4 * XB > 2 * DMB ? (4 * XB > H1B ? 4 * XB : H1B) : (2 * DMB > H1B ? 2 * DMB : H1B);
  static constexpr int LDS_BYTES = kLdsBytes + 2 * XBUF_BLOCKS * 64 * 16;
};

// One Linear layer, output blocks split over the NWS wavefronts of the workgroup.
//   inload(): fills in[] -- called after the first chunk's barrier, i.e. when the previous layer's
//             LDS writes of every wavefront are visible;
//   emit(mb, v): called by the wavefront that owns output block mb.
template <class Seq, int LI, bool RELU, bool BIAS, int NWS, int CH, class WS, class InLoad, class Emit>
__device__ __forceinline__ void linear_split_chunk(WS& ws, bool more, const v4f* __restrict__ in, LinIn<Seq::kp(LI), Seq::bf(LI)>& xin,
                                                   InLoad& inload, Emit& emit) {
  constexpr int KP = Seq::kp(LI), NP = Seq::np(LI);
  constexpr bool BF = Seq::bf(LI);
  constexpr int KB = KP / 16, NB = NP / 16;
  constexpr int STRIDE = row_stride(KP, BF);
  constexpr int CR = chunk_rows(KP, NP, BF);
  constexpr int C0 = Seq::first_chunk(LI);
  constexpr int mb0 = CH * (CR / 16);
  constexpr int mbn = (mb0 + CR / 16 < NB) ? mb0 + CR / 16 : NB;
  constexpr int JMAX = (mbn - mb0 + NWS - 1) / NWS;          // owned blocks in this chunk (upper bound)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int m = lane & 15, q = lane >> 4;
  // the wavefronts that own a block of THIS chunk leave the fetch of the next one to the others (WStreamT::issue)
  constexpr int NACT = B3D_NODE_IDLE_ISSUE && (mbn - mb0) < NWS ? mbn - mb0 : 0;
  B3D_ACQ_T0();
  const float* w = ws.template acquire<Seq, C0 + CH, mb0 % NWS, NACT>(more);
  B3D_ACQ_ADD();
  if constexpr (CH == 0) { inload(); xin.prepare(in); }       // the operand (bf16 pieces or the blocks themselves): once per layer
  const int first = mb0 + ((wave - mb0 % NWS) + NWS) % NWS;   // first owned block of the chunk
  const v4f zero4 = {0.f, 0.f, 0.f, 0.f};
  auto emit_one = [&](int mb, v4f v, int slot) {
    // emit(mb, v) or emit(mb, v, slot): slot = mb / NWS, the index of the block among those this
    // wavefront owns -- a compile-time constant after unrolling (for per-wave register arrays)
    if constexpr (requires { emit(0, zero4, 0); }) {
      static_assert(mb0 % NWS == 0, "slot numbering needs chunk boundaries on multiples of the wave count");
      emit(mb, v, slot);
    } else {
      emit(mb, v);
    }
  };
  if constexpr (WS::kDirect) {
    // weights straight from global memory (WDirectT, b3d_dev.hpp): this wavefront's blocks only, fragments two groups ahead of the
    // MFMAs that consume them; one accumulator chain per block, the same fmaf / bf16x6 order as the LDS form
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
      const int mb = first + NWS * j;
      if (mb >= mbn) break;                                    // wave-uniform
      const float* wa = w + ((mb - mb0) * 16 + m) * STRIDE + 4 * q;
      v4f acc = zero4;
      if constexpr (BIAS) {
        const gbl_f_cp ba = (gbl_f_cp)(w + ((mb - mb0) * 16 + 4 * q) * STRIDE + bias_col(KP, BF));
        acc = v4f{ba[0], ba[STRIDE], ba[2 * STRIDE], ba[3 * STRIDE]};
      }
      if constexpr (BF) {
        constexpr int KG = KP / 32, G = 2, NG = (KG + G - 1) / G;          // groups of two 32-wide k groups: six 16-byte loads
        typedef const __attribute__((address_space(1))) u4v* gbl_u4v_cp;
        auto ld = [&](int c) {
          Bf3 f;
          f.p0 = __builtin_bit_cast(bf8, *(gbl_u4v_cp)(wa + 16 * c));
          f.p1 = __builtin_bit_cast(bf8, *(gbl_u4v_cp)(wa + 16 * c + KP / 2));
          f.p2 = __builtin_bit_cast(bf8, *(gbl_u4v_cp)(wa + 16 * c + KP));
          return f;
        };
        Bf3 f[2][G];
#pragma unroll
        for (int g = 0; g < 2 && g < NG; ++g)
#pragma unroll
          for (int i = 0; i < G; ++i)
            if (g * G + i < KG) f[g][i] = ld(g * G + i);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
#pragma unroll
          for (int i = 0; i < G; ++i)
            if (g * G + i < KG) acc = bf_mfma6(f[g & 1][i], xin.x[g * G + i], acc);
          if (g + 2 < NG) {
#pragma unroll
            for (int i = 0; i < G; ++i)
              if ((g + 2) * G + i < KG) f[g & 1][i] = ld((g + 2) * G + i);
          }
        }
      } else {
        constexpr int G = 4, NG = (KB + G - 1) / G;                        // groups of four 16-wide k blocks: four 16-byte loads
        v4f f[2][G];
#pragma unroll
        for (int g = 0; g < 2 && g < NG; ++g)
#pragma unroll
          for (int i = 0; i < G; ++i)
            if (g * G + i < KB) f[g][i] = *(gbl_v4f_cp)(wa + 16 * (g * G + i));
#pragma unroll
        for (int g = 0; g < NG; ++g) {
#pragma unroll
          for (int i = 0; i < G; ++i)
            if (g * G + i < KB) acc = mfma4(f[g & 1][i], in[g * G + i], acc);
          if (g + 2 < NG) {
#pragma unroll
            for (int i = 0; i < G; ++i)
              if ((g + 2) * G + i < KB) f[g & 1][i] = *(gbl_v4f_cp)(wa + 16 * ((g + 2) * G + i));
          }
        }
      }
      emit_one(mb, RELU ? relu4(acc) : acc, mb0 / NWS + j);
    }
  } else if constexpr (BF) {
    constexpr int KG = KP / 32;
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
      const int mb = first + NWS * j;
      if (mb >= mbn) break;                                    // wave-uniform
      const float* wa = w + ((mb - mb0) * 16 + m) * STRIDE + 4 * q;
      v4f acc = zero4;
      if constexpr (BIAS) {
        const float* ba = w + ((mb - mb0) * 16 + 4 * q) * STRIDE + bias_col(KP, BF);
        acc = v4f{ba[0], ba[STRIDE], ba[2 * STRIDE], ba[3 * STRIDE]};
      }
      Bf3 cur = bf_load<KP>(wa);
#pragma unroll
      for (int c = 0; c < KG; ++c) {
        Bf3 nxt = cur;
        if (c + 1 < KG) nxt = bf_load<KP>(wa + 16 * (c + 1));
        __builtin_amdgcn_sched_barrier(0);
        acc = bf_mfma6(cur, xin.x[c], acc);
        cur = nxt;
      }
      emit_one(mb, RELU ? relu4(acc) : acc, mb0 / NWS + j);
    }
  } else {
#pragma unroll
  for (int j = 0; j < JMAX; j += 2) {
    const int mbA = first + NWS * j, mbB = mbA + NWS;
    const bool hasA = mbA < mbn, hasB = (j + 1 < JMAX) && (mbB < mbn);     // wave-uniform
    if (!hasA) break;
    const float* wa = w + ((mbA - mb0) * 16 + m) * STRIDE + 4 * q;
    const float* wb = w + (((hasB ? mbB : mbA) - mb0) * 16 + m) * STRIDE + 4 * q;
    v4f acc0 = zero4, acc1 = zero4;
    if constexpr (BIAS) {
      const float* ba = w + ((mbA - mb0) * 16 + 4 * q) * STRIDE + KP;
      const float* bb = w + (((hasB ? mbB : mbA) - mb0) * 16 + 4 * q) * STRIDE + KP;
      acc0 = v4f{ba[0], ba[STRIDE], ba[2 * STRIDE], ba[3 * STRIDE]};
      acc1 = v4f{bb[0], bb[STRIDE], bb[2 * STRIDE], bb[3 * STRIDE]};
    }
    v4f fa = *reinterpret_cast<const v4f*>(wa), fb = *reinterpret_cast<const v4f*>(wb);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      v4f na = zero4, nb = zero4;
      if (kb + 1 < KB) {
        na = *reinterpret_cast<const v4f*>(wa + 16 * (kb + 1));
        nb = *reinterpret_cast<const v4f*>(wb + 16 * (kb + 1));
      }
      __builtin_amdgcn_sched_barrier(0);
      acc0 = mfma4(fa, in[kb], acc0);
      if (hasB) acc1 = mfma4(fb, in[kb], acc1);
      fa = na; fb = nb;
    }
    emit_one(mbA, RELU ? relu4(acc0) : acc0, mb0 / NWS + j);
    if (hasB) emit_one(mbB, RELU ? relu4(acc1) : acc1, mb0 / NWS + j + 1);
  }
  }
}

template <class Seq, int LI, bool RELU, bool BIAS, int NWS, class WS, class InLoad, class Emit, int... CH>
__device__ __forceinline__ void linear_split_impl(WS& ws, bool more, const v4f* __restrict__ in, InLoad& inload, Emit& emit,
                                                  std::integer_sequence<int, CH...>) {
  LinIn<Seq::kp(LI), Seq::bf(LI)> xin;
  if constexpr (!Seq::bf(LI)) xin.prepare(in);
  (linear_split_chunk<Seq, LI, RELU, BIAS, NWS, CH>(ws, more, in, xin, inload, emit), ...);
}

template <class Seq, int LI, bool RELU, bool BIAS, int NWS, class WS, class InLoad, class Emit>
__device__ __forceinline__ void linear_split(WS& ws, bool more, const v4f* __restrict__ in, InLoad inload, Emit emit) {
  linear_split_impl<Seq, LI, RELU, BIAS, NWS>(ws, more, in, inload, emit,
                                              std::make_integer_sequence<int, Seq::layer_chunks(LI)>{});
}

// ------------------------------------------------------------------------------------------
// NWS == 4: two wavefronts per list.  NWS == 8 (kNodeWavesWide): two for the by-destination list (`past`: the
// in-degree is a handful of frames) and six for the by-source list (`fut`), three per feature half, each a
// contiguous third of the list -- a launch lasts as long as its worst tile, and the out-degree of a tracking
// graph reaches 40+.
constexpr int kNodeWavesWide = 8;
template <class D, class Seq, bool PROJ, int NWS = kNodeWaves>
__device__ __forceinline__ void node_fwd_split_body(const NodeFwdArgs& a, float* smem) {
  using NS = NodeSplit<D>;
  static_assert(NWS == kNodeWaves || NWS == kNodeWavesWide, "four or eight wavefronts per tile");
  constexpr int XB = NS::XB, DMB = NS::DMB, H1B = NS::H1B, H2B = NS::H2B;
  constexpr int FP = NWS == kNodeWavesWide ? 3 : 1;          // wavefronts per half of the by-source list
  constexpr int MB2 = 2 * DMB, BPW = DMB / 2;                // blocks of M per gathering wavefront
  static_assert(DMB % 2 == 0 && 2 + 2 * FP == NWS, "message width must split over the wavefronts");
  static_assert(FP == 1 || (FP - 1) * DMB <= NS::XBUF_BLOCKS, "partial sums live in the second exchange buffer");
  if constexpr (PROJ) { B3D_STAMP(0, 0); B3D_ACQ_ZERO(); }
  NodeRing<NWS * 64> ws;      // (ring form:) one barrier per weight chunk: it is also what publishes the previous layer's LDS activations
  ws.init(a.wpack, smem);
  ws.template start<Seq>();
  v4f* xb0 = reinterpret_cast<v4f*>(smem + 2 * kWBufFloats);
  v4f* xb1 = xb0 + NS::XBUF_BLOCKS * 64;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 16 + (lane & 15);
  const bool valid = row < a.N;

  // Every weight-chunk acquire drains vmcnt: stores are issued right AFTER the next layer's barrier (by the
  // block's owner, from the values every wavefront reads back), loads a stage ahead of their use.
  // PROJ: the x0 terms of the table blocks this wavefront will own (block mb = wave + NWS * slot, columns F | P only)
  constexpr int PFB = 2 * D::EH1 / 16, PT0B = 2 * D::MH / 16;
  constexpr int PS0 = PFB / NWS, PS1 = (PFB + PT0B - 1) / NWS;       // first and last slot that can hold such a block
  v4f t0[PROJ ? PS1 - PS0 + 1 : 1];
  if constexpr (PROJ) {
#pragma unroll
    for (int s = PS0; s <= PS1; ++s) {
      const int mb = wave + NWS * s;                                   // wave-uniform
      t0[s - PS0] = v4f{0.f, 0.f, 0.f, 0.f};
      if (mb >= PFB && mb < PFB + PT0B) load_row<1>(a.T0, row, 16 * PT0B, 16 * (mb - PFB), valid, &t0[s - PS0]);
    }
  }
  // ---- segment sums: waves [0, NWS/2) own the `past` half of M, the others the `fut` half ----
  // The gathers run in layout Q (lane = 4 row + q: contiguous 64-byte reads per lane quad, b3d_dev.hpp); the
  // sums reach the MFMA layout through their LDS slot.
  v4f part[BPW];
  const bool is_fut = wave >= 2;
  const int fpart = is_fut ? (wave - 2) >> 1 : 0;            // which third of the by-source list
  const int blk0 = is_fut ? DMB + ((wave - 2) & 1) * BPW : wave * BPW;
  const bool owner = fpart == 0;                             // holds the finished sums of its blocks
  const long rowq = (long)blockIdx.x * 16 + q_row(lane);
  const bool validq = rowq < a.N;
  {
#pragma unroll
    for (int b = 0; b < BPW; ++b) part[b] = v4f{0.f, 0.f, 0.f, 0.f};
    if (validq) {
      constexpr int U = BPW <= 3 ? 8 : 4;                    // rows in flight per lane (MI355X: 8 beats 4 and 16)
      if (!is_fut) {
        segment_sum_deep_q<BPW, U>(a.past, D::DM, 16 * blk0, a.dst_perm, a.dst_ptr[rowq], a.dst_ptr[rowq + 1], part, q_piece(lane));
      } else {
        const int beg = a.src_ptr[rowq], len = a.src_ptr[rowq + 1] - beg;
        segment_sum_deep_q<BPW, U>(a.fut, D::DM, 16 * (blk0 - DMB), a.src_perm, beg + len * fpart / FP, beg + len * (fpart + 1) / FP, part,
                                   q_piece(lane));
      }
    }
    v4f* dstp = owner ? xb0 + blk0 * 64 : xb1 + ((fpart - 1) * DMB + blk0 - DMB) * 64;
#pragma unroll
    for (int b = 0; b < BPW; ++b) dstp[b * 64 + q_slot(lane)] = part[b];
    if constexpr (FP > 1) {
      __syncthreads();
      if (is_fut && owner) {
#pragma unroll
        for (int b = 0; b < BPW; ++b) {
#pragma unroll
          for (int t = 1; t < FP; ++t) part[b] += xb1[((t - 1) * DMB + blk0 - DMB + b) * 64 + q_slot(lane)];
          xb0[(blk0 + b) * 64 + q_slot(lane)] = part[b];
        }
      }
    }
  }
  if constexpr (PROJ) B3D_STAMP(0, 1);
  v4f m[MB2], h1[H1B], h2[H2B];
  linear_split<Seq, 0, true, true, NWS>(
      ws, false, m,
      [&]() {
        if (a.M && owner) store_row_q<BPW>(a.M, rowq, 2 * D::DM, 16 * blk0, validq, part);
#pragma unroll
        for (int b = 0; b < MB2; ++b) m[b] = xb0[b * 64 + lane];
      },
      [&](int mb, v4f v) { xb1[mb * 64 + lane] = v; });
  if constexpr (PROJ) B3D_STAMP(0, 2);
  linear_split<Seq, 1, true, true, NWS>(
      ws, false, h1,
      [&]() {
#pragma unroll
        for (int b = 0; b < H1B; ++b) h1[b] = xb1[b * 64 + lane];
        if (a.sH1) {
#pragma unroll
          for (int b = 0; b < H1B; ++b)
            if (b % NWS == wave) store_row<1>(a.sH1, row, D::NH1, 16 * b, valid, &h1[b]);
        }
      },
      [&](int mb, v4f v) { xb0[mb * 64 + lane] = v; });
  if constexpr (PROJ) B3D_STAMP(0, 3);
  linear_split<Seq, 2, false, true, NWS>(
      ws, false, h2,
      [&]() {
#pragma unroll
        for (int b = 0; b < H2B; ++b) h2[b] = xb0[b * 64 + lane];
        if (a.sH2) {
#pragma unroll
          for (int b = 0; b < H2B; ++b)
            if (b % NWS == wave) store_row<1>(a.sH2, row, D::NH2, 16 * b, valid, &h2[b]);
        }
      },
      [&](int mb, v4f v) {
        if constexpr (PROJ) xb1[mb * 64 + lane] = v;                  // stored behind the projection's barrier
        else store_row<1>(a.x_out, row, D::DX, 16 * mb, valid, &v);
      });
  if constexpr (PROJ) B3D_STAMP(0, 4);
  if constexpr (PROJ) {
    // per-node parts of the NEXT layer's three first Linear layers (b3d_hoist.hpp): T = Wp x' + bp (+ x0 terms)
    // columns: first-layer parts (A | B | F | P), then GATConv.lin(x) of the discarded k-NN block; only F | P
    // carry x0 terms
    constexpr int FB = PFB, T0B = PT0B, TB = FB + T0B + D::DX / 16, TW = 16 * TB;
    v4f xn[XB];
    linear_split<Seq, 3, false, true, NWS>(
        ws, false, xn,
        [&]() {
#pragma unroll
          for (int b = 0; b < XB; ++b) xn[b] = xb1[b * 64 + lane];
#pragma unroll
          for (int b = 0; b < XB; ++b)
            if (b % NWS == wave) store_row<1>(a.x_out, row, D::DX, 16 * b, valid, &xn[b]);
        },
        [&](int mb, v4f v, int slot) {
          if (mb >= FB && mb < FB + T0B) v += t0[slot - PS0];
          store_row<1>(a.T, row, TW, 16 * mb, valid, &v);
        });
  }
  if constexpr (PROJ) { B3D_STAMP(0, 5); B3D_ACQ_SAVE(0, 31); }
  (void)XB;
}

template <class D, int NWS = kNodeWaves>
__global__ __launch_bounds__(NWS * 64, 1) void mp_node_fwd_split_kernel(const NodeFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  node_fwd_split_body<D, typename D::NodeFwdSeq, false, NWS>(a, smem);
}

// ------------------------------------------------------------------------------------------
template <class D>
__global__ __launch_bounds__(kNodeWaves * 64, 1) void mp_node_bwd_split_kernel(const NodeBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  using Seq = typename D::NodeBwdSeq;
  using NS = NodeSplit<D>;
  constexpr int NWS = kNodeWaves;
  constexpr int XB = NS::XB, DMB = NS::DMB, H1B = NS::H1B, H2B = NS::H2B;
  constexpr int GB = 2 * XB;                                  // d x' | d x0 contribution
  constexpr int GPW = GB / (NWS / 2);                         // gradient blocks per wavefront (one list each half)
  static_assert(NWS % 2 == 0 && GB % (NWS / 2) == 0, "gradient width must split over half the wavefronts");
  NodeRing<NWS * 64> ws;      // (ring form:) one barrier per weight chunk: it is also what publishes the previous layer's LDS activations
  ws.init(a.wpack, smem);
  ws.template start<Seq>();
  v4f* xb0 = reinterpret_cast<v4f*>(smem + 2 * kWBufFloats);
  v4f* xb1 = xb0 + NS::XBUF_BLOCKS * 64;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 16 + (lane & 15);
  const bool valid = row < a.N;

  // saved activations of this layer (whole rows: every wavefront masks the full gradient it reads)
  v4f act2[H2B], act1[H1B];
  load_row<H2B>(a.sH2, row, D::NH2, 0, valid, act2);
  load_row<H1B>(a.sH1, row, D::NH1, 0, valid, act1);

  v4f g[GB];
  if (a.g_direct && !a.g_direct_wide) {
    // standalone layer: d x' is given per node
    load_row<XB>(a.g_direct, row, D::DX, 0, valid, g);
#pragma unroll
    for (int b = XB; b < GB; ++b) g[b] = v4f{0.f, 0.f, 0.f, 0.f};
  } else {
    if (a.g_direct) {
      load_row<GB>(a.g_direct, row, 2 * D::DX, 0, valid, g);     // (d x' | d x0 contribution) per node
    } else {
    // ---- transposed gathers: waves [0, NWS/2) sum the by-destination list, the others by-source ----
    {
      v4f part[GPW];
#pragma unroll
      for (int b = 0; b < GPW; ++b) part[b] = v4f{0.f, 0.f, 0.f, 0.f};
      const int half = wave / (NWS / 2), blk0 = (wave % (NWS / 2)) * GPW;
      const long rowq = (long)blockIdx.x * 16 + q_row(lane);       // gathers in layout Q (b3d_dev.hpp)
      if (rowq < a.N) {
        constexpr int U = GPW <= 3 ? 8 : 4;
        if (half == 0) segment_sum_deep_q<GPW, U>(a.gdst, 2 * D::DX, 16 * blk0, a.dst_perm, a.dst_ptr[rowq], a.dst_ptr[rowq + 1], part, q_piece(lane));
        else segment_sum_deep_q<GPW, U>(a.gsrc, 2 * D::DX, 16 * blk0, a.src_perm, a.src_ptr[rowq], a.src_ptr[rowq + 1], part, q_piece(lane));
      }
#pragma unroll
      for (int b = 0; b < GPW; ++b) xb0[(half * GB + blk0 + b) * 64 + q_slot(lane)] = part[b];
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < GB; ++b) g[b] = xb0[b * 64 + lane] + xb0[(GB + b) * 64 + lane];
    }
#pragma unroll
    for (int b = 0; b < GB; ++b) {
      if (b % NWS != wave) continue;                            // wave-uniform: one owner per block
      if (b < XB) {
        store_row<1>(a.Gdx, row, D::DX, 16 * b, valid, &g[b]);
      } else {
        v4f t = g[b];
        if (!a.dx0_first) {
          v4f prev;
          load_row<1>(a.dx0_acc, row, D::DX, 16 * (b - XB), valid, &prev);
          t += prev;
        }
        store_row<1>(a.dx0_acc, row, D::DX, 16 * (b - XB), valid, &t);
      }
    }
  }

  v4f d2[H2B], d1[H1B];
  linear_split<Seq, 0, false, false, NWS>(
      ws, false, g, [&]() {}, [&](int mb, v4f v) { xb1[mb * 64 + lane] = v; });
  linear_split<Seq, 1, false, false, NWS>(
      ws, false, d2,
      [&]() {
#pragma unroll
        for (int b = 0; b < H2B; ++b) d2[b] = xb1[b * 64 + lane];
        relu_bwd<H2B>(d2, act2);
#pragma unroll
        for (int b = 0; b < H2B; ++b)
          if (b % NWS == wave) store_row<1>(a.GdH2, row, D::NH2, 16 * b, valid, &d2[b]);
      },
      [&](int mb, v4f v) { xb0[mb * 64 + lane] = v; });
  linear_split<Seq, 2, false, false, NWS>(
      ws, false, d1,
      [&]() {
#pragma unroll
        for (int b = 0; b < H1B; ++b) d1[b] = xb0[b * 64 + lane];
        relu_bwd<H1B>(d1, act1);
#pragma unroll
        for (int b = 0; b < H1B; ++b)
          if (b % NWS == wave) store_row<1>(a.GdH1, row, D::NH1, 16 * b, valid, &d1[b]);
      },
      [&](int mb, v4f v) { store_row<1>(a.dM, row, 2 * D::DM, 16 * mb, valid, &v); });
  (void)DMB;
}

}  // namespace b3d
