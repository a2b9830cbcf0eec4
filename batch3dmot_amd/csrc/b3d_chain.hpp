// Plain row-wise MLP stacks (edge / node encoders, edge classifier, modality heads, the
// cross-edge attention encoder): the same register-resident MFMA chain as the message-passing
// kernels, parameterised by how a row's input blocks are fetched and how its result is written.
#pragma once
#include "b3d_dev.hpp"

namespace b3d {

constexpr int kChainMaxLayers = 6;

// ---- row loaders: fill NB feature blocks (layout L) for one row -------------------------------
template <int NB_>
struct LoadAligned {                 // [rows, stride] fp32, optional row gather
  static constexpr int NB = NB_;
  const float* ptr; const int* idx; int stride; int col0;
  __device__ __forceinline__ void operator()(long row, bool valid, v4f* dst) const {
    long r = row;
    if (idx && valid) r = idx[row];
    load_row<NB>(ptr, r, stride, col0, valid, dst);
  }
};

struct LoadEdgeAttrF64 {             // edge_attr [E,4] float64 -> .float() (pose_gnn.py:67)
  static constexpr int NB = 1;
  const double* ptr;
  __device__ __forceinline__ void operator()(long row, bool valid, v4f* dst) const {
    const int q = (threadIdx.x & 63) >> 4;
    v4f v = {0.f, 0.f, 0.f, 0.f};
    if (valid && q == 0) {
      const double* p = ptr + row * 4;
      v.x = (float)p[0]; v.y = (float)p[1]; v.z = (float)p[2]; v.w = (float)p[3];
    }
    dst[0] = v;
  }
};

template <int W>                     // [rows, W] fp32 with W not a multiple of 4 (pose_feats, W = 19)
struct LoadUnaligned {
  static constexpr int NB = (W + 15) / 16;
  const float* ptr;
  __device__ __forceinline__ void operator()(long row, bool valid, v4f* dst) const {
    const int q = (threadIdx.x & 63) >> 4;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      float* v = reinterpret_cast<float*>(&dst[b]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = 16 * b + 4 * q + e;
        v[e] = (valid && c < W) ? ptr[row * W + c] : 0.f;
      }
    }
  }
};

struct LoadScalar {                  // [rows, 1] -> feature 0
  static constexpr int NB = 1;
  const float* ptr;                  // nullptr = zero
  __device__ __forceinline__ void operator()(long row, bool valid, v4f* dst) const {
    const int q = (threadIdx.x & 63) >> 4;
    v4f v = {0.f, 0.f, 0.f, 0.f};
    if (valid && q == 0 && ptr) v.x = ptr[row];
    dst[0] = v;
  }
};

// Gradient arriving at the node encoder output (x = initial_x = x_enc):
//   upstream d x_enc + running d initial_x + the layer-0 edge backward's per-edge rows
//   (d x | d x0 at dst, d x | d x0 at src) summed over the node's CSR / CSC lists.
template <int XB>
struct LoadNodeEncGrad {
  static constexpr int NB = XB;
  const float* d_x_enc;   // [N, DX] or nullptr
  const float* dx0_acc;   // [N, DX] or nullptr
  const float* gdst; const float* gsrc;   // [E, 2 DX]
  const int* dst_ptr; const int* dst_perm; const int* src_ptr; const int* src_perm;
  __device__ __forceinline__ void operator()(long row, bool valid, v4f* dst) const {
    v4f g[2 * XB];
#pragma unroll
    for (int b = 0; b < 2 * XB; ++b) g[b] = v4f{0.f, 0.f, 0.f, 0.f};
    if (valid) {
      constexpr int U = XB <= 3 ? 4 : 2;          // rows in flight per lane
      segment_sum_deep<2 * XB, U>(gdst, 32 * XB, 0, dst_perm, dst_ptr[row], dst_ptr[row + 1], g);
      segment_sum_deep<2 * XB, U>(gsrc, 32 * XB, 0, src_perm, src_ptr[row], src_ptr[row + 1], g);
    }
    v4f t[XB];
    if (d_x_enc) { load_row<XB>(d_x_enc, row, 16 * XB, 0, valid, t); add_blocks<XB>(g, t); }
    if (dx0_acc) { load_row<XB>(dx0_acc, row, 16 * XB, 0, valid, t); add_blocks<XB>(g, t); }
#pragma unroll
    for (int b = 0; b < XB; ++b) dst[b] = g[b] + g[XB + b];
  }
};

// two aligned sources added together (optional row gather on both)
template <int NB_>
struct LoadAdd2 {
  static constexpr int NB = NB_;
  const float* p0; int stride0; int col0;
  const float* p1; int stride1; int col1;     // p1 may be nullptr
  const int* idx;                             // row gather applied to both sources, or nullptr
  __device__ __forceinline__ void operator()(long row, bool valid, v4f* dst) const {
    long r = row;
    if (idx && valid) r = idx[row];
    load_row<NB>(p0, r, stride0, col0, valid, dst);
    if (p1) {
      v4f t[NB];
      load_row<NB>(p1, r, stride1, col1, valid, t);
      add_blocks<NB>(dst, t);
    }
  }
};

// d(pre-sigmoid) = d_prob * p * (1 - p)   (Sigmoid at clr_att_gnn.py:57)
struct LoadSigmoidGrad {
  static constexpr int NB = 1;
  const float* d_prob;   // [rows,1] or nullptr
  const float* prob;     // [rows,1]
  __device__ __forceinline__ void operator()(long row, bool valid, v4f* dst) const {
    const int q = (threadIdx.x & 63) >> 4;
    v4f v = {0.f, 0.f, 0.f, 0.f};
    if (valid && q == 0 && d_prob) { const float p = prob[row]; v.x = d_prob[row] * p * (1.f - p); }
    dst[0] = v;
  }
};

// ---- row storers ------------------------------------------------------------------------------
template <int NB_>
struct StoreAligned {
  static constexpr int NB = NB_;
  float* ptr; const int* idx; int stride; int col0;
  __device__ __forceinline__ void operator()(long row, bool valid, const v4f* src) const {
    long r = row;
    if (idx && valid) r = idx[row];
    if (ptr) store_row<NB>(ptr, r, stride, col0, valid, src);
  }
};

template <int NB_>
struct StoreTwo {                    // the same rows to two destinations (e.g. x[0] and the returned x_enc)
  static constexpr int NB = NB_;
  float* p0; float* p1; int stride;
  __device__ __forceinline__ void operator()(long row, bool valid, const v4f* src) const {
    store_row<NB>(p0, row, stride, 0, valid, src);
    store_row<NB>(p1, row, stride, 0, valid, src);
  }
};

struct StoreScalar {                 // feature 0 -> [rows, 1], optional sigmoid (clr_att_gnn.py:57)
  static constexpr int NB = 1;
  float* ptr; int sigmoid;
  __device__ __forceinline__ void operator()(long row, bool valid, const v4f* src) const {
    const int q = (threadIdx.x & 63) >> 4;
    if (valid && q == 0) {
      float v = src[0].x;
      if (sigmoid) v = 1.f / (1.f + __expf(-v));
      ptr[row] = v;
    }
  }
};

struct StoreNone {
  static constexpr int NB = 1;
  __device__ __forceinline__ void operator()(long, bool, const v4f*) const {}
};

// ---- forward chain ------------------------------------------------------------------------------
template <class In, class Out>
struct ChainFwdArgs {
  int rows;
  In in;
  Out out;
  float* save_in;                       // padded copy of the input row [rows, KP0] or nullptr
  float* save[kChainMaxLayers];         // hidden activations [rows, NP_l] (training) or nullptr
  const float* wpack;
};

template <class Seq, unsigned RELU_MASK, int LI, class Args, class WS>
__device__ __forceinline__ void chain_fwd_rec(WS& ws, bool more, const Args& a, long row,
                                              bool valid, const v4f* in) {
  constexpr int NB = Seq::np(LI) / 16;
  v4f out[NB];
  linear<Seq, LI, ((RELU_MASK >> LI) & 1u) != 0>(ws, more, in, out);
  if constexpr (LI + 1 < Seq::NL) {
    if (a.save[LI]) store_row<NB>(a.save[LI], row, Seq::np(LI), 0, valid, out);
    chain_fwd_rec<Seq, RELU_MASK, LI + 1>(ws, more, a, row, valid, out);
  } else {
    a.out(row, valid, out);
  }
}

template <class Seq, unsigned RELU_MASK, class In, class Out, int NW>
__global__ __launch_bounds__(NW * 64, NW >= 8 ? 2 : 1) void chain_fwd_kernel(const ChainFwdArgs<In, Out> a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  static_assert(In::NB == Seq::kp(0) / 16, "loader width != first layer input width");
  WStreamG<NW * 64, Seq, Seq::SLOT, 2 * Seq::SLOT> ws;       // narrow stacks: the whole image usually fits where the ring was
  ws.init(a.wpack, smem);
  ws.template start<Seq>();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ntiles = (a.rows + NW * 16 - 1) / (NW * 16);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const bool more = tile + (int)gridDim.x < ntiles;
    const long row = (long)tile * (NW * 16) + wave * kRowsPerWave + (lane & 15);
    const bool valid = row < a.rows;
    v4f in[In::NB];
    a.in(row, valid, in);
    if (a.save_in) store_row<In::NB>(a.save_in, row, 16 * In::NB, 0, valid, in);
    chain_fwd_rec<Seq, RELU_MASK, 0>(ws, more, a, row, valid, in);
  }
}

// ---- backward chain (data gradient) -------------------------------------------------------------
// SeqT lists the TRANSPOSED images from the last layer down: L<NP_L, NP_{L-1}>, ..., and, when the
// gradient of the input is wanted, finally L<NP_1, KP_1>.  G_L comes from the loader; for each
// further step G_{l-1} = (W_l^T G_l) * (h_{l-1} > 0).  Every G_l (l < L) is stored for the weight
// gradient; act[i] is the saved activation that masks the output of step i (nullptr: no mask,
// i.e. the final step that yields the input gradient).
template <class In, class Out>
struct ChainBwdArgs {
  int rows;
  In in;
  Out out;                              // receives the result of the LAST step
  float* gtop;                          // padded copy of G_L [rows, KP of SeqT layer 0] or nullptr
  const float* act[kChainMaxLayers];
  float* gsave[kChainMaxLayers];        // G after step i [rows, NP of SeqT layer i] or nullptr
  const float* wpack;
};

template <class SeqT, int LI, class Args, class WS>
__device__ __forceinline__ void chain_bwd_rec(WS& ws, bool more, const Args& a, long row,
                                              bool valid, const v4f* g) {
  constexpr int NB = SeqT::np(LI) / 16;
  v4f d[NB], act[NB];
  // the saved activation is fetched under this layer's products (round 5: it used to be loaded behind them, a round trip per layer
  // in front of every relu')
  linear<SeqT, LI, false, false>(ws, more, g, d, [&]() {
    if (a.act[LI]) load_row<NB>(a.act[LI], row, SeqT::np(LI), 0, valid, act);
  });
  if (a.act[LI]) relu_bwd<NB>(d, act);
  if (a.gsave[LI]) store_row<NB>(a.gsave[LI], row, SeqT::np(LI), 0, valid, d);
  if constexpr (LI + 1 < SeqT::NL) {
    chain_bwd_rec<SeqT, LI + 1>(ws, more, a, row, valid, d);
  } else {
    a.out(row, valid, d);
  }
}

template <class SeqT, class In, class Out, int NW>
__global__ __launch_bounds__(NW * 64, NW >= 8 ? 2 : 1) void chain_bwd_kernel(const ChainBwdArgs<In, Out> a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  static_assert(In::NB == SeqT::kp(0) / 16, "loader width != top gradient width");
  WStreamG<NW * 64, SeqT, SeqT::SLOT, 2 * SeqT::SLOT> ws;
  ws.init(a.wpack, smem);
  ws.template start<SeqT>();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ntiles = (a.rows + NW * 16 - 1) / (NW * 16);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const bool more = tile + (int)gridDim.x < ntiles;
    const long row = (long)tile * (NW * 16) + wave * kRowsPerWave + (lane & 15);
    const bool valid = row < a.rows;
    v4f g[In::NB];
    a.in(row, valid, g);
    if (a.gtop) store_row<In::NB>(a.gtop, row, 16 * In::NB, 0, valid, g);
    chain_bwd_rec<SeqT, 0>(ws, more, a, row, valid, g);
  }
}

// ---- one WIDE Linear layer (input up to 640 features in registers, output streamed to memory) ------
// Used for att_edge_encoder (640-512-384-256-128-64, clr_att_gnn.py:81-91): input + output of such a
// layer do not fit the register file together, so every finished 16-feature output block goes
// straight to HBM.  mask != nullptr: multiply by (mask > 0) -- the ReLU derivative in backward.
template <class In>
struct WideArgs {
  int rows;
  In in;
  float* out;            // [rows, out_stride]
  int out_stride;
  int out_col0;
  // ReLU masks (b3d_dev.hpp), one [rows, 16]-float plane per tensor: the lane (row m, quarter q) owns 16 bytes, word b / 8 holds its
  // blocks 8 (b / 8) ... (at most 512 outputs = 4 words)
  const unsigned* mask_in;   // backward: the output is multiplied by (forward activation > 0) read from here, or nullptr
  unsigned* mask_out;        // forward (RELU): the bits of THIS layer's output are written here, or nullptr
  const float* wpack;
};

template <class Seq, bool RELU, bool BIAS, class In, int NW>
__global__ __launch_bounds__(NW * 64, NW >= 8 ? 2 : 1) void wide_linear_kernel(const WideArgs<In> a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  static_assert(Seq::NL == 1 && In::NB == Seq::kp(0) / 16, "one layer; loader width = input width");
  constexpr int NB = Seq::np(0) / 16;
  static_assert(NB <= 32, "four mask words per lane");
  WStreamT<NW * 64, Seq::SLOT> ws;
  ws.init(a.wpack, smem);
  ws.template start<Seq>();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, q = lane >> 4;
  const int ntiles = (a.rows + NW * 16 - 1) / (NW * 16);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const bool more = tile + (int)gridDim.x < ntiles;
    const long row = (long)tile * (NW * 16) + wave * kRowsPerWave + (lane & 15);
    const bool valid = row < a.rows;
    v4f in[In::NB];
    a.in(row, valid, in);
    // Output / mask rows as (uniform base) + 32-bit byte offset: hipcc then selects the saddr form of global_store / global_load and
    // keeps no 64-bit row address alive across the layer -- with pointer pairs the 512-wide instances ran out of registers and
    // RELOADED the output base from scratch in front of every store, behind an s_waitcnt vmcnt(0) that also drained the weight
    // chunk in flight (round 5).  The launcher checks that the buffers stay below 4 GB.
    const unsigned ooff = ((unsigned)row * (unsigned)a.out_stride + (unsigned)a.out_col0 + 4u * (unsigned)q) * 4u;
    const unsigned moff = ((unsigned)row * 4u + (unsigned)q) * 16u;      // this lane's four words of a mask plane
    char* const obase = reinterpret_cast<char*>(a.out);
    u4v mk = {0u, 0u, 0u, 0u};
    if (a.mask_in && valid) mk = *reinterpret_cast<const u4v*>(reinterpret_cast<const char*>(a.mask_in) + moff);
    unsigned word = 0u;                                            // the mask word being consumed (backward) / built (forward)
    unsigned* wp = &word;
    const u4v* mkp = &mk;
    const bool masked = a.mask_in != nullptr;
    char* const mbase = reinterpret_cast<char*>(a.mask_out);
    const bool mwrite = RELU && a.mask_out != nullptr && valid;
    linear_emit<Seq, 0, RELU, BIAS>(ws, more, in, [=](int mb, v4f v) {
      if (masked) {                                                // blocks arrive in ascending order
        if ((mb & 7) == 0) *wp = (*mkp)[mb >> 3];
        v.x = keep_if_msb(*wp, v.x); v.y = keep_if_msb(*wp, v.y); v.z = keep_if_msb(*wp, v.z); v.w = keep_if_msb(*wp, v.w);
      }
      if constexpr (RELU) {
        if (mwrite) {
          if ((mb & 7) == 0) *wp = 0u;
          *wp = push_positive(*wp, v.x); *wp = push_positive(*wp, v.y); *wp = push_positive(*wp, v.z); *wp = push_positive(*wp, v.w);
          if ((mb & 7) == 7 || mb == NB - 1) *reinterpret_cast<unsigned*>(mbase + moff + 4u * (unsigned)(mb >> 3)) = *wp << (28 - 4 * (mb & 7));
        }
      }
      if (valid) *reinterpret_cast<v4f*>(obase + ooff + 64u * (unsigned)mb) = v;
    });
  }
}

}  // namespace b3d
