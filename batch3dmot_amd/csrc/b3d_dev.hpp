// Device building blocks for the gfx950 (MI355X, CDNA4) message-passing kernels.
//
// Execution shape shared by every MLP kernel in this library
// ----------------------------------------------------------
//  * workgroup = 512 threads = 8 wavefronts (2 per SIMD); a wavefront owns 16 rows (edges or
//    nodes) and carries their activations through a whole MLP stack IN REGISTERS.
//  * every Linear layer is computed transposed, Y^T[out][row] = W[out][in] . X^T[in][row], with
//    v_mfma_f32_16x16x4_f32 (exact fp32, bitwise an fmaf chain): the 16x16 accumulator of one
//    layer (lane = row, registers = 4 consecutive output features) is directly the B operand of
//    the next layer, so activations never leave the register file and need no transposition.
//    "Layout L": lane l = (i = l & 15, q = l >> 4) holds, for feature block b (16 features),
//    features 16 b + 4 q + {0,1,2,3} of row i as one float4.
//  * weights are the A operand.  They are pre-packed ("images", two formats: see the geometry section) and streamed global -> LDS in chunks of <= 52 KB through a two-buffer ring,
//    asynchronously with LDS-DMA (global_load_lds_dwordx4) while the previous chunk is being
//    multiplied; one workgroup barrier per chunk.
//
// No CUDA compatibility layer, no multi-backend dispatch: this file only builds for gfx950.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>

namespace b3d {

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kWaves = 8;                 // wavefronts per workgroup
constexpr int kThreads = kWaves * 64;     // 512
constexpr int kRowsPerWave = 16;
constexpr int kTileRows = kWaves * kRowsPerWave;   // 128 rows per workgroup tile
constexpr int kWBufFloats = 13312;        // one weight ring slot: 52 KB
constexpr int kChunkAlign = 256;          // floats; 1 KB = one wave-wide 16-byte LDS-DMA
constexpr int kLdsBytes = 2 * kWBufFloats * 4;

__host__ __device__ constexpr int round_up(int a, int b) { return (a + b - 1) / b * b; }
__host__ __device__ constexpr int pad16(int a) { return round_up(a, 16); }

// ---- packed weight image geometry (shared by host packer and device consumer) ------------
// Two image formats, chosen per layer:
//  * fp32 (KP not a multiple of 32, or opted out): NP = pad16(N) rows, row stride KP + 8 floats, columns [0,K) =
//    W[r][c], column KP = bias[r], everything else zero.  Consumed by v_mfma_f32_16x16x4_f32 (exact fp32).
//  * bf16x3 ("BF", KP a multiple of 32): every weight is split EXACTLY into three bf16 pieces (8 + 8 + 8 significand
//    bits by truncation, w = w0 + w1 + w2); a row is [piece 0: KP/2 dwords][piece 1][piece 2][bias (fp32)][pad],
//    stride 3 KP / 2 + 8 dwords.  Inside a piece the 32 features of group c sit at dwords 16 c .. 16 c + 15 in the
//    order in which a wavefront HOLDS the next layer's operand (bf_pos below), so that the fp32 accumulator of one
//    layer turns into the bf16 B operand of the next without any lane movement.  Consumed by
//    v_mfma_f32_16x16x32_bf16 as "bf16x6": the activations are split the same way in registers and a product is
//    formed from the six piece products of weight >= 2^-24 (w0x0, w0x1, w1x0, w0x2, w1x1, w2x0), accumulated in
//    fp32: fp32-class accuracy (3.5e-7 vs 2.0e-7 of the fmaf chain on two chained layers,
//    tools/micro/bf16x6_linear.hip) at 16 / 6 of the fp32 MFMA rate (measured 258 vs 134 TFLOP/s fp32-equivalent).
// Both strides are = 8 mod 16 dwords: conflict-free ds_read_b128 of a 16-row fragment (a stride = 4 mod 16 is 2-way
// conflicted on gfx950's b128 lane groups).  The image is cut into row chunks that fit one ring slot; every chunk is
// padded to a multiple of 1 KB.
#ifndef B3D_BF16X6
#define B3D_BF16X6 1
#endif
__host__ __device__ constexpr bool bf_auto(int KP) { return B3D_BF16X6 != 0 && KP % 32 == 0 && KP >= 64 && KP <= 256; }   // narrower layers are not MFMA bound
__host__ __device__ constexpr int row_stride(int KP, bool bf) { return bf ? 3 * KP / 2 + 8 : KP + 8; }
__host__ __device__ constexpr int bias_col(int KP, bool bf) { return bf ? 3 * KP / 2 : KP; }
// position (0..31) inside its 32-feature group at which feature f (0..31) of the group is stored / held:
// lane quarter g = (f & 15) >> 2 holds element j = 4 (f >> 4) + (f & 3) of the bf16 operand fragment
__host__ __device__ constexpr int bf_pos(int f) { return 8 * ((f & 15) >> 2) + 4 * (f >> 4) + (f & 3); }
__host__ __device__ constexpr int chunk_rows(int KP, int NP, bool bf) {
  // whole 16-row blocks; 128-row multiples where they fit, so that a chunk boundary is a multiple of the wave count
  // for the kernels that deal output blocks round-robin to 4 or 8 wavefronts (b3d_node.hpp)
  int r = kWBufFloats / row_stride(KP, bf);
  r = r >= 128 ? r / 128 * 128 : r / 16 * 16;
  return r < NP ? r : NP;
}
__host__ __device__ constexpr int n_chunks(int KP, int NP, bool bf) {
  return (NP + chunk_rows(KP, NP, bf) - 1) / chunk_rows(KP, NP, bf);
}
__host__ __device__ constexpr int chunk_nrows(int KP, int NP, bool bf, int c) {   // rows in chunk c
  int cr = chunk_rows(KP, NP, bf);
  int left = NP - c * cr;
  return left < cr ? left : cr;
}
__host__ __device__ constexpr int chunk_floats(int KP, bool bf, int rows) {
  return round_up(rows * row_stride(KP, bf), kChunkAlign);
}
__host__ __device__ constexpr int image_floats(int KP, int NP, bool bf) {
  int tot = 0;
  for (int c = 0; c < n_chunks(KP, NP, bf); ++c) tot += chunk_floats(KP, bf, chunk_nrows(KP, NP, bf, c));
  return tot;
}

// A layer of a kernel's weight sequence: padded input width KP, padded output width NP, image format
// (BF_: -1 = by the rule above, 0 = fp32, 1 = bf16x3).
template <int KP_, int NP_, int BF_ = -1>
struct L {
  static constexpr int KP = KP_, NP = NP_;
  static constexpr bool BF = BF_ < 0 ? bf_auto(KP_) : BF_ != 0;
  static_assert(KP_ % 16 == 0 && NP_ % 16 == 0, "pad layer widths to multiples of 16");
  static_assert(!BF || KP_ % 32 == 0, "the bf16 operand covers 32 features");
};
template <int KP_, int NP_> using LF = L<KP_, NP_, 0>;       // fp32 image, exact fmaf chain

// The ordered list of layers a kernel consumes (one weight image each).
template <class... Ls>
struct LayerSeq {
  static constexpr int NL = sizeof...(Ls);
  __host__ __device__ static constexpr int kp(int li) { constexpr int a[] = {Ls::KP...}; return a[li]; }
  __host__ __device__ static constexpr int np(int li) { constexpr int a[] = {Ls::NP...}; return a[li]; }
  __host__ __device__ static constexpr bool bf(int li) { constexpr bool a[] = {Ls::BF...}; return a[li]; }
  __host__ __device__ static constexpr int layer_chunks(int li) { return n_chunks(kp(li), np(li), bf(li)); }
  __host__ __device__ static constexpr int first_chunk(int li) {
    int c = 0;
    for (int i = 0; i < li; ++i) c += layer_chunks(i);
    return c;
  }
  static constexpr int NCH = first_chunk(NL);
  __host__ __device__ static constexpr int layer_off(int li) {     // float offset of the image
    int o = 0;
    for (int i = 0; i < li; ++i) o += image_floats(kp(i), np(i), bf(i));
    return o;
  }
  static constexpr int TOTAL_FLOATS = layer_off(NL);
  __host__ __device__ static constexpr int chunk_layer(int ci) {
    int li = 0;
    while (ci >= first_chunk(li + 1)) ++li;
    return li;
  }
  __host__ __device__ static constexpr int chunk_off(int ci) {
    int li = chunk_layer(ci);
    int o = layer_off(li);
    for (int c = 0; c < ci - first_chunk(li); ++c) o += chunk_floats(kp(li), bf(li), chunk_nrows(kp(li), np(li), bf(li), c));
    return o;
  }
  __host__ __device__ static constexpr int chunk_size(int ci) {    // floats, padded
    int li = chunk_layer(ci);
    return chunk_floats(kp(li), bf(li), chunk_nrows(kp(li), np(li), bf(li), ci - first_chunk(li)));
  }
  __host__ __device__ static constexpr int max_chunk() {           // floats of the largest chunk
    int m = 0;
    for (int ci = 0; ci < NCH; ++ci) m = chunk_size(ci) > m ? chunk_size(ci) : m;
    return m;
  }
  // ring slot size for kernels that size their LDS by the sequence (narrow stacks); the message-passing
  // kernels use the full kWBufFloats
  static constexpr int SLOT = max_chunk();
};

// ---- experiment support: per-workgroup phase time stamps (off unless built with -DB3D_EXP_STAMPS) ----
#ifdef B3D_EXP_STAMPS
static __device__ long long g_stamps[4][512 * 32];     // one copy per translation unit (no relocatable device code)
#define B3D_STAMP(k, i) do { if (threadIdx.x == 0 && blockIdx.x < 512) b3d::g_stamps[k][blockIdx.x * 32 + (i)] = wall_clock64(); } while (0)
// time wavefront 0 spends inside the ring's acquire (counted wait + barrier) of the node-sized kernels, summed per workgroup:
// zeroed where a kernel starts its stamps, saved into one of its stamp slots where it ends them
static __device__ long long g_acq[512];
#define B3D_ACQ_ZERO() do { if (threadIdx.x == 0 && blockIdx.x < 512) b3d::g_acq[blockIdx.x] = 0; } while (0)
#define B3D_ACQ_T0() const long long acq_t0_ = wall_clock64()
#define B3D_ACQ_ADD() do { if (threadIdx.x == 0 && blockIdx.x < 512) b3d::g_acq[blockIdx.x] += wall_clock64() - acq_t0_; } while (0)
#define B3D_ACQ_SAVE(k, i) do { if (threadIdx.x == 0 && blockIdx.x < 512) b3d::g_stamps[k][blockIdx.x * 32 + (i)] = b3d::g_acq[blockIdx.x]; } while (0)
#else
#define B3D_STAMP(k, i) do {} while (0)
#define B3D_ACQ_ZERO() do {} while (0)
#define B3D_ACQ_T0() do {} while (0)
#define B3D_ACQ_ADD() do {} while (0)
#define B3D_ACQ_SAVE(k, i) do {} while (0)
#endif

// ---- weight stream: global -> LDS, two slots ----------------------------------------------
#ifndef B3D_USE_LDS_DMA
#define B3D_USE_LDS_DMA 1
#endif

template <int NT, int SLOT = kWBufFloats>
struct WStreamT {
  static constexpr bool kDirect = false;
  const float* g;      // packed images of this kernel (global)
  float* lds;          // 2 * SLOT floats
  int slot;            // slot that the NEXT acquire returns
#if !B3D_USE_LDS_DMA
  v4f pre[(SLOT / 4 + NT - 1) / NT];
#endif

  __device__ __forceinline__ void init(const float* gw, float* l) { g = gw; lds = l; slot = 0; }

  // A0, NA (round 6, node-sized kernels): wavefronts [A0, A0 + NA) mod NT / 64 have MFMA work in the chunk that is consumed while this
  // one is fetched; the OTHERS issue its pieces (an LDS-DMA instruction stalls its issuer for 175-350 cycles: with every wavefront
  // issuing, the few that own a block of a 3-4 block chunk started their MFMAs that much later).  NA == 0: everyone issues.
  template <class Seq, int CI, int A0 = 0, int NA = 0>
  __device__ __forceinline__ void issue(int to_slot) {
    constexpr int off = Seq::chunk_off(CI);
    constexpr int n4 = Seq::chunk_size(CI) / 4;          // multiple of 64
    static_assert(Seq::chunk_size(CI) <= SLOT, "chunk larger than a ring slot");
#if B3D_USE_LDS_DMA
    const int lane = threadIdx.x & 63;
    constexpr int NW = NT / 64, NI = NW - NA;            // issuers
    static_assert(NA >= 0 && NI >= 1, "somebody has to issue");
    const int wave = NA == 0 ? (int)(threadIdx.x >> 6) : ((int)(threadIdx.x >> 6) - A0 - NA + 2 * NW) % NW;   // issuer index
    const float* src = g + off;
    // opaque per call: hipcc otherwise hoists the per-piece 64-bit source addresses of EVERY chunk of the sequence
    // out of the tile loop (up to ~150 address pairs) and spills them
    // ... and SCALAR (round 5: "+s", it was "+v"): the chunk base stays in an SGPR pair and a piece's address is base + a 32-bit lane
    // offset -- the saddr form of global_load_lds.  As a VGPR pair every piece carried its own 64-bit address (v_lshl_add_u64 per
    // piece, five pairs kept live in the 512-wide kernels, which then spilled the output base and reloaded it behind a vmcnt(0)).
    asm volatile("" : "+s"(src));
    float* dst = lds + to_slot * SLOT;
#pragma unroll
    for (int i0 = 0; i0 < n4; i0 += NI * 64) {
      const int base = i0 + wave * 64;                   // wave-uniform
      if ((NA == 0 || wave < NI) && base < n4) {
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(src + (unsigned)(base + lane) * 4u),
            (__attribute__((address_space(3))) void*)(dst + (size_t)base * 4), 16, 0, 0);
      }
    }
#else
    (void)to_slot;
    const float* src = g + off;
#pragma unroll
    for (int j = 0; j < (n4 + NT - 1) / NT; ++j) {
      const int i = j * NT + threadIdx.x;
      if (i < n4) pre[j] = *reinterpret_cast<const v4f*>(src + (size_t)i * 4);
    }
#endif
  }

  // Start the stream: chunk 0 of the first tile.
  template <class Seq>
  __device__ __forceinline__ void start() { issue<Seq, 0>(slot); }

  // Make chunk CI visible to the whole workgroup and start fetching the next one.
  // more == false suppresses the wrap-around prefetch after the kernel's last chunk.
  template <class Seq, int CI, int A0 = 0, int NA = 0>
  __device__ __forceinline__ const float* acquire(bool more) {
    constexpr int NXT = (CI + 1) % Seq::NCH;
#if B3D_USE_LDS_DMA
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (NXT != 0 || more) issue<Seq, NXT, A0, NA>(slot ^ 1);
#else
    {
      constexpr int n4 = Seq::chunk_size(CI) / 4;
      float* dst = lds + slot * SLOT;
#pragma unroll
      for (int j = 0; j < (n4 + NT - 1) / NT; ++j) {
        const int i = j * NT + threadIdx.x;
        if (i < n4) *reinterpret_cast<v4f*>(dst + (size_t)i * 4) = pre[j];
      }
    }
    __syncthreads();
    if (NXT != 0 || more) issue<Seq, NXT>(slot ^ 1);
#endif
    const float* cur = lds + slot * SLOT;
    slot ^= 1;
    return cur;
  }
};

// ---- no ring at all (round 6 EXPERIMENT behind -DB3D_NODE_DIRECT=1: 3 x SLOWER on row-major images, see b3d_node.hpp) ---------------
// linear_split gives every 16-row weight block to ONE wavefront, so nothing is shared through the LDS copy the ring makes: with 16 rows
// per workgroup against ~1 MB of weights those kernels were a chain of 16+ (LDS-DMA chunk in flight -> vmcnt(0) -> barrier -> a few
// MFMAs) steps, one chunk in flight at a time -- bound by DMA latency at 14-18 % MFMA-busy.  Here a wavefront reads the fragments of its
// own blocks straight from global memory (L2-resident images) into registers, several groups ahead; the only barrier left per layer
// is the one that publishes the previous layer's activations.  Same images, same MFMA order, same bits.
template <int NT>
struct WDirectT {
  static constexpr bool kDirect = true;
  const float* g;
  __device__ __forceinline__ void init(const float* gw, float*) { g = gw; }
  template <class Seq>
  __device__ __forceinline__ void start() {}
  template <class Seq, int CI, int A0 = 0, int NA = 0>
  __device__ __forceinline__ const float* acquire(bool) {
    if constexpr (CI == Seq::first_chunk(Seq::chunk_layer(CI))) __syncthreads();   // the previous layer's LDS writes of every wavefront
    return g + Seq::chunk_off(CI);
  }
};
typedef const __attribute__((address_space(1))) v4f* gbl_v4f_cp;
typedef const __attribute__((address_space(1))) float* gbl_f_cp;

// ---- grouped weight stream: consecutive chunks that fit one slot together travel under ONE barrier ----------
// A Linear of the hoisted edge stacks is 9-25 KB of weights and 32-96 MFMAs per wavefront: with one barrier per
// layer a tile spends about as long in barriers (~0.7 us each: drain, rendezvous, first LDS fragment) as in some of
// its layers.  Chunks are grouped greedily in sequence order up to GSLOT floats; the group leader's acquire waits,
// synchronises and prefetches the next group, the other chunks of the group only compute their address.
template <class Seq, int GSLOT, int FIRST = GSLOT>
struct ChunkGroups {                                          // FIRST: cap of the first group (a short one starts the MFMAs early)
  __host__ __device__ static constexpr int leader(int ci) {
    int lead = 0, acc = 0;
    for (int c = 0; c <= ci; ++c) {
      const int sz = Seq::chunk_size(c);
      if (c == 0 || acc + sz > (lead == 0 ? FIRST : GSLOT)) { lead = c; acc = sz; } else acc += sz;
    }
    return lead;
  }
  __host__ __device__ static constexpr int next_leader(int ci) {     // first chunk of the following group (NCH: none)
    int c = ci + 1;
    while (c < Seq::NCH && leader(c) != c) ++c;
    return c;
  }
  __host__ __device__ static constexpr int group_floats(int lead) {
    const int nl = next_leader(lead);
    return (nl < Seq::NCH ? Seq::chunk_off(nl) : Seq::TOTAL_FLOATS) - Seq::chunk_off(lead);
  }
  __host__ __device__ static constexpr int group_index(int ci) {
    int k = 0;
    for (int c = 1; c <= ci; ++c) k += (leader(c) == c) ? 1 : 0;
    return k;
  }
};

// RESIDENT (the whole sequence fits LDS next to nothing else): the image sits in LDS at its global offsets and is
// loaded ONCE per workgroup, in front of the first layer (under the tile's gather prologue); one barrier per
// kernel, none between layers, and later tiles of the workgroup find everything in place.
// Otherwise: a two-slot ring of GSLOT floats, one group per slot.
constexpr int kResidentMaxFloats = 35 * 1024;                 // 140 KB of the CU's 160 KB
template <class Seq, int RING_SLOT = kWBufFloats, int RES_MAX = kResidentMaxFloats>
__host__ __device__ constexpr int stream_lds_bytes() { return (Seq::TOTAL_FLOATS <= RES_MAX ? Seq::TOTAL_FLOATS : 2 * RING_SLOT) * 4; }

// RING_SLOT: slot size of the ring form; RES_MAX: the sequence is kept resident if it is no larger than this.
template <int NT, class SeqT, int RING_SLOT = kWBufFloats, int RES_MAX = kResidentMaxFloats>
struct WStreamG {
  static constexpr bool RESIDENT = SeqT::TOTAL_FLOATS <= RES_MAX;
  static constexpr int GSLOT = RESIDENT ? SeqT::TOTAL_FLOATS : RING_SLOT;
  // One group when resident: a short first group (MFMAs start earlier, the rest streams underneath) was measured
  // SLOWER, 27.9 vs 24.5 us -- while any LDS-DMA is pending hipcc drains vmcnt(0) at every use of a load result.
  static constexpr int FIRST = GSLOT;
  using G = ChunkGroups<SeqT, GSLOT, FIRST>;
  const float* g;
  float* lds;
  const float* base;   // start of the group being consumed
  int slot;            // ring form: slot that the next leader acquire returns
  int loaded;          // resident form: groups already in place

  __device__ __forceinline__ void init(const float* gw, float* l) { g = gw; lds = l; slot = 0; base = l; loaded = 0; }

  template <int LEAD>
  __device__ __forceinline__ void issue(float* dst) {
    constexpr int off = SeqT::chunk_off(LEAD);
    constexpr int n4 = G::group_floats(LEAD) / 4;
    static_assert(G::group_floats(LEAD) <= GSLOT && n4 % 64 == 0, "group larger than a slot");
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float* src = g + off;
    asm volatile("" : "+s"(src));          // see WStreamT::issue
#pragma unroll
    for (int i0 = 0; i0 < n4; i0 += NT) {
      const int b = i0 + wave * 64;
      if (b < n4) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (unsigned)(b + lane) * 4u),
                                         (__attribute__((address_space(3))) void*)(dst + (size_t)b * 4), 16, 0, 0);
      }
    }
  }
  template <class Seq>
  __device__ __forceinline__ void start() {
    static_assert(std::is_same<Seq, SeqT>::value, "stream bound to another sequence");
    issue<0>(lds);
  }

  template <class Seq, int CI>
  __device__ __forceinline__ const float* acquire(bool more) {
    static_assert(std::is_same<Seq, SeqT>::value, "stream bound to another sequence");
    constexpr int LEAD = G::leader(CI);
    if constexpr (LEAD != CI) {
      constexpr int OFF = SeqT::chunk_off(CI) - SeqT::chunk_off(LEAD);    // forced constant: the offset functions loop
      return base + OFF;
    } else if constexpr (RESIDENT) {
      constexpr int K = G::group_index(CI), NXT = G::next_leader(CI), OFF = SeqT::chunk_off(CI);
      if (loaded <= K) {                                       // wave-uniform; false on every tile after the first
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if constexpr (NXT < SeqT::NCH) issue<NXT>(lds + SeqT::chunk_off(NXT));
        loaded = K + 1;
      }
      base = lds + OFF;
      return base;
    } else {
      constexpr int NXT = G::next_leader(CI) % SeqT::NCH;     // NCH -> 0: the next tile's first group
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (NXT != 0 || more) issue<NXT>(lds + (slot ^ 1) * GSLOT);
      base = lds + slot * GSLOT;
      slot ^= 1;
      return base;
    }
  }
};

using WStream = WStreamT<kThreads>;

// ---- one Linear layer on a 16-row wave tile, activations in registers ---------------------
__device__ __forceinline__ v4f mfma4(const v4f a, const v4f b, v4f c) {
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, c, 0, 0, 0);
  return c;
}

__device__ __forceinline__ v4f relu4(v4f a) {
  v4f r;
  r.x = relu1(a.x); r.y = relu1(a.y); r.z = relu1(a.z); r.w = relu1(a.w);
  return r;
}

// ---- bf16x6: operand fragments of v_mfma_f32_16x16x32_bf16 --------------------------------------------------
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
struct Bf3 { bf8 p0, p1, p2; };                             // the three pieces of 8 operand elements

// Exact three-way split of two layout-L blocks of one row (features 32 c + 4 q + {0..3} and 32 c + 16 + 4 q + {0..3})
// into the B-operand fragments of feature group c.  Truncation split: x0 = top 16 bits of x, r = x - x0 (exact, <= 16
// significant bits), x1 = top 16 bits of r, x2 = r - x1 (exact, <= 8 bits).
__device__ __forceinline__ Bf3 bf_split(const v4f a, const v4f b) {
  const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  unsigned h[8], m[8], l[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    h[i] = __float_as_uint(x[i]);
    const float r1 = x[i] - __uint_as_float(h[i] & 0xffff0000u);
    m[i] = __float_as_uint(r1);
    l[i] = __float_as_uint(r1 - __uint_as_float(m[i] & 0xffff0000u));
  }
  u4v q0, q1, q2;
#pragma unroll
  for (int d = 0; d < 4; ++d) {                             // v_perm_b32: the upper halves of two dwords
    q0[d] = __builtin_amdgcn_perm(h[2 * d + 1], h[2 * d], 0x07060302u);
    q1[d] = __builtin_amdgcn_perm(m[2 * d + 1], m[2 * d], 0x07060302u);
    q2[d] = __builtin_amdgcn_perm(l[2 * d + 1], l[2 * d], 0x07060302u);
  }
  return Bf3{__builtin_bit_cast(bf8, q0), __builtin_bit_cast(bf8, q1), __builtin_bit_cast(bf8, q2)};
}
__device__ __forceinline__ v4f bf_mfma6(const Bf3& w, const Bf3& x, v4f acc) {   // smallest terms first
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.p0, x.p2, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.p1, x.p1, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.p2, x.p0, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.p0, x.p1, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.p1, x.p0, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.p0, x.p0, acc, 0, 0, 0);
  return acc;
}
template <int KP>
__device__ __forceinline__ Bf3 bf_load(const float* p) {    // p: this lane's 16 bytes of piece 0; pieces are KP/2 dwords apart
  Bf3 f;
  f.p0 = __builtin_bit_cast(bf8, *reinterpret_cast<const u4v*>(p));
  f.p1 = __builtin_bit_cast(bf8, *reinterpret_cast<const u4v*>(p + KP / 2));
  f.p2 = __builtin_bit_cast(bf8, *reinterpret_cast<const u4v*>(p + KP));
  return f;
}
// The operand a layer consumes: the fp32 blocks themselves (fp32 image) or their bf16 pieces (BF image), made once per
// layer, in front of its first weight chunk.
template <int KP, bool BF>
struct LinIn {
  const v4f* in;
  __device__ __forceinline__ void prepare(const v4f* __restrict__ x) { in = x; }
};
template <int KP>
struct LinIn<KP, true> {
  Bf3 x[KP / 32];
  __device__ __forceinline__ void prepare(const v4f* __restrict__ in) {
#pragma unroll
    for (int c = 0; c < KP / 32; ++c) x[c] = bf_split(in[2 * c], in[2 * c + 1]);
  }
};

// act(W . in + b) for layer LI of Seq.  in: KB feature blocks (layout L); every finished output
// block is handed to emit(mb, value) -- either kept in registers (linear) or streamed to memory
// (linear_emit, for layers whose input + output do not fit the register file together).
struct NoHook { __device__ __forceinline__ void operator()() const {} };

template <class Seq, int LI, bool RELU, bool BIAS, int CH, class WS, class Emit, class Hook>
__device__ __forceinline__ void linear_chunk(WS& ws, bool more, const v4f* __restrict__ in_blocks, LinIn<Seq::kp(LI), Seq::bf(LI)>& xin,
                                             Emit& emit, Hook& hook, const v4f* init = nullptr) {
  constexpr int KP = Seq::kp(LI), NP = Seq::np(LI);
  constexpr bool BF = Seq::bf(LI);
  constexpr int KB = KP / 16, NB = NP / 16;
  constexpr int STRIDE = row_stride(KP, BF);
  constexpr int CR = chunk_rows(KP, NP, BF);
  constexpr int C0 = Seq::first_chunk(LI);
  constexpr int mb0 = CH * (CR / 16);
  constexpr int mbn = (mb0 + CR / 16 < NB) ? mb0 + CR / 16 : NB;
  const int lane = threadIdx.x & 63;
  const int m = lane & 15, q = lane >> 4;
  const float* w = ws.template acquire<Seq, C0 + CH>(more);
  // work that should overlap this layer's MFMAs instead of sitting in front of its barrier (the
  // acquire above drains vmcnt, stores included): e.g. the previous layer's activation stores.  The operand is
  // made AFTER the hook: a hook that stores the input blocks is the last reader of their fp32 form, so those
  // registers are free as soon as the bf16 pieces exist.
  if constexpr (CH == 0) { hook(); xin.prepare(in_blocks); }
  const float* wrow = w + m * STRIDE + 4 * q;                 // A fragment: row m of a block, this lane's 16 bytes
  const float* wbias = w + 4 * q * STRIDE + bias_col(KP, BF); // bias of output rows 4q..4q+3
  const v4f zero4 = {0.f, 0.f, 0.f, 0.f};
  if constexpr (BF) {
    // One output block at a time (a single accumulation chain of v_mfma_f32_16x16x32_bf16 issues back to back);
    // the three weight fragments (and the bias) of step t+1 are read from LDS BEFORE the 6 MFMAs of step t.
    constexpr int KG = KP / 32;
    auto frag = [&](int mb, int c) { return bf_load<KP>(wrow + (mb - mb0) * 16 * STRIDE + 16 * c); };
    auto bias = [&](int mb) -> v4f {
      const float* wb = wbias + (mb - mb0) * 16 * STRIDE;
      return v4f{wb[0], wb[STRIDE], wb[2 * STRIDE], wb[3 * STRIDE]};
    };
    Bf3 cur = frag(mb0, 0);
    v4f nb = BIAS ? bias(mb0) : zero4;
#pragma unroll
    for (int mb = mb0; mb < mbn; ++mb) {
      v4f acc = nb;
      if (init) acc += init[mb];
#pragma unroll
      for (int c = 0; c < KG; ++c) {
        Bf3 nxt = cur;
        if (c + 1 < KG) nxt = frag(mb, c + 1);
        else if (mb + 1 < mbn) {
          nxt = frag(mb + 1, 0);
          if constexpr (BIAS) nb = bias(mb + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        acc = bf_mfma6(cur, xin.x[c], acc);
        cur = nxt;
      }
      emit(mb, RELU ? relu4(acc) : acc);
    }
  } else {
  const v4f* __restrict__ in = xin.in;
  // Two output blocks at a time (two independent accumulator chains hide the 40-cycle dependent
  // latency of v_mfma_f32_16x16x4_f32 behind its 32-cycle issue interval).  The weight fragments
  // (and the bias) of step t+1 are read from LDS BEFORE the 8 MFMAs of step t are issued, so the
  // LDS latency is always covered by a full step of matrix work.
  constexpr int NPAIR = (mbn - mb0 + 1) / 2;
  auto frag = [&](int pair, int half, int kb) -> v4f {
    return *reinterpret_cast<const v4f*>(wrow + (pair * 2 + half) * 16 * STRIDE + 16 * kb);
  };
  auto bias = [&](int pair, int half) -> v4f {
    const float* wb = wbias + (pair * 2 + half) * 16 * STRIDE;
    return v4f{wb[0], wb[STRIDE], wb[2 * STRIDE], wb[3 * STRIDE]};
  };
  v4f fa0 = frag(0, 0, 0);
  v4f fa1 = (mb0 + 1 < mbn) ? frag(0, 1, 0) : zero4;
  v4f nb0 = BIAS ? bias(0, 0) : zero4;
  v4f nb1 = (BIAS && mb0 + 1 < mbn) ? bias(0, 1) : zero4;
#pragma unroll
  for (int pr = 0; pr < NPAIR; ++pr) {
    const int mb = mb0 + 2 * pr;
    const bool two = (mb + 1 < mbn);
    v4f acc0 = nb0, acc1 = nb1;
    if (init) {                                  // accumulators start from a caller-provided partial result
      acc0 += init[mb];
      if (two) acc1 += init[mb + 1];
    }
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      v4f na0 = zero4, na1 = zero4;
      if (kb + 1 < KB) {
        na0 = frag(pr, 0, kb + 1);
        if (two) na1 = frag(pr, 1, kb + 1);
      } else if (pr + 1 < NPAIR) {
        const bool two_n = (mb + 3 < mbn);
        na0 = frag(pr + 1, 0, 0);
        if (two_n) na1 = frag(pr + 1, 1, 0);
        if constexpr (BIAS) {
          nb0 = bias(pr + 1, 0);
          if (two_n) nb1 = bias(pr + 1, 1);
        }
      }
      // keep the LDS reads of the next step in front of this step's MFMAs (hipcc otherwise sinks
      // them next to their use and every 8-MFMA group starts with an exposed lgkmcnt(0) wait)
      __builtin_amdgcn_sched_barrier(0);
      acc0 = mfma4(fa0, in[kb], acc0);
      if (two) acc1 = mfma4(fa1, in[kb], acc1);
      fa0 = na0; fa1 = na1;
    }
    emit(mb, RELU ? relu4(acc0) : acc0);
    if (two) emit(mb + 1, RELU ? relu4(acc1) : acc1);
  }
  }
}

template <class Seq, int LI, bool RELU, bool BIAS, class WS, class Emit, class Hook, int... CH>
__device__ __forceinline__ void linear_impl(WS& ws, bool more, const v4f* __restrict__ in, Emit& emit, Hook& hook,
                                            std::integer_sequence<int, CH...>, const v4f* init = nullptr) {
  LinIn<Seq::kp(LI), Seq::bf(LI)> xin;
  (linear_chunk<Seq, LI, RELU, BIAS, CH, WS, Emit, Hook>(ws, more, in, xin, emit, hook, init), ...);
}

template <class Seq, int LI, bool RELU, bool BIAS = true, class WS, class Emit, class Hook = NoHook>
__device__ __forceinline__ void linear_emit(WS& ws, bool more, const v4f* __restrict__ in, Emit emit, Hook hook = Hook{}) {
  linear_impl<Seq, LI, RELU, BIAS, WS, Emit, Hook>(ws, more, in, emit, hook,
                                                   std::make_integer_sequence<int, Seq::layer_chunks(LI)>{});
}

template <class Seq, int LI, bool RELU, bool BIAS = true, class WS, class Hook = NoHook>
__device__ __forceinline__ void linear(WS& ws, bool more, const v4f* __restrict__ in, v4f* __restrict__ out,
                                       Hook hook = Hook{}) {
  linear_emit<Seq, LI, RELU, BIAS>(ws, more, in, [out](int mb, v4f v) { out[mb] = v; }, hook);
}

// out = act(W . in (+ b) + init): the accumulators start from `init` (e.g. the per-node part of a
// Linear over a concatenation, gathered per edge).  out may alias init.
template <class Seq, int LI, bool RELU, bool BIAS, class WS, class Hook = NoHook>
__device__ __forceinline__ void linear_init(WS& ws, bool more, const v4f* __restrict__ in, const v4f* init, v4f* out,
                                            Hook hook = Hook{}) {
  auto emit = [out](int mb, v4f v) { out[mb] = v; };
  linear_impl<Seq, LI, RELU, BIAS, WS, decltype(emit), Hook>(ws, more, in, emit, hook,
                                                             std::make_integer_sequence<int, Seq::layer_chunks(LI)>{}, init);
}

// ---- row <-> register helpers (layout L) -----------------------------------------------------
// NBLK feature blocks of one row, starting at column col0 (multiple of 4 floats).
template <int NBLK>
__device__ __forceinline__ void load_row(const float* __restrict__ base, long row, int stride,
                                         int col0, bool valid, v4f* __restrict__ dst) {
  const int q = (threadIdx.x & 63) >> 4;
  const float* p = base + row * (long)stride + col0 + 4 * q;
#pragma unroll
  for (int b = 0; b < NBLK; ++b) {
    dst[b] = valid ? *reinterpret_cast<const v4f*>(p + 16 * b) : v4f{0.f, 0.f, 0.f, 0.f};
  }
}

// Unconditional form: the caller guarantees a readable row (out-of-range tile rows are clamped to a valid
// row; their results are never stored).  A predicated load makes hipcc stage the result through temporaries
// and wait for every pair of loads before reusing them -- the gathers of a prologue then run serially.
template <int NBLK>
__device__ __forceinline__ void load_row_u(const float* __restrict__ base, long row, int stride, int col0,
                                           v4f* __restrict__ dst) {
  const int q = (threadIdx.x & 63) >> 4;
  const float* p = base + row * (long)stride + col0 + 4 * q;
#pragma unroll
  for (int b = 0; b < NBLK; ++b) dst[b] = *reinterpret_cast<const v4f*>(p + 16 * b);
}

// Make the compiler wait for a loaded value HERE.  While an LDS-DMA is pending hipcc waits vmcnt(0) (not a
// counted vmcnt) at the first use of any ordinary load result; if that first use sits behind the issue of
// the next weight chunk, the wavefront waits for that chunk before computing on the current one and the
// ring overlaps nothing.  Touching the value right in front of an acquire (which waits vmcnt(0) anyway)
// moves the wait to where it is free.
__device__ __forceinline__ void wait_for(const v4f& v) { asm volatile("" ::"v"(v.x)); }
template <int N>
__device__ __forceinline__ void wait_for(const v4f (&a)[N]) {      // every block: hipcc reorders the loads
#pragma unroll
  for (int b = 0; b < N; ++b) wait_for(a[b]);
}

template <int NBLK>
__device__ __forceinline__ void store_row(float* __restrict__ base, long row, int stride, int col0,
                                          bool valid, const v4f* __restrict__ src) {
  const int q = (threadIdx.x & 63) >> 4;
  float* p = base + row * (long)stride + col0 + 4 * q;
  if (valid) {
#pragma unroll
    for (int b = 0; b < NBLK; ++b) *reinterpret_cast<v4f*>(p + 16 * b) = src[b];
  }
}

// relu mask: g where act > 0 else 0
template <int NBLK>
__device__ __forceinline__ void relu_bwd(v4f* __restrict__ g, const v4f* __restrict__ act) {
#pragma unroll
  for (int b = 0; b < NBLK; ++b) {
    g[b].x = act[b].x > 0.f ? g[b].x : 0.f;
    g[b].y = act[b].y > 0.f ? g[b].y : 0.f;
    g[b].z = act[b].z > 0.f ? g[b].z : 0.f;
    g[b].w = act[b].w > 0.f ? g[b].w : 0.f;
  }
}

// ---- ReLU masks: one bit per saved hidden value (round 5) ----------------------------------------------------------------------
// The backward sweep needs of a saved hidden activation only (value > 0).  The forward packs that bit for the lane's values of a
// tensor into words (first value in the highest bit of word 0; 8 sixteen-feature blocks = 32 values per word); the backward consumes
// a word from its top bit.  Two instructions per value on either side, as the compare + select on the activation itself.
__device__ __forceinline__ unsigned push_positive(unsigned acc, float v) {
  float t;
  asm("v_sub_f32 %0, 0, %1" : "=v"(t) : "v"(v));                   // sign set iff v > 0 (0 - (-0) = +0); not foldable to a negation
  return __builtin_amdgcn_alignbit(acc, __float_as_uint(t), 31);   // (acc << 1) | sign
}
template <int NB>
__device__ __forceinline__ void relu_mask_words(const v4f (&h)[NB], unsigned (&w)[(NB + 7) / 8]) {
#pragma unroll
  for (int wi = 0; wi < (NB + 7) / 8; ++wi) {
    const int b1 = NB < 8 * wi + 8 ? NB : 8 * wi + 8;
    unsigned acc = 0u;
#pragma unroll
    for (int b = 8 * wi; b < b1; ++b) {
      acc = push_positive(acc, h[b].x); acc = push_positive(acc, h[b].y);
      acc = push_positive(acc, h[b].z); acc = push_positive(acc, h[b].w);
    }
    w[wi] = acc << (32 - 4 * (b1 - 8 * wi));
  }
}
// g = top bit of w ? g : +0;  w <<= 1.  (w + w leaves the bit in the carry, the select reads it: no temporaries -- an
// extract-and-mask form let the scheduler keep a tensor's worth of extracted bits alive: 256 registers and spills.)
__device__ __forceinline__ float keep_if_msb(unsigned& w, float g) {
  asm("v_add_co_u32 %0, vcc, %0, %0\n\tv_cndmask_b32 %1, 0, %1, vcc" : "+v"(w), "+v"(g) : : "vcc");
  return g;
}
// the blocks [8 wi, 8 wi + 8) of g against word wi of a tensor's mask, for every word
template <int NB>
__device__ __forceinline__ void relu_bwd_words(v4f (&g)[NB], const unsigned* __restrict__ words) {
#pragma unroll
  for (int wi = 0; wi < (NB + 7) / 8; ++wi) {
    unsigned w = words[wi];
#pragma unroll
    for (int b = 8 * wi; b < (NB < 8 * wi + 8 ? NB : 8 * wi + 8); ++b) {
      g[b].x = keep_if_msb(w, g[b].x); g[b].y = keep_if_msb(w, g[b].y);
      g[b].z = keep_if_msb(w, g[b].z); g[b].w = keep_if_msb(w, g[b].w);
    }
  }
}

template <int NBLK>
__device__ __forceinline__ void add_blocks(v4f* __restrict__ a, const v4f* __restrict__ b) {
#pragma unroll
  for (int i = 0; i < NBLK; ++i) a[i] += b[i];
}

template <int NBLK>
__device__ __forceinline__ void copy_blocks(v4f* __restrict__ a, const v4f* __restrict__ b) {
#pragma unroll
  for (int i = 0; i < NBLK; ++i) a[i] = b[i];
}

// Sum of rows listed in perm[beg..end) (a CSR / CSC segment), NBLK blocks starting at col0.
// perm == nullptr means identity.  Rows are fetched four (two for wide rows) at a time so that
// several independent loads are in flight per lane, and added in list order (fixed summation
// order -> bitwise reproducible).  (Measured on MI355X: deeper unrolling, index prefetch with
// branch-free masked batches, and a fused two-list variant were all slower than this form.)
template <int NBLK>
__device__ __forceinline__ void segment_sum(const float* __restrict__ base, int stride, int col0,
                                            const int* __restrict__ perm, int beg, int end,
                                            v4f* __restrict__ acc) {
  constexpr int U = (NBLK <= 6) ? 4 : 2;
  const int q = (threadIdx.x & 63) >> 4;
  const float* b0 = base + col0 + 4 * q;
  int k = beg;
  for (; k + U <= end; k += U) {
    long r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) r[u] = perm ? perm[k + u] : (k + u);
    v4f t[U][NBLK];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float* p = b0 + r[u] * (long)stride;
#pragma unroll
      for (int b = 0; b < NBLK; ++b) t[u][b] = *reinterpret_cast<const v4f*>(p + 16 * b);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int b = 0; b < NBLK; ++b) acc[b] += t[u][b];
  }
  for (; k < end; ++k) {
    const long r = perm ? perm[k] : k;
    const float* p = b0 + r * (long)stride;
#pragma unroll
    for (int b = 0; b < NBLK; ++b) acc[b] += *reinterpret_cast<const v4f*>(p + 16 * b);
  }
}

// Deep variant for kernels with few feature blocks per wavefront: U rows in flight per lane, the
// gather indices of the next batch fetched while the rows of this one are in flight, slots beyond
// the segment end masked off.  Same summation order as above.
//
// `q` is the 16-byte piece of every 64-byte feature block this lane reads.  In the register layout the MFMAs
// want ("L": lane = row + 16 q) the four pieces of a block sit 16 lanes apart and neighbouring lanes read
// different rows: the texture path then looks up one cache line per lane and a gather moves ~20 B/clk/CU.
// With "Q" (lane = 4 row + q) each lane quad reads one contiguous 64-byte block: ~36-43 B/clk/CU measured on
// MI355X (tools/micro/gather_pattern.hip).  Sums taken in Q go through LDS (slot row + 16 q) to reach L.
template <int NBLK, int U>
__device__ __forceinline__ void segment_sum_deep_q(const float* __restrict__ base, int stride, int col0,
                                                   const int* __restrict__ perm, int beg, int end,
                                                   v4f* __restrict__ acc, int q) {
  const float* b0 = base + col0 + 4 * q;
  if (beg >= end) return;
  const int last = end - 1;
  int r[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int kk = (beg + u < end) ? beg + u : last;
    r[u] = perm ? perm[kk] : kk;
  }
  for (int k = beg; k < end; k += U) {
    // slots beyond the segment end are NOT loaded (lanes masked off: in layout Q a row is a lane quad, and the
    // texture path skips idle quads); they contribute zeros
    v4f t[U][NBLK];
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int b = 0; b < NBLK; ++b) t[u][b] = v4f{0.f, 0.f, 0.f, 0.f};
      if (k + u < end) {
        const float* p = b0 + (long)r[u] * stride;
#pragma unroll
        for (int b = 0; b < NBLK; ++b) t[u][b] = *reinterpret_cast<const v4f*>(p + 16 * b);
      }
    }
    if (k + U < end) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int kk = (k + U + u < end) ? k + U + u : last;
        r[u] = perm ? perm[kk] : kk;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int b = 0; b < NBLK; ++b) acc[b] += t[u][b];
    }
  }
}
template <int NBLK, int U>
__device__ __forceinline__ void segment_sum_deep(const float* __restrict__ base, int stride, int col0,
                                                 const int* __restrict__ perm, int beg, int end,
                                                 v4f* __restrict__ acc) {
  segment_sum_deep_q<NBLK, U>(base, stride, col0, perm, beg, end, acc, (threadIdx.x & 63) >> 4);
}

// Layout Q helpers: lane = 4 * (row in tile) + q.
__device__ __forceinline__ int q_row(int lane) { return lane >> 2; }
__device__ __forceinline__ int q_piece(int lane) { return lane & 3; }
__device__ __forceinline__ int q_slot(int lane) { return (lane >> 2) + 16 * (lane & 3); }   // lane of layout L holding the same data
template <int NBLK>
__device__ __forceinline__ void store_row_q(float* __restrict__ base, long row, int stride, int col0, bool valid,
                                            const v4f* __restrict__ src) {
  float* p = base + row * (long)stride + col0 + 4 * (threadIdx.x & 3);
  if (valid) {
#pragma unroll
    for (int b = 0; b < NBLK; ++b) *reinterpret_cast<v4f*>(p + 16 * b) = src[b];
  }
}

}  // namespace b3d
