// Stand-alone harness for the camera+LiDAR+radar EDGE stack (hoisted form, forward), the kernel shape under development:
//   4 wavefronts per workgroup (one per SIMD, up to 512 registers), each owning ONE 32-row tile (32 edges), 128 edges per
//   workgroup.  Every Linear is Y^T[out][row] = W[out][in] . X^T[in][row] on v_mfma_f32_32x32x16_bf16 ("bf16x6": exact
//   three-way bf16 split of both operands, six piece products, fp32 accumulation): ONE wavefront per SIMD issues the
//   32-cycle instruction back to back (tools/micro/mfma_rate.hip: the 16-cycle 16x16x32 form needs two wavefronts per SIMD:
//   24.7 vs 13.8 cycles per MFMA and SIMD with the LDS fragment reads in the loop).  The 32x32 accumulator of a layer
//   (lane = row, 16 registers = features (i & 3) + 8 (i >> 2) + 4 (lane >> 5)) is the B operand of the next layer without
//   lane movement; the weight images are stored fragment by fragment in exactly that k order.
//   Weights stream global -> LDS through a ring of 24 KB slots by LDS-DMA issued from inline asm (invisible to hipcc's
//   s_waitcnt bookkeeping) with counted vmcnt waits; one s_barrier per 8 k-steps (48 MFMAs per wavefront).
// Checks the result against a float64 CPU evaluation and times the launch.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 edge_stack.hip -o edge_stack && ./edge_stack
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <utility>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

constexpr int DE = 64, DA = 64, EH1 = 256, EH2 = 128, MH = 192, DM = 128;
constexpr int TW = 992, OA = 0, OB = 256, OF = 512, OP = 704;

// ---- weight stream geometry ---------------------------------------------------------------------------------------------
// A layer with K inputs and N outputs is (N / 32) x (K / 16) steps; a step is the three 1 KB fragments (bf16 pieces 0, 1, 2) of
// one 32 x 16 block of W: lane l = (r = l & 31, h = l >> 5) holds the 8 bf16 W[32 ob + r][32 ib + 16 s + 8 (j >> 2) + 4 h + (j & 3)],
// j = 0..7, for k-step ks = 2 ib + s.  Steps are stored in execution order; a chunk is kChunkSteps consecutive steps.
constexpr int kWaves = 4, kChunkSteps = 8, kStepBytes = 3072, kChunkBytes = kChunkSteps * kStepBytes, kSlots = 6;
constexpr int kPiecesPerWave = kChunkBytes / 1024 / kWaves;          // 6
template <int K_, int N_>
struct LY {
  static constexpr int K = K_, N = N_, KS = K / 16, OB = N / 32, STEPS = KS * OB;
  static_assert(K % 32 == 0 && N % 32 == 0 && STEPS % kChunkSteps == 0, "layer geometry");
};
template <class... Ls>
struct SeqT {
  static constexpr int NL = sizeof...(Ls);
  __host__ __device__ static constexpr int k(int li) { constexpr int a[] = {Ls::K...}; return a[li]; }
  __host__ __device__ static constexpr int n(int li) { constexpr int a[] = {Ls::N...}; return a[li]; }
  __host__ __device__ static constexpr int steps(int li) { constexpr int a[] = {Ls::STEPS...}; return a[li]; }
  __host__ __device__ static constexpr int first_step(int li) { int c = 0; for (int i = 0; i < li; ++i) c += steps(i); return c; }
  __host__ __device__ static constexpr int first_chunk(int li) { return first_step(li) / kChunkSteps; }
  __host__ __device__ static constexpr int bias_off(int li) { int c = 0; for (int i = 0; i < li; ++i) c += n(i); return c; }   // floats
  static constexpr int NSTEPS = first_step(NL), NCH = NSTEPS / kChunkSteps, NBIAS = bias_off(NL);
  static constexpr int WEIGHT_BYTES = NSTEPS * kStepBytes;
  static constexpr int BIAS_BYTES = (NBIAS * 4 + 4095) / 4096 * 4096;      // one DMA piece per wavefront and 4 KB
  static constexpr int TOTAL_BYTES = WEIGHT_BYTES + BIAS_BYTES;
};
using FwdSeq = SeqT<LY<128, 256>, LY<256, 128>, LY<128, 64>, LY<64, 192>, LY<192, 128>, LY<64, 192>, LY<192, 128>>;
static_assert(FwdSeq::NCH % kSlots == 0, "the slot of a chunk must not depend on the tile");
static_assert(FwdSeq::BIAS_BYTES == 8192, "bias DMA below moves two pieces per wavefront");
constexpr int kLdsBytes = kSlots * kChunkBytes + FwdSeq::BIAS_BYTES;

// ---- bf16x6 ---------------------------------------------------------------------------------------------------------------
struct Bf3 { bf8 p0, p1, p2; };
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
  for (int d = 0; d < 4; ++d) {
    q0[d] = __builtin_amdgcn_perm(h[2 * d + 1], h[2 * d], 0x07060302u);
    q1[d] = __builtin_amdgcn_perm(m[2 * d + 1], m[2 * d], 0x07060302u);
    q2[d] = __builtin_amdgcn_perm(l[2 * d + 1], l[2 * d], 0x07060302u);
  }
  return Bf3{__builtin_bit_cast(bf8, q0), __builtin_bit_cast(bf8, q1), __builtin_bit_cast(bf8, q2)};
}
// A 32-feature block of one row IS the accumulator vector: element 4 g + e = feature 8 g + 4 h + e (g = 0..3, e = 0..3).
typedef v16f Blk;
typedef float v8f __attribute__((ext_vector_type(8)));
__device__ __forceinline__ Blk blk_from(const v4f a, const v4f b, const v4f c, const v4f d) {
  const v8f lo = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  const v8f hi = __builtin_shufflevector(c, d, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
}
template <int G>
__device__ __forceinline__ v4f blk_part(const Blk b) { return __builtin_shufflevector(b, b, 4 * G, 4 * G + 1, 4 * G + 2, 4 * G + 3); }
__device__ __forceinline__ Blk relu16(const Blk a) {
  const Blk z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  return __builtin_elementwise_max(a, z);
}

// ---- LDS-DMA ring -----------------------------------------------------------------------------------------------------------
// One asm statement per chunk and wavefront: 6 pieces of 1 KB (64 lanes x 16 B), contiguous in global memory and in LDS.  The
// instruction offset advances the global AND the LDS address (tools/micro/dma_probe.hip).  M0 carries the LDS byte address;
// it is compiler-reserved, so it is saved and restored inside the statement.
__device__ __forceinline__ void dma6(const void* gsrc, unsigned lds_dst, unsigned voff) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
               "global_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072\n\t"
               "s_add_u32 m0, m0, 0x1000\n\tv_add_u32 %1, 0x1000, %1\n\t"
               "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep), "+v"(voff) : "s"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma2(const void* gsrc, unsigned lds_dst, unsigned voff) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(gsrc), "s"(lds_dst) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// HK::before(ci): ordinary vector-memory instructions (stores / loads that hipcc issues) between acquire<ci - 1> and
// acquire<ci>.  They are YOUNGER than the DMA pieces issued at acquire<ci - 1>: a wait that does not count them drains every DMA
// in flight and every store.  The counts must not exceed what is really issued (an over-count would let the wait return
// early): rows past the end are stored too (into the buffers' padding), so every store is unconditional.
template <class S, class HK>
struct Ring {
  const char* g;        // images: weights, then the biases
  unsigned lds0;        // byte address of the ring in the LDS address space
  int wave, lane;
  __device__ __forceinline__ void init(const void* gw, const void* lds) {
    g = (const char*)gw;
    lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)lds;
    wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    lane = threadIdx.x & 63;
  }
  __device__ __forceinline__ unsigned bias_lds() const { return lds0 + kSlots * kChunkBytes; }
  template <int CI>
  __device__ __forceinline__ void issue() {
#ifdef NO_DMA
    return;          // timing experiment: no weight stream (results are garbage)
#endif
    constexpr int SLOT = CI % kSlots;
    const unsigned woff = (unsigned)wave * (kPiecesPerWave * 1024);
    dma6(g + (size_t)CI * kChunkBytes, lds0 + SLOT * kChunkBytes + woff, woff + lane * 16);
  }
  // pieces of this wavefront that may still be in flight when chunk CI is needed: those of the chunks issued after it
  template <int CI>
  static constexpr int pending(bool more) {
    int p = 0;
    for (int c = CI + 1; c < CI + kSlots - 1; ++c)
      if (c < S::NCH || more) p += kPiecesPerWave;
    for (int j = 0; j < kSlots - 1; ++j)
      if (CI - j >= 0) p += HK::before(CI - j);          // (first tile: nothing before chunk 0; later tiles: under-counted, safe)
    return p < 63 ? p : 63;
  }
  __device__ __forceinline__ void start() {
    const unsigned woff = (unsigned)wave * 2048;
    dma2(g + S::WEIGHT_BYTES, bias_lds() + woff, woff + lane * 16);          // biases: 8 KB, older than every chunk
    issue<0>(); issue<1>(); issue<2>(); issue<3>(); issue<4>();
  }
  // chunk CI has landed for every wavefront, the slot of chunk CI - 1 is free: refill it with chunk CI + kSlots - 1
  template <int CI>
  __device__ __forceinline__ unsigned acquire(bool more) {
    constexpr int PM = pending<CI>(true), PN = pending<CI>(false);
    if constexpr (PM == PN) wait_vm<PM>();
    else { if (more) wait_vm<PM>(); else wait_vm<PN>(); }
    __builtin_amdgcn_s_barrier();
    constexpr int NXT = CI + kSlots - 1;
    if constexpr (NXT < S::NCH) issue<NXT>();
    else if (more) issue<NXT - S::NCH>();
    return lds0 + (CI % kSlots) * kChunkBytes;
  }
};

typedef __attribute__((address_space(3))) const u4v* lds_u4v_p;
typedef __attribute__((address_space(3))) const v4f* lds_v4f_p;
__device__ __forceinline__ Bf3 frag_load(unsigned addr) {          // addr: this lane's 16 bytes of piece 0 of the step
  Bf3 f;
  f.p0 = __builtin_bit_cast(bf8, *(lds_u4v_p)(size_t)addr);
  f.p1 = __builtin_bit_cast(bf8, *(lds_u4v_p)(size_t)(addr + 1024));
  f.p2 = __builtin_bit_cast(bf8, *(lds_u4v_p)(size_t)(addr + 2048));
  return f;
}
__device__ __forceinline__ v16f mfma6(const Bf3& w, const Bf3& x, v16f acc) {   // smallest terms first
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.p0, x.p2, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.p1, x.p1, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.p2, x.p0, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.p0, x.p1, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.p1, x.p0, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.p0, x.p0, acc, 0, 0, 0);
  return acc;
}

// State of a layer's step loop (everything else is compile-time).
struct StepState {
  unsigned base;     // LDS address of the current chunk
  Bf3 cur;           // fragments of the current step
  v16f acc;
};

// One step of layer LI: k-step ks of output block ob.  io[ob] holds the initial value (INIT) on entry of the block and the
// activation on exit.
template <class S, int LI, int ST, bool RELU, bool BIAS, bool INIT, class RingT>
__device__ __forceinline__ void step(RingT& ring, bool more, StepState& st, const Bf3 (&x)[S::k(LI) / 16], Blk (&io)[S::n(LI) / 32]) {
  constexpr int KS = S::k(LI) / 16, NST = S::steps(LI);
  constexpr int ob = ST / KS, ks = ST % KS;
  constexpr int GST = S::first_step(LI) + ST;                    // step of the tile
  constexpr int IN_CHUNK = GST % kChunkSteps;
  if constexpr (IN_CHUNK == 0) {
    st.base = ring.template acquire<GST / kChunkSteps>(more);
    st.cur = frag_load(st.base + ring.lane * 16);
  }
  if constexpr (ks == 0) {
    v16f a = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (BIAS) {
      const unsigned ba = ring.bias_lds() + (S::bias_off(LI) + 32 * ob + 4 * (ring.lane >> 5)) * 4;
      a = blk_from(*(lds_v4f_p)(size_t)(ba), *(lds_v4f_p)(size_t)(ba + 32), *(lds_v4f_p)(size_t)(ba + 64), *(lds_v4f_p)(size_t)(ba + 96));
    }
    if constexpr (INIT) a += io[ob];
    st.acc = a;
  }
  // The three LDS reads of the NEXT step go one each into the first three MFMA gaps of this step: issued between MFMAs a
  // read costs a few cycles of the gap; issued as a group in front of them the matrix pipe drains meanwhile
  // (tools/micro/mfma_rate.hip: 39 vs 47 cycles per MFMA for one wavefront per SIMD).
  __builtin_amdgcn_sched_barrier(0);
  Bf3 nxt = st.cur;
  constexpr bool PREFETCH = IN_CHUNK + 1 < kChunkSteps && ST + 1 < NST;
  if constexpr (PREFETCH) nxt = frag_load(st.base + (IN_CHUNK + 1) * kStepBytes + ring.lane * 16);
  st.acc = mfma6(st.cur, x[ks], st.acc);
  if constexpr (PREFETCH) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
  st.cur = nxt;
  if constexpr (ks == KS - 1) io[ob] = RELU ? relu16(st.acc) : st.acc;
}
template <class S, int LI, bool RELU, bool BIAS, bool INIT, class RingT, int... ST>
__device__ __forceinline__ void layer_impl(RingT& ring, bool more, const Bf3 (&x)[S::k(LI) / 16], Blk (&io)[S::n(LI) / 32],
                                           std::integer_sequence<int, ST...>) {
  StepState st;
  (step<S, LI, ST, RELU, BIAS, INIT>(ring, more, st, x, io), ...);
}
template <class S, int LI, bool RELU, bool BIAS, bool INIT, class RingT>
__device__ __forceinline__ void layer(RingT& ring, bool more, const Bf3 (&x)[S::k(LI) / 16], Blk (&io)[S::n(LI) / 32]) {
  layer_impl<S, LI, RELU, BIAS, INIT>(ring, more, x, io, std::make_integer_sequence<int, S::steps(LI)>{});
}

template <int NB>
__device__ __forceinline__ void split_blocks(const Blk (&a)[NB], Bf3 (&x)[2 * NB]) {
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    x[2 * b] = bf_split(blk_part<0>(a[b]), blk_part<1>(a[b]));
    x[2 * b + 1] = bf_split(blk_part<2>(a[b]), blk_part<3>(a[b]));
  }
}
// Row tables are addressed as (uniform base pointer) + (32-bit byte offset): one VGPR per row and table, and hipcc selects
// the saddr form of global_load / global_store (no 64-bit address arithmetic, no address pairs to keep alive).
template <int NB>
__device__ __forceinline__ void load_row(const float* __restrict__ base, unsigned row, int stride, int col0, Blk (&dst)[NB]) {
  const unsigned h = (threadIdx.x & 63) >> 5;
  const unsigned off = (row * (unsigned)stride + (unsigned)col0 + 4u * h) * 4u;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const char* p = reinterpret_cast<const char*>(base) + off + 128u * b;
    dst[b] = blk_from(*reinterpret_cast<const v4f*>(p), *reinterpret_cast<const v4f*>(p + 32), *reinterpret_cast<const v4f*>(p + 64),
                      *reinterpret_cast<const v4f*>(p + 96));
  }
}
template <int NB>
__device__ __forceinline__ void store_row(float* __restrict__ base, unsigned row, int stride, const Blk (&src)[NB]) {
#ifdef NO_STORE
  return;            // timing experiment
#endif
  const unsigned h = (threadIdx.x & 63) >> 5;
  const unsigned off = (row * (unsigned)stride + 4u * h) * 4u;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    char* p = reinterpret_cast<char*>(base) + off + 128u * b;
    *reinterpret_cast<v4f*>(p) = blk_part<0>(src[b]);
    *reinterpret_cast<v4f*>(p + 32) = blk_part<1>(src[b]);
    *reinterpret_cast<v4f*>(p + 64) = blk_part<2>(src[b]);
    *reinterpret_cast<v4f*>(p + 96) = blk_part<3>(src[b]);
  }
}

#ifdef STAMPS
#define STAMP(i) do { if (threadIdx.x == 0) { a.stamps[blockIdx.x * 16 + (i)] = wall_clock64(); if ((i) == 0 || (i) == 9) a.stamps[blockIdx.x * 16 + 10 + ((i) != 0)] = clock64(); } } while (0)
#else
#define STAMP(i) do {} while (0)
#endif
struct Args {
  long long* stamps;
  int E;
  const int *src, *dst;
  const float *T, *e_in, *a_in;
  float *e_out, *fut, *past, *sH1, *sH2, *sF1, *sP1;
  const void* wpack;
};

struct FwdHooks {     // stores + loads in front of the first chunk of layers 1 .. 6 (edge_fwd_kernel below)
  __host__ __device__ static constexpr int before(int ci) {
    using S = FwdSeq;
    return ci == S::first_chunk(1) ? 32 + 24 : ci == S::first_chunk(2) ? 16 + 24 : ci == S::first_chunk(3) ? 8 : ci == S::first_chunk(4) ? 24
         : ci == S::first_chunk(5) ? 16 : ci == S::first_chunk(6) ? 24 : 0;
  }
};
__global__ __launch_bounds__(kWaves * 64, 1) void edge_fwd_kernel(const Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using S = FwdSeq;
  STAMP(0);
  Ring<S, FwdHooks> ring;
  ring.init(a.wpack, smem);
  ring.start();
  const int lane = threadIdx.x & 63;
  const int ntiles = (a.E + 127) / 128;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const bool more = tile + (int)gridDim.x < ntiles;
    const unsigned row = (unsigned)tile * 128u + ring.wave * 32 + (lane & 31);
    const unsigned rc = row < (unsigned)a.E ? row : (unsigned)a.E - 1u;         // rows past the end compute on the last edge
    const unsigned s = (unsigned)a.src[rc], d = (unsigned)a.dst[rc];
    Blk ein[4];
    {
      Blk e0[2], a0[2];
      load_row<2>(a.e_in, rc, DE, 0, e0);
      load_row<2>(a.a_in, rc, DA, 0, a0);
      ein[0] = e0[0]; ein[1] = e0[1]; ein[2] = a0[0]; ein[3] = a0[1];
    }
    Blk h1[8];
    {
      Blk tb[8];
      load_row<8>(a.T, d, TW, OA, h1);
      load_row<8>(a.T, s, TW, OB, tb);
#pragma unroll
      for (int b = 0; b < 8; ++b) h1[b] += tb[b];
    }
    // ---- edge_update ----
    {
      Bf3 x0[8];
      split_blocks<4>(ein, x0);
      STAMP(1);
      layer<S, 0, true, false, true>(ring, more, x0, h1);
    }
    STAMP(2);
    store_row<8>(a.sH1, row, EH1, h1);
    Blk fi[6];
    load_row<6>(a.T, d, TW, OF, fi);
    Blk h2[4];
    {
      Bf3 x1[16];
      split_blocks<8>(h1, x1);
      layer<S, 1, true, true, false>(ring, more, x1, h2);
    }
    STAMP(3);
    store_row<4>(a.sH2, row, EH2, h2);
    Blk pi[6];
    load_row<6>(a.T, s, TW, OP, pi);
    Blk en[2];
    {
      Bf3 x2[8];
      split_blocks<4>(h2, x2);
      layer<S, 2, false, true, false>(ring, more, x2, en);
    }
    STAMP(4);
    store_row<2>(a.e_out, row, DE, en);
    Bf3 xe[4];
    split_blocks<2>(en, xe);
    // ---- create_future_msgs ----
    layer<S, 3, true, false, true>(ring, more, xe, fi);
    STAMP(5);
    store_row<6>(a.sF1, row, MH, fi);
    {
      Blk mo[4];
      Bf3 x4[12];
      split_blocks<6>(fi, x4);
      layer<S, 4, false, true, false>(ring, more, x4, mo);
      STAMP(6);
      store_row<4>(a.fut, row, DM, mo);
    }
    // ---- create_past_msgs ----
    layer<S, 5, true, false, true>(ring, more, xe, pi);
    STAMP(7);
    store_row<6>(a.sP1, row, MH, pi);
    {
      Blk mo[4];
      Bf3 x6[12];
      split_blocks<6>(pi, x6);
      layer<S, 6, false, true, false>(ring, more, x6, mo);
      STAMP(8);
      store_row<4>(a.past, row, DM, mo);
      STAMP(9);
    }
  }
}

// ---- host ---------------------------------------------------------------------------------------------------------------
struct HostLayer { int K, N; std::vector<float> w, b; bool has_bias; };

static void split3(float x, unsigned short (&p)[3]) {
  unsigned xb; memcpy(&xb, &x, 4);
  unsigned hb = xb & 0xffff0000u; float hf; memcpy(&hf, &hb, 4);
  const float r1 = x - hf;
  unsigned mb; memcpy(&mb, &r1, 4);
  unsigned mbh = mb & 0xffff0000u; float mf; memcpy(&mf, &mbh, 4);
  const float r2 = r1 - mf;
  unsigned lb; memcpy(&lb, &r2, 4);
  p[0] = (unsigned short)(xb >> 16); p[1] = (unsigned short)(mb >> 16); p[2] = (unsigned short)(lb >> 16);
}
// steps of a layer in execution order: for ob: for ks: pieces 0, 1, 2, each 64 lanes x 8 bf16
static void pack_layer(unsigned short* img, const HostLayer& L) {
  const int KS = L.K / 16, OBN = L.N / 32;
  for (int ob = 0; ob < OBN; ++ob)
    for (int ks = 0; ks < KS; ++ks) {
      unsigned short* stp = img + ((size_t)(ob * KS + ks) * kStepBytes) / 2;
      const int ib = ks / 2, s = ks % 2;
      for (int lane = 0; lane < 64; ++lane) {
        const int r = lane & 31, h = lane >> 5;
        for (int j = 0; j < 8; ++j) {
          const int col = 32 * ib + 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
          unsigned short p[3];
          split3(L.w[(size_t)(32 * ob + r) * L.K + col], p);
          for (int pc = 0; pc < 3; ++pc) stp[(size_t)pc * 512 + lane * 8 + j] = p[pc];
        }
      }
    }
}

int main(int argc, char** argv) {
  const int E = argc > 1 ? atoi(argv[1]) : 31104, N = 3000, reps = argc > 2 ? atoi(argv[2]) : 20;
  const size_t EP = (size_t)(E + 127) / 128 * 128;      // rows past the end are stored as well
  std::mt19937 rng(1234);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  using S = FwdSeq;
  HostLayer L[7] = {{128, 256}, {256, 128}, {128, 64}, {64, 192}, {192, 128}, {64, 192}, {192, 128}};
  const bool hb[7] = {false, true, true, false, true, false, true};
  for (int i = 0; i < 7; ++i) {
    L[i].has_bias = hb[i];
    L[i].w.resize((size_t)L[i].K * L[i].N);
    L[i].b.resize(L[i].N);
    const float sc = 1.45f / sqrtf((float)L[i].K);
    for (auto& v : L[i].w) v = U(rng) * sc;
    for (auto& v : L[i].b) v = U(rng) * 0.1f;
  }
  std::vector<unsigned short> img(S::TOTAL_BYTES / 2, 0);
  for (int i = 0; i < 7; ++i) pack_layer(img.data() + (size_t)S::first_step(i) * kStepBytes / 2, L[i]);
  {
    float* bias = reinterpret_cast<float*>(img.data() + S::WEIGHT_BYTES / 2);
    for (int i = 0; i < 7; ++i)
      for (int n = 0; n < L[i].N; ++n) bias[S::bias_off(i) + n] = L[i].has_bias ? L[i].b[n] : 0.f;
  }
  printf("weights: %d steps, %d chunks, %.1f KB streamed per tile, LDS %d bytes\n", S::NSTEPS, S::NCH, S::TOTAL_BYTES / 1024.0, kLdsBytes);
  std::vector<int> src(E), dst(E);
  for (int k = 0; k < E; ++k) { dst[k] = (int)((long)k * N / E); src[k] = (int)(rng() % N); }
  std::vector<float> T((size_t)N * TW), ein((size_t)E * DE), ain((size_t)E * DA);
  for (auto& v : T) v = U(rng) * 0.5f;
  for (auto& v : ein) v = U(rng);
  for (auto& v : ain) v = U(rng);
  auto dev = [&](const void* h, size_t bytes) { void* p; CHECK(hipMalloc(&p, bytes)); if (h) CHECK(hipMemcpy(p, h, bytes, hipMemcpyHostToDevice)); return p; };
  Args a;
  a.E = E;
  a.src = (int*)dev(src.data(), E * 4); a.dst = (int*)dev(dst.data(), E * 4);
  a.T = (float*)dev(T.data(), T.size() * 4); a.e_in = (float*)dev(ein.data(), ein.size() * 4); a.a_in = (float*)dev(ain.data(), ain.size() * 4);
  a.e_out = (float*)dev(nullptr, EP * DE * 4); a.fut = (float*)dev(nullptr, EP * DM * 4); a.past = (float*)dev(nullptr, EP * DM * 4);
  a.sH1 = (float*)dev(nullptr, EP * EH1 * 4); a.sH2 = (float*)dev(nullptr, EP * EH2 * 4);
  a.sF1 = (float*)dev(nullptr, EP * MH * 4); a.sP1 = (float*)dev(nullptr, EP * MH * 4);
  a.wpack = dev(img.data(), img.size() * 2);
  a.stamps = (long long*)dev(nullptr, (size_t)4096 * 16 * 8);
  const int lds = kLdsBytes;
  CHECK(hipFuncSetAttribute((const void*)edge_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  const int ntiles = (E + 127) / 128;
  const int grid = ntiles;
  hipLaunchKernelGGL(edge_fwd_kernel, dim3(grid), dim3(kWaves * 64), lds, 0, a);
  CHECK(hipDeviceSynchronize());
  // ---- check rows against float64 ----
  std::vector<float> fut((size_t)E * DM), past((size_t)E * DM), eo((size_t)E * DE), sH1((size_t)E * EH1);
  CHECK(hipMemcpy(fut.data(), a.fut, fut.size() * 4, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(past.data(), a.past, past.size() * 4, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(eo.data(), a.e_out, eo.size() * 4, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(sH1.data(), a.sH1, sH1.size() * 4, hipMemcpyDeviceToHost));
  double worst[4] = {0, 0, 0, 0}, scale[4] = {0, 0, 0, 0};
  auto lin = [&](const HostLayer& Lr, const std::vector<double>& x, const double* init, bool relu) {
    std::vector<double> y(Lr.N);
    for (int n = 0; n < Lr.N; ++n) {
      double s0 = (Lr.has_bias ? (double)Lr.b[n] : 0.0) + (init ? init[n] : 0.0);
      for (int k = 0; k < Lr.K; ++k) s0 += (double)Lr.w[(size_t)n * Lr.K + k] * x[k];
      y[n] = relu ? std::max(s0, 0.0) : s0;
    }
    return y;
  };
  std::vector<int> rows;
  for (int r = 0; r < 200; ++r) rows.push_back(r);
  for (int r = E - 300; r < E; ++r) rows.push_back(r);
  for (int r = 5000; r < 5100; ++r) rows.push_back(r);
  for (int r : rows) {
    std::vector<double> x(128), ia(256), ifu(192), ipa(192);
    for (int k = 0; k < 64; ++k) { x[k] = ein[(size_t)r * DE + k]; x[64 + k] = ain[(size_t)r * DA + k]; }
    const float* td = &T[(size_t)dst[r] * TW];
    const float* tsr = &T[(size_t)src[r] * TW];
    for (int k = 0; k < 256; ++k) ia[k] = (double)(td[OA + k] + tsr[OB + k]);      // the kernel adds them in fp32
    for (int k = 0; k < 192; ++k) { ifu[k] = td[OF + k]; ipa[k] = tsr[OP + k]; }
    auto h1 = lin(L[0], x, ia.data(), true);
    auto h2 = lin(L[1], h1, nullptr, true);
    auto en = lin(L[2], h2, nullptr, false);
    auto f1 = lin(L[3], en, ifu.data(), true);
    auto fu = lin(L[4], f1, nullptr, false);
    auto p1 = lin(L[5], en, ipa.data(), true);
    auto pa = lin(L[6], p1, nullptr, false);
    for (int k = 0; k < DM; ++k) {
      worst[0] = std::max(worst[0], std::fabs(fu[k] - (double)fut[(size_t)r * DM + k])); scale[0] = std::max(scale[0], std::fabs(fu[k]));
      worst[1] = std::max(worst[1], std::fabs(pa[k] - (double)past[(size_t)r * DM + k])); scale[1] = std::max(scale[1], std::fabs(pa[k]));
    }
    for (int k = 0; k < DE; ++k) { worst[2] = std::max(worst[2], std::fabs(en[k] - (double)eo[(size_t)r * DE + k])); scale[2] = std::max(scale[2], std::fabs(en[k])); }
    for (int k = 0; k < EH1; ++k) { worst[3] = std::max(worst[3], std::fabs(h1[k] - (double)sH1[(size_t)r * EH1 + k])); scale[3] = std::max(scale[3], std::fabs(h1[k])); }
  }
  const char* nm[4] = {"fut", "past", "e_out", "sH1"};
  bool ok = true;
  for (int i = 0; i < 4; ++i) {
    printf("%-6s max |err| / max |ref| = %.3e\n", nm[i], worst[i] / scale[i]);
    ok = ok && worst[i] / scale[i] < 2e-6;
  }
  printf(ok ? "CHECK ok\n" : "CHECK FAILED\n");
  // ---- time ----
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(edge_fwd_kernel, dim3(grid), dim3(kWaves * 64), lds, 0, a);
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(edge_fwd_kernel, dim3(grid), dim3(kWaves * 64), lds, 0, a);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double us = 1e3 * ms / reps;
  const double flops = 2.0 * 147456.0 * E;
  printf("E = %d, %d workgroups: %.2f us per launch, %.1f TFLOP/s fp32-equivalent (%.3f of 416.7)\n", E, grid, us, flops / us * 1e-6, flops / us * 1e-6 / 416.7);
#ifdef STAMPS
  {
    std::vector<long long> st((size_t)grid * 16);
    CHECK(hipMemcpy(st.data(), a.stamps, st.size() * 8, hipMemcpyDeviceToHost));
    const char* names[9] = {"prologue (gathers, split)", "L0 128>256", "L1 256>128 (+sH1 st, fi ld)", "L2 128>64 (+sH2 st, pi ld)", "L3 64>192 (+e_out st)",
                            "L4 192>128 (+sF1 st)", "L5 64>192 (+fut st)", "L6 192>128 (+sP1 st)", "past store"};
    long long t0 = st[0];
    for (int g = 0; g < grid; ++g) t0 = std::min(t0, st[(size_t)g * 16]);
    double spread = 0, endm = 0, endx = 0;
    for (int g = 0; g < grid; ++g) { spread = std::max(spread, (st[(size_t)g * 16] - t0) * 0.01); endm += (st[(size_t)g * 16 + 9] - t0) * 0.01 / grid; endx = std::max(endx, (st[(size_t)g * 16 + 9] - t0) * 0.01); }
    printf("stamps (last launch): start spread %.2f us, end mean %.2f max %.2f us\n", spread, endm, endx);
    {
      double ghz = 0;
      for (int g = 0; g < grid; ++g) ghz += (double)(st[(size_t)g * 16 + 11] - st[(size_t)g * 16 + 10]) / ((st[(size_t)g * 16 + 9] - st[(size_t)g * 16]) * 10.0) / grid;
      printf("in-kernel shader clock (s_memtime / s_memrealtime): %.3f GHz\n", ghz);
    }
    for (int i = 0; i < 9; ++i) {
      double m = 0, mx = 0;
      for (int g = 0; g < grid; ++g) { const double d = (st[(size_t)g * 16 + i + 1] - st[(size_t)g * 16 + i]) * 0.01; m += d / grid; mx = std::max(mx, d); }
      printf("   %-30s mean %6.2f  max %6.2f us\n", names[i], m, mx);
    }
  }
#endif
  return ok ? 0 : 1;
}
