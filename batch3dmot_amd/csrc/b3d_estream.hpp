// Fragment-streamed MLP stacks on 16-row wave tiles (round 3): the edge kernels of the camera+LiDAR+radar model.
//
// What round 2's kernels (b3d_dev.hpp: 8 wavefronts per workgroup, weight rows in a two-slot LDS ring filled by the
// LDS-DMA BUILTIN) lost their time on -- measured on MI355X with tools/micro/edge_stack*.hip, mfma_rate.hip, dma_probe.hip:
//   * hipcc waits vmcnt(0) for every ordinary load (and every scratch reload) while an LDS-DMA it knows about is in
//     flight: the spilled activations of a 256-register kernel were reloaded block by block behind a drain of the whole
//     next weight group.  Here the DMA is issued from inline asm (hipcc does not see it) and waited for with COUNTED
//     vmcnt immediates: the count of younger DMA pieces plus the loads / stores the kernel itself issued since (a static
//     table per kernel); rows past the end are computed and stored too (buffers are padded), so those counts are exact.
//   * one accumulator chain per wavefront: dependent v_mfma_f32_16x16x32_bf16 issue ~42 cycles apart, two wavefronts per
//     SIMD then reach 21 cycles per MFMA; with two chains (two 16-row output blocks against the same inputs) 13.8.
//   * a workgroup barrier per weight chunk between wavefronts that SHARE a SIMD: whichever the arbiter prefers (the
//     older one) finishes its chunk first and the other completes alone at a single wavefront's rate (2,300 - 2,500
//     cycles per chunk against 1,540 free-running; staggering, per-chunk flag words and priorities did not recover it).
//     Here a workgroup is 4 wavefronts, ONE PER SIMD, and TWO workgroups share a CU (80 KB of LDS each): the wavefronts
//     that meet at a barrier do not compete for a matrix pipe, the two that share one belong to different workgroups
//     and drift freely.  The price is that every CU stages the weights twice (L2 -> LDS, 2 x 0.87 MB per 128 edges).
//   * 64-bit row addresses (pairs of VGPRs per table, kept alive across the kernel): rows are addressed as uniform
//     base + 32-bit byte offset (global_load / store saddr form).
//   * row-major weight images with padded strides: the image is a sequence of 1 KB FRAGMENTS in execution order, each
//     exactly what one ds_read_b128 of a wavefront (= one LDS-DMA piece) moves: no strides, no bank conflicts.
// What bounds the kernels now: a 16-row tile needs one 1 KB LDS fragment per two MFMAs (bf16x6: three weight pieces per
// six products), which holds two wavefronts per SIMD at ~19 cycles per MFMA (tools/micro/mfma_rate.hip: 21.4 for this
// read rate); 32-row tiles would halve it but 31 k edges are only 121 rows per CU.
#pragma once
#include <utility>
#include "b3d_dev.hpp"

namespace b3d {
namespace es {

// A wavefront carries kRB 16-row blocks through the stack.  kRB = 1 (shipped): 16 rows per wavefront, 64-row tiles, TWO workgroups
// per CU (two wavefronts of different workgroups share a SIMD and cover each other's waits).  kRB = 2 (round-4 experiment, -DB3D_ES_RB=2):
// 32 rows per wavefront, 128-row tiles, ONE workgroup per CU with the whole 512-register budget per wavefront -- every weight
// fragment read from LDS feeds four MFMA chains, half the fragment reads and half the L2 -> LDS weight stream per edge, 0 spills
// (446 - 509 registers), ISA audit clean -- and 13 - 16 % SLOWER (forward 454 vs 400 us per step of six launches, backward 491 vs
// 422: profiles/r04_e_edge_rowblocks_ab.txt): with one wavefront per SIMD nothing covers the rendezvous, the LDS latencies and the
// ~880 v_accvgpr moves per tile.  The limiter of these kernels is the per-SIMD dependency chain, not LDS bandwidth.
#ifndef B3D_ES_RB
#define B3D_ES_RB 1
#endif
constexpr int kRB = B3D_ES_RB;
// TIMING ABLATIONS (tools only, results are garbage): bit 0 no refill DMA, bit 1 no fragment reads, bit 2 two of twelve MFMAs per step,
// bit 3 no operand split, bit 4 no rendezvous barrier
#ifndef B3D_ES_ABL
#define B3D_ES_ABL 0
#endif
// EXPERIMENT (round 6, -DB3D_ES_WAVES=8 [-DB3D_ES_SLOTS=6]): ONE workgroup of 8 wavefronts per CU on a 128-row tile, one ring for all of
// them -- half the LDS-DMA pieces per wavefront and half the L2 -> LDS stream per edge, and room for a deeper ring.
#ifndef B3D_ES_WAVES
#define B3D_ES_WAVES 4
#endif
#ifndef B3D_ES_SLOTS
#define B3D_ES_SLOTS 3
#endif
// EXPERIMENT (round 6, -DB3D_ES_LATE=1): the ordinary loads / stores of a layer boundary are issued BEHIND the rendezvous of the
// layer's first chunk and behind ALL refill pieces of that chunk (issued at its step 0 instead of spread over its steps): the pieces
// the next rendezvous waits for are then OLDER than the stores, so a counted wait passes them -- vmcnt is in order, and as shipped the
// wait one chunk later (vmcnt(0)) drains the boundary's stores; late, they get two chunks.
#ifndef B3D_ES_LATE
#define B3D_ES_LATE 0
#endif
constexpr bool kLate = B3D_ES_LATE != 0;
constexpr int kWaves = B3D_ES_WAVES, kTileRows = kWaves * 16 * kRB;    // rows per workgroup
constexpr int kWgPerCu = (kRB == 1 && kWaves == 4) ? 2 : 1;
constexpr int kChunkSteps = 4, kStepBytes = 6144, kChunkBytes = kChunkSteps * kStepBytes, kSlots = B3D_ES_SLOTS;
constexpr int kPiecesPerWave = kChunkBytes / 1024 / kWaves;  // 6

// ---- weight stream geometry (host packer: b3d_prep.hip pack_frag_kernel) -------------------------------------------------
// A layer with K inputs and N outputs is (N / 32) x (K / 32) steps; a step is TWO 16 x 32 blocks of W (output blocks 2 p and
// 2 p + 1 against the same 32 inputs), each as three 1 KB fragments (bf16 pieces 0, 1, 2 of the exact split w = w0 + w1 + w2):
// lane l = (m = l & 15, q = l >> 4) holds the 8 bf16 W[16 ob + m][32 c + 16 (j >> 2) + 4 q + (j & 3)], j = 0..7 -- the k order in
// which a wavefront holds the previous layer's accumulator (b3d_dev.hpp bf_pos).  Steps are stored in execution order
// (for p: for c); a chunk is kChunkSteps consecutive steps of the kernel's whole sequence.  The biases of all layers follow
// the steps as fp32 (zeros for layers without one), padded to 4 KB.
template <int K_, int N_>
struct LY {
  static constexpr int K = K_, N = N_, KS = K / 32, OB = N / 32, STEPS = KS * OB;
  static_assert(K % 32 == 0 && N % 32 == 0, "layer widths in multiples of 32");
};
template <class... Ls>
struct Seq {
  static constexpr int NL = sizeof...(Ls);
  __host__ __device__ static constexpr int k(int li) { constexpr int a[] = {Ls::K...}; return a[li]; }
  __host__ __device__ static constexpr int n(int li) { constexpr int a[] = {Ls::N...}; return a[li]; }
  __host__ __device__ static constexpr int steps(int li) { constexpr int a[] = {Ls::STEPS...}; return a[li]; }
  __host__ __device__ static constexpr int first_step(int li) { int c = 0; for (int i = 0; i < li; ++i) c += steps(i); return c; }
  __host__ __device__ static constexpr int bias_off(int li) { int c = 0; for (int i = 0; i < li; ++i) c += n(i); return c; }   // floats
  static constexpr int NSTEPS = first_step(NL), NCH = NSTEPS / kChunkSteps, NBIAS = bias_off(NL);
  static_assert(NSTEPS % kChunkSteps == 0 && NCH % kSlots == 0, "whole chunks; the slot of a chunk must not depend on the tile");
  // the hook tables (b3d_edge2.hpp) place a layer's loads / stores in front of ITS first chunk: every layer starts a chunk
  __host__ __device__ static constexpr bool layers_start_chunks() {
    for (int i = 0; i < NL; ++i)
      if (first_step(i) % kChunkSteps != 0) return false;
    return true;
  }
  static_assert(layers_start_chunks(), "a layer boundary inside a chunk: FwdHooks / BwdHooks::before(first_chunk(li)) would be mis-attributed");
  // chunk that holds the first step of layer li (hook tables)
  __host__ __device__ static constexpr int first_chunk(int li) { return first_step(li) / kChunkSteps; }
  static constexpr int WEIGHT_BYTES = NSTEPS * kStepBytes;
  static constexpr int BIAS_BYTES = (NBIAS * 4 + 4095) / 4096 * 4096;
  static constexpr int TOTAL_BYTES = WEIGHT_BYTES + BIAS_BYTES;
  static constexpr int TOTAL_FLOATS = TOTAL_BYTES / 4;
  static constexpr int LDS_BYTES = kSlots * kChunkBytes + BIAS_BYTES;
  static_assert(BIAS_BYTES == 4096 || BIAS_BYTES == 8192, "bias DMA: one or two pieces per wavefront");
  static_assert(LDS_BYTES <= (kWgPerCu == 2 ? 80 : 160) * 1024, "two workgroups per CU (kRB = 1)");
};

// ---- LDS-DMA from inline asm ---------------------------------------------------------------------------------------------
// N pieces of 1 KB (64 lanes x 16 B), contiguous in global memory and in LDS.  The instruction offset advances the global AND
// the LDS address (tools/micro/dma_probe.hip).  M0 carries the LDS byte address; it is compiler-reserved, so it is saved and
// restored inside the statement (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void dma1(const void* gsrc, unsigned lds_dst, unsigned voff) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma2(const void* gsrc, unsigned lds_dst, unsigned voff) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(gsrc), "s"(lds_dst) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ---- the ring --------------------------------------------------------------------------------------------------------------
// RENDEZVOUS.  One s_barrier in front of step 0 of every chunk C, with the meaning "every wavefront's pieces of chunk C + 1 have
// landed (each waited for its own with a counted vmcnt in front of the barrier) and every wavefront is done with chunk C - 1":
// chunk C + 1 is known to be complete a whole chunk before it is needed, so the fragments of its first step are fetched
// during the last step of chunk C like any others -- no LDS latency is exposed behind the barrier -- and the slot of chunk
// C - 1 is refilled with chunk C + 2, a share of the pieces in front of every step of chunk C.
//
// HK::before(c): ordinary vector-memory instructions (stores / loads that hipcc issues) in front of step 0 of chunk c, i.e.
// between rendezvous<c - 1> and rendezvous<c>.  They are YOUNGER than the DMA pieces of chunk c + 1 (issued during chunk c - 1):
// a wait that does not count them drains every store (measured: microseconds per layer boundary).  The counts must not
// exceed what is really issued -- an over-count would let the wait return before the pieces have landed.
template <class S, class HK>
struct Ring {
  const char* g;        // images: steps, then the biases
  unsigned lds0;        // byte address of the ring in the LDS address space
  int wave, lane;
  bool first;           // rendezvous<0> of the first tile is Ring::start's barrier
  __device__ __forceinline__ void init(const void* gw, const void* lds) {
    g = (const char*)gw;
    lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)lds;
    wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    lane = threadIdx.x & 63;
    first = true;
  }
  __device__ __forceinline__ unsigned bias_lds() const { return lds0 + kSlots * kChunkBytes; }
  // the pieces [J0, J1) of this wavefront's share of chunk CI
  template <int CI, int J0, int J1>
  __device__ __forceinline__ void issue_pieces() {
    constexpr int SLOT = CI % kSlots;
    if constexpr (J1 > J0) {
      const unsigned woff = (unsigned)wave * (kPiecesPerWave * 1024) + J0 * 1024;
      if constexpr (J1 - J0 == 1) dma1(g + (size_t)CI * kChunkBytes, lds0 + SLOT * kChunkBytes + woff, woff + lane * 16);
      else if constexpr (J1 - J0 == 2) dma2(g + (size_t)CI * kChunkBytes, lds0 + SLOT * kChunkBytes + woff, woff + lane * 16);
      else { issue_pieces<CI, J0, J0 + 2>(); issue_pieces<CI, J0 + 2, J1>(); }
    }
  }
  // Pieces of this wavefront younger than its pieces of chunk C + 1 when rendezvous<C> waits: none (chunk C + 2 is issued behind the
  // barrier) -- only the ordinary loads / stores in front of chunk C.
  template <int C>
  static constexpr int pending() {
    // kLate: the boundary operations of chunk C - 1 (issued behind that chunk's rendezvous and behind the pieces of chunk C + 1)
    const int p = kLate ? (C >= 1 ? HK::before(C - 1) : 0) : HK::before(C);
    return p < 63 ? p : 63;
  }
  // kLate: does chunk C carry boundary operations (then all its refill pieces are issued at step 0, in front of them)?
  template <int C>
  static constexpr bool late_chunk() { return kLate && HK::before(C) > 0; }
  // Stream start (once per kernel): the biases and chunks 0, 1 in flight, then complete for everybody.
  __device__ __forceinline__ void start() {
    if constexpr (S::BIAS_BYTES / kWaves == 2048) {
      const unsigned woff = (unsigned)wave * 2048;
      dma2(g + S::WEIGHT_BYTES, bias_lds() + woff, woff + lane * 16);
    } else if constexpr (S::BIAS_BYTES / kWaves == 1024) {
      const unsigned woff = (unsigned)wave * 1024;
      dma1(g + S::WEIGHT_BYTES, bias_lds() + woff, woff + lane * 16);
    } else {                                    // 4 KB over 8 wavefronts: the first four
      static_assert(S::BIAS_BYTES / kWaves == 512, "bias DMA");
      if (wave < 4) {
        const unsigned woff = (unsigned)wave * 1024;
        dma1(g + S::WEIGHT_BYTES, bias_lds() + woff, woff + lane * 16);
      }
    }
    issue_pieces<0, 0, kPiecesPerWave>();
    issue_pieces<1, 0, kPiecesPerWave>();
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
  }
  template <int C>
  __device__ __forceinline__ unsigned slot_addr() const { return lds0 + (C % kSlots) * kChunkBytes; }
  template <int C>
  __device__ __forceinline__ void rendezvous() {
    if constexpr (C == 0) { if (first) { first = false; return; } }
    wait_vm<pending<C>()>();
    if constexpr ((B3D_ES_ABL & 16) == 0) __builtin_amdgcn_s_barrier();
  }
  template <int C>
  __device__ __forceinline__ void refill_all(bool more) {
    constexpr int NXT = C + kSlots - 1;
    if constexpr ((B3D_ES_ABL & 1) != 0) return;
    if constexpr (NXT < S::NCH) issue_pieces<NXT, 0, kPiecesPerWave>();
    else if (more) issue_pieces<NXT - S::NCH, 0, kPiecesPerWave>();
  }
  // in front of step J of chunk C: this step's share of the pieces of chunk C + 2
  template <int C, int J>
  __device__ __forceinline__ void refill(bool more) {
    constexpr int NXT = C + kSlots - 1, J0 = J * kPiecesPerWave / kChunkSteps, J1 = (J + 1) * kPiecesPerWave / kChunkSteps;
    if constexpr ((B3D_ES_ABL & 1) != 0) return;
    if constexpr (NXT < S::NCH) issue_pieces<NXT, J0, J1>();
    else if (more) issue_pieces<NXT - S::NCH, J0, J1>();
  }
};

// ---- steps ---------------------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) const u4v* lds_u4v_p;
typedef __attribute__((address_space(3))) const v4f* lds_v4f_p;
typedef unsigned u2w __attribute__((ext_vector_type(2)));
#ifndef B3D_ES_F16
#define B3D_ES_F16 0
#endif
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ Bf3 frag_load(unsigned addr) {          // addr: this lane's 16 bytes of piece 0 of a 16 x 32 block
  Bf3 f;
  f.p0 = __builtin_bit_cast(bf8, *(lds_u4v_p)(size_t)addr);
  f.p1 = __builtin_bit_cast(bf8, *(lds_u4v_p)(size_t)(addr + 1024));
#if B3D_ES_F16
  f.p2 = f.p1;
#else
  f.p2 = __builtin_bit_cast(bf8, *(lds_u4v_p)(size_t)(addr + 2048));
#endif
  return f;
}
// EXPERIMENT (B3D_ES_F16): two fp16 pieces (round-to-nearest value + the fp16 of its exact residual), three products
__device__ __forceinline__ Bf3 f16_split(const v4f a, const v4f b) {
  const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  u4v q0, q1;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const f2v v = {x[2 * d], x[2 * d + 1]};
    const h2v hi = __builtin_convertvector(v, h2v);
    const f2v r = v - __builtin_convertvector(hi, f2v);
    const h2v lo = __builtin_convertvector(r, h2v);
    q0[d] = __builtin_bit_cast(unsigned, hi);
    q1[d] = __builtin_bit_cast(unsigned, lo);
  }
  return Bf3{__builtin_bit_cast(bf8, q0), __builtin_bit_cast(bf8, q1), __builtin_bit_cast(bf8, q1)};
}
__device__ __forceinline__ void frag_load2(unsigned addr, Bf3& f0, Bf3& f1) { f0 = frag_load(addr); f1 = frag_load(addr + 3072); }
// 2 kRB independent chains (output blocks 2 p, 2 p + 1 x the wavefront's row blocks), interleaved; smallest terms first (as
// bf_mfma6)
template <int KS>
__device__ __forceinline__ void mfma_step(const Bf3& w0, const Bf3& w1, const Bf3 (&x)[kRB][KS], int ks, v4f (&a0)[kRB], v4f (&a1)[kRB]) {
#define B3D_ES_PROD(WP, XP)                                                                                   \
  _Pragma("unroll") for (int rb = 0; rb < kRB; ++rb) {                                                         \
    a0[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0.WP, x[rb][ks].XP, a0[rb], 0, 0, 0);                   \
    a1[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1.WP, x[rb][ks].XP, a1[rb], 0, 0, 0);                   \
  }
#if B3D_ES_F16
#define B3D_ES_PRODH(WP, XP)                                                                                  \
  _Pragma("unroll") for (int rb = 0; rb < kRB; ++rb) {                                                         \
    a0[rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8v, w0.WP), __builtin_bit_cast(h8v, x[rb][ks].XP), a0[rb], 0, 0, 0);  \
    a1[rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8v, w1.WP), __builtin_bit_cast(h8v, x[rb][ks].XP), a1[rb], 0, 0, 0);  \
  }
  B3D_ES_PRODH(p0, p1) B3D_ES_PRODH(p1, p0) B3D_ES_PRODH(p0, p0)
#undef B3D_ES_PRODH
#elif (B3D_ES_ABL & 4)
  B3D_ES_PROD(p0, p0)
#else
  B3D_ES_PROD(p0, p2) B3D_ES_PROD(p1, p1) B3D_ES_PROD(p2, p0) B3D_ES_PROD(p0, p1) B3D_ES_PROD(p1, p0) B3D_ES_PROD(p0, p0)
#endif
#undef B3D_ES_PROD
}

// State of the step loop, carried across layers (everything else is compile-time).
struct StepState {
  unsigned base;     // LDS address of the current chunk
  Bf3 cur0, cur1;    // fragments of the current step (output blocks 2 p, 2 p + 1)
  v4f acc0[kRB], acc1[kRB];
};

// One step of layer LI: 32 inputs (group ks) against the output blocks 2 ob, 2 ob + 1.  io[b] holds the initial value (INIT) on
// entry of a block and the activation on exit.
struct NoHook { __device__ __forceinline__ void operator()() const {} };
template <class S, int LI, int ST, bool RELU, bool BIAS, bool INIT, class RingT, class Hook>
__device__ __forceinline__ void step(RingT& ring, bool more, StepState& st, const Bf3 (&x)[kRB][S::k(LI) / 32], v4f (&io)[kRB][S::n(LI) / 16],
                                     Hook& hook) {
  constexpr int KS = S::k(LI) / 32;
  constexpr int ob = ST / KS, ks = ST % KS;
  constexpr int GST = S::first_step(LI) + ST;                    // step of the tile
  constexpr int IN_CHUNK = GST % kChunkSteps, CJ = GST / kChunkSteps;
  if constexpr (IN_CHUNK == 0) {
    ring.template rendezvous<CJ>();
    st.base = ring.template slot_addr<CJ>();       // (its first fragments were fetched during the previous step)
  }
  if constexpr (RingT::template late_chunk<CJ>()) {
    static_assert(ST < kChunkSteps, "boundary operations belong to the first chunk of a layer");
    if constexpr (ST == 0) { ring.template refill_all<CJ>(more); hook(); }
  } else {
    ring.template refill<CJ, IN_CHUNK>(more);
  }
  if constexpr (ks == 0) {
    v4f a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (BIAS) {
      const unsigned ba = ring.bias_lds() + (S::bias_off(LI) + 32 * ob + 4 * (ring.lane >> 4)) * 4;
      a0 = *(lds_v4f_p)(size_t)ba;
      a1 = *(lds_v4f_p)(size_t)(ba + 64);
    }
#pragma unroll
    for (int rb = 0; rb < kRB; ++rb) {
      st.acc0[rb] = a0; st.acc1[rb] = a1;
      if constexpr (INIT) { st.acc0[rb] += io[rb][2 * ob]; st.acc1[rb] += io[rb][2 * ob + 1]; }
    }
  }
  // the LDS reads of the NEXT step are issued in front of this step's MFMAs (hipcc otherwise sinks them next to their use)
  Bf3 n0 = st.cur0, n1 = st.cur1;
  if constexpr ((B3D_ES_ABL & 2) != 0) {}
  else if constexpr (IN_CHUNK + 1 < kChunkSteps) frag_load2(st.base + (IN_CHUNK + 1) * kStepBytes + ring.lane * 16, n0, n1);
  else if constexpr (CJ + 1 < S::NCH) frag_load2(ring.template slot_addr<CJ + 1>() + ring.lane * 16, n0, n1);
  else { if (more) frag_load2(ring.template slot_addr<0>() + ring.lane * 16, n0, n1); }
  __builtin_amdgcn_sched_barrier(0);
  mfma_step<KS>(st.cur0, st.cur1, x, ks, st.acc0, st.acc1);
  __builtin_amdgcn_sched_barrier(0);
  st.cur0 = n0; st.cur1 = n1;
  if constexpr (ks == KS - 1) {
#pragma unroll
    for (int rb = 0; rb < kRB; ++rb) {
      io[rb][2 * ob] = RELU ? relu4(st.acc0[rb]) : st.acc0[rb];
      io[rb][2 * ob + 1] = RELU ? relu4(st.acc1[rb]) : st.acc1[rb];
    }
  }
}
template <class S, int LI, bool RELU, bool BIAS, bool INIT, class RingT, class Hook, int... ST>
__device__ __forceinline__ void layer_impl(RingT& ring, bool more, const Bf3 (&x)[kRB][S::k(LI) / 32], v4f (&io)[kRB][S::n(LI) / 16],
                                           StepState& st, Hook& hook, std::integer_sequence<int, ST...>) {
  (step<S, LI, ST, RELU, BIAS, INIT>(ring, more, st, x, io, hook), ...);
}
// io = act(W . x (+ b) (+ io)) for every row block of the wavefront.  `hook`: the ordinary loads / stores of the layer boundary in front
// of this layer (HK::before(first_chunk(LI)) of them): issued here, in front of the layer -- or, kLate, inside its first step.
template <class S, int LI, bool RELU, bool BIAS, bool INIT, class RingT, class Hook = NoHook>
__device__ __forceinline__ void layer(RingT& ring, bool more, StepState& st, const Bf3 (&x)[kRB][S::k(LI) / 32], v4f (&io)[kRB][S::n(LI) / 16],
                                      Hook hook = Hook{}) {
  if constexpr (!RingT::template late_chunk<S::first_chunk(LI)>()) hook();
  layer_impl<S, LI, RELU, BIAS, INIT>(ring, more, x, io, st, hook, std::make_integer_sequence<int, S::steps(LI)>{});
}

template <int NB>
__device__ __forceinline__ void split_blocks(const v4f (&a)[kRB][NB], Bf3 (&x)[kRB][NB / 2]) {
#pragma unroll
  for (int rb = 0; rb < kRB; ++rb)
#pragma unroll
    for (int c = 0; c < NB / 2; ++c) {
#if B3D_ES_F16
      x[rb][c] = f16_split(a[rb][2 * c], a[rb][2 * c + 1]);
#elif (B3D_ES_ABL & 8)
      x[rb][c] = Bf3{__builtin_bit_cast(bf8, u4v{__builtin_bit_cast(unsigned, a[rb][2 * c].x), __builtin_bit_cast(unsigned, a[rb][2 * c].y), __builtin_bit_cast(unsigned, a[rb][2 * c].z), __builtin_bit_cast(unsigned, a[rb][2 * c].w)}),
                     __builtin_bit_cast(bf8, u4v{__builtin_bit_cast(unsigned, a[rb][2 * c + 1].x), __builtin_bit_cast(unsigned, a[rb][2 * c + 1].y), __builtin_bit_cast(unsigned, a[rb][2 * c + 1].z), __builtin_bit_cast(unsigned, a[rb][2 * c + 1].w)}),
                     __builtin_bit_cast(bf8, u4v{__builtin_bit_cast(unsigned, a[rb][2 * c].x), __builtin_bit_cast(unsigned, a[rb][2 * c + 1].y), __builtin_bit_cast(unsigned, a[rb][2 * c].z), __builtin_bit_cast(unsigned, a[rb][2 * c + 1].w)})};
#else
      x[rb][c] = bf_split(a[rb][2 * c], a[rb][2 * c + 1]);
#endif
    }
}
// Row tables are addressed as (uniform base pointer) + (32-bit byte offset): one VGPR per row and table, and hipcc selects
// the saddr form of global_load / global_store (no 64-bit address arithmetic, no address pairs to keep alive).  Unconditional.
template <int NB>
__device__ __forceinline__ void load_row(const float* __restrict__ base, unsigned row, int stride, int col0, v4f (&dst)[NB]) {
  const unsigned q = (threadIdx.x & 63) >> 4;
  const unsigned off = (row * (unsigned)stride + (unsigned)col0 + 4u * q) * 4u;
#pragma unroll
  for (int b = 0; b < NB; ++b) dst[b] = *reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(base) + off + 64u * b);
}
template <int NB>
__device__ __forceinline__ void store_row(float* __restrict__ base, unsigned row, int stride, const v4f (&src)[NB]) {
  const unsigned q = (threadIdx.x & 63) >> 4;
  const unsigned off = (row * (unsigned)stride + 4u * q) * 4u;
#pragma unroll
  for (int b = 0; b < NB; ++b) *reinterpret_cast<v4f*>(reinterpret_cast<char*>(base) + off + 64u * b) = src[b];
}
template <int NB>
__device__ __forceinline__ void relu_bwd_blocks(v4f (&g)[kRB][NB], const v4f (&act)[kRB][NB]) {
#pragma unroll
  for (int rb = 0; rb < kRB; ++rb)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      g[rb][b].x = act[rb][b].x > 0.f ? g[rb][b].x : 0.f;
      g[rb][b].y = act[rb][b].y > 0.f ? g[rb][b].y : 0.f;
      g[rb][b].z = act[rb][b].z > 0.f ? g[rb][b].z : 0.f;
      g[rb][b].w = act[rb][b].w > 0.f ? g[rb][b].w : 0.f;
    }
}
// ---- ReLU masks (round 5) ---------------------------------------------------------------------------------------------------
// The backward sweep needs of the saved hidden activations only their SIGN pattern (relu'); reading sH1 | sH2 | sF1 | sP1 back
// cost 3 KB per edge and layer and up to 64 registers per wavefront.  The forward writes, next to the activations (the weight
// gradient still consumes those), one bit per value: 128 bytes per edge in two planes of 16 bytes per lane (row m, quarter q),
// both addressed like a 16-float row table ((row * 4 + q) * 16 bytes: no new loop-invariant address register) --
//   plane A: words 0..1 sH1 (16 blocks), word 2 sH2 (8 blocks);   plane B: words 0..1 sF1 (12 blocks), words 2..3 sP1 (12 blocks);
// bit 31 - (4 (b & 7) + j) of word b / 8 <-> component j of the lane's block b, set iff the value is > 0 (the test the backward
// made on the activation itself: -0 and 0 are not).
constexpr int kMaskFloatsPerRow = 32;                       // both planes
struct MaskPlanes { unsigned* a; unsigned* b; };
struct MaskPlanesC { const unsigned* a; const unsigned* b; };
// words [WORD0, WORD0 + ceil(NB / 8)) of the lane's 16 bytes in `plane`
template <int NB, int WORD0>
__device__ __forceinline__ void store_masks(unsigned* __restrict__ plane, const unsigned (&row)[kRB], const v4f (&h)[kRB][NB]) {
  static_assert((NB + 7) / 8 <= 2 && WORD0 + (NB + 7) / 8 <= 4, "one or two words per tensor");
  const unsigned q = (threadIdx.x & 63) >> 4;
#pragma unroll
  for (int rb = 0; rb < kRB; ++rb) {
    unsigned w[(NB + 7) / 8];
    relu_mask_words<NB>(h[rb], w);
    char* p = reinterpret_cast<char*>(plane) + (row[rb] * 16u + 4u * q) * 4u + WORD0 * 4;
    if constexpr ((NB + 7) / 8 == 2) *reinterpret_cast<u2w*>(p) = u2w{w[0], w[1]};
    else *reinterpret_cast<unsigned*>(p) = w[0];
  }
}
__device__ __forceinline__ void load_mask_plane(const unsigned* __restrict__ plane, const unsigned (&row)[kRB], u4v (&mk)[kRB]) {
  const unsigned q = (threadIdx.x & 63) >> 4;
#pragma unroll
  for (int rb = 0; rb < kRB; ++rb) mk[rb] = *reinterpret_cast<const u4v*>(reinterpret_cast<const char*>(plane) + (row[rb] * 16u + 4u * q) * 4u);
}
// g = mask bit ? g : +0 (what relu_bwd_blocks computes from the activation): b3d_dev.hpp relu_bwd_words on words [WORD0, ...)
template <int NB, int WORD0>
__device__ __forceinline__ void relu_bwd_mask(v4f (&g)[kRB][NB], const u4v (&mk)[kRB]) {
#pragma unroll
  for (int rb = 0; rb < kRB; ++rb) {
    unsigned words[(NB + 7) / 8];
#pragma unroll
    for (int wi = 0; wi < (NB + 7) / 8; ++wi) words[wi] = mk[rb][WORD0 + wi];
    relu_bwd_words<NB>(g[rb], words);
  }
}
// ---- per-destination sums inside the wavefront (round 6) -------------------------------------------------------------------------
// The 16 rows of a block sit in the 16 lanes of a DPP row (lane = 16 q + m): with the edges grouped by destination a run of equal
// keys is contiguous in m, and a Hillis-Steele pass of row_shr 1, 2, 4, 8 leaves in every lane the sum of its run up to itself --
// a lane adds its left neighbour at distance s iff that neighbour carries the same key (sorted keys: then everything between does).
// The run's LAST lane holds the whole run.  Fixed order: the result depends on the run's position in the block only.
constexpr int kPastDumpRows = 1024;
template <int SH>
__device__ __forceinline__ int dpp_row_shr_i(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x110 + SH, 0xf, 0xf, false); }
template <int SH>
__device__ __forceinline__ float dpp_row_shr_f0(float v) {        // 0 where the row has no lane m - SH
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x110 + SH, 0xf, 0xf, true));
}
template <int NB, int SH>
__device__ __forceinline__ void run_scan_step(v4f (&v)[NB], int key) {
  const bool same = dpp_row_shr_i<SH>(key, -1) == key;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const float ux = dpp_row_shr_f0<SH>(v[b].x), uy = dpp_row_shr_f0<SH>(v[b].y), uz = dpp_row_shr_f0<SH>(v[b].z), uw = dpp_row_shr_f0<SH>(v[b].w);
    v[b].x = same ? v[b].x + ux : v[b].x; v[b].y = same ? v[b].y + uy : v[b].y;
    v[b].z = same ? v[b].z + uz : v[b].z; v[b].w = same ? v[b].w + uw : v[b].w;
  }
}
// v: the block's rows (lane m = row m).  Returns true in the lanes that hold a whole run (the run's last row); v is summed in place.
template <int NB>
__device__ __forceinline__ bool run_sums(v4f (&v)[NB], int key) {
  run_scan_step<NB, 1>(v, key);
  run_scan_step<NB, 2>(v, key);
  run_scan_step<NB, 4>(v, key);
  run_scan_step<NB, 8>(v, key);
  return __builtin_amdgcn_update_dpp(-1, key, 0x100 + 1, 0xf, 0xf, false) != key;     // row_shl:1: the key of lane m + 1 (-1 behind lane 15)
}

// the same row table access for every row block of the wavefront
template <int NB>
__device__ __forceinline__ void load_rows(const float* __restrict__ base, const unsigned (&row)[kRB], int stride, int col0, v4f (&dst)[kRB][NB]) {
#pragma unroll
  for (int rb = 0; rb < kRB; ++rb) load_row<NB>(base, row[rb], stride, col0, dst[rb]);
}
template <int NB>
__device__ __forceinline__ void store_rows(float* __restrict__ base, const unsigned (&row)[kRB], int stride, const v4f (&src)[kRB][NB]) {
#pragma unroll
  for (int rb = 0; rb < kRB; ++rb) store_row<NB>(base, row[rb], stride, src[rb]);
}

}  // namespace es
}  // namespace b3d
