// Stand-alone harness for the camera+LiDAR+radar EDGE stack (hoisted form, forward), the kernel shape under development:
//   4 wavefronts per workgroup (one per SIMD, up to 512 VGPRs), each owning TWO 16-row tiles (32 edges), 128 edges per workgroup;
//   weights (bf16x3 images, 16-row blocks) streamed global -> LDS through a 5-slot ring by LDS-DMA issued from inline asm
//   (invisible to hipcc's s_waitcnt bookkeeping) with counted vmcnt waits; one s_barrier per chunk.
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
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

constexpr int DE = 64, DA = 64, EH1 = 256, EH2 = 128, MH = 192, DM = 128;
constexpr int TW = 992, OA = 0, OB = 256, OF = 512, OP = 704;

__host__ __device__ constexpr int round_up(int a, int b) { return (a + b - 1) / b * b; }
__host__ __device__ constexpr int bf_pos(int f) { return 8 * ((f & 15) >> 2) + 4 * (f >> 4) + (f & 3); }

// ---- weight image geometry ---------------------------------------------------------------------------------------------
// layer: K inputs, N outputs, BPC 16-row blocks per chunk.  Row = [piece0: K/2 dwords][piece1][piece2][bias][pad 7], stride 3K/2 + 8.
constexpr int kRing = 5, kSlotBytes = 28672, kWaves = 4;
template <int K_, int N_, int BPC_>
struct LY {
  static constexpr int K = K_, N = N_, BPC = BPC_;
  static constexpr int STRIDE = 3 * K / 2 + 8;                                    // dwords
  static constexpr int CH_BYTES = round_up(BPC * 16 * STRIDE * 4, 1024 * kWaves);  // every wavefront moves the same number of 1 KB pieces
  static constexpr int NCH = N / 16 / BPC;
  static constexpr int PIECES = CH_BYTES / 1024 / kWaves;
  static_assert(N % (16 * BPC) == 0 && K % 32 == 0 && CH_BYTES <= kSlotBytes, "layer geometry");
};
template <class... Ls>
struct SeqT {
  static constexpr int NL = sizeof...(Ls);
  __host__ __device__ static constexpr int k(int li) { constexpr int a[] = {Ls::K...}; return a[li]; }
  __host__ __device__ static constexpr int n(int li) { constexpr int a[] = {Ls::N...}; return a[li]; }
  __host__ __device__ static constexpr int bpc(int li) { constexpr int a[] = {Ls::BPC...}; return a[li]; }
  __host__ __device__ static constexpr int stride(int li) { constexpr int a[] = {Ls::STRIDE...}; return a[li]; }
  __host__ __device__ static constexpr int chb(int li) { constexpr int a[] = {Ls::CH_BYTES...}; return a[li]; }
  __host__ __device__ static constexpr int nch(int li) { constexpr int a[] = {Ls::NCH...}; return a[li]; }
  __host__ __device__ static constexpr int pieces(int li) { constexpr int a[] = {Ls::PIECES...}; return a[li]; }
  __host__ __device__ static constexpr int first(int li) { int c = 0; for (int i = 0; i < li; ++i) c += nch(i); return c; }
  static constexpr int NCH = first(NL);
  __host__ __device__ static constexpr int layer_of(int ci) { int li = 0; while (ci >= first(li + 1)) ++li; return li; }
  __host__ __device__ static constexpr int goff(int ci) {                        // byte offset of chunk ci
    int o = 0;
    for (int c = 0; c < ci; ++c) o += chb(layer_of(c));
    return o;
  }
  static constexpr int TOTAL_BYTES = goff(NCH);
  __host__ __device__ static constexpr int cpieces(int ci) { return pieces(layer_of(ci)); }
};
using FwdSeq = SeqT<LY<128, 256, 2>, LY<256, 128, 1>, LY<128, 64, 2>, LY<64, 192, 4>, LY<192, 128, 1>, LY<64, 192, 4>, LY<192, 128, 1>>;
static_assert(FwdSeq::NCH % kRing == 0, "the slot of a chunk must not depend on the tile");

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
__device__ __forceinline__ v4f relu4(v4f a) { return v4f{fmaxf(a.x, 0.f), fmaxf(a.y, 0.f), fmaxf(a.z, 0.f), fmaxf(a.w, 0.f)}; }

// ---- LDS-DMA ring -----------------------------------------------------------------------------------------------------------
// One asm statement per chunk and wavefront: P pieces of 1 KB (64 lanes x 16 B), contiguous in global memory and in LDS.
// M0 carries the LDS byte address of the piece; it is compiler-reserved, so it is saved and restored inside the statement.
// The instruction offset advances the global AND the LDS address (tools/micro/dma_probe.hip): four pieces per M0 value.
#define DMA4 "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072\n\t"
#define DMA_BUMP "s_add_u32 m0, m0, 0x1000\n\tv_add_u32 %1, 0x1000, %1\n\t"
template <int P>
__device__ __forceinline__ void dma_pieces(const void* gsrc, unsigned lds_dst, unsigned voff) {
  unsigned keep;
  static_assert(P == 5 || P == 7, "piece counts of the sequence");
  if constexpr (P == 7)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t" DMA4 DMA_BUMP
                 "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep), "+v"(voff) : "s"(gsrc), "s"(lds_dst) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t" DMA4 DMA_BUMP "global_load_lds_dwordx4 %1, %2\n\t" "s_mov_b32 m0, %0"
                 : "=&s"(keep), "+v"(voff) : "s"(gsrc), "s"(lds_dst) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// HK::before(ci): ordinary vector-memory instructions (stores / loads that hipcc issues) between acquire<ci - 1> and
// acquire<ci>.  They are YOUNGER than the DMA pieces issued at acquire<ci - 1>: a wait that does not count them drains
// every DMA in flight and every store (measured: 6 us per such wait).  The counts must not exceed what is really issued
// (an over-count would let the wait return early): rows past the end are stored too (into the buffers' padding).
template <class S, class HK>
struct Ring {
  const char* g;        // images
  unsigned lds0;        // byte address of the ring in the LDS address space
  int wave, lane;
  __device__ __forceinline__ void init(const void* gw, const void* lds) {
    g = (const char*)gw;
    lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)lds;
    wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    lane = threadIdx.x & 63;
  }
  template <int CI>
  __device__ __forceinline__ void issue() {
#ifdef NO_DMA
    return;          // timing experiment: no weight stream (results are garbage)
#endif
    constexpr int P = S::cpieces(CI), SLOT = CI % kRing, OFF = S::goff(CI);
    const unsigned woff = (unsigned)wave * (P * 1024);
    dma_pieces<P>(g + OFF, lds0 + SLOT * kSlotBytes + woff, woff + lane * 16);
  }
  // pieces of this wavefront that may still be in flight when chunk CI is needed: those of the chunks issued after it
  template <int CI>
  static constexpr int pending(bool more) {
    int p = 0;
    for (int c = CI + 1; c < CI + kRing - 1; ++c)
      if (c < S::NCH || more) p += S::cpieces(c % S::NCH);
    for (int j = 0; j < kRing - 1; ++j)
      if (CI - j >= 0) p += HK::before(CI - j);          // (first tile: nothing before chunk 0; later tiles: under-counted, safe)
    return p < 63 ? p : 63;
  }
  __device__ __forceinline__ void start() { issue<0>(); issue<1>(); issue<2>(); issue<3>(); }
  // chunk CI has landed for every wavefront, the slot of chunk CI - 1 is free: refill it with chunk CI + 4
  template <int CI>
  __device__ __forceinline__ unsigned acquire(bool more) {
    constexpr int PM = pending<CI>(true), PN = pending<CI>(false);
    if constexpr (PM == PN) wait_vm<PM>();
    else { if (more) wait_vm<PM>(); else wait_vm<PN>(); }
    __builtin_amdgcn_s_barrier();
    constexpr int NXT = CI + kRing - 1;
    if constexpr (NXT < S::NCH) issue<NXT>();
    else if (more) issue<NXT - S::NCH>();
    return lds0 + (CI % kRing) * kSlotBytes;
  }
};

typedef __attribute__((address_space(3))) const u4v* lds_u4v_p;
typedef __attribute__((address_space(3))) const float* lds_f_p;
template <int K>
__device__ __forceinline__ Bf3 frag_load(unsigned addr) {          // addr: LDS byte address of this lane's 16 bytes of piece 0
  Bf3 f;
  f.p0 = __builtin_bit_cast(bf8, *(lds_u4v_p)(size_t)addr);
  f.p1 = __builtin_bit_cast(bf8, *(lds_u4v_p)(size_t)(addr + 2 * K));
  f.p2 = __builtin_bit_cast(bf8, *(lds_u4v_p)(size_t)(addr + 4 * K));
  return f;
}
__device__ __forceinline__ v4f mfma12(const Bf3& w, const Bf3& x, v4f acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.p0, x.p2, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.p1, x.p1, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.p2, x.p0, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.p0, x.p1, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.p1, x.p0, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.p0, x.p0, acc, 0, 0, 0);
  return acc;
}

// One chunk of layer LI: output blocks [CH * BPC, (CH + 1) * BPC) of both row tiles.  acc[t][mb] holds the initial value
// (INIT) on entry and the activation on return.
template <class S, int LI, int CH, bool RELU, bool BIAS, bool INIT, class RingT>
__device__ __forceinline__ void chunk(RingT& ring, bool more, const Bf3 (&x)[2][S::k(LI) / 32], v4f (&acc)[2][S::n(LI) / 16]) {
  constexpr int K = S::k(LI), KG = K / 32, BPC = S::bpc(LI), STRIDE = S::stride(LI);
  constexpr int CI = S::first(LI) + CH;
  const unsigned base = ring.template acquire<CI>(more);
  const int m = ring.lane & 15, q = ring.lane >> 4;
  const unsigned wrow = base + (m * STRIDE + 4 * q) * 4;
  const unsigned wbias = base + (4 * q * STRIDE + 3 * K / 2) * 4;
  auto bias = [&](int b) -> v4f {
    const unsigned a = wbias + b * 16 * STRIDE * 4;
    return v4f{*(lds_f_p)(size_t)a, *(lds_f_p)(size_t)(a + STRIDE * 4), *(lds_f_p)(size_t)(a + 2 * STRIDE * 4), *(lds_f_p)(size_t)(a + 3 * STRIDE * 4)};
  };
  auto frag = [&](int b, int c) { return frag_load<K>(wrow + (b * 16 * STRIDE + 16 * c) * 4); };
  Bf3 cur = frag(0, 0);
  v4f nb = BIAS ? bias(0) : v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int b = 0; b < BPC; ++b) {
    constexpr int dummy = 0; (void)dummy;
    const int mb = CH * BPC + b;
    v4f a0 = nb, a1 = nb;
    if constexpr (INIT) { a0 += acc[0][mb]; a1 += acc[1][mb]; }
#pragma unroll
    for (int c = 0; c < KG; ++c) {
      Bf3 nxt = cur;
      if (c + 1 < KG) nxt = frag(b, c + 1);
      else if (b + 1 < BPC) { nxt = frag(b + 1, 0); if constexpr (BIAS) nb = bias(b + 1); }
      // the LDS reads of step t + 1 stay IN FRONT of the 12 MFMAs of step t (hipcc otherwise sinks them behind the
      // tenth MFMA: a single wavefront per SIMD then waits out the LDS latency at the head of every step)
      __builtin_amdgcn_sched_barrier(0);
      a0 = mfma12(cur, x[0][c], a0);
      a1 = mfma12(cur, x[1][c], a1);
      __builtin_amdgcn_sched_barrier(0);
      cur = nxt;
    }
    acc[0][mb] = RELU ? relu4(a0) : a0;
    acc[1][mb] = RELU ? relu4(a1) : a1;
  }
}
template <class S, int LI, bool RELU, bool BIAS, bool INIT, class RingT, int... CH>
__device__ __forceinline__ void layer_impl(RingT& ring, bool more, const Bf3 (&x)[2][S::k(LI) / 32], v4f (&acc)[2][S::n(LI) / 16],
                                           std::integer_sequence<int, CH...>) {
  (chunk<S, LI, CH, RELU, BIAS, INIT>(ring, more, x, acc), ...);
}
template <class S, int LI, bool RELU, bool BIAS, bool INIT, class RingT>
__device__ __forceinline__ void layer(RingT& ring, bool more, const Bf3 (&x)[2][S::k(LI) / 32], v4f (&acc)[2][S::n(LI) / 16]) {
  layer_impl<S, LI, RELU, BIAS, INIT>(ring, more, x, acc, std::make_integer_sequence<int, S::nch(LI)>{});
}

template <int NB>
__device__ __forceinline__ void split_blocks(const v4f (&a)[2][NB], Bf3 (&x)[2][NB / 2]) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int c = 0; c < NB / 2; ++c) x[t][c] = bf_split(a[t][2 * c], a[t][2 * c + 1]);
}
// Row tables are addressed as (uniform base pointer) + (32-bit byte offset): one VGPR per row and table, and hipcc selects
// the saddr form of global_load / global_store (no 64-bit address arithmetic, no address pairs to keep alive).
template <int NB>
__device__ __forceinline__ void load_rows(const float* __restrict__ base, const unsigned (&row)[2], int stride, int col0, v4f (&dst)[2][NB]) {
  const unsigned q = (threadIdx.x & 63) >> 4;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const unsigned off = (row[t] * (unsigned)stride + (unsigned)col0 + 4u * q) * 4u;
#pragma unroll
    for (int b = 0; b < NB; ++b) dst[t][b] = *reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(base) + off + 64u * b);
  }
}
template <int NB>
__device__ __forceinline__ void store_rows(float* __restrict__ base, const unsigned (&row)[2], int stride, const v4f (&src)[2][NB]) {
  const unsigned q = (threadIdx.x & 63) >> 4;
#ifdef NO_STORE
  return;            // timing experiment
#endif
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const unsigned off = (row[t] * (unsigned)stride + 4u * q) * 4u;
#pragma unroll
    for (int b = 0; b < NB; ++b) *reinterpret_cast<v4f*>(reinterpret_cast<char*>(base) + off + 64u * b) = src[t][b];
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
    return ci == S::first(1) ? 32 + 24 : ci == S::first(2) ? 16 + 24 : ci == S::first(3) ? 8 : ci == S::first(4) ? 24
         : ci == S::first(5) ? 16 : ci == S::first(6) ? 24 : 0;
  }
};
__global__ __launch_bounds__(kWaves * 64, 1) void edge_fwd_kernel(const Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using S = FwdSeq;
  Ring<S, FwdHooks> ring;
  STAMP(0);
  ring.init(a.wpack, smem);
  ring.start();
  const int lane = threadIdx.x & 63;
  const int ntiles = (a.E + 127) / 128;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const bool more = tile + (int)gridDim.x < ntiles;
    unsigned row[2], rc[2], s[2], d[2];
    bool valid[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      row[t] = (unsigned)tile * 128u + ring.wave * 32 + t * 16 + (lane & 15);
      valid[t] = row[t] < (unsigned)a.E;
      rc[t] = valid[t] ? row[t] : (unsigned)a.E - 1u;
      s[t] = (unsigned)a.src[rc[t]];
      d[t] = (unsigned)a.dst[rc[t]];
    }
    v4f ein[2][8];
    {
      v4f e0[2][4], a0[2][4];
      load_rows<4>(a.e_in, rc, DE, 0, e0);
      load_rows<4>(a.a_in, rc, DA, 0, a0);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int b = 0; b < 4; ++b) { ein[t][b] = e0[t][b]; ein[t][4 + b] = a0[t][b]; }
    }
    v4f h1[2][16];
    {
      v4f tb[2][16];
      load_rows<16>(a.T, d, TW, OA, h1);
      load_rows<16>(a.T, s, TW, OB, tb);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int b = 0; b < 16; ++b) h1[t][b] += tb[t][b];
    }
    // ---- edge_update ----
    {
      Bf3 x0[2][4];
      split_blocks<8>(ein, x0);
      STAMP(1);
      layer<S, 0, true, false, true>(ring, more, x0, h1);
    }
    STAMP(2);
    store_rows<16>(a.sH1, row, EH1, h1);
    v4f fi[2][12];
    load_rows<12>(a.T, d, TW, OF, fi);
    v4f h2[2][8];
    {
      Bf3 x1[2][8];
      split_blocks<16>(h1, x1);
      layer<S, 1, true, true, false>(ring, more, x1, h2);
    }
    STAMP(3);
    store_rows<8>(a.sH2, row, EH2, h2);
    v4f pi[2][12];
    load_rows<12>(a.T, s, TW, OP, pi);
    v4f en[2][4];
    {
      Bf3 x2[2][4];
      split_blocks<8>(h2, x2);
      layer<S, 2, false, true, false>(ring, more, x2, en);
    }
    STAMP(4);
    store_rows<4>(a.e_out, row, DE, en);
    Bf3 xe[2][2];
    split_blocks<4>(en, xe);
    // ---- create_future_msgs ----
    layer<S, 3, true, false, true>(ring, more, xe, fi);
    STAMP(5);
    store_rows<12>(a.sF1, row, MH, fi);
    {
      v4f mo[2][8];
      Bf3 x4[2][6];
      split_blocks<12>(fi, x4);
      layer<S, 4, false, true, false>(ring, more, x4, mo);
      STAMP(6);
      store_rows<8>(a.fut, row, DM, mo);
    }
    // ---- create_past_msgs ----
    layer<S, 5, true, false, true>(ring, more, xe, pi);
    STAMP(7);
    store_rows<12>(a.sP1, row, MH, pi);
    {
      v4f mo[2][8];
      Bf3 x6[2][6];
      split_blocks<12>(pi, x6);
      layer<S, 6, false, true, false>(ring, more, x6, mo);
      STAMP(8);
      store_rows<8>(a.past, row, DM, mo);
      STAMP(9);
    }
  }
}

// ---- host ---------------------------------------------------------------------------------------------------------------
struct HostLayer { int K, N; std::vector<float> w, b; bool has_bias; };

static void pack_layer(std::vector<unsigned>& img, size_t byte_off, const HostLayer& L, int bpc, int stride, int chb) {
  // chunk ch: rows [ch * bpc * 16, ...), chb bytes each
  const int K = L.K;
  for (int r = 0; r < L.N; ++r) {
    const int ch = r / (16 * bpc), rl = r % (16 * bpc);
    unsigned* row = img.data() + (byte_off + (size_t)ch * chb) / 4 + (size_t)rl * stride;
    for (int c = 0; c < K; c += 2) {
      unsigned pc[3] = {0, 0, 0};
      for (int e = 0; e < 2; ++e) {
        const float x = L.w[(size_t)r * K + c + e];
        unsigned xb; memcpy(&xb, &x, 4);
        unsigned hb = xb & 0xffff0000u; float hf; memcpy(&hf, &hb, 4);
        const float r1 = x - hf;
        unsigned mb; memcpy(&mb, &r1, 4);
        unsigned mbh = mb & 0xffff0000u; float mf; memcpy(&mf, &mbh, 4);
        const float r2 = r1 - mf;
        unsigned lb; memcpy(&lb, &r2, 4);
        pc[0] |= (xb >> 16) << (16 * e);
        pc[1] |= (mb >> 16) << (16 * e);
        pc[2] |= (lb >> 16) << (16 * e);
      }
      const int pos = 32 * (c / 32) + bf_pos(c % 32);
      for (int p = 0; p < 3; ++p) row[p * (K / 2) + pos / 2] = pc[p];
    }
    float bv = L.has_bias ? L.b[r] : 0.f;
    memcpy(&row[3 * K / 2], &bv, 4);
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
  std::vector<unsigned> img(S::TOTAL_BYTES / 4, 0u);
  for (int i = 0; i < 7; ++i) pack_layer(img, S::goff(S::first(i)), L[i], S::bpc(i), S::stride(i), S::chb(i));
  printf("weights: %d chunks, %.1f KB streamed per tile\n", S::NCH, S::TOTAL_BYTES / 1024.0);
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
  a.wpack = dev(img.data(), img.size() * 4);
  a.stamps = (long long*)dev(nullptr, (size_t)4096 * 16 * 8);
  const int lds = kRing * kSlotBytes;
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
