// Fully connected heads of the frozen point encoders (reference batch_3dmot/models/pointnet.py:46-48
// STN3d fc1-bn4-relu, fc2-bn5-relu, fc3; pointnet.py:188-192 and radarnet.py:60-64 forward_feat: fc1-bn-relu,
// fc2-dropout-bn-relu) as a chain of ONE kernel per Linear:
//
//     y = mask * ( act_in(x) . W^T + b )          act_in(x) = relu(x * in_scale + in_shift)  (the producer's BatchNorm + ReLU,
//                                                  applied while the tile is staged; identity for the first Linear)
//
// with the per-column sum and squared deviations of y accumulated on the way out, from which the LAST workgroup to finish forms
// this layer's BatchNorm affine (batch statistics in train mode, running statistics in eval mode) and updates the running
// statistics as nn.BatchNorm1d does.  BatchNorm, ReLU and Dropout never run as kernels of their own, the normalised
// activations never exist in memory; a final elementwise kernel (b3d_affine_relu) materialises the last activation.
// Products are bf16x6 (below): fp32-class accuracy, NOT bitwise an fp32 fmaf chain, and a +-inf input gives NaN (inf - inf in the
// exact split) where torch's Linear gives +-inf.  Statistics: per-tile (sum, M2 about the tile mean) in a slab, combined in tile
// order in float64 (Chan): bitwise reproducible, and as well conditioned as torch's Welford for badly centred activations.
#include "b3d_common.hpp"
#include "b3d_launch.hpp"
#include "b3d_dev.hpp"

namespace b3d {
namespace {

// 64 x 64 tiles, EIGHT wavefronts (two per SIMD), a 32 x 16 block each: a 2,100 x 512 layer is 264 workgroups, about one per CU, and
// a single wavefront per SIMD ran at a third of the MFMA rate (LDS and barrier waits with nobody to cover them); 32 x 32 tiles
// (four workgroups per CU) doubled the operand traffic and were slower still.
//
// Products are bf16x6 (b3d_dev.hpp): both operands are split exactly into three bf16 pieces WHILE THEY ARE STAGED -- the LDS tiles
// are three [row][k] bf16 images each -- and a 32-wide k group of a 16 x 16 block is six v_mfma_f32_16x16x32_bf16 (96 cycles)
// instead of eight v_mfma_f32_16x16x4_f32 (256 cycles); 64 k per barrier instead of 32.  The exact-fp32 form of this kernel sat
// at 19 % MFMA-busy and 52 us for [1,500 x 1,024] . [1,024 x 512]: one barrier per 16 MFMAs of a wavefront.
constexpr int kTM = 64, kTN = 64, kFcThreads = 512;
// TK: k per barrier.  64: 110 KB of LDS, one workgroup per CU -- the form of every launch that fits the chip in one round; 32: 61 KB, two
// workgroups per CU -- for the launches with more tiles than CUs (a 2,100 x 512 layer is 264 tiles: at one workgroup per CU the last 8
// ran as a second round, 53 us for a 22 us workgroup).
template <int TK>
struct FcGeo {
  static constexpr int kTK = TK;
  static constexpr int kTPR = TK / 4, kRPP = kFcThreads / kTPR;   // staging: threads per tile row, rows per pass
  // dwords per tile row.  Round 6: 40 (24) instead of TK / 2 + 4 = 36 (20).  A ds_read_b128 is served in lane groups
  // {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS table), not in runs of 16 lanes: with lane = (row li, k part
  // lk = lane / 16) a group mixes rows of two k parts, and pitches of 36 / 20 dwords put two of its 16-byte slots on the same banks
  // (PMC: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.37 / 0.50); 40 / 24 are conflict-free for every group (enumerated).
  static constexpr int kPitch = TK == 64 ? 40 : 24;
  static constexpr int kPiece = kTM * kPitch; // dwords of one piece image
  static constexpr int kTileDw = 3 * kPiece;  // one operand tile: three pieces
  static constexpr int kLdsBytes = 2 * 2 * kTileDw * 4;   // x and w tiles, double buffered: 122,880 B (73,728 B: two workgroups per CU)
};

struct FcArgs {
  const float* x;        // [B, K]
  const float* w;        // [N, K]
  const float* bias;     // [N] or nullptr
  const float* in_scale; // [K] or nullptr: x <- relu(x * in_scale + in_shift)
  const float* in_shift;
  const float* mask;     // [B, N] or nullptr: y <- y * mask (Dropout: 0 or 1 / (1 - p), drawn by the caller)
  const float* add;      // [N] or nullptr: y <- y + add (STN3d: the flattened identity)
  float* y;              // [B, N]
  int B, K, N;
  // statistics / BatchNorm of this layer (gamma == nullptr: none)
  float* part;           // [row tiles][2][N] partial sums
  unsigned* ticket;
  const float *gamma, *beta;
  float *running_mean, *running_var;
  long long* nbt;
  float momentum, eps;
  int train;
  float *out_scale, *out_shift;   // [N]
};

// AFFINE: the producer's BatchNorm (1: + ReLU, 2: without -- the point stacks' last BatchNorm in front of fc1, pointnet.py:188,
// radarnet.py:60) is applied to x while it is staged (compile-time: a run-time test around the loads made hipcc wait vmcnt(0) behind
// every one of them, i.e. no tile was ever in flight under the MFMAs)
template <int AFFINE, int TK>
__global__ __launch_bounds__(kFcThreads, TK == 32 ? 2 : 1) void fc_kernel(const FcArgs a) {
  using G = FcGeo<TK>;
  constexpr int kTK = G::kTK, kTPR = G::kTPR, kRPP = G::kRPP, kPitch = G::kPitch, kPiece = G::kPiece, kTileDw = G::kTileDw;
  extern __shared__ __attribute__((aligned(16))) unsigned fc_lds[];
  unsigned* const As = fc_lds;                 // [2][3 pieces][kTM][kPitch]
  unsigned* const Bs = fc_lds + 2 * kTileDw;
  __shared__ float colsum[2][2][kTN];       // [row half][sum | sum of squares][column]
  constexpr int HS = kTM / kRPP;                // staging passes (kRPP rows each)
  __shared__ int s_last;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wr = wave >> 2, wc = wave & 3;                   // 2 x 4 wavefronts, 32 x 16 outputs each
  // XCD-aware tile order (blocks b and b + 8 share an XCD and its L2): the column tiles of one row tile run on ONE XCD, next to
  // each other in time, so a row tile of x is fetched into one L2 once (with x = blockIdx.x, y = blockIdx.y every row tile was read
  // by all eight XCDs: 8 x the input through the Infinity Cache).  Speed only: any placement is correct.
  const int nct = (a.N + kTN - 1) / kTN, nrt = (a.B + kTM - 1) / kTM;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int rt = (slot / nct) * 8 + xcd, ct = slot % nct;
  if (rt >= nrt) return;               // padding of the last round (not counted by the arrival counter)
  const int m0 = rt * kTM, n0 = ct * kTN;
  const int li = lane & 15, lk = lane >> 4;
  // staging: thread t moves one float4 (4 consecutive k) of row t / 16 (+ 32 per pass) of each tile
  // TK = 32: eight threads per row, so a ds_write_b64 lane group (16 contiguous lanes, banks mod 32) holds TWO rows: rows r and r + 1 at
  // a pitch of 24 dwords overlap in half their banks; r and r + 2 (48 dwords = 16 mod 32) do not -- bits 0 and 1 of the row are swapped
  const int sj = tid / kTPR;
  const int sr = (kTPR == 8) ? ((sj & ~3) | ((sj & 1) << 1) | ((sj >> 1) & 1)) : sj;
  const int sq = tid % kTPR, sk = sq * 4;
  // Three register sets: the tile of chunk kc is loaded during chunk kc - 3 (a chunk is a fraction of a microsecond of MFMAs, an
  // HBM round trip several times that), staged at the end of chunk kc - 1.  Loads are unconditional (indices clamped, values
  // zeroed by a select when they are staged): nothing consumes a loaded value before its tile is staged, two chunks later --
  // a run-time test around the loads made hipcc wait vmcnt(0) behind every one of them.
  v4f ax[3][HS], bx[3][HS], sc[3], sh[3];
  int rA[HS], rB[HS];
  bool okA[HS], okB[HS];
#pragma unroll
  for (int h = 0; h < HS; ++h) {
    rA[h] = min(m0 + sr + kRPP * h, a.B - 1); rB[h] = min(n0 + sr + kRPP * h, a.N - 1);
    okA[h] = m0 + sr + kRPP * h < a.B; okB[h] = n0 + sr + kRPP * h < a.N;
  }
  auto load = [&](int k0, v4f (&axs)[HS], v4f (&bxs)[HS], v4f& scs, v4f& shs) {
    const int k = min(k0 + sk, a.K - 4);
#pragma unroll
    for (int h = 0; h < HS; ++h) {
      axs[h] = *reinterpret_cast<const v4f*>(a.x + (size_t)rA[h] * a.K + k);
      bxs[h] = *reinterpret_cast<const v4f*>(a.w + (size_t)rB[h] * a.K + k);
    }
    if constexpr (AFFINE != 0) {
      scs = *reinterpret_cast<const v4f*>(a.in_scale + k);
      shs = *reinterpret_cast<const v4f*>(a.in_shift + k);
    }
  };
  typedef unsigned u2v __attribute__((ext_vector_type(2)));
  // x = p0 + p1 + p2 exactly (truncation split, b3d_dev.hpp); a dword holds elements 2 d (low half) and 2 d + 1
  auto put = [&](unsigned* dst, const v4f x) {
    const float f[4] = {x.x, x.y, x.z, x.w};
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      h[i] = __float_as_uint(f[i]);
      const float r1 = f[i] - __uint_as_float(h[i] & 0xffff0000u);
      m[i] = __float_as_uint(r1);
      l[i] = __float_as_uint(r1 - __uint_as_float(m[i] & 0xffff0000u));
    }
    *reinterpret_cast<u2v*>(dst) = u2v{__builtin_amdgcn_perm(h[1], h[0], 0x07060302u), __builtin_amdgcn_perm(h[3], h[2], 0x07060302u)};
    *reinterpret_cast<u2v*>(dst + kPiece) = u2v{__builtin_amdgcn_perm(m[1], m[0], 0x07060302u), __builtin_amdgcn_perm(m[3], m[2], 0x07060302u)};
    *reinterpret_cast<u2v*>(dst + 2 * kPiece) = u2v{__builtin_amdgcn_perm(l[1], l[0], 0x07060302u), __builtin_amdgcn_perm(l[3], l[2], 0x07060302u)};
  };
  auto stage = [&](int buf, int k0, const v4f (&axs)[HS], const v4f (&bxs)[HS], const v4f& scs, const v4f& shs) {
    const bool kok = k0 + sk < a.K;
    const v4f zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int h = 0; h < HS; ++h) {
      v4f v = axs[h];
      if constexpr (AFFINE == 1) {
        v.x = relu1(fmaf(v.x, scs.x, shs.x)); v.y = relu1(fmaf(v.y, scs.y, shs.y));
        v.z = relu1(fmaf(v.z, scs.z, shs.z)); v.w = relu1(fmaf(v.w, scs.w, shs.w));
      } else if constexpr (AFFINE == 2) {
        v.x = fmaf(v.x, scs.x, shs.x); v.y = fmaf(v.y, scs.y, shs.y);
        v.z = fmaf(v.z, scs.z, shs.z); v.w = fmaf(v.w, scs.w, shs.w);
      }
      put(As + buf * kTileDw + (sr + kRPP * h) * kPitch + 2 * sq, (okA[h] && kok) ? v : zero);
      put(Bs + buf * kTileDw + (sr + kRPP * h) * kPitch + 2 * sq, (okB[h] && kok) ? bxs[h] : zero);
    }
  };
  auto frag = [&](const unsigned* p) {                      // this lane's 8 k of one row, three pieces
    Bf3 f;
    f.p0 = __builtin_bit_cast(bf8, *reinterpret_cast<const u4v*>(p));
    f.p1 = __builtin_bit_cast(bf8, *reinterpret_cast<const u4v*>(p + kPiece));
    f.p2 = __builtin_bit_cast(bf8, *reinterpret_cast<const u4v*>(p + 2 * kPiece));
    return f;
  };
  v4f acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};     // rows 32 wr + {0..15}, {16..31}
  const int nk = (a.K + kTK - 1) / kTK;
  load(0, ax[0], bx[0], sc[0], sh[0]);
  load(kTK, ax[1], bx[1], sc[1], sh[1]);             // (chunks past the end are never staged)
  load(2 * kTK, ax[2], bx[2], sc[2], sh[2]);
  stage(0, 0, ax[0], bx[0], sc[0], sh[0]);
  __syncthreads();
  for (int kc3 = 0; kc3 < nk; kc3 += 3) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int kc = kc3 + u;
      if (kc >= nk) break;
      const int buf = kc & 1;
      load((kc + 3) * kTK, ax[u], bx[u], sc[u], sh[u]);      // set u held chunk kc, which is in LDS
      const unsigned* A = As + buf * kTileDw + (32 * wr + li) * kPitch + 4 * lk;
      const unsigned* Bt = Bs + buf * kTileDw + (16 * wc + li) * kPitch + 4 * lk;
#pragma unroll
      for (int kg = 0; kg < kTK / 32; ++kg) {
        const Bf3 bw = frag(Bt + 16 * kg);
        const Bf3 a0 = frag(A + 16 * kg), a1 = frag(A + 16 * kPitch + 16 * kg);
        acc0 = bf_mfma6(a0, bw, acc0);
        acc1 = bf_mfma6(a1, bw, acc1);
      }
      if (kc + 1 < nk) stage(buf ^ 1, (kc + 1) * kTK, ax[(u + 1) % 3], bx[(u + 1) % 3], sc[(u + 1) % 3], sh[(u + 1) % 3]);
      __syncthreads();
    }
  }
  // ---- epilogue: D of block bi: row = m0 + 32 wr + 16 bi + 4 lk + reg, column = n0 + 16 wc + li ----
  float cs = 0.f;                                            // this lane's partial sum of its column
  float vals[8];                                             // ... and its eight outputs (zero where the row or column does not exist)
  {
    const int n = n0 + 16 * wc + li;
    const bool nok = n < a.N;
    const float bv = (a.bias && nok) ? a.bias[n] : 0.f;
    const float av = (a.add && nok) ? a.add[n] : 0.f;
#pragma unroll
    for (int bi = 0; bi < 2; ++bi) {
      const v4f acc = bi ? acc1 : acc0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + 32 * wr + 16 * bi + 4 * lk + r;
        float v = 0.f;
        if (row < a.B && nok) {
          v = acc[r] + bv;
          if (a.mask) v *= a.mask[(size_t)row * a.N + n];
          v += av;
          a.y[(size_t)row * a.N + n] = v;
          cs += v;
        }
        vals[4 * bi + r] = v;
      }
    }
  }
  if (!a.gamma) return;
  // Per-tile column statistics as (sum, M2 = sum of squared deviations from the TILE's mean) -- E[y^2] - E[y]^2 of fp32 sums cancels
  // when |mean| >> std -- in a fixed order: the four row groups of a wavefront (lanes li + 16 lk), then the two row halves.
  {
    float s = cs;
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (lk == 0) colsum[wr][0][16 * wc + li] = s;
  }
  __syncthreads();
  {
    const int rows_here = min(kTM, a.B - m0);
    const float tmean = (colsum[0][0][16 * wc + li] + colsum[1][0][16 * wc + li]) / (float)rows_here;
    float q = 0.f;
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + 32 * wr + 16 * bi + 4 * lk + r;
        const float dv = vals[4 * bi + r] - tmean;
        q = row < a.B ? fmaf(dv, dv, q) : q;
      }
    q += __shfl_xor(q, 16, 64);
    q += __shfl_xor(q, 32, 64);
    if (lk == 0) colsum[wr][1][16 * wc + li] = q;
  }
  __syncthreads();
  if (tid < 2 * kTN) {
    const int st = tid / kTN, c = tid % kTN, n = n0 + c;
    if (n < a.N) a.part[((size_t)rt * 2 + st) * a.N + n] = colsum[0][st][c] + colsum[1][st][c];
  }
  // The last workgroup OF A COLUMN TILE to finish forms the BatchNorm affine of its 64 columns (MI355X_MICROARCH.md hand-off: drained
  // stores, barrier, one lane's agent-scope release in front of the ticket; one acquire on the last workgroup in front of its reads).
  // (Until round 4 the last workgroup of the LAUNCH did all N columns, one thread per column walking the row tiles twice with one
  // dependent L2 round trip per tile: a 15 - 25 us single-workgroup tail behind every layer -- 38 us for a 132-workgroup launch whose
  // workgroups live 16 us.)  Here: thread (c, g) = (tid % 64, wavefront g) takes the row tiles t = g, g + 8, ... of column n0 + c --
  // four or five independent loads per pass -- and the eight partials of a column are added in wavefront order.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(a.ticket + 1 + ct, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = (t == (unsigned)nrt - 1u) ? 1 : 0;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    s_last = last;
  }
  __syncthreads();
  if (!s_last) return;
  double* const red = reinterpret_cast<double*>(fc_lds);      // [8][64] (the operand tiles are done with)
  const int c = tid & 63, g = tid >> 6, n = n0 + c;
  const bool nok = n < a.N;
  const long long nbt_after = a.nbt ? *a.nbt + 1 : 1;          // (written only after EVERY column tile has been finalised, below)
  const double cnt = (double)a.B;
  double m = 0.0, m2 = 0.0;
  if (a.train) {
    // Chan's combination of the per-tile (count, sum, M2) in float64: the mean first, then the squared deviations about it
    double s = 0.0;
    if (nok)
      for (int t = g; t < nrt; t += 8) s += (double)a.part[((size_t)t * 2) * a.N + n];
    red[g * 64 + c] = s;
    __syncthreads();
    s = 0.0;
#pragma unroll
    for (int gg = 0; gg < 8; ++gg) s += red[gg * 64 + c];
    m = s / cnt;
    __syncthreads();
    double q = 0.0;
    if (nok)
      for (int t = g; t < nrt; t += 8) {
        const double nt = (double)min(kTM, a.B - t * kTM);
        const double dm = (double)a.part[((size_t)t * 2) * a.N + n] / nt - m;
        q += (double)a.part[((size_t)t * 2 + 1) * a.N + n] + nt * dm * dm;
      }
    red[g * 64 + c] = q;
    __syncthreads();
#pragma unroll
    for (int gg = 0; gg < 8; ++gg) m2 += red[gg * 64 + c];
  }
  if (g == 0 && nok) {
    float mean, var;
    if (a.train) {
      mean = (float)m; var = (float)(m2 / cnt);
      if (a.running_mean) {
        const double mom = a.momentum >= 0.f ? (double)a.momentum : 1.0 / (double)nbt_after;
        const double unbiased = (double)var * (cnt / (cnt > 1.0 ? cnt - 1.0 : 1.0));
        a.running_mean[n] = (float)((1.0 - mom) * (double)a.running_mean[n] + mom * (double)mean);
        a.running_var[n] = (float)((1.0 - mom) * (double)a.running_var[n] + mom * unbiased);
      }
    } else {
      mean = a.running_mean[n]; var = a.running_var[n];
    }
    const float sc = a.gamma[n] / sqrtf(var + a.eps);
    a.out_scale[n] = sc;
    a.out_shift[n] = a.beta[n] - mean * sc;
  }
  // num_batches_tracked moves once per call, after the last column tile has read it; both tickets are re-armed for the next call
  // (stream order)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    a.ticket[1 + ct] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == (unsigned)nct - 1u) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      if (a.train && a.nbt) *a.nbt = nbt_after;
      a.ticket[0] = 0u;
    }
  }
}

__global__ __launch_bounds__(256) void affine_relu_kernel(const float* __restrict__ y, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, long total, int N, float* __restrict__ out) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int n = (int)(i % N);
    out[i] = relu1(fmaf(y[i], scale[n], shift[n]));
  }
}

}  // namespace
}  // namespace b3d

using namespace b3d;

extern "C" size_t b3d_fc_bn_workspace_bytes(int32_t B, int32_t N) {
  if (B < 0) B = 0;
  if (N < 0) N = 0;
  return 256 + (size_t)((B + kTM - 1) / kTM + 1) * 2 * (size_t)N * 4 + 256;
}

extern "C" int b3d_fc_bn_forward(const float* x, int32_t B, int32_t K, const float* w, const float* bias, int32_t N,
                                 const float* in_scale, const float* in_shift, int32_t in_relu, const float* mask, const float* add,
                                 const b3d_batchnorm* bn, int32_t train, float* y, float* out_scale, float* out_shift,
                                 void* workspace, size_t workspace_bytes, b3d_stream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  B3D_REQUIRE(B >= 0 && K > 0 && N > 0, "b3d_fc_bn_forward: bad shape [%d, %d] x [%d, %d]", (int)B, (int)K, (int)N, (int)K);
  if (B == 0) return B3D_OK;
  B3D_REQUIRE(x && w && y && workspace, "b3d_fc_bn_forward: null argument");
  B3D_REQUIRE(K % 4 == 0, "b3d_fc_bn_forward: K %d must be a multiple of 4 (16-byte rows)", (int)K);
  B3D_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "b3d_fc_bn_forward: input affine comes as a pair");
  if (bn) {
    B3D_REQUIRE(bn->gamma && bn->beta && out_scale && out_shift, "b3d_fc_bn_forward: BatchNorm without gamma / beta / outputs");
    B3D_REQUIRE((bn->running_mean == nullptr) == (bn->running_var == nullptr), "b3d_fc_bn_forward: running statistics come as a pair");
    B3D_REQUIRE(train || bn->running_mean, "b3d_fc_bn_forward: eval mode needs running statistics");
    B3D_REQUIRE(!train || B > 1, "b3d_fc_bn_forward: batch statistics need more than one row");
  }
  if (workspace_bytes < b3d_fc_bn_workspace_bytes(B, N)) return fail(B3D_ERR_WORKSPACE, "b3d_fc_bn_forward: workspace too small");
  uintptr_t p = ((uintptr_t)workspace + 255) & ~(uintptr_t)255;
  FcArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.w = w; a.bias = bias; a.in_scale = in_scale; a.in_shift = in_shift; a.mask = mask; a.add = add; a.y = y;
  a.B = B; a.K = K; a.N = N;
  a.ticket = (unsigned*)p;
  a.part = (float*)(p + 256);
  a.train = train ? 1 : 0;
  if (bn) {
    a.gamma = bn->gamma; a.beta = bn->beta; a.running_mean = bn->running_mean; a.running_var = bn->running_var;
    a.nbt = train ? (long long*)bn->num_batches_tracked : nullptr;
    a.momentum = bn->momentum; a.eps = bn->eps;
    a.out_scale = out_scale; a.out_shift = out_shift;
  }
  const int nrt = (B + kTM - 1) / kTM, nct = (N + kTN - 1) / kTN;
  B3D_REQUIRE(nct <= 63, "b3d_fc_bn_forward: N %d > 4032 columns (one arrival counter per 64-column tile in the workspace header)", (int)N);
  const dim3 grid((unsigned)((nrt + 7) / 8 * 8 * nct));                // row tiles padded to whole rounds of the eight XCDs
  // more tiles than CUs: the 61 KB form, two workgroups per CU, keeps the launch to one round
  static int cus_of[64] = {};                                          // compute units per device ordinal (0 = not asked yet)
  int dev = 0;
  B3D_HIP_CHECK(hipGetDevice(&dev));
  if (dev >= 0 && dev < 64 && cus_of[dev] == 0) {
    int cus = 0;
    B3D_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    cus_of[dev] = cus > 0 ? cus : 256;
  }
  const int n_cus = (dev >= 0 && dev < 64) ? cus_of[dev] : 256;
  const bool small = nrt * nct > n_cus;
  auto go = [&](auto kern, int lds) -> int {
    B3D_TRY(set_lds_cached(reinterpret_cast<const void*>(kern), lds));
    hipLaunchKernelGGL(kern, grid, dim3(kFcThreads), lds, stream, a);
    return B3D_OK;
  };
  if (in_scale && in_relu) B3D_TRY(small ? go(fc_kernel<1, 32>, FcGeo<32>::kLdsBytes) : go(fc_kernel<1, 64>, FcGeo<64>::kLdsBytes));
  else if (in_scale) B3D_TRY(small ? go(fc_kernel<2, 32>, FcGeo<32>::kLdsBytes) : go(fc_kernel<2, 64>, FcGeo<64>::kLdsBytes));
  else B3D_TRY(small ? go(fc_kernel<0, 32>, FcGeo<32>::kLdsBytes) : go(fc_kernel<0, 64>, FcGeo<64>::kLdsBytes));
  return launch_check("fc_kernel");
}

extern "C" int b3d_fc_ticket_init(void* workspace, size_t workspace_bytes, b3d_stream stream_) {
  B3D_REQUIRE(workspace && workspace_bytes >= 512, "b3d_fc_ticket_init: workspace");
  uintptr_t p = ((uintptr_t)workspace + 255) & ~(uintptr_t)255;
  B3D_HIP_CHECK(hipMemsetAsync((void*)p, 0, 256, (hipStream_t)stream_));
  return B3D_OK;
}

extern "C" int b3d_affine_relu(const float* y, const float* scale, const float* shift, int32_t B, int32_t N, float* out,
                               b3d_stream stream_) {
  B3D_REQUIRE(B >= 0 && N > 0, "b3d_affine_relu: bad shape");
  if (B == 0) return B3D_OK;
  B3D_REQUIRE(y && scale && shift && out, "b3d_affine_relu: null argument");
  const long total = (long)B * N;
  long blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(affine_relu_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, y, scale, shift, total, N, out);
  return launch_check("affine_relu_kernel");
}
