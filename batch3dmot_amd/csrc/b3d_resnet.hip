// ResNetAE.encode -- the frozen camera encoder of the camera+LiDAR+radar model (reference
// batch_3dmot/models/resnet_fully_conv.py:42-82,84-161: conv(3,12,4,2,1) -> ResidualBlock(12,24,k4,s2) -> ResidualBlock(24,48,k3,s1) ->
// ResidualBlock(48,96,k3,s2) on [N,3,32,32] crops -> [N,96]), train-mode BatchNorm included.
//
// The whole encoder is 1.5 MMAC per crop (9 GFLOP per 3,000 crops) on activations of at most 12 KB per crop: what it
// costs through the library path is launches -- ten convolutions, nine batch-statistics BatchNorms (one workgroup per
// CHANNEL there: 12-96 workgroups on a 256-CU part), ReLUs and residual adds, ~60 launches, 0.9 ms.  Here: six phase
// kernels.  A phase boundary sits where train-mode BatchNorm needs the statistics of the WHOLE batch before its output
// can be formed; inside a phase everything stays in LDS.  Every phase
//   * stages its input crops into zero-bordered LDS tiles, applying the producer's BatchNorm (scale / shift formed from
//     the batch sums the previous phase accumulated, or from the running statistics in eval mode), the residual add and
//     the ReLU on the way -- BatchNorm / ReLU / add never run as kernels of their own;
//   * evaluates its convolutions on the MATRIX CORES (round 4; rounds 1-3: direct convolutions on the vector ALUs with the weights
//     in SGPRs, 0 % MFMA, 0.42 ms for 3,000 crops and a chain of dependent scalar-load round trips per input channel): an
//     implicit GEMM out[c][pixel] = sum_k W[c][k] col[k][pixel] with k = (ci, ky, kx) in the weight's own order, as
//     v_mfma_f32_16x16x32_bf16 tiles in the bf16x6 form of b3d_dev.hpp (fp32-class accuracy).  The weights are split once per call
//     into operand fragments (resnet_pack_frag_kernel) and travel global -> registers -> a two-slot LDS buffer one k-step ahead,
//     shared by the 16 wavefronts; a wavefront owns 16 (crop, pixel) pairs, gathers the 8 tile words of its im2col fragment per
//     k-step from LDS through a per-convolution tap-offset table, splits them into three bf16 pieces and issues 6 MFMAs per
//     16 output channels; one barrier per k-step;
//   * writes the RAW convolution outputs and accumulates their per-channel sum and sum of squares (16-lane DPP reduction ->
//     one row of sums per WAVEFRONT in LDS, every address with a single writer -> rows added in wavefront order -> a
//     per-workgroup row of partial sums -> added in block order by the last workgroup to arrive: bitwise reproducible).
// Phases (crops per workgroup): P0 conv + block1.conv1 + block1.downsample (2), P1 block1.conv2 (12), P2 block2.conv1 +
// downsample (16), P3 block2.conv2 (16), P4 block3.conv1 + downsample (16), P5 block3.conv2 (16), then the output
// kernel (BatchNorm + add + ReLU of the last block) and the running-statistics update of all nine BatchNorms.
// Measured alone on 3,000 crops (tools/resnet_phase_times.sh): 119 + 27 + 30 + 31 + 49 + 18 us (VALU form: 125 + 36 + 46 + 60 + 65 + 35);
// what is left per k-step is the split of the gathered words on the vector ALUs (~100 instructions per wavefront, 16 wavefronts
// per CU) and, per phase, ~10 us of launch + tile zero fill + BatchNorm affine + statistics hand-off.
#include "b3d_common.hpp"
#include "b3d_launch.hpp"
#include "b3d_dev.hpp"

namespace b3d {
namespace {

constexpr int kConvs = 10, kBns = 9;
// convolution order: 0 conv, 1 b1.conv1, 2 b1.conv2, 3 b1.down, 4 b2.conv1, 5 b2.conv2, 6 b2.down, 7 b3.conv1, 8 b3.conv2, 9 b3.down
// BatchNorm order:   0 b1.bn1, 1 b1.bn2, 2 b1.down, 3 b2.bn1, 4 b2.bn2, 5 b2.down, 6 b3.bn1, 7 b3.bn2, 8 b3.down
constexpr int kCin[kConvs] = {3, 12, 24, 12, 24, 48, 24, 48, 96, 48};
constexpr int kCout[kConvs] = {12, 24, 24, 24, 48, 48, 48, 96, 96, 96};
constexpr int kKer[kConvs] = {4, 4, 4, 5, 3, 3, 1, 3, 3, 3};
constexpr int kBnC[kBns] = {24, 24, 24, 48, 48, 48, 96, 96, 96};
constexpr int kBnPix[kBns] = {64, 16, 16, 16, 16, 16, 4, 1, 1};        // output pixels per crop behind each BatchNorm
__host__ __device__ constexpr int bn_off(int i) { int o = 0; for (int k = 0; k < i; ++k) o += kBnC[k]; return o; }
constexpr int kBnChannels = bn_off(kBns);                              // 504
// Matrix-core form of a convolution (round 4): out[c][pixel] = sum_k W[c][k] col[k][pixel], k = (ci, ky, kx) in the weight's own
// order, as v_mfma_f32_16x16x32_bf16 tiles with W as the first operand (16 output channels x 32 k) and the im2col columns as the
// second (32 k x 16 pixels), bf16x6 products (b3d_dev.hpp).  The weights are split ONCE into operand fragments: for k-step ks and
// channel tile ct, piece p: 64 lanes x 16 bytes, lane l = (channel 16 ct + l % 16, k = 32 ks + 8 (l / 16) + 0..7).
// (block3.conv2, i = 8, is a k3 s2 p1 convolution of a 2 x 2 input to ONE output pixel: only the taps (1..2, 1..2) meet the input, so
// its k runs over (ci, y, x) of the 2 x 2 input -- 384 instead of 864.)
__host__ __device__ constexpr int conv_kdim(int i) { return i == 8 ? kCin[8] * 4 : kCin[i] * kKer[i] * kKer[i]; }
// index of tap k of convolution i inside a torch weight row [ci][ky][kx]
__host__ __device__ constexpr int conv_tap_index(int i, int k) { return i == 8 ? ((k >> 2) * 9 + (1 + ((k >> 1) & 1)) * 3 + 1 + (k & 1)) : k; }
__host__ __device__ constexpr int conv_ksteps(int i) { return (conv_kdim(i) + 31) / 32; }
__host__ __device__ constexpr int conv_ctiles(int i) { return (kCout[i] + 15) / 16; }
__host__ __device__ constexpr int frag_u4(int i) { return conv_ksteps(i) * conv_ctiles(i) * 3 * 64; }
__host__ __device__ constexpr int frag_off(int i) { int o = 0; for (int k = 0; k < i; ++k) o += frag_u4(k); return o; }
constexpr int kFragU4 = frag_off(kConvs);
constexpr int kResThreads = 1024, kResWaves = 16;                      // every phase kernel
constexpr int kResMaxGrid = 512;                                       // persistent workgroups of a phase, at most

struct BnDev {
  const float *gamma, *beta;
  float *running_mean, *running_var;
  long long* nbt;
  float momentum, eps;
};
struct ResArgs {
  int N, train;
  const float* x;          // [N, 3, 32, 32]
  const u4v* wfrag;        // bf16x3 operand fragments of every convolution: conv i at frag_off(i) (see conv_kdim above)
  const float* bias[kConvs];
  BnDev bn[kBns];
  double* sums;            // [kBnChannels][2]  batch sums, written by the LAST workgroup of the producing phase
  float* part;             // [kResMaxGrid][2 kBnChannels]  per-workgroup partial sums (train mode)
  unsigned* tickets;       // [kBns] arrival counters of the workgroups of a phase (zeroed per call)
  float *z1, *zd1, *z2, *z3, *zd2, *z4, *z5, *zd3, *z6;
  float* out;              // [N, 96]
};

struct PackW { const float* src[kConvs]; };
// fragments of all convolutions: one thread per (conv, k-step, channel tile, lane)
__global__ __launch_bounds__(256) void resnet_pack_frag_kernel(const PackW p, u4v* __restrict__ dst) {
  const int i = blockIdx.y;
  const int KD = conv_kdim(i), KST = conv_ksteps(i), CT = conv_ctiles(i), CO = kCout[i];
  for (int t = blockIdx.x * 256 + threadIdx.x; t < KST * CT * 64; t += gridDim.x * 256) {
    const int lane = t & 63, ct = (t >> 6) % CT, ks = (t >> 6) / CT;
    const int c = 16 * ct + (lane & 15), k0 = 32 * ks + 8 * (lane >> 4);
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float w = (c < CO && k0 + j < KD) ? p.src[i][(long)c * (kCin[i] * kKer[i] * kKer[i]) + conv_tap_index(i, k0 + j)] : 0.f;
      h[j] = __float_as_uint(w);
      const float r1 = w - __uint_as_float(h[j] & 0xffff0000u);
      m[j] = __float_as_uint(r1);
      l[j] = __float_as_uint(r1 - __uint_as_float(m[j] & 0xffff0000u));
    }
    u4v q0, q1, q2;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      q0[d] = __builtin_amdgcn_perm(h[2 * d + 1], h[2 * d], 0x07060302u);
      q1[d] = __builtin_amdgcn_perm(m[2 * d + 1], m[2 * d], 0x07060302u);
      q2[d] = __builtin_amdgcn_perm(l[2 * d + 1], l[2 * d], 0x07060302u);
    }
    u4v* o = dst + frag_off(i) + ((size_t)(ks * CT + ct) * 3) * 64 + lane;
    o[0] = q0; o[64] = q1; o[128] = q2;
  }
}

// ---- per-channel affine of a BatchNorm: batch statistics (train) or running statistics (eval) -----------------------------
__device__ __forceinline__ void bn_affine(const ResArgs& a, int b, int c, float& scale, float& shift) {
  const BnDev& bn = a.bn[b];
  float mean, var;
  if (a.train) {
    const double cnt = (double)a.N * kBnPix[b];
    const double m = a.sums[2 * (bn_off(b) + c)] / cnt;
    double v = a.sums[2 * (bn_off(b) + c) + 1] / cnt - m * m;
    if (v < 0.0) v = 0.0;
    mean = (float)m; var = (float)v;
  } else {
    mean = bn.running_mean[c]; var = bn.running_var[c];
  }
  scale = bn.gamma[c] / sqrtf(var + bn.eps);
  shift = bn.beta[c] - mean * scale;
}
// aff[0..C) = scale, aff[C..2C) = shift
__device__ __forceinline__ void bn_affine_to_lds(const ResArgs& a, int b, float* aff) {
  const int C = kBnC[b];
  for (int c = threadIdx.x; c < C; c += blockDim.x) bn_affine(a, b, c, aff[c], aff[C + c]);
}

// ---- wavefront sums of a channel's outputs -> LDS accumulators ------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row_sum16(float v) {
  v += dpp_mov<0x128>(v);      // row_ror:8
  v += dpp_mov<0x124>(v);
  v += dpp_mov<0x122>(v);
  v += dpp_mov<0x121>(v);
  return v;
}
// Batch sums in a FIXED order (the result must not depend on which workgroup arrives first: no float or double atomics
// across workgroups).  Every workgroup parks its partial sums in its own row of `part`; the workgroup whose ticket is the
// last one adds the rows in block order (float64) into `sums`, which the NEXT launch reads.  Hand-off as
// MI355X_MICROARCH.md prescribes: stores drained by every storing wavefront, workgroup barrier, one lane's agent-scope
// release in front of the ticket; on the last workgroup one agent-scope acquire in front of the barrier that precedes the reads.
typedef float v4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void stat_flush(const ResArgs& a, int b, const float* stat, double* scratch) {
  __shared__ int s_last;
  const int n2 = 2 * kBnC[b], off = 2 * bn_off(b);
  for (int i = threadIdx.x; i < n2; i += blockDim.x) a.part[(size_t)blockIdx.x * (2 * kBnChannels) + off + i] = stat[i];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(&a.tickets[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = (t == gridDim.x - 1) ? 1 : 0;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    s_last = last;
  }
  __syncthreads();
  if (s_last) {
    // rows added in block order, the blocks cut into contiguous slices that the threads share out (the partition depends
    // on the launch geometry only): a thread adds four neighbouring columns of its slice (16-byte loads, eight rows in
    // flight), float64; then the slices are added in order.  `scratch`: 32 KB of this workgroup's LDS that is dead by now.
    double* s_slice = scratch;
    const int n4 = n2 / 4;
    const int nsl = (int)blockDim.x / n4;                    // 21 .. 85 slices
    const int per = ((int)gridDim.x + nsl - 1) / nsl;
    const int sl = (int)threadIdx.x / n4, q = (int)threadIdx.x - sl * n4;
    if (sl < nsl) {
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
      const int b0 = sl * per, b1 = min(b0 + per, (int)gridDim.x);
      const float* base = a.part + off + 4 * q;
      for (int blk = b0; blk < b1; blk += 8) {
        v4s v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          v[u] = (blk + u < b1) ? *reinterpret_cast<const v4s*>(base + (size_t)(blk + u) * (2 * kBnChannels)) : v4s{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 8; ++u) { s0 += (double)v[u].x; s1 += (double)v[u].y; s2 += (double)v[u].z; s3 += (double)v[u].w; }
      }
      double* d = s_slice + (size_t)sl * n2 + 4 * q;
      d[0] = s0; d[1] = s1; d[2] = s2; d[3] = s3;
    }
    __syncthreads();
    if ((int)threadIdx.x < n2) {
      double sum = 0.0;
      for (int k = 0; k < nsl; ++k) sum += s_slice[k * n2 + threadIdx.x];
      a.sums[off + threadIdx.x] = sum;
    }
  }
  __syncthreads();                                  // s_last is reused by a second flush of the same kernel
}

__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// ---- convolution on the matrix cores -----------------------------------------------------------------------------------------
// LDS of one convolution: koff[32 KST] (tile offset of tap k: (ci HP + ky) HP + kx; 0 behind the last tap, where the weights are
// zero), wbuf[2][CT * 192] u4v (the weight fragments of the current and the next k-step, shared by all wavefronts), and in train
// mode wstat[waves][2 COUT] (per-wavefront sums: every address has ONE writer, the rows are added in wavefront order).
template <int CI, int HP>
__device__ __forceinline__ void conv_koff_fill(int* koff) {
  constexpr int K = kKer[CI], KD = conv_kdim(CI), KST = conv_ksteps(CI);
  for (int k = threadIdx.x; k < 32 * KST; k += blockDim.x) {
    int o = 0;
    if (k < KD) {
      if constexpr (CI == 8) o = k;                          // rows [ci][2][2] of the 2 x 2 input
      else { const int ci = k / (K * K), r = k - ci * (K * K), ky = r / K, kx = r - ky * K; o = (ci * HP + ky) * HP + kx; }
    }
    koff[k] = o;
  }
}
__device__ __forceinline__ Bf3 split8(const float (&v)[8]) {
  unsigned h[8], m[8], l[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    h[j] = __float_as_uint(v[j]);
    const float r1 = v[j] - __uint_as_float(h[j] & 0xffff0000u);
    m[j] = __float_as_uint(r1);
    l[j] = __float_as_uint(r1 - __uint_as_float(m[j] & 0xffff0000u));
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
// One convolution of a crop group.  Wavefront w takes the pixel tiles w, w + NW, ... (16 consecutive (crop, pixel) pairs each; at
// most TPW of them) and ALL channel tiles, so that the im2col fragment of a (tile, k-step) -- 8 gathered LDS words per lane, split
// into three bf16 pieces -- is built once.  Every k-step: the weight fragments of step ks + 1 travel global -> registers under the
// MFMAs of step ks and are parked in the other half of wbuf behind them; ONE barrier per k-step.
// S: stride; OFF: 0 for a padded convolution (pad 1), 1 for an unpadded one; HO: output height = width.
// z == nullptr: the raw output (+ bias) goes to the LDS tile `ztile` [crop][COUT][ZP][ZP] at (+1, +1) instead (P0's first layer).
template <int CI, int HP, int S, int OFF, int HO, int CROPS, int NTS, int CSPLIT = 1, int ZP = 0, int TSTRIDE = kCin[CI] * HP * HP>
__device__ __forceinline__ void conv_mfma(const ResArgs& a, int img0, const float* tiles, const int* koff, u4v* wbuf, const float* bias,
                                          float* z, float* wstat, float* ztile = nullptr) {
  constexpr int COUT = kCout[CI], KST = conv_ksteps(CI), CT = conv_ctiles(CI);
  constexpr int PIX = HO * HO, NT = (CROPS * PIX + 15) / 16, TPW = (NT + NTS - 1) / NTS, FR = CT * 192;
  constexpr int NW = NTS, CTW = CT / CSPLIT;                 // pixel-tile slots; channel tiles per wavefront
  static_assert(CT % CSPLIT == 0, "channel tiles divide over the channel groups");
  const int wave_id = uniform(threadIdx.x >> 6), lane = threadIdx.x & 63, n = lane & 15, kq = lane >> 4;
  // wavefront -> (pixel-tile slot, channel group); wavefronts past NTS * CSPLIT only take part in the barriers and the weight loads
  const bool computes = wave_id < NTS * CSPLIT;
  const int wave = computes ? wave_id % NTS : 0, ct0 = computes ? (wave_id / NTS) * CTW : 0;
  const u4v* wf = a.wfrag + frag_off(CI);
  // this lane's pixel (as the column of the second operand) in each of its tiles
  const float* base[TPW];
  bool act[TPW];
  int imv[TPW], pxv[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int gp = (wave + NW * i) * 16 + n;
    act[i] = wave + NW * i < NT && gp < CROPS * PIX;
    const int im = act[i] ? gp / PIX : 0, px = act[i] ? gp % PIX : 0;
    imv[i] = im; pxv[i] = px;
    base[i] = tiles + im * TSTRIDE + ((px / HO) * S + OFF) * HP + (px % HO) * S + OFF;
  }
  v4f acc[TPW][CTW];
#pragma unroll
  for (int i = 0; i < TPW; ++i)
#pragma unroll
    for (int ct = 0; ct < CTW; ++ct) acc[i][ct] = v4f{0.f, 0.f, 0.f, 0.f};
  constexpr int LPT = (FR + kResThreads - 1) / kResThreads;  // fragment words per thread and k-step (6 channel tiles: 1,152 words)
  auto wload = [&](int ks, u4v (&r)[LPT]) {
#pragma unroll
    for (int u = 0; u < LPT; ++u) {
      const int t = threadIdx.x + u * kResThreads;
      r[u] = (t < FR && ks < KST) ? wf[(size_t)ks * FR + t] : u4v{0u, 0u, 0u, 0u};
    }
  };
  auto wstore = [&](int ks, const u4v (&r)[LPT]) {
#pragma unroll
    for (int u = 0; u < LPT; ++u) {
      const int t = threadIdx.x + u * kResThreads;
      if (t < FR && ks < KST) wbuf[(ks & 1) * FR + t] = r[u];
    }
  };
  u4v pre[LPT], pre2[LPT];
  wload(0, pre);
  wstore(0, pre);
  // software pipeline: the weight fragments of step ks + 2 are in flight (registers) during step ks, the tile offsets of step ks + 2
  // and the gathered tile words of step ks + 1 are fetched under the split + MFMAs of step ks
  wload(1, pre);
  auto offsets = [&](int ks, int4& o0, int4& o1) {
    const int kk = ks < KST ? ks : KST - 1;
    o0 = *reinterpret_cast<const int4*>(koff + 32 * kk + 8 * kq);
    o1 = *reinterpret_cast<const int4*>(koff + 32 * kk + 8 * kq + 4);
  };
  auto gather = [&](const int4& o0, const int4& o1, float (&v)[TPW][8]) {
    if (!computes) return;                                   // (wave-uniform: the other wavefronts only move weight fragments)
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      const float* b = base[i];
      v[i][0] = b[o0.x]; v[i][1] = b[o0.y]; v[i][2] = b[o0.z]; v[i][3] = b[o0.w];
      v[i][4] = b[o1.x]; v[i][5] = b[o1.y]; v[i][6] = b[o1.z]; v[i][7] = b[o1.w];
    }
  };
  int4 oa, ob;
  float vc[TPW][8] = {};
  offsets(0, oa, ob);
  gather(oa, ob, vc);
  offsets(1, oa, ob);
  __syncthreads();
#pragma unroll 1
  for (int ks = 0; ks < KST; ++ks) {
    const u4v* cur = wbuf + (ks & 1) * FR;
    wload(ks + 2, pre2);
    float vn[TPW][8] = {};
    gather(oa, ob, vn);                                      // step ks + 1 (the last step gathers its own taps again: unused)
    offsets(ks + 2, oa, ob);
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      if (computes && wave + NW * i < NT) {                  // wave-uniform
        const Bf3 col = split8(vc[i]);
#pragma unroll
        for (int ct = 0; ct < CTW; ++ct) {
          Bf3 w;
          w.p0 = __builtin_bit_cast(bf8, cur[((ct0 + ct) * 3 + 0) * 64 + lane]);
          w.p1 = __builtin_bit_cast(bf8, cur[((ct0 + ct) * 3 + 1) * 64 + lane]);
          w.p2 = __builtin_bit_cast(bf8, cur[((ct0 + ct) * 3 + 2) * 64 + lane]);
          acc[i][ct] = bf_mfma6(w, col, acc[i][ct]);
        }
      }
    }
    wstore(ks + 1, pre);
#pragma unroll
    for (int u = 0; u < LPT; ++u) pre[u] = pre2[u];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) vc[i][j] = vn[i][j];
    __syncthreads();
  }
  // D: lane (n, kq) holds channels 16 ct + 4 kq + r of pixel n
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    if (computes && wave + NW * i < NT) {
      const bool valid = act[i] && img0 + imv[i] < a.N;
#pragma unroll
      for (int ct = 0; ct < CTW; ++ct) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 16 * (ct0 + ct) + 4 * kq + r;
          {
            const bool cok = c < COUT;
            const float val = cok ? acc[i][ct][r] + bias[c] : 0.f;
            if constexpr (ZP > 0) {
              if (act[i] && cok) ztile[(imv[i] * COUT + c) * (ZP * ZP) + (pxv[i] / HO + 1) * ZP + (pxv[i] % HO) + 1] = val;
            } else {
              if (valid && cok) z[((long)(img0 + imv[i]) * COUT + c) * PIX + pxv[i]] = val;
              if (a.train) {
                const float sv = valid ? val : 0.f;
                const float s1 = row_sum16(sv), s2 = row_sum16(sv * sv);
                if (n == 0 && cok) { wstat[wave_id * (2 * COUT) + 2 * c] += s1; wstat[wave_id * (2 * COUT) + 2 * c + 1] += s2; }
              }
            }
          }
        }
      }
    }
  }
}
// stat[i] = sum over the wavefront rows of wstat, in wavefront order (call behind a barrier; one more before stat is read)
template <int COUT, int NW>
__device__ __forceinline__ void wstat_reduce(const float* wstat, float* stat) {
  for (int i = threadIdx.x; i < 2 * COUT; i += blockDim.x) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += wstat[w * (2 * COUT) + i];
    stat[i] = s;
  }
}

// ---- P0: conv -> block1.conv1 (+ statistics) and block1.downsample (+ statistics): 2 crops, 16 wavefronts, matrix cores --------
// conv 3 -> 12 k4 s2 p1 (32 -> 16): 512 pixels = 32 pixel tiles, two per wavefront, K = 48 in 2 k-steps, its output (+ bias) lands
// in the zero-bordered LDS tile a0; block1.conv1 12 -> 24 k4 s2 p1 (16 -> 8): 8 pixel tiles x 2 channel tiles = 16 wavefronts,
// K = 192; block1.downsample 12 -> 24 k5 s3 p0 (16 -> 4): 2 pixel tiles x 2 channel tiles = 4 wavefronts, K = 300.
// EXPERIMENT (round 6, -DB3D_P0_CROPS=4): four crops per pass (half the passes, twice the pixel tiles per weight fragment): 119.9 -> 116.2 us
// on 3,000 crops with 23 spilled registers -- not worth them; the phase is bound by the gathers of its im2col operand, not by passes.
#ifndef B3D_P0_CROPS
#define B3D_P0_CROPS 2
#endif
constexpr int kP0Crops = B3D_P0_CROPS;
constexpr int kP0X = 3 * 34 * 34, kP0A = 12 * 18 * 18;
constexpr int kP0Lds = kP0Crops * (kP0X + kP0A) * 4 + 2 * 2 * 192 * 16 + 32 * (conv_ksteps(0) + conv_ksteps(1) + conv_ksteps(3)) * 4
                       + (2 * 48 + 2 * kResWaves * 48) * 4;
__global__ __launch_bounds__(kResThreads) void resnet_p0_kernel(const ResArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xin = smem;                               // [2][3][34][34]
  float* a0 = xin + kP0Crops * kP0X;               // [2][12][18][18]
  u4v* wbuf = reinterpret_cast<u4v*>(a0 + kP0Crops * kP0A);
  int* koff0 = reinterpret_cast<int*>(wbuf + 2 * 2 * 192);
  int* koff1 = koff0 + 32 * conv_ksteps(0);
  int* koffd = koff1 + 32 * conv_ksteps(1);
  float* st1 = reinterpret_cast<float*>(koffd + 32 * conv_ksteps(3));   // [24][2] block1.bn1
  float* std_ = st1 + 48;                          // [24][2] block1.downsample
  float* wst1 = std_ + 48;
  float* wstd = wst1 + kResWaves * 48;
  for (int i = threadIdx.x; i < kP0Crops * (kP0X + kP0A); i += kResThreads) smem[i] = 0.f;
  for (int i = threadIdx.x; i < 96 + 2 * kResWaves * 48; i += kResThreads) st1[i] = 0.f;
  conv_koff_fill<0, 34>(koff0);
  conv_koff_fill<1, 18>(koff1);
  conv_koff_fill<3, 18>(koffd);
  const int groups = (a.N + kP0Crops - 1) / kP0Crops;
  for (int g = blockIdx.x; g < groups; g += gridDim.x) {
    const int img0 = g * kP0Crops;
    __syncthreads();                               // the previous pass is done with the tiles (and the zero fill is visible)
    for (int i = threadIdx.x; i < kP0Crops * 3 * 1024; i += kResThreads) {
      const int im = i / 3072, r = i - im * 3072, c = r >> 10, y = (r >> 5) & 31, xx = r & 31;
      const float v = img0 + im < a.N ? a.x[(long)(img0 + im) * 3072 + r] : 0.f;
      xin[im * kP0X + (c * 34 + y + 1) * 34 + xx + 1] = v;
    }
    __syncthreads();
    conv_mfma<0, 34, 2, 0, 16, kP0Crops, 16, 1, 18>(a, img0, xin, koff0, wbuf, a.bias[0], nullptr, nullptr, a0);
    __syncthreads();                               // a0 is complete
    conv_mfma<1, 18, 2, 0, 8, kP0Crops, 8, 2>(a, img0, a0, koff1, wbuf, a.bias[1], a.z1, wst1);
    conv_mfma<3, 18, 3, 1, 4, kP0Crops, kP0Crops, 2>(a, img0, a0, koffd, wbuf, a.bias[3], a.zd1, wstd);     // one 16-pixel tile per crop
  }
  if (a.train) {
    __syncthreads();
    wstat_reduce<24, kResWaves>(wst1, st1);
    wstat_reduce<24, kResWaves>(wstd, std_);
    __syncthreads();
    stat_flush(a, 0, st1, reinterpret_cast<double*>(smem));
    stat_flush(a, 2, std_, reinterpret_cast<double*>(smem));
  }
}

// ---- middle phases: stage relu(bn(zA) [+ bn(zB)]) into [CROPS][CIN][HP][HP] tiles, then 1-2 convolutions ------------------
// HP: HIN + 2 for a full zero border, HIN + 1 when no tap reaches past the bottom / right edge.
template <int CROPS, int CIN, int HIN, int HP, int NT>
__device__ __forceinline__ void stage_tiles(const ResArgs& a, int img0, const float* zA, const float* affA, const float* zB,
                                            const float* affB, float* tiles) {
  constexpr int PIX = HIN * HIN;
  for (int i = threadIdx.x; i < CROPS * CIN * PIX; i += NT) {
    const int im = i / (CIN * PIX), r = i - im * (CIN * PIX), c = r / PIX, p = r - c * PIX;
    float v = 0.f;
    if (img0 + im < a.N) {
      const long o = (long)(img0 + im) * (CIN * PIX) + r;
      v = fmaf(zA[o], affA[c], affA[CIN + c]);
      if (zB) v += fmaf(zB[o], affB[c], affB[CIN + c]);
      v = relu1(v);
    }
    tiles[(im * CIN + c) * HP * HP + (p / HIN + 1) * HP + (p % HIN) + 1] = v;
  }
}

// P1: block1.conv2 24 -> 24, k4 s2 p1, 8 -> 4, on the matrix cores: 12 crops per workgroup = 12 pixel tiles (one per wavefront,
// four wavefronts only load weights); K = 384 in 12 k-steps, 2 channel tiles.
constexpr int kP1Crops = 12;
constexpr int kP1Tiles = kP1Crops * 24 * 100;
constexpr int kP1Lds = kP1Tiles * 4 + 2 * conv_ctiles(2) * 192 * 16 + 32 * conv_ksteps(2) * 4 + (48 + 48 + kResWaves * 48) * 4;
__global__ __launch_bounds__(kResThreads) void resnet_p1_kernel(const ResArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* tiles = smem;
  u4v* wbuf = reinterpret_cast<u4v*>(tiles + kP1Tiles);
  int* koff = reinterpret_cast<int*>(wbuf + 2 * conv_ctiles(2) * 192);
  float* aff = reinterpret_cast<float*>(koff + 32 * conv_ksteps(2));
  float* stat = aff + 48;
  float* wstat = stat + 48;
  for (int i = threadIdx.x; i < kP1Tiles; i += kResThreads) tiles[i] = 0.f;
  for (int i = threadIdx.x; i < 48 + kResWaves * 48; i += kResThreads) stat[i] = 0.f;
  conv_koff_fill<2, 10>(koff);
  __syncthreads();
  bn_affine_to_lds(a, 0, aff);
  const int groups = (a.N + kP1Crops - 1) / kP1Crops;
  for (int g = blockIdx.x; g < groups; g += gridDim.x) {
    const int img0 = g * kP1Crops;
    __syncthreads();
    stage_tiles<kP1Crops, 24, 8, 10, kResThreads>(a, img0, a.z1, aff, nullptr, nullptr, tiles);
    __syncthreads();
    conv_mfma<2, 10, 2, 0, 4, kP1Crops, 12>(a, img0, tiles, koff, wbuf, a.bias[2], a.z2, wstat);
  }
  if (a.train) {
    __syncthreads();
    wstat_reduce<24, kResWaves>(wstat, stat);
    __syncthreads();
    stat_flush(a, 1, stat, reinterpret_cast<double*>(smem));
  }
}

// P2: y1 = relu(bn2(z2) + bn_d(zd1)); block2.conv1 24 -> 48 k3 s1 p1 and block2.downsample 24 -> 48 k1, 4 -> 4, on the matrix
// cores: 16 crops = 16 pixel tiles, one per wavefront; K = 216 in 7 k-steps and K = 24 in one; 3 channel tiles.
constexpr int kP2Crops = 16;
constexpr int kP2Tiles = kP2Crops * 24 * 36;
constexpr int kP2Lds = kP2Tiles * 4 + 2 * 3 * 192 * 16 + 32 * (conv_ksteps(4) + conv_ksteps(6)) * 4 + (48 + 48 + 96 + 96 + 2 * kResWaves * 96) * 4;
__global__ __launch_bounds__(kResThreads) void resnet_p2_kernel(const ResArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* tiles = smem;
  u4v* wbuf = reinterpret_cast<u4v*>(tiles + kP2Tiles);
  int* koff1 = reinterpret_cast<int*>(wbuf + 2 * 3 * 192);
  int* koffd = koff1 + 32 * conv_ksteps(4);
  float* affA = reinterpret_cast<float*>(koffd + 32 * conv_ksteps(6));
  float* affB = affA + 48;
  float* stat1 = affB + 48;
  float* statd = stat1 + 96;
  float* wstat1 = statd + 96;
  float* wstatd = wstat1 + kResWaves * 96;
  for (int i = threadIdx.x; i < kP2Tiles; i += kResThreads) tiles[i] = 0.f;
  for (int i = threadIdx.x; i < 192 + 2 * kResWaves * 96; i += kResThreads) stat1[i] = 0.f;
  conv_koff_fill<4, 6>(koff1);
  conv_koff_fill<6, 6>(koffd);
  __syncthreads();
  bn_affine_to_lds(a, 1, affA);
  bn_affine_to_lds(a, 2, affB);
  const int groups = (a.N + kP2Crops - 1) / kP2Crops;
  for (int g = blockIdx.x; g < groups; g += gridDim.x) {
    const int img0 = g * kP2Crops;
    __syncthreads();
    stage_tiles<kP2Crops, 24, 4, 6, kResThreads>(a, img0, a.z2, affA, a.zd1, affB, tiles);
    __syncthreads();
    conv_mfma<4, 6, 1, 0, 4, kP2Crops, 16>(a, img0, tiles, koff1, wbuf, a.bias[4], a.z3, wstat1);
    conv_mfma<6, 6, 1, 1, 4, kP2Crops, 16>(a, img0, tiles, koffd, wbuf, a.bias[6], a.zd2, wstatd);
  }
  if (a.train) {
    __syncthreads();
    wstat_reduce<48, kResWaves>(wstat1, stat1);
    wstat_reduce<48, kResWaves>(wstatd, statd);
    __syncthreads();
    stat_flush(a, 3, stat1, reinterpret_cast<double*>(smem));
    stat_flush(a, 5, statd, reinterpret_cast<double*>(smem));
  }
}

// P3: block2.conv2 48 -> 48 k3 s1 p1 on relu(bn1(z3)), on the matrix cores: 16 crops per workgroup = 16 pixel tiles, one per
// wavefront; K = 432 in 14 k-steps, 3 channel tiles.
constexpr int kP3Crops = 16;
constexpr int kP3Tiles = kP3Crops * 48 * 36;
constexpr int kP3Lds = kP3Tiles * 4 + 2 * conv_ctiles(5) * 192 * 16 + 32 * conv_ksteps(5) * 4 + (96 + 96 + kResWaves * 96) * 4;
__global__ __launch_bounds__(kResThreads) void resnet_p3_kernel(const ResArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* tiles = smem;
  u4v* wbuf = reinterpret_cast<u4v*>(tiles + kP3Tiles);
  int* koff = reinterpret_cast<int*>(wbuf + 2 * conv_ctiles(5) * 192);
  float* aff = reinterpret_cast<float*>(koff + 32 * conv_ksteps(5));
  float* stat = aff + 96;
  float* wstat = stat + 96;
  for (int i = threadIdx.x; i < kP3Tiles; i += kResThreads) tiles[i] = 0.f;
  for (int i = threadIdx.x; i < 96 + kResWaves * 96; i += kResThreads) stat[i] = 0.f;
  conv_koff_fill<5, 6>(koff);
  __syncthreads();
  bn_affine_to_lds(a, 3, aff);
  const int groups = (a.N + kP3Crops - 1) / kP3Crops;
  for (int g = blockIdx.x; g < groups; g += gridDim.x) {
    const int img0 = g * kP3Crops;
    __syncthreads();
    stage_tiles<kP3Crops, 48, 4, 6, kResThreads>(a, img0, a.z3, aff, nullptr, nullptr, tiles);
    __syncthreads();
    conv_mfma<5, 6, 1, 0, 4, kP3Crops, kResWaves>(a, img0, tiles, koff, wbuf, a.bias[5], a.z4, wstat);
  }
  if (a.train) {
    __syncthreads();
    wstat_reduce<48, kResWaves>(wstat, stat);
    __syncthreads();
    stat_flush(a, 4, stat, reinterpret_cast<double*>(smem));
  }
}

// P4: y2 = relu(bn2(z4) + bn_d(zd2)); block3.conv1 48 -> 96 k3 s2 p1 (4 -> 2) and block3.downsample 48 -> 96 k3 s2 p0 (4 -> 1), on
// the matrix cores.  16 crops in 5x5 tiles (no tap reaches the bottom / right border): conv1 = 64 pixels = 4 pixel tiles x 3 groups of
// 2 channel tiles (12 wavefronts), downsample = 16 pixels = 1 pixel tile x 6 channel tiles (6 wavefronts); K = 432 in 14 k-steps.
constexpr int kP4Crops = 16;
constexpr int kP4Tiles = kP4Crops * 48 * 25;
constexpr int kP4Lds = kP4Tiles * 4 + 2 * 6 * 192 * 16 + 32 * (conv_ksteps(7) + conv_ksteps(9)) * 4 + (96 + 96 + 192 + 192 + 2 * kResWaves * 192) * 4;
static_assert(kP4Lds <= 160 * 1024, "LDS of one CU");
__global__ __launch_bounds__(kResThreads) void resnet_p4_kernel(const ResArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* tiles = smem;
  u4v* wbuf = reinterpret_cast<u4v*>(tiles + kP4Tiles);
  int* koff1 = reinterpret_cast<int*>(wbuf + 2 * 6 * 192);
  int* koffd = koff1 + 32 * conv_ksteps(7);
  float* affA = reinterpret_cast<float*>(koffd + 32 * conv_ksteps(9));
  float* affB = affA + 96;
  float* stat1 = affB + 96;
  float* statd = stat1 + 192;
  float* wstat1 = statd + 192;
  float* wstatd = wstat1 + kResWaves * 192;
  for (int i = threadIdx.x; i < kP4Tiles; i += kResThreads) tiles[i] = 0.f;
  for (int i = threadIdx.x; i < 384 + 2 * kResWaves * 192; i += kResThreads) stat1[i] = 0.f;
  conv_koff_fill<7, 5>(koff1);
  conv_koff_fill<9, 5>(koffd);
  __syncthreads();
  bn_affine_to_lds(a, 4, affA);
  bn_affine_to_lds(a, 5, affB);
  const int groups = (a.N + kP4Crops - 1) / kP4Crops;
  for (int g = blockIdx.x; g < groups; g += gridDim.x) {
    const int img0 = g * kP4Crops;
    __syncthreads();
    stage_tiles<kP4Crops, 48, 4, 5, kResThreads>(a, img0, a.z4, affA, a.zd2, affB, tiles);
    __syncthreads();
    conv_mfma<7, 5, 2, 0, 2, kP4Crops, 4, 3>(a, img0, tiles, koff1, wbuf, a.bias[7], a.z5, wstat1);
    conv_mfma<9, 5, 2, 1, 1, kP4Crops, 1, 6>(a, img0, tiles, koffd, wbuf, a.bias[9], a.zd3, wstatd);
  }
  if (a.train) {
    __syncthreads();
    wstat_reduce<96, kResWaves>(wstat1, stat1);
    wstat_reduce<96, kResWaves>(wstatd, statd);
    __syncthreads();
    stat_flush(a, 6, stat1, reinterpret_cast<double*>(smem));
    stat_flush(a, 8, statd, reinterpret_cast<double*>(smem));
  }
}

// P5: block3.conv2 96 -> 96 k3 s2 p1 on relu(bn1(z5)), 2 -> 1: only the taps (1..2, 1..2) meet the 2x2 input, a 384-wide
// matrix-vector product per crop; on the matrix cores with the crops as the pixel dimension: 16 crops per workgroup = ONE pixel tile,
// six wavefronts take one channel tile each; K = 384 in 12 k-steps.  Rows of 388 floats: the 16 crops of a tile on distinct banks.
constexpr int kP5Crops = 16, kP5Row = 388, kP5Threads = kResThreads;
constexpr int kP5Lds = kP5Crops * kP5Row * 4 + 2 * 6 * 192 * 16 + 32 * conv_ksteps(8) * 4 + (192 + 192 + kResWaves * 192) * 4;
__global__ __launch_bounds__(kP5Threads) void resnet_p5_kernel(const ResArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* rows = smem;
  u4v* wbuf = reinterpret_cast<u4v*>(rows + kP5Crops * kP5Row);
  int* koff = reinterpret_cast<int*>(wbuf + 2 * 6 * 192);
  float* aff = reinterpret_cast<float*>(koff + 32 * conv_ksteps(8));
  float* stat = aff + 192;
  float* wstat = stat + 192;
  for (int i = threadIdx.x; i < 192 + kResWaves * 192; i += kP5Threads) stat[i] = 0.f;
  conv_koff_fill<8, 2>(koff);
  bn_affine_to_lds(a, 6, aff);
  const int groups = (a.N + kP5Crops - 1) / kP5Crops;
  for (int g = blockIdx.x; g < groups; g += gridDim.x) {
    const int img0 = g * kP5Crops;
    __syncthreads();
    for (int i = threadIdx.x; i < kP5Crops * 384; i += kP5Threads) {
      const int im = i / 384, r = i - im * 384;
      float v = 0.f;
      if (img0 + im < a.N) v = relu1(fmaf(a.z5[(long)(img0 + im) * 384 + r], aff[r >> 2], aff[96 + (r >> 2)]));
      rows[im * kP5Row + r] = v;
    }
    __syncthreads();
    conv_mfma<8, 2, 1, 0, 1, kP5Crops, 1, 6, 0, kP5Row>(a, img0, rows, koff, wbuf, a.bias[8], a.z6, wstat);
  }
  if (a.train) {
    __syncthreads();
    wstat_reduce<96, kResWaves>(wstat, stat);
    __syncthreads();
    stat_flush(a, 7, stat, reinterpret_cast<double*>(smem));
  }
}

// out = relu(bn2(z6) + bn_d(zd3)); in train mode block 0.. also update the running statistics of all nine BatchNorms
__global__ __launch_bounds__(256) void resnet_out_kernel(const ResArgs a) {
  const long n = (long)a.N * 96;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = (int)(i % 96);
    float s1, t1, s2, t2;
    bn_affine(a, 7, c, s1, t1);
    bn_affine(a, 8, c, s2, t2);
    a.out[i] = relu1(fmaf(a.z6[i], s1, t1) + fmaf(a.zd3[i], s2, t2));
  }
}
__global__ __launch_bounds__(128) void resnet_track_kernel(const ResArgs a) {
  const int b = blockIdx.x, c = threadIdx.x;
  const BnDev& bn = a.bn[b];
  if (!bn.running_mean) return;
  const long long nbt_after = bn.nbt ? *bn.nbt + 1 : 1;
  if (c < kBnC[b]) {
    const double cnt = (double)a.N * kBnPix[b];
    const double m = a.sums[2 * (bn_off(b) + c)] / cnt;
    double v = a.sums[2 * (bn_off(b) + c) + 1] / cnt - m * m;
    if (v < 0.0) v = 0.0;
    const double mom = bn.momentum >= 0.f ? (double)bn.momentum : 1.0 / (double)nbt_after;
    const double unbiased = (double)(float)v * (cnt / (cnt > 1.0 ? cnt - 1.0 : 1.0));
    bn.running_mean[c] = (float)((1.0 - mom) * (double)bn.running_mean[c] + mom * (double)(float)m);
    bn.running_var[c] = (float)((1.0 - mom) * (double)bn.running_var[c] + mom * (double)(float)unbiased);
  }
  __syncthreads();
  if (c == 0 && bn.nbt) *bn.nbt = nbt_after;
}

constexpr size_t kActFloatsPerCrop = 24 * 64 + 24 * 16 + 24 * 16 + 48 * 16 + 48 * 16 + 48 * 16 + 96 * 4 + 96 + 96;

}  // namespace
}  // namespace b3d

using namespace b3d;

extern "C" size_t b3d_resnet_encode_workspace_bytes(int32_t N) {
  if (N < 0) N = 0;
  return 256 + (size_t)kFragU4 * 16 + 256 + (size_t)kBnChannels * 2 * 8 + 64 + 256
         + (size_t)kResMaxGrid * 2 * kBnChannels * 4 + 256 + (size_t)N * kActFloatsPerCrop * 4 + 64;
}

extern "C" int b3d_resnet_encode(const b3d_linear* conv, const b3d_batchnorm* bn, const float* x, int32_t N, int32_t train,
                                 void* workspace, size_t workspace_bytes, float* out, b3d_stream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  B3D_REQUIRE(conv && bn && workspace && (N == 0 || (x && out)), "b3d_resnet_encode: null argument");
  B3D_REQUIRE(N >= 0, "b3d_resnet_encode: N %d", (int)N);
  for (int i = 0; i < kConvs; ++i) B3D_REQUIRE(conv[i].w && conv[i].b, "b3d_resnet_encode: null convolution %d", i);
  for (int i = 0; i < kBns; ++i) {
    B3D_REQUIRE(bn[i].gamma && bn[i].beta, "b3d_resnet_encode: null BatchNorm %d", i);
    B3D_REQUIRE((bn[i].running_mean == nullptr) == (bn[i].running_var == nullptr), "b3d_resnet_encode: running statistics come as a pair");
    B3D_REQUIRE(train || bn[i].running_mean, "b3d_resnet_encode: eval mode needs the running statistics of BatchNorm %d", i);
  }
  B3D_REQUIRE(!train || N != 1, "b3d_resnet_encode: batch statistics need more than one value per channel");
  if (workspace_bytes < b3d_resnet_encode_workspace_bytes(N)) return fail(B3D_ERR_WORKSPACE, "b3d_resnet_encode: workspace too small");
  if (N == 0) return B3D_OK;
  auto align = [](uintptr_t p) { return (p + 255) & ~(uintptr_t)255; };
  uintptr_t p = align((uintptr_t)workspace);
  u4v* wfrag = (u4v*)p; p = align(p + (size_t)kFragU4 * 16);
  double* sums = (double*)p; p = align(p + (size_t)kBnChannels * 16 + 64);      // + the arrival counters behind the sums
  float* part = (float*)p; p = align(p + (size_t)kResMaxGrid * 2 * kBnChannels * 4);
  float* act = (float*)p;
  ResArgs a;
  a.N = N; a.train = train ? 1 : 0; a.x = x; a.wfrag = wfrag; a.sums = sums; a.part = part; a.tickets = (unsigned*)(sums + 2 * kBnChannels); a.out = out;
  for (int i = 0; i < kConvs; ++i) a.bias[i] = (const float*)conv[i].b;
  for (int i = 0; i < kBns; ++i) {
    a.bn[i].gamma = bn[i].gamma; a.bn[i].beta = bn[i].beta;
    a.bn[i].running_mean = bn[i].running_mean; a.bn[i].running_var = bn[i].running_var;
    a.bn[i].nbt = (long long*)bn[i].num_batches_tracked; a.bn[i].momentum = bn[i].momentum; a.bn[i].eps = bn[i].eps;
  }
  const size_t n = (size_t)N;
  a.z1 = act; act += n * 24 * 64;
  a.zd1 = act; act += n * 24 * 16;
  a.z2 = act; act += n * 24 * 16;
  a.z3 = act; act += n * 48 * 16;
  a.zd2 = act; act += n * 48 * 16;
  a.z4 = act; act += n * 48 * 16;
  a.z5 = act; act += n * 96 * 4;
  a.zd3 = act; act += n * 96;
  a.z6 = act;
  PackW pw;
  for (int i = 0; i < kConvs; ++i) pw.src[i] = (const float*)conv[i].w;
  hipLaunchKernelGGL(resnet_pack_frag_kernel, dim3(8, kConvs), dim3(256), 0, stream, pw, wfrag);
  B3D_TRY(launch_check("resnet_pack_frag_kernel"));
  if (train) B3D_HIP_CHECK(hipMemsetAsync(sums, 0, (size_t)kBnChannels * 16 + 64, stream));
  // persistent workgroups (as many as are resident at once): the tile zero fill and the BatchNorm affine are per workgroup
  auto grid = [&](int crops, int resident = kResMaxGrid) { const int g = (N + crops - 1) / crops; return dim3((unsigned)(g < resident ? g : resident)); };
  B3D_TRY(set_lds(resnet_p0_kernel, kP0Lds));
  hipLaunchKernelGGL(resnet_p0_kernel, grid(kP0Crops), dim3(kResThreads), kP0Lds, stream, a);
  B3D_TRY(launch_check("resnet_p0_kernel"));
  B3D_TRY(set_lds(resnet_p1_kernel, kP1Lds));
  hipLaunchKernelGGL(resnet_p1_kernel, grid(kP1Crops), dim3(kResThreads), kP1Lds, stream, a);
  B3D_TRY(launch_check("resnet_p1_kernel"));
  B3D_TRY(set_lds(resnet_p2_kernel, kP2Lds));
  hipLaunchKernelGGL(resnet_p2_kernel, grid(kP2Crops), dim3(kResThreads), kP2Lds, stream, a);
  B3D_TRY(launch_check("resnet_p2_kernel"));
  B3D_TRY(set_lds(resnet_p3_kernel, kP3Lds));
  hipLaunchKernelGGL(resnet_p3_kernel, grid(kP3Crops), dim3(kResThreads), kP3Lds, stream, a);
  B3D_TRY(launch_check("resnet_p3_kernel"));
  B3D_TRY(set_lds(resnet_p4_kernel, kP4Lds));
  hipLaunchKernelGGL(resnet_p4_kernel, grid(kP4Crops, 256), dim3(kResThreads), kP4Lds, stream, a);
  B3D_TRY(launch_check("resnet_p4_kernel"));
  B3D_TRY(set_lds(resnet_p5_kernel, kP5Lds));
  hipLaunchKernelGGL(resnet_p5_kernel, grid(kP5Crops, 256), dim3(kP5Threads), kP5Lds, stream, a);
  B3D_TRY(launch_check("resnet_p5_kernel"));
  hipLaunchKernelGGL(resnet_out_kernel, dim3((unsigned)((n * 96 + 255) / 256 < 1024 ? (n * 96 + 255) / 256 : 1024)), dim3(256), 0, stream, a);
  B3D_TRY(launch_check("resnet_out_kernel"));
  if (train) {
    hipLaunchKernelGGL(resnet_track_kernel, dim3(kBns), dim3(128), 0, stream, a);
    B3D_TRY(launch_check("resnet_track_kernel"));
  }
  return B3D_OK;
}
