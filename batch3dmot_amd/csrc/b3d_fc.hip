// Fully connected heads of the frozen point encoders (reference batch_3dmot/models/pointnet.py:46-48
// STN3d fc1-bn4-relu, fc2-bn5-relu, fc3; pointnet.py:188-192 and radarnet.py:60-64 forward_feat: fc1-bn-relu,
// fc2-dropout-bn-relu) as a chain of ONE kernel per Linear:
//
//     y = mask * ( act_in(x) . W^T + b )          act_in(x) = relu(x * in_scale + in_shift)  (the producer's BatchNorm + ReLU,
//                                                  applied while the tile is staged; identity for the first Linear)
//
// with the per-column sum and sum of squares of y accumulated on the way out, from which the LAST workgroup to finish forms
// this layer's BatchNorm affine (batch statistics in train mode, running statistics in eval mode) and updates the running
// statistics as nn.BatchNorm1d does.  BatchNorm, ReLU and Dropout never run as kernels of their own, the normalised
// activations never exist in memory; a final elementwise kernel (b3d_affine_relu) materialises the last activation.
// Exact fp32: v_mfma_f32_16x16x4_f32 (bitwise an fmaf chain over k).  Statistics are summed in a fixed order (per-tile
// partials in a slab, added in tile order in float64): bitwise reproducible.
#include "b3d_common.hpp"
#include "b3d_launch.hpp"

namespace b3d {
namespace {

// 64 x 64 tiles, EIGHT wavefronts (two per SIMD), a 32 x 16 block each: a 2,100 x 512 layer is 264 workgroups, about one per CU, and
// a single wavefront per SIMD ran at a third of the MFMA rate (LDS and barrier waits with nobody to cover them); 32 x 32 tiles
// (four workgroups per CU) doubled the operand traffic and were slower still.
constexpr int kTM = 64, kTN = 64, kTK = 32, kFcThreads = 512;
constexpr int kLd = kTK + 2;     // [row][k] tiles, pitch 34 dwords: bank = 2 row + k -- the ds_read_b32 of a 16x16x4 operand (16 rows x
                                 // 2 k per 32-lane group) and the staging ds_write_b64 (2 rows x 8 k-quads per 16-lane group) are conflict-free

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

// AFFINE: the producer's BatchNorm + ReLU is applied to x while it is staged (compile-time: a run-time test around the loads made
// hipcc wait vmcnt(0) behind every one of them, i.e. no tile was ever in flight under the MFMAs)
template <bool AFFINE>
__global__ __launch_bounds__(kFcThreads) void fc_kernel(const FcArgs a) {
  __shared__ __attribute__((aligned(16))) float As[2][kTM * kLd];
  __shared__ __attribute__((aligned(16))) float Bs[2][kTN * kLd];
  __shared__ float colsum[2][2][kTN];       // [row half][sum | sum of squares][column]
  constexpr int HS = kTM / (kFcThreads / 8);    // staging passes (64 rows each)
  __shared__ int s_last;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wr = wave >> 2, wc = wave & 3;                   // 2 x 4 wavefronts, 32 x 16 outputs each
  // XCD-aware tile order (blocks b and b + 8 share an XCD and its L2): the column tiles of one row tile run on ONE XCD, next to
  // each other in time, so a row tile of x is fetched into one L2 once (with x = blockIdx.x, y = blockIdx.y every row tile was read
  // by all eight XCDs: 8 x the input through the Infinity Cache, 74 -> see profiles/r03_*).  Speed only: any placement is correct.
  const int nct = (a.N + kTN - 1) / kTN, nrt = (a.B + kTM - 1) / kTM;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int rt = (slot / nct) * 8 + xcd, ct = slot % nct;
  if (rt >= nrt) return;               // padding of the last round (not counted by the arrival counter)
  const int m0 = rt * kTM, n0 = ct * kTN;
  const int li = lane & 15, lk = lane >> 4;
  // staging: thread t moves one float4 (4 consecutive k) of row t / 8 of each tile
  const int sr = tid >> 3, sk = (tid & 7) * 4;
  // Three register sets: the tile of chunk kc is loaded during chunk kc - 3 (a chunk is ~0.4 us of MFMAs, an HBM round trip several
  // times that: with one chunk of distance the kernel ran at a quarter of the MFMA rate), staged at the end of chunk kc - 1.
  // Loads are unconditional (indices clamped, values zeroed by a select when they are staged): nothing consumes a loaded value
  // before its tile is staged, two chunks later.
  v4f ax[3][HS], bx[3][HS], sc[3], sh[3];
  int rA[HS], rB[HS];
  bool okA[HS], okB[HS];
#pragma unroll
  for (int h = 0; h < HS; ++h) {
    rA[h] = min(m0 + sr + 64 * h, a.B - 1); rB[h] = min(n0 + sr + 64 * h, a.N - 1);
    okA[h] = m0 + sr + 64 * h < a.B; okB[h] = n0 + sr + 64 * h < a.N;
  }
  auto load = [&](int k0, v4f (&axs)[HS], v4f (&bxs)[HS], v4f& scs, v4f& shs) {
    const int k = min(k0 + sk, a.K - 4);
#pragma unroll
    for (int h = 0; h < HS; ++h) {
      axs[h] = *reinterpret_cast<const v4f*>(a.x + (size_t)rA[h] * a.K + k);
      bxs[h] = *reinterpret_cast<const v4f*>(a.w + (size_t)rB[h] * a.K + k);
    }
    if constexpr (AFFINE) {
      scs = *reinterpret_cast<const v4f*>(a.in_scale + k);
      shs = *reinterpret_cast<const v4f*>(a.in_shift + k);
    }
  };
  auto stage = [&](int buf, int k0, const v4f (&axs)[HS], const v4f (&bxs)[HS], const v4f& scs, const v4f& shs) {
    const bool kok = k0 + sk < a.K;
#pragma unroll
    for (int h = 0; h < HS; ++h) {
      typedef float v2f __attribute__((ext_vector_type(2)));
      v4f v = axs[h];
      if constexpr (AFFINE) {
        v.x = relu1(fmaf(v.x, scs.x, shs.x)); v.y = relu1(fmaf(v.y, scs.y, shs.y));
        v.z = relu1(fmaf(v.z, scs.z, shs.z)); v.w = relu1(fmaf(v.w, scs.w, shs.w));
      }
      const bool oa = okA[h] && kok, ob = okB[h] && kok;
      const v4f u = bxs[h];
      float* pa = &As[buf][(sr + 64 * h) * kLd + sk];
      *reinterpret_cast<v2f*>(pa) = v2f{oa ? v.x : 0.f, oa ? v.y : 0.f};
      *reinterpret_cast<v2f*>(pa + 2) = v2f{oa ? v.z : 0.f, oa ? v.w : 0.f};
      float* pb = &Bs[buf][(sr + 64 * h) * kLd + sk];
      *reinterpret_cast<v2f*>(pb) = v2f{ob ? u.x : 0.f, ob ? u.y : 0.f};
      *reinterpret_cast<v2f*>(pb + 2) = v2f{ob ? u.z : 0.f, ob ? u.w : 0.f};
    }
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
      const float* A = &As[buf][(32 * wr + li) * kLd + lk];
      const float* Bt = &Bs[buf][(16 * wc + li) * kLd + lk];
#pragma unroll
      for (int ks = 0; ks < kTK / 4; ++ks) {
        const float b = Bt[ks * 4];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[ks * 4], b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[16 * kLd + ks * 4], b, acc1, 0, 0, 0);
      }
      if (kc + 1 < nk) stage(buf ^ 1, (kc + 1) * kTK, ax[(u + 1) % 3], bx[(u + 1) % 3], sc[(u + 1) % 3], sh[(u + 1) % 3]);
      __syncthreads();
    }
  }
  // ---- epilogue: D of block bi: row = m0 + 32 wr + 16 bi + 4 lk + reg, column = n0 + 16 wc + li ----
  float cs = 0.f, cq = 0.f;                                  // this lane's partial sums of its column
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
        if (row < a.B && nok) {
          float v = acc[r] + bv;
          if (a.mask) v *= a.mask[(size_t)row * a.N + n];
          v += av;
          a.y[(size_t)row * a.N + n] = v;
          cs += v;
          cq = fmaf(v, v, cq);
        }
      }
    }
  }
  if (!a.gamma) return;
  // column sums in a fixed order: the four row groups of a wavefront (lanes li + 16 lk), then the two row halves
  {
    float s = cs, q = cq;
    s += __shfl_xor(s, 16, 64); q += __shfl_xor(q, 16, 64);
    s += __shfl_xor(s, 32, 64); q += __shfl_xor(q, 32, 64);
    if (lk == 0) { colsum[wr][0][16 * wc + li] = s; colsum[wr][1][16 * wc + li] = q; }
  }
  __syncthreads();
  if (tid < 2 * kTN) {
    const int st = tid / kTN, c = tid % kTN, n = n0 + c;
    if (n < a.N) a.part[((size_t)rt * 2 + st) * a.N + n] = colsum[0][st][c] + colsum[1][st][c];
  }
  // last workgroup: batch statistics -> affine of this layer's BatchNorm (MI355X_MICROARCH.md hand-off: drained stores, barrier,
  // one lane's agent-scope release in front of the ticket; one acquire on the last workgroup in front of its reads)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = (t == (unsigned)(nrt * nct) - 1u) ? 1 : 0;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    s_last = last;
  }
  __syncthreads();
  if (!s_last) return;
  const long long nbt_after = a.nbt ? *a.nbt + 1 : 1;
  for (int n = tid; n < a.N; n += kFcThreads) {
    float mean, var;
    if (a.train) {
      double s = 0.0, q = 0.0;
      for (int t = 0; t < nrt; ++t) {
        s += (double)a.part[((size_t)t * 2) * a.N + n];
        q += (double)a.part[((size_t)t * 2 + 1) * a.N + n];
      }
      const double cnt = (double)a.B;
      const double m = s / cnt;
      double v = q / cnt - m * m;
      if (v < 0.0) v = 0.0;
      mean = (float)m; var = (float)v;
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
  __syncthreads();
  if (tid == 0) {
    if (a.train && a.nbt) *a.nbt = nbt_after;
    *a.ticket = 0u;                                          // re-armed for the next call (stream order)
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
                                 const float* in_scale, const float* in_shift, const float* mask, const float* add,
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
  const dim3 grid((unsigned)((nrt + 7) / 8 * 8 * nct));                // row tiles padded to whole rounds of the eight XCDs
  if (in_scale) hipLaunchKernelGGL(fc_kernel<true>, grid, dim3(kFcThreads), 0, stream, a);
  else hipLaunchKernelGGL(fc_kernel<false>, grid, dim3(kFcThreads), 0, stream, a);
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
