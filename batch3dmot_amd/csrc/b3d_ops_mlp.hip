// Row-wise MLP stacks and the one-key cross-edge attention as standalone operators of the C ABI (SURVEY.md section 8b:
// b3d_mlp_fwd/bwd = K1, K2, K13, K15, K18; b3d_xattn_node_affine = K17):
//
//   b3d_mlp_forward / _backward        nn.Sequential(Linear, ReLU, Linear, ...[, Sigmoid]) of up to five Linear layers with run-time
//                                      widths -- the reference's edge_encoder / node_encoder / edge_classifier / fc_lidar_encoder /
//                                      fc_radar_encoder / att_edge_encoder (clr_att_gnn.py:35-72,81-91; pose_gnn.py:29-53)
//   b3d_xattn_node_affine_forward /    nn.MultiheadAttention called with ONE query and ONE key per edge (clr_att_gnn.py:143-159):
//   _backward                          softmax over one key == 1, so the module is out_proj(v_proj(value)) per NODE
//
// The whole-model entry points (b3d_pose_forward, b3d_clr_forward) run these stacks inside their fused, compile-time-shaped kernels
// (b3d_chain.hpp); this file is the same arithmetic for callers that drive the layers themselves (GNN.knn_writeback, where the
// per-node tables of the model plan cannot be used).  One tiled GEMM kernel on the exact-fp32 matrix instruction
// (v_mfma_f32_16x16x4_f32: bitwise an fmaf chain, no split products) serves all three products of a Linear layer:
//
//   forward          Y  [rows, N] = act(X [rows, K] . W^T + b)
//   data gradient    dX [rows, K] = dZ [rows, N] . W           (* (X > 0): the ReLU in front of this layer, fused into the epilogue)
//   weight gradient  dW [N, K]    = dZ^T . X,  db = column sums of dZ (an extra all-ones column of X), contraction over the rows split
//                                   into fixed chunks -> slabs -> summed in chunk order: bitwise reproducible, no float atomics
#include "b3d_common.hpp"

namespace b3d {
namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int kTM = 64, kTN = 64, kTK = 32, kPitch = kTK + 4, kThreads = 256;
constexpr int kTileFloats = kTM * kPitch;                    // one operand tile in LDS: [64 rows][32 k + pad]

// C[r][j] = sum_c A(r, c) * B(j, c),  r < M, j < Nj, c in the split's share of [0, Kc).
//   A(r, c) = A[r * sAr + c * sAc],  B(j, c) = B[j * sBr + c * sBc]  (one of the two strides of an operand is 1)
struct GemmArgs {
  const float* A; long sAr, sAc;
  const float* B; long sBr, sBc;
  int M, Nj, Kc;
  int vecA, vecB;          // 16-byte loads along the operand's unit-stride dimension are legal (alignment, multiples of 4)
  int ones_col;            // B(ones_col, c) = 1 for every c (bias-gradient column); -1: none
  float* C; long ldc;
  const float* bias;       // [Nj] or nullptr
  const float* mask;       // same geometry as C or nullptr: C <- mask > 0 ? C : 0
  int act;                 // 0: none, 1: ReLU, 2: sigmoid
  int c_per_split;         // contraction elements per blockIdx.z (a multiple of kTK)
  long slab_stride;        // C of split s starts at C + s * slab_stride
};

struct Stage { f4 v[2]; };

// One operand's share of a 64 x 32 tile per thread: two float4.  `along_c`: the float4 runs along the contraction (unit stride sc),
// thread t -> row t / 8 (+ 32), k 4 (t % 8); otherwise it runs along the rows (unit stride sr), thread t -> rows 4 (t % 16), k t / 16 (+ 16).
__device__ __forceinline__ Stage fetch(const float* __restrict__ p, long sr, long sc, bool along_c, bool vec, int r0, int R, int c0,
                                       int c_hi, int ones_row, int tid) {
  Stage s;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    f4 v = {0.f, 0.f, 0.f, 0.f};
    if (along_c) {
      const int r = r0 + (tid >> 3) + 32 * h, c = c0 + 4 * (tid & 7);
      if (r == ones_row) {
        v.x = c < c_hi ? 1.f : 0.f; v.y = c + 1 < c_hi ? 1.f : 0.f; v.z = c + 2 < c_hi ? 1.f : 0.f; v.w = c + 3 < c_hi ? 1.f : 0.f;
      } else if (r < R && c < c_hi) {
        const float* q = p + (long)r * sr + (long)c * sc;
        if (vec && c + 3 < c_hi) v = *reinterpret_cast<const f4*>(q);
        else {
          v.x = q[0];
          if (c + 1 < c_hi) v.y = q[sc];
          if (c + 2 < c_hi) v.z = q[2 * sc];
          if (c + 3 < c_hi) v.w = q[3 * sc];
        }
      }
    } else {
      const int r = r0 + 4 * (tid & 15), c = c0 + (tid >> 4) + 16 * h;
      if (c < c_hi) {
        const float* q = p + (long)r * sr + (long)c * sc;
        if (vec && r + 3 < R) v = *reinterpret_cast<const f4*>(q);
        else {
          if (r < R) v.x = q[0];
          if (r + 1 < R) v.y = q[sr];
          if (r + 2 < R) v.z = q[2 * sr];
          if (r + 3 < R) v.w = q[3 * sr];
        }
        if (ones_row >= r && ones_row < r + 4) {
          float* e = reinterpret_cast<float*>(&v);
          e[ones_row - r] = 1.f;
        }
      }
    }
    s.v[h] = v;
  }
  return s;
}

__device__ __forceinline__ void put(float* __restrict__ tile, const Stage& s, bool along_c, int tid) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (along_c) {
      *reinterpret_cast<f4*>(tile + ((tid >> 3) + 32 * h) * kPitch + 4 * (tid & 7)) = s.v[h];
    } else {
      float* d = tile + (4 * (tid & 15)) * kPitch + (tid >> 4) + 16 * h;
      d[0] = s.v[h].x; d[kPitch] = s.v[h].y; d[2 * kPitch] = s.v[h].z; d[3 * kPitch] = s.v[h].w;
    }
  }
}

__global__ __launch_bounds__(kThreads) void mlp_gemm_kernel(const GemmArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[4 * kTileFloats];       // A and B tiles, double buffered: 36 KB
  float* const As = lds;
  float* const Bs = lds + 2 * kTileFloats;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lq = lane >> 4;
  const int m0 = blockIdx.y * kTM, n0 = blockIdx.x * kTN;
  const int c_lo = blockIdx.z * a.c_per_split;
  const int c_hi = min(a.Kc, c_lo + a.c_per_split);
  const bool acA = a.sAc == 1, acB = a.sBc == 1;
  f4 acc[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) acc[b] = f4{0.f, 0.f, 0.f, 0.f};
  const int nch = c_hi > c_lo ? (c_hi - c_lo + kTK - 1) / kTK : 0;
  if (nch > 0) {
    Stage sa = fetch(a.A, a.sAr, a.sAc, acA, a.vecA != 0, m0, a.M, c_lo, c_hi, -1, tid);
    Stage sb = fetch(a.B, a.sBr, a.sBc, acB, a.vecB != 0, n0, a.ones_col >= 0 ? a.ones_col : a.Nj, c_lo, c_hi, a.ones_col, tid);
    put(As, sa, acA, tid);
    put(Bs, sb, acB, tid);
    __syncthreads();
    for (int ch = 0; ch < nch; ++ch) {
      const int buf = ch & 1;
      const bool more = ch + 1 < nch;
      if (more) {                                            // the next tile travels while this one is multiplied
        sa = fetch(a.A, a.sAr, a.sAc, acA, a.vecA != 0, m0, a.M, c_lo + (ch + 1) * kTK, c_hi, -1, tid);
        sb = fetch(a.B, a.sBr, a.sBc, acB, a.vecB != 0, n0, a.ones_col >= 0 ? a.ones_col : a.Nj, c_lo + (ch + 1) * kTK, c_hi, a.ones_col, tid);
      }
      const float* At = As + buf * kTileFloats + (16 * wave + li) * kPitch + 4 * lq;
      const float* Bt = Bs + buf * kTileFloats + li * kPitch + 4 * lq;
#pragma unroll
      for (int g = 0; g < kTK / 16; ++g) {
        const f4 av = *reinterpret_cast<const f4*>(At + 16 * g);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const f4 bv = *reinterpret_cast<const f4*>(Bt + 16 * b * kPitch + 16 * g);
          // lane (li, lq) supplies contraction elements 16 g + 4 lq + {0,1,2,3} of its row to four instructions; both operands use
          // the same assignment, so the four together cover the 16-wide group once
          acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc[b], 0, 0, 0);
          acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc[b], 0, 0, 0);
          acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc[b], 0, 0, 0);
          acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc[b], 0, 0, 0);
        }
      }
      if (more) {
        put(As + (buf ^ 1) * kTileFloats, sa, acA, tid);
        put(Bs + (buf ^ 1) * kTileFloats, sb, acB, tid);
      }
      __syncthreads();
    }
  }
  // accumulator register r of lane (li, lq), block b: C[m0 + 16 wave + 4 lq + r][n0 + 16 b + li]
  float* const C = a.C + (long)blockIdx.z * a.slab_stride;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int col = n0 + 16 * b + li;
    if (col >= a.Nj) continue;
    const float bv = a.bias ? a.bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + 16 * wave + 4 * lq + r;
      if (row >= a.M) continue;
      float v = acc[b][r] + bv;
      if (a.act == 1) v = relu1(v);
      else if (a.act == 2) v = 1.f / (1.f + __expf(-v));
      if (a.mask) v = a.mask[(long)row * a.ldc + col] > 0.f ? v : 0.f;
      C[(long)row * a.ldc + col] = v;
    }
  }
}

// dW [N, K] (+ db [N]) = slabs [S][N][K + 1] summed in slab order
__global__ __launch_bounds__(256) void mlp_slab_reduce_kernel(const float* __restrict__ slab, int S, int N, int K, float* __restrict__ dw,
                                                              float* __restrict__ db) {
  const long total = (long)N * (K + 1);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    float s = 0.f;
    for (int k = 0; k < S; ++k) s += slab[(long)k * total + i];
    const int n = (int)(i / (K + 1)), c = (int)(i % (K + 1));
    if (c < K) { if (dw) dw[(long)n * K + c] = s; }
    else if (db) db[n] = s;
  }
}

// g = d_y * f'(y): kind 1 = ReLU (y > 0), kind 2 = sigmoid (y (1 - y)); y is the stack's OUTPUT
__global__ __launch_bounds__(256) void mlp_top_grad_kernel(const float* __restrict__ dy, const float* __restrict__ y, long total, int kind,
                                                           float* __restrict__ g) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const float p = y[i];
    g[i] = kind == 2 ? dy[i] * p * (1.f - p) : (p > 0.f ? dy[i] : 0.f);
  }
}

__global__ __launch_bounds__(256) void mlp_zero_kernel(float* __restrict__ p, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) p[i] = 0.f;
}

bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

int launch_gemm(GemmArgs& a, int splits, hipStream_t stream) {
  if (a.M <= 0 || a.Nj <= 0) return B3D_OK;
  const dim3 grid((unsigned)((a.Nj + kTN - 1) / kTN), (unsigned)((a.M + kTM - 1) / kTM), (unsigned)splits);
  hipLaunchKernelGGL(mlp_gemm_kernel, grid, dim3(kThreads), 0, stream, a);
  return launch_check("mlp_gemm_kernel");
}

constexpr int kMaxLayers = 5, kMaxWidth = 1024;

int check_desc(const b3d_mlp_desc* d, const char* who) {
  B3D_REQUIRE(d, "%s: null descriptor", who);
  B3D_REQUIRE(d->n_layers >= 1 && d->n_layers <= kMaxLayers, "%s: n_layers %d not in 1..%d", who, (int)d->n_layers, kMaxLayers);
  for (int l = 0; l <= d->n_layers; ++l)
    B3D_REQUIRE(d->widths[l] >= 1 && d->widths[l] <= kMaxWidth, "%s: widths[%d] = %d not in 1..%d", who, l, (int)d->widths[l], kMaxWidth);
  B3D_REQUIRE((d->relu_mask >> d->n_layers) == 0, "%s: relu_mask 0x%x has bits beyond layer %d", who, d->relu_mask, (int)d->n_layers - 1);
  B3D_REQUIRE(!(d->final_sigmoid && ((d->relu_mask >> (d->n_layers - 1)) & 1u)), "%s: ReLU and Sigmoid behind the last layer", who);
  return B3D_OK;
}

int max_width(const b3d_mlp_desc* d, int from, int to) {
  int m = 1;
  for (int l = from; l <= to; ++l) m = d->widths[l] > m ? d->widths[l] : m;
  return m;
}

// contraction chunks of a weight gradient: enough workgroups to fill the chip, at least 256 rows each
int wgrad_splits(int64_t rows, int N, int K, int* c_per_split) {
  const long tiles = (long)((N + kTM - 1) / kTM) * ((K + 1 + kTN - 1) / kTN);
  long s = (1024 + tiles - 1) / tiles;
  const long maxs = (rows + 255) / 256;
  if (s > maxs) s = maxs;
  if (s < 1) s = 1;
  long cps = ((rows + s - 1) / s + kTK - 1) / kTK * kTK;
  if (cps < kTK) cps = kTK;
  *c_per_split = (int)cps;
  return (int)((rows + cps - 1) / cps > 0 ? (rows + cps - 1) / cps : 1);
}

struct FwdWs { float* h[kMaxLayers]; };          // h[l] = output of layer l (l < L - 1)

bool carve_fwd(const b3d_mlp_desc* d, int64_t rows, bool training, void* ws, size_t bytes, FwdWs& out, size_t* need) {
  Carver c(ws, bytes);
  const int L = d->n_layers;
  if (training) {
    for (int l = 0; l + 1 < L; ++l) out.h[l] = c.take<float>((size_t)rows * d->widths[l + 1]);
  } else if (L > 1) {
    const size_t w = (size_t)max_width(d, 1, L - 1);
    float* p0 = c.take<float>((size_t)rows * w);
    float* p1 = L > 2 ? c.take<float>((size_t)rows * w) : nullptr;
    for (int l = 0; l + 1 < L; ++l) out.h[l] = (l & 1) ? p1 : p0;
  }
  if (need) *need = c.off + 256;
  return c.ok();
}

struct BwdWs { float* g[2]; float* slab; };

bool carve_bwd(const b3d_mlp_desc* d, int64_t rows, void* ws, size_t bytes, BwdWs& out, size_t* need) {
  Carver c(ws, bytes);
  const size_t w = (size_t)max_width(d, 0, d->n_layers);
  out.g[0] = c.take<float>((size_t)rows * w);
  out.g[1] = c.take<float>((size_t)rows * w);
  size_t slab = 0;
  for (int l = 0; l < d->n_layers; ++l) {
    int cps;
    const int s = wgrad_splits(rows, d->widths[l + 1], d->widths[l], &cps);
    const size_t f = (size_t)s * d->widths[l + 1] * (d->widths[l] + 1);
    slab = f > slab ? f : slab;
  }
  out.slab = c.take<float>(slab);
  if (need) *need = c.off + 256;
  return c.ok();
}

int grid1d(long total) {
  long b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace
}  // namespace b3d

using namespace b3d;

extern "C" size_t b3d_mlp_workspace_bytes(const b3d_mlp_desc* d, int64_t rows, uint32_t flags) {
  if (check_desc(d, "b3d_mlp_workspace_bytes") != B3D_OK || rows < 0) return 0;
  FwdWs w{};
  size_t need = 0;
  carve_fwd(d, rows, (flags & B3D_FLAG_TRAINING) != 0, nullptr, 0, w, &need);
  return need;
}

extern "C" int b3d_mlp_forward(const b3d_mlp_desc* d, const b3d_linear* layers, const float* x, int64_t rows, uint32_t flags,
                               void* workspace, size_t workspace_bytes, float* y, b3d_stream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  B3D_TRY(check_desc(d, "b3d_mlp_forward"));
  B3D_REQUIRE(rows >= 0 && rows < (1ll << 31), "b3d_mlp_forward: rows %lld", (long long)rows);
  if (rows == 0) return B3D_OK;
  B3D_REQUIRE(layers && x && y, "b3d_mlp_forward: null argument");
  const int L = d->n_layers;
  for (int l = 0; l < L; ++l) B3D_REQUIRE(layers[l].w, "b3d_mlp_forward: layer %d has no weight", l);
  FwdWs w{};
  if (L > 1) {
    B3D_REQUIRE(workspace, "b3d_mlp_forward: null workspace");
    if (!carve_fwd(d, rows, (flags & B3D_FLAG_TRAINING) != 0, workspace, workspace_bytes, w, nullptr))
      return fail(B3D_ERR_WORKSPACE, "b3d_mlp_forward: workspace too small (%zu bytes)", workspace_bytes);
  }
  const float* in = x;
  for (int l = 0; l < L; ++l) {
    const int K = d->widths[l], N = d->widths[l + 1];
    float* out = l + 1 < L ? w.h[l] : y;
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.A = in; a.sAr = K; a.sAc = 1;
    a.B = layers[l].w; a.sBr = K; a.sBc = 1;
    a.M = (int)rows; a.Nj = N; a.Kc = K;
    a.vecA = (K % 4 == 0 && aligned16(in)) ? 1 : 0;
    a.vecB = (K % 4 == 0 && aligned16(layers[l].w)) ? 1 : 0;
    a.ones_col = -1;
    a.C = out; a.ldc = N;
    a.bias = layers[l].b;
    a.act = ((d->relu_mask >> l) & 1u) ? 1 : ((l + 1 == L && d->final_sigmoid) ? 2 : 0);
    a.c_per_split = (K + kTK - 1) / kTK * kTK;
    B3D_TRY(launch_gemm(a, 1, stream));
    in = out;
  }
  return B3D_OK;
}

extern "C" size_t b3d_mlp_backward_scratch_bytes(const b3d_mlp_desc* d, int64_t rows) {
  if (check_desc(d, "b3d_mlp_backward_scratch_bytes") != B3D_OK || rows < 0) return 0;
  BwdWs w{};
  size_t need = 0;
  carve_bwd(d, rows, nullptr, 0, w, &need);
  return need;
}

extern "C" int b3d_mlp_backward(const b3d_mlp_desc* d, const b3d_linear* layers, const float* x, const float* y, int64_t rows,
                                void* workspace, size_t workspace_bytes, void* scratch, size_t scratch_bytes, const float* d_y,
                                float* d_x, const b3d_linear_grad* grads, b3d_stream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  B3D_TRY(check_desc(d, "b3d_mlp_backward"));
  B3D_REQUIRE(rows >= 0 && rows < (1ll << 31), "b3d_mlp_backward: rows %lld", (long long)rows);
  B3D_REQUIRE(layers && grads, "b3d_mlp_backward: null argument");
  const int L = d->n_layers;
  if (rows == 0) {                                  // an empty batch: every gradient is zero
    for (int l = 0; l < L; ++l) {
      const long nw = (long)d->widths[l] * d->widths[l + 1];
      if (grads[l].w) hipLaunchKernelGGL(mlp_zero_kernel, dim3(grid1d(nw)), dim3(256), 0, stream, grads[l].w, nw);
      if (grads[l].b) hipLaunchKernelGGL(mlp_zero_kernel, dim3(1), dim3(256), 0, stream, grads[l].b, (long)d->widths[l + 1]);
    }
    return launch_check("mlp_zero_kernel");
  }
  B3D_REQUIRE(x && y && d_y && scratch, "b3d_mlp_backward: null argument");
  for (int l = 0; l < L; ++l) B3D_REQUIRE(layers[l].w, "b3d_mlp_backward: layer %d has no weight", l);
  FwdWs fw{};
  if (L > 1) {
    B3D_REQUIRE(workspace, "b3d_mlp_backward: null workspace (the one a B3D_FLAG_TRAINING forward filled)");
    if (!carve_fwd(d, rows, true, workspace, workspace_bytes, fw, nullptr))
      return fail(B3D_ERR_WORKSPACE, "b3d_mlp_backward: workspace too small (%zu bytes)", workspace_bytes);
  }
  BwdWs bw{};
  if (!carve_bwd(d, rows, scratch, scratch_bytes, bw, nullptr))
    return fail(B3D_ERR_WORKSPACE, "b3d_mlp_backward: scratch too small (%zu bytes)", scratch_bytes);
  // gradient at the last layer's pre-activation
  const float* g = d_y;
  int gi = 0;
  const int top = d->final_sigmoid ? 2 : (((d->relu_mask >> (L - 1)) & 1u) ? 1 : 0);
  if (top) {
    const long total = (long)rows * d->widths[L];
    hipLaunchKernelGGL(mlp_top_grad_kernel, dim3(grid1d(total)), dim3(256), 0, stream, d_y, y, total, top, bw.g[0]);
    B3D_TRY(launch_check("mlp_top_grad_kernel"));
    g = bw.g[0];
    gi = 1;
  }
  for (int l = L - 1; l >= 0; --l) {
    const int K = d->widths[l], N = d->widths[l + 1];
    const float* in = l == 0 ? x : fw.h[l - 1];
    if (grads[l].w || grads[l].b) {
      // dW' [N, K + 1] = g^T . (in | 1), the rows cut into chunks -> slabs -> summed in chunk order
      int cps;
      const int S = wgrad_splits(rows, N, K, &cps);
      GemmArgs a;
      memset(&a, 0, sizeof(a));
      a.A = g; a.sAr = 1; a.sAc = N;
      a.B = in; a.sBr = 1; a.sBc = K;
      a.M = N; a.Nj = K + 1; a.Kc = (int)rows;
      a.vecA = (N % 4 == 0 && aligned16(g)) ? 1 : 0;
      a.vecB = (K % 4 == 0 && aligned16(in)) ? 1 : 0;
      a.ones_col = K;
      a.C = bw.slab; a.ldc = K + 1;
      a.c_per_split = cps;
      a.slab_stride = (long)N * (K + 1);
      B3D_TRY(launch_gemm(a, S, stream));
      hipLaunchKernelGGL(mlp_slab_reduce_kernel, dim3(grid1d((long)N * (K + 1))), dim3(256), 0, stream, bw.slab, S, N, K, grads[l].w,
                         grads[l].b);
      B3D_TRY(launch_check("mlp_slab_reduce_kernel"));
    }
    if (l > 0 || d_x) {
      // d in [rows, K] = g . W, masked by the ReLU that produced `in` (layer l - 1's)
      float* out = l == 0 ? d_x : bw.g[gi];
      GemmArgs a;
      memset(&a, 0, sizeof(a));
      a.A = g; a.sAr = N; a.sAc = 1;
      a.B = layers[l].w; a.sBr = 1; a.sBc = K;
      a.M = (int)rows; a.Nj = K; a.Kc = N;
      a.vecA = (N % 4 == 0 && aligned16(g)) ? 1 : 0;
      a.vecB = (K % 4 == 0 && aligned16(layers[l].w)) ? 1 : 0;
      a.ones_col = -1;
      a.C = out; a.ldc = K;
      a.mask = (l > 0 && ((d->relu_mask >> (l - 1)) & 1u)) ? fw.h[l - 1] : nullptr;
      a.c_per_split = (N + kTK - 1) / kTK * kTK;
      B3D_TRY(launch_gemm(a, 1, stream));
      g = out;
      gi ^= 1;
    }
  }
  return B3D_OK;
}

// ---- nn.MultiheadAttention with one query and one key per edge = out_proj(v_proj(value)) per node -------------------------------
namespace {
void xattn_desc(int D, b3d_mlp_desc* d) {
  memset(d, 0, sizeof(*d));
  d->n_layers = 2;
  d->widths[0] = d->widths[1] = d->widths[2] = D;
}
}  // namespace

extern "C" size_t b3d_xattn_node_affine_workspace_bytes(int64_t N, int32_t D, uint32_t flags) {
  if (D < 1 || D > kMaxWidth) return 0;
  b3d_mlp_desc d;
  xattn_desc(D, &d);
  return b3d_mlp_workspace_bytes(&d, N, flags);
}

extern "C" int b3d_xattn_node_affine_forward(const b3d_mha* att, int32_t D, const float* x, int64_t N, uint32_t flags, void* workspace,
                                             size_t workspace_bytes, float* y, b3d_stream stream) {
  B3D_REQUIRE(att && att->in_proj_weight && att->in_proj_bias && att->out_proj_weight && att->out_proj_bias,
              "b3d_xattn_node_affine_forward: null parameter");
  B3D_REQUIRE(D >= 1 && D <= kMaxWidth, "b3d_xattn_node_affine_forward: D %d", (int)D);
  b3d_mlp_desc d;
  xattn_desc(D, &d);
  // in_proj = (q | k | v) rows: the value projection is the last third (clr_att_gnn.py:77-79: kdim = vdim = embed_dim)
  const b3d_linear layers[2] = {{att->in_proj_weight + 2 * (size_t)D * D, att->in_proj_bias + 2 * (size_t)D},
                                {att->out_proj_weight, att->out_proj_bias}};
  return b3d_mlp_forward(&d, layers, x, N, flags, workspace, workspace_bytes, y, stream);
}

extern "C" size_t b3d_xattn_node_affine_scratch_bytes(int64_t N, int32_t D) {
  if (D < 1 || D > kMaxWidth) return 0;
  b3d_mlp_desc d;
  xattn_desc(D, &d);
  return b3d_mlp_backward_scratch_bytes(&d, N);
}

extern "C" int b3d_xattn_node_affine_backward(const b3d_mha* att, int32_t D, const float* x, const float* y, int64_t N, void* workspace,
                                              size_t workspace_bytes, void* scratch, size_t scratch_bytes, const float* d_y, float* d_x,
                                              const b3d_mha_grad* grads, b3d_stream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  B3D_REQUIRE(att && att->in_proj_weight && att->in_proj_bias && att->out_proj_weight && att->out_proj_bias && grads,
              "b3d_xattn_node_affine_backward: null parameter");
  B3D_REQUIRE(D >= 1 && D <= kMaxWidth, "b3d_xattn_node_affine_backward: D %d", (int)D);
  b3d_mlp_desc d;
  xattn_desc(D, &d);
  const b3d_linear layers[2] = {{att->in_proj_weight + 2 * (size_t)D * D, att->in_proj_bias + 2 * (size_t)D},
                                {att->out_proj_weight, att->out_proj_bias}};
  // the query / key projections are dead (softmax over one key): their thirds of the in_proj gradients are exact zeros
  if (grads->in_proj_weight) {
    const long n = 2l * D * D;
    hipLaunchKernelGGL(mlp_zero_kernel, dim3(grid1d(n)), dim3(256), 0, stream, grads->in_proj_weight, n);
  }
  if (grads->in_proj_bias) hipLaunchKernelGGL(mlp_zero_kernel, dim3(1), dim3(256), 0, stream, grads->in_proj_bias, 2l * D);
  B3D_TRY(launch_check("mlp_zero_kernel"));
  const b3d_linear_grad g[2] = {{grads->in_proj_weight ? grads->in_proj_weight + 2 * (size_t)D * D : nullptr,
                                 grads->in_proj_bias ? grads->in_proj_bias + 2 * (size_t)D : nullptr},
                                {grads->out_proj_weight, grads->out_proj_bias}};
  return b3d_mlp_backward(&d, layers, x, y, N, workspace, workspace_bytes, scratch, scratch_bytes, d_y, d_x, g, stream);
}
