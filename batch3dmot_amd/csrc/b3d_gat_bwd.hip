// Backward of the frame-wise k-NN + GATConv block (reference pose_gnn.py:74-80, clr_att_gnn.py:178-184)
// for `knn_writeback=True` -- the mode in which the block's result is USED (the reference computes it and drops it: `x[mask] ==
// x_t` is a comparison).  The neighbour lists are data (arg-min indices): no gradient flows through the selection.
//
//   h = x W^T;  s_q = <h_q, a_src>, d_c = <h_c, a_dst>;  z_cj = s_{q_j} + d_c;  a = leaky_relu(z, 0.2);
//   alpha_c. = softmax_j(a_c.);  y_c = sum_j alpha_cj h_{q_j} + b                        (forward: b3d_knn.hpp gat_aggregate_kernel)
//
//   d alpha_cj = <dy_c, h_{q_j}>;  t_c = sum_j alpha_cj d alpha_cj;  d z_cj = alpha_cj (d alpha_cj - t_c) slope(z_cj)
//   d h_q = sum over edges (c, j) with q_j = q of (alpha_cj dy_c + d z_cj a_src)  +  (sum_j d z_qj) a_dst
//   d a_src = sum_q (sum over edges out of q of d z) h_q;  d a_dst = sum_c (sum_j d z_cj) h_c;  d b = sum_c dy_c
//   d W = d h^T x;  d x = d h W
//
// Every sum runs in a fixed order (no float atomics): the per-source sums walk the CSC lists of the k-NN graph that
// b3d_graph_build produces (segments sorted by edge id), column sums run over node ids in order.
#include "b3d_common.hpp"
#include "b3d_knn.hpp"

namespace b3d {
namespace {

// h = x W^T (no bias), one thread per output element; D <= 96: 3,000 x 96 x 96 MACs
__global__ __launch_bounds__(256) void gatb_lin_kernel(const float* __restrict__ x, const float* __restrict__ W, int N, int D, float* __restrict__ h) {
  const long id = (long)blockIdx.x * 256 + threadIdx.x;
  if (id >= (long)N * D) return;
  const int n = (int)(id / D), o = (int)(id % D);
  float s = 0.f;
  for (int k = 0; k < D; ++k) s = fmaf(x[(size_t)n * D + k], W[(size_t)o * D + k], s);
  h[id] = s;
}

// k-NN edges as an [2, N K] int64 list for b3d_graph_build over N + 1 nodes: slot (c, j) is the edge q_j -> c; empty slots
// (j >= cnt[c]) hang off the dummy node N
__global__ __launch_bounds__(256) void gatb_edges_kernel(const int* __restrict__ nbr, const int* __restrict__ cnt, int N, int K,
                                                         long long* __restrict__ ei) {
  const long id = (long)blockIdx.x * 256 + threadIdx.x;
  const long E = (long)N * K;
  if (id >= E) return;
  const int c = (int)(id / K), j = (int)(id % K);
  const bool live = j < cnt[c];
  ei[id] = live ? nbr[(size_t)c * kKnnMaxK + j] : N;
  ei[E + id] = live ? c : N;
}

// one wavefront per centre: attention coefficients again (as the forward computes them), then d z per edge and its sum
template <int D>
__global__ __launch_bounds__(256) void gatb_center_kernel(const float* __restrict__ h, const float* __restrict__ dy, int N, int K,
                                                          const int* __restrict__ nbr, const int* __restrict__ cnt,
                                                          const float* __restrict__ att_src, const float* __restrict__ att_dst,
                                                          float* __restrict__ e_alpha, float* __restrict__ e_dz, float* __restrict__ dd) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + wave;
  if (c >= N) return;
  const int kk = cnt[c];
  float a = -__builtin_inff(), z = 0.f, dal = 0.f;
  if (lane < kk) {
    const int q = nbr[(size_t)c * kKnnMaxK + lane];
    float ss = 0.f, sd = 0.f;
#pragma unroll
    for (int d = 0; d < D; d += 4) {
      const v4f vq = *reinterpret_cast<const v4f*>(h + (size_t)q * D + d);
      const v4f vc = *reinterpret_cast<const v4f*>(h + (size_t)c * D + d);
      const v4f g = *reinterpret_cast<const v4f*>(dy + (size_t)c * D + d);
      ss = fmaf(vq.x, att_src[d], ss); ss = fmaf(vq.y, att_src[d + 1], ss); ss = fmaf(vq.z, att_src[d + 2], ss); ss = fmaf(vq.w, att_src[d + 3], ss);
      sd = fmaf(vc.x, att_dst[d], sd); sd = fmaf(vc.y, att_dst[d + 1], sd); sd = fmaf(vc.z, att_dst[d + 2], sd); sd = fmaf(vc.w, att_dst[d + 3], sd);
      dal = fmaf(g.x, vq.x, dal); dal = fmaf(g.y, vq.y, dal); dal = fmaf(g.z, vq.z, dal); dal = fmaf(g.w, vq.w, dal);
    }
    z = ss + sd;
    a = z > 0.f ? z : 0.2f * z;
  }
  float m = a;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  const float ex = (lane < kk) ? __expf(a - m) : 0.f;
  float den = ex;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) den += __shfl_xor(den, off, 64);
  const float alpha = ex / (den + 1e-16f);
  float t = alpha * dal;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) t += __shfl_xor(t, off, 64);
  const float dz = (lane < kk) ? alpha * (dal - t) * (z > 0.f ? 1.f : 0.2f) : 0.f;
  float s = dz;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
  if (lane < K) {
    e_alpha[(size_t)c * K + lane] = (lane < kk) ? alpha : 0.f;
    e_dz[(size_t)c * K + lane] = dz;
  }
  if (lane == 0) dd[c] = s;
}

// one wavefront per node q: d h_q over the CSC list of q (edges out of q, ascending edge id = ascending (centre, slot))
template <int D>
__global__ __launch_bounds__(256) void gatb_source_kernel(const float* __restrict__ dy, int N, int K, const int* __restrict__ src_ptr,
                                                          const int* __restrict__ src_perm, const float* __restrict__ e_alpha,
                                                          const float* __restrict__ e_dz, const float* __restrict__ dd,
                                                          const float* __restrict__ att_src, const float* __restrict__ att_dst,
                                                          float* __restrict__ dh, float* __restrict__ ds) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + wave;
  if (q >= N) return;
  constexpr int PER = (D + 63) / 64;
  float acc[PER];
#pragma unroll
  for (int p = 0; p < PER; ++p) acc[p] = 0.f;
  float sz = 0.f;
  const int beg = src_ptr[q], end = src_ptr[q + 1];
  for (int i = beg; i < end; ++i) {
    const int e = src_perm[i];
    const int c = e / K;
    const float al = e_alpha[e], dz = e_dz[e];
    sz += dz;
#pragma unroll
    for (int p = 0; p < PER; ++p) {
      const int d = lane + 64 * p;
      if (d < D) acc[p] = fmaf(al, dy[(size_t)c * D + d], acc[p]);
    }
  }
  const float ddq = dd[q];
#pragma unroll
  for (int p = 0; p < PER; ++p) {
    const int d = lane + 64 * p;
    if (d < D) dh[(size_t)q * D + d] = acc[p] + sz * att_src[d] + ddq * att_dst[d];
  }
  if (lane == 0) ds[q] = sz;
}

// d x = d h W: one thread per element
__global__ __launch_bounds__(256) void gatb_dx_kernel(const float* __restrict__ dh, const float* __restrict__ W, int N, int D, float* __restrict__ dx) {
  const long id = (long)blockIdx.x * 256 + threadIdx.x;
  if (id >= (long)N * D) return;
  const int n = (int)(id / D), k = (int)(id % D);
  float s = 0.f;
  for (int o = 0; o < D; ++o) s = fmaf(dh[(size_t)n * D + o], W[(size_t)o * D + k], s);
  dx[id] = s;
}

// Column reductions over the nodes, in node order, split into kParts row ranges whose partials are added in order:
//   d W [o][k] = sum_n dh[n][o] x[n][k]      (blocks 0 .. D-1: one block per o, one thread per k)
//   d a_src [k] = sum_n ds[n] h[n][k], d a_dst [k] = sum_n dd[n] h[n][k], d b [k] = sum_n dy[n][k]      (block D)
constexpr int kParts = 8;
__global__ __launch_bounds__(128) void gatb_param_partial_kernel(const float* __restrict__ x, const float* __restrict__ h,
                                                                 const float* __restrict__ dh, const float* __restrict__ dy,
                                                                 const float* __restrict__ ds, const float* __restrict__ dd, int N, int D,
                                                                 float* __restrict__ part /* [kParts][(D + 3) * D] */) {
  const int o = blockIdx.x, part_id = blockIdx.y, k = threadIdx.x;
  if (k >= D) return;
  const int per = (N + kParts - 1) / kParts;
  const int n0 = part_id * per, n1 = (n0 + per < N) ? n0 + per : N;
  float* dst = part + (size_t)part_id * (D + 3) * D;
  if (o < D) {
    float s = 0.f;
    for (int n = n0; n < n1; ++n) s = fmaf(dh[(size_t)n * D + o], x[(size_t)n * D + k], s);
    dst[(size_t)o * D + k] = s;
  } else {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int n = n0; n < n1; ++n) {
      const float hv = h[(size_t)n * D + k];
      s0 = fmaf(ds[n], hv, s0);
      s1 = fmaf(dd[n], hv, s1);
      s2 += dy[(size_t)n * D + k];
    }
    dst[(size_t)D * D + k] = s0;
    dst[(size_t)(D + 1) * D + k] = s1;
    dst[(size_t)(D + 2) * D + k] = s2;
  }
}
__global__ __launch_bounds__(256) void gatb_param_finish_kernel(const float* __restrict__ part, int D, float* __restrict__ d_lin,
                                                                float* __restrict__ d_att_src, float* __restrict__ d_att_dst,
                                                                float* __restrict__ d_bias) {
  const int id = blockIdx.x * 256 + threadIdx.x;
  const int total = (D + 3) * D;
  if (id >= total) return;
  float s = 0.f;
  for (int p = 0; p < kParts; ++p) s += part[(size_t)p * total + id];
  if (id < D * D) { if (d_lin) d_lin[id] = s; }
  else if (id < (D + 1) * D) { if (d_att_src) d_att_src[id - D * D] = s; }
  else if (id < (D + 2) * D) { if (d_att_dst) d_att_dst[id - (D + 1) * D] = s; }
  else if (d_bias) d_bias[id - (D + 2) * D] = s;
}

struct GatBwdWs {
  float *h, *dh, *e_alpha, *e_dz, *dd, *ds, *part;
  long long* ei;
  void* gws;
  size_t gws_bytes, bytes;
  bool ok;
};
GatBwdWs gatb_carve(void* ws, size_t ws_bytes, int N, int D, int K) {
  Carver c(ws, ws_bytes);
  GatBwdWs w;
  const size_t n = (size_t)(N > 0 ? N : 1);
  w.h = c.take<float>(n * D); w.dh = c.take<float>(n * D);
  w.e_alpha = c.take<float>(n * K); w.e_dz = c.take<float>(n * K);
  w.dd = c.take<float>(n); w.ds = c.take<float>(n);
  w.part = c.take<float>((size_t)kParts * (D + 3) * D);
  w.ei = c.take<long long>(2 * n * K);
  w.gws_bytes = b3d_graph_workspace_bytes(N + 1, N * K);
  w.gws = c.take<char>(w.gws_bytes);
  w.bytes = c.off + 256;
  w.ok = c.ok();
  return w;
}

}  // namespace
}  // namespace b3d

using namespace b3d;

extern "C" size_t b3d_knn_gat_backward_workspace_bytes(int32_t N, int32_t D, int32_t k) {
  if (N <= 0 || k < 1 || k > kKnnMaxK) return 0;
  return gatb_carve(nullptr, 0, N, D, k).bytes;
}

extern "C" int b3d_knn_gat_backward(const float* x, int32_t N, int32_t D, int32_t k, const b3d_gat* gat, const int32_t* nbr,
                                    const int32_t* cnt, const float* d_y, void* workspace, size_t workspace_bytes, float* d_x,
                                    const b3d_gat_grad* grads, b3d_stream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  B3D_REQUIRE(x && gat && nbr && cnt && d_y && workspace && grads, "b3d_knn_gat_backward: null argument");
  B3D_REQUIRE(gat->lin && gat->att_src && gat->att_dst, "b3d_knn_gat_backward: knn_conv parameters are null");
  B3D_REQUIRE(D == 48 || D == 96, "b3d_knn_gat_backward: D must be 48 or 96, got %d", D);
  B3D_REQUIRE(N > 0 && k >= 1 && k <= kKnnMaxK, "b3d_knn_gat_backward: bad N / k");
  GatBwdWs w = gatb_carve(workspace, workspace_bytes, N, D, k);
  if (!w.ok) return fail(B3D_ERR_WORKSPACE, "b3d_knn_gat_backward: workspace %zu < %zu bytes", workspace_bytes, w.bytes);
  const long nd = (long)N * D, ne = (long)N * k;
  hipLaunchKernelGGL(gatb_lin_kernel, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, stream, x, gat->lin, N, D, w.h);
  B3D_TRY(launch_check("gatb_lin_kernel"));
  hipLaunchKernelGGL(gatb_edges_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, stream, (const int*)nbr, (const int*)cnt, N, k, w.ei);
  B3D_TRY(launch_check("gatb_edges_kernel"));
  b3d_graph g;
  B3D_TRY(b3d_graph_build((const int64_t*)w.ei, N + 1, (int32_t)ne, w.gws, w.gws_bytes, &g, stream_));
  if (D == 48) {
    hipLaunchKernelGGL(gatb_center_kernel<48>, dim3((N + 3) / 4), dim3(256), 0, stream, w.h, d_y, N, k, (const int*)nbr, (const int*)cnt,
                       gat->att_src, gat->att_dst, w.e_alpha, w.e_dz, w.dd);
    B3D_TRY(launch_check("gatb_center_kernel"));
    hipLaunchKernelGGL(gatb_source_kernel<48>, dim3((N + 3) / 4), dim3(256), 0, stream, d_y, N, k, g.src_ptr, g.src_perm, w.e_alpha, w.e_dz,
                       w.dd, gat->att_src, gat->att_dst, w.dh, w.ds);
  } else {
    hipLaunchKernelGGL(gatb_center_kernel<96>, dim3((N + 3) / 4), dim3(256), 0, stream, w.h, d_y, N, k, (const int*)nbr, (const int*)cnt,
                       gat->att_src, gat->att_dst, w.e_alpha, w.e_dz, w.dd);
    B3D_TRY(launch_check("gatb_center_kernel"));
    hipLaunchKernelGGL(gatb_source_kernel<96>, dim3((N + 3) / 4), dim3(256), 0, stream, d_y, N, k, g.src_ptr, g.src_perm, w.e_alpha, w.e_dz,
                       w.dd, gat->att_src, gat->att_dst, w.dh, w.ds);
  }
  B3D_TRY(launch_check("gatb_source_kernel"));
  if (d_x) {
    hipLaunchKernelGGL(gatb_dx_kernel, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, stream, w.dh, gat->lin, N, D, d_x);
    B3D_TRY(launch_check("gatb_dx_kernel"));
  }
  hipLaunchKernelGGL(gatb_param_partial_kernel, dim3(D + 1, kParts), dim3(128), 0, stream, x, w.h, w.dh, d_y, w.ds, w.dd, N, D, w.part);
  B3D_TRY(launch_check("gatb_param_partial_kernel"));
  hipLaunchKernelGGL(gatb_param_finish_kernel, dim3(((D + 3) * D + 255) / 256), dim3(256), 0, stream, w.part, D, grads->lin, grads->att_src,
                     grads->att_dst, grads->bias);
  return launch_check("gatb_param_finish_kernel");
}
