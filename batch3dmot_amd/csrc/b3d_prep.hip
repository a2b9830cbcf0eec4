// Graph structure (int64 COO -> int32 src/dst + CSR by destination + CSC by source) and weight
// image packing.  Small integer / copy kernels; HBM- and latency-bound, nothing to tile.
#include "b3d_common.hpp"
#include "b3d_dev.hpp"
#include "b3d_pack.hpp"

namespace b3d {

// ---- graph ------------------------------------------------------------------------------------
__global__ void graph_convert_count(const int64_t* __restrict__ ei, int E, int N, int* __restrict__ src,
                                    int* __restrict__ dst, int* __restrict__ cnt_dst,
                                    int* __restrict__ cnt_src, int* __restrict__ bad) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= E) return;
  const long s = ei[k], d = ei[(long)E + k];
  long ss = s, dd = d;
  // bad[1] = b3d_graph.dst_unsorted: the edges are not grouped by destination (or one of them is invalid)
  if (k + 1 < E && ei[(long)E + k + 1] < d) atomicOr(bad + 1, 1);
  if (s < 0 || s >= N || d < 0 || d >= N) {
    atomicOr(bad + 1, 1);
    // the reference raises an index error for this input (pose_gnn.py:180); here the edge is counted in `bad`
    // (b3d_graph.invalid_edges, which the caller turns into that error) and rewritten to the self loop (0, 0), which
    // keeps CSR / CSC consistent: nothing downstream can index out of bounds before the caller has looked
    atomicAdd(bad, 1);
    ss = 0; dd = 0;
    if (N <= 0) { src[k] = 0; dst[k] = 0; return; }
  }
  src[k] = (int)ss;
  dst[k] = (int)dd;
  atomicAdd(&cnt_dst[dd], 1);
  atomicAdd(&cnt_src[ss], 1);
}

// Exclusive scan of two count arrays (one workgroup each): ptr[0..N], cursor copy for the fill.
__global__ __launch_bounds__(1024) void graph_scan(const int* __restrict__ cnt_dst, const int* __restrict__ cnt_src,
                                                   int N, int* __restrict__ dst_ptr, int* __restrict__ src_ptr,
                                                   int* __restrict__ cur_dst, int* __restrict__ cur_src) {
  const int* cnt = blockIdx.x == 0 ? cnt_dst : cnt_src;
  int* ptr = blockIdx.x == 0 ? dst_ptr : src_ptr;
  int* cur = blockIdx.x == 0 ? cur_dst : cur_src;
  __shared__ int part[1024];
  const int per = (N + 1023) / 1024;
  const int b = threadIdx.x * per;
  int s = 0;
  for (int i = b; i < b + per && i < N; ++i) s += cnt[i];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    int v = (threadIdx.x >= off) ? part[threadIdx.x - off] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  int run = part[threadIdx.x] - s;
  for (int i = b; i < b + per && i < N; ++i) {
    ptr[i] = run;
    cur[i] = run;
    run += cnt[i];
  }
  if (threadIdx.x == 1023) ptr[N] = part[1023];
}

__global__ void graph_fill(const int* __restrict__ src, const int* __restrict__ dst, int E, int N,
                           int* __restrict__ cur_dst, int* __restrict__ cur_src,
                           int* __restrict__ dst_perm, int* __restrict__ src_perm) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= E) return;
  if (N <= 0) { dst_perm[k] = k; src_perm[k] = k; return; }        // every edge invalid: identity, flagged in `bad`
  dst_perm[atomicAdd(&cur_dst[dst[k]], 1)] = k;
  src_perm[atomicAdd(&cur_src[src[k]], 1)] = k;
}

// The atomic fill leaves each segment in arrival order; sort every segment by edge id so that the
// summation order of all segment sums is fixed (bitwise reproducible results).  One wavefront per
// list: lists of <= 64 edges are rank-sorted with wavefront shuffles, longer ones by lane 0.
__global__ __launch_bounds__(256) void graph_sort_segments(const int* __restrict__ dst_ptr, const int* __restrict__ src_ptr,
                                                           int N, int* __restrict__ dst_perm, int* __restrict__ src_perm) {
  const int w = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (w >= 2 * N) return;
  const int n = w >> 1;
  const int* ptr = (w & 1) ? src_ptr : dst_ptr;
  int* perm = (w & 1) ? src_perm : dst_perm;
  const int b = ptr[n], e = ptr[n + 1], len = e - b;
  if (len <= 1) return;
  if (len <= 64) {
    const int v = (lane < len) ? perm[b + lane] : 0x7fffffff;
    int rank = 0;
    for (int j = 0; j < len; ++j) rank += (__shfl(v, j, 64) < v);
    if (lane < len) perm[b + rank] = v;
  } else if (lane == 0) {
    for (int i = b + 1; i < e; ++i) {
      const int v = perm[i];
      int j = i - 1;
      while (j >= b && perm[j] > v) { perm[j + 1] = perm[j]; --j; }
      perm[j + 1] = v;
    }
  }
}

// The rows of `past` the node kernel sums per node (b3d.h: b3d_graph.past_ptr / past_rows): with edges grouped by destination, the
// last edge of every (destination, aligned 16-edge block) run -- the edge kernel has added the run there --, otherwise every edge of the
// destination's list.  One workgroup: a scan over the nodes in chunks of 1,024.
constexpr int kPastBlock = 16;                  // rows of a wavefront of the edge kernels (b3d_estream.hpp)
__global__ __launch_bounds__(1024) void graph_past_lists(const int* __restrict__ dst_ptr, const int* __restrict__ dst_perm, int N,
                                                         const int* __restrict__ unsorted, int* __restrict__ past_ptr,
                                                         int* __restrict__ past_rows) {
  __shared__ int part[1024];
  __shared__ int base_s;
  const int tid = threadIdx.x;
  const bool tails = *unsorted == 0;
  if (tid == 0) base_s = 0;
  __syncthreads();
  for (int n0 = 0; n0 < N; n0 += 1024) {
    const int n = n0 + tid;
    int b = 0, e = 0, c = 0;
    if (n < N) {
      b = dst_ptr[n]; e = dst_ptr[n + 1];
      c = tails ? (e > b ? (e - 1) / kPastBlock - b / kPastBlock + 1 : 0) : e - b;
    }
    part[tid] = c;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const int v = tid >= off ? part[tid - off] : 0;
      __syncthreads();
      part[tid] += v;
      __syncthreads();
    }
    int o = base_s + part[tid] - c;
    if (n < N) {
      past_ptr[n] = o;
      if (tails) {
        for (int t = b / kPastBlock; e > b && t <= (e - 1) / kPastBlock; ++t) {
          const int last = kPastBlock * t + kPastBlock - 1;
          past_rows[o++] = last < e - 1 ? last : e - 1;    // (grouped by destination: the list's edge ids are b .. e - 1)
        }
      } else {
        for (int i = b; i < e; ++i) past_rows[o++] = dst_perm[i];
      }
    }
    __syncthreads();
    if (tid == 1023) base_s += part[1023];
    __syncthreads();
  }
  if (tid == 0) past_ptr[N] = base_s;
}

struct GraphLayout {
  int *src, *dst, *dst_ptr, *dst_perm, *src_ptr, *src_perm, *cnt_dst, *cnt_src, *cur_dst, *cur_src, *bad, *past_ptr, *past_rows;
  size_t bytes;
  bool ok;
};

static GraphLayout graph_layout(void* ws, size_t ws_bytes, int N, int E) {
  Carver c(ws, ws_bytes);
  GraphLayout g;
  g.src = c.take<int>(E > 0 ? E : 1);
  g.dst = c.take<int>(E > 0 ? E : 1);
  g.dst_ptr = c.take<int>(N + 1);
  g.src_ptr = c.take<int>(N + 1);
  g.dst_perm = c.take<int>(E > 0 ? E : 1);
  g.src_perm = c.take<int>(E > 0 ? E : 1);
  // zero-initialised block: counts + bad flag, contiguous
  g.cnt_dst = c.take<int>(2 * (size_t)N + 64);
  g.cnt_src = g.cnt_dst ? g.cnt_dst + N : nullptr;
  g.bad = g.cnt_dst ? g.cnt_dst + 2 * (size_t)N : nullptr;
  g.cur_dst = c.take<int>(N > 0 ? N : 1);
  g.cur_src = c.take<int>(N > 0 ? N : 1);
  g.past_ptr = c.take<int>(N + 1);
  g.past_rows = c.take<int>((size_t)(E > 0 ? E : 1) + (size_t)N + 16);       // >= max(E, E / 16 + N)
  g.bytes = c.off + 256;
  g.ok = c.ok();
  return g;
}

}  // namespace b3d

using namespace b3d;

extern "C" size_t b3d_graph_workspace_bytes(int32_t N, int32_t E) {
  return graph_layout(nullptr, 0, N, E).bytes;
}

extern "C" int b3d_graph_build(const int64_t* edge_index, int32_t N, int32_t E, void* workspace,
                               size_t workspace_bytes, b3d_graph* out, b3d_stream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  B3D_REQUIRE(out != nullptr && workspace != nullptr, "b3d_graph_build: null argument");
  B3D_REQUIRE(N >= 0 && E >= 0, "b3d_graph_build: negative size");
  B3D_REQUIRE(E == 0 || edge_index != nullptr, "b3d_graph_build: edge_index is null");
  GraphLayout g = graph_layout(workspace, workspace_bytes, N, E);
  if (!g.ok) return fail(B3D_ERR_WORKSPACE, "b3d_graph_build: workspace %zu < %zu bytes", workspace_bytes, g.bytes);
  B3D_HIP_CHECK(hipMemsetAsync(g.cnt_dst, 0, (2 * (size_t)N + 64) * sizeof(int), stream));
  if (E > 0) {
    hipLaunchKernelGGL(graph_convert_count, dim3((E + 255) / 256), dim3(256), 0, stream, edge_index, E, N,
                       g.src, g.dst, g.cnt_dst, g.cnt_src, g.bad);
    B3D_TRY(launch_check("graph_convert_count"));
  }
  hipLaunchKernelGGL(graph_scan, dim3(2), dim3(1024), 0, stream, g.cnt_dst, g.cnt_src, N, g.dst_ptr,
                     g.src_ptr, g.cur_dst, g.cur_src);
  B3D_TRY(launch_check("graph_scan"));
  if (E > 0) {
    hipLaunchKernelGGL(graph_fill, dim3((E + 255) / 256), dim3(256), 0, stream, g.src, g.dst, E, N, g.cur_dst,
                       g.cur_src, g.dst_perm, g.src_perm);
    B3D_TRY(launch_check("graph_fill"));
    hipLaunchKernelGGL(graph_sort_segments, dim3((2 * N + 3) / 4), dim3(256), 0, stream, g.dst_ptr,
                       g.src_ptr, N, g.dst_perm, g.src_perm);
    B3D_TRY(launch_check("graph_sort_segments"));
  }
  hipLaunchKernelGGL(graph_past_lists, dim3(1), dim3(1024), 0, stream, g.dst_ptr, g.dst_perm, N, g.bad + 1, g.past_ptr, g.past_rows);
  B3D_TRY(launch_check("graph_past_lists"));
  out->N = N;
  out->E = E;
  out->src = g.src;
  out->dst = g.dst;
  out->dst_ptr = g.dst_ptr;
  out->dst_perm = g.dst_perm;
  out->src_ptr = g.src_ptr;
  out->src_perm = g.src_perm;
  out->invalid_edges = g.bad;
  out->dst_unsorted = g.bad + 1;
  out->past_ptr = g.past_ptr;
  out->past_rows = g.past_rows;
  return B3D_OK;
}

// ---- weight images -----------------------------------------------------------------------------
namespace b3d {

__global__ void pack_kernel(const PackArgs a) {
  // one workgroup column per descriptor (blockIdx.y), grid-stride over the image's floats
  const PackDesc& d = a.d[blockIdx.y];
  if (d.transposed >= 2) {                      // index / zero fill
    int* p = reinterpret_cast<int*>(d.dst);
    for (int id = blockIdx.x * blockDim.x + threadIdx.x; id < d.N; id += gridDim.x * blockDim.x)
      p[id] = (d.transposed == 2) ? id : 0;
    return;
  }
  const bool bf = d.bf != 0;
  const int stride = row_stride(d.KP, bf);
  const int cr = chunk_rows(d.KP, d.NP, bf);
  auto chunk_base = [&](int r, int& rc) -> size_t {          // chunk ch holds image rows [ch*cr, ...), each chunk padded to 1 KB
    const int ch = r / cr;
    rc = r - ch * cr;
    size_t off = 0;
    for (int i = 0; i < ch; ++i) off += chunk_floats(d.KP, bf, chunk_nrows(d.KP, d.NP, bf, i));
    return off;
  };
  auto sign_of = [&](int rl) -> float { return (d.row_sign && !d.transposed && rl < d.N && d.row_sign[rl] < 0.f) ? -1.f : 1.f; };
  auto value = [&](int rl, int c) -> float {                 // element (rl, c) of the slice, zero outside [N, K]
    if (rl >= d.N || c >= d.K || !d.w) return 0.f;
    return d.transposed ? d.w[(size_t)c * d.ld + rl] : d.w[(size_t)rl * d.ld + c] * sign_of(rl);
  };
  if (bf) {
    // bf16x3 image: one thread per PAIR of adjacent columns (they share a dword in each of the three pieces) + one
    // per bias.  A partial descriptor covers columns [col0, col0 + K): col0 and K even.
    const int c0 = d.partial ? d.col0 : 0;
    const int ncol = d.partial ? d.K : d.KP;                 // columns this descriptor writes
    const int span = ncol / 2 + (d.partial ? 0 : 8);         // + bias and row padding (zeroed)
    const int total = d.nrows * span;
    unsigned* dst = reinterpret_cast<unsigned*>(d.dst);
    for (int id = blockIdx.x * blockDim.x + threadIdx.x; id < total; id += gridDim.x * blockDim.x) {
      const int rl = id / span, t = id - rl * span;
      int rc;
      const size_t base = chunk_base(d.row0 + rl, rc) + (size_t)rc * stride;
      if (t >= ncol / 2) {                                   // bias (fp32) and padding
        const int k = t - ncol / 2;
        float v = 0.f;
        if (k == 0 && rl < d.N && d.b && !d.transposed) v = d.b[rl] * sign_of(rl);
        dst[base + 3 * d.KP / 2 + k] = __float_as_uint(v);
        continue;
      }
      const int c = 2 * t;                                   // slice columns c, c + 1 -> image columns c0 + c, c0 + c + 1
      unsigned pc[3] = {0u, 0u, 0u};
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float x = value(rl, c + e);
        const unsigned xb = __float_as_uint(x);
        const float r1 = x - __uint_as_float(xb & 0xffff0000u);
        const unsigned mb = __float_as_uint(r1);
        const float r2 = r1 - __uint_as_float(mb & 0xffff0000u);
        pc[0] |= (xb >> 16) << (16 * e);
        pc[1] |= (mb >> 16) << (16 * e);
        pc[2] |= (__float_as_uint(r2) >> 16) << (16 * e);
      }
      const int col = c0 + c;
      const int pos = 32 * (col / 32) + bf_pos(col % 32);    // even: the pair (pos, pos + 1) is one dword
#pragma unroll
      for (int p = 0; p < 3; ++p) dst[base + p * (d.KP / 2) + pos / 2] = pc[p];
    }
    return;
  }
  const int span = d.partial ? d.K : stride;                 // columns this descriptor writes per row
  const int total = d.nrows * span;
  for (int id = blockIdx.x * blockDim.x + threadIdx.x; id < total; id += gridDim.x * blockDim.x) {
    const int rl = id / span, c = id - rl * span;            // row / column within the slice
    float v = 0.f;
    if (rl < d.N && d.w) {
      if (!d.transposed) {
        if (c < d.K) v = d.w[(size_t)rl * d.ld + c] * sign_of(rl);
        else if (c == d.KP && d.b) v = d.b[rl] * sign_of(rl);
      } else if (c < d.K) {
        // image of W^T: rows = input features of the forward layer, cols = its outputs
        v = d.w[(size_t)c * d.ld + rl];
      }
    }
    int rc;
    const size_t off = chunk_base(d.row0 + rl, rc);
    d.dst[off + (size_t)rc * stride + d.col0 + c] = v;
  }
}

// Fragment-stream image of one layer (b3d_estream.hpp): steps in execution order (for p: for c), a step = the output blocks
// 2 p and 2 p + 1 against the inputs [32 c, 32 c + 32), each block three 1 KB fragments (bf16 pieces 0, 1, 2), a fragment = 64
// lanes x 8 bf16: lane (m = l & 15, q = l >> 4), element j <-> W[16 ob + m][32 c + 16 (j >> 2) + 4 q + (j & 3)].
// One thread per pair of adjacent elements (one dword in each of the three pieces).
__global__ __launch_bounds__(256) void pack_frag_kernel(const FragArgs a) {
  const FragDesc& d = a.d[blockIdx.y];
  const int KS = d.K / 32;
  const int total = d.N * d.K / 2;
  unsigned* dst = reinterpret_cast<unsigned*>(d.steps);
  for (int t = blockIdx.x * 256 + threadIdx.x; t < total; t += gridDim.x * 256) {
    const int step = t >> 9, r = t & 511;
    const int half = r >> 8, lane = (r & 255) >> 2, jp = r & 3;
    const int p = step / KS, c = step - p * KS;
    const int row = 32 * p + 16 * half + (lane & 15);
    unsigned pc[3] = {0u, 0u, 0u};
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int j = 2 * jp + e;
      const int col = 32 * c + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3);
      const float x = d.transposed ? d.w[(size_t)col * d.ld + row] : d.w[(size_t)row * d.ld + col];
#if defined(B3D_ES_F16) && B3D_ES_F16
      const _Float16 hi = (_Float16)x;
      const _Float16 lo = (_Float16)(x - (float)hi);
      pc[0] |= (unsigned)__builtin_bit_cast(unsigned short, hi) << (16 * e);
      pc[1] |= (unsigned)__builtin_bit_cast(unsigned short, lo) << (16 * e);
#else
      const unsigned xb = __float_as_uint(x);
      const float r1 = x - __uint_as_float(xb & 0xffff0000u);
      const unsigned mb = __float_as_uint(r1);
      const float r2 = r1 - __uint_as_float(mb & 0xffff0000u);
      pc[0] |= (xb >> 16) << (16 * e);
      pc[1] |= (mb >> 16) << (16 * e);
      pc[2] |= (__float_as_uint(r2) >> 16) << (16 * e);
#endif
    }
    const size_t o = (size_t)step * 1536 + half * 768 + lane * 4 + jp;     // dwords
#pragma unroll
    for (int q = 0; q < 3; ++q) dst[o + q * 256] = pc[q];
  }
  if (blockIdx.x == 0)
    for (int n = threadIdx.x; n < d.N; n += 256) d.bias[n] = d.b ? d.b[n] : 0.f;
}

int pack_frags(const FragDesc* descs, int n, hipStream_t stream) {
  if (n > kFragMax) return fail(B3D_ERR_ARG, "fragment pack table overflow");
  if (n == 0) return B3D_OK;
  FragArgs a;
  a.n = n;
  for (int i = 0; i < n; ++i) a.d[i] = descs[i];
  hipLaunchKernelGGL(pack_frag_kernel, dim3(32, n), dim3(256), 0, stream, a);
  return launch_check("pack_frag_kernel");
}

int pack_images(const PackDesc* descs, int n, hipStream_t stream) {
  for (int i0 = 0; i0 < n; i0 += kPackMax) {
    PackArgs a;
    a.n = (n - i0 < kPackMax) ? n - i0 : kPackMax;
    int maxtot = 0;
    for (int i = 0; i < a.n; ++i) {
      a.d[i] = descs[i0 + i];
      const int t = (a.d[i].transposed >= 2) ? a.d[i].N : a.d[i].nrows * (a.d[i].partial ? a.d[i].K : row_stride(a.d[i].KP, a.d[i].bf != 0));
      if (t > maxtot) maxtot = t;
    }
    int gx = (maxtot + 255) / 256;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(pack_kernel, dim3(gx, a.n), dim3(256), 0, stream, a);
    B3D_TRY(launch_check("pack_kernel"));
  }
  return B3D_OK;
}

}  // namespace b3d
