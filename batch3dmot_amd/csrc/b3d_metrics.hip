// Average precision of the edge scores, overall and per edge class (reference train.py:18,143-150 and :188-196:
// torchmetrics.functional average_precision(out, gt, pos_label=1) on the whole batch and on out[edge_classes == c]).
//
// torchmetrics is a third-party dependency of the reference that this repository does not vendor; its published
// algorithm for the binary case (_binary_clf_curve + _average_precision_compute_with_precision_recall): sort the
// scores descending, one curve point per DISTINCT score (a tie group ends at its last element), tps = cumulative
// positives, precision = tps / rank, recall = tps / total positives, AP = sum over points (R_n - R_{n-1}) P_n.
// The points after full recall add nothing and the appended (P = 1, R = 0) end point makes R_0 = 0, so
//      AP = (1 / positives) * sum over tie groups g of  positives(g) * tps(end of g) / rank(end of g),
// NaN when the set holds no positive (0 / 0 in torchmetrics).  sklearn.metrics.average_precision_score is the same
// sum; tests/test_metrics_hip.py checks both the numpy restatement in oracle/ and sklearn.
//
// One stable device radix sort orders every set at once: each edge is entered twice, under key
// (set << 32 | descending-score bits) with set 0 = all edges and set c = its class; two scans (cumulative positives,
// start of the current tie group) and one fixed-order reduction per set finish it.  float64 throughout, no atomics.
#include <string.h>
#include "b3d_common.hpp"
#include <rocprim/rocprim.hpp>

namespace b3d {
namespace {

__device__ __forceinline__ unsigned desc_bits(float s) {
  unsigned u = __float_as_uint(s);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);          // ascending-orderable
  return ~u;                                               // descending
}

__global__ void ap_keys_kernel(const float* __restrict__ scores, const void* __restrict__ y, int y_is_int64,
                               const float* __restrict__ edge_classes, long long E, int C, long long* key, int* val) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= E) return;
  const unsigned d = desc_bits(scores[i] + 0.0f);        // -0.0 and +0.0 are one threshold
  const int pos = y_is_int64 ? (((const long long*)y)[i] == 1) : (((const float*)y)[i] == 1.0f);
  key[i] = (long long)d;                                   // set 0: every edge
  val[i] = pos;
  if (edge_classes) {
    const float cf = edge_classes[i];
    const int c = (int)cf;
    const bool in_range = (float)c == cf && c >= 1 && c <= C;      // the reference compares edge_classes == cls_idx
    key[E + i] = ((long long)(in_range ? c : C + 1) << 32) | (long long)d;
    val[E + i] = pos;
  }
}
// group_start_mark[i] = i where a tie group starts, else 0 (max-scanned into "start of the group that holds i");
// seg_start[s] = first sorted position of set s.
__global__ void ap_marks_kernel(const long long* __restrict__ skey, long long M, int* mark, int* seg_start) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  const bool head = (i == 0) || skey[i] != skey[i - 1];
  mark[i] = head ? (int)i : 0;
  const int s = (int)(skey[i] >> 32);
  if (i == 0 || (int)(skey[i - 1] >> 32) != s) seg_start[s] = (int)i;
}
__global__ void ap_init_kernel(int* seg_start, int n, int fill) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) seg_start[i] = fill;
}
// One workgroup per set: AP of the set = sum over tie-group ends of positives(group) * tps / rank, / positives.
__global__ __launch_bounds__(256) void ap_reduce_kernel(const long long* __restrict__ skey, const int* __restrict__ cpos,
                                                        const int* __restrict__ gstart, const int* __restrict__ seg_start,
                                                        long long M, int C, double* ap, int* count) {
  const int s = blockIdx.x;                                // 0 .. C
  __shared__ double red[256];
  const int beg = seg_start[s];
  double part = 0.0;
  int end = beg;
  if (beg >= 0) {
    // the set ends where the next non-empty set starts
    end = (int)M;
    for (int t = s + 1; t <= C + 1; ++t) {
      if (seg_start[t] >= 0) { end = seg_start[t]; break; }
    }
    const int base = beg > 0 ? cpos[beg - 1] : 0;          // positives in front of the set
    for (int i = beg + (int)threadIdx.x; i < end; i += 256) {
      const bool last_of_group = (i + 1 == end) || skey[i + 1] != skey[i];
      if (!last_of_group) continue;
      const int gs = gstart[i];
      const int gpos = cpos[i] - (gs > 0 ? cpos[gs - 1] : 0);
      if (gpos > 0) part += (double)gpos * (double)(cpos[i] - base) / (double)(i - beg + 1);
    }
  }
  red[threadIdx.x] = part;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {                      // fixed tree: bitwise reproducible
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int n = end - beg, positives = (beg >= 0 && n > 0) ? cpos[end - 1] - (beg > 0 ? cpos[beg - 1] : 0) : 0;
    count[s] = beg >= 0 ? n : 0;
    ap[s] = positives > 0 ? red[0] / (double)positives : __longlong_as_double(0x7ff8000000000000ll);
  }
}

// ---- direct form for training-batch sizes (round 5) --------------------------------------------------------------------------
// The sum above, regrouped per POSITIVE: every positive p of a tie group contributes tps(end of its group) / rank(end of its group)
//      = #{positives j of the set : s_j >= s_p} / #{edges j of the set : s_j >= s_p},
// so AP(set) = (1 / positives) * sum over its positives of that ratio -- no sort, no scan: two counts per positive over the set's
// scores.  A training batch has ~31,000 edges and ~1,500 positives (5 * 10^7 comparisons: microseconds of vector work on scores that
// stay in L2), where the radix-sort form costs ~25 rocPRIM launches, 0.3 ms inside a 4.2 ms step (bench.py: clr_with_ap_metrics).
// Four launches: pack (label + class per edge), compaction of the positives (one workgroup, ballots), counts (one wavefront per pair
// of positives, every count exact in integers), a fixed-tree sum per set (float64): bitwise reproducible.  Used for E <= kApDirectMaxE;
// larger inputs keep the sort.
constexpr long long kApDirectMaxE = 131072;

// meta[j]: bit 15 = positive, low byte = class in 1..C (0: outside every per-class set)
__global__ void ap_pack_kernel(const void* __restrict__ y, int y_is_int64, const float* __restrict__ edge_classes, long long E, int C,
                               unsigned short* __restrict__ meta) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= E) return;
  const int pos = y_is_int64 ? (((const long long*)y)[i] == 1) : (((const float*)y)[i] == 1.0f);
  int c = 0;
  if (edge_classes) {
    const float cf = edge_classes[i];
    const int ci = (int)cf;
    if ((float)ci == cf && ci >= 1 && ci <= C) c = ci;             // the reference compares edge_classes == cls_idx
  }
  meta[i] = (unsigned short)((pos ? 0x8000 : 0) | c);
}

// ascending indices of the positives + their number (one workgroup; per 4,096 edges: four ballots per wavefront, one LDS scan over
// the 64 (sub-block, wavefront) counts, two barriers)
__global__ __launch_bounds__(1024) void ap_compact_kernel(const unsigned short* __restrict__ meta, long long E, int* __restrict__ pos_idx,
                                                          int* __restrict__ count) {
  __shared__ int wsum[4][16];
  __shared__ int base_s;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (tid == 0) base_s = 0;
  __syncthreads();
  for (long long n0 = 0; n0 < E; n0 += 4096) {
    bool f[4];
    int below[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long n = n0 + 1024 * u + tid;
      f[u] = n < E && (meta[n < E ? n : 0] & 0x8000u) != 0;
      const unsigned long long bb = __ballot(f[u]);
      below[u] = __popcll(bb & ((1ull << lane) - 1ull));
      if (lane == 0) wsum[u][wave] = __popcll(bb);
    }
    __syncthreads();
    int off = base_s, run = 0;
    int mine[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int w = 0; w < 16; ++w) {
        if (w == wave) mine[u] = run;
        run += wsum[u][w];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (f[u]) pos_idx[off + mine[u] + below[u]] = (int)(n0 + 1024 * u + tid);
    __syncthreads();
    if (tid == 0) base_s = off + run;
    __syncthreads();
  }
  if (tid == 0) *count = base_s;
}

// One wavefront per positive, sixteen positives per workgroup pass; the set's scores and labels go through LDS in tiles of 16,384
// edges (every wavefront of the workgroup reads the same tile: one trip to L2 per tile and workgroup instead of one per wavefront
// and 256 edges -- the first form of this kernel was a chain of 122 dependent L2 round trips, 146 us).
constexpr int kApTile = 16384, kApWaves = 16;
__global__ __launch_bounds__(kApWaves * 64) void ap_direct_kernel(const float* __restrict__ scores, const unsigned short* __restrict__ meta,
                                                                    long long E, const int* __restrict__ pos_idx, const int* __restrict__ count,
                                                                    double* __restrict__ term_all, double* __restrict__ term_cls) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ap_lds[];
  float* const ts = reinterpret_cast<float*>(ap_lds);                                   // [kApTile]
  unsigned short* const tm = reinterpret_cast<unsigned short*>(ap_lds + kApTile * 4);   // [kApTile]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int P = *count;
  const int groups = (P + kApWaves - 1) / kApWaves;
  for (int g = blockIdx.x; g < groups; g += gridDim.x) {
    const int k = g * kApWaves + wave;
    const bool live = k < P;
    const int pi = pos_idx[live ? k : P - 1];
    const float sp = scores[pi];
    const unsigned cls = meta[pi] & 0xffu;
    int ca = 0, cp = 0, cca = 0, ccp = 0;
    for (long long t0 = 0; t0 < E; t0 += kApTile) {
      __syncthreads();                                         // the previous tile has been consumed
      const int n = (int)((E - t0) < kApTile ? (E - t0) : kApTile);
      for (int i0 = 0; i0 < n; i0 += 4 * kApWaves * 64) {   // eight independent loads per thread and round trip
        float v[4];
        unsigned short mm[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + u * kApWaves * 64 + tid;
          const long long j = t0 + (i < n ? i : 0);
          v[u] = scores[j]; mm[u] = meta[j];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + u * kApWaves * 64 + tid;
          if (i < n) { ts[i] = v[u]; tm[i] = mm[u]; }
        }
      }
      __syncthreads();
      for (int j0 = 0; j0 < n; j0 += 256) {                    // four independent LDS reads per lane
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int j = j0 + 64 * u + lane;
          const bool in = j < n;
          const float sj = ts[in ? j : 0];
          const unsigned m = tm[in ? j : 0];
          const int pos = (m >> 15) & 1;
          const int ge = (in && sj >= sp) ? 1 : 0;
          const int same = (ge && cls != 0 && (m & 0xffu) == cls) ? 1 : 0;
          ca += ge; cp += ge & pos;
          cca += same; ccp += same & pos;
        }
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      ca += __shfl_xor(ca, off); cp += __shfl_xor(cp, off);
      cca += __shfl_xor(cca, off); ccp += __shfl_xor(ccp, off);
    }
    if (lane == 0 && live) {
      term_all[k] = (double)cp / (double)ca;                                  // ca >= 1: the positive itself
      term_cls[k] = cls != 0 ? (double)ccp / (double)cca : 0.0;
    }
  }
}

// one workgroup per set s (0 = every edge, c = class c): sum of its positives' terms (fixed tree), its positives, its edges
__global__ __launch_bounds__(256) void ap_direct_reduce_kernel(const unsigned short* __restrict__ meta, long long E, const int* __restrict__ pos_idx,
                                                               const int* __restrict__ count, const double* __restrict__ term_all,
                                                               const double* __restrict__ term_cls, double* ap, int* cnt) {
  const int s = blockIdx.x;
  __shared__ double red[256];
  __shared__ int redp[256], rede[256];
  const int P = *count;
  double part = 0.0;
  int npos = 0, nedge = 0;
  for (int k = threadIdx.x; k < P; k += 256) {
    if (s == 0) { part += term_all[k]; ++npos; }
    else if ((int)(meta[pos_idx[k]] & 0xffu) == s) { part += term_cls[k]; ++npos; }
  }
  if (s == 0) nedge = threadIdx.x == 0 ? (int)E : 0;
  else {
    for (long long j0 = 0; j0 < E; j0 += 2048) {             // eight independent loads per thread and round trip
      unsigned m[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const long long j = j0 + 256 * u + threadIdx.x; m[u] = j < E ? (unsigned)meta[j] : 0u; }
#pragma unroll
      for (int u = 0; u < 8; ++u) nedge += ((int)(m[u] & 0xffu) == s) ? 1 : 0;
    }
  }
  red[threadIdx.x] = part; redp[threadIdx.x] = npos; rede[threadIdx.x] = nedge;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {                      // fixed tree: bitwise reproducible
    if ((int)threadIdx.x < w) {
      red[threadIdx.x] += red[threadIdx.x + w]; redp[threadIdx.x] += redp[threadIdx.x + w]; rede[threadIdx.x] += rede[threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    cnt[s] = rede[0];
    ap[s] = redp[0] > 0 ? red[0] / (double)redp[0] : __longlong_as_double(0x7ff8000000000000ll);
  }
}

struct ApWs {
  long long *key, *skey;
  int *val, *sval, *cpos, *mark, *gstart, *seg_start;
  void* tmp;
  size_t tmp_bytes, bytes;
  bool ok;
};
size_t ap_tmp_bytes(long long M) {
  size_t a = 0, b = 0, c = 0;
  (void)rocprim::radix_sort_pairs(nullptr, a, (long long*)nullptr, (long long*)nullptr, (int*)nullptr, (int*)nullptr, (size_t)M, 0, 40, nullptr);
  (void)rocprim::inclusive_scan(nullptr, b, (int*)nullptr, (int*)nullptr, (size_t)M, rocprim::plus<int>(), nullptr);
  (void)rocprim::inclusive_scan(nullptr, c, (int*)nullptr, (int*)nullptr, (size_t)M, rocprim::maximum<int>(), nullptr);
  size_t m = a > b ? a : b;
  return (m > c ? m : c) + 256;
}
void ap_carve(ApWs& w, void* p, size_t bytes, long long E, int C) {
  Carver c(p, bytes);
  const size_t m = (size_t)(2 * (E > 0 ? E : 1));
  w.key = c.take<long long>(m); w.skey = c.take<long long>(m);
  w.val = c.take<int>(m); w.sval = c.take<int>(m); w.cpos = c.take<int>(m); w.mark = c.take<int>(m); w.gstart = c.take<int>(m);
  w.seg_start = c.take<int>((size_t)C + 2);
  w.tmp_bytes = ap_tmp_bytes((long long)m);
  w.tmp = c.take<char>(w.tmp_bytes);
  w.bytes = c.off + 256;
  w.ok = c.ok();
}

}  // namespace
}  // namespace b3d

using namespace b3d;

extern "C" size_t b3d_average_precision_workspace_bytes(int64_t E, int32_t num_classes) {
  ApWs w;
  ap_carve(w, nullptr, 0, E, num_classes);
  return w.bytes;
}

extern "C" int b3d_average_precision(const float* scores, const void* y, int32_t y_is_int64, const float* edge_classes, int64_t E,
                                     int32_t num_classes, void* workspace, size_t workspace_bytes, double* ap, int32_t* count,
                                     b3d_stream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  B3D_REQUIRE(workspace && ap && count, "b3d_average_precision: null argument");
  B3D_REQUIRE(E == 0 || (scores && y), "b3d_average_precision: null scores / labels");
  B3D_REQUIRE(E >= 0 && E < (1ll << 30) && num_classes >= 0 && num_classes < 250, "b3d_average_precision: E %lld, classes %d",
              (long long)E, num_classes);
  const int C = num_classes;
  ApWs w;
  ap_carve(w, workspace, workspace_bytes, E, C);
  if (!w.ok) return fail(B3D_ERR_WORKSPACE, "b3d_average_precision: workspace %zu < %zu bytes", workspace_bytes, w.bytes);
  if (E > 0 && E <= kApDirectMaxE) {
    // scratch of the sort form, reused: meta <- val, positives' indices <- mark, their number <- seg_start, terms <- key / skey
    unsigned short* meta = (unsigned short*)w.val;
    int* pos_idx = w.mark;
    int* npos = w.seg_start;
    double* term_all = (double*)w.key;
    double* term_cls = (double*)w.skey;
    const unsigned eb = (unsigned)((E + 255) / 256);
    hipLaunchKernelGGL(ap_pack_kernel, dim3(eb), dim3(256), 0, stream, y, (int)y_is_int64, edge_classes, (long long)E, C, meta);
    hipLaunchKernelGGL(ap_compact_kernel, dim3(1), dim3(1024), 0, stream, (const unsigned short*)meta, (long long)E, pos_idx, npos);
    // persistent: one workgroup per 16 positives of a pass, at most one per CU-pair's worth (the count is on the device)
    long long groups_max = (E + kApWaves - 1) / kApWaves;
    const unsigned wg = (unsigned)(groups_max < 256 ? groups_max : 256);
    constexpr int ap_lds_bytes = kApTile * 6;
    B3D_TRY(set_lds_cached(reinterpret_cast<const void*>(ap_direct_kernel), ap_lds_bytes));
    hipLaunchKernelGGL(ap_direct_kernel, dim3(wg), dim3(kApWaves * 64), ap_lds_bytes, stream, scores, (const unsigned short*)meta,
                       (long long)E, (const int*)pos_idx, (const int*)npos, term_all, term_cls);
    hipLaunchKernelGGL(ap_direct_reduce_kernel, dim3(C + 1), dim3(256), 0, stream, (const unsigned short*)meta, (long long)E,
                       (const int*)pos_idx, (const int*)npos, (const double*)term_all, (const double*)term_cls, ap, count);
    B3D_HIP_CHECK(hipGetLastError());
    return B3D_OK;
  }
  hipLaunchKernelGGL(ap_init_kernel, dim3(1), dim3(256), 0, stream, w.seg_start, C + 2, -1);
  const long long M = edge_classes ? 2 * E : E;
  if (M > 0) {
    const unsigned eb = (unsigned)((E + 255) / 256), mb = (unsigned)((M + 255) / 256);
    hipLaunchKernelGGL(ap_keys_kernel, dim3(eb), dim3(256), 0, stream, scores, y, (int)y_is_int64, edge_classes, (long long)E, C, w.key, w.val);
    size_t tb = w.tmp_bytes;
    B3D_HIP_CHECK(rocprim::radix_sort_pairs(w.tmp, tb, w.key, w.skey, w.val, w.sval, (size_t)M, 0, 40, stream));
    hipLaunchKernelGGL(ap_marks_kernel, dim3(mb), dim3(256), 0, stream, w.skey, M, w.mark, w.seg_start);
    tb = w.tmp_bytes;
    B3D_HIP_CHECK(rocprim::inclusive_scan(w.tmp, tb, w.sval, w.cpos, (size_t)M, rocprim::plus<int>(), stream));
    tb = w.tmp_bytes;
    B3D_HIP_CHECK(rocprim::inclusive_scan(w.tmp, tb, w.mark, w.gstart, (size_t)M, rocprim::maximum<int>(), stream));
  }
  hipLaunchKernelGGL(ap_reduce_kernel, dim3(C + 1), dim3(256), 0, stream, w.skey, w.cpos, w.gstart, w.seg_start, M, C, ap, count);
  B3D_HIP_CHECK(hipGetLastError());
  return B3D_OK;
}
