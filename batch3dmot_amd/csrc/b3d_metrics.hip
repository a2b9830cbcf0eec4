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
