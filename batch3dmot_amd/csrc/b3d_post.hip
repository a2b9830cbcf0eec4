// Post-processing of per-window edge scores (reference predict.py:199-233 and :92-124): mean score per
// global edge over the overlapping windows, per-class threshold, best predecessor / successor per node.
// The reference walks str(meta)-keyed Python dictionaries edge by edge; here the grouping is a stable
// device radix sort (rocPRIM) and everything else a handful of small kernels.  Deterministic and
// order-exact: the mean is the float64 sum of an edge's scores in order of appearance divided by their
// count (np.mean of the reference's list), kept edges come out in first-appearance order (dict insertion
// order), and ties of max() go to the entry inserted first.
#include <string.h>
#include "b3d_common.hpp"
#include <rocprim/rocprim.hpp>

namespace b3d {
namespace {

// Order-preserving map of a double onto an unsigned integer (any sign; 0 is below every value and means "none").
__device__ __forceinline__ unsigned long long ordered_bits(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
// A node id outside [0, N) or a class id outside [0, C) is counted in *invalid (the caller raises on it) and
// clamped, so that nothing below indexes out of bounds.
__global__ void post_keys_kernel(const long long* __restrict__ pairs, long long M, long long N, long long* key, int* pos, int* invalid) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  long long s = pairs[2 * i], d = pairs[2 * i + 1];
  if (s < 0 || s >= N || d < 0 || d >= N) {
    atomicAdd(invalid, 1);
    s = s < 0 ? 0 : (s >= N ? N - 1 : s);
    d = d < 0 ? 0 : (d >= N ? N - 1 : d);
  }
  key[i] = s * N + d;
  pos[i] = (int)i;
}
__global__ void post_check_classes_kernel(const long long* __restrict__ node_class, long long N, int C, int* invalid) {
  const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n < N && (node_class[n] < 0 || node_class[n] >= C)) atomicAdd(invalid, 1);
}
__global__ void post_heads_kernel(const long long* __restrict__ skey, long long M, int* head) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  head[i] = (i == 0 || skey[i] != skey[i - 1]) ? 1 : 0;
}
// one thread per sorted position that starts a run: mean over the run in order of appearance
__global__ void post_means_kernel(const long long* __restrict__ skey, const int* __restrict__ spos, const int* __restrict__ head,
                                  const int* __restrict__ seg, const float* __restrict__ scores, long long M,
                                  int* first_pos, double* mean, long long* ukey, int* uid) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M || !head[i]) return;
  const int u = seg[i];                                  // exclusive scan of head = index of this unique edge
  double s = 0.0;
  long long j = i;
  int cnt = 0;
  for (; j < M && skey[j] == skey[i]; ++j) { s += (double)scores[spos[j]]; ++cnt; }   // stable sort: ascending position
  first_pos[u] = spos[i];
  mean[u] = s / (double)cnt;
  ukey[u] = skey[i];
  uid[u] = u;
}
__global__ void post_init_kernel(long long N, unsigned long long* best_in, unsigned long long* best_out, int* first_in, int* first_out) {
  const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  best_in[n] = 0ull; best_out[n] = 0ull;
  first_in[n] = 0x7fffffff; first_out[n] = 0x7fffffff;
}
// entries past the number of distinct edges sort behind every real first position
__global__ void post_pad_kernel(const int* __restrict__ U, long long M, int* first_pos, int* uid) {
  const long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= M || u < *U) return;
  first_pos[u] = 0x7fffffff;
  uid[u] = (int)u;
}
// uniques in first-appearance order -> keep flag by the class threshold of the SOURCE node
__global__ void post_keep_kernel(const int* __restrict__ order, const long long* __restrict__ ukey, const double* __restrict__ mean,
                                 const long long* __restrict__ node_class, const double* __restrict__ thr, int C, long long N,
                                 const int* __restrict__ U, long long M, int* keep) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r > M) return;
  if (r >= *U) { keep[r] = 0; return; }
  const int u = order[r];
  const long long src = ukey[u] / N;
  long long c = node_class[src];
  c = c < 0 ? 0 : (c >= C ? C - 1 : c);                  // counted by post_check_classes_kernel
  keep[r] = mean[u] > thr[c] ? 1 : 0;
}
__global__ void post_emit_kernel(const int* __restrict__ order, const long long* __restrict__ ukey, const double* __restrict__ mean,
                                 const int* __restrict__ keep, const int* __restrict__ slot, long long N,
                                 const int* __restrict__ U, long long* kept_pairs, double* kept_scores,
                                 unsigned long long* best_in, unsigned long long* best_out) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= *U || !keep[r]) return;
  const int u = order[r], k = slot[r];
  const long long src = ukey[u] / N, dst = ukey[u] - src * N;
  kept_pairs[2 * k] = src; kept_pairs[2 * k + 1] = dst;
  kept_scores[k] = mean[u];
  const unsigned long long bits = ordered_bits(mean[u]);
  atomicMax(best_in + dst, bits);
  atomicMax(best_out + src, bits);
}
// among the kept edges that reach a node's best score, the one inserted first (smallest k)
__global__ void post_first_kernel(const long long* __restrict__ kept_pairs, const double* __restrict__ kept_scores,
                                  const int* __restrict__ Kp,
                                  const unsigned long long* __restrict__ best_in, const unsigned long long* __restrict__ best_out,
                                  int* first_in, int* first_out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= *Kp) return;
  const long long src = kept_pairs[2 * k], dst = kept_pairs[2 * k + 1];
  const unsigned long long bits = ordered_bits(kept_scores[k]);
  if (bits == best_in[dst]) atomicMin(first_in + dst, k);
  if (bits == best_out[src]) atomicMin(first_out + src, k);
}
__global__ void post_flux_kernel(const long long* __restrict__ kept_pairs, const int* __restrict__ first_in,
                                 const int* __restrict__ first_out, long long N, long long* pred, long long* succ) {
  const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const int ki = first_in[n], ko = first_out[n];
  pred[n] = ki == 0x7fffffff ? -1 : kept_pairs[2 * ki];
  succ[n] = ko == 0x7fffffff ? -1 : kept_pairs[2 * ko + 1];
}

struct PostWs {
  long long *key, *skey, *ukey;
  int *pos, *spos, *head, *seg, *first_pos, *uid, *sfirst, *order, *keep, *slot, *first_in, *first_out, *count;
  double* mean;
  unsigned long long *best_in, *best_out;
  void* tmp;
  size_t tmp_bytes, bytes;
  bool ok;
};

size_t post_tmp_bytes(long long M) {
  size_t a = 0, b = 0, c = 0;
  (void)rocprim::radix_sort_pairs(nullptr, a, (long long*)nullptr, (long long*)nullptr, (int*)nullptr, (int*)nullptr, (size_t)M, 0, 64, nullptr);
  (void)rocprim::radix_sort_pairs(nullptr, b, (int*)nullptr, (int*)nullptr, (int*)nullptr, (int*)nullptr, (size_t)M, 0, 32, nullptr);
  (void)rocprim::exclusive_scan(nullptr, c, (int*)nullptr, (int*)nullptr, 0, (size_t)M + 1, rocprim::plus<int>(), nullptr);
  size_t m = a > b ? a : b;
  return (m > c ? m : c) + 256;
}

void post_carve(PostWs& w, void* p, size_t bytes, long long M, long long N) {
  Carver c(p, bytes);
  const size_t m = (size_t)(M > 0 ? M : 1), n = (size_t)(N > 0 ? N : 1);
  w.key = c.take<long long>(m); w.skey = c.take<long long>(m); w.ukey = c.take<long long>(m);
  w.pos = c.take<int>(m); w.spos = c.take<int>(m); w.head = c.take<int>(m + 1); w.seg = c.take<int>(m + 1);
  w.first_pos = c.take<int>(m); w.uid = c.take<int>(m); w.sfirst = c.take<int>(m); w.order = c.take<int>(m);
  w.keep = c.take<int>(m + 1); w.slot = c.take<int>(m + 1);
  w.first_in = c.take<int>(n); w.first_out = c.take<int>(n); w.count = c.take<int>(4);
  w.mean = c.take<double>(m);
  w.best_in = c.take<unsigned long long>(n); w.best_out = c.take<unsigned long long>(n);
  w.tmp_bytes = post_tmp_bytes(M);
  w.tmp = c.take<char>(w.tmp_bytes);
  w.bytes = c.off + 256;
  w.ok = c.ok();
}

}  // namespace
}  // namespace b3d

using namespace b3d;

extern "C" size_t b3d_post_workspace_bytes(int64_t M, int64_t N) {
  PostWs w;
  post_carve(w, nullptr, 0, M, N);
  return w.bytes;
}

// The number of kept edges is data dependent: counts[0] = distinct edges, counts[1] = kept edges, counts[2] = entries
// of `pairs` / `node_class` outside their range (device int32[3]); kept_pairs [M,2] / kept_scores [M] are filled for
// the first counts[1] rows.  The caller reads the counts when it needs the size (one 12-byte copy), exactly as
// torch's boolean indexing would, and treats counts[2] != 0 as the index error the reference's dictionaries raise.
extern "C" int b3d_post_greedy(const int64_t* pairs, const float* scores, int64_t M, const int64_t* node_class, int64_t N,
                               const double* class_threshold, int32_t num_classes, void* workspace, size_t workspace_bytes, int64_t* kept_pairs,
                               double* kept_scores, int64_t* pred, int64_t* succ, int32_t* counts, b3d_stream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  B3D_REQUIRE(node_class && class_threshold && workspace && kept_pairs && kept_scores && pred && succ && counts,
              "b3d_post_greedy: null argument");
  B3D_REQUIRE(M == 0 || (pairs && scores), "b3d_post_greedy: null edge list");
  B3D_REQUIRE(M >= 0 && N > 0 && M < (1ll << 31) && N < (1ll << 31), "b3d_post_greedy: M %lld, N %lld", (long long)M, (long long)N);
  B3D_REQUIRE(num_classes > 0, "b3d_post_greedy: num_classes %d", (int)num_classes);
  PostWs w;
  post_carve(w, workspace, workspace_bytes, M, N);
  if (!w.ok) return fail(B3D_ERR_WORKSPACE, "b3d_post_greedy: workspace %zu < %zu bytes", workspace_bytes, w.bytes);
  const unsigned nb = (unsigned)((N + 255) / 256);
  hipLaunchKernelGGL(post_init_kernel, dim3(nb), dim3(256), 0, stream, (long long)N, w.best_in, w.best_out, w.first_in, w.first_out);
  B3D_HIP_CHECK(hipMemsetAsync(counts, 0, 3 * sizeof(int32_t), stream));
  hipLaunchKernelGGL(post_check_classes_kernel, dim3(nb), dim3(256), 0, stream, (const long long*)node_class, (long long)N, (int)num_classes, counts + 2);
  if (M > 0) {
    const unsigned mb = (unsigned)((M + 256) / 256);       // covers M + 1 items where a scan total is written
    hipLaunchKernelGGL(post_keys_kernel, dim3(mb), dim3(256), 0, stream, (const long long*)pairs, (long long)M, (long long)N, w.key, w.pos, counts + 2);
    size_t tb = w.tmp_bytes;
    B3D_HIP_CHECK(rocprim::radix_sort_pairs(w.tmp, tb, w.key, w.skey, w.pos, w.spos, (size_t)M, 0, 64, stream));
    hipLaunchKernelGGL(post_heads_kernel, dim3(mb), dim3(256), 0, stream, w.skey, (long long)M, w.head);
    B3D_HIP_CHECK(hipMemsetAsync(w.head + M, 0, 4, stream));
    tb = w.tmp_bytes;
    B3D_HIP_CHECK(rocprim::exclusive_scan(w.tmp, tb, w.head, w.seg, 0, (size_t)M + 1, rocprim::plus<int>(), stream));
    const int* U = w.seg + M;                               // number of distinct edges, on the device
    // The launches below are sized for the worst case (U <= M) and masked by the device-side count: no value
    // has to visit the host in the middle of the call.
    hipLaunchKernelGGL(post_means_kernel, dim3(mb), dim3(256), 0, stream, w.skey, w.spos, w.head, w.seg, scores, (long long)M,
                       w.first_pos, w.mean, w.ukey, w.uid);
    hipLaunchKernelGGL(post_pad_kernel, dim3(mb), dim3(256), 0, stream, U, (long long)M, w.first_pos, w.uid);
    tb = w.tmp_bytes;
    B3D_HIP_CHECK(rocprim::radix_sort_pairs(w.tmp, tb, w.first_pos, w.sfirst, w.uid, w.order, (size_t)M, 0, 32, stream));
    hipLaunchKernelGGL(post_keep_kernel, dim3(mb), dim3(256), 0, stream, w.order, w.ukey, w.mean, (const long long*)node_class,
                       class_threshold, (int)num_classes, (long long)N, U, (long long)M, w.keep);
    tb = w.tmp_bytes;
    B3D_HIP_CHECK(rocprim::exclusive_scan(w.tmp, tb, w.keep, w.slot, 0, (size_t)M + 1, rocprim::plus<int>(), stream));
    const int* K = w.slot + M;                              // number of kept edges
    hipLaunchKernelGGL(post_emit_kernel, dim3(mb), dim3(256), 0, stream, w.order, w.ukey, w.mean, w.keep, w.slot, (long long)N, U,
                       (long long*)kept_pairs, kept_scores, w.best_in, w.best_out);
    hipLaunchKernelGGL(post_first_kernel, dim3(mb), dim3(256), 0, stream, (const long long*)kept_pairs, kept_scores, K, w.best_in,
                       w.best_out, w.first_in, w.first_out);
    B3D_HIP_CHECK(hipMemcpyAsync(counts, U, 4, hipMemcpyDeviceToDevice, stream));
    B3D_HIP_CHECK(hipMemcpyAsync(counts + 1, K, 4, hipMemcpyDeviceToDevice, stream));
  }
  hipLaunchKernelGGL(post_flux_kernel, dim3(nb), dim3(256), 0, stream, (const long long*)kept_pairs, w.first_in, w.first_out,
                     (long long)N, (long long*)pred, (long long*)succ);
  B3D_HIP_CHECK(hipGetLastError());
  return B3D_OK;
}
