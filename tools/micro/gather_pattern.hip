// Microbenchmark: TA/L1 rate of 16-byte-per-lane row gathers for two lane <-> (row, piece) mappings.
//   A ("layout L"): lane = m + 16 q  -> row m, 16-byte piece q of a 64-byte block   (adjacent lanes: different rows)
//   B (quad-contiguous): lane = 4 m + q                                                (adjacent lanes: one 64-byte block)
// hipcc --offload-arch=gfx950 -O3 gather_pattern.hip -o gather_pattern && ./gather_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE, int NB, int U>
__global__ __launch_bounds__(512, 1) void gather(const float* __restrict__ tab, int stride, const int* __restrict__ idx, int iters,
                                                 float* __restrict__ out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int m = MODE == 0 ? (lane & 15) : (lane >> 2), q = MODE == 0 ? (lane >> 4) : (lane & 3);
  const int* ix = idx + ((size_t)(blockIdx.x * 8 + wave) * iters) * 16 * U;
  v4f acc[NB];
  for (int b = 0; b < NB; ++b) acc[b] = v4f{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    v4f t[U][NB];
    if constexpr (MODE == 2) {                       // 8 lanes per row: 128 contiguous bytes, 8 rows per instruction
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int r = ix[(it * U + u) * 16 + 8 * h + (lane >> 3)];
          const float* p = tab + (size_t)r * stride + 4 * (lane & 7);
#pragma unroll
          for (int c = 0; c < NB / 2; ++c) t[u][h * (NB / 2) + c] = *reinterpret_cast<const v4f*>(p + 32 * c);
        }
      }
    } else {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = ix[(it * U + u) * 16 + m];
      const float* p = tab + (size_t)r * stride + 4 * q;
#pragma unroll
      for (int b = 0; b < NB; ++b) t[u][b] = *reinterpret_cast<const v4f*>(p + 16 * b);
    }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[b] += t[u][b];
  }
  v4f s = acc[0];
  for (int b = 1; b < NB; ++b) s += acc[b];
  out[(size_t)(blockIdx.x * 512 + threadIdx.x)] = s.x + s.y + s.z + s.w;
}

template <int MODE, int NB, int U>
void run(const char* name, const float* tab, int stride, const int* idx, int iters, float* out, bool seq) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gather<MODE, NB, U>), dim3(256), dim3(512), 0, 0, tab, stride, idx, iters, out);
  hipEventRecord(e0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((gather<MODE, NB, U>), dim3(256), dim3(512), 0, 0, tab, stride, idx, iters, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps;
  const double bytes = 256.0 * 8 * iters * U * 16 * NB * 64;
  printf("%-28s %s  NB=%d U=%d: %8.2f us  %7.1f GB/s  %6.1f B/clk/CU (2.4 GHz)\n", name, seq ? "seq " : "rand", NB, U, us, bytes / us * 1e-3,
         bytes / 256 / (us * 2400));
}

int main() {
  const int N = 3000, stride = 432, iters = 32, UMAX = 8;
  std::vector<float> h((size_t)N * stride, 1.f);
  float* tab; hipMalloc(&tab, h.size() * 4); hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const size_t nidx = (size_t)256 * 8 * iters * 16 * UMAX;
  std::vector<int> hi(nidx), hs(nidx);
  srand(1);
  for (size_t i = 0; i < nidx; ++i) { hi[i] = rand() % N; hs[i] = (int)(i % N); }
  int *idx, *ids; hipMalloc(&idx, nidx * 4); hipMalloc(&ids, nidx * 4);
  hipMemcpy(idx, hi.data(), nidx * 4, hipMemcpyHostToDevice);
  hipMemcpy(ids, hs.data(), nidx * 4, hipMemcpyHostToDevice);
  float* out; hipMalloc(&out, 256 * 512 * 4);
  run<0, 3, 8>("A lane=m+16q", tab, stride, idx, iters, out, false);
  run<1, 3, 8>("B lane=4m+q", tab, stride, idx, iters, out, false);
  run<0, 6, 4>("A lane=m+16q", tab, stride, idx, iters, out, false);
  run<1, 6, 4>("B lane=4m+q", tab, stride, idx, iters, out, false);
  run<0, 3, 8>("A lane=m+16q", tab, stride, ids, iters, out, true);
  run<1, 3, 8>("B lane=4m+q", tab, stride, ids, iters, out, true);
  run<2, 6, 4>("C 8 lanes/row", tab, stride, idx, iters, out, false);
  run<2, 2, 8>("C 8 lanes/row", tab, stride, idx, iters, out, false);
  run<1, 2, 8>("B lane=4m+q", tab, stride, idx, iters, out, false);
  run<0, 2, 8>("A lane=m+16q", tab, stride, idx, iters, out, false);
  run<0, 1, 8>("A lane=m+16q", tab, stride, idx, iters, out, false);
  run<1, 1, 8>("B lane=4m+q", tab, stride, idx, iters, out, false);
  return 0;
}
