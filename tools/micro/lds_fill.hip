// Microbenchmark: how fast can ONE workgroup per CU fill LDS from an L2-resident buffer?
//   mode 0: LDS-DMA (global_load_lds, 16 B per lane), D instructions per lane in flight, then vmcnt(0) + barrier
//   mode 1: global_load_dwordx4 -> registers -> ds_write_b128, D loads per lane in flight
// hipcc --offload-arch=gfx950 -O3 lds_fill.hip -o lds_fill
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE, int NW, int D>
__global__ __launch_bounds__(NW * 64, 1) void fill(const float* __restrict__ src, int chunks, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int NT = NW * 64;
  float acc = 0.f;
  for (int c = 0; c < chunks; ++c) {
    const float* s = src + (size_t)c * D * NT * 4;
    if constexpr (MODE == 0) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const int base = d * NT + wave * 64;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + (size_t)(base + lane) * 4),
                                         (__attribute__((address_space(3))) void*)(lds + (size_t)base * 4), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      v4f t[D];
#pragma unroll
      for (int d = 0; d < D; ++d) t[d] = *reinterpret_cast<const v4f*>(s + (size_t)(d * NT + threadIdx.x) * 4);
#pragma unroll
      for (int d = 0; d < D; ++d) *reinterpret_cast<v4f*>(lds + (size_t)(d * NT + threadIdx.x) * 4) = t[d];
    }
    __syncthreads();
    acc += lds[(threadIdx.x * 5 + c) % (D * NT * 4)];
    __syncthreads();
  }
  out[blockIdx.x * NT + threadIdx.x] = acc;
}

template <int MODE, int NW, int D>
void run(const char* name, const float* src, float* out) {
  const int chunk_bytes = D * NW * 64 * 16;
  const int chunks = (4 << 20) / chunk_bytes;          // 4 MB per workgroup, all from the same L2-resident 4 MB
  hipFuncSetAttribute((const void*)fill<MODE, NW, D>, hipFuncAttributeMaxDynamicSharedMemorySize, chunk_bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((fill<MODE, NW, D>), dim3(256), dim3(NW * 64), chunk_bytes, 0, src, chunks, out);
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((fill<MODE, NW, D>), dim3(256), dim3(NW * 64), chunk_bytes, 0, src, chunks, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / 5, bytes = (double)chunks * chunk_bytes;
  printf("%-10s waves=%2d in-flight/lane=%2d chunk=%3d KB: %8.1f us  %6.1f B/clk/CU  (%.2f us per chunk)\n", name, NW, D, chunk_bytes >> 10, us,
         bytes / (us * 2400), us / chunks);
}

int main() {
  float* src; hipMalloc(&src, 8 << 20); hipMemset(src, 0, 8 << 20);
  float* out; hipMalloc(&out, 256 * 1024 * 4);
  run<0, 8, 2>("lds-dma", src, out);
  run<0, 8, 6>("lds-dma", src, out);
  run<0, 8, 12>("lds-dma", src, out);
  run<0, 16, 6>("lds-dma", src, out);
  run<0, 4, 12>("lds-dma", src, out);
  run<1, 8, 2>("via-vgpr", src, out);
  run<1, 8, 6>("via-vgpr", src, out);
  run<1, 8, 12>("via-vgpr", src, out);
  run<1, 16, 6>("via-vgpr", src, out);
  run<1, 4, 12>("via-vgpr", src, out);
  return 0;
}
