// How fast does ONE wavefront per SIMD issue v_mfma_f32_16x16x32_bf16?  Variants: 1 / 2 / 4 accumulator chains, optional LDS
// fragment reads (3 x ds_read_b128 per 6 or 12 MFMAs).   hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));
template <int CHAINS, bool LDS, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 1) void k(const u4v* __restrict__ in, float* __restrict__ out, long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) u4v lds[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4096; i += WAVES * 64) lds[i] = in[i & 1023];
  __syncthreads();
  bf8 x[CHAINS][3];
  for (int c = 0; c < CHAINS; ++c) for (int p = 0; p < 3; ++p) x[c][p] = __builtin_bit_cast(bf8, in[(c * 3 + p) * 64 + lane]);
  bf8 w[3];
  for (int p = 0; p < 3; ++p) w[p] = __builtin_bit_cast(bf8, in[(16 + p) * 64 + lane]);
  v4f acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = v4f{0.f, 0.f, 0.f, 0.f};
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    bf8 nw[3];
    if constexpr (LDS) {
      const u4v* p = lds + ((it * 192 + lane) & 2047);
      nw[0] = __builtin_bit_cast(bf8, p[0]); nw[1] = __builtin_bit_cast(bf8, p[64]); nw[2] = __builtin_bit_cast(bf8, p[128]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
      acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], x[c][2], acc[c], 0, 0, 0);
      acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1], x[c][1], acc[c], 0, 0, 0);
      acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[2], x[c][0], acc[c], 0, 0, 0);
      acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], x[c][1], acc[c], 0, 0, 0);
      acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1], x[c][0], acc[c], 0, 0, 0);
      acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], x[c][0], acc[c], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (LDS) { w[0] = nw[0]; w[1] = nw[1]; w[2] = nw[2]; }
  }
  const long long t1 = clock64();
  v4f s = acc[0];
  for (int c = 1; c < CHAINS; ++c) s += acc[c];
  out[blockIdx.x * WAVES * 64 + threadIdx.x] = s.x + s.y + s.z + s.w;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
typedef float v16f __attribute__((ext_vector_type(16)));
template <int CHAINS, bool LDS, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 1) void k32(const u4v* __restrict__ in, float* __restrict__ out, long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) u4v lds[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4096; i += WAVES * 64) lds[i] = in[i & 1023];
  __syncthreads();
  bf8 x[CHAINS][3];
  for (int c = 0; c < CHAINS; ++c) for (int p = 0; p < 3; ++p) x[c][p] = __builtin_bit_cast(bf8, in[(c * 3 + p) * 64 + lane]);
  bf8 w[3];
  for (int p = 0; p < 3; ++p) w[p] = __builtin_bit_cast(bf8, in[(16 + p) * 64 + lane]);
  v16f acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    bf8 nw[3];
    if constexpr (LDS) {
      const u4v* p = lds + ((it * 192 + lane) & 2047);
      nw[0] = __builtin_bit_cast(bf8, p[0]); nw[1] = __builtin_bit_cast(bf8, p[64]); nw[2] = __builtin_bit_cast(bf8, p[128]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
      acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[c][2], acc[c], 0, 0, 0);
      acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], x[c][1], acc[c], 0, 0, 0);
      acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[2], x[c][0], acc[c], 0, 0, 0);
      acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[c][1], acc[c], 0, 0, 0);
      acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], x[c][0], acc[c], 0, 0, 0);
      acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[c][0], acc[c], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (LDS) { w[0] = nw[0]; w[1] = nw[1]; w[2] = nw[2]; }
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
  out[blockIdx.x * WAVES * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int CHAINS, bool LDS, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 1) void k32i(const u4v* __restrict__ in, float* __restrict__ out, long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) u4v lds[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4096; i += WAVES * 64) lds[i] = in[i & 1023];
  __syncthreads();
  bf8 x[CHAINS][3];
  for (int c = 0; c < CHAINS; ++c) for (int p = 0; p < 3; ++p) x[c][p] = __builtin_bit_cast(bf8, in[(c * 3 + p) * 64 + lane]);
  bf8 w[3];
  for (int p = 0; p < 3; ++p) w[p] = __builtin_bit_cast(bf8, in[(16 + p) * 64 + lane]);
  v16f acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    bf8 nw[3];
    if constexpr (LDS) {
      const u4v* p = lds + ((it * 192 + lane) & 2047);
      nw[0] = __builtin_bit_cast(bf8, p[0]); nw[1] = __builtin_bit_cast(bf8, p[64]); nw[2] = __builtin_bit_cast(bf8, p[128]);
    }
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
      acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[c][2], acc[c], 0, 0, 0);
      acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], x[c][1], acc[c], 0, 0, 0);
      acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[2], x[c][0], acc[c], 0, 0, 0);
      acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[c][1], acc[c], 0, 0, 0);
      acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], x[c][0], acc[c], 0, 0, 0);
      acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[c][0], acc[c], 0, 0, 0);
    }
    // one LDS read in each of the first three MFMA gaps (MI355X_MICROARCH.md: reads issued between MFMAs cost ~3 cycles per gap)
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 6 * CHAINS - 3, 0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (LDS) { w[0] = nw[0]; w[1] = nw[1]; w[2] = nw[2]; }
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
  out[blockIdx.x * WAVES * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int CHAINS, bool LDS, int WAVES>
void run32(const char* name, const u4v* in, float* out, long long* cyc) {
  const int iters = 2000, grid = 256;
  hipFuncSetAttribute((const void*)k32<CHAINS, LDS, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k32<CHAINS, LDS, WAVES>), dim3(grid), dim3(WAVES * 64), 65536, 0, in, out, cyc, iters);
  hipDeviceSynchronize();
  long long h[256];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double m = 0;
  for (int i = 0; i < grid; ++i) m += (double)h[i] / grid;
  printf("32x32x16: %-34s %6.2f cycles per MFMA (per wave; %d waves/SIMD)\n", name, m / iters / (6 * CHAINS), WAVES / 4);
}
template <int CHAINS, bool LDS, int WAVES>
void run(const char* name, const u4v* in, float* out, long long* cyc) {
  const int iters = 2000, grid = 256;
  hipFuncSetAttribute((const void*)k<CHAINS, LDS, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<CHAINS, LDS, WAVES>), dim3(grid), dim3(WAVES * 64), 65536, 0, in, out, cyc, iters);
  hipDeviceSynchronize();
  long long h[256];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double m = 0;
  for (int i = 0; i < grid; ++i) m += (double)h[i] / grid;
  printf("%-44s %6.2f cycles per MFMA (per wave; %d waves/SIMD)\n", name, m / iters / (6 * CHAINS), WAVES / 4);
}
int main() {
  u4v* in; float* out; long long* cyc;
  hipMalloc(&in, 65536 * 16); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
  unsigned* h = new unsigned[65536 * 4];
  for (int i = 0; i < 65536 * 4; ++i) h[i] = 0x3f803f80u + (i * 2654435761u >> 20);
  hipMemcpy(in, h, 65536 * 16, hipMemcpyHostToDevice);
  run<1, false, 4>("1 chain, registers only", in, out, cyc);
  run<2, false, 4>("2 chains, registers only", in, out, cyc);
  run<4, false, 4>("4 chains, registers only", in, out, cyc);
  run<1, true, 4>("1 chain + 3 ds_read_b128 per 6 MFMAs", in, out, cyc);
  run<2, true, 4>("2 chains + 3 ds_read_b128 per 12 MFMAs", in, out, cyc);
  run<4, true, 4>("4 chains + 3 ds_read_b128 per 24 MFMAs", in, out, cyc);
  run<1, true, 8>("1 chain + 3 ds_read_b128 per 6 MFMAs", in, out, cyc);
  run<2, true, 8>("2 chains + 3 ds_read_b128 per 12 MFMAs", in, out, cyc);
  run32<1, false, 4>("1 chain, registers only", in, out, cyc);
  run32<2, false, 4>("2 chains, registers only", in, out, cyc);
  run32<1, true, 4>("1 chain + 3 ds_read_b128 per 6", in, out, cyc);
  run32<2, true, 4>("2 chains + 3 ds_read_b128 per 12", in, out, cyc);
  run32<1, true, 8>("1 chain + 3 ds_read_b128 per 6", in, out, cyc);
  {
    const int iters = 2000, grid = 256;
    hipFuncSetAttribute((const void*)k32i<1, true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k32i<1, true, 4>), dim3(grid), dim3(256), 65536, 0, in, out, cyc, iters);
    hipDeviceSynchronize();
    long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < grid; ++i) m += (double)h[i] / grid;
    printf("32x32x16: 1 chain, reads interleaved in the MFMA gaps   %6.2f cycles per MFMA (1 wave/SIMD)\n", m / iters / 6);
  }
  return 0;
}
