// How fast does ONE workgroup per CU stream an L2-resident weight image (1 MB, the same for every workgroup) through an LDS ring
// by LDS-DMA, as a function of slot size and of the number of chunks in flight?  (The node kernels' ring: 2 x 52 KB slots, one chunk
// in flight, s_waitcnt vmcnt(0) + barrier per chunk.)
//   build: hipcc -O3 --offload-arch=gfx950 -o ring_rate ring_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int NT, int SLOT_BYTES, int NSLOT, int CONSUME>
__global__ __launch_bounds__(NT, 1) void ring_kernel(const float* __restrict__ w, int total_bytes, float* out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int DEPTH = NSLOT - 1;                       // chunks in flight while one is consumed
  constexpr int PER = SLOT_BYTES / 16 / NT;              // global_load_lds per thread and chunk
  static_assert(SLOT_BYTES % (16 * NT) == 0, "slot = whole passes of the workgroup");
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int nchunks = total_bytes / SLOT_BYTES;
  auto issue = [&](int c) {
    const float* src = w + (size_t)c * (SLOT_BYTES / 4);
    float* dst = smem + (size_t)(c % NSLOT) * (SLOT_BYTES / 4);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int base = (i * NT / 64 + wave) * 64;        // float4 index, wave-uniform
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)(base + lane) * 4),
                                       (__attribute__((address_space(3))) void*)(dst + (size_t)base * 4), 16, 0, 0);
    }
  };
  for (int c = 0; c < DEPTH && c < nchunks; ++c) issue(c);
  float s = 0.f;
  for (int c = 0; c < nchunks; ++c) {
    // chunk c has landed once at most the loads of the (DEPTH - 1) younger chunks are outstanding
    const int younger = (nchunks - 1 - c) < (DEPTH - 1) ? (nchunks - 1 - c) : (DEPTH - 1);
    if (younger * PER >= 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else {
      switch (younger * PER) {
#define W(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
        W(0) W(1) W(2) W(3) W(4) W(5) W(6) W(7) W(8) W(9) W(10) W(11) W(12) W(13) W(14) W(15) W(16) W(17) W(18) W(19) W(20) W(21) W(22) W(23)
#undef W
      }
    }
    __syncthreads();
    if (c + DEPTH < nchunks) issue(c + DEPTH);           // into the slot of chunk c - 1: everybody is past it
    const float* cur = smem + (size_t)(c % NSLOT) * (SLOT_BYTES / 4);
    if (CONSUME) {
      // read the chunk once (b128 per thread per pass) and spend CONSUME x 4 cycles of VALU per float4
#pragma unroll 4
      for (int i = threadIdx.x; i < SLOT_BYTES / 16; i += NT) {
        const float4 v = reinterpret_cast<const float4*>(cur)[i];
        float t = v.x + v.y + v.z + v.w;
#pragma unroll
        for (int k = 0; k < CONSUME; ++k) t = t * 1.0001f + 0.5f;
        s += t;
      }
    }
  }
  if (s == 1234.5f) out[blockIdx.x] = s;
}

template <int NT, int SLOT_BYTES, int NSLOT, int CONSUME>
static void run(const float* w, int total, float* out, int grid) {
  const int lds = SLOT_BYTES * NSLOT;
  hipFuncSetAttribute(reinterpret_cast<const void*>(ring_kernel<NT, SLOT_BYTES, NSLOT, CONSUME>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 6; ++rep) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((ring_kernel<NT, SLOT_BYTES, NSLOT, CONSUME>), dim3(grid), dim3(NT), lds, 0, w, total, out);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  const hipError_t err = hipGetLastError();
  printf("threads %4d  slot %6d B x %d slots (%d in flight)  consume %2d  grid %3d : %7.1f us  -> %6.1f GB/s per CU%s\n", NT, SLOT_BYTES, NSLOT,
         NSLOT - 1, CONSUME, grid, best * 1e3, total / (best * 1e-3) * 1e-9, err == hipSuccess ? "" : "  LAUNCH ERROR");
}

int main() {
  const int total = 1 << 20;
  float *w, *out;
  (void)hipMalloc(&w, total);
  (void)hipMemset(w, 0, total);
  (void)hipMalloc(&out, 4096);
  printf("# one workgroup per CU streams the same %d bytes (L2-resident) through an LDS ring; barrier per chunk\n", total);
  for (int grid : {188, 256}) {
    run<256, 53248 / 13 * 16, 2, 0>(w, total, out, grid);       // 64 KB slots, 1 in flight
    run<256, 32768, 2, 0>(w, total, out, grid);
    run<256, 32768, 3, 0>(w, total, out, grid);
    run<256, 32768, 4, 0>(w, total, out, grid);
    run<256, 16384, 2, 0>(w, total, out, grid);
    run<256, 16384, 4, 0>(w, total, out, grid);
    run<256, 16384, 8, 0>(w, total, out, grid);
    run<256, 8192, 8, 0>(w, total, out, grid);
    run<256, 8192, 16, 0>(w, total, out, grid);
    run<512, 32768, 4, 0>(w, total, out, grid);
    run<512, 16384, 8, 0>(w, total, out, grid);
    run<256, 32768, 2, 8>(w, total, out, grid);
    run<256, 32768, 4, 8>(w, total, out, grid);
    run<256, 16384, 8, 8>(w, total, out, grid);
    run<256, 32768, 2, 32>(w, total, out, grid);
    run<256, 32768, 4, 32>(w, total, out, grid);
    run<256, 16384, 8, 32>(w, total, out, grid);
  }
  return 0;
}
