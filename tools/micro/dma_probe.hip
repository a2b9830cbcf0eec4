// Probe: where does global_load_lds_dwordx4 with an instruction offset land in LDS?  (M0 base + inst_offset?)
//   hipcc --offload-arch=gfx950 -O3 dma_probe.hip -o dma_probe && ./dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(64) void probe(const unsigned* __restrict__ src, unsigned* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned lds[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 4096; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)lds;
  unsigned voff = lane * 16;
  unsigned keep;
  // piece 0: m0 = base, offset 0; piece 1: m0 = base (unchanged), inst offset 1024 -> does it land at base + 1024?
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2 offset:0\n\t"
      "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
      "s_add_u32 m0, m0, 0x2000\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
      "s_mov_b32 m0, %0\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&s"(keep)
      : "v"(voff), "s"(src), "s"(lds_base)
      : "memory");
  __syncthreads();
  for (int i = lane; i < 4096; i += 64) out[i] = lds[i];
}
int main() {
  std::vector<unsigned> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = i;
  unsigned *d, *o;
  hipMalloc(&d, 16384); hipMalloc(&o, 16384);
  hipMemcpy(d, h.data(), 16384, hipMemcpyHostToDevice);
  probe<<<1, 64, 16384>>>(d, o);
  std::vector<unsigned> r(4096);
  hipMemcpy(r.data(), o, 16384, hipMemcpyDeviceToHost);
  for (int i = 0; i < 4096; i += 256) printf("lds[%4d] = %08x  lds[%4d] = %08x\n", i, r[i], i + 255, r[i + 255]);
  return 0;
}
