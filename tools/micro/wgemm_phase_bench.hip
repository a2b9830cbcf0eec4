// Which phase of the cooperative weight-gradient step (csrc/b3d_wgemm.hpp) costs what, in isolation and together?
// One workgroup per CU (150 KB of LDS), 128 x 256 output tile, 32-row steps, operands as three bf16 piece images in LDS --
// the geometry of the shipped kernel's largest shape -- with the global loads replaced by register values (the launch
// measured the same with its loads removed, profiles/r04_c_wgemm_experiments.txt).
//
//   build:  hipcc -O3 --offload-arch=gfx950 -o wgemm_phase_bench wgemm_phase_bench.hip
//   run:    ./wgemm_phase_bench            (prints ns and cycles@2.0GHz per 32-row step for each combination)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef short s8 __attribute__((ext_vector_type(8)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) float lds_float;

constexpr int kRows = 32, kNG = 128, kKG = 256, kW = kNG + kKG;
constexpr int kPitch = ((kW / 2 - 8 + 63) / 64) * 64 + 8;        // dwords
constexpr int kPiece = kRows * kPitch, kBuf = 3 * kPiece;
constexpr int kLdsBytes = 2 * kBuf * 4 + 256;     // + stamps of the flag-synchronised form

enum { READS = 1, MFMA = 2, SPLIT = 4, WRITES = 8, BARRIER = 16, PRIO = 32, FUSED = 64, FLAGS = 128 };

__device__ __forceinline__ bf8 frag(const lds_float* img, int lane_off, int cd) {
  auto* p0 = (__attribute__((address_space(3))) s4*)(img + lane_off + cd);
  auto* p1 = (__attribute__((address_space(3))) s4*)(img + lane_off + 16 * kPitch + cd);
  const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p0);
  const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p1);
  const s8 v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  return __builtin_bit_cast(bf8, v);
}
struct Bf3 { bf8 p0, p1, p2; };

__device__ __forceinline__ void split4(const v4f x, unsigned (&p0)[2], unsigned (&p1)[2], unsigned (&p2)[2]) {
  const float f[4] = {x.x, x.y, x.z, x.w};
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    h[i] = __float_as_uint(f[i]);
    const float r1 = f[i] - __uint_as_float(h[i] & 0xffff0000u);
    m[i] = __float_as_uint(r1);
    l[i] = __float_as_uint(r1 - __uint_as_float(m[i] & 0xffff0000u));
  }
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    p0[d] = __builtin_amdgcn_perm(h[2 * d + 1], h[2 * d], 0x07060302u);
    p1[d] = __builtin_amdgcn_perm(m[2 * d + 1], m[2 * d], 0x07060302u);
    p2[d] = __builtin_amdgcn_perm(l[2 * d + 1], l[2 * d], 0x07060302u);
  }
}

// MULT wavefronts x (MBW x NBW) blocks cover 128 x 256: MULT = 4 -> 2 x 2 grid of 4 x 8 blocks; MULT = 8 -> 2 x 4 grid of 4 x 4.
template <int MODE, int MULT>
__global__ __launch_bounds__(512, 1) void phase_kernel(float* out, int steps, float seed) {
  extern __shared__ __attribute__((aligned(16))) float lds_raw[];
  lds_float* lds = (lds_float*)lds_raw;
  constexpr int WC = MULT == 4 ? 2 : 4, MBW = 4, NBW = MULT == 4 ? 8 : 4;
  constexpr bool fused = (MODE & FUSED) != 0;          // every wavefront stages AND multiplies (the shipped form)
  constexpr int LT = fused ? 512 : 256;                // staging threads
  constexpr int G4 = kNG / 4, A4 = kKG / 4, RG = LT / G4, RA = LT / A4;
  constexpr int GLs = kRows / RG, ALs = kRows / RA;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const bool mult = fused || wave < MULT;
  const bool loader = fused || wave >= 4;
  for (int i = tid; i < 2 * kBuf; i += 512) lds[i] = 0.f;
  __syncthreads();
  const int lt = fused ? tid : tid - 256;
  const int grow0 = lt / G4, gc4 = lt % G4, arow0 = lt / A4, ac4 = lt % A4;
  const int gdst0 = grow0 * kPitch + 2 * gc4, adst0 = arow0 * kPitch + kNG / 2 + 2 * ac4;
  v4f gx[GLs], ax[ALs];
#pragma unroll
  for (int i = 0; i < GLs; ++i) gx[i] = v4f{seed + i, seed * 3.f, seed + lane, 1.f / (1 + lane)};
#pragma unroll
  for (int i = 0; i < ALs; ++i) ax[i] = v4f{seed - i, seed * 5.f, seed - lane, 2.f / (1 + lane)};
  auto put = [&](lds_float* buf, int off, const v4f x) {
    unsigned p0[2], p1[2], p2[2];
    if constexpr (MODE & SPLIT) split4(x, p0, p1, p2);
    else {
      p0[0] = __float_as_uint(x.x); p0[1] = __float_as_uint(x.y); p1[0] = __float_as_uint(x.z); p1[1] = __float_as_uint(x.w);
      p2[0] = p0[0]; p2[1] = p1[1];
    }
    typedef __attribute__((address_space(3))) u2 lds_u2;
    lds_float* d = buf + off;
    if constexpr (MODE & WRITES) {
      *(lds_u2*)d = u2{p0[0], p0[1]};
      *(lds_u2*)(d + kPiece) = u2{p1[0], p1[1]};
      *(lds_u2*)(d + 2 * kPiece) = u2{p2[0], p2[1]};
    } else {
      asm volatile("" ::"v"(p0[0]), "v"(p0[1]), "v"(p1[0]), "v"(p1[1]), "v"(p2[0]), "v"(p2[1]));
    }
  };
  v4f bs = {0.f, 0.f, 0.f, 0.f};
  auto stage = [&](lds_float* buf, int t) {
#pragma unroll
    for (int i = 0; i < GLs; ++i) {
      v4f x = gx[i];
      x.x += (float)t;                                  // (a fresh value per step: nothing hoists out of the loop)
      put(buf, gdst0 + RG * i * kPitch, x);
      bs += x;
    }
#pragma unroll
    for (int i = 0; i < ALs; ++i) {
      v4f x = ax[i];
      x.y += (float)t;
      put(buf, adst0 + RA * i * kPitch, x);
    }
  };
  const int mw = fused ? wave : wave % MULT;
  const int wr = mw / WC, wc = mw % WC;
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const int lane_off = (4 * g + q) * kPitch + 2 * p;
  v4f acc[MBW][NBW];
#pragma unroll
  for (int a = 0; a < MBW; ++a)
#pragma unroll
    for (int b = 0; b < NBW; ++b) acc[a][b] = v4f{0.f, 0.f, 0.f, 0.f};
  bf8 dummy;
#pragma unroll
  for (int i = 0; i < 8; ++i) dummy[i] = (__bf16)(seed + i);
  auto rd3 = [&](const lds_float* cur, int cd) {
    Bf3 f;
    if constexpr (MODE & READS) {
      f.p0 = frag(cur, lane_off, cd);
      f.p1 = frag(cur + kPiece, lane_off, cd);
      f.p2 = frag(cur + 2 * kPiece, lane_off, cd);
    } else {
      f.p0 = dummy; f.p1 = dummy; f.p2 = dummy;
      asm volatile("" : "+v"(f.p0), "+v"(f.p1), "+v"(f.p2));
    }
    return f;
  };
  auto compute = [&](const lds_float* cur) {
    Bf3 af[MBW];
#pragma unroll
    for (int a = 0; a < MBW; ++a) af[a] = rd3(cur, 8 * (wr * MBW + a));
    Bf3 bc = rd3(cur, kNG / 2 + 8 * (wc * NBW));
#pragma unroll
    for (int b = 0; b < NBW; ++b) {
      Bf3 bn = bc;
      if (b + 1 < NBW) bn = rd3(cur, kNG / 2 + 8 * (wc * NBW + b + 1));
      if constexpr (MODE & MFMA) {
#pragma unroll
        for (int a = 0; a < MBW; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a].p0, bc.p2, acc[a][b], 0, 0, 0);
#pragma unroll
        for (int a = 0; a < MBW; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a].p1, bc.p1, acc[a][b], 0, 0, 0);
#pragma unroll
        for (int a = 0; a < MBW; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a].p2, bc.p0, acc[a][b], 0, 0, 0);
#pragma unroll
        for (int a = 0; a < MBW; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a].p0, bc.p1, acc[a][b], 0, 0, 0);
#pragma unroll
        for (int a = 0; a < MBW; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a].p1, bc.p0, acc[a][b], 0, 0, 0);
#pragma unroll
        for (int a = 0; a < MBW; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a].p0, bc.p0, acc[a][b], 0, 0, 0);
      } else {
#pragma unroll
        for (int a = 0; a < MBW; ++a) asm volatile("" ::"v"(af[a].p0), "v"(af[a].p1), "v"(af[a].p2));
        asm volatile("" ::"v"(bc.p0), "v"(bc.p1), "v"(bc.p2));
      }
      bc = bn;
    }
  };
  if constexpr (MODE & PRIO) { if (mult && !fused) __builtin_amdgcn_s_setprio(2); }
  if constexpr (fused) {
    for (int t = 0; t < steps; ++t) {
      stage(lds + (t & 1) * kBuf, t);
      if constexpr (MODE & BARRIER) __syncthreads();
      if (wave < MULT || MULT == 8) compute(lds + (t & 1) * kBuf);
    }
  } else if constexpr ((MODE & FLAGS) != 0) {
    // No barrier: stamps in LDS.  full[b][w] = last step loader wavefront w has completely staged into buffer b (+1);
    // done[b][m] = last step multiplier wavefront m has completely READ out of buffer b (+1).  A wavefront's LDS operations are
    // executed in order, so data written before a stamp is visible to whoever has seen the stamp.
    volatile int* full = reinterpret_cast<volatile int*>(lds_raw + 2 * kBuf);        // [2][4]
    volatile int* done = full + 8;                                                     // [2][4]
    if (tid < 16) const_cast<int*>(full)[tid] = 0;
    __syncthreads();
    auto wait4 = [&](volatile int* st, int want) {
      while (true) {
        const int a0 = st[0], a1 = st[1], a2 = st[2], a3 = st[3];
        if (min(min(a0, a1), min(a2, a3)) >= want) break;
        __builtin_amdgcn_s_sleep(1);
      }
    };
    if (loader && !mult) {
      const int lw = wave - 4;
      for (int t = 0; t < steps; ++t) {
        const int b = t & 1;
        if (t >= 2) wait4(done + 4 * b, t - 1);             // the multipliers are done reading step t - 2 (stamp t - 1)
        stage(lds + b * kBuf, t);
        __builtin_amdgcn_s_waitcnt(0xc07f);                 // lgkmcnt(0): this wavefront's tile words are in LDS
        if (lane == 0) full[4 * b + lw] = t + 1;
      }
    } else if (mult) {
      for (int t = 0; t < steps; ++t) {
        const int b = t & 1;
        wait4(full + 4 * b, t + 1);
        compute(lds + b * kBuf);
        if (lane == 0) done[4 * b + wave] = t + 1;          // (behind the step's last fragment read in program order)
      }
    }
  } else if (loader && !mult) {
    for (int t = 0; t < steps; ++t) {
      if constexpr (MODE & (SPLIT | WRITES)) stage(lds + ((t + 1) & 1) * kBuf, t);
      if constexpr (MODE & BARRIER) __syncthreads();
    }
  } else if (mult) {
    for (int t = 0; t < steps; ++t) {
      if constexpr (MODE & (READS | MFMA)) compute(lds + (t & 1) * kBuf);
      if constexpr (MODE & BARRIER) __syncthreads();
    }
  }
  float s = bs.x + bs.y + bs.z + bs.w;
#pragma unroll
  for (int a = 0; a < MBW; ++a)
#pragma unroll
    for (int b = 0; b < NBW; ++b) s += acc[a][b].x + acc[a][b].y + acc[a][b].z + acc[a][b].w;
  if (s == 12345.678f) out[blockIdx.x * 512 + tid] = s;
}

template <int MODE, int MULT>
static void run(const char* name, float* out, int steps) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(phase_kernel<MODE, MULT>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((phase_kernel<MODE, MULT>), dim3(256), dim3(512), kLdsBytes, 0, out, steps, 0.37f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  const hipError_t err = hipGetLastError();
  const double ns = best * 1e6 / steps;
  printf("%-86s %8.1f ns/step  %7.0f cyc@2.0GHz  (MFMA pipe floor 3072)%s\n", name, ns, ns * 2.0, err == hipSuccess ? "" : "  LAUNCH ERROR");
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 512 * 4);
  const int steps = 2000;
  printf("# 256 workgroups (one per CU), 512 threads, 128 x 256 tile, %d steps of 32 rows; LDS %d bytes\n", steps, kLdsBytes);
  run<MFMA, 4>("4 multiplier wavefronts: MFMAs only (operands in registers)", out, steps);
  run<READS, 4>("4 multiplier wavefronts: fragment reads only", out, steps);
  run<READS | MFMA, 4>("4 multiplier wavefronts: reads + MFMAs, no barrier", out, steps);
  run<READS | MFMA | BARRIER, 4>("4 multiplier wavefronts: reads + MFMAs + barrier (4 idle wavefronts join it)", out, steps);
  run<SPLIT, 4>("4 loader wavefronts: split only", out, steps);
  run<WRITES, 4>("4 loader wavefronts: LDS writes only", out, steps);
  run<SPLIT | WRITES, 4>("4 loader wavefronts: split + LDS writes", out, steps);
  run<SPLIT | WRITES | READS | MFMA, 4>("4 + 4: everything, no barrier", out, steps);
  run<SPLIT | WRITES | READS | MFMA | BARRIER, 4>("4 + 4: everything + barrier per step (v3's form)", out, steps);
  run<SPLIT | WRITES | READS | MFMA | BARRIER | PRIO, 4>("4 + 4: everything + barrier, multipliers at s_setprio 2", out, steps);
  run<SPLIT | WRITES | READS | MFMA | FLAGS, 4>("4 + 4: everything, LDS stamps instead of the barrier (loaders run ahead)", out, steps);
  run<SPLIT | MFMA, 4>("4 + 4: split beside MFMAs only (no LDS traffic)", out, steps);
  run<WRITES | MFMA, 4>("4 + 4: LDS writes beside MFMAs only", out, steps);
  run<WRITES | READS, 4>("4 + 4: LDS writes beside fragment reads (no VALU work, no MFMA)", out, steps);
  run<MFMA | FUSED, 8>("8 fused wavefronts: MFMAs only", out, steps);
  run<READS | MFMA | FUSED, 8>("8 fused wavefronts: reads + MFMAs", out, steps);
  run<SPLIT | WRITES | FUSED, 8>("8 fused wavefronts: split + writes", out, steps);
  run<SPLIT | WRITES | READS | MFMA | FUSED, 8>("8 fused wavefronts: everything, no barrier", out, steps);
  run<SPLIT | WRITES | READS | MFMA | BARRIER | FUSED, 8>("8 fused wavefronts: everything, one barrier per step (single buffer; shipped form has 1)", out, steps);
  hipFree(out);
  return 0;
}
