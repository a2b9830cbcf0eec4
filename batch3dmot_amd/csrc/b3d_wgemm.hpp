// Cooperative weight gradient (same jobs, tasks and slabs as b3d_wstream.hpp; the hoisted plan's launch).
//
//   dW[n][k] = sum over layer variants v, rows r of  G_v[r][n] * Act_v[r][k]        db[n] = sum G_v[r][n]
//
// wstream gives every WAVEFRONT a 64 x 96 block of dW and lets it stream its own rows: a 384 x 512 matrix becomes 36
// jobs that read G six times and Act six times (measured: 4.7 GB of HBM traffic per backward pass for 1.7 GB of
// operands, 5.3 TB/s -- the launch sits on the HBM roofline of its OWN re-reads), on exact-fp32 MFMA.  Here a WORKGROUP
// of 8 wavefronts owns up to 128 x 256 (256 x 96, 192 x 128 ...) of dW: per step it stages 32 rows of G and Act ONCE
// -- split on the way into three bf16 pieces each (x = x0 + x1 + x2 exactly, b3d_dev.hpp) -- as row-major [32][W] bf16
// images in LDS, and every wavefront reads the operands of its 2-24 output blocks from there.  The contraction index
// of v_mfma_f32_16x16x32_bf16 is the ROW of those images, i.e. the operands are needed column-major:
// ds_read_b64_tr_b16 delivers exactly that (per 16 lanes a 4 rows x 16 columns block, transposed), so nothing is
// transposed in registers or at staging time.  Lane group g takes rows {4g..4g+3} and {16+4g..16+4g+3} as its eight
// k-slots -- any assignment works as long as both operands use the same one -- which with a row pitch = 8 (mod 64)
// dwords makes both transposed reads of a 32-lane half conflict-free.  Six piece products per block (bf16x6) in fp32
// accumulators; two LDS buffers, one barrier per 32-row step; the global loads run one or two steps ahead of the MFMAs.
// A job may take its activation columns from two sources (edge_update.0: e | att), so that G is read once for both.
#pragma once
#include "b3d_wstream.hpp"

namespace b3d {

constexpr int kWgmThreads = 512, kWgmRows = 32, kWgmMaxW = 384;
__host__ __device__ constexpr int wgm_pitch(int W) { return ((W / 2 - 8 + 63) / 64) * 64 + 8; }     // dwords, >= W / 2, = 8 mod 64
// 32 x 32 x 16 tiles (B32): the two 16-lane groups of a 32-lane half read the SAME four rows at column blocks 16 apart (8 dwords), so
// the rows must land 16 banks apart: pitch = 16 (mod 64) dwords -> rows q = 0..3 at banks 16 q, the second column block at + 8
__host__ __device__ constexpr int wgm_pitch32(int W) { return ((W / 2 - 16 + 63) / 64) * 64 + 16; }
#ifndef B3D_WGM_B32
#define B3D_WGM_B32 1          // 1: v_mfma_f32_32x32x16_bf16 for the shapes whose wave grid divides into 32 x 32 blocks
#endif
constexpr int kWgmLdsBytes = 2 * 3 * kWgmRows * (B3D_WGM_B32 ? wgm_pitch32(kWgmMaxW) : wgm_pitch(kWgmMaxW)) * 4;
typedef float v16f __attribute__((ext_vector_type(16)));

// LDS pointers stay in the LDS address space end to end: through a generic float* (a noinline function's argument) every one
// of the 48 transposed reads of a step paid an address-space cast with its null check (cmp + cndmask + 64-bit add).
typedef __attribute__((address_space(3))) float lds_float;
// ... and the operand rows are read as GLOBAL loads: the job's pointers come out of a struct, i.e. generic, and generic
// loads are flat_load (they also tick the LDS counter and check the LDS aperture)
typedef const __attribute__((address_space(1))) v4f* gbl_v4f_p;
typedef short wgm_s4 __attribute__((ext_vector_type(4)));
typedef short wgm_s8 __attribute__((ext_vector_type(8)));

// shapes: (wave grid WR x WC) x (blocks per wavefront MBW x NBW): G width NG = 16 WR MBW, activation width KG = 16 WC NBW
enum {
  WGM_128_256 = 0, WGM_128_192, WGM_128_128, WGM_64_128, WGM_256_64, WGM_256_96, WGM_192_64, WGM_192_96, WGM_96_128,
  WGM_192_128, WGM_128_96, WGM_256_128, WGM_SHAPES
};
inline int wgm_shape(int ng, int kg) {
  static const int t[WGM_SHAPES][2] = {{128, 256}, {128, 192}, {128, 128}, {64, 128}, {256, 64}, {256, 96}, {192, 64}, {192, 96},
                                       {96, 128}, {192, 128}, {128, 96}, {256, 128}};
  for (int i = 0; i < WGM_SHAPES; ++i)
    if (t[i][0] == ng && t[i][1] == kg) return i;
  return -1;
}

__device__ __forceinline__ void wgm_split4(const v4f x, unsigned (&p0)[2], unsigned (&p1)[2], unsigned (&p2)[2]) {
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

// operand fragment: rows {4g+q} and {16+4g+q} of 16 columns starting at dword column `cd` of one piece image
__device__ __forceinline__ bf8 wgm_frag(const lds_float* img, int lane_off /* (4g+q) * pitch + 2p, dwords */, int pitch, int cd) {
  auto* p0 = (__attribute__((address_space(3))) wgm_s4*)(img + lane_off + cd);
  auto* p1 = (__attribute__((address_space(3))) wgm_s4*)(img + lane_off + 16 * pitch + cd);
  const wgm_s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p0);
  const wgm_s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p1);
  const wgm_s8 v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  return __builtin_bit_cast(bf8, v);
}

// not inlined: one register allocation per shape (inlined into the dispatch switch the kernel spilled 164 VGPRs)
// B32: blocks are 32 x 32 (v_mfma_f32_32x32x16_bf16, two per 32-row step and block: half the matrix instructions -- and half of
// their issue slots, which this kernel is short of: per step a wavefront issues ~170 vector instructions of operand split, ~57 LDS
// instructions and 96 matrix instructions of the 16 x 16 x 32 form, each of which holds the SIMD's issue port for 8 of its 16 cycles)
template <int WR, int WC, int MBW, int NBW, bool GATHER, bool B32 = false>
__device__ __attribute__((noinline)) void wgm_task(const WsJob& job, int chunk, lds_float* lds) {
  constexpr int BS = B32 ? 32 : 16;
  constexpr int NG = BS * WR * MBW, KG = BS * WC * NBW, W = NG + KG, PITCH = B32 ? wgm_pitch32(W) : wgm_pitch(W);
  constexpr int PIECE = kWgmRows * PITCH, BUF = 3 * PIECE;
  constexpr int G4 = NG / 4, A4 = KG / 4;
  static_assert(WR * WC == 8 && W <= kWgmMaxW && 2 * BUF * 4 <= kWgmLdsBytes, "shape does not fit the workgroup / LDS");
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wr = wave / WC, wc = wave % WC;
  const int r0 = chunk * job.rows_per_task;
  int r1 = r0 + job.rows_per_task;
  if (r1 > job.rows) r1 = job.rows;
  const int nrows = r1 > r0 ? r1 - r0 : 0;
  const int nsteps = (nrows + kWgmRows - 1) / kWgmRows;
  const int total = nsteps * job.nvar;
  const int gstride = job.g.stride;
  const long gvs = job.g.vstride;
  const int* gidx = job.g.idx + r0;
  const int split4 = job.wcol[1] > 0 ? job.wcol[1] / 4 : A4;          // float4 columns served by act[0]; the rest by act[1]

  // Staging: thread tid owns float4 column tid % G4 of the gradient rows tid / G4 + RG i (i = 0 .. GL-1; threads past
  // RG * G4 idle), likewise for the activation rows: one source pointer and one LDS offset per thread, every slot a
  // compile-time multiple of the row stride away.
  constexpr int RG = kWgmThreads / G4, RA = kWgmThreads / A4;
  constexpr int GLs = (kWgmRows + RG - 1) / RG, ALs = (kWgmRows + RA - 1) / RA;
  const int grow0 = tid / G4, gc4 = tid % G4, arow0 = tid / A4, ac4 = tid % A4;
  const bool gthread = grow0 < RG, athread = arow0 < RA;
  const bool second = ac4 >= split4;
  const WsSeg& asg = second ? job.act[1] : job.act[0];
  const int astride = asg.stride;
  const long avs = asg.vstride;
  const float* abase = asg.ptr + asg.col0 + (long)(r0 + arow0) * astride + 4 * (second ? ac4 - split4 : ac4);
  const float* gbase = job.g.ptr + job.g.col0 + 4 * gc4 + (GATHER ? 0L : (long)(r0 + grow0) * gstride);
  const int gdst0 = grow0 * PITCH + 2 * gc4, adst0 = arow0 * PITCH + NG / 2 + 2 * ac4;
  constexpr bool DEEP = (B32 ? 4 : 1) * MBW * NBW <= 9;      // accumulator registers: 16 per 32 x 32 block, 4 per 16 x 16 block
  v4f gx[DEEP ? 2 : 1][GLs], ax[DEEP ? 2 : 1][ALs];
  v4f bs = {0.f, 0.f, 0.f, 0.f};
  int gi[GLs];
#pragma unroll
  for (int i = 0; i < GLs; ++i) gi[i] = 0;
  const v4f zero4 = {0.f, 0.f, 0.f, 0.f};

  auto load_idx = [&](int t) {                               // gather indices of step t (GATHER only)
    if constexpr (GATHER) {
      const int s = (t < total ? t : 0) % nsteps;
#pragma unroll
      for (int i = 0; i < GLs; ++i) {
        const int row = kWgmRows * s + grow0 + RG * i;
        gi[i] = (t < total && gthread && grow0 + RG * i < kWgmRows && row < nrows) ? gidx[row] : 0;
      }
    }
  };
  auto load_rows = [&](int t, v4f (&gq)[GLs], v4f (&aq)[ALs]) {    // rows of step t -> registers
    const bool live = t < total;
    const int tt = live ? t : 0;
    const int v = tt / nsteps, s = tt - v * nsteps;
    const float* gv = gbase + v * gvs + (GATHER ? 0L : (long)(kWgmRows * s) * gstride);
    const float* av = abase + v * avs + (long)(kWgmRows * s) * astride;
    const int left = nrows - kWgmRows * s;                   // rows of this step that exist
#pragma unroll
    for (int i = 0; i < GLs; ++i) {
      const int row = grow0 + RG * i;
      const bool ok = live && gthread && row < kWgmRows && row < left;
      const float* ptr = GATHER ? gv + (long)gi[i] * gstride : gv + (long)(RG * i) * gstride;
      gq[i] = ok ? *(gbl_v4f_p)ptr : zero4;
    }
#pragma unroll
    for (int i = 0; i < ALs; ++i) {
      const int row = arow0 + RA * i;
      const bool ok = live && athread && row < kWgmRows && row < left;
      aq[i] = ok ? *(gbl_v4f_p)(av + (long)(RA * i) * astride) : zero4;
    }
  };
  auto put = [&](lds_float* buf, int off, const v4f x) {
    unsigned p0[2], p1[2], p2[2];
    wgm_split4(x, p0, p1, p2);
    typedef unsigned wgm_u2 __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) wgm_u2 lds_u2;
    lds_float* d = buf + off;
    *(lds_u2*)d = wgm_u2{p0[0], p0[1]};
    *(lds_u2*)(d + PIECE) = wgm_u2{p1[0], p1[1]};
    *(lds_u2*)(d + 2 * PIECE) = wgm_u2{p2[0], p2[1]};
  };
  auto stage = [&](lds_float* buf, const v4f (&gq)[GLs], const v4f (&aq)[ALs]) {    // registers -> three bf16 piece images
#pragma unroll
    for (int i = 0; i < GLs; ++i) {
      if (gthread && grow0 + RG * i < kWgmRows) {
        put(buf, gdst0 + RG * i * PITCH, gq[i]);
        bs += gq[i];
      }
    }
#pragma unroll
    for (int i = 0; i < ALs; ++i)
      if (athread && arow0 + RA * i < kWgmRows) put(buf, adst0 + RA * i * PITCH, aq[i]);
  };

  using acc_t = std::conditional_t<B32, v16f, v4f>;
  acc_t acc[MBW][NBW];
#pragma unroll
  for (int a = 0; a < MBW; ++a)
#pragma unroll
    for (int b = 0; b < NBW; ++b) {
      if constexpr (B32) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
      }
      else acc[a][b] = zero4;
    }

  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  // 16 x 16 x 32: lane group g takes rows {4g..4g+3} and {16+4g..} as its eight k-slots.  32 x 32 x 16: lane = (m = lane % 32,
  // kg = lane / 32) takes rows {8 kg .. 8 kg + 7} of a 16-row half; its 16-lane group reads column block (lane / 16) % 2
  const int lane_off = B32 ? (8 * (lane >> 5) + q) * PITCH + 2 * p + 8 * ((lane >> 4) & 1) : (4 * g + q) * PITCH + 2 * p;
  auto frag32 = [&](const lds_float* img, int cd) {                 // 8 k-slots (rows) of this lane's column, one piece
    auto* p0 = (__attribute__((address_space(3))) wgm_s4*)(img + lane_off + cd);
    auto* p1 = (__attribute__((address_space(3))) wgm_s4*)(img + lane_off + 4 * PITCH + cd);
    const wgm_s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p0);
    const wgm_s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p1);
    const wgm_s8 v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    return __builtin_bit_cast(bf8, v);
  };
  auto mfma6_32 = [&](const Bf3& x, const Bf3& w, v16f c) {          // smallest terms first, as bf_mfma6
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x.p0, w.p2, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x.p1, w.p1, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x.p2, w.p0, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x.p0, w.p1, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x.p1, w.p0, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x.p0, w.p0, c, 0, 0, 0);
    return c;
  };
  auto compute = [&](const lds_float* cur) {
    if constexpr (B32) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {                                  // rows 16 h .. 16 h + 15 of the step
        const lds_float* ch = cur + 16 * h * PITCH;
        Bf3 af[MBW];
#pragma unroll
        for (int a = 0; a < MBW; ++a) {
          const int cd = 16 * (wr * MBW + a);
          af[a].p0 = frag32(ch, cd);
          af[a].p1 = frag32(ch + PIECE, cd);
          af[a].p2 = frag32(ch + 2 * PIECE, cd);
        }
#pragma unroll
        for (int b = 0; b < NBW; ++b) {
          const int cd = NG / 2 + 16 * (wc * NBW + b);
          Bf3 bfr;
          bfr.p0 = frag32(ch, cd);
          bfr.p1 = frag32(ch + PIECE, cd);
          bfr.p2 = frag32(ch + 2 * PIECE, cd);
#pragma unroll
          for (int a = 0; a < MBW; ++a) acc[a][b] = mfma6_32(af[a], bfr, acc[a][b]);
        }
      }
      return;
    } else {
    Bf3 af[MBW];
#pragma unroll
    for (int a = 0; a < MBW; ++a) {
      const int cd = 8 * (wr * MBW + a);
      af[a].p0 = wgm_frag(cur, lane_off, PITCH, cd);
      af[a].p1 = wgm_frag(cur + PIECE, lane_off, PITCH, cd);
      af[a].p2 = wgm_frag(cur + 2 * PIECE, lane_off, PITCH, cd);
    }
#pragma unroll
    for (int b = 0; b < NBW; ++b) {
      const int cd = NG / 2 + 8 * (wc * NBW + b);
      Bf3 bfr;
      bfr.p0 = wgm_frag(cur, lane_off, PITCH, cd);
      bfr.p1 = wgm_frag(cur + PIECE, lane_off, PITCH, cd);
      bfr.p2 = wgm_frag(cur + 2 * PIECE, lane_off, PITCH, cd);
#pragma unroll
      for (int a = 0; a < MBW; ++a) acc[a][b] = bf_mfma6(af[a], bfr, acc[a][b]);
    }
    }
  };
  // Prefetch depth: rows are fetched TWO steps ahead where the registers allow it (a step of a small shape is far
  // shorter than an HBM round trip); the shapes with 12+ accumulator blocks per wavefront keep one step in flight
  // -- their MFMA phase (>= 1 us) covers most of the latency and a second register set made them spill.
  __syncthreads();                                           // the previous task of this workgroup is done with the LDS
  if constexpr (DEEP) {
    // at the top of iteration t, LDS buffer t & 1 holds step t, register set (t + 1) & 1 step t + 1
    load_idx(0);
    load_rows(0, gx[0], ax[0]);
    load_idx(1);
    load_rows(1, gx[1], ax[1]);
    load_idx(2);
    stage(lds, gx[0], ax[0]);
    __syncthreads();
    for (int t = 0; t < total; t += 2) {
      load_rows(t + 2, gx[0], ax[0]);
      load_idx(t + 3);
      compute(lds);
      stage(lds + BUF, gx[1], ax[1]);                        // step t + 1 (zeros past the end)
      __syncthreads();
      if (t + 1 >= total) break;
      load_rows(t + 3, gx[1], ax[1]);
      load_idx(t + 4);
      compute(lds + BUF);
      stage(lds, gx[0], ax[0]);                              // step t + 2
      __syncthreads();
    }
  } else {
    load_idx(0);
    load_rows(0, gx[0], ax[0]);
    load_idx(1);
    stage(lds, gx[0], ax[0]);
    __syncthreads();
    for (int t = 0; t < total; ++t) {
      load_rows(t + 1, gx[0], ax[0]);                        // in flight during this step's MFMAs
      load_idx(t + 2);
      compute(lds + (t & 1) * BUF);
      stage(lds + ((t + 1) & 1) * BUF, gx[0], ax[0]);
      __syncthreads();
    }
  }

  // ---- partial -> slab (row-major [NP][KP], then bias) ---------------------------------------
  float* slab = job.slab + (size_t)chunk * ((size_t)job.NP * job.KP + job.NP);
  if constexpr (B32) {
    // register r of lane l of a 32 x 32 block: row 8 (r / 4) + 4 (l / 32) + r % 4, column l % 32
    const int n = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int a = 0; a < MBW; ++a) {
#pragma unroll
      for (int b = 0; b < NBW; ++b) {
        const int colf = job.wcol[0] + 32 * (wc * NBW + b) + n;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rowf = job.wrow + 32 * (wr * MBW + a) + 8 * (r >> 2) + 4 * hh + (r & 3);
          slab[(size_t)rowf * job.KP + colf] = acc[a][b][r];
        }
      }
    }
  } else {
  const int n = lane & 15, qq = lane >> 4;
#pragma unroll
  for (int a = 0; a < MBW; ++a) {
#pragma unroll
    for (int b = 0; b < NBW; ++b) {
      const float* vv = reinterpret_cast<const float*>(&acc[a][b]);
      const int colf = job.wcol[0] + 16 * (wc * NBW + b) + n;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int rowf = job.wrow + 16 * (wr * MBW + a) + 4 * qq + j;
        slab[(size_t)rowf * job.KP + colf] = vv[j];
      }
    }
  }
  }
  if (job.write_bias) {
    // column sums of G in a FIXED order (no float atomics: the result must not depend on which wavefront arrives first):
    // thread (grow0, gc4) parks its partial float4 at [grow0][gc4]; thread i then adds the RG partials of column i in
    // row order.  Every wavefront is past its last LDS read (barrier above).
    lds_float* colsum = lds;
    if (gthread) {
      lds_float* d = colsum + (grow0 * G4 + gc4) * 4;
      d[0] = bs.x; d[1] = bs.y; d[2] = bs.z; d[3] = bs.w;
    }
    __syncthreads();
    for (int i = tid; i < NG; i += kWgmThreads) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < RG; ++r) s += colsum[r * NG + i];
      slab[(size_t)job.NP * job.KP + job.wrow + i] = s;
    }
  }
}

static __global__ __launch_bounds__(kWgmThreads, 1) void wgemm_kernel(const WsJob* __restrict__ table, const int* __restrict__ task_job,
                                                                      int total_tasks, const int* __restrict__ iota) {
  extern __shared__ __attribute__((aligned(16))) float wgm_lds[];
  __shared__ WsJob sj;
  for (int task = blockIdx.x; task < total_tasks; task += gridDim.x) {
    __syncthreads();
    {
      const int* srcw = reinterpret_cast<const int*>(&table[task_job[task]]);
      int* dstw = reinterpret_cast<int*>(&sj);
      for (int i = threadIdx.x; i < (int)(sizeof(WsJob) / 4); i += kWgmThreads) dstw[i] = srcw[i];
    }
    __syncthreads();
    const WsJob& job = sj;
    const int chunk = task - job.task_begin;
    const bool gather = job.g.idx != iota;
    switch (job.shape) {
#if B3D_WGM_B32
      // shapes whose dW tile divides into 32 x 32 blocks over 8 wavefronts: (wave grid) x (32-blocks per wavefront)
      case WGM_128_256: wgm_task<2, 4, 2, 2, false, true>(job, chunk, (lds_float*)wgm_lds); break;
      case WGM_128_192:
        if (gather) wgm_task<4, 2, 1, 3, true, true>(job, chunk, (lds_float*)wgm_lds);
        else wgm_task<4, 2, 1, 3, false, true>(job, chunk, (lds_float*)wgm_lds);
        break;
      case WGM_128_128: wgm_task<4, 2, 1, 2, false, true>(job, chunk, (lds_float*)wgm_lds); break;
      case WGM_64_128: wgm_task<2, 4, 1, 1, false, true>(job, chunk, (lds_float*)wgm_lds); break;
      case WGM_256_64: wgm_task<8, 1, 1, 2, false, true>(job, chunk, (lds_float*)wgm_lds); break;
      case WGM_256_96: wgm_task<8, 1, 1, 3, false, true>(job, chunk, (lds_float*)wgm_lds); break;
      case WGM_192_128: wgm_task<2, 4, 3, 1, false, true>(job, chunk, (lds_float*)wgm_lds); break;
      case WGM_256_128: wgm_task<4, 2, 2, 2, false, true>(job, chunk, (lds_float*)wgm_lds); break;
#else
      case WGM_128_256: wgm_task<2, 4, 4, 4, false>(job, chunk, (lds_float*)wgm_lds); break;
      case WGM_128_192:
        if (gather) wgm_task<2, 4, 4, 3, true>(job, chunk, (lds_float*)wgm_lds);
        else wgm_task<2, 4, 4, 3, false>(job, chunk, (lds_float*)wgm_lds);
        break;
      case WGM_128_128: wgm_task<2, 4, 4, 2, false>(job, chunk, (lds_float*)wgm_lds); break;
      case WGM_64_128: wgm_task<2, 4, 2, 2, false>(job, chunk, (lds_float*)wgm_lds); break;
      case WGM_256_64: wgm_task<8, 1, 2, 4, false>(job, chunk, (lds_float*)wgm_lds); break;
      case WGM_256_96: wgm_task<8, 1, 2, 6, false>(job, chunk, (lds_float*)wgm_lds); break;
      case WGM_192_128: wgm_task<2, 4, 6, 2, false>(job, chunk, (lds_float*)wgm_lds); break;
      case WGM_256_128: wgm_task<8, 1, 2, 8, false>(job, chunk, (lds_float*)wgm_lds); break;
#endif
      // 12 or 18 blocks of 32 x 32 do not divide over 8 wavefronts: these stay on 16 x 16 x 32
      case WGM_192_64: wgm_task<4, 2, 3, 2, false>(job, chunk, (lds_float*)wgm_lds); break;
      case WGM_192_96: wgm_task<4, 2, 3, 3, false>(job, chunk, (lds_float*)wgm_lds); break;
      case WGM_96_128: wgm_task<2, 4, 3, 2, false>(job, chunk, (lds_float*)wgm_lds); break;
      case WGM_128_96: wgm_task<4, 2, 2, 3, false>(job, chunk, (lds_float*)wgm_lds); break;
      default: break;
    }
  }
}

}  // namespace b3d
