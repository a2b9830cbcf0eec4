// Streaming weight gradient, LDS-DMA form (same jobs, tasks and slabs as b3d_wstream.hpp).
//
// wstream (v1) keeps its prefetched rows in registers: 24 accumulator blocks + two register sets of
// 2-4 row steps fill the 256-VGPR budget of a 2-waves-per-SIMD kernel, so a wavefront has ~5 KB of
// HBM reads in flight and the launch runs at ~3.5 of the ~6.3 TB/s the chip can stream.  Here every
// wavefront owns a private LDS ring of kWs2Depth 8-row steps that global_load_lds fills directly: the
// rows in flight cost no registers, the steady loop contains no ordinary global load (hipcc waits
// vmcnt(0) on the first use of one while an LDS-DMA is pending), and the ring is retired with counted
// s_waitcnt vmcnt(N).  Gather indices (dM[dst] / dM[src] rows of the message stacks) travel through a
// second small ring the same way.  One activation segment per job (all jobs of the hoisted plan).
#pragma once
#include "b3d_wstream.hpp"

namespace b3d {

constexpr int kWs2Rows = 8;                                // rows per step = two MFMA k-steps
constexpr int kWs2Depth = 3;                               // row steps in flight per wavefront
constexpr int kWs2StepFloats = kWs2Rows * 160;             // largest (gradient + activation) row step: 64 + 96 features
constexpr int kWs2IdxSlots = 2 * kWs2Depth;                // an index is fetched 2 * depth steps ahead of its row
constexpr int kWs2WaveFloats = kWs2Depth * kWs2StepFloats + kWs2IdxSlots * 64;
constexpr int kWs2LdsBytes = kWsWaves * kWs2WaveFloats * 4;

// W-wide segment of an 8-row step, as LDS-DMA pieces (one wave-instruction each, 64 lanes x 16 or 4 bytes):
//   * groups of 64 features: two 16-byte pieces (rows 0-3 and 4-7; lane (m, q): row 4h + q, features 64 g + 4 m ..)
//   * a remainder of 32:     ONE 16-byte piece over all 8 rows (lane l: row l >> 3, features 4 (l & 7) ..)
//   * a remainder of 16:     two 4-byte pieces (rows 0-3 and 4-7; lane (m, q): row 4h + q, feature m)
// The operands of k-step h come back with b128 / b64 / b32 LDS reads in the feature order of SegMap<W>.
template <int W>
struct Seg2 {
  static constexpr int n4 = W / 64, n2 = (W % 64) / 32, n1 = (W % 32) / 16;
  static constexpr int PIECES = 2 * n4 + n2 + 2 * n1, NB = W / 16;
  static constexpr int O2 = 512 * n4, O1 = O2 + 256 * n2;       // float offsets of the 32- and 16-wide parts
  static_assert(W % 16 == 0, "segment width must be a multiple of 16");
};

__device__ __forceinline__ void glds16(const float* g, float* lds) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}
__device__ __forceinline__ void glds4(const void* g, void* lds) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds, 4, 0, 0);
}

// rowp(r): pointer to column 0 of the segment in row r (0..7) of the step, for this lane
template <int W, class RowPtr>
__device__ __forceinline__ void seg2_issue(RowPtr rowp, float* lds, int lane) {
  using S = Seg2<W>;
  const int m = lane & 15, q = lane >> 4;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float* r = rowp(4 * h + q);
#pragma unroll
    for (int g = 0; g < S::n4; ++g) glds16(r + 64 * g + 4 * m, lds + 512 * g + 256 * h);
    if constexpr (S::n1 > 0) glds4(r + 64 * S::n4 + 32 * S::n2 + m, lds + S::O1 + 64 * h);
  }
  if constexpr (S::n2 > 0) glds16(rowp(lane >> 3) + 64 * S::n4 + 4 * (lane & 7), lds + S::O2);
}
template <int W>
__device__ __forceinline__ void seg2_read(const float* lds, int h, int lane, float* __restrict__ out) {
  using S = Seg2<W>;
  const int m = lane & 15, q = lane >> 4;
#pragma unroll
  for (int g = 0; g < S::n4; ++g) {
    const v4f t = *reinterpret_cast<const v4f*>(lds + 512 * g + 256 * h + 4 * lane);
    out[4 * g + 0] = t.x; out[4 * g + 1] = t.y; out[4 * g + 2] = t.z; out[4 * g + 3] = t.w;
  }
  if constexpr (S::n2 > 0) {
    const v2f t = *reinterpret_cast<const v2f*>(lds + S::O2 + 32 * (4 * h + q) + 2 * m);
    out[4 * S::n4 + 0] = t.x; out[4 * S::n4 + 1] = t.y;
  }
  if constexpr (S::n1 > 0) out[4 * S::n4 + 2 * S::n2] = lds[S::O1 + 64 * h + lane];
}

template <int N>
__device__ __forceinline__ void ws2_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int GW, int S0, bool GATHER>
__device__ __forceinline__ void ws2_task(const WsJob& job, int chunk, const float* __restrict__ zero_row, float* wlds) {
  constexpr int MB = GW / 16, NB = S0 / 16, P = kWs2Depth, R = kWs2Rows;
  constexpr int PC = Seg2<GW>::PIECES + Seg2<S0>::PIECES + (GATHER ? 1 : 0);   // VM instructions per step
  static_assert(MB * NB <= 24 && R * (GW + S0) <= kWs2StepFloats, "shape does not fit the accumulator / ring budget");
  static_assert((P - 1) * PC < 64, "vmcnt field");
  const int lane = threadIdx.x & 63, m = lane & 15, q = lane >> 4;
  float* ring = wlds;
  int* iring = reinterpret_cast<int*>(wlds + kWs2Depth * kWs2StepFloats);
  v4f acc[MB][NB];
#pragma unroll
  for (int a = 0; a < MB; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[a][b] = v4f{0.f, 0.f, 0.f, 0.f};
  float bsum[MB];
#pragma unroll
  for (int a = 0; a < MB; ++a) bsum[a] = 0.f;

  const int r0 = chunk * job.rows_per_task;
  int r1 = r0 + job.rows_per_task;
  if (r1 > job.rows) r1 = job.rows;
  const int nsteps = (r1 > r0) ? (r1 - r0 + R - 1) / R : 0;
  const int total = nsteps * job.nvar;                       // steps over all layer variants, one pipeline
  const int gstride = job.g.stride, astride = job.act[0].stride;
  const float* gbase = job.g.ptr + job.g.col0;
  const float* abase = job.act[0].ptr + job.act[0].col0;
  const long gvs = job.g.vstride, avs = job.act[0].vstride;
  const int* gidx = job.g.idx;

  // Two cursors walk the steps of all layer variants as ONE pipeline: `di` issues rows, `ii` issues gather
  // indices 2P steps ahead of their rows.  Rows past the range (and steps past the end) pair a zero
  // gradient row with activation row r0 of variant 0: no contribution, always a valid address.
  struct Cursor { int t, v, s; };
  auto advance = [&](Cursor& c) {
    ++c.t;
    if (++c.s == nsteps) { c.s = 0; ++c.v; }
  };
  auto issue_idx = [&](const Cursor& c) {
    if constexpr (GATHER) {
      const int row = r0 + R * c.s + (lane & 7);             // 8 rows per step: lanes 8.. repeat them
      const bool ok = (c.t < total) && (row < r1);
      glds4(gidx + (ok ? row : r0), iring + (c.t % kWs2IdxSlots) * 64);
    }
  };
  auto issue_data = [&](const Cursor& c) {
    const bool live = c.t < total;
    const int v = live ? c.v : 0;
    const int base_row = r0 + R * c.s;
    const int* islot = iring + (c.t % kWs2IdxSlots) * 64;
    const float* gv = gbase + v * gvs;
    const float* av = abase + v * avs;
    auto grow = [&](int r) -> const float* {
      const int row = base_row + r;
      if (!(live && row < r1)) return zero_row;
      if constexpr (GATHER) return gv + (long)islot[r] * gstride;
      return gv + (long)row * gstride;
    };
    auto arow = [&](int r) -> const float* {
      const int row = base_row + r;
      return av + (long)((live && row < r1) ? row : r0) * astride;
    };
    float* slot = ring + (c.t % P) * kWs2StepFloats;
    seg2_issue<GW>(grow, slot, lane);
    seg2_issue<S0>(arow, slot + R * GW, lane);
  };
  Cursor di{0, 0, 0}, ii{0, 0, 0};

  // Every step -- prologue included -- issues its VM instructions in the same order (one index piece, then the
  // row pieces), so that "all but the (P-1) * PC youngest" always covers the step about to be read.
  if constexpr (GATHER) {
#pragma unroll
    for (int t = 0; t < P; ++t) { issue_idx(ii); advance(ii); }
    ws2_wait<0>();
  }
#pragma unroll
  for (int t = 0; t < P; ++t) {
    issue_idx(ii); advance(ii);                              // index of step t + P
    issue_data(di); advance(di);                             // rows of step t
  }

  for (int t = 0; t < total; ++t) {
    ws2_wait<(P - 1) * PC>();                                // step t has landed (and every index issued before it)
    float sa[2][MB], sb[2][NB];
    const float* slot = ring + (t % P) * kWs2StepFloats;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      seg2_read<GW>(slot, h, lane, sa[h]);
      seg2_read<S0>(slot + R * GW, h, lane, sb[h]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // operands are in registers: the slot may be refilled
    issue_idx(ii); advance(ii);                              // index of step t + 2P (older than the rows below)
    issue_data(di); advance(di);                             // rows of step t + P into the slot just read
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int a = 0; a < MB; ++a) bsum[a] += sa[h][a];
#pragma unroll
      for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(sa[h][a], sb[h][b], acc[a][b], 0, 0, 0);
    }
  }
  ws2_wait<0>();                                             // drain the tail of the pipeline before the ring is reused

  // ---- partial -> slab (row-major [NP][KP], then bias) ---------------------------------------
  float* slab = job.slab + (size_t)chunk * ((size_t)job.NP * job.KP + job.NP);
#pragma unroll
  for (int a = 0; a < MB; ++a) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const float* vv = reinterpret_cast<const float*>(&acc[a][b]);
      const int colf = job.wcol[0] + SegMap<S0>::feat(b, m);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int rowf = job.wrow + SegMap<GW>::feat(a, 4 * q + j);
        slab[(size_t)rowf * job.KP + colf] = vv[j];
      }
    }
  }
  if (job.write_bias) {
#pragma unroll
    for (int a = 0; a < MB; ++a) {
      float sum = bsum[a];
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      if (q == 0) slab[(size_t)job.NP * job.KP + job.wrow + SegMap<GW>::feat(a, m)] = sum;
    }
  }
}

// job.gather: the gradient rows are gathered (job.g.idx is not the identity array)
static __global__ __launch_bounds__(kWsWaves * 64, 2) void wstream2_kernel(const WsJob* __restrict__ table,
                                                                    const int* __restrict__ task_job, int total_tasks,
                                                                    const float* __restrict__ zero_row,
                                                                    const int* __restrict__ iota) {
  extern __shared__ __attribute__((aligned(16))) float ws2_lds[];
  __shared__ WsJob sj[kWsWaves];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int task = blockIdx.x * kWsWaves + wave;
  if (task >= total_tasks) return;
  {
    const int* srcw = reinterpret_cast<const int*>(&table[task_job[task]]);
    int* dstw = reinterpret_cast<int*>(&sj[wave]);
    for (int i = lane; i < (int)(sizeof(WsJob) / 4); i += 64) dstw[i] = srcw[i];
  }
  __builtin_amdgcn_wave_barrier();
  const WsJob& job = sj[wave];
  const int chunk = task - job.task_begin;
  float* wlds = ws2_lds + wave * kWs2WaveFloats;
  const bool gather = job.g.idx != iota;
  switch (job.shape) {
    case WS_64_96:
      if (gather) ws2_task<64, 96, true>(job, chunk, zero_row, wlds);
      else ws2_task<64, 96, false>(job, chunk, zero_row, wlds);
      break;
    case WS_32_64: ws2_task<32, 64, false>(job, chunk, zero_row, wlds); break;
    case WS_48_64: ws2_task<48, 64, false>(job, chunk, zero_row, wlds); break;
    case WS_96_64: ws2_task<96, 64, false>(job, chunk, zero_row, wlds); break;
    case WS_64_64: ws2_task<64, 64, false>(job, chunk, zero_row, wlds); break;
    case WS_16_16: ws2_task<16, 16, false>(job, chunk, zero_row, wlds); break;
    case WS_16_32: ws2_task<16, 32, false>(job, chunk, zero_row, wlds); break;
    case WS_32_16: ws2_task<32, 16, false>(job, chunk, zero_row, wlds); break;
    case WS_32_32: ws2_task<32, 32, false>(job, chunk, zero_row, wlds); break;
    case WS_48_32: ws2_task<48, 32, false>(job, chunk, zero_row, wlds); break;
    case WS_48_48: ws2_task<48, 48, false>(job, chunk, zero_row, wlds); break;
    case WS_96_32: ws2_task<96, 32, false>(job, chunk, zero_row, wlds); break;
    case WS_96_48: ws2_task<96, 48, false>(job, chunk, zero_row, wlds); break;
    default: break;                                          // multi-segment shapes: wstream_kernel only
  }
}

}  // namespace b3d
