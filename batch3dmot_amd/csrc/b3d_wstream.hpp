// Streaming weight gradient (the large, 16-aligned Linear layers of the message-passing stacks).
//
//   dW[n][k] = sum over layers v, rows r of  G_v[r][n] * Act_v[r][k]        db[n] = sum G_v[r][n]
//
// The GNN applies ONE set of message-passing weights in all of its layers, so the contraction runs
// over (layer, edge): a single launch after the backward sweep handles every layer.
//
// One wavefront owns a whole weight matrix for a range of rows ("task").  Per step it takes 4 rows:
// lane (m, q) loads row r+q -- 16 bytes at a time, feature order permuted so that the 16 lanes of a
// quarter cover 64 consecutive features with one dwordx4 each -- straight into the registers that
// v_mfma_f32_16x16x4_f32 consumes as A (gradient features) and B (activation features); the rows are
// the MFMA K index.  Every byte of G and Act is read exactly once, by exactly one wavefront, in full
// 256-byte segments; there is no LDS staging, no barrier and no float atomic.  All (NP/16)x(KP/16)
// accumulator blocks stay in registers for the whole task; the partial goes to a per-task slab with
// plain stores and a later fixed-order sum makes the result bitwise reproducible.
//
// Gathered operands (x[dst], x[src], dM[dst] ...) take their row through an index that is fetched two
// steps ahead of the row itself, which is fetched two steps ahead of its use.
#pragma once
#include "b3d_common.hpp"
#include "b3d_dev.hpp"

namespace b3d {

typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int kWsMaxJobs = 40;
constexpr int kWsWaves = 4;            // tasks per workgroup

struct WsSeg {
  const float* ptr;      // [*, stride] of layer variant 0
  const int* idx;        // row gather (never null: identity segments use an iota array)
  long vstride;          // floats between consecutive layer variants
  int stride;            // floats per row
  int col0;
};

struct WsJob {
  WsSeg g;               // gradient rows, width GW
  WsSeg act[3];          // activation segments, widths S0, S1, S2
  int wcol[3];           // column of dW where each segment's features start
  int wrow;              // row of dW where this job's gradient features start
  int write_bias;        // exactly one of the jobs that share a slab writes the bias gradient
  int shape;             // index into the compiled shape list
  int rows;
  int nvar;              // layer variants accumulated into the same partial
  int rows_per_task;     // multiple of 4
  int ntasks;
  int NP, KP;
  float* slab;           // [ntasks][NP*KP + NP]
  int task_begin;        // first task (wavefront) of this job in the launch
};

// Jobs live in a device table (a backward pass of the camera+LiDAR+radar model has > 100 of them);
// ws_table_kernel copies them there kWsMaxJobs at a time through its kernel arguments.
struct WsTableArgs {
  int n;
  int first;             // index of jobs[0] in the device table
  WsJob* table;
  int* task_job;         // [total tasks] job index of every task
  WsJob jobs[kWsMaxJobs];
};

// Feature <-> (virtual 16-block, lane) map of a W-wide segment loaded with 16/8/4-byte pieces:
// groups of 64 features (dwordx4: lane j holds 4j..4j+3), then 32 (dwordx2), then 16 (dword).
template <int W>
struct SegMap {
  static constexpr int n4 = W / 64, n2 = (W % 64) / 32, n1 = (W % 32) / 16;
  static constexpr int NB = W / 16;
  static_assert(W % 16 == 0, "segment width must be a multiple of 16");
  __device__ static constexpr int feat(int vb, int j) {
    if (vb < 4 * n4) return 64 * (vb / 4) + 4 * j + (vb % 4);
    vb -= 4 * n4;
    if (vb < 2 * n2) return 64 * n4 + 2 * j + vb;
    return 64 * n4 + 32 * n2 + j;
  }
};

// Plain loads only: nothing at load time may depend on the loaded VALUES (rows that must not
// contribute are redirected by the caller to an all-zero gradient row).
template <int W>
__device__ __forceinline__ void seg_load(const float* __restrict__ p, int m, float* __restrict__ out) {
  using M = SegMap<W>;
#pragma unroll
  for (int g = 0; g < M::n4; ++g) {
    const v4f t = *reinterpret_cast<const v4f*>(p + 64 * g + 4 * m);
    out[4 * g + 0] = t.x; out[4 * g + 1] = t.y; out[4 * g + 2] = t.z; out[4 * g + 3] = t.w;
  }
  if constexpr (M::n2 > 0) {
    const v2f t = *reinterpret_cast<const v2f*>(p + 64 * M::n4 + 2 * m);
    out[4 * M::n4 + 0] = t.x; out[4 * M::n4 + 1] = t.y;
  }
  if constexpr (M::n1 > 0) {
    out[4 * M::n4 + 2 * M::n2] = p[64 * M::n4 + 32 * M::n2 + m];
  }
}
template <int GW, int S0, int S1, int S2>
struct WsShape {
  static constexpr int MB = GW / 16, K = S0 + S1 + S2, NB = K / 16;
  static constexpr int B0 = S0 / 16, B1 = S1 / 16, B2 = S2 / 16;
};

template <class SH>
struct WsStage {
  float a[SH::MB];
  float b[SH::NB];
};

// Up to 48 accumulator blocks (192 registers, the MFMA accumulator file) per wavefront: a 96x128
// matrix is owned whole, so its gradient rows are read once.  Wider matrices are split into
// (row group x column group) jobs that share one slab.
template <int GW, int S0, int S1, int S2>
__device__ __forceinline__ void ws_task(const WsJob& job, int chunk, const float* __restrict__ zero_row) {
  using SH = WsShape<GW, S0, S1, S2>;
  constexpr int MB = SH::MB, NB = SH::NB;
  static_assert(MB * NB <= 24, "too many accumulator blocks for one wavefront");
  const int lane = threadIdx.x & 63, m = lane & 15, q = lane >> 4;
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
  const int nsteps = (r1 > r0) ? (r1 - r0 + 3) / 4 : 0;
  const int gstride = job.g.stride, s0stride = job.act[0].stride, s1stride = (S1 > 0) ? job.act[1].stride : 0,
            s2stride = (S2 > 0) ? job.act[2].stride : 0;
  // every segment carries an index array (identity segments point at an iota array), so the loop
  // body is straight-line code: loads, selects, MFMAs.
  const int* ig = job.g.idx;
  const int* i0 = job.act[0].idx;
  const int* i1 = (S1 > 0) ? job.act[1].idx : i0;
  const int* i2 = (S2 > 0) ? job.act[2].idx : i0;

  for (int v = 0; v < job.nvar; ++v) {
    const float* gp = job.g.ptr + v * job.g.vstride + job.g.col0;
    const float* p0 = job.act[0].ptr + v * job.act[0].vstride + job.act[0].col0;
    const float* p1 = (S1 > 0) ? job.act[1].ptr + v * job.act[1].vstride + job.act[1].col0 : p0;
    const float* p2 = (S2 > 0) ? job.act[2].ptr + v * job.act[2].vstride + job.act[2].col0 : p0;

    struct Rows { int g, a0, a1, a2; bool ok; };
    auto rows_of = [&](int step) {
      Rows r;
      const int row = r0 + 4 * step + q;
      r.ok = (step < nsteps) && (row < r1);
      const int rr = r.ok ? row : r0;
      r.g = ig[rr];
      r.a0 = i0[rr];
      r.a1 = (S1 > 0) ? i1[rr] : 0;
      r.a2 = (S2 > 0) ? i2[rr] : 0;
      return r;
    };
    auto fetch = [&](const Rows& r, WsStage<SH>& st) {
      const float* pg = r.ok ? gp + (long)r.g * gstride : zero_row;   // zero gradient row: no contribution
      seg_load<GW>(pg, m, st.a);
      seg_load<S0>(p0 + (long)r.a0 * s0stride, m, st.b);
      if constexpr (S1 > 0) seg_load<S1>(p1 + (long)r.a1 * s1stride, m, st.b + SH::B0);
      if constexpr (S2 > 0) seg_load<S2>(p2 + (long)r.a2 * s2stride, m, st.b + SH::B0 + SH::B1);
    };
    auto compute = [&](const WsStage<SH>& st) {
#pragma unroll
      for (int a = 0; a < MB; ++a) bsum[a] += st.a[a];
#pragma unroll
      for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(st.a[a], st.b[b], acc[a][b], 0, 0, 0);
    };

    // Blocks of BS steps (4 BS rows), two register sets.  While block k is multiplied, the rows of
    // block k+1 are in flight (issued right after the first step of block k) and the gather indices
    // of block k+2 are being fetched: the only wait on the critical path is the one in front of a
    // block's first step, BS-1 steps of MFMAs after its loads were issued.
    constexpr int BS = (MB * NB >= 24) ? 2 : 4;
    WsStage<SH> SA[BS], SB[BS];
    Rows RA[BS], RB[BS];
    const int nblk = (nsteps + BS - 1) / BS;
#pragma unroll
    for (int i = 0; i < BS; ++i) RA[i] = rows_of(i);
#pragma unroll
    for (int i = 0; i < BS; ++i) RB[i] = rows_of(BS + i);
#pragma unroll
    for (int i = 0; i < BS; ++i) fetch(RA[i], SA[i]);
#pragma unroll
    for (int i = 0; i < BS; ++i) RA[i] = rows_of(2 * BS + i);
    for (int k = 0; k < nblk; k += 2) {
      compute(SA[0]);                                   // block k
#pragma unroll
      for (int i = 0; i < BS; ++i) fetch(RB[i], SB[i]); // block k+1
#pragma unroll
      for (int i = 0; i < BS; ++i) RB[i] = rows_of((k + 3) * BS + i);
#pragma unroll
      for (int i = 1; i < BS; ++i) compute(SA[i]);
      compute(SB[0]);                                   // block k+1 (zero rows beyond the range)
#pragma unroll
      for (int i = 0; i < BS; ++i) fetch(RA[i], SA[i]); // block k+2
#pragma unroll
      for (int i = 0; i < BS; ++i) RA[i] = rows_of((k + 4) * BS + i);
#pragma unroll
      for (int i = 1; i < BS; ++i) compute(SB[i]);
    }
  }

  // ---- partial -> slab (row-major [NP][KP], then bias) ---------------------------------------
  float* slab = job.slab + (size_t)chunk * ((size_t)job.NP * job.KP + job.NP);
#pragma unroll
  for (int a = 0; a < MB; ++a) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const float* vv = reinterpret_cast<const float*>(&acc[a][b]);
      const int colf = (b < SH::B0) ? job.wcol[0] + SegMap<S0>::feat(b, m)
                       : (b < SH::B0 + SH::B1) ? job.wcol[1] + SegMap<(S1 > 0 ? S1 : 16)>::feat(b - SH::B0, m)
                                               : job.wcol[2] + SegMap<(S2 > 0 ? S2 : 16)>::feat(b - SH::B0 - SH::B1, m);
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

// compiled shapes (GW; S0, S1, S2).  At most 24 accumulator blocks (96 VGPRs) per wavefront so that
// two wavefronts fit a SIMD without touching the AGPR half of the register file (the compiler
// otherwise shuttles accumulators through v_accvgpr moves that stall behind the MFMAs).
enum {
  WS_96_48_16 = 0,      // P 96x128 stacks, columns [48-wide segment | 16 columns of the next]
  WS_96_32_32 = 1,      // P edge_update.0, columns [x[src] 16:48 | e]
  WS_64_96 = 2,
  WS_32_64 = 3,
  WS_48_64 = 4,
  WS_96_64 = 5,         // P combine_future_past.0 (two column halves)
  WS_64_64 = 6,
  // narrow stacks (encoders, classifier): widths padded to 16 by their row stride
  WS_16_16 = 7,
  WS_16_32 = 8,
  WS_32_16 = 9,
  WS_32_32 = 10,
  WS_48_32 = 11,
  WS_48_48 = 12,
  // hoisted first layers (per-node and per-edge parts of a Linear over a concatenation)
  WS_96_32 = 13,
  WS_96_48 = 14,
  WS_SHAPES = 15
};

static __global__ void ws_table_kernel(const WsTableArgs a) {
  const int j = blockIdx.x;
  if (j >= a.n) return;
  const int* srcw = reinterpret_cast<const int*>(&a.jobs[j]);
  int* dstw = reinterpret_cast<int*>(&a.table[a.first + j]);
  for (int i = threadIdx.x; i < (int)(sizeof(WsJob) / 4); i += blockDim.x) dstw[i] = srcw[i];
  for (int t = threadIdx.x; t < a.jobs[j].ntasks; t += blockDim.x) a.task_job[a.jobs[j].task_begin + t] = a.first + j;
}

static __global__ __launch_bounds__(kWsWaves * 64, 2) void wstream_kernel(const WsJob* __restrict__ table,
                                                                   const int* __restrict__ task_job, int total_tasks,
                                                                   const float* __restrict__ zero_row) {
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
  switch (job.shape) {
    case WS_96_48_16: ws_task<96, 48, 16, 0>(job, chunk, zero_row); break;
    case WS_96_32_32: ws_task<96, 32, 32, 0>(job, chunk, zero_row); break;
    case WS_64_96: ws_task<64, 96, 0, 0>(job, chunk, zero_row); break;
    case WS_32_64: ws_task<32, 64, 0, 0>(job, chunk, zero_row); break;
    case WS_48_64: ws_task<48, 64, 0, 0>(job, chunk, zero_row); break;
    case WS_96_64: ws_task<96, 64, 0, 0>(job, chunk, zero_row); break;
    case WS_64_64: ws_task<64, 64, 0, 0>(job, chunk, zero_row); break;
    case WS_16_16: ws_task<16, 16, 0, 0>(job, chunk, zero_row); break;
    case WS_16_32: ws_task<16, 32, 0, 0>(job, chunk, zero_row); break;
    case WS_32_16: ws_task<32, 16, 0, 0>(job, chunk, zero_row); break;
    case WS_32_32: ws_task<32, 32, 0, 0>(job, chunk, zero_row); break;
    case WS_48_32: ws_task<48, 32, 0, 0>(job, chunk, zero_row); break;
    case WS_48_48: ws_task<48, 48, 0, 0>(job, chunk, zero_row); break;
    case WS_96_32: ws_task<96, 32, 0, 0>(job, chunk, zero_row); break;
    case WS_96_48: ws_task<96, 48, 0, 0>(job, chunk, zero_row); break;
    default: break;
  }
}

// blocks (MFMAs per 4-row step) of a shape: the unit of work used to balance tasks
inline int ws_shape_blocks(int shape) {
  static const int b[WS_SHAPES] = {6 * 4, 6 * 4, 4 * 6, 2 * 4, 3 * 4, 6 * 4, 4 * 4, 1, 2, 2, 4, 6, 9, 6 * 2, 6 * 3};
  return b[shape];
}

// Host side: collects jobs, uploads the table, launches.
struct WsLauncher {
  WsJob* table;          // device, capacity `cap`
  int* task_job;         // device, capacity `task_cap`
  int cap, task_cap;
  int njobs, total_tasks;
  WsTableArgs pending;
  hipStream_t stream;
  int status;
  void begin(WsJob* t, int c, int* tj, int tc, hipStream_t s) {
    table = t; cap = c; task_job = tj; task_cap = tc; njobs = 0; total_tasks = 0; stream = s; status = 0;
    pending.n = 0; pending.first = 0; pending.table = t; pending.task_job = tj;
  }
  void flush() {
    if (pending.n == 0) return;
    hipLaunchKernelGGL(ws_table_kernel, dim3(pending.n), dim3(64), 0, stream, pending);
    pending.first += pending.n;
    pending.n = 0;
  }
  // job.task_begin / ntasks are filled here from rows / rows_per_task
  void add(WsJob job) {
    if (njobs >= cap || status) { status = -1; return; }
    job.ntasks = (job.rows + job.rows_per_task - 1) / job.rows_per_task;
    if (job.ntasks < 1) job.ntasks = 1;
    job.task_begin = total_tasks;
    if (total_tasks + job.ntasks > task_cap) { status = -1; return; }
    total_tasks += job.ntasks;
    pending.jobs[pending.n++] = job;
    ++njobs;
    if (pending.n == kWsMaxJobs) flush();
  }
  void launch(const float* zero_row, int family) {
    flush();
    if (total_tasks == 0 || status) return;
    ProfScope ps(family, stream);
    hipLaunchKernelGGL(wstream_kernel, dim3((total_tasks + kWsWaves - 1) / kWsWaves), dim3(kWsWaves * 64), 0, stream,
                       table, task_job, total_tasks, zero_row);
  }
  // LDS-DMA form (b3d_wstream2.hpp): single-segment jobs only
  template <class Kern>
  int launch2(Kern kernel, int lds_bytes, const float* zero_row, const int* iota, int family) {
    flush();
    if (total_tasks == 0 || status) return 0;
    if (lds_bytes > 64 * 1024 && set_lds_cached(reinterpret_cast<const void*>(kernel), lds_bytes) != 0) return -1;
    ProfScope ps(family, stream);
    hipLaunchKernelGGL(kernel, dim3((total_tasks + kWsWaves - 1) / kWsWaves), dim3(kWsWaves * 64), lds_bytes, stream,
                       table, task_job, total_tasks, zero_row, iota);
    return 0;
  }
};

}  // namespace b3d
