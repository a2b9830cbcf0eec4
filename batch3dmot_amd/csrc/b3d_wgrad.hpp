// Weight gradients: dW[n][k] = sum_rows G[row][n] * Act[row][k],  db[n] = sum_rows G[row][n].
//
// The contraction runs over the edge (or node) dimension, i.e. the rows are the MFMA K index:
// v_mfma_f32_16x16x4_f32 with A[m][k] = G[row k][16 mb + m], B[k][n] = Act[row k][16 nb + n].
// A workgroup owns a chunk of rows and (a group of 16-row blocks of) one weight matrix; it stages
// 32-row tiles of G and of the concatenated activation segments in LDS (coalesced float4 loads,
// row gathers through an index for x[src] / x[dst] / dM[dst] ...), every wave accumulates its
// 16x16 blocks of dW in registers over the whole chunk, and the partial lands in a per-chunk slab
// (plain stores, no float atomics: the later chunk-order sum is bitwise reproducible).  Because
// the GNN's weights are shared by its 6 layers, a layer's partial is accumulated into the same
// slab (read-modify-write by the owning workgroup) and one reduce kernel runs per backward.
#pragma once
#include "b3d_dev.hpp"

namespace b3d {

constexpr int kWgRT = 32;          // rows per LDS tile
constexpr int kWgMaxJobs = 8;
constexpr int kWgMaxSegs = 4;

struct WgSeg {
  const float* ptr;   // [*, stride]
  const int* idx;     // row gather (nullptr = identity)
  int stride;         // floats
  int col0;
  int width;          // floats taken from each row
  int aligned;        // 1: stride, col0, width all multiples of 4 and ptr 16-byte aligned
};

struct WgJob {
  WgSeg g;                    // G rows: width = N (true output width)
  WgSeg act[kWgMaxSegs];      // concatenated activation segments: sum of widths = K
  int nact;
  int NP, KP;                 // padded dims (multiples of 16)
  int rows;
  int rows_per_chunk;         // multiple of kWgRT
  int nchunks;
  int mgroups;                // NP/16 split over this many workgroups (MAXMB blocks each)
  float* slab;                // [nchunks][NP*KP + NP]
  int accumulate;             // 0: overwrite slab, 1: slab += partial
  int wg_begin;               // first workgroup of this job in the launch
};

struct WgArgs {
  int njobs;
  WgJob jobs[kWgMaxJobs];
};

__device__ __forceinline__ float wg_load1(const WgSeg& s, long r, int c) {
  return s.ptr[r * (long)s.stride + s.col0 + c];
}

// One float4 of the LDS tile and where it comes from (fixed per thread for the whole kernel).
struct WgSlot {
  const float* sp;   // source base + col0 + column (fast slots)
  const int* ip;     // row gather or nullptr
  int stride;
  int rl;            // tile-local row
  int lds_off;       // float offset inside the tile
  int seg;           // -1: G, >= 0: activation segment that holds element 0 of this float4
  int c;             // column inside that segment (for G: true output column)
  int mode;          // 0: always zero / unused, 1: one aligned 16-byte load, 2: element-wise (slow)
};

// MAXMB: 16-row output blocks per workgroup; MAXNBW: 16-col input blocks per wave (8 waves);
// SLOTS: float4 loads per thread per 32-row tile (ceil(32 * (16 MAXMB + KPmax) / 4 / 512)).
template <int MAXMB, int MAXNBW, int SLOTS>
__global__ __launch_bounds__(kThreads, 4) void wgrad_kernel(const WgArgs args) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ WgJob sjob;
  // ---- which job / chunk / output-row group ---------------------------------------------
  int j = 0;
#pragma unroll
  for (int t = 1; t < kWgMaxJobs; ++t)
    if (t < args.njobs && (int)blockIdx.x >= args.jobs[t].wg_begin) j = t;
  if (threadIdx.x < sizeof(WgJob) / 4) {
    reinterpret_cast<int*>(&sjob)[threadIdx.x] = reinterpret_cast<const int*>(&args.jobs[j])[threadIdx.x];
  }
  __syncthreads();
  const WgJob& job = sjob;
  const int local = blockIdx.x - job.wg_begin;
  const int chunk = local / job.mgroups, mg = local % job.mgroups;
  const int NB = job.KP / 16;
  const int mb_base = mg * MAXMB;
  int MBW = job.NP / 16 - mb_base;
  if (MBW > MAXMB) MBW = MAXMB;
  const int gw = MBW * 16;                       // G columns staged by this workgroup
  const int gstride = gw + 4, astride = job.KP + 4;
  float* Gl = smem;                              // [kWgRT][gstride]
  float* Al = smem + kWgRT * gstride;            // [kWgRT][astride]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int m = lane & 15, q = lane >> 4;

  // ---- per-thread load plan -----------------------------------------------------------------
  const int gc4 = gw / 4, ac4 = job.KP / 4, c4tot = gc4 + ac4;
  WgSlot slot[SLOTS];
#pragma unroll
  for (int e = 0; e < SLOTS; ++e) {
    const int id = threadIdx.x + e * kThreads;
    WgSlot sl;
    sl.sp = nullptr; sl.ip = nullptr; sl.stride = 0; sl.rl = 0; sl.lds_off = -1; sl.seg = -1; sl.c = 0; sl.mode = 0;
    if (id < kWgRT * c4tot) {
      const int rl = id / c4tot, c4 = id - rl * c4tot;
      sl.rl = rl;
      if (c4 < gc4) {
        sl.c = mb_base * 16 + c4 * 4;
        sl.lds_off = rl * gstride + c4 * 4;
        if (sl.c < job.g.width) {
          sl.mode = (job.g.aligned && sl.c + 4 <= job.g.width) ? 1 : 2;
          sl.sp = job.g.ptr + job.g.col0 + sl.c; sl.ip = job.g.idx; sl.stride = job.g.stride;
        }
      } else {
        int c = (c4 - gc4) * 4;
        sl.lds_off = kWgRT * gstride + rl * astride + c;
        for (int sgi = 0; sgi < job.nact; ++sgi) {
          const int w = job.act[sgi].width;
          if (c < w) {
            sl.seg = sgi; sl.c = c;
            sl.mode = (job.act[sgi].aligned && (c & 3) == 0 && c + 4 <= w) ? 1 : 2;
            sl.sp = job.act[sgi].ptr + job.act[sgi].col0 + c; sl.ip = job.act[sgi].idx; sl.stride = job.act[sgi].stride;
            break;
          }
          c -= w;
        }
      }
    }
    slot[e] = sl;
  }

  // element-wise path: unaligned source or a float4 that straddles segments (rare, small jobs)
  auto fetch_slow = [&](const WgSlot& sl, long row) -> v4f {
    v4f v = {0.f, 0.f, 0.f, 0.f};
    float* vp = reinterpret_cast<float*>(&v);
    if (sl.seg == -1) {
      const WgSeg& sg = job.g;
      const long r = sg.idx ? sg.idx[row] : row;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (sl.c + e < sg.width) vp[e] = wg_load1(sg, r, sl.c + e);
      return v;
    }
    int sgi = sl.seg, c = sl.c;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      while (sgi < job.nact && c >= job.act[sgi].width) { c -= job.act[sgi].width; ++sgi; }
      if (sgi < job.nact) {
        const WgSeg& s2 = job.act[sgi];
        const long r2 = s2.idx ? s2.idx[row] : row;
        vp[e] = wg_load1(s2, r2, c);
      }
      ++c;
    }
    return v;
  };

  // All slots of a thread are fetched together: first every row index, then every row chunk, so
  // that the (up to SLOTS) dependent index -> data chains overlap instead of running one by one.
  auto fetch_all = [&](long rt, long r1, v4f* pre) {
    long r[SLOTS];
    bool ok[SLOTS];
#pragma unroll
    for (int e = 0; e < SLOTS; ++e) {
      const long row = rt + slot[e].rl;
      ok[e] = (slot[e].mode == 1) && (row < r1);
      r[e] = row;
      if (ok[e] && slot[e].ip) r[e] = slot[e].ip[row];
    }
#pragma unroll
    for (int e = 0; e < SLOTS; ++e) {
      pre[e] = v4f{0.f, 0.f, 0.f, 0.f};
      if (ok[e]) pre[e] = *reinterpret_cast<const v4f*>(slot[e].sp + r[e] * (long)slot[e].stride);
    }
#pragma unroll
    for (int e = 0; e < SLOTS; ++e) {
      const long row = rt + slot[e].rl;
      if (slot[e].mode == 2 && row < r1) pre[e] = fetch_slow(slot[e], row);
    }
  };

  v4f acc[MAXMB][MAXNBW];
#pragma unroll
  for (int a = 0; a < MAXMB; ++a)
#pragma unroll
    for (int b = 0; b < MAXNBW; ++b) acc[a][b] = v4f{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;

  const long r0 = (long)chunk * job.rows_per_chunk;
  long r1 = r0 + job.rows_per_chunk;
  if (r1 > job.rows) r1 = job.rows;

  v4f pre[SLOTS];
  fetch_all(r0, r1, pre);

  for (long rt = r0; rt < r1; rt += kWgRT) {
    // ---- registers -> LDS tile, then prefetch the next tile while this one is multiplied ---
#pragma unroll
    for (int e = 0; e < SLOTS; ++e)
      if (slot[e].lds_off >= 0) *reinterpret_cast<v4f*>(smem + slot[e].lds_off) = pre[e];
    __syncthreads();
    if (rt + kWgRT < r1) fetch_all(rt + kWgRT, r1, pre);
    // ---- bias gradient: column sums of G -----------------------------------------------
    if ((int)threadIdx.x < gw) {
#pragma unroll 8
      for (int rl = 0; rl < kWgRT; ++rl) bsum += Gl[rl * gstride + threadIdx.x];
    }
    // ---- MFMA over the tile's 32 rows (8 k-steps of 4 rows) -----------------------------
#pragma unroll 2
    for (int st = 0; st < kWgRT / 4; ++st) {
      const float* gr = Gl + (4 * st + q) * gstride + m;
      const float* ar = Al + (4 * st + q) * astride + m;
      float bv[MAXNBW];
#pragma unroll
      for (int b = 0; b < MAXNBW; ++b) {
        const int nb = wave + 8 * b;
        bv[b] = (nb < NB) ? ar[16 * nb] : 0.f;
      }
#pragma unroll
      for (int a = 0; a < MAXMB; ++a) {
        if (a < MBW) {
          const float av = gr[16 * a];
#pragma unroll
          for (int b = 0; b < MAXNBW; ++b) {
            if (wave + 8 * b < NB)
              acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[b], acc[a][b], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();
  }

  // ---- partial -> slab ---------------------------------------------------------------------
  float* slab = job.slab + (size_t)chunk * ((size_t)job.NP * job.KP + job.NP);
#pragma unroll
  for (int a = 0; a < MAXMB; ++a) {
    if (a < MBW) {
#pragma unroll
      for (int b = 0; b < MAXNBW; ++b) {
        const int nb = wave + 8 * b;
        if (nb < NB) {
          float* p = slab + (size_t)((mb_base + a) * 16 + 4 * q) * job.KP + 16 * nb + m;
          const float* v = reinterpret_cast<const float*>(&acc[a][b]);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* pp = p + (size_t)r * job.KP;
            *pp = job.accumulate ? *pp + v[r] : v[r];
          }
        }
      }
    }
  }
  if ((int)threadIdx.x < gw) {
    float* pb = slab + (size_t)job.NP * job.KP + mb_base * 16 + threadIdx.x;
    *pb = job.accumulate ? *pb + bsum : bsum;
  }
}

// ---- slab reduce: parameter gradients in torch layout ([N,K] weight, [N] bias) ---------------
constexpr int kRedMaxEntries = 32;
struct RedEntry {
  const float* slab;
  int nchunks, NP, KP, N, K;
  float* dw;   // [N, K] or nullptr
  float* db;   // [N] or nullptr
  int begin;   // first flat element id of this entry in the launch
};
struct RedArgs {
  int nentries;
  int total;
  RedEntry e[kRedMaxEntries];
};

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const RedArgs a) {
  const int id = blockIdx.x * 256 + threadIdx.x;
  if (id >= a.total) return;
  int j = 0;
  for (int t = 1; t < a.nentries; ++t)
    if (id >= a.e[t].begin) j = t;
  const RedEntry& e = a.e[j];
  const int l = id - e.begin;                   // [0, N*K + N)
  const size_t cs = (size_t)e.NP * e.KP + e.NP;
  size_t off;
  float* out;
  if (l < e.N * e.K) {
    const int n = l / e.K, k = l - n * e.K;
    off = (size_t)n * e.KP + k;
    out = e.dw ? e.dw + l : nullptr;
  } else {
    const int n = l - e.N * e.K;
    off = (size_t)e.NP * e.KP + n;
    out = e.db ? e.db + n : nullptr;
  }
  if (!out) return;
  float s = 0.f;
  int c = 0;
  for (; c + 8 <= e.nchunks; c += 8) {
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = e.slab[(size_t)(c + u) * cs + off];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += t[u];
  }
  for (; c < e.nchunks; ++c) s += e.slab[(size_t)c * cs + off];
  *out = s;
}

}  // namespace b3d
