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
constexpr int kWgMaxJobs = 12;
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

// MAXMB: 16-row output blocks per workgroup; MAXNBW: 16-col input blocks per wave (8 waves);
// SLOTS: row passes per 32-row tile.  Thread t stages ONE fixed 16-byte column chunk
// (t % c4tot) of rows (t / c4tot) + e * RPT, e < SLOTS, so it carries a single source descriptor.
//
// Pipeline per workgroup: tile t is multiplied out of LDS buffer t & 1 while the row chunks of tiles
// t+1 and t+2 are in flight in two register sets and the gather indices of tile t+3 are being
// fetched -- the dependent index -> row chain and the HBM/L2 latency are both off the critical
// path; one workgroup barrier per tile.
template <int MAXMB, int MAXNBW, int SLOTS>
__global__ __launch_bounds__(kThreads, 4) void wgrad_kernel(const WgArgs args) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ WgJob sjob;
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
  const int tile_floats = kWgRT * (gstride + astride);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int m = lane & 15, q = lane >> 4;

  // ---- this thread's column chunk ----------------------------------------------------------
  const int gc4 = gw / 4, ac4 = job.KP / 4, c4tot = gc4 + ac4;
  const int RPT = kThreads / c4tot;              // rows staged per pass (host guarantees >= 1)
  const int my_c4 = threadIdx.x % c4tot, my_r = threadIdx.x / c4tot;
  const bool active = my_r < RPT;
  const float* sp = nullptr;                     // source base + col0 + column
  const int* ip = nullptr;                       // row gather
  int sstride = 0, mode = 0, seg = -1, col = 0;  // mode 0: zero, 1: one 16-byte load, 2: element-wise
  int lds_off, lds_rstride;
  if (my_c4 < gc4) {
    col = mb_base * 16 + my_c4 * 4;
    lds_off = my_r * gstride + my_c4 * 4;
    lds_rstride = gstride;
    if (col < job.g.width) {
      mode = (job.g.aligned && col + 4 <= job.g.width) ? 1 : 2;
      sp = job.g.ptr + job.g.col0 + col; ip = job.g.idx; sstride = job.g.stride;
    }
  } else {
    int c = (my_c4 - gc4) * 4;
    lds_off = kWgRT * gstride + my_r * astride + c;
    lds_rstride = astride;
    for (int sgi = 0; sgi < job.nact; ++sgi) {
      const int w = job.act[sgi].width;
      if (c < w) {
        seg = sgi; col = c;
        mode = (job.act[sgi].aligned && (c & 3) == 0 && c + 4 <= w) ? 1 : 2;
        sp = job.act[sgi].ptr + job.act[sgi].col0 + c; ip = job.act[sgi].idx; sstride = job.act[sgi].stride;
        break;
      }
      c -= w;
    }
  }
  if (!active) mode = 0;

  // element-wise path: unaligned source or a float4 that straddles segments (rare, small jobs)
  auto fetch_slow = [&](long row) -> v4f {
    v4f v = {0.f, 0.f, 0.f, 0.f};
    float* vp = reinterpret_cast<float*>(&v);
    if (seg == -1) {
      const WgSeg& sg = job.g;
      const long r = sg.idx ? sg.idx[row] : row;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (col + e < sg.width) vp[e] = wg_load1(sg, r, col + e);
      return v;
    }
    int sgi = seg, c = col;
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

  const long r0 = (long)chunk * job.rows_per_chunk;
  long r1 = r0 + job.rows_per_chunk;
  if (r1 > job.rows) r1 = job.rows;
  const int T = (r1 > r0) ? (int)((r1 - r0 + kWgRT - 1) / kWgRT) : 0;

  // gather indices of tile `tile` (-1: nothing to load -> zeros)
  auto load_idx = [&](int tile, int* ridx) {
#pragma unroll
    for (int e = 0; e < SLOTS; ++e) {
      const int rl = e * RPT + my_r;
      const long row = r0 + (long)tile * kWgRT + rl;
      int v = -1;
      if (mode != 0 && tile < T && rl < kWgRT && row < r1) v = (mode == 1 && ip) ? ip[row] : (int)row;
      ridx[e] = v;
    }
  };
  auto load_data = [&](const int* ridx, v4f* P) {
#pragma unroll
    for (int e = 0; e < SLOTS; ++e) {
      P[e] = v4f{0.f, 0.f, 0.f, 0.f};
      if (mode == 1 && ridx[e] >= 0) P[e] = *reinterpret_cast<const v4f*>(sp + (long)ridx[e] * sstride);
    }
    if (mode == 2) {
#pragma unroll
      for (int e = 0; e < SLOTS; ++e)
        if (ridx[e] >= 0) P[e] = fetch_slow(ridx[e]);
    }
  };
  auto write_tile = [&](const v4f* P, float* buf) {
    if (active) {
#pragma unroll
      for (int e = 0; e < SLOTS; ++e)
        if (e * RPT + my_r < kWgRT) *reinterpret_cast<v4f*>(buf + lds_off + e * RPT * lds_rstride) = P[e];
    }
  };

  v4f acc[MAXMB][MAXNBW];
#pragma unroll
  for (int a = 0; a < MAXMB; ++a)
#pragma unroll
    for (int b = 0; b < MAXNBW; ++b) acc[a][b] = v4f{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;

  auto multiply = [&](const float* buf) {
    const float* Gl = buf;
    const float* Al = buf + kWgRT * gstride;
    if ((int)threadIdx.x < gw) {
#pragma unroll 8
      for (int rl = 0; rl < kWgRT; ++rl) bsum += Gl[rl * gstride + threadIdx.x];
    }
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
  };

  // ---- software pipeline ----------------------------------------------------------------------
  v4f P0[SLOTS], P1[SLOTS];
  int I0[SLOTS], I1[SLOTS];
  float* buf0 = smem;
  float* buf1 = smem + tile_floats;
  load_idx(0, I0);
  load_idx(1, I1);
  load_data(I0, P0);            // tile 0
  load_idx(2, I0);
  load_data(I1, P1);            // tile 1
  load_idx(3, I1);
  write_tile(P0, buf0);
  load_data(I0, P0);            // tile 2
  load_idx(4, I0);
  __syncthreads();
  // invariant at the top of iteration t (even): buf0 = tile t, P1 = tile t+1, P0 = tile t+2,
  // I1 = indices of tile t+3, I0 = indices of tile t+4
  for (int t = 0; t < T; t += 2) {
    write_tile(P1, buf1);       // tile t+1
    load_data(I1, P1);          // tile t+3
    load_idx(t + 5, I1);
    multiply(buf0);             // tile t
    __syncthreads();
    if (t + 1 >= T) break;
    write_tile(P0, buf0);       // tile t+2
    load_data(I0, P0);          // tile t+4
    load_idx(t + 6, I0);
    multiply(buf1);             // tile t+1
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
  float* dw;   // [N, K] block with leading dimension ld (a column slice of a wider gradient), or nullptr
  int ld;
  float* db;   // [N] or nullptr
  int begin;   // first QUAD (4 consecutive elements) of this entry in the launch
};
struct RedArgs {
  int nentries;
  int total;
  RedEntry e[kRedMaxEntries];
};

// A workgroup reduces kRedQuads consecutive QUADS of output elements (an entry's elements are numbered [0, N K + N)
// and padded to a multiple of 4; entry.begin counts quads); kRedParts threads share one quad: each sums every
// kRedParts-th slab, 8 slabs in flight per thread.  Where the quad is four consecutive floats of the slab -- four columns
// of one weight row (K % 4 == 0) or four bias entries, 16-byte aligned: every layer of both models -- one 16-byte load
// per slab, 1 KB runs per wavefront (64-element runs of 4-byte loads ran this at 0.7 TB/s); otherwise four 4-byte loads
// per slab, still 8 slabs in flight (a plain scalar loop here -- it held the BIAS quads in round 2 -- walks the slabs one
// load latency at a time: 450 slabs / 8 threads x 4 elements x ~0.4 us was the whole 80 us of the PoseGNN launch).  The
// partials meet in LDS and are added in a fixed order.  Bitwise reproducible.
// Geometry by slab count (round 4): few slabs per entry (the camera+LiDAR+radar step: 21 or 41) -> 128 quads x 4 parts (64 x 8 took
// 75 + 17 us for its two launches, 128 x 4 takes 52 + 13; 256 x 2: 91 + 14, 256 x 1: 130 + 15); hundreds of slabs (PoseGNN: ~450) ->
// 64 x 8, twice the slabs in flight per quad (128 x 4 took 29 us there against 23).
template <int kRedQuads, int kRedParts>
static __global__ __launch_bounds__(kRedQuads * kRedParts) void wgrad_reduce_kernel(const RedArgs a) {
  __shared__ v4f part[kRedParts][kRedQuads];
  const int el = threadIdx.x % kRedQuads, sub = threadIdx.x / kRedQuads;
  const int id = blockIdx.x * kRedQuads + el;
  const bool live = id < a.total;
  int j = 0;
  for (int t = 1; t < a.nentries; ++t)
    if (live && id >= a.e[t].begin) j = t;
  const RedEntry& e = a.e[j];
  const int l0 = live ? 4 * (id - e.begin) : 0;  // first element of the quad, in [0, N*K + N)
  const size_t cs = (size_t)e.NP * e.KP + e.NP;
  const int nk = e.N * e.K, tot = nk + e.N;
  const bool in_w = l0 + 3 < nk, in_b = l0 >= nk && l0 + 3 < tot;
  // slab offset of element l of this entry, -1: nothing to sum (past the end, or no output requested)
  auto off_of = [&](int l) -> long {
    if (l >= tot) return -1;
    if (l < nk) { const int n = l / e.K, k = l - n * e.K; return e.dw ? (long)n * e.KP + k : -1; }
    return e.db ? (long)e.NP * e.KP + (l - nk) : -1;
  };
  const bool aligned = (e.KP & 3) == 0 && (cs & 3) == 0 && ((uintptr_t)e.slab & 15) == 0;
  const bool vec = live && aligned && ((in_w && e.dw && (e.K & 3) == 0) || (in_b && e.db && (nk & 3) == 0));
  v4f s = {0.f, 0.f, 0.f, 0.f};
  if (vec) {
    const float* src = e.slab + off_of(l0);
    int c = sub;
    for (; c + 7 * kRedParts < e.nchunks; c += 8 * kRedParts) {
      v4f t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const v4f*>(src + (size_t)(c + u * kRedParts) * cs);
#pragma unroll
      for (int u = 0; u < 8; ++u) s += t[u];
    }
    // the rest (every entry of the camera+LiDAR+radar step: 21 or 41 slabs are fewer than 8 x kRedParts): up to eight loads in flight
    // again -- clamped slab index, the value of a slab past the end replaced by zero -- instead of one dependent round trip per
    // slab (round 4: this loop was the whole 80 us of the launch); same slabs in the same order
    if (c < e.nchunks) {
      v4f t[8];
      const int last = e.nchunks - 1;
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int cc = c + u * kRedParts; t[u] = *reinterpret_cast<const v4f*>(src + (size_t)(cc < last ? cc : last) * cs); }
#pragma unroll
      for (int u = 0; u < 8; ++u) if (c + u * kRedParts < e.nchunks) s += t[u];
    }
  } else if (live) {
    long off[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) off[r] = off_of(l0 + r);
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    int c = sub;
    for (; c + 7 * kRedParts < e.nchunks; c += 8 * kRedParts) {
      float t[8][4];
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) t[u][r] = off[r] >= 0 ? e.slab[(size_t)(c + u * kRedParts) * cs + off[r]] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += t[u][r];
    }
    for (; c < e.nchunks; c += kRedParts)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] += off[r] >= 0 ? e.slab[(size_t)c * cs + off[r]] : 0.f;
    s = v4f{v[0], v[1], v[2], v[3]};
  }
  part[sub][el] = s;
  __syncthreads();
  if (sub == 0 && live) {
    v4f t = part[0][el];
#pragma unroll
    for (int p = 1; p < kRedParts; ++p) t += part[p][el];
    if (vec && in_w && (e.ld & 3) == 0 && ((uintptr_t)e.dw & 15) == 0) {
      const int n = l0 / e.K, k = l0 - n * e.K;
      *reinterpret_cast<v4f*>(e.dw + (size_t)n * e.ld + k) = t;
    } else {
      const float v[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int l = l0 + r;
        if (l >= tot) continue;
        if (l < nk) { const int n = l / e.K, k = l - n * e.K; if (e.dw) e.dw[(size_t)n * e.ld + k] = v[r]; }
        else if (e.db) e.db[l - nk] = v[r];
      }
    }
  }
}

}  // namespace b3d
