// Frame-wise k-NN in feature space + GATConv (reference pose_gnn.py:74-80, clr_att_gnn.py:178-184).
//
// For every distinct timestamp value t: X = x[ts == t]; k-NN graph (k = 20, no self loops) on X in
// D-dimensional feature space; GATConv(D, D, heads = 1, add_self_loops = False) on that graph.
// The reference then evaluates `x[ts == t] == x_t` -- a comparison -- so the result never reaches
// x.  The block is still executed by the reference (and by its CPU baseline), hence these
// kernels; whether the result is written back is the caller's choice.
//
// Third-party semantics (torch_cluster.knn / torch_geometric GATConv are absent from the
// reference tree and un-pinned): Euclidean k nearest excluding self, fewer
// than k when the frame has <= k nodes; h = W x; a = leaky_relu(h_q.att_src + h_c.att_dst, 0.2);
// softmax over the neighbours of c as exp(a - max) / (sum + 1e-16); y_c = sum alpha h_q + bias.
#pragma once
#include "b3d_launch.hpp"

namespace b3d {

constexpr int kKnnMaxK = 32;
constexpr int kKnnLdsCand = 2048;     // candidates per wave whose distances are cached in LDS

struct KnnWs {
  int* rank;      // [N] position of node in (timestamp, id) order
  int* order;     // [N] inverse of rank
  int* fbeg;      // [N] first position of the node's frame in `order`
  int* fend;      // [N] one past the last
  int* nbr;       // [N, kKnnMaxK] neighbour node ids (-1 padded)
  int* cnt;       // [N] number of neighbours
  int* rcnt;      // [3, N] rank counters
  float* h;       // [N, D]
  float* y;       // [N, D] GAT output
  float* wp;      // packed image of lin (L<D,D>)
  bool ranked;
  bool packed;    // wp already holds the image of `lin` (the model's own pack launch wrote it)
};

inline void knn_carve(KnnWs& k, Carver& c, int N, int D) {
  const size_t n = (size_t)(N > 0 ? N : 1);
  k.rank = c.take<int>(n); k.order = c.take<int>(n); k.fbeg = c.take<int>(n); k.fend = c.take<int>(n);
  k.nbr = c.take<int>(n * kKnnMaxK); k.cnt = c.take<int>(n); k.rcnt = c.take<int>(3 * n + 64);
  k.h = c.take<float>(n * D); k.y = c.take<float>(n * D);
  k.wp = c.take<float>(D == 48 ? LayerSeq<L<48, 48>>::TOTAL_FLOATS : LayerSeq<L<96, 96>>::TOTAL_FLOATS);
  k.ranked = false;
  k.packed = false;
}

// Rank by (timestamp, node id).  O(N^2) comparisons spread over N x kRankSlices threads: thread
// (i, s) counts over slice s of the nodes (timestamps staged through LDS as 32-bit offsets from
// ts[0]) and adds its partial counts with integer atomics (exact, order independent).
constexpr int kRankSlices = 16;
static __global__ __launch_bounds__(256) void knn_rank_count_kernel(const int64_t* __restrict__ ts, int N,
                                                             int* __restrict__ cnt /* [3][N] zeroed */) {
  __shared__ int tile[256];
  const int64_t t0 = ts[0];
  auto rel = [&](int64_t v) {
    int64_t d = v - t0;
    d = d > (1 << 30) ? (1 << 30) : d;
    d = d < -(1 << 30) ? -(1 << 30) : d;
    return (int)d;
  };
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int t = (i < N) ? rel(ts[i]) : 0;
  const int per = ((N + kRankSlices - 1) / kRankSlices + 255) / 256 * 256;
  const int jb = blockIdx.y * per;
  const int je = (jb + per < N) ? jb + per : N;
  int less = 0, leq = 0, r = 0;
  for (int j0 = jb; j0 < je; j0 += 256) {
    const int j = j0 + threadIdx.x;
    tile[threadIdx.x] = (j < N) ? rel(ts[j]) : 0x7fffffff;
    __syncthreads();
    const int split = i - j0;                    // tile entries below `split` have a smaller node id
#pragma unroll 16
    for (int jj = 0; jj < 256; ++jj) {
      const int u = tile[jj];
      less += (u < t);
      leq += (u <= t);
      r += (u < t) || (u == t && jj < split);
    }
    __syncthreads();
  }
  if (i < N) {
    atomicAdd(&cnt[i], less);
    atomicAdd(&cnt[N + i], leq);
    atomicAdd(&cnt[2 * N + i], r);
  }
}

static __global__ void knn_rank_finish_kernel(const int* __restrict__ cnt, int N, int* __restrict__ rank,
                                       int* __restrict__ order, int* __restrict__ fbeg, int* __restrict__ fend) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int r = cnt[2 * N + i];
  rank[i] = r;
  order[r] = i;
  fbeg[i] = cnt[i];
  fend[i] = cnt[N + i];
}

// Wavefront minimum of 64-bit keys (hi, lo) with DPP row shifts / row broadcasts (GFX9 wave64
// reduction; the result lands in lane 63 and is broadcast through an SGPR).  Distances are
// non-negative floats, so their bit patterns order like unsigned integers.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void dpp_min_step(unsigned& hi, unsigned& lo) {
  const unsigned th = (unsigned)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)hi, CTRL, ROW_MASK, 0xF, false);
  const unsigned tl = (unsigned)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)lo, CTRL, ROW_MASK, 0xF, false);
  const bool less = (th < hi) || (th == hi && tl < lo);
  hi = less ? th : hi;
  lo = less ? tl : lo;
}
__device__ __forceinline__ void wave_min_key(unsigned& hi, unsigned& lo) {
  dpp_min_step<0x111, 0xF>(hi, lo);      // row_shr:1
  dpp_min_step<0x112, 0xF>(hi, lo);      // row_shr:2
  dpp_min_step<0x114, 0xF>(hi, lo);      // row_shr:4
  dpp_min_step<0x118, 0xF>(hi, lo);      // row_shr:8   -> lane 15 of every row holds the row minimum
  dpp_min_step<0x142, 0xA>(hi, lo);      // row_bcast:15 into rows 1 and 3
  dpp_min_step<0x143, 0xC>(hi, lo);      // row_bcast:31 into rows 2 and 3 -> lane 63 holds the minimum
  hi = (unsigned)__builtin_amdgcn_readlane((int)hi, 63);
  lo = (unsigned)__builtin_amdgcn_readlane((int)lo, 63);
}

__device__ __forceinline__ void wave_argmin(float& d, int& j) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const float od = __shfl_xor(d, off, 64);
    const int oj = __shfl_xor(j, off, 64);
    if (od < d || (od == d && oj < j)) { d = od; j = oj; }
  }
}

// One wavefront per centre, kKnnCentres centres (consecutive in frame order) per workgroup.
// Candidate rows are staged through LDS in tiles of kKnnTileRows, so a candidate row is fetched from
// L2 once per workgroup instead of once per centre; each lane computes the squared distance of its
// candidate to the wave's centre from LDS, the distances of the whole frame stay in a per-wave LDS
// list, and the k nearest are extracted by k rounds of a wavefront arg-min.  A frame longer than the list (kKnnList
// positions) is taken in segments: whenever the next tile would not fit, the k nearest of the positions listed so far are
// extracted and merged with the running k nearest (two sorted lists of <= 32 entries side by side in one wavefront, k
// more arg-min rounds) -- exact, same (distance, position) order, ~n_t / 960 extra extractions.  The footprint (4 waves,
// < 45 KB LDS) is chosen so that these workgroups fit on a CU NEXT to a resident edge-phase
// workgroup (104 KB LDS, 8 waves): the block runs on a side stream underneath the layer it belongs to.
constexpr int kKnnCentres = 4;
constexpr int kKnnTileRows = 64;
constexpr int kKnnList = 1024;        // positions in the per-wave LDS distance list (longer frames: several segments)
template <int D>
__global__ __launch_bounds__(kKnnCentres * 64) void knn_tile_kernel(const float* __restrict__ x, int N, int k,
                                                                    const int* __restrict__ order,
                                                                    const int* __restrict__ fbeg,
                                                                    const int* __restrict__ fend,
                                                                    int* __restrict__ nbr, int* __restrict__ cnt) {
  // padded tile row: 16-byte aligned rows whose 16-lane groups start on distinct banks (100 l mod 64 and 52 l mod 64 are 16 different
  // multiples of 4 for l = 0..15), so a lane reads its candidate row as ds_read_b128 -- with D + 1 floats per row (rounds 1-3) every
  // FMA of the distance had its own 4-byte LDS read and the kernel was bound by LDS instruction issue
  constexpr int TS = D + 4;                               // (D = 48, PoseGNN: 33.6 us against 35.4 with D + 1 floats per row)
  constexpr int TR = kKnnTileRows;                        // candidate rows per tile
  constexpr int NT = kKnnCentres * 64;
  __shared__ __attribute__((aligned(16))) float tile[2][TR * TS];   // double buffered: one barrier per tile
  __shared__ float dist[kKnnCentres][kKnnList];
  __shared__ int wb[kKnnCentres], we[kKnnCentres];
  __shared__ float pickd[kKnnCentres][32];               // the k survivors of a selection (b3d_knn.hpp: extract), one per lane
  __shared__ int pickp[kKnnCentres][32];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int pos = blockIdx.x * kKnnCentres + wave;        // position in frame order
  const bool live = pos < N;
  const int c = live ? order[pos] : 0;
  const int b = live ? fbeg[c] : 0x7fffffff, e = live ? fend[c] : 0;
  const int nt = live ? e - b : 0;
  if (lane == 0) { wb[wave] = b; we[wave] = e; }
  __syncthreads();
  int ub = 0x7fffffff, ue = 0;                            // union of the candidate ranges of the workgroup
#pragma unroll
  for (int w = 0; w < kKnnCentres; ++w) { ub = min(ub, wb[w]); ue = max(ue, we[w]); }
  float xc[D];
#pragma unroll
  for (int d = 0; d < D; d += 4) {
    const v4f v = *reinterpret_cast<const v4f*>(x + (size_t)c * D + d);
    xc[d] = v.x; xc[d + 1] = v.y; xc[d + 2] = v.z; xc[d + 3] = v.w;
  }
  const float INF = __builtin_inff();
  const int kk = (k < nt - 1) ? k : (nt - 1);
  static_assert(kKnnMaxK <= 32, "the merge holds the two sorted lists in the two halves of a wavefront");
  // Segment state of this wavefront: list entry i is frame position segbase + i; lane r holds the r-th nearest found so
  // far (distance, position relative to b), INF where there is none yet.
  int segbase = b;
  float bestd = INF;
  int bestp = 0x7fffffff;
  bool have = false;
  // k nearest of the listed positions [segbase, upto), merged into (bestd, bestp); the list restarts at `upto`.
  auto extract = [&](int upto) __attribute__((always_inline)) {
    const int fill = upto - segbase;
    if (fill <= 0) return;
    // The lane's candidates (list entries lane, lane + 64, ...) move to registers.
    constexpr int SL = kKnnList / 64;
    float cd[SL];
#pragma unroll
    for (int j = 0; j < SL; ++j) cd[j] = (64 * j + lane < fill) ? dist[wave][64 * j + lane] : INF;
    // Selection by threshold (round 3; k rounds of scan + arg-min before: ~2,400 vector instructions per centre, the largest
    // part of this kernel).  (1) The `want`-th smallest distance V by bisection on the float BITS (non-negative floats order
    // like unsigned integers): a step is one compare per list register, the count is ballots + scalar popcounts.  (2)
    // Exactly `want` survivors: everything below V, and of the entries equal to V those with the smallest positions (a
    // second bisection, on the position, only if V is tied beyond what is needed).  (3) Survivors compacted through LDS,
    // one per lane.  (4) A bitonic sort of the <= 32 (distance, position) keys across lanes 0..31 leaves the r-th nearest
    // in lane r -- the same order the rounds produced.
    unsigned cb[SL];
#pragma unroll
    for (int j = 0; j < SL; ++j) cb[j] = __float_as_uint(cd[j]);
    auto count = [&](auto pred) __attribute__((always_inline)) {   // entries of the whole list that satisfy pred(j)
      int c = 0;
#pragma unroll
      for (int j = 0; j < SL; ++j) c += __popcll(__ballot(pred(j)));
      return c;
    };
    const unsigned kInfBits = 0x7f800000u;
    const int navail = count([&](int j) { return cb[j] < kInfBits; });
    const int want = kk < navail ? kk : navail;
    float sd = INF;
    int sp = 0x7fffffff;
    if (want > 0) {
      unsigned mn = 0xffffffffu, mx = 0u;
#pragma unroll
      for (int j = 0; j < SL; ++j) { mn = min(mn, cb[j]); mx = max(mx, cb[j] < kInfBits ? cb[j] : 0u); }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) { mn = min(mn, (unsigned)__shfl_xor((int)mn, off, 64)); mx = max(mx, (unsigned)__shfl_xor((int)mx, off, 64)); }
      unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)mn), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)mx);
      // smallest V with count(<= V) >= want.  (Round 4) The walk stops as soon as a probe has EXACTLY `want` entries at or below it
      // -- any such value separates the survivors, it need not be the want-th distance itself: ~log2(entries x spread) probes
      // instead of one per bit of the range; only a tie across the want-th place runs to the end and into the position bisection.
      bool cut = false;
      unsigned thr = 0u;
      while (lo < hi) {
        const unsigned mid = lo + ((hi - lo) >> 1);
        const int cle = count([&](int j) { return cb[j] <= mid; });
        if (cle == want) { cut = true; thr = mid; break; }
        if (cle > want) hi = mid; else lo = mid + 1;
      }
      const unsigned V = cut ? thr + 1u : lo;                  // survivors: everything below V (+ entries equal to V up to ilim)
      const int need_eq = cut ? 0 : want - count([&](int j) { return cb[j] < V; });      // >= 1 without a clean cut
      int ilim = cut ? -1 : 0x7fffffff;                        // largest list index taken among the entries equal to V
      if (!cut && count([&](int j) { return cb[j] == V; }) > need_eq) {
        int il = 0, ih = kKnnList - 1;
        while (il < ih) {
          const int im = (il + ih) >> 1;
          if (count([&](int j) { return cb[j] == V && 64 * j + lane <= im; }) >= need_eq) ih = im; else il = im + 1;
        }
        ilim = il;
      }
      // compaction: survivor number (entries of earlier registers) + (earlier lanes of this register)
      int base = 0;
#pragma unroll
      for (int j = 0; j < SL; ++j) {
        const bool sel = cb[j] < V || (cb[j] == V && 64 * j + lane <= ilim);
        const unsigned long long m = __ballot(sel);
        if (sel) {
          const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
          pickd[wave][slot] = cd[j];
          pickp[wave][slot] = 64 * j + lane;
        }
        base += __popcll(m);
      }
      float kd = INF;
      int kp = 0x7fffffff;
      if (lane < want) { kd = pickd[wave][lane]; kp = pickp[wave][lane]; }
      // bitonic sort, ascending in (distance, position), over lanes 0..31 (lanes >= want hold +inf keys and stay behind)
#pragma unroll
      for (int ksz = 2; ksz <= 32; ksz <<= 1) {
#pragma unroll
        for (int st = ksz >> 1; st >= 1; st >>= 1) {
          const float od = __shfl_xor(kd, st, 64);
          const int op = __shfl_xor(kp, st, 64);
          const bool up = (lane & ksz) == 0;                   // this block sorts ascending
          const bool lower = (lane & st) == 0;                 // this lane keeps the smaller key of the pair (if ascending)
          const bool o_less = (od < kd) || (od == kd && op < kp);
          const bool take = (up == lower) ? o_less : !o_less && !(od == kd && op == kp);
          if (take) { kd = od; kp = op; }
        }
      }
      if (lane < want) { sd = kd; sp = kp + (segbase - b); }
    }
    if (!have) {
      bestd = sd; bestp = sp; have = true;
    } else {
      // lanes 0..31: the running list, lanes 32..63: this segment's list; k rounds take the smallest (distance, position)
      // (the shuffles run with every lane active: ds_bpermute returns 0 for a source lane that is masked off)
      const float sdx = __shfl(sd, lane & 31, 64);
      const int spx = __shfl(sp, lane & 31, 64);
      float md = lane < 32 ? bestd : sdx;
      int mp = lane < 32 ? bestp : spx;
      float nd = INF;
      int np = 0x7fffffff;
      for (int r = 0; r < kk; ++r) {
        unsigned hi = __float_as_uint(md), lo = (md < INF) ? (unsigned)mp : 0x7fffffffu;
        const unsigned myhi = hi, mylo = lo;
        wave_min_key(hi, lo);
        if (lane == r && lo != 0x7fffffffu) { nd = __uint_as_float(hi); np = (int)lo; }
        if (myhi == hi && mylo == lo) md = INF;              // positions are unique: exactly one lane retires its entry
      }
      bestd = nd; bestp = np;
    }
    segbase = upto;
  };
  constexpr int PER = TR * (D / 4) / NT;                  // v4f staged per thread per tile
  static_assert(PER * NT == TR * (D / 4), "tile must divide over the workgroup");
  // Software pipeline, two tiles deep: while tile i is evaluated from LDS, the rows of tiles i+1 and i+2 are in flight to two
  // register sets and the gather indices of tile i+3 are being fetched (a tile step is ~0.3 us of LDS reads and FMAs, the
  // index -> row chain two dependent L2 round trips: with one tile in flight the launch was a latency chain).  Loads are
  // unconditional (positions past the end of the union range read a clamped row nobody evaluates): a test around a load
  // makes hipcc drain every load in flight behind it.
  int ridx[PER];
  v4f va[PER], vb[PER];
  auto load_idx = [&](int t0) {
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int r = (threadIdx.x + j * NT) / (D / 4);
      ridx[j] = order[min(t0 + r, N - 1)];
    }
  };
  auto load_rows = [&](v4f (&v)[PER]) {
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int c4 = (threadIdx.x + j * NT) % (D / 4);
      v[j] = *reinterpret_cast<const v4f*>(x + (size_t)ridx[j] * D + 4 * c4);
    }
  };
  auto tile_step = [&](int t0, int cur, v4f (&v)[PER]) __attribute__((always_inline)) {      // rows of tile t0 (in v) -> LDS; v <- rows of tile t0 + 2 TR; evaluate
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int i = threadIdx.x + j * NT;
      const int r = i / (D / 4), c4 = i - r * (D / 4);
      *reinterpret_cast<v4f*>(tile[cur] + r * TS + 4 * c4) = v[j];
    }
    __syncthreads();
    load_rows(v);                                         // tile t0 + 2 TR (indices fetched one step ago)
    load_idx(t0 + 3 * TR);
    if (live && t0 > segbase && t0 + TR - segbase > kKnnList) extract(t0 < e ? t0 : e);   // the list cannot take this tile
#pragma unroll
    for (int u = 0; u < TR / 64; ++u) {
      const int p = t0 + 64 * u + lane;                   // this lane's candidate position
      if (live && p >= b && p < e) {
        const float* row = tile[cur] + (64 * u + lane) * TS;
        float s2 = 0.f;
#pragma unroll
        for (int d = 0; d < D; d += 4) {                    // (the same fmaf chain, feature by feature)
          const v4f r4 = *reinterpret_cast<const v4f*>(row + d);
          float a = r4.x - xc[d]; s2 = fmaf(a, a, s2);
          a = r4.y - xc[d + 1]; s2 = fmaf(a, a, s2);
          a = r4.z - xc[d + 2]; s2 = fmaf(a, a, s2);
          a = r4.w - xc[d + 3]; s2 = fmaf(a, a, s2);
        }
        dist[wave][p - segbase] = (p == pos) ? INF : s2;
      }
    }
  };
  load_idx(ub);
  load_rows(va);
  load_idx(ub + TR);
  load_rows(vb);
  load_idx(ub + 2 * TR);
  for (int t0 = ub; t0 < ue; t0 += 2 * TR) {
    tile_step(t0, 0, va);
    if (t0 + TR >= ue) break;
    tile_step(t0 + TR, 1, vb);
  }
  if (!live) return;
  extract(e);
  if (lane < kk) nbr[(size_t)c * kKnnMaxK + lane] = order[b + bestp];
  if (lane == 0) cnt[c] = kk > 0 ? kk : 0;
}

// one wavefront per destination: per-destination segmented softmax with wavefront shuffles,
// attention-weighted sum of neighbour rows
template <int D>
__global__ __launch_bounds__(256) void gat_aggregate_kernel(const float* __restrict__ h, int hs, int N,
                                                            const int* __restrict__ nbr, const int* __restrict__ cnt,
                                                            const float* __restrict__ att_src, const float* __restrict__ att_dst,
                                                            const float* __restrict__ bias, float* __restrict__ y) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + wave;
  if (c >= N) return;
  const int kk = cnt[c];
  int q = -1;
  float a = -__builtin_inff();
  if (lane < kk) {
    // attention logit of edge (q -> c): leaky_relu(<h[q], att_src> + <h[c], att_dst>), GATConv heads = 1
    q = nbr[(size_t)c * kKnnMaxK + lane];
    float ss = 0.f, sd = 0.f;
#pragma unroll
    for (int d = 0; d < D; d += 4) {
      const v4f vq = *reinterpret_cast<const v4f*>(h + (size_t)q * hs + d);
      const v4f vc = *reinterpret_cast<const v4f*>(h + (size_t)c * hs + d);
      ss = fmaf(vq.x, att_src[d], ss); ss = fmaf(vq.y, att_src[d + 1], ss); ss = fmaf(vq.z, att_src[d + 2], ss); ss = fmaf(vq.w, att_src[d + 3], ss);
      sd = fmaf(vc.x, att_dst[d], sd); sd = fmaf(vc.y, att_dst[d + 1], sd); sd = fmaf(vc.z, att_dst[d + 2], sd); sd = fmaf(vc.w, att_dst[d + 3], sd);
    }
    const float z = ss + sd;
    a = z > 0.f ? z : 0.2f * z;
  }
  float m = a;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  float ex = (lane < kk) ? __expf(a - m) : 0.f;
  float den = ex;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) den += __shfl_xor(den, off, 64);
  const float alpha = ex / (den + 1e-16f);
  constexpr int PER = (D + 63) / 64;
  float acc[PER];
#pragma unroll
  for (int p = 0; p < PER; ++p) acc[p] = 0.f;
  for (int j = 0; j < kk; ++j) {
    const float aj = __shfl(alpha, j, 64);
    const int qj = __shfl(q, j, 64);
#pragma unroll
    for (int p = 0; p < PER; ++p) {
      const int d = lane + 64 * p;
      if (d < D) acc[p] = fmaf(aj, h[(size_t)qj * hs + d], acc[p]);
    }
  }
#pragma unroll
  for (int p = 0; p < PER; ++p) {
    const int d = lane + 64 * p;
    if (d < D) y[(size_t)c * D + d] = acc[p] + bias[d];
  }
}

// x [N, D]; result left in ws.y / ws.nbr / ws.cnt
template <int D>
inline int knn_gat_block(KnnWs& ws, const float* x, const int64_t* ts, int N, const b3d_gat& gat, int k,
                         hipStream_t stream, const float* h_pre = nullptr, int h_stride = 0) {
  // h_pre: GATConv.lin(x) already computed by the caller ([N, h_stride] rows, D valid columns)
  B3D_REQUIRE(gat.lin && gat.att_src && gat.att_dst && gat.bias, "knn_conv parameters are null");
  B3D_REQUIRE(k >= 1 && k <= kKnnMaxK, "k-NN k=%d outside [1,%d]", k, kKnnMaxK);
  if (N <= 0) return B3D_OK;
  if (!ws.ranked) {
    B3D_HIP_CHECK(hipMemsetAsync(ws.rcnt, 0, (3 * (size_t)N + 64) * sizeof(int), stream));
    hipLaunchKernelGGL(knn_rank_count_kernel, dim3((N + 255) / 256, kRankSlices), dim3(256), 0, stream, ts, N, ws.rcnt);
    B3D_TRY(launch_check("knn_rank_count_kernel"));
    hipLaunchKernelGGL(knn_rank_finish_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, ws.rcnt, N, ws.rank, ws.order, ws.fbeg, ws.fend);
    B3D_TRY(launch_check("knn_rank_finish_kernel"));
    if (!ws.packed) {
      using S = LayerSeq<L<D, D>>;
      PackDesc d = pack_desc<S>(0, ws.wp, gat.lin, nullptr, D, D, false);
      B3D_TRY(pack_images(&d, 1, stream));
      ws.packed = true;
    }
    ws.ranked = true;
  }
  {
    ProfScope ps(B3D_K_KNN, stream);
    hipLaunchKernelGGL(knn_tile_kernel<D>, dim3((N + kKnnCentres - 1) / kKnnCentres), dim3(kKnnCentres * 64), 0, stream,
                       x, N, k, ws.order, ws.fbeg, ws.fend, ws.nbr, ws.cnt);
  }
  B3D_TRY(launch_check("knn_tile_kernel"));
  const float* h = h_pre ? h_pre : ws.h;
  const int hs = h_pre ? h_stride : D;
  if (!h_pre) {
    using S = LayerSeq<L<D, D>>;
    ChainFwdArgs<LoadAligned<D / 16>, StoreAligned<D / 16>> a;
    memset(&a, 0, sizeof(a));
    a.rows = N;
    a.in = LoadAligned<D / 16>{x, nullptr, D, 0};
    a.out = StoreAligned<D / 16>{ws.h, nullptr, D, 0};
    a.wpack = ws.wp;
    B3D_TRY(launch_rows<kNWNode>(chain_fwd_kernel<S, 0u, LoadAligned<D / 16>, StoreAligned<D / 16>, kNWNode>, "gat_linear", a, N, stream, B3D_K_KNN, chain_lds<S>()));
  }
  {
    ProfScope ps(B3D_K_KNN, stream);
    hipLaunchKernelGGL(gat_aggregate_kernel<D>, dim3((N + 3) / 4), dim3(256), 0, stream, h, hs, N, ws.nbr, ws.cnt, gat.att_src, gat.att_dst, gat.bias, ws.y);
  }
  return launch_check("gat_aggregate_kernel");
}

}  // namespace b3d
