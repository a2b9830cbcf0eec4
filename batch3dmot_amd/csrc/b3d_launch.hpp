// Host-side launch helpers shared by the model orchestration files.
#pragma once
#include <stdlib.h>
#include "b3d_common.hpp"
#include "b3d_chain.hpp"
#include "b3d_mp.hpp"
#include "b3d_node.hpp"
#include "b3d_pack.hpp"
#include "b3d_wgrad.hpp"

namespace b3d {

template <class K>
inline int set_lds(K kernel, int bytes) {
  if (bytes <= 64 * 1024) return B3D_OK;
  return set_lds_cached(reinterpret_cast<const void*>(kernel), bytes);
}

// Row-tiled MLP kernels.  NW = wavefronts per workgroup (each owns 16 rows).  Edge-sized inputs
// use 8; node-sized chain kernels (encoders) use 1 so the launch still covers ~200 CUs.
template <int NW, class Kern, class Args>
inline int launch_rows(Kern kernel, const char* name, const Args& a, long rows, hipStream_t stream,
                       int family = B3D_K_OTHER, int lds_bytes = kLdsBytes) {
  if (rows <= 0) return B3D_OK;
  B3D_TRY(set_lds(kernel, lds_bytes));
  ProfScope ps(family, stream);
  hipLaunchKernelGGL(kernel, dim3(grid_for_tiles(rows, NW * 16)), dim3(NW * 64), lds_bytes, stream, a);
  return launch_check(name);
}
// LDS of the chain / wide-linear kernels: two ring slots of the sequence's largest chunk (a 48x48 layer
// needs 20 KB, not the 104 KB of the message-passing kernels -- it must not evict them from a CU).
template <class Seq>
constexpr int chain_lds() { return 2 * Seq::SLOT * 4; }     // ring of the largest chunk; a resident image (WStreamG) is never larger
constexpr int kNWEdge = 8, kNWNode = 1;

// Node phase of a message-passing layer: 4 wavefronts per 16-row tile (b3d_node.hpp).
template <class D, int NW = kNodeWaves, class Kern, class Args>
inline int launch_node_split(Kern kernel, const char* name, const Args& a, long rows, hipStream_t stream, int family) {
  if (rows <= 0) return B3D_OK;
  B3D_TRY(set_lds(kernel, NodeSplit<D>::LDS_BYTES));
  ProfScope ps(family, stream);
  hipLaunchKernelGGL(kernel, dim3((unsigned)((rows + 15) / 16)), dim3(NW * 64), NodeSplit<D>::LDS_BYTES, stream, a);
  return launch_check(name);
}

inline WgSeg seg(const float* p, const int* idx, int stride, int col0, int width) {
  WgSeg s;
  s.ptr = p; s.idx = idx; s.stride = stride; s.col0 = col0; s.width = width;
  s.aligned = ((stride & 3) == 0 && (col0 & 3) == 0 && ((uintptr_t)p & 15) == 0) ? 1 : 0;
  return s;
}

// Chunk geometry of a weight-gradient job.  A workgroup's time is dominated by per-tile latency,
// not by the matrix size, so the chunk length depends first on the row count: ~128 chunks per
// matrix, between 1 and 8 tiles of 32 rows each.  Every chunk writes a whole partial matrix (NP x KP floats), so a
// wide matrix additionally gets at least as many rows per chunk as make that slab no larger than the rows it
// reads (NP KP / (NP + KP)): 128 for the 192 x 256 fc heads of the camera+LiDAR+radar model, where 32-row chunks
// spent more on slabs than on inputs (0.70 -> 0.49 ms of LDS-staged weight gradients per step).
inline int wg_rows_per_chunk(long rows, int NP, int KP) {
  long rpc = ((rows + 127) / 128 + kWgRT - 1) / kWgRT * kWgRT;
  if (rpc < kWgRT) rpc = kWgRT;
  if (NP > 0 && KP > 0) {
    const long bal = ((long)NP * KP / (NP + KP) + kWgRT - 1) / kWgRT * kWgRT;
    if (bal > rpc) rpc = bal;
  }
  if (rpc > 8 * kWgRT) rpc = 8 * kWgRT;
  return (int)rpc;
}
inline int wg_nchunks(long rows, int NP, int KP, long /*launch_weight*/) {
  if (rows <= 0) return 1;
  const int rpc = wg_rows_per_chunk(rows, NP, KP);
  return (int)((rows + rpc - 1) / rpc);
}
inline size_t wg_slab_floats(int nchunks, int NP, int KP) { return (size_t)nchunks * ((size_t)NP * KP + NP); }

template <int MAXMB, int MAXNBW>
inline int launch_wgrad(WgArgs& a, hipStream_t stream, int family = B3D_K_WGRAD_OTHER) {
  // row passes per 32-row tile: a thread stages one 16-byte column chunk of kThreads / c4tot rows
  constexpr int kMaxC4 = (16 * MAXMB + 128 * MAXNBW) / 4;
  constexpr int SLOTS = (kWgRT + (kThreads / kMaxC4) - 1) / (kThreads / kMaxC4);
  static_assert(kMaxC4 <= kThreads, "tile row wider than the workgroup");
  if (a.njobs == 0) return B3D_OK;
  int wgs = 0, maxkp = 0;
  for (int j = 0; j < a.njobs; ++j) {
    WgJob& job = a.jobs[j];
    if (job.KP / 16 > 8 * MAXNBW)
      return fail(B3D_ERR_ARG, "wgrad job %d (%dx%d) exceeds kernel configuration <%d,%d>", j, job.NP, job.KP, MAXMB, MAXNBW);
    job.mgroups = (job.NP / 16 + MAXMB - 1) / MAXMB;
    job.wg_begin = wgs;
    wgs += job.nchunks * job.mgroups;
    if (job.KP > maxkp) maxkp = job.KP;
  }
  const int lds = 2 * kWgRT * ((MAXMB * 16 + 4) + (maxkp + 4)) * 4;
  auto kern = wgrad_kernel<MAXMB, MAXNBW, SLOTS>;
  B3D_TRY(set_lds(kern, lds));
  ProfScope ps(family, stream);
  hipLaunchKernelGGL(kern, dim3(wgs), dim3(kThreads), lds, stream, a);
  return launch_check("wgrad_kernel");
}

inline int launch_reduce(RedArgs& a, hipStream_t stream) {
  int total = 0;                                 // in quads of elements
  for (int i = 0; i < a.nentries; ++i) {
    a.e[i].begin = total;
    total += (a.e[i].N * a.e[i].K + a.e[i].N + 3) / 4;
  }
  a.total = total;
  if (total == 0) return B3D_OK;
  // slabs per output quad, averaged over the launch (a launch is mostly made of its largest matrices)
  double weighted = 0.0;
  for (int i = 0; i < a.nentries; ++i) weighted += (double)a.e[i].nchunks * ((a.e[i].N * a.e[i].K + a.e[i].N + 3) / 4);
  if (weighted / total > 64.0) hipLaunchKernelGGL((wgrad_reduce_kernel<64, 8>), dim3((total + 63) / 64), dim3(512), 0, stream, a);
  else hipLaunchKernelGGL((wgrad_reduce_kernel<128, 4>), dim3((total + 127) / 128), dim3(512), 0, stream, a);
  return launch_check("wgrad_reduce_kernel");
}

// One Linear layer's weight-gradient bookkeeping: slab + job template.
struct LinSlab {
  float* slab;
  int nchunks, NP, KP, N, K;
  bool used;     // already written during this backward -> accumulate
};

inline WgJob make_job(LinSlab& ls, long rows, const WgSeg& g) {
  WgJob j;
  memset(&j, 0, sizeof(j));
  j.g = g;
  j.nact = 0;
  j.NP = ls.NP; j.KP = ls.KP;
  j.rows = (int)rows;
  j.nchunks = ls.nchunks;
  j.rows_per_chunk = wg_rows_per_chunk(rows, ls.NP, ls.KP);
  j.slab = ls.slab;
  j.accumulate = ls.used ? 1 : 0;
  ls.used = true;
  return j;
}
inline void add_act(WgJob& j, const WgSeg& s) { j.act[j.nact++] = s; }

inline RedEntry red_entry(const LinSlab& ls, float* dw, float* db) {
  RedEntry e;
  e.slab = ls.slab; e.nchunks = ls.nchunks; e.NP = ls.NP; e.KP = ls.KP; e.N = ls.N; e.K = ls.K;
  e.dw = dw; e.db = db; e.begin = 0; e.ld = ls.K;
  return e;
}

}  // namespace b3d
