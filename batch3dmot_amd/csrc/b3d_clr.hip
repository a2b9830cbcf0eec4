// GNN (camera + LiDAR + radar) forward / backward -- reference batch_3dmot/models/clr_att_gnn.py:16-188
// -- as a sequence of gfx950 kernel launches on one stream.  All scratch lives in the caller's
// workspace.  The frozen encoders are adjacent (their outputs are inputs here, see include/b3d.h).
#include "b3d_launch.hpp"
#include "b3d_knn.hpp"
#include "b3d_wstream.hpp"
#include "b3d_wstream2.hpp"
#ifndef B3D_NW_HEAD
#define B3D_NW_HEAD 4
#endif
#include "b3d_wgemm.hpp"
#include <algorithm>
#include <vector>
#include "b3d_hoist.hpp"
#include "b3d_att.hpp"
#include "b3d_edge2.hpp"

namespace b3d {
namespace clr {
// several small buffers zeroed by one launch (one memset node each costs ~4 us of a captured step)
constexpr int kZeroMax = 24;
struct ZeroArgs { float* p[kZeroMax]; int n[kZeroMax]; int count; };
__global__ __launch_bounds__(256) void zero_many_kernel(const ZeroArgs a) {
  const int k = blockIdx.y;
  if (k >= a.count) return;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < a.n[k]; i += gridDim.x * 256) a.p[k][i] = 0.f;
}
inline void zero_add(ZeroArgs& z, float* p, size_t n) { if (p && n && z.count < kZeroMax) { z.p[z.count] = p; z.n[z.count] = (int)n; ++z.count; } }
inline int zero_launch(ZeroArgs& z, hipStream_t stream) {
  if (z.count == 0) return B3D_OK;
  hipLaunchKernelGGL(zero_many_kernel, dim3(16, z.count), dim3(256), 0, stream, z);
  z.count = 0;
  return launch_check("zero_many_kernel");
}

// B3D_PAST_RUNS=0: one `past` row per edge whatever the edge order (A/B switch of the in-wave per-destination sums, b3d_edge2.hpp)
static bool past_runs_enabled() {
  static const bool on = []() { const char* ev = getenv("B3D_PAST_RUNS"); return !(ev && atoi(ev) == 0); }();
  return on;
}
using D = DimsC;
using DB = DimsCB;                       // hoisted kernels: bf16x3 images for the edge stacks
using HC = Hoist<DB>;
// The model runs ONE plan: first layers hoisted to per-node tables (b3d_hoist.hpp, b3d_att.hpp), edge stacks on the
// fragment-streamed kernels (b3d_edge2.hpp), weight gradients on the cooperative kernel (b3d_wgemm.hpp) with the
// per-wavefront LDS-DMA kernel (b3d_wstream2.hpp) for the shapes it has no tile for.  The unsplit kernels of b3d_mp.hpp
// serve the single-layer operator at the end of this file.
using ES = es::EdgeSeqs<DB>;
// fragment-streamed edge kernels: one workgroup of es::kWaves wavefronts per es::kTileRows-row tile
template <class Kern, class Args>
static int launch_es(Kern kernel, const char* name, const Args& a, long rows, hipStream_t stream, int family, int lds_bytes) {
  if (rows <= 0) return B3D_OK;
  B3D_TRY(set_lds(kernel, lds_bytes));
  ProfScope ps(family, stream);
  hipLaunchKernelGGL(kernel, dim3(grid_for_tiles(rows, es::kTileRows)), dim3(es::kWaves * 64), lds_bytes, stream, a);
  return launch_check(name);
}
// rows of every per-edge workspace buffer: the fragment-streamed kernels store whole tiles (64 or 128 rows)
static size_t edge_rows(int E) { return ((size_t)(E > 0 ? E : 1) + es::kTileRows - 1) / es::kTileRows * es::kTileRows; }
constexpr int XS = 288;                                                  // x_sens / s width (96 + 128 + 64)
using SeqEE = LayerSeq<L<16, 16>, L<16, 32>, L<32, 64>>;                 // 4-16-32-64          :35-41
using SeqNE = LayerSeq<L<32, 48>, L<48, 96>>;                            // 19-48-96            :43-47
using SeqCls = LayerSeq<L<64, 32>, L<32, 16>, L<16, 16>, L<16, 16>>;     // 64-32-16-8-1        :49-58
using SeqFL = LayerSeq<L<256, 192>, L<192, 128>>;                        // 256-192-128         :60-64
using SeqFR = LayerSeq<L<256, 192>, L<192, 128>, L<128, 64>>;            // 256-192-128-64      :66-72
template <int DD> using SeqAff = LayerSeq<L<DD, DD>, L<DD, DD>>;         // out_proj(v_proj(x)) :77-79,148-155
// att_edge_encoder :81-91 (640-512-384-256-128-64); .0 runs hoisted (b3d_att.hpp)
using SeqAT1 = LayerSeq<L<512, 384, 1>>;     // 384 / 512 inputs: bf16x6 too (one wide layer per kernel: the operand pieces fit)
using SeqAT2 = LayerSeq<L<384, 256, 1>>;
using SeqAT3 = LayerSeq<L<256, 128>>;
using SeqAT4 = LayerSeq<L<128, 64>>;
// att_edge_encoder.0 with its node columns hoisted (b3d_att.hpp)
using SeqAttU = LayerSeq<LF<96, 1024>, LF<96, 1024>, LF<96, 1024>>;       // U = (W0[:, 0:288] s + b0 | W0[:, 288:576] s), three K-slices
template <class... Ls> struct Rep16 { using type = LayerSeq<Ls..., Ls..., Ls..., Ls..., Ls..., Ls..., Ls..., Ls..., Ls..., Ls..., Ls..., Ls..., Ls..., Ls..., Ls..., Ls...>; };
// d s = W0[:, 0:288]^T dU_i + W0[:, 288:576]^T dU_j as K-slices.  Round 6: EIGHT slices of 128 (B3D_ATT_DS_SLICE) instead of sixteen of 64 --
// a 64-wide slice is 128 + 128 + 32 rows in three part-filled ring chunks (48 chunk barriers per launch for 1.3 MB); a 128-wide one is
// three full 96-row chunks (24 barriers), six blocks of a chunk busy with 32-MFMA chains.
#ifndef B3D_PROJ0_WAVES
#define B3D_PROJ0_WAVES 4     // wavefronts per 16-row tile of node_proj0_split_kernel; 8 (one block of a 128-row chunk per wavefront instead of two) measured 29.8 vs 25.9 us: profiles/r06_experiments.txt
#endif
#ifndef B3D_ATT_DS_SLICE
#define B3D_ATT_DS_SLICE 128
#endif
template <class... Ls> struct Rep8 { using type = LayerSeq<Ls..., Ls..., Ls..., Ls..., Ls..., Ls..., Ls..., Ls...>; };
using SeqAttDs = std::conditional_t<B3D_ATT_DS_SLICE == 128, Rep8<LF<128, 288>>::type, Rep16<LF<64, 288>>::type>;
using SeqAT0eT = LayerSeq<L<512, 64, 1>>;                                    // d e0 = W0[:, 576:640]^T d A0
// transposed (data gradient)
using SeqClsT = LayerSeq<L<16, 16>, L<16, 16>, L<16, 32>, L<32, 64>>;
using SeqEET = LayerSeq<L<64, 32>, L<32, 16>>;
using SeqNET = LayerSeq<L<96, 48>>;
using SeqFLT = LayerSeq<L<128, 192>>;
using SeqFRT = LayerSeq<L<64, 128>, L<128, 192>>;
template <int DD> using SeqAffT = LayerSeq<L<DD, DD>, L<DD, DD>>;
using SeqAT4T = LayerSeq<L<64, 128>>;
using SeqAT3T = LayerSeq<L<128, 256>>;
using SeqAT2T = LayerSeq<L<256, 384>>;
using SeqAT1T = LayerSeq<L<384, 512, 1>>;

enum { EE0, EE1, EE2, NE0, NE1, C0, C1, C2, C3, FL0, FL1, FR0, FR1, FR2,
       AVC, AOC, AVL, AOL, AVR, AOR, AT0, AT1, AT2, AT3, AT4,
       EU0, EU1, EU2, PA0, PA1, FU0, FU1, CF0, CF1, CF2, LIN_COUNT };
struct LinDim { int N, K; };
static const LinDim kDims[LIN_COUNT] = {
    {16, 4}, {32, 16}, {64, 32}, {48, 19}, {96, 48}, {32, 64}, {16, 32}, {8, 16}, {1, 8},
    {192, 256}, {128, 192}, {192, 256}, {128, 192}, {64, 128},
    {96, 96}, {96, 96}, {128, 128}, {128, 128}, {64, 64}, {64, 64},
    {512, 640}, {384, 512}, {256, 384}, {128, 256}, {64, 128},
    {256, 320}, {128, 256}, {64, 128}, {192, 256}, {128, 192}, {192, 256}, {128, 192}, {192, 256}, {128, 192}, {96, 128}};
// rows the layer is applied to: 0 = edges, 1 = nodes, 2 = lidar rows, 3 = radar rows
static const int kRowKind[LIN_COUNT] = {0, 0, 0, 1, 1, 0, 0, 0, 0, 2, 2, 3, 3, 3, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0,
                                        0, 0, 0, 0, 0, 0, 0, 1, 1, 1};
// Rows per weight-gradient task (measured on the benchmark batch: 512 / 1,024 / 64 / 64 in round 2; larger tasks = fewer,
// smaller slab sets -- 175 instead of 340 MB per step -- at the same load balance)
// fc_lidar_encoder / fc_radar_encoder (256-192-128(-64) on the 2,100 / 750 rows that carry the modality): ~0.5 MB of bf16x3 weight
// images per workgroup.  With one wavefront per workgroup (kNWNode: right for the 19-48-96 node encoder) that stream passes through
// ONE loader wavefront -- ~25 GB/s, 20 us of a 25 us launch; four wavefronts load it four times faster for four 16-row tiles.
constexpr int kNWHead = B3D_NW_HEAD;
#ifndef B3D_RPT_MP
#define B3D_RPT_MP 768
#endif
#ifndef B3D_RPT_ATT
#define B3D_RPT_ATT 1536
#endif
constexpr int kStreamRowsPerTask = B3D_RPT_MP;   // message-passing stacks (x up to 6 layer variants)
constexpr int kStreamNodeRowsPerTask = 128;      // hoisted first layers: node columns contract over N rows x depth layers
constexpr int kStreamNodeRowsPerTaskAtt = 512;   // att_edge_encoder.0's node columns: one variant, a [512, 288] partial per task
constexpr int kStreamRowsPerTaskAtt = B3D_RPT_ATT;   // att_edge_encoder (one variant)

// column blocks of the hoisted first-layer gradients: (linear, first column, width, rows contracted over edges?)
enum { VL_EU0XI, VL_EU0XJ, VL_EU0E, VL_FU0X, VL_FU0E, VL_FU0X0, VL_PA0X, VL_PA0E, VL_PA0X0, VL_AT0I, VL_AT0J, VL_AT0E, VL_COUNT };
struct VlDesc { int lin, col0, width; bool on_edges, bias; };
static const VlDesc kVl[VL_COUNT] = {
    {EU0, 0, 96, false, false}, {EU0, 96, 96, false, false}, {EU0, 192, 128, true, true},
    {FU0, 0, 96, false, false}, {FU0, 96, 64, true, true}, {FU0, 160, 96, false, false},
    {PA0, 0, 96, false, false}, {PA0, 96, 64, true, true}, {PA0, 160, 96, false, false},
    {AT0, 0, 288, false, false}, {AT0, 288, 288, false, false}, {AT0, 576, 64, true, true}};
struct Ws {
  // forward images
  float *wp_ee, *wp_ne, *wp_cls, *wp_fl, *wp_fr, *wp_aff[3], *wp_at[5], *wp_nfwd;
  // backward images
  // hoisted first layers
  float *wp_proj0, *wp_nfwd_h, *wp_gproj, *wp_nbwd_h;
  float *wp_efwd2, *wp_ebwd2, *wp_ebwd_nm2;       // fragment-stream images (b3d_estream.hpp)
  float *wp_efwd_nm2;                             // ... of the last layer's edge_update alone (B3D_FLAG_SKIP_DEAD_LAST_MESSAGES)
  float *T, *T0, *dT, *gx;
  float *wp_attU, *wp_att0, *wp_attDs, *wp_at0eT, *U, *dU, *ds, *de0;     // att_edge_encoder.0 hoisted
  float *wp_clsT, *wp_eeT, *wp_neT, *wp_flT, *wp_frT, *wp_affT[3], *wp_atT[5], *wp_nbwd;
  // activations
  float *ea_pad, *ee_a1, *ee_a2, *pose_pad, *ne_a1, *xsens, *fl_a1, *fr_a1, *fr_a2, *aff_v[3], *s;
  float *A[4], *att;                 // att_edge_encoder hidden [E,512/384/256/128], output [E,64]
  unsigned* amask[4];                // training: ReLU masks of A[0..3], [E, 16] words each (b3d_dev.hpp); nullptr in inference
  float *x[16], *e[16];
  float *sH1[16], *sH2[16], *sF1[16], *sP1[16], *M[16], *nH1[16], *nH2[16];
  float* rmask[16];       // ReLU masks of sH1 | sH2 | sF1 | sP1 (b3d_estream.hpp: 128 bytes per edge and layer)
  float *fut, *past, *c_a1, *c_a2, *c_a3, *prob;
  // backward scratch
  float *de[2], *da_acc, *gdst, *gsrc, *dx0_acc;
  float *dM, *GdH1, *GdH2, *Gde, *GdF1, *GdP1, *Gdx, *GnH2, *GnH1;     // per layer, uniform stride
  float *gc_top, *gc3, *gc2, *gc1, *ge_top, *ge2, *ge1, *gn_top, *gn1;
  float *dA[4];                      // att backward: [E,128/256/384/512]
  float *gaff_top[3], *gaff_v[3], *dxs;   // affine backward; dxs [N,288] gradient of x_sens columns
  float *gfl_top, *gfl1, *gfr_top, *gfr2, *gfr1;
  float* zrow;
  int* iota;
  WsJob* ws_table;
  int* ws_task_job;
  LinSlab lin[LIN_COUNT];
  LinSlab vlin[VL_COUNT];          // hoisted first layers: column blocks of edge_update.0 / create_*_msgs.0 with slabs of their own
  KnnWs knn;
  size_t bytes;
  bool ok;
};
constexpr int kTableCap = 320, kTaskCap = 32768;

// ---- weight-gradient job builder shared by the model's backward and the standalone layer's ------------------------------------
// The cooperative bf16x6 kernel (b3d_wgemm.hpp) takes every block it has a shape for; the per-wavefront streaming kernel
// (b3d_wstream2.hpp) keeps the rest.  Both share the device tables, split in halves.
struct Col { const float* p; const int* idx; long vstride; int stride; int col0; int width; };   // activation columns
// EXPERIMENT (round 6, -DB3D_WGM_HYBRID=1 or B3D_WGM_HYBRID=1 in the environment; default OFF): hybrid task sizes of the cooperative
// kernel -- the first ~3/4 of a job's rows in tasks of 4 x rows_per_task, the rest in tasks of rows_per_task, so that a matrix gets
// half as many slabs (31,078 rows at 768 per task: 7 + 13 = 20 instead of 41).  Measured: wgrad family 584 -> 1,513 us per step
// (gpurun_out/ab_hybrid.txt; parity green): a 768-row task of the largest shapes (six layer variants) already runs ~290 us of the
// launch's 530, four of them in a row ARE the launch.  The slab traffic (215 MB written, 215 MB read back) is the price of tasks
// short enough to balance.
#ifndef B3D_WGM_HYBRID
#define B3D_WGM_HYBRID 0
#endif
struct WgSplit { int r1, big, na, nb; };          // rows in long tasks, their size, number of long / short tasks
static WgSplit wg_hybrid(long rows, int rpt) {
  WgSplit s{0, 4 * rpt, 0, (int)((rows + rpt - 1) / rpt)};
  static const bool on = []() { const char* ev = getenv("B3D_WGM_HYBRID"); return ev ? atoi(ev) != 0 : (B3D_WGM_HYBRID != 0); }();
  if (!on || rows < 8L * rpt) { if (s.nb < 1) s.nb = 1; return s; }
  s.na = (int)((rows * 3 / 4) / s.big);
  s.r1 = s.na * s.big;
  s.nb = (int)((rows - s.r1 + rpt - 1) / rpt);
  return s;
}
struct WgBuilder {
  const int* iota = nullptr;
  WsLauncher wl, wlc;
  std::vector<WsJob> coop;                       // added to wlc longest task first (launch)
  void begin(WsJob* table, int table_cap, int* task_job, int task_cap, const int* iota_, hipStream_t stream) {
    iota = iota_;
    const int jobs_c = table_cap / 2, tasks_c = task_cap / 2;
    wl.begin(table, table_cap - jobs_c, task_job, task_cap - tasks_c, stream);
    wlc.begin(table + (table_cap - jobs_c), jobs_c, task_job + (task_cap - tasks_c), tasks_c, stream);
    coop.clear();
    coop.reserve(128);
  }
  // Cooperative decomposition: column groups of <= 256 per activation segment, row groups chosen per column group
  // from the compiled shapes; false (nothing added) if some pair has none.
  static int col_groups(int width, int* out) {
    switch (width) {
      case 512: out[0] = 256; out[1] = 256; return 2;
      case 384: out[0] = 256; out[1] = 128; return 2;
      case 288: out[0] = 192; out[1] = 96; return 2;
      case 256: case 192: case 128: case 96: case 64: out[0] = width; return 1;
      default: return 0;
    }
  }
  static int row_groups(int n, int kg, int* out) {
    int k = 0;
    if (kg <= 96) {
      while (n >= 256) { out[k++] = 256; n -= 256; }
      while (n >= 192) { out[k++] = 192; n -= 192; }
      if (n == 128 && kg == 96) { out[k++] = 128; n = 0; }
    } else if (kg == 128) {
      if (n % 192 == 0) while (n > 0) { out[k++] = 192; n -= 192; }
      while (n >= 128) { out[k++] = 128; n -= 128; }
      if (n == 96 || n == 64) { out[k++] = n; n = 0; }
    } else {
      while (n >= 128) { out[k++] = 128; n -= 128; }
    }
    return n == 0 ? k : -1;
  }
  bool add_block_coop(LinSlab& ls, long rows, int nvar, int rpt, const float* gp, const int* gidx, long gvs, int gstride,
                      int gcol0, const Col* cols, int ncols, bool with_bias) {
    WsJob jobs[48];
    int nj = 0;
    int wcol = 0;
    for (int ci = 0; ci < ncols; ++ci) {
      if (cols[ci].idx) return false;                      // gathered activations: streaming kernel
      if ((uintptr_t)cols[ci].p % 16 != 0 || cols[ci].stride % 4 != 0 || cols[ci].col0 % 4 != 0 || cols[ci].vstride % 4 != 0) return false;
      int cg[4];
      int ncg = col_groups(cols[ci].width, cg);
      if (ncg == 0) return false;
      int c0 = 0;
      for (int k = 0; k < ncg; ++k) {
        int kgs[4] = {cg[k], 0, 0, 0}, nk = 1;
        int rg[8];
        int nr = row_groups(ls.N, cg[k], rg);
        if (nr < 0 && cg[k] > 128) {                       // 192 x 256 and the like: narrower column groups
          nk = 0;
          for (int left = cg[k]; left > 0;) { const int t = left >= 128 ? 128 : left; kgs[nk++] = t; left -= t; }
        }
        for (int kk = 0; kk < nk; ++kk) {
          nr = row_groups(ls.N, kgs[kk], rg);
          if (nr < 0) return false;
          int g0 = 0;
          for (int r = 0; r < nr; ++r) {
            const int shape = wgm_shape(rg[r], kgs[kk]);
            if (shape < 0 || nj == 48) return false;
            WsJob jb;
            memset(&jb, 0, sizeof(jb));
            jb.g.ptr = gp; jb.g.idx = gidx ? gidx : iota; jb.g.vstride = gvs; jb.g.stride = gstride; jb.g.col0 = gcol0 + g0;
            jb.act[0].ptr = cols[ci].p; jb.act[0].idx = iota; jb.act[0].vstride = cols[ci].vstride;
            jb.act[0].stride = cols[ci].stride; jb.act[0].col0 = cols[ci].col0 + c0;
            jb.act[1] = jb.act[0]; jb.act[2] = jb.act[0];
            jb.wcol[0] = wcol + c0; jb.wrow = g0;
            jb.write_bias = (with_bias && ci == 0 && c0 == 0) ? 1 : 0;     // the first column group of every row group
            jb.shape = shape;
            jb.rows = (int)rows; jb.nvar = nvar; jb.rows_per_task = rpt;
            jb.NP = ls.NP; jb.KP = ls.KP; jb.slab = ls.slab;
            if (gidx && shape != WGM_128_192) return false;             // the only compiled gathered shape
            jobs[nj++] = jb;
            g0 += rg[r];
          }
          c0 += kgs[kk];
        }
      }
      wcol += cols[ci].width;
    }
    for (int j = 0; j < nj; ++j) push_coop(jobs[j]);
    ls.nchunks = coop_chunks(rows, rpt);          // (<= the chunks carve() sized the slab for: the reduction sums these)
    return true;
  }
  static int coop_chunks(long rows, int rpt) { const WgSplit sp = wg_hybrid(rows, rpt); return sp.na + sp.nb; }
  // one job -> its long-task part and its short-task part (slabs [0, na) and [na, na + nb) of the same matrix)
  void push_coop(const WsJob& j) {
    const WgSplit sp = wg_hybrid(j.rows, j.rows_per_task);
    if (sp.na == 0) { coop.push_back(j); return; }
    const size_t cs = (size_t)j.NP * j.KP + j.NP;
    WsJob a = j;
    a.rows = sp.r1; a.rows_per_task = sp.big;
    coop.push_back(a);
    if (j.rows > sp.r1) {
      WsJob b = j;
      b.rows = j.rows - sp.r1;
      b.slab = j.slab + (size_t)sp.na * cs;
      if (j.g.idx != iota) b.g.idx = j.g.idx + sp.r1;                       // gathered gradient rows: the index list moves
      else b.g.ptr = j.g.ptr + (size_t)sp.r1 * j.g.stride;
      for (int k = 0; k < 3; ++k) b.act[k].ptr = j.act[k].ptr + (size_t)sp.r1 * j.act[k].stride;   // (coop jobs: activations are never gathered)
      coop.push_back(b);
    }
  }
  void add_block(LinSlab& ls, long rows, int nvar, int rpt, const float* gp, const int* gidx, long gvs, int gstride, int gcol0,
                 const Col* cols, int ncols, bool with_bias) {
    if (nvar <= 0) return;
    ls.used = true;
    if (((uintptr_t)gp % 16 == 0) && gstride % 4 == 0 && gcol0 % 4 == 0 &&
        add_block_coop(ls, rows, nvar, rpt, gp, gidx, gvs, gstride, gcol0, cols, ncols, with_bias))
      return;
    bool first_job_of_group = true;
    for (int g0 = 0; g0 < ls.N; g0 += 64) {
      const int gw = (ls.N - g0 >= 64) ? 64 : ls.N - g0;       // 64, or the 32-row tail of a 96-row matrix
      int wcol = 0;
      first_job_of_group = with_bias;
      for (int ci = 0; ci < ncols; ++ci) {
        for (int c0 = 0; c0 < cols[ci].width;) {
          int cw = cols[ci].width - c0;           // column groups of 96 or 64: 128 -> 64+64, 256 -> 96+96+64
          cw = (cw == 128 || cw < 96) ? 64 : 96;
          WsJob jb;
          memset(&jb, 0, sizeof(jb));
          jb.g.ptr = gp; jb.g.idx = gidx ? gidx : iota; jb.g.vstride = gvs; jb.g.stride = gstride; jb.g.col0 = gcol0 + g0;
          jb.act[0].ptr = cols[ci].p; jb.act[0].idx = cols[ci].idx ? cols[ci].idx : iota; jb.act[0].vstride = cols[ci].vstride;
          jb.act[0].stride = cols[ci].stride; jb.act[0].col0 = cols[ci].col0 + c0;
          jb.act[1] = jb.act[0]; jb.act[2] = jb.act[0];
          jb.wcol[0] = wcol + c0; jb.wcol[1] = 0; jb.wcol[2] = 0; jb.wrow = g0;
          jb.write_bias = first_job_of_group ? 1 : 0;
          first_job_of_group = false;
          jb.shape = (gw == 64) ? (cw == 96 ? WS_64_96 : WS_64_64) : WS_32_64;   // 32-row tail only with 64-col groups
          jb.rows = (int)rows; jb.nvar = nvar; jb.rows_per_task = rpt;
          jb.NP = ls.NP; jb.KP = ls.KP; jb.slab = ls.slab;
          wl.add(jb);
          c0 += cw;
        }
        wcol += cols[ci].width;
      }
    }
  }
  // edge_update.0's per-edge columns [e | att]: one job over both sources (GdH1, the bulk of the bytes, is read once)
  void add_eu0_edge_columns(LinSlab& ls, long rows, int nvar, int rpt, const float* gdh1, long gvs, int gstride, const float* e, long evs,
                            int estride, const float* att) {
    ls.used = true;
    WsJob jb;
    memset(&jb, 0, sizeof(jb));
    jb.g.ptr = gdh1; jb.g.idx = iota; jb.g.vstride = gvs; jb.g.stride = gstride; jb.g.col0 = 0;
    jb.act[0].ptr = e; jb.act[0].idx = iota; jb.act[0].vstride = evs; jb.act[0].stride = estride; jb.act[0].col0 = 0;
    jb.act[1].ptr = att; jb.act[1].idx = iota; jb.act[1].vstride = 0; jb.act[1].stride = 64; jb.act[1].col0 = 0;
    jb.act[2] = jb.act[0];
    jb.wcol[0] = 0; jb.wcol[1] = 64; jb.wrow = 0; jb.write_bias = 1; jb.shape = WGM_256_128;
    jb.rows = (int)rows; jb.nvar = nvar; jb.rows_per_task = rpt; jb.NP = ls.NP; jb.KP = ls.KP; jb.slab = ls.slab;
    push_coop(jb);
    ls.nchunks = coop_chunks(rows, rpt);
  }
  int launch(const float* zrow, hipStream_t stream) {
    // every streaming job has one un-gathered activation segment: LDS-DMA ring form
    B3D_REQUIRE(wl.launch2(wstream2_kernel, kWs2LdsBytes, zrow, iota, B3D_K_WGRAD_EDGE) == 0, "wstream2: LDS attribute");
    B3D_REQUIRE(wl.status == 0, "streaming weight gradient: job table overflow (%d jobs, %d tasks)", wl.njobs, wl.total_tasks);
    B3D_TRY(launch_check("wstream_kernel"));
    // One workgroup per task, dispatched in task order as CUs free up: longest tasks first, or the 60 us tasks of
    // att_edge_encoder (last in plan order) start when the rest of the chip has run dry.  Cost of a task ~ its 32-row
    // steps times the bytes of a step.
    auto cost = [](const WsJob& j) {
      static const int w[WGM_SHAPES] = {384, 320, 256, 192, 320, 352, 256, 288, 224, 320, 224, 384};
      const long rows = j.rows < j.rows_per_task ? j.rows : j.rows_per_task;
      return ((rows + kWgmRows - 1) / kWgmRows) * (long)j.nvar * w[j.shape];
    };
    std::stable_sort(coop.begin(), coop.end(), [&](const WsJob& a, const WsJob& b) { return cost(a) > cost(b); });
    for (const WsJob& j : coop) wlc.add(j);
    wlc.flush();
    B3D_REQUIRE(wlc.status == 0, "cooperative weight gradient: job table overflow (%d jobs, %d tasks)", wlc.njobs, wlc.total_tasks);
    if (wlc.total_tasks > 0) {
      B3D_TRY(set_lds(wgemm_kernel, kWgmLdsBytes));
      ProfScope ps(B3D_K_WGRAD_EDGE, stream);
      hipLaunchKernelGGL(wgemm_kernel, dim3((unsigned)(wlc.total_tasks < 2048 ? wlc.total_tasks : 2048)), dim3(kWgmThreads), kWgmLdsBytes,
                         stream, (const WsJob*)wlc.table, (const int*)wlc.task_job, wlc.total_tasks, iota);
      B3D_TRY(launch_check("wgemm_kernel"));
    }
    return B3D_OK;
  }
};

constexpr int kStreamRowsPerTaskFc = 128;     // modality heads: a few thousand rows
static bool is_fc(int lin) { return lin >= FL0 && lin <= FR2; }
static bool is_streamed(int lin) { return lin >= AT0 || is_fc(lin); }   // att_edge_encoder, message passing, fc heads

static void carve(Ws& w, void* ws, size_t ws_bytes, int N, int E, int nl, int nr, int depth, uint32_t flags) {
  Carver c(ws, ws_bytes);
  const bool tr = flags & B3D_FLAG_TRAINING;
  const size_t e_ = edge_rows(E), n_ = (size_t)(N > 0 ? N : 1);
  const size_t l_ = (size_t)(nl > 0 ? nl : 1), r_ = (size_t)(nr > 0 ? nr : 1);
  memset(&w, 0, sizeof(w));
  w.wp_ee = c.take<float>(SeqEE::TOTAL_FLOATS);
  w.wp_ne = c.take<float>(SeqNE::TOTAL_FLOATS);
  w.wp_cls = c.take<float>(SeqCls::TOTAL_FLOATS);
  w.wp_fl = c.take<float>(SeqFL::TOTAL_FLOATS);
  w.wp_fr = c.take<float>(SeqFR::TOTAL_FLOATS);
  w.wp_aff[0] = c.take<float>(SeqAff<96>::TOTAL_FLOATS);
  w.wp_aff[1] = c.take<float>(SeqAff<128>::TOTAL_FLOATS);
  w.wp_aff[2] = c.take<float>(SeqAff<64>::TOTAL_FLOATS);
  w.wp_at[1] = c.take<float>(SeqAT1::TOTAL_FLOATS);
  w.wp_at[2] = c.take<float>(SeqAT2::TOTAL_FLOATS);
  w.wp_at[3] = c.take<float>(SeqAT3::TOTAL_FLOATS);
  w.wp_at[4] = c.take<float>(SeqAT4::TOTAL_FLOATS);
  w.wp_nfwd = c.take<float>(D::NodeFwdSeq::TOTAL_FLOATS);
  w.wp_proj0 = c.take<float>(Proj0Seq2<DB>::TOTAL_FLOATS);
  w.wp_nfwd_h = c.take<float>(NodeFwdHSeq<DB>::TOTAL_FLOATS);
  w.wp_efwd2 = c.take<float>(ES::Fwd::TOTAL_FLOATS);
  w.wp_efwd_nm2 = c.take<float>(ES::FwdNoMsg::TOTAL_FLOATS);     // (always carved: the layout must not depend on a forward-only flag)
  w.T = c.take<float>(n_ * HC::TW);
  w.T0 = c.take<float>(n_ * 2 * DB::MH);
  w.wp_attU = c.take<float>(SeqAttU::TOTAL_FLOATS);
  w.wp_att0 = c.take<float>(Att0Seq::TOTAL_FLOATS);
  w.U = c.take<float>(n_ * 1024);
  w.xsens = c.take<float>(n_ * XS);
  w.s = c.take<float>(n_ * XS);
  w.att = c.take<float>(e_ * 64);
  w.fut = c.take<float>(e_ * D::DM);
  w.past = c.take<float>((e_ + es::kPastDumpRows) * D::DM);      // + the dump rows of the in-wave per-destination sums (b3d_edge2.hpp)
  w.prob = c.take<float>(e_);
  const int adims[4] = {512, 384, 256, 128};
  const int affd[3] = {96, 128, 64};
  if (!tr) {
    // inference: hidden activations of the wide encoder ping-pong through two buffers
    w.A[0] = c.take<float>(e_ * 512); w.A[1] = c.take<float>(e_ * 384); w.A[2] = w.A[0]; w.A[3] = w.A[1];
    w.x[0] = c.take<float>(n_ * D::DX); w.x[1] = c.take<float>(n_ * D::DX); w.x[2] = c.take<float>(n_ * D::DX);
    w.e[0] = c.take<float>(e_ * D::DE); w.e[1] = c.take<float>(e_ * D::DE); w.e[2] = c.take<float>(e_ * D::DE);
    for (int l = 3; l <= depth; ++l) { w.x[l] = w.x[1 + (l - 1) % 2]; w.e[l] = w.e[1 + (l - 1) % 2]; }
    for (int m = 0; m < 3; ++m) w.aff_v[m] = nullptr;
  } else {
    w.wp_clsT = c.take<float>(SeqClsT::TOTAL_FLOATS);
    w.wp_eeT = c.take<float>(SeqEET::TOTAL_FLOATS);
    w.wp_neT = c.take<float>(SeqNET::TOTAL_FLOATS);
    w.wp_flT = c.take<float>(SeqFLT::TOTAL_FLOATS);
    w.wp_frT = c.take<float>(SeqFRT::TOTAL_FLOATS);
    w.wp_affT[0] = c.take<float>(SeqAffT<96>::TOTAL_FLOATS);
    w.wp_affT[1] = c.take<float>(SeqAffT<128>::TOTAL_FLOATS);
    w.wp_affT[2] = c.take<float>(SeqAffT<64>::TOTAL_FLOATS);
    w.wp_atT[1] = c.take<float>(SeqAT1T::TOTAL_FLOATS);
    w.wp_atT[2] = c.take<float>(SeqAT2T::TOTAL_FLOATS);
    w.wp_atT[3] = c.take<float>(SeqAT3T::TOTAL_FLOATS);
    w.wp_atT[4] = c.take<float>(SeqAT4T::TOTAL_FLOATS);
    w.wp_ebwd2 = c.take<float>(ES::Bwd::TOTAL_FLOATS);
    w.wp_ebwd_nm2 = c.take<float>(ES::BwdNoMsg::TOTAL_FLOATS);
    w.wp_nbwd_h = c.take<float>(NodeBwdHSeq<DB>::TOTAL_FLOATS);
    w.wp_gproj = c.take<float>(HC::GradProjSeq::TOTAL_FLOATS);
    w.wp_attDs = c.take<float>(SeqAttDs::TOTAL_FLOATS);
    w.wp_at0eT = c.take<float>(SeqAT0eT::TOTAL_FLOATS);
    w.dU = c.take<float>(n_ * 1024);
    w.ds = c.take<float>(n_ * XS);
    w.de0 = c.take<float>(e_ * 64);
    w.dT = c.take<float>((size_t)depth * n_ * HC::GW);
    w.gx = c.take<float>(n_ * 2 * D::DX);
    w.wp_nbwd = c.take<float>(D::NodeBwdSeq::TOTAL_FLOATS);
    w.ea_pad = c.take<float>(e_ * 16); w.ee_a1 = c.take<float>(e_ * 16); w.ee_a2 = c.take<float>(e_ * 32);
    w.pose_pad = c.take<float>(n_ * 32); w.ne_a1 = c.take<float>(n_ * 48);
    w.fl_a1 = c.take<float>(l_ * 192); w.fr_a1 = c.take<float>(r_ * 192); w.fr_a2 = c.take<float>(r_ * 128);
    for (int m = 0; m < 3; ++m) w.aff_v[m] = c.take<float>(n_ * affd[m]);
    for (int i = 0; i < 4; ++i) w.A[i] = c.take<float>(e_ * adims[i]);
    for (int i = 0; i < 4; ++i) w.amask[i] = reinterpret_cast<unsigned*>(c.take<float>(e_ * 16));
    {
      float* xb = c.take<float>((size_t)(depth + 1) * n_ * D::DX);
      float* eb = c.take<float>((size_t)(depth + 1) * e_ * D::DE);
      for (int l = 0; l <= depth; ++l) {
        w.x[l] = xb ? xb + (size_t)l * n_ * D::DX : nullptr;
        w.e[l] = eb ? eb + (size_t)l * e_ * D::DE : nullptr;
      }
    }
    auto per_layer = [&](float** arr, size_t per) {
      float* b = c.take<float>((size_t)depth * per);
      for (int l = 0; l < depth; ++l) arr[l] = b ? b + (size_t)l * per : nullptr;
    };
    per_layer(w.sH1, e_ * D::EH1); per_layer(w.sH2, e_ * D::EH2); per_layer(w.sF1, e_ * D::MH); per_layer(w.sP1, e_ * D::MH);
    per_layer(w.rmask, e_ * es::kMaskFloatsPerRow);
    per_layer(w.M, n_ * D::NIN); per_layer(w.nH1, n_ * D::NH1); per_layer(w.nH2, n_ * D::NH2);
    w.c_a1 = c.take<float>(e_ * 32); w.c_a2 = c.take<float>(e_ * 16); w.c_a3 = c.take<float>(e_ * 16);
    w.de[0] = c.take<float>(e_ * D::DE); w.de[1] = c.take<float>(e_ * D::DE);
    w.da_acc = c.take<float>(e_ * D::DA);
    w.gdst = c.take<float>(e_ * 2 * D::DX); w.gsrc = c.take<float>(e_ * 2 * D::DX);
    w.dx0_acc = c.take<float>(n_ * D::DX);
    w.dM = c.take<float>((size_t)depth * n_ * D::NIN);
    w.GdH1 = c.take<float>((size_t)depth * e_ * D::EH1);
    w.GdH2 = c.take<float>((size_t)depth * e_ * D::EH2);
    w.Gde = c.take<float>((size_t)depth * e_ * D::DE);
    w.GdF1 = c.take<float>((size_t)depth * e_ * D::MH);
    w.GdP1 = c.take<float>((size_t)depth * e_ * D::MH);
    w.Gdx = c.take<float>((size_t)depth * n_ * D::DX);
    w.GnH2 = c.take<float>((size_t)depth * n_ * D::NH2);
    w.GnH1 = c.take<float>((size_t)depth * n_ * D::NH1);
    w.gc_top = c.take<float>(e_ * 16); w.gc3 = c.take<float>(e_ * 16); w.gc2 = c.take<float>(e_ * 16); w.gc1 = c.take<float>(e_ * 32);
    w.ge_top = c.take<float>(e_ * 64); w.ge2 = c.take<float>(e_ * 32); w.ge1 = c.take<float>(e_ * 16);
    w.gn_top = c.take<float>(n_ * 96); w.gn1 = c.take<float>(n_ * 48);
    w.dA[0] = c.take<float>(e_ * 128); w.dA[1] = c.take<float>(e_ * 256); w.dA[2] = c.take<float>(e_ * 384); w.dA[3] = c.take<float>(e_ * 512);
    for (int m = 0; m < 3; ++m) { w.gaff_top[m] = c.take<float>(n_ * affd[m]); w.gaff_v[m] = c.take<float>(n_ * affd[m]); }
    w.dxs = c.take<float>(n_ * XS);
    w.gfl_top = c.take<float>(l_ * 128); w.gfl1 = c.take<float>(l_ * 192);
    w.gfr_top = c.take<float>(r_ * 64); w.gfr2 = c.take<float>(r_ * 128); w.gfr1 = c.take<float>(r_ * 192);
    w.zrow = c.take<float>(256);
    w.iota = c.take<int>((size_t)(E > N ? E : N) + 64);
    w.ws_table = c.take<WsJob>(kTableCap);
    w.ws_task_job = c.take<int>(kTaskCap);
    for (int i = 0; i < LIN_COUNT; ++i) {
      LinSlab& ls = w.lin[i];
      ls.N = kDims[i].N; ls.K = kDims[i].K; ls.NP = pad16(ls.N); ls.KP = pad16(ls.K);
      const long rows = kRowKind[i] == 0 ? E : kRowKind[i] == 1 ? N : kRowKind[i] == 2 ? nl : nr;
      if (is_streamed(i)) {
        const int rpt = is_fc(i) ? kStreamRowsPerTaskFc : (i <= AT4) ? kStreamRowsPerTaskAtt : kStreamRowsPerTask;
        ls.nchunks = (int)((rows + rpt - 1) / rpt);
        if (ls.nchunks < 1) ls.nchunks = 1;
      } else {
        ls.nchunks = wg_nchunks(rows, ls.NP, ls.KP, 0);
      }
      ls.slab = c.take<float>(wg_slab_floats(ls.nchunks, ls.NP, ls.KP));
      ls.used = false;
    }
    for (int v = 0; v < VL_COUNT; ++v) {
      LinSlab& ls = w.vlin[v];
      ls.N = kDims[kVl[v].lin].N; ls.K = kVl[v].width; ls.NP = pad16(ls.N); ls.KP = pad16(ls.K);
      const long rows = kVl[v].on_edges ? E : N;
      const int rpt = kVl[v].on_edges ? (kVl[v].lin == AT0 ? kStreamRowsPerTaskAtt : kStreamRowsPerTask)
                                      : (kVl[v].lin == AT0 ? kStreamNodeRowsPerTaskAtt : kStreamNodeRowsPerTask);
      ls.nchunks = (int)((rows + rpt - 1) / rpt);
      if (ls.nchunks < 1) ls.nchunks = 1;
      ls.slab = c.take<float>(wg_slab_floats(ls.nchunks, ls.NP, ls.KP));
      ls.used = false;
    }
  }
  if (flags & B3D_FLAG_RUN_DEAD_KNN) knn_carve(w.knn, c, N, D::DX);
  w.bytes = c.off + 256;
  w.ok = c.ok();
}

__global__ void iota_kernel(int* p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = i;
}

// strided row copy: dst[r][dcol0 + c] = src[r][c]
__global__ void copy_cols_kernel(const float* __restrict__ src, int sstride, float* __restrict__ dst, int dstride,
                                 int dcol0, int rows, int width) {
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (long)rows * width) return;
  const long r = id / width;
  const int cc = (int)(id - r * width);
  dst[r * dstride + dcol0 + cc] = src[r * sstride + cc];
}

// clr_att_gnn.py:107-121: has[n] = (sum of row n) != 0.  One wavefront per row.
__global__ __launch_bounds__(256) void modality_mask_kernel(const float* __restrict__ f, int N, int width,
                                                            uint8_t* __restrict__ has) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= N) return;
  float s = 0.f;
  for (int c = lane; c < width; c += 64) s += f[(size_t)row * width + c];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
  if (lane == 0) has[row] = (s != 0.f) ? 1 : 0;
}

// Ascending ids of the rows with has[n] != 0 (torch.nonzero of clr_att_gnn.py:107-121's masks) and their count: ONE workgroup, a
// ballot per wavefront and an LDS scan over the wavefronts per 1,024 rows (N is a few thousand: rocPRIM's partition + reduce +
// lookback launches cost 75 us per modality in front of every step).
// `cap` / `mismatch` (b3d_modality_rows_expect): at most cap ids are written, and *mismatch is incremented when the count is not cap --
// the captured-step form, where the count is a SHAPE baked into the graph and the device, not the host, checks it.
__global__ __launch_bounds__(1024) void compact_rows_kernel(const uint8_t* __restrict__ has, int N, long long* __restrict__ rows,
                                                            int* __restrict__ count, int cap, int* __restrict__ mismatch) {
  __shared__ int wsum[16];
  __shared__ int base_s;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (tid == 0) base_s = 0;
  __syncthreads();
  for (int n0 = 0; n0 < N; n0 += 1024) {
    const int n = n0 + tid;
    const bool f = n < N && has[n] != 0;
    const unsigned long long b = __ballot(f);
    const int below = __popcll(b & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(b);
    __syncthreads();
    int off = base_s;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    if (f && off + below < cap) rows[off + below] = n;
    __syncthreads();
    if (tid == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += wsum[w]; base_s += t; }
    __syncthreads();
  }
  if (tid == 0) {
    *count = base_s;
    if (mismatch && base_s != cap) atomicAdd(mismatch, 1);
  }
}

struct LinPtrs { const float* w; const float* b; };
static void gather_linears(const b3d_clr_weights* pw, LinPtrs* L) {
  auto put = [&](int first, const b3d_linear* a, int n) { for (int i = 0; i < n; ++i) L[first + i] = LinPtrs{a[i].w, a[i].b}; };
  put(EE0, pw->edge_encoder, 3); put(NE0, pw->node_encoder, 2); put(C0, pw->edge_classifier, 4);
  put(FL0, pw->fc_lidar_encoder, 2); put(FR0, pw->fc_radar_encoder, 3); put(AT0, pw->att_edge_encoder, 5);
  put(EU0, pw->mp.edge_update, 3); put(PA0, pw->mp.create_past_msgs, 2); put(FU0, pw->mp.create_future_msgs, 2);
  put(CF0, pw->mp.combine_future_past, 3);
  const b3d_mha* mh[3] = {&pw->c2c_att, &pw->l2l_att, &pw->r2r_att};
  const int dd[3] = {96, 128, 64};
  for (int m = 0; m < 3; ++m) {
    // value projection = rows [2D, 3D) of in_proj (clr_att_gnn.py:148-155 with one key: softmax == 1)
    L[AVC + 2 * m] = LinPtrs{mh[m]->in_proj_weight + (size_t)2 * dd[m] * dd[m], mh[m]->in_proj_bias + 2 * dd[m]};
    L[AOC + 2 * m] = LinPtrs{mh[m]->out_proj_weight, mh[m]->out_proj_bias};
  }
}

static int check_weights(const b3d_clr_weights* pw) {
  B3D_REQUIRE(pw != nullptr, "weights struct is null");
  LinPtrs L[LIN_COUNT];
  gather_linears(pw, L);
  for (int i = 0; i < LIN_COUNT; ++i) B3D_REQUIRE(L[i].w && L[i].b, "null weight/bias pointer (linear %d)", i);
  return B3D_OK;
}

// `bwd_stream`: where the images only the backward sweep reads (transposed weights, the backward fragment streams) are packed --
// the launch stream, or (round 6) a library side stream forked at the forward's entry and joined at its end: ~40 % of the packing
// leaves the launch stream's chain and runs under the forward's kernels.
static int pack_all(const b3d_clr_weights* pw, Ws& w, bool training, bool knn, hipStream_t stream, hipStream_t bwd_stream, bool fwd_nomsg) {
  LinPtrs L[LIN_COUNT];
  gather_linears(pw, L);
  PackDesc d[224];
  int n = 0;
  auto F = [&](auto tag, int li, float* base, int lin) {
    using S = decltype(tag);
    d[n++] = pack_desc<S>(li, base, L[lin].w, L[lin].b, kDims[lin].N, kDims[lin].K, false);
  };
  auto T = [&](auto tag, int li, float* base, int lin) {
    using S = decltype(tag);
    d[n++] = pack_desc<S>(li, base, L[lin].w, nullptr, kDims[lin].K, kDims[lin].N, true);
  };
  for (int i = 0; i < 3; ++i) F(SeqEE{}, i, w.wp_ee, EE0 + i);
  for (int i = 0; i < 2; ++i) F(SeqNE{}, i, w.wp_ne, NE0 + i);
  for (int i = 0; i < 4; ++i) F(SeqCls{}, i, w.wp_cls, C0 + i);
  for (int i = 0; i < 2; ++i) F(SeqFL{}, i, w.wp_fl, FL0 + i);
  for (int i = 0; i < 3; ++i) F(SeqFR{}, i, w.wp_fr, FR0 + i);
  F(SeqAff<96>{}, 0, w.wp_aff[0], AVC); F(SeqAff<96>{}, 1, w.wp_aff[0], AOC);
  F(SeqAff<128>{}, 0, w.wp_aff[1], AVL); F(SeqAff<128>{}, 1, w.wp_aff[1], AOL);
  F(SeqAff<64>{}, 0, w.wp_aff[2], AVR); F(SeqAff<64>{}, 1, w.wp_aff[2], AOR);
  F(SeqAT1{}, 0, w.wp_at[1], AT1); F(SeqAT2{}, 0, w.wp_at[2], AT2);
  F(SeqAT3{}, 0, w.wp_at[3], AT3); F(SeqAT4{}, 0, w.wp_at[4], AT4);
  for (int i = 0; i < 3; ++i) F(D::NodeFwdSeq{}, i, w.wp_nfwd, CF0 + i);
  constexpr int DX = DB::DX, DE = DB::DE, EIN = DB::EIN, MIN = DB::MIN, H1 = DB::EH1, MH = DB::MH;
  const LinPtrs &eu0 = L[EU0], &fu0 = L[FU0], &pa0 = L[PA0];
  // per-node table T = (eu0[:, x_i] x + b | eu0[:, x_j] x | fu0[:, x] x + b | pa0[:, x] x + b | GATConv.lin x)
  auto proj = [&](auto tag, int li, float* base) {
    using S = decltype(tag);
    d[n++] = pack_slice<S>(li, base, eu0.w, eu0.b, H1, DX, EIN, HC::OA, H1, false);
    d[n++] = pack_slice<S>(li, base, eu0.w + DX, nullptr, H1, DX, EIN, HC::OB, H1, false);
    d[n++] = pack_slice<S>(li, base, fu0.w, fu0.b, MH, DX, MIN, HC::OF, MH, false);
    d[n++] = pack_slice<S>(li, base, pa0.w, pa0.b, MH, DX, MIN, HC::OP, MH, false);
    d[n++] = pack_slice<S>(li, base, knn ? pw->knn_conv.lin : nullptr, nullptr, DX, DX, DX, HC::OG, DX, false);
  };
  d[n++] = pack_slice<Proj0Seq2<DB>>(0, w.wp_proj0, fu0.w + DX + DE, nullptr, MH, DX, MIN, 0, MH, false);    // x0 columns
  d[n++] = pack_slice<Proj0Seq2<DB>>(0, w.wp_proj0, pa0.w + DX + DE, nullptr, MH, DX, MIN, MH, MH, false);
  proj(Proj0Seq2<DB>{}, 1, w.wp_proj0);
  for (int i = 0; i < 3; ++i) F(NodeFwdHSeq<DB>{}, i, w.wp_nfwd_h, CF0 + i);
  proj(NodeFwdHSeq<DB>{}, 3, w.wp_nfwd_h);
  const LinPtrs& a0 = L[AT0];                               // att_edge_encoder.0 [512, 640]
  for (int k = 0; k < 3; ++k) {
    d[n++] = pack_slice<SeqAttU>(k, w.wp_attU, a0.w + 96 * k, k == 0 ? a0.b : nullptr, 512, 96, 640, 0, 512, false);
    d[n++] = pack_slice<SeqAttU>(k, w.wp_attU, a0.w + XS + 96 * k, nullptr, 512, 96, 640, 512, 512, false);
  }
  for (int cc = 0; cc < 4; ++cc)
    d[n++] = pack_slice<Att0Seq>(cc, w.wp_att0, a0.w + (size_t)128 * cc * 640 + 2 * XS, nullptr, 128, 64, 640, 0, 128, false);
  const int n_fwd = n;
  if (training) {
    T(SeqClsT{}, 0, w.wp_clsT, C3); T(SeqClsT{}, 1, w.wp_clsT, C2); T(SeqClsT{}, 2, w.wp_clsT, C1); T(SeqClsT{}, 3, w.wp_clsT, C0);
    T(SeqEET{}, 0, w.wp_eeT, EE2); T(SeqEET{}, 1, w.wp_eeT, EE1);
    T(SeqNET{}, 0, w.wp_neT, NE1);
    T(SeqFLT{}, 0, w.wp_flT, FL1);
    T(SeqFRT{}, 0, w.wp_frT, FR2); T(SeqFRT{}, 1, w.wp_frT, FR1);
    T(SeqAffT<96>{}, 0, w.wp_affT[0], AOC); T(SeqAffT<96>{}, 1, w.wp_affT[0], AVC);
    T(SeqAffT<128>{}, 0, w.wp_affT[1], AOL); T(SeqAffT<128>{}, 1, w.wp_affT[1], AVL);
    T(SeqAffT<64>{}, 0, w.wp_affT[2], AOR); T(SeqAffT<64>{}, 1, w.wp_affT[2], AVR);
    T(SeqAT1T{}, 0, w.wp_atT[1], AT1); T(SeqAT2T{}, 0, w.wp_atT[2], AT2);
    T(SeqAT3T{}, 0, w.wp_atT[3], AT3); T(SeqAT4T{}, 0, w.wp_atT[4], AT4);
    using NB = D::NodeBwdSeq;
    T(NB{}, 0, w.wp_nbwd, CF2); T(NB{}, 1, w.wp_nbwd, CF1); T(NB{}, 2, w.wp_nbwd, CF0);
    // (dx | dx0) = sum over the four lists of (node columns)^T . dT_list: rows 0:DX from the x columns, rows DX:2DX
    // (future / past only) from the x0 columns
    using NH = NodeBwdHSeq<DB>;
    auto gp = [&](auto tag, float* base) {
      using S = decltype(tag);
      d[n++] = pack_slice<S>(0, base, eu0.w, nullptr, DX, H1, EIN, 0, DX, true);
      d[n++] = pack_slice<S>(1, base, eu0.w + DX, nullptr, DX, H1, EIN, 0, DX, true);
      d[n++] = pack_slice<S>(2, base, fu0.w, nullptr, DX, MH, MIN, 0, DX, true);
      d[n++] = pack_slice<S>(2, base, fu0.w + DX + DE, nullptr, DX, MH, MIN, DX, DX, true);
      d[n++] = pack_slice<S>(3, base, pa0.w, nullptr, DX, MH, MIN, 0, DX, true);
      d[n++] = pack_slice<S>(3, base, pa0.w + DX + DE, nullptr, DX, MH, MIN, DX, DX, true);
    };
    gp(NH{}, w.wp_nbwd_h);
    gp(HC::GradProjSeq{}, w.wp_gproj);
    T(NH{}, 4, w.wp_nbwd_h, CF2); T(NH{}, 5, w.wp_nbwd_h, CF1); T(NH{}, 6, w.wp_nbwd_h, CF0);
    constexpr int DSK = B3D_ATT_DS_SLICE, DSH = 512 / DSK;   // slice width, slices per half
    static_assert(SeqAttDs::NL == 2 * DSH && SeqAttDs::kp(0) == DSK, "K-slices of the two 512-wide halves of dU");
    for (int k = 0; k < 2 * DSH; ++k)                        // slice k: DSK columns of dU_i (k < DSH) or dU_j
      d[n++] = pack_slice<SeqAttDs>(k, w.wp_attDs, a0.w + (size_t)DSK * (k % DSH) * 640 + (k < DSH ? 0 : XS), nullptr, XS, DSK, 640, 0, XS, true);
    d[n++] = pack_slice<SeqAT0eT>(0, w.wp_at0eT, a0.w + 2 * XS, nullptr, 64, 512, 640, 0, 64, true);
  }
  if (n > 224) return fail(B3D_ERR_ARG, "pack descriptor table overflow");
  if (bwd_stream == stream) {
    B3D_TRY(pack_images(d, n, stream));                    // one table: the launches are filled to kPackMax descriptors
  } else {
    B3D_TRY(pack_images(d, n_fwd, stream));
    if (n > n_fwd) B3D_TRY(pack_images(d + n_fwd, n - n_fwd, bwd_stream));
  }
  FragDesc f[kFragMax];
  int m = 0;
  using FS = ES::Fwd;
  f[m++] = frag_desc<FS>(0, w.wp_efwd2, eu0.w + 2 * DX, nullptr, EIN, false);        // e | att columns (bias: in the table T)
  f[m++] = frag_desc<FS>(1, w.wp_efwd2, L[EU1].w, L[EU1].b, kDims[EU1].K, false);
  f[m++] = frag_desc<FS>(2, w.wp_efwd2, L[EU2].w, L[EU2].b, kDims[EU2].K, false);
  f[m++] = frag_desc<FS>(3, w.wp_efwd2, fu0.w + DX, nullptr, MIN, false);             // e' columns
  f[m++] = frag_desc<FS>(4, w.wp_efwd2, L[FU1].w, L[FU1].b, kDims[FU1].K, false);
  f[m++] = frag_desc<FS>(5, w.wp_efwd2, pa0.w + DX, nullptr, MIN, false);
  f[m++] = frag_desc<FS>(6, w.wp_efwd2, L[PA1].w, L[PA1].b, kDims[PA1].K, false);
  if (fwd_nomsg) {
    using FN = ES::FwdNoMsg;
    f[m++] = frag_desc<FN>(0, w.wp_efwd_nm2, eu0.w + 2 * DX, nullptr, EIN, false);
    f[m++] = frag_desc<FN>(1, w.wp_efwd_nm2, L[EU1].w, L[EU1].b, kDims[EU1].K, false);
    f[m++] = frag_desc<FN>(2, w.wp_efwd_nm2, L[EU2].w, L[EU2].b, kDims[EU2].K, false);
  }
  const int m_fwd = m;
  if (training) {
    using BS = ES::Bwd;
    f[m++] = frag_desc<BS>(0, w.wp_ebwd2, L[PA1].w, nullptr, kDims[PA1].K, true);
    f[m++] = frag_desc<BS>(1, w.wp_ebwd2, pa0.w + DX, nullptr, MIN, true);
    f[m++] = frag_desc<BS>(2, w.wp_ebwd2, L[FU1].w, nullptr, kDims[FU1].K, true);
    f[m++] = frag_desc<BS>(3, w.wp_ebwd2, fu0.w + DX, nullptr, MIN, true);
    f[m++] = frag_desc<BS>(4, w.wp_ebwd2, L[EU2].w, nullptr, kDims[EU2].K, true);
    f[m++] = frag_desc<BS>(5, w.wp_ebwd2, L[EU1].w, nullptr, kDims[EU1].K, true);
    f[m++] = frag_desc<BS>(6, w.wp_ebwd2, eu0.w + 2 * DX, nullptr, EIN, true);
    using NS = ES::BwdNoMsg;
    f[m++] = frag_desc<NS>(0, w.wp_ebwd_nm2, L[EU2].w, nullptr, kDims[EU2].K, true);
    f[m++] = frag_desc<NS>(1, w.wp_ebwd_nm2, L[EU1].w, nullptr, kDims[EU1].K, true);
    f[m++] = frag_desc<NS>(2, w.wp_ebwd_nm2, eu0.w + 2 * DX, nullptr, EIN, true);
  }
  if (bwd_stream == stream) {
    B3D_TRY(pack_frags(f, m, stream));
  } else {
    B3D_TRY(pack_frags(f, m_fwd, stream));
    if (m > m_fwd) B3D_TRY(pack_frags(f + m_fwd, m - m_fwd, bwd_stream));
  }
  return B3D_OK;
}

template <class Seq, bool RELU, bool BIAS, class In>
static int wide(const char* name, const In& in, long rows, float* out, int ostride, int ocol0, const unsigned* mask_in, unsigned* mask_out,
                const float* wp, hipStream_t stream, int family = B3D_K_ATT_FWD) {
  B3D_REQUIRE((unsigned long long)rows * (unsigned long long)ostride * 4ull < (1ull << 32),
              "wide_linear: %ld rows x %d columns exceed the 32-bit row offsets of this kernel", rows, ostride);
  WideArgs<In> a;
  a.rows = (int)rows; a.in = in; a.out = out; a.out_stride = ostride; a.out_col0 = ocol0; a.mask_in = mask_in; a.mask_out = mask_out; a.wpack = wp;
  return launch_rows<kNWEdge>(wide_linear_kernel<Seq, RELU, BIAS, In, kNWEdge>, name, a, rows, stream, family, chain_lds<Seq>());
}

template <class Seq>
static int node_linear(const char* name, const float* in, int in_stride, int in_col0, float* out, int out_stride, int N,
                       const float* wp, hipStream_t stream, int family) {
  NodeLinArgs a;
  a.N = N; a.in = in; a.in_stride = in_stride; a.in_col0 = in_col0; a.out = out; a.out_stride = out_stride; a.out_col0 = 0; a.wpack = wp;
  B3D_TRY(set_lds(att_node_linear_kernel<Seq>, kNodeLinLds));
  ProfScope ps(family, stream);
  hipLaunchKernelGGL(att_node_linear_kernel<Seq>, dim3((N + 15) / 16), dim3(kNodeLinWaves * 64), kNodeLinLds, stream, a);
  return launch_check(name);
}

// One modality's out_proj(v_proj(x)) on all nodes: x = xsens[:, xc : xc+DD] -> s[:, sc : sc+DD]
template <int DD>
static int affine_fwd(Ws& w, int m, int N, int xc, int sc, hipStream_t stream) {
  using In = LoadAligned<DD / 16>;
  using Out = StoreAligned<DD / 16>;
  ChainFwdArgs<In, Out> a;
  memset(&a, 0, sizeof(a));
  a.rows = N; a.in = In{w.xsens, nullptr, XS, xc}; a.out = Out{w.s, nullptr, XS, sc};
  a.save[0] = w.aff_v[m];
  a.wpack = w.wp_aff[m];
  return launch_rows<kNWNode>(chain_fwd_kernel<SeqAff<DD>, 0u, In, Out, kNWNode>, "modality_affine", a, N, stream, B3D_K_OTHER, chain_lds<SeqAff<DD>>());
}

// backward of one modality: d s[:, sc : sc+DD] -> d x_m = dxs[:, xc : xc+DD]
template <int DD>
static int affine_bwd(Ws& w, int m, int N, int xc, int sc, hipStream_t stream) {      // d s arrives per node (att_edge_encoder.0 is hoisted)
  using In = LoadAligned<DD / 16>;
  using Out = StoreAligned<DD / 16>;
  ChainBwdArgs<In, Out> a;
  memset(&a, 0, sizeof(a));
  a.rows = N;
  a.in = In{w.ds, nullptr, XS, sc};
  a.out = Out{w.dxs, nullptr, XS, xc};
  a.gtop = w.gaff_top[m];
  a.gsave[0] = w.gaff_v[m];
  a.wpack = w.wp_affT[m];
  return launch_rows<kNWNode>(chain_bwd_kernel<SeqAffT<DD>, In, Out, kNWNode>, "modality_affine_bwd", a, N, stream, B3D_K_OTHER, chain_lds<SeqAffT<DD>>());
}
}  // namespace clr
}  // namespace b3d

using namespace b3d;
using namespace b3d::clr;

extern "C" uint32_t b3d_features(void) {
  return B3D_FEATURE_POSE_HOIST | B3D_FEATURE_CLR_HOIST_MP | B3D_FEATURE_CLR_HOIST_ATT;
}

extern "C" int b3d_modality_mask(const float* feats, int32_t N, int32_t width, uint8_t* has, b3d_stream stream_) {
  B3D_REQUIRE(feats && has && N >= 0 && width > 0, "b3d_modality_mask: bad argument");
  if (N == 0) return B3D_OK;
  hipLaunchKernelGGL(modality_mask_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream_, feats, N, width, has);
  return launch_check("modality_mask_kernel");
}

extern "C" int b3d_modality_rows(const float* feats, int32_t N, int32_t width, uint8_t* has, int64_t* rows, int32_t* count,
                                 b3d_stream stream_) {
  B3D_REQUIRE(feats && has && rows && count && N >= 0 && width > 0, "b3d_modality_rows: bad argument");
  hipStream_t stream = (hipStream_t)stream_;
  if (N > 0) {
    hipLaunchKernelGGL(modality_mask_kernel, dim3((N + 3) / 4), dim3(256), 0, stream, feats, N, width, has);
    B3D_TRY(launch_check("modality_mask_kernel"));
  }
  hipLaunchKernelGGL(compact_rows_kernel, dim3(1), dim3(1024), 0, stream, has, N, (long long*)rows, count, N, (int*)nullptr);
  return launch_check("compact_rows_kernel");
}

extern "C" int b3d_modality_rows_expect(const float* feats, int32_t N, int32_t width, uint8_t* has, int64_t* rows, int32_t expected,
                                        int32_t* count, int32_t* mismatch, b3d_stream stream_) {
  B3D_REQUIRE(feats && has && rows && count && mismatch && N >= 0 && width > 0 && expected >= 0 && expected <= N,
              "b3d_modality_rows_expect: bad argument");
  hipStream_t stream = (hipStream_t)stream_;
  if (N > 0) {
    hipLaunchKernelGGL(modality_mask_kernel, dim3((N + 3) / 4), dim3(256), 0, stream, feats, N, width, has);
    B3D_TRY(launch_check("modality_mask_kernel"));
  }
  hipLaunchKernelGGL(compact_rows_kernel, dim3(1), dim3(1024), 0, stream, has, N, (long long*)rows, count, expected, mismatch);
  return launch_check("compact_rows_kernel");
}

extern "C" size_t b3d_clr_workspace_bytes(int32_t N, int32_t E, int32_t n_lidar, int32_t n_radar, int32_t depth, uint32_t flags) {
  if (depth < 1 || depth > 15) return 0;
  Ws w;
  carve(w, nullptr, 0, N, E, n_lidar, n_radar, depth, flags);
  return w.bytes;
}

extern "C" int b3d_clr_debug_ptrs(void* workspace, size_t workspace_bytes, int32_t N, int32_t E, int32_t nl, int32_t nr,
                                  int32_t depth, uint32_t flags, int32_t layer, float** x, float** e, float** att) {
  B3D_REQUIRE(depth >= 1 && depth <= 15 && layer >= 0 && layer <= depth, "bad layer");
  Ws w;
  carve(w, workspace, workspace_bytes, N, E, nl, nr, depth, flags);
  if (!w.ok) return fail(B3D_ERR_WORKSPACE, "workspace too small");
  *x = w.x[layer]; *e = w.e[layer]; *att = w.att;
  return B3D_OK;
}

extern "C" int b3d_clr_forward(const b3d_clr_weights* pw, const b3d_graph* g, const b3d_clr_inputs* in, int32_t depth,
                               uint32_t flags, void* workspace, size_t workspace_bytes, float* out_prob,
                               float* out_x_sens, b3d_stream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  B3D_TRY(check_weights(pw));
  B3D_REQUIRE(g && in && workspace && out_prob && out_x_sens, "b3d_clr_forward: null argument");
  B3D_REQUIRE(in->pose_feats && in->edge_attr && in->x_img, "b3d_clr_forward: null input tensor");
  B3D_REQUIRE(depth >= 1 && depth <= 15, "b3d_clr_forward: depth %d outside [1,15]", depth);
  const int N = g->N, E = g->E, nl = in->n_lidar, nr = in->n_radar;
  B3D_REQUIRE(N > 0 && E > 0, "b3d_clr_forward: empty graph (N=%d, E=%d)", N, E);
  B3D_REQUIRE(nl >= 0 && nl <= N && nr >= 0 && nr <= N, "b3d_clr_forward: bad modality row counts");
  B3D_REQUIRE(nl == 0 || (in->pointnet_out && in->lidar_nodes), "b3d_clr_forward: LiDAR rows without data");
  B3D_REQUIRE(nr == 0 || (in->radarnet_out && in->radar_nodes), "b3d_clr_forward: radar rows without data");
  const bool tr = flags & B3D_FLAG_TRAINING;
  Ws w;
  carve(w, workspace, workspace_bytes, N, E, nl, nr, depth, flags);
  if (!w.ok) return fail(B3D_ERR_WORKSPACE, "b3d_clr_forward: workspace %zu < %zu bytes", workspace_bytes, w.bytes);
  if (flags & B3D_FLAG_RUN_DEAD_KNN)
    B3D_REQUIRE(pw->knn_conv.lin && pw->knn_conv.att_src && pw->knn_conv.att_dst && pw->knn_conv.bias,
                "b3d_clr_forward: knn_conv pointers are required with B3D_FLAG_RUN_DEAD_KNN");
  Side* pack_side = nullptr;
  static const bool pack_on_side = []() { const char* ev = getenv("B3D_PACK_SIDE"); return ev && atoi(ev) != 0; }();   // A/B switch, default OFF: measured neutral (round 6: 3.976 / 3.981 vs 3.963 / 3.958 ms) -- with the encode-ahead branch beside it the step is bound by the SUM of its kernels, not by the launch stream's chain
  if (tr && !(flags & B3D_FLAG_SINGLE_STREAM) && pack_on_side) {
    B3D_TRY(side_get(1, &pack_side));
    B3D_TRY(side_fork(stream, pack_side));                  // (the images' previous readers -- the last backward sweep -- are behind us on `stream`)
  }
  const bool skip_last_msgs = (flags & B3D_FLAG_SKIP_DEAD_LAST_MESSAGES) != 0;
  B3D_TRY(pack_all(pw, w, tr, (flags & B3D_FLAG_RUN_DEAD_KNN) != 0, stream, pack_side ? pack_side->s : stream, skip_last_msgs));

  // ---- the part that does not read the frozen encoders' outputs: edge / node encoder, layer 0's per-node table, the
  //      first k-NN block (x[0]); the encoders may still be running on other streams (b3d_clr_inputs::encoders_ready) ----
  {  // edge encoder: edge_attr.float() -> 4-16-32-64 (:123)
    ChainFwdArgs<LoadEdgeAttrF64, StoreAligned<4>> a;
    memset(&a, 0, sizeof(a));
    a.rows = E; a.in.ptr = in->edge_attr; a.out = StoreAligned<4>{w.e[0], nullptr, D::DE, 0};
    a.save_in = w.ea_pad; a.save[0] = w.ee_a1; a.save[1] = w.ee_a2; a.wpack = w.wp_ee;
    B3D_TRY(launch_rows<kNWEdge>(chain_fwd_kernel<SeqEE, 0x3u, LoadEdgeAttrF64, StoreAligned<4>, kNWEdge>, "edge_encoder", a, E, stream, B3D_K_OTHER, chain_lds<SeqEE>()));
  }
  {  // node encoder 19-48-96 (:174-176)
    ChainFwdArgs<LoadUnaligned<19>, StoreAligned<6>> a;
    memset(&a, 0, sizeof(a));
    a.rows = N; a.in.ptr = in->pose_feats; a.out = StoreAligned<6>{w.x[0], nullptr, D::DX, 0};
    a.save_in = w.pose_pad; a.save[0] = w.ne_a1; a.wpack = w.wp_ne;
    B3D_TRY(launch_rows<kNWNode>(chain_fwd_kernel<SeqNE, 0x1u, LoadUnaligned<19>, StoreAligned<6>, kNWNode>, "node_encoder", a, N, stream, B3D_K_OTHER, chain_lds<SeqNE>()));
  }
  // x0 terms of the future / past columns (once per forward) + the per-node table of layer 0
  NodeProj0Args a;
  a.N = N; a.x0 = w.x[0]; a.T0 = w.T0; a.T = w.T; a.wpack = w.wp_proj0;
  B3D_TRY((launch_node_split<DB, B3D_PROJ0_WAVES>(node_proj0_split_kernel<DB, B3D_PROJ0_WAVES>, "node_proj0", a, N, stream, B3D_K_OTHER)));
  Side* knn_side = nullptr;
  auto knn_block = [&](int l) -> int {                         // the discarded k-NN + GAT block on x[l] (:180-184)
    B3D_REQUIRE(in->node_timestamps != nullptr, "node_timestamps required with B3D_FLAG_RUN_DEAD_KNN");
    hipStream_t ks = stream;
    if (!(flags & B3D_FLAG_SINGLE_STREAM)) {
      if (!knn_side) B3D_TRY(side_get(0, &knn_side));
      B3D_TRY(side_fork(stream, knn_side));                // x[l] is complete on `stream` here
      ks = knn_side->s;
    }
    // on the launch stream the block reads GATConv.lin(x[l]) from the per-node table
    const bool pre = ks == stream;
    return knn_gat_block<D::DX>(w.knn, w.x[l], in->node_timestamps, N, pw->knn_conv, 20, ks, pre ? w.T + HC::OG : nullptr, HC::TW);
  };
  if (flags & B3D_FLAG_RUN_DEAD_KNN) B3D_TRY(knn_block(0));
  if (in->encoders_ready) B3D_HIP_CHECK(hipStreamWaitEvent(stream, (hipEvent_t)in->encoders_ready, 0));

  // ---- x_sens = x_img | x_lidar | x_radar; rows without a modality stay zero (:127-141,172) ------
  B3D_HIP_CHECK(hipMemsetAsync(w.xsens, 0, (size_t)N * XS * sizeof(float), stream));
  hipLaunchKernelGGL(copy_cols_kernel, dim3(((long)N * 96 + 255) / 256), dim3(256), 0, stream, in->x_img, 96, w.xsens, XS, 0, N, 96);
  B3D_TRY(launch_check("copy_cols_kernel"));
  if (nl > 0) {  // fc_lidar_encoder 256-192-128 on the LiDAR rows, scattered to their nodes
    using In = LoadAligned<16>;
    using Out = StoreAligned<8>;
    ChainFwdArgs<In, Out> a;
    memset(&a, 0, sizeof(a));
    a.rows = nl; a.in = In{in->pointnet_out, nullptr, 256, 0}; a.out = Out{w.xsens, in->lidar_nodes, XS, 96};
    a.save[0] = w.fl_a1; a.wpack = w.wp_fl;
    B3D_TRY(launch_rows<kNWHead>(chain_fwd_kernel<SeqFL, 0x1u, In, Out, kNWHead>, "fc_lidar_encoder", a, nl, stream, B3D_K_OTHER, chain_lds<SeqFL>()));
  }
  if (nr > 0) {  // fc_radar_encoder 256-192-128-64
    using In = LoadAligned<16>;
    using Out = StoreAligned<4>;
    ChainFwdArgs<In, Out> a;
    memset(&a, 0, sizeof(a));
    a.rows = nr; a.in = In{in->radarnet_out, nullptr, 256, 0}; a.out = Out{w.xsens, in->radar_nodes, XS, 224};
    a.save[0] = w.fr_a1; a.save[1] = w.fr_a2; a.wpack = w.wp_fr;
    B3D_TRY(launch_rows<kNWHead>(chain_fwd_kernel<SeqFR, 0x3u, In, Out, kNWHead>, "fc_radar_encoder", a, nr, stream, B3D_K_OTHER, chain_lds<SeqFR>()));
  }
  B3D_HIP_CHECK(hipMemcpyAsync(out_x_sens, w.xsens, (size_t)N * XS * sizeof(float), hipMemcpyDeviceToDevice, stream));

  // ---- cross-edge modality attention, hoisted to nodes: s = z_radar | z_lidar | z_img (:143-161) ----
  B3D_TRY(affine_fwd<96>(w, 0, N, 0, 192, stream));
  B3D_TRY(affine_fwd<128>(w, 1, N, 96, 64, stream));
  B3D_TRY(affine_fwd<64>(w, 2, N, 224, 0, stream));

  {  // att_edge_encoder( s[dst] | s[src] | e ) 640-512-384-256-128-64 (:161-164)
    B3D_TRY(node_linear<SeqAttU>("att_node_linear", w.s, XS, 0, w.U, 1024, N, w.wp_attU, stream, B3D_K_ATT_FWD));
    Att0FwdArgs fa;
    fa.E = E; fa.src = g->src; fa.dst = g->dst; fa.U = w.U; fa.e0 = w.e[0]; fa.A0 = w.A[0]; fa.wpack = w.wp_att0; fa.rmask = w.amask[0];
    B3D_TRY(launch_rows<kNWEdge>(att0_fwd_kernel<kNWEdge>, "att_edge_encoder.0", fa, E, stream, B3D_K_ATT_FWD, stream_lds_bytes<Att0Seq>()));
    B3D_TRY((wide<SeqAT1, true, true>("att_edge_encoder.2", LoadAligned<32>{w.A[0], nullptr, 512, 0}, E, w.A[1], 384, 0, nullptr, w.amask[1], w.wp_at[1], stream)));
    B3D_TRY((wide<SeqAT2, true, true>("att_edge_encoder.4", LoadAligned<24>{w.A[1], nullptr, 384, 0}, E, w.A[2], 256, 0, nullptr, w.amask[2], w.wp_at[2], stream)));
    B3D_TRY((wide<SeqAT3, true, true>("att_edge_encoder.6", LoadAligned<16>{w.A[2], nullptr, 256, 0}, E, w.A[3], 128, 0, nullptr, w.amask[3], w.wp_at[3], stream)));
    B3D_TRY((wide<SeqAT4, false, true>("att_edge_encoder.8", LoadAligned<8>{w.A[3], nullptr, 128, 0}, E, w.att, 64, 0, nullptr, nullptr, w.wp_at[4], stream)));
  }
  for (int l = 0; l < depth; ++l) {
    if ((flags & B3D_FLAG_RUN_DEAD_KNN) && l > 0 && (l % 2 == 0)) B3D_TRY(knn_block(l));
    NodeFwdArgs na;
    memset(&na, 0, sizeof(na));
    // the rows of `past` a node sums: the run tails the edge kernel left (edges grouped by destination), or its whole list
    const bool runs = past_runs_enabled() && g->past_ptr != nullptr && g->past_rows != nullptr && g->dst_unsorted != nullptr;
    na.N = N; na.dst_ptr = runs ? g->past_ptr : g->dst_ptr; na.dst_perm = runs ? g->past_rows : g->dst_perm;
    na.src_ptr = g->src_ptr; na.src_perm = g->src_perm;
    na.past = w.past; na.fut = w.fut; na.M = w.M[l]; na.x_out = w.x[l + 1]; na.sH1 = w.nH1[l]; na.sH2 = w.nH2[l];
    EdgeFwdHArgs ea;
    memset(&ea, 0, sizeof(ea));
    ea.E = E; ea.src = g->src; ea.dst = g->dst; ea.T = w.T; ea.e_in = w.e[l]; ea.a_in = w.att;
    ea.e_out = w.e[l + 1]; ea.fut = w.fut; ea.past = w.past;
    ea.dst_unsorted = runs ? g->dst_unsorted : nullptr; ea.past_dump0 = (unsigned)edge_rows(E);
    ea.sH1 = w.sH1[l]; ea.sH2 = w.sH2[l]; ea.sF1 = w.sF1[l]; ea.sP1 = w.sP1[l]; ea.wpack = w.wp_efwd2;
    ea.rmask = reinterpret_cast<unsigned*>(w.rmask[l]); ea.rmask2 = ea.rmask ? ea.rmask + edge_rows(E) * 16 : nullptr;
    if (skip_last_msgs && l + 1 == depth) {
      // the last layer's messages and node update feed nothing (clr_att_gnn.py:188): edge_update alone, no node launch
      ea.wpack = w.wp_efwd_nm2;
      if (tr) B3D_TRY(launch_es((es::edge_fwd_kernel<DB, true, false>), "edge_fwd_last", ea, E, stream, B3D_K_EDGE_FWD, ES::FwdNoMsg::LDS_BYTES));
      else B3D_TRY(launch_es((es::edge_fwd_kernel<DB, false, false>), "edge_fwd_last", ea, E, stream, B3D_K_EDGE_FWD, ES::FwdNoMsg::LDS_BYTES));
      continue;
    }
    if (tr) B3D_TRY(launch_es(es::edge_fwd_kernel<DB, true>, "edge_fwd", ea, E, stream, B3D_K_EDGE_FWD, ES::Fwd::LDS_BYTES));
    else B3D_TRY(launch_es(es::edge_fwd_kernel<DB, false>, "edge_fwd", ea, E, stream, B3D_K_EDGE_FWD, ES::Fwd::LDS_BYTES));
    if (l + 1 < depth) {                                   // + the per-node table the next layer's edge phase gathers
      na.wpack = w.wp_nfwd_h; na.T = w.T; na.T0 = w.T0;
      B3D_TRY((launch_node_split<DB, kNodeWavesWide>(mp_node_fwd_split_h_kernel<DB>, "mp_node_fwd", na, N, stream, B3D_K_NODE_FWD)));
    } else {
      na.wpack = w.wp_nfwd;
      B3D_TRY((launch_node_split<D, kNodeWavesWide>(mp_node_fwd_split_kernel<D, kNodeWavesWide>, "mp_node_fwd", na, N, stream, B3D_K_NODE_FWD)));
    }
  }
  {  // edge classifier 64-32-16-8-1 + Sigmoid (:49-58,188)
    ChainFwdArgs<LoadAligned<4>, StoreScalar> a;
    memset(&a, 0, sizeof(a));
    a.rows = E; a.in = LoadAligned<4>{w.e[depth], nullptr, D::DE, 0}; a.out = StoreScalar{w.prob, 1};
    a.save[0] = w.c_a1; a.save[1] = w.c_a2; a.save[2] = w.c_a3; a.wpack = w.wp_cls;
    B3D_TRY(launch_rows<kNWEdge>(chain_fwd_kernel<SeqCls, 0x7u, LoadAligned<4>, StoreScalar, kNWEdge>, "edge_classifier", a, E, stream, B3D_K_OTHER, chain_lds<SeqCls>()));
    B3D_HIP_CHECK(hipMemcpyAsync(out_prob, w.prob, (size_t)E * sizeof(float), hipMemcpyDeviceToDevice, stream));
  }
  if (knn_side && !((flags & B3D_FLAG_DEFER_SIDE_JOIN) && (flags & B3D_FLAG_TRAINING))) B3D_TRY(side_join(knn_side, stream));
  if (pack_side) B3D_TRY(side_join(pack_side, stream));     // the backward images are complete before the caller can enqueue the sweep
  return B3D_OK;
}

extern "C" int b3d_clr_backward(const b3d_clr_weights* pw, const b3d_graph* g, const b3d_clr_inputs* in, int32_t depth,
                                void* workspace, size_t workspace_bytes, const float* d_prob, const float* d_x_sens,
                                const b3d_clr_grads* gr, b3d_stream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  B3D_TRY(check_weights(pw));
  B3D_REQUIRE(g && in && workspace && gr, "b3d_clr_backward: null argument");
  B3D_REQUIRE(depth >= 1 && depth <= 15, "b3d_clr_backward: depth %d outside [1,15]", depth);
  const int N = g->N, E = g->E, nl = in->n_lidar, nr = in->n_radar;
  B3D_REQUIRE(N > 0 && E > 0, "b3d_clr_backward: empty graph");
  Ws w;
  carve(w, workspace, workspace_bytes, N, E, nl, nr, depth, B3D_FLAG_TRAINING);
  if (!w.ok) return fail(B3D_ERR_WORKSPACE, "b3d_clr_backward: workspace %zu < %zu bytes", workspace_bytes, w.bytes);
  const int* src = g->src;
  const int* dst = g->dst;
  const size_t EP = edge_rows(E);           // per-edge buffers are padded to whole 64-row tiles (carve)
  const size_t eL1 = EP * D::EH1, eL2 = EP * D::EH2, eLe = EP * D::DE, eLm = EP * D::MH;
  const size_t nLm = (size_t)N * D::NIN, nLx = (size_t)N * D::DX, nL1 = (size_t)N * D::NH1, nL2 = (size_t)N * D::NH2;
  WgArgs smallE, smallN;           // LDS-staged weight gradients of the narrow / unaligned layers
  smallE.njobs = smallN.njobs = 0;

  // ---- classifier (with the sigmoid derivative) -> d e[depth] --------------------------------------
  int cur = 0;
  {
    ChainBwdArgs<LoadSigmoidGrad, StoreAligned<4>> a;
    memset(&a, 0, sizeof(a));
    a.rows = E; a.in = LoadSigmoidGrad{d_prob, w.prob}; a.out = StoreAligned<4>{w.de[cur], nullptr, D::DE, 0};
    a.gtop = w.gc_top;
    a.act[0] = w.c_a3; a.act[1] = w.c_a2; a.act[2] = w.c_a1; a.act[3] = nullptr;
    a.gsave[0] = w.gc3; a.gsave[1] = w.gc2; a.gsave[2] = w.gc1; a.gsave[3] = nullptr;
    a.wpack = w.wp_clsT;
    B3D_TRY(launch_rows<kNWEdge>(chain_bwd_kernel<SeqClsT, LoadSigmoidGrad, StoreAligned<4>, kNWEdge>, "edge_classifier_bwd", a, E, stream, B3D_K_OTHER, chain_lds<SeqClsT>()));
    WgJob j3 = make_job(w.lin[C3], E, seg(w.gc_top, nullptr, 16, 0, 1)); add_act(j3, seg(w.c_a3, nullptr, 16, 0, 8)); smallE.jobs[smallE.njobs++] = j3;
    WgJob j2 = make_job(w.lin[C2], E, seg(w.gc3, nullptr, 16, 0, 8)); add_act(j2, seg(w.c_a2, nullptr, 16, 0, 16)); smallE.jobs[smallE.njobs++] = j2;
    WgJob j1 = make_job(w.lin[C1], E, seg(w.gc2, nullptr, 16, 0, 16)); add_act(j1, seg(w.c_a1, nullptr, 32, 0, 32)); smallE.jobs[smallE.njobs++] = j1;
    WgJob j0 = make_job(w.lin[C0], E, seg(w.gc1, nullptr, 32, 0, 32)); add_act(j0, seg(w.e[depth], nullptr, D::DE, 0, D::DE)); smallE.jobs[smallE.njobs++] = j0;
  }

  // ---- message-passing layers, last to first (data gradients; G tensors kept per layer) -------------
  bool dx0_first = true, da_first = true;
  // hoisted first layers: per-node gradient of T from layer `lay`'s first-layer gradients (kept per layer: weight gradient)
  auto listsum = [&](int lay) -> int {
    NodeGradProjArgs ga;
    memset(&ga, 0, sizeof(ga));
    ga.N = N; ga.dst_ptr = g->dst_ptr; ga.dst_perm = g->dst_perm; ga.src_ptr = g->src_ptr; ga.src_perm = g->src_perm;
    ga.GdH1 = w.GdH1 + lay * eL1;
    ga.GdF1 = (lay < depth - 1) ? w.GdF1 + lay * eLm : nullptr;
    ga.GdP1 = (lay < depth - 1) ? w.GdP1 + lay * eLm : nullptr;
    ga.dT = w.dT + (size_t)lay * N * HC::GW;
    const long tasks = (long)((N + 15) / 16) * ListSumGeom<DB>::TASKS;
    ProfScope ps(B3D_K_NODE_BWD, stream);
    hipLaunchKernelGGL(node_listsum_kernel<DB>, dim3((unsigned)((tasks + 3) / 4)), dim3(256), 0, stream, ga);
    return launch_check("node_listsum_kernel");
  };
  for (int l = depth - 1; l >= 0; --l) {
    const bool msgs = (l < depth - 1);
    if (msgs) {
      B3D_TRY(listsum(l + 1));
      NodeBwdGArgs nb;
      memset(&nb, 0, sizeof(nb));
      nb.N = N; nb.dT = w.dT + (size_t)(l + 1) * N * HC::GW;
      nb.dx0_acc = w.dx0_acc; nb.dx0_first = dx0_first ? 1 : 0;
      nb.sH1 = w.nH1[l]; nb.sH2 = w.nH2[l];
      nb.dM = w.dM + l * nLm; nb.Gdx = w.Gdx + l * nLx; nb.GdH2 = w.GnH2 + l * nL2; nb.GdH1 = w.GnH1 + l * nL1;
      nb.wpack = w.wp_nbwd_h;
      B3D_TRY(set_lds(node_bwd_g_kernel<DB, true>, NodeBwdGLds<DB>::BYTES));
      {
        ProfScope ps(B3D_K_NODE_BWD, stream);
        hipLaunchKernelGGL((node_bwd_g_kernel<DB, true>), dim3((N + 15) / 16), dim3(kNodeBwdGWaves * 64), NodeBwdGLds<DB>::BYTES, stream, nb);
      }
      B3D_TRY(launch_check("node_bwd_g_kernel"));
      dx0_first = false;
    }
    EdgeBwdHArgs eb;
    memset(&eb, 0, sizeof(eb));
    eb.E = E; eb.src = src; eb.dst = dst;
    eb.dM = msgs ? w.dM + l * nLm : nullptr;
    eb.de_out = w.de[cur]; eb.de_in = w.de[cur ^ 1];
    eb.sH1 = w.sH1[l]; eb.sH2 = w.sH2[l]; eb.sF1 = w.sF1[l]; eb.sP1 = w.sP1[l];
    eb.rmask = reinterpret_cast<const unsigned*>(w.rmask[l]); eb.rmask2 = eb.rmask + edge_rows(E) * 16;
    eb.da_acc = w.da_acc; eb.da_first = da_first ? 1 : 0;
    eb.GdH1 = w.GdH1 + l * eL1; eb.GdH2 = w.GdH2 + l * eL2; eb.Gde = w.Gde + l * eLe;
    eb.GdF1 = w.GdF1 + l * eLm; eb.GdP1 = w.GdP1 + l * eLm;
    if (msgs) {
      eb.wpack = w.wp_ebwd2;
      B3D_TRY(launch_es(es::edge_bwd_kernel<DB, true>, "edge_bwd", eb, E, stream, B3D_K_EDGE_BWD, ES::Bwd::LDS_BYTES));
    } else {                                                 // last layer: its node output is not used (only e feeds the classifier)
      eb.wpack = w.wp_ebwd_nm2;
      B3D_TRY(launch_es(es::edge_bwd_kernel<DB, false>, "edge_bwd_last", eb, E, stream, B3D_K_EDGE_BWD, ES::BwdNoMsg::LDS_BYTES));
    }
    da_first = false;
    cur ^= 1;
  }
  // layer 0's (dx | dx0) for the node encoder
  B3D_TRY(listsum(0));
  NodeBwdGArgs nb;
  memset(&nb, 0, sizeof(nb));
  nb.N = N; nb.dT = w.dT; nb.gx = w.gx; nb.wpack = w.wp_gproj;
  B3D_TRY(set_lds(node_bwd_g_kernel<DB, false>, NodeBwdGLds<DB>::BYTES));
  {
    ProfScope ps(B3D_K_NODE_BWD, stream);
    hipLaunchKernelGGL((node_bwd_g_kernel<DB, false>), dim3((N + 15) / 16), dim3(kNodeBwdGWaves * 64), NodeBwdGLds<DB>::BYTES, stream, nb);
  }
  B3D_TRY(launch_check("node_bwd_g_kernel"));

  // ---- att_edge_encoder backward: d att (summed over the layers) -> d(s[dst] | s[src] | e) ------------
  B3D_TRY((wide<SeqAT4T, false, false>("att_edge_encoder.8^T", LoadAligned<4>{w.da_acc, nullptr, 64, 0}, E, w.dA[0], 128, 0, w.amask[3], nullptr, w.wp_atT[4], stream, B3D_K_ATT_BWD)));
  B3D_TRY((wide<SeqAT3T, false, false>("att_edge_encoder.6^T", LoadAligned<8>{w.dA[0], nullptr, 128, 0}, E, w.dA[1], 256, 0, w.amask[2], nullptr, w.wp_atT[3], stream, B3D_K_ATT_BWD)));
  B3D_TRY((wide<SeqAT2T, false, false>("att_edge_encoder.4^T", LoadAligned<16>{w.dA[1], nullptr, 256, 0}, E, w.dA[2], 384, 0, w.amask[1], nullptr, w.wp_atT[2], stream, B3D_K_ATT_BWD)));
  B3D_TRY((wide<SeqAT1T, false, false>("att_edge_encoder.2^T", LoadAligned<24>{w.dA[2], nullptr, 384, 0}, E, w.dA[3], 512, 0, w.amask[0], nullptr, w.wp_atT[1], stream, B3D_K_ATT_BWD)));
  // d e0 per edge; d U per node = sums of d A0 over the CSR / CSC lists; d s = W0[:, 0:288]^T dU_i + W0[:, 288:576]^T dU_j
  B3D_TRY((wide<SeqAT0eT, false, false>("att_edge_encoder.0[e]^T", LoadAligned<32>{w.dA[3], nullptr, 512, 0}, E, w.de0, 64, 0, nullptr, nullptr, w.wp_at0eT, stream, B3D_K_ATT_BWD)));
  AttListSumArgs la;
  la.N = N; la.W = 512; la.dst_ptr = g->dst_ptr; la.dst_perm = g->dst_perm; la.src_ptr = g->src_ptr; la.src_perm = g->src_perm;
  la.G = w.dA[3]; la.dU = w.dU;
  {
    const long tasks = (long)((N + 15) / 16) * 2 * (512 / 64);
    ProfScope ps(B3D_K_ATT_BWD, stream);
    hipLaunchKernelGGL(att_listsum_kernel, dim3((unsigned)((tasks + 3) / 4)), dim3(256), 0, stream, la);
  }
  B3D_TRY(launch_check("att_listsum_kernel"));
  B3D_TRY(node_linear<SeqAttDs>("att_node_linear^T", w.dU, 1024, 0, w.ds, XS, N, w.wp_attDs, stream, B3D_K_ATT_BWD));
  B3D_TRY(affine_bwd<96>(w, 0, N, 0, 192, stream));
  B3D_TRY(affine_bwd<128>(w, 1, N, 96, 64, stream));
  B3D_TRY(affine_bwd<64>(w, 2, N, 224, 0, stream));
  {
    const int affd[3] = {96, 128, 64}, xc[3] = {0, 96, 224};
    for (int m = 0; m < 3; ++m) {
      WgJob jo = make_job(w.lin[AOC + 2 * m], N, seg(w.gaff_top[m], nullptr, affd[m], 0, affd[m]));
      add_act(jo, seg(w.aff_v[m], nullptr, affd[m], 0, affd[m]));
      smallN.jobs[smallN.njobs++] = jo;
      WgJob jv = make_job(w.lin[AVC + 2 * m], N, seg(w.gaff_v[m], nullptr, affd[m], 0, affd[m]));
      add_act(jv, seg(w.xsens, nullptr, XS, xc[m], affd[m]));
      smallN.jobs[smallN.njobs++] = jv;
    }
  }
  if (nl > 0) {  // fc_lidar_encoder: gradient of x_lidar = attention path + upstream d x_sens
    using In = LoadAdd2<8>;
    ChainBwdArgs<In, StoreNone> a;
    memset(&a, 0, sizeof(a));
    a.rows = nl; a.in = In{w.dxs, XS, 96, d_x_sens, XS, 96, in->lidar_nodes};
    a.gtop = w.gfl_top; a.act[0] = w.fl_a1; a.gsave[0] = w.gfl1; a.wpack = w.wp_flT;
    B3D_TRY(launch_rows<kNWHead>(chain_bwd_kernel<SeqFLT, In, StoreNone, kNWHead>, "fc_lidar_encoder_bwd", a, nl, stream, B3D_K_OTHER, chain_lds<SeqFLT>()));

  }
  if (nr > 0) {  // fc_radar_encoder
    using In = LoadAdd2<4>;
    ChainBwdArgs<In, StoreNone> a;
    memset(&a, 0, sizeof(a));
    a.rows = nr; a.in = In{w.dxs, XS, 224, d_x_sens, XS, 224, in->radar_nodes};
    a.gtop = w.gfr_top; a.act[0] = w.fr_a2; a.act[1] = w.fr_a1; a.gsave[0] = w.gfr2; a.gsave[1] = w.gfr1; a.wpack = w.wp_frT;
    B3D_TRY(launch_rows<kNWHead>(chain_bwd_kernel<SeqFRT, In, StoreNone, kNWHead>, "fc_radar_encoder_bwd", a, nr, stream, B3D_K_OTHER, chain_lds<SeqFRT>()));

  }

  // ---- encoders ----------------------------------------------------------------------------------------
  // node encoder (x = initial_x): running d initial_x + layer 0's (dx | dx0)
  using In = LoadNodeEncGradH<6>;
  ChainBwdArgs<In, StoreNone> a;
  memset(&a, 0, sizeof(a));
  a.rows = N;
  a.in = In{nullptr, dx0_first ? nullptr : w.dx0_acc, w.gx};
  a.gtop = w.gn_top; a.act[0] = w.ne_a1; a.gsave[0] = w.gn1; a.wpack = w.wp_neT;
  B3D_TRY(launch_rows<kNWNode>(chain_bwd_kernel<SeqNET, In, StoreNone, kNWNode>, "node_encoder_bwd", a, N, stream, B3D_K_OTHER, chain_lds<SeqNET>()));
  WgJob n1 = make_job(w.lin[NE1], N, seg(w.gn_top, nullptr, 96, 0, 96)); add_act(n1, seg(w.ne_a1, nullptr, 48, 0, 48)); smallN.jobs[smallN.njobs++] = n1;
  WgJob n0 = make_job(w.lin[NE0], N, seg(w.gn1, nullptr, 48, 0, 48)); add_act(n0, seg(w.pose_pad, nullptr, 32, 0, 19)); smallN.jobs[smallN.njobs++] = n0;
  {  // edge encoder: e[0] feeds layer 0 AND att_edge_encoder (columns 576:640 of its input)
    using In = LoadAdd2<4>;
    ChainBwdArgs<In, StoreNone> a;
    memset(&a, 0, sizeof(a));
    a.rows = E;
    a.in = In{w.de[cur], D::DE, 0, w.de0, 64, 0, nullptr};
    a.gtop = w.ge_top; a.act[0] = w.ee_a2; a.act[1] = w.ee_a1; a.gsave[0] = w.ge2; a.gsave[1] = w.ge1; a.wpack = w.wp_eeT;
    B3D_TRY(launch_rows<kNWEdge>(chain_bwd_kernel<SeqEET, In, StoreNone, kNWEdge>, "edge_encoder_bwd", a, E, stream, B3D_K_OTHER, chain_lds<SeqEET>()));
    WgJob e2 = make_job(w.lin[EE2], E, seg(w.ge_top, nullptr, 64, 0, 64)); add_act(e2, seg(w.ee_a2, nullptr, 32, 0, 32)); smallE.jobs[smallE.njobs++] = e2;
    WgJob e1 = make_job(w.lin[EE1], E, seg(w.ge2, nullptr, 32, 0, 32)); add_act(e1, seg(w.ee_a1, nullptr, 16, 0, 16)); smallE.jobs[smallE.njobs++] = e1;
    WgJob e0 = make_job(w.lin[EE0], E, seg(w.ge1, nullptr, 16, 0, 16)); add_act(e0, seg(w.ea_pad, nullptr, 16, 0, 4)); smallE.jobs[smallE.njobs++] = e0;
  }
  B3D_TRY((launch_wgrad<8, 1>(smallE, stream, B3D_K_WGRAD_OTHER)));
  B3D_TRY((launch_wgrad<8, 1>(smallN, stream, B3D_K_WGRAD_OTHER)));

  // ---- streamed weight gradients: message-passing stacks (all layers) + att_edge_encoder -------------
  {
    hipLaunchKernelGGL(iota_kernel, dim3(((E > N ? E : N) + 255) / 256), dim3(256), 0, stream, w.iota, E > N ? E : N);
    B3D_TRY(launch_check("iota_kernel"));
    B3D_HIP_CHECK(hipMemsetAsync(w.zrow, 0, 256 * sizeof(float), stream));
    WgBuilder wb;
    wb.begin(w.ws_table, kTableCap, w.ws_task_job, kTaskCap, w.iota, stream);
    auto add_block = [&](LinSlab& ls, long rows, int nvar, int rpt, const float* gp, const int* gidx, long gvs, int gstride, int gcol0,
                         const Col* cols, int ncols, bool with_bias) {
      wb.add_block(ls, rows, nvar, rpt, gp, gidx, gvs, gstride, gcol0, cols, ncols, with_bias);
    };
    auto add_matrix = [&](int lin, long rows, int nvar, int rpt, const float* gp, const int* gidx, long gvs, int gstride, int gcol0,
                          const Col* cols, int ncols) {
      wb.add_block(w.lin[lin], rows, nvar, rpt, gp, gidx, gvs, gstride, gcol0, cols, ncols, true);
    };
    const int rp = kStreamRowsPerTask, rpa = kStreamRowsPerTaskAtt;
    // First layers: per-edge columns contract over edges, node columns over NODES (G = column blocks of dT).
    const int rn = kStreamNodeRowsPerTask;
    const long tLs = (long)N * HC::GW;
    {  // edge_update.0 [256, 320]: x[dst] 0:96 | x[src] 96:192 | e 192:256 | att 256:320
      wb.add_eu0_edge_columns(w.vlin[VL_EU0E], E, depth, rp, w.GdH1, eL1, D::EH1, w.e[0], (long)eLe, D::DE, w.att);
      Col cx[1] = {{w.x[0], nullptr, (long)nLx, D::DX, 0, 96}};
      add_block(w.vlin[VL_EU0XI], N, depth, rn, w.dT, nullptr, tLs, HC::GW, HC::OA, cx, 1, false);
      add_block(w.vlin[VL_EU0XJ], N, depth, rn, w.dT, nullptr, tLs, HC::GW, HC::OB, cx, 1, false);
      Col c1[1] = {{w.sH1[0], nullptr, (long)eL1, D::EH1, 0, 256}};
      add_matrix(EU1, E, depth, rp, w.GdH2, nullptr, eL2, D::EH2, 0, c1, 1);
      Col c2[1] = {{w.sH2[0], nullptr, (long)eL2, D::EH2, 0, 128}};
      add_matrix(EU2, E, depth, rp, w.Gde, nullptr, eLe, D::DE, 0, c2, 1);
    }
    {  // message stacks .0 [192, 256] (layers 0 .. depth-2): x[.] 0:96 | e' 96:160 | x0[.] 160:256
      Col ce[1] = {{w.e[1], nullptr, (long)eLe, D::DE, 0, 64}};
      Col cx[1] = {{w.x[0], nullptr, (long)nLx, D::DX, 0, 96}};
      Col c0[1] = {{w.x[0], nullptr, 0, D::DX, 0, 96}};
      add_block(w.vlin[VL_PA0E], E, depth - 1, rp, w.GdP1, nullptr, eLm, D::MH, 0, ce, 1, true);
      add_block(w.vlin[VL_PA0X], N, depth - 1, rn, w.dT, nullptr, tLs, HC::GW, HC::OP, cx, 1, false);
      add_block(w.vlin[VL_PA0X0], N, depth - 1, rn, w.dT, nullptr, tLs, HC::GW, HC::OP, c0, 1, false);
      add_block(w.vlin[VL_FU0E], E, depth - 1, rp, w.GdF1, nullptr, eLm, D::MH, 0, ce, 1, true);
      add_block(w.vlin[VL_FU0X], N, depth - 1, rn, w.dT, nullptr, tLs, HC::GW, HC::OF, cx, 1, false);
      add_block(w.vlin[VL_FU0X0], N, depth - 1, rn, w.dT, nullptr, tLs, HC::GW, HC::OF, c0, 1, false);
      Col cp1[1] = {{w.sP1[0], nullptr, (long)eLm, D::MH, 0, 192}};
      add_matrix(PA1, E, depth - 1, rp, w.dM, dst, nLm, D::NIN, 0, cp1, 1);
      Col cf1[1] = {{w.sF1[0], nullptr, (long)eLm, D::MH, 0, 192}};
      add_matrix(FU1, E, depth - 1, rp, w.dM, src, nLm, D::NIN, D::DM, cf1, 1);
    }
    // modality heads on the rows that carry the modality (the LDS-staged kernel spilled ~3,000 VGPRs at these widths)
    const int rf = kStreamRowsPerTaskFc;
    if (nl > 0) {
      Col a1[1] = {{w.fl_a1, nullptr, 0, 192, 0, 192}};
      add_matrix(FL1, nl, 1, rf, w.gfl_top, nullptr, 0, 128, 0, a1, 1);
      Col a0[1] = {{in->pointnet_out, nullptr, 0, 256, 0, 256}};
      add_matrix(FL0, nl, 1, rf, w.gfl1, nullptr, 0, 192, 0, a0, 1);
    }
    if (nr > 0) {
      Col a2[1] = {{w.fr_a2, nullptr, 0, 128, 0, 128}};
      add_matrix(FR2, nr, 1, rf, w.gfr_top, nullptr, 0, 64, 0, a2, 1);
      Col a1[1] = {{w.fr_a1, nullptr, 0, 192, 0, 192}};
      add_matrix(FR1, nr, 1, rf, w.gfr2, nullptr, 0, 128, 0, a1, 1);
      Col a0[1] = {{in->radarnet_out, nullptr, 0, 256, 0, 256}};
      add_matrix(FR0, nr, 1, rf, w.gfr1, nullptr, 0, 192, 0, a0, 1);
    }
    {  // node update (layers 0 .. depth-2)
      Col c0[1] = {{w.M[0], nullptr, (long)nLm, D::NIN, 0, 256}};
      add_matrix(CF0, N, depth - 1, rp, w.GnH1, nullptr, nL1, D::NH1, 0, c0, 1);
      Col c1[1] = {{w.nH1[0], nullptr, (long)nL1, D::NH1, 0, 192}};
      add_matrix(CF1, N, depth - 1, rp, w.GnH2, nullptr, nL2, D::NH2, 0, c1, 1);
      Col c2[1] = {{w.nH2[0], nullptr, (long)nL2, D::NH2, 0, 128}};
      add_matrix(CF2, N, depth - 1, rp, w.Gdx, nullptr, nLx, D::DX, 0, c2, 1);
    }
    {  // att_edge_encoder
      // .0: edge columns over edges, node columns over nodes (G = the per-node sums dU_i | dU_j)
      Col ce[1] = {{w.e[0], nullptr, 0, D::DE, 0, 64}};
      add_block(w.vlin[VL_AT0E], E, 1, rpa, w.dA[3], nullptr, 0, 512, 0, ce, 1, true);
      Col cs[1] = {{w.s, nullptr, 0, XS, 0, 288}};
      add_block(w.vlin[VL_AT0I], N, 1, kStreamNodeRowsPerTaskAtt, w.dU, nullptr, 0, 1024, 0, cs, 1, false);
      add_block(w.vlin[VL_AT0J], N, 1, kStreamNodeRowsPerTaskAtt, w.dU, nullptr, 0, 1024, 512, cs, 1, false);
      Col c1[1] = {{w.A[0], nullptr, 0, 512, 0, 512}};
      add_matrix(AT1, E, 1, rpa, w.dA[2], nullptr, 0, 384, 0, c1, 1);
      Col c2[1] = {{w.A[1], nullptr, 0, 384, 0, 384}};
      add_matrix(AT2, E, 1, rpa, w.dA[1], nullptr, 0, 256, 0, c2, 1);
      Col c3[1] = {{w.A[2], nullptr, 0, 256, 0, 256}};
      add_matrix(AT3, E, 1, rpa, w.dA[0], nullptr, 0, 128, 0, c3, 1);
      Col c4[1] = {{w.A[3], nullptr, 0, 128, 0, 128}};
      add_matrix(AT4, E, 1, rpa, w.da_acc, nullptr, 0, 64, 0, c4, 1);
    }
    B3D_TRY(wb.launch(w.zrow, stream));
  }

  // ---- slabs -> parameter gradients -----------------------------------------------------------------------
  {
    float* dw[LIN_COUNT];
    float* db[LIN_COUNT];
    auto put = [&](int first, const b3d_linear_grad* a, int n) { for (int i = 0; i < n; ++i) { dw[first + i] = a[i].w; db[first + i] = a[i].b; } };
    put(EE0, gr->edge_encoder, 3); put(NE0, gr->node_encoder, 2); put(C0, gr->edge_classifier, 4);
    put(FL0, gr->fc_lidar_encoder, 2); put(FR0, gr->fc_radar_encoder, 3); put(AT0, gr->att_edge_encoder, 5);
    put(EU0, gr->mp.edge_update, 3); put(PA0, gr->mp.create_past_msgs, 2); put(FU0, gr->mp.create_future_msgs, 2);
    put(CF0, gr->mp.combine_future_past, 3);
    const b3d_mha_grad* mg[3] = {&gr->c2c_att, &gr->l2l_att, &gr->r2r_att};
    const int dd[3] = {96, 128, 64};
    ZeroArgs zz;
    zz.count = 0;
    for (int m = 0; m < 3; ++m) {
      // q / k thirds of in_proj receive no gradient (softmax over one key is constant)
      zero_add(zz, mg[m]->in_proj_weight, (size_t)3 * dd[m] * dd[m]);
      zero_add(zz, mg[m]->in_proj_bias, (size_t)3 * dd[m]);
      dw[AVC + 2 * m] = mg[m]->in_proj_weight ? mg[m]->in_proj_weight + (size_t)2 * dd[m] * dd[m] : nullptr;
      db[AVC + 2 * m] = mg[m]->in_proj_bias ? mg[m]->in_proj_bias + 2 * dd[m] : nullptr;
      dw[AOC + 2 * m] = mg[m]->out_proj_weight;
      db[AOC + 2 * m] = mg[m]->out_proj_bias;
    }
    B3D_TRY(zero_launch(zz, stream));            // in front of every reduction: they write the v third of in_proj
    RedArgs ra;
    ra.nentries = 0;
    for (int i = 0; i < LIN_COUNT; ++i) {
      LinSlab& ls = w.lin[i];
      if (i == EU0 || i == FU0 || i == PA0 || i == AT0) {
        bool any = false;
        for (int v = 0; v < VL_COUNT; ++v) any = any || (kVl[v].lin == i && w.vlin[v].used);
        if (any) {          // depth == 1: the message stacks receive no gradient (falls through to the zero fill)
          for (int v = 0; v < VL_COUNT; ++v) {
            if (kVl[v].lin != i) continue;
            RedEntry e = red_entry(w.vlin[v], dw[i] ? dw[i] + kVl[v].col0 : nullptr, kVl[v].bias ? db[i] : nullptr);
            e.ld = ls.K;
            ra.e[ra.nentries++] = e;
            if (ra.nentries == kRedMaxEntries) { B3D_TRY(launch_reduce(ra, stream)); ra.nentries = 0; }
          }
          continue;
        }
      }
      if (!ls.used) {
        if (zz.count + 2 > kZeroMax) B3D_TRY(zero_launch(zz, stream));
        zero_add(zz, dw[i], (size_t)ls.N * ls.K);
        zero_add(zz, db[i], (size_t)ls.N);
        continue;
      }
      ra.e[ra.nentries++] = red_entry(ls, dw[i], db[i]);
      if (ra.nentries == kRedMaxEntries) { B3D_TRY(launch_reduce(ra, stream)); ra.nentries = 0; }
    }
    B3D_TRY(zero_launch(zz, stream));            // gradients of layers that took no part (disjoint from the reductions)
    B3D_TRY(launch_reduce(ra, stream));
  }
  B3D_TRY(b3d_side_join(stream_));
  return B3D_OK;
}

// ---- standalone CausalMessagePassing layer, camera+LiDAR+radar widths (clr_att_gnn.py:227-356) --------------
// forward(x, edge_index, edge_attr, initial_x, att_edge_attr) -> (x', e') and its backward, as an operator of its own.  Since round 4
// it runs the model's plan: the node columns of the three first layers evaluated per NODE (a per-node table T from the caller's
// x / initial_x: node_proj0_split_kernel), the fragment-streamed edge kernels (b3d_edge2.hpp), per-node gradient sums
// (node_listsum + node_bwd_g) and the cooperative weight gradient -- the round-1 unsplit kernels (75 / 138 spilled registers at
// these widths) are gone.  Caller tensors have exactly E rows; the edge kernels store whole tiles, so their outputs go through
// padded workspace buffers and the E valid rows are copied out.
namespace b3d {
namespace clr {
constexpr int kLayerTableCap = 160, kLayerTaskCap = 16384;
enum { LL_EU0, LL_EU1, LL_EU2, LL_PA0, LL_PA1, LL_FU0, LL_FU1, LL_CF0, LL_CF1, LL_CF2, LL_COUNT };
static const int kLayerLin[LL_COUNT] = {EU0, EU1, EU2, PA0, PA1, FU0, FU1, CF0, CF1, CF2};
constexpr int kLayerVl = 9;                        // the first nine column blocks of kVl: edge_update.0, create_*_msgs.0
struct ClrLayerWs {
  float *wp_proj0, *wp_efwd2, *wp_nfwd, *wp_ebwd2, *wp_gproj, *wp_nbwd;
  float *T, *T0, *e_out, *fut, *past;
  float *sH1, *sH2, *sF1, *sP1, *M, *nH1, *nH2;
  float* rmask;
  float *dM, *GdH1, *GdH2, *Gde, *GdF1, *GdP1, *GnH2, *GnH1, *dT, *gx, *de_in, *da, *zero_e, *zero_n;
  float* zrow;
  int* iota;
  WsJob* ws_table;
  int* ws_task_job;
  LinSlab lin[LL_COUNT];
  LinSlab vlin[kLayerVl];
  size_t bytes;
  bool ok;
};
static void carve_layer(ClrLayerWs& w, void* ws, size_t ws_bytes, int N, int E, bool tr) {
  Carver c(ws, ws_bytes);
  memset(&w, 0, sizeof(w));
  const size_t e_ = edge_rows(E), n_ = (size_t)(N > 0 ? N : 1);
  w.wp_proj0 = c.take<float>(Proj0Seq2<DB>::TOTAL_FLOATS);
  w.wp_efwd2 = c.take<float>(ES::Fwd::TOTAL_FLOATS);
  w.wp_nfwd = c.take<float>(D::NodeFwdSeq::TOTAL_FLOATS);
  w.T = c.take<float>(n_ * HC::TW);
  w.T0 = c.take<float>(n_ * 2 * DB::MH);
  w.e_out = c.take<float>(e_ * D::DE);
  w.fut = c.take<float>(e_ * D::DM);
  w.past = c.take<float>((e_ + es::kPastDumpRows) * D::DM);      // + the dump rows of the in-wave per-destination sums (b3d_edge2.hpp)
  if (tr) {
    w.wp_ebwd2 = c.take<float>(ES::Bwd::TOTAL_FLOATS);
    w.wp_gproj = c.take<float>(HC::GradProjSeq::TOTAL_FLOATS);
    w.wp_nbwd = c.take<float>(D::NodeBwdSeq::TOTAL_FLOATS);
    w.sH1 = c.take<float>(e_ * D::EH1); w.sH2 = c.take<float>(e_ * D::EH2); w.sF1 = c.take<float>(e_ * D::MH); w.sP1 = c.take<float>(e_ * D::MH);
    w.rmask = c.take<float>(e_ * es::kMaskFloatsPerRow);
    w.M = c.take<float>(n_ * D::NIN); w.nH1 = c.take<float>(n_ * D::NH1); w.nH2 = c.take<float>(n_ * D::NH2);
    w.dM = c.take<float>(n_ * D::NIN);
    w.GdH1 = c.take<float>(e_ * D::EH1); w.GdH2 = c.take<float>(e_ * D::EH2); w.Gde = c.take<float>(e_ * D::DE);
    w.GdF1 = c.take<float>(e_ * D::MH); w.GdP1 = c.take<float>(e_ * D::MH);
    w.GnH2 = c.take<float>(n_ * D::NH2); w.GnH1 = c.take<float>(n_ * D::NH1);
    w.dT = c.take<float>(n_ * HC::GW); w.gx = c.take<float>(n_ * 2 * D::DX);
    w.de_in = c.take<float>(e_ * D::DE); w.da = c.take<float>(e_ * D::DA);
    w.zero_e = c.take<float>(e_ * D::DE); w.zero_n = c.take<float>(n_ * D::DX);
    w.zrow = c.take<float>(256);
    w.iota = c.take<int>((size_t)(E > N ? E : N) + 64);
    w.ws_table = c.take<WsJob>(kLayerTableCap);
    w.ws_task_job = c.take<int>(kLayerTaskCap);
    for (int i = 0; i < LL_COUNT; ++i) {
      LinSlab& ls = w.lin[i];
      const int li = kLayerLin[i];
      ls.N = kDims[li].N; ls.K = kDims[li].K; ls.NP = pad16(ls.N); ls.KP = pad16(ls.K);
      const long rows = kRowKind[li] == 0 ? E : N;
      ls.nchunks = (int)((rows + kStreamRowsPerTask - 1) / kStreamRowsPerTask);
      if (ls.nchunks < 1) ls.nchunks = 1;
      ls.slab = c.take<float>(wg_slab_floats(ls.nchunks, ls.NP, ls.KP));
      ls.used = false;
    }
    for (int v = 0; v < kLayerVl; ++v) {
      LinSlab& ls = w.vlin[v];
      ls.N = kDims[kVl[v].lin].N; ls.K = kVl[v].width; ls.NP = pad16(ls.N); ls.KP = pad16(ls.K);
      const long rows = kVl[v].on_edges ? E : N;
      const int rpt = kVl[v].on_edges ? kStreamRowsPerTask : kStreamNodeRowsPerTask;
      ls.nchunks = (int)((rows + rpt - 1) / rpt);
      if (ls.nchunks < 1) ls.nchunks = 1;
      ls.slab = c.take<float>(wg_slab_floats(ls.nchunks, ls.NP, ls.KP));
      ls.used = false;
    }
  }
  w.bytes = c.off + 256;
  w.ok = c.ok();
}
static int pack_layer(const b3d_mp_weights* mw, ClrLayerWs& w, bool tr, hipStream_t stream) {
  const b3d_linear* stacks[] = {mw->edge_update, mw->create_future_msgs, mw->create_past_msgs, mw->combine_future_past};
  const int cnt[] = {3, 2, 2, 3};
  for (int s = 0; s < 4; ++s)
    for (int i = 0; i < cnt[s]; ++i)
      B3D_REQUIRE(stacks[s][i].w && stacks[s][i].b, "CausalMessagePassing layer: null weight/bias pointer (stack %d layer %d)", s, i);
  constexpr int DX = DB::DX, DE = DB::DE, EIN = DB::EIN, MIN = DB::MIN, H1 = DB::EH1, MH = DB::MH;
  const b3d_linear &eu0 = mw->edge_update[0], &fu0 = mw->create_future_msgs[0], &pa0 = mw->create_past_msgs[0];
  PackDesc d[48];
  int n = 0;
  // per-node table T = (eu0[:, x_i] x + b | eu0[:, x_j] x | fu0[:, x] x + fu0[:, x0] x0 + b | pa0[:, x] x + pa0[:, x0] x0 + b | 0)
  d[n++] = pack_slice<Proj0Seq2<DB>>(0, w.wp_proj0, fu0.w + DX + DE, nullptr, MH, DX, MIN, 0, MH, false);    // x0 columns
  d[n++] = pack_slice<Proj0Seq2<DB>>(0, w.wp_proj0, pa0.w + DX + DE, nullptr, MH, DX, MIN, MH, MH, false);
  d[n++] = pack_slice<Proj0Seq2<DB>>(1, w.wp_proj0, eu0.w, eu0.b, H1, DX, EIN, HC::OA, H1, false);
  d[n++] = pack_slice<Proj0Seq2<DB>>(1, w.wp_proj0, eu0.w + DX, nullptr, H1, DX, EIN, HC::OB, H1, false);
  d[n++] = pack_slice<Proj0Seq2<DB>>(1, w.wp_proj0, fu0.w, fu0.b, MH, DX, MIN, HC::OF, MH, false);
  d[n++] = pack_slice<Proj0Seq2<DB>>(1, w.wp_proj0, pa0.w, pa0.b, MH, DX, MIN, HC::OP, MH, false);
  d[n++] = pack_slice<Proj0Seq2<DB>>(1, w.wp_proj0, nullptr, nullptr, DX, DX, DX, HC::OG, DX, false);         // (the k-NN block's columns: unused)
  for (int i = 0; i < 3; ++i)
    d[n++] = pack_desc<D::NodeFwdSeq>(i, w.wp_nfwd, mw->combine_future_past[i].w, mw->combine_future_past[i].b, kDims[CF0 + i].N, kDims[CF0 + i].K, false);
  if (tr) {
    using NB = D::NodeBwdSeq;
    for (int i = 0; i < 3; ++i)
      d[n++] = pack_desc<NB>(i, w.wp_nbwd, mw->combine_future_past[2 - i].w, nullptr, kDims[CF2 - i].K, kDims[CF2 - i].N, true);
    using GP = HC::GradProjSeq;
    d[n++] = pack_slice<GP>(0, w.wp_gproj, eu0.w, nullptr, DX, H1, EIN, 0, DX, true);
    d[n++] = pack_slice<GP>(1, w.wp_gproj, eu0.w + DX, nullptr, DX, H1, EIN, 0, DX, true);
    d[n++] = pack_slice<GP>(2, w.wp_gproj, fu0.w, nullptr, DX, MH, MIN, 0, DX, true);
    d[n++] = pack_slice<GP>(2, w.wp_gproj, fu0.w + DX + DE, nullptr, DX, MH, MIN, DX, DX, true);
    d[n++] = pack_slice<GP>(3, w.wp_gproj, pa0.w, nullptr, DX, MH, MIN, 0, DX, true);
    d[n++] = pack_slice<GP>(3, w.wp_gproj, pa0.w + DX + DE, nullptr, DX, MH, MIN, DX, DX, true);
  }
  B3D_TRY(pack_images(d, n, stream));
  FragDesc f[kFragMax];
  int m = 0;
  using FS = ES::Fwd;
  f[m++] = frag_desc<FS>(0, w.wp_efwd2, eu0.w + 2 * DX, nullptr, EIN, false);
  f[m++] = frag_desc<FS>(1, w.wp_efwd2, mw->edge_update[1].w, mw->edge_update[1].b, kDims[EU1].K, false);
  f[m++] = frag_desc<FS>(2, w.wp_efwd2, mw->edge_update[2].w, mw->edge_update[2].b, kDims[EU2].K, false);
  f[m++] = frag_desc<FS>(3, w.wp_efwd2, fu0.w + DX, nullptr, MIN, false);
  f[m++] = frag_desc<FS>(4, w.wp_efwd2, mw->create_future_msgs[1].w, mw->create_future_msgs[1].b, kDims[FU1].K, false);
  f[m++] = frag_desc<FS>(5, w.wp_efwd2, pa0.w + DX, nullptr, MIN, false);
  f[m++] = frag_desc<FS>(6, w.wp_efwd2, mw->create_past_msgs[1].w, mw->create_past_msgs[1].b, kDims[PA1].K, false);
  if (tr) {
    using BS = ES::Bwd;
    f[m++] = frag_desc<BS>(0, w.wp_ebwd2, mw->create_past_msgs[1].w, nullptr, kDims[PA1].K, true);
    f[m++] = frag_desc<BS>(1, w.wp_ebwd2, pa0.w + DX, nullptr, MIN, true);
    f[m++] = frag_desc<BS>(2, w.wp_ebwd2, mw->create_future_msgs[1].w, nullptr, kDims[FU1].K, true);
    f[m++] = frag_desc<BS>(3, w.wp_ebwd2, fu0.w + DX, nullptr, MIN, true);
    f[m++] = frag_desc<BS>(4, w.wp_ebwd2, mw->edge_update[2].w, nullptr, kDims[EU2].K, true);
    f[m++] = frag_desc<BS>(5, w.wp_ebwd2, mw->edge_update[1].w, nullptr, kDims[EU1].K, true);
    f[m++] = frag_desc<BS>(6, w.wp_ebwd2, eu0.w + 2 * DX, nullptr, EIN, true);
  }
  return pack_frags(f, m, stream);
}
}  // namespace clr
}  // namespace b3d

extern "C" size_t b3d_clr_layer_workspace_bytes(int32_t N, int32_t E, uint32_t flags) {
  ClrLayerWs w;
  carve_layer(w, nullptr, 0, N, E, (flags & B3D_FLAG_TRAINING) != 0);
  return w.bytes;
}

extern "C" int b3d_clr_layer_forward(const b3d_mp_weights* mw, const b3d_graph* g, const float* x, const float* x0,
                                     const float* e, const float* att, uint32_t flags, void* workspace, size_t workspace_bytes,
                                     float* x_new, float* e_new, b3d_stream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  B3D_REQUIRE(mw && g && x && x0 && e && att && workspace && x_new && e_new, "b3d_clr_layer_forward: null argument");
  const int N = g->N, E = g->E;
  B3D_REQUIRE(N > 0 && E > 0, "b3d_clr_layer_forward: empty graph (N=%d, E=%d)", N, E);
  const bool tr = (flags & B3D_FLAG_TRAINING) != 0;
  ClrLayerWs w;
  carve_layer(w, workspace, workspace_bytes, N, E, tr);
  if (!w.ok) return fail(B3D_ERR_WORKSPACE, "b3d_clr_layer_forward: workspace %zu < %zu bytes", workspace_bytes, w.bytes);
  B3D_TRY(pack_layer(mw, w, tr, stream));
  NodeProj0Args pa;
  pa.N = N; pa.x0 = x0; pa.x = x; pa.T0 = w.T0; pa.T = w.T; pa.wpack = w.wp_proj0;
  B3D_TRY((launch_node_split<DB, B3D_PROJ0_WAVES>(node_proj0_split_kernel<DB, B3D_PROJ0_WAVES>, "node_proj0", pa, N, stream, B3D_K_OTHER)));
  EdgeFwdHArgs ea;
  memset(&ea, 0, sizeof(ea));
  ea.E = E; ea.src = g->src; ea.dst = g->dst; ea.T = w.T; ea.e_in = e; ea.a_in = att;
  ea.e_out = w.e_out; ea.fut = w.fut; ea.past = w.past;
  const bool runs = past_runs_enabled() && g->past_ptr != nullptr && g->past_rows != nullptr && g->dst_unsorted != nullptr;
  ea.dst_unsorted = runs ? g->dst_unsorted : nullptr; ea.past_dump0 = (unsigned)edge_rows(E);
  ea.sH1 = w.sH1; ea.sH2 = w.sH2; ea.sF1 = w.sF1; ea.sP1 = w.sP1; ea.wpack = w.wp_efwd2;
  ea.rmask = reinterpret_cast<unsigned*>(w.rmask); ea.rmask2 = ea.rmask ? ea.rmask + edge_rows(E) * 16 : nullptr;
  if (tr) B3D_TRY(launch_es(es::edge_fwd_kernel<DB, true>, "edge_fwd", ea, E, stream, B3D_K_EDGE_FWD, ES::Fwd::LDS_BYTES));
  else B3D_TRY(launch_es(es::edge_fwd_kernel<DB, false>, "edge_fwd", ea, E, stream, B3D_K_EDGE_FWD, ES::Fwd::LDS_BYTES));
  B3D_HIP_CHECK(hipMemcpyAsync(e_new, w.e_out, (size_t)E * D::DE * sizeof(float), hipMemcpyDeviceToDevice, stream));
  NodeFwdArgs na;
  memset(&na, 0, sizeof(na));
  na.N = N; na.dst_ptr = runs ? g->past_ptr : g->dst_ptr; na.dst_perm = runs ? g->past_rows : g->dst_perm;
  na.src_ptr = g->src_ptr; na.src_perm = g->src_perm;
  na.past = w.past; na.fut = w.fut; na.x_out = x_new; na.wpack = w.wp_nfwd;
  if (tr) { na.M = w.M; na.sH1 = w.nH1; na.sH2 = w.nH2; }
  B3D_TRY((launch_node_split<D, kNodeWavesWide>(mp_node_fwd_split_kernel<D, kNodeWavesWide>, "mp_node_fwd", na, N, stream, B3D_K_NODE_FWD)));
  return B3D_OK;
}

extern "C" int b3d_clr_layer_backward(const b3d_mp_weights* mw, const b3d_graph* g, const float* x, const float* x0,
                                      const float* e, const float* att, const float* e_new, void* workspace, size_t workspace_bytes,
                                      const float* d_x_new, const float* d_e_new, float* d_x, float* d_x0, float* d_e, float* d_att,
                                      const b3d_mp_grads* gr, b3d_stream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  B3D_REQUIRE(mw && g && x && x0 && e && att && e_new && workspace && gr, "b3d_clr_layer_backward: null argument");
  const int N = g->N, E = g->E;
  B3D_REQUIRE(N > 0 && E > 0, "b3d_clr_layer_backward: empty graph");
  ClrLayerWs w;
  carve_layer(w, workspace, workspace_bytes, N, E, true);
  if (!w.ok) return fail(B3D_ERR_WORKSPACE, "b3d_clr_layer_backward: workspace %zu < %zu bytes", workspace_bytes, w.bytes);
  // missing upstream gradients are zeros
  if (!d_x_new) { B3D_HIP_CHECK(hipMemsetAsync(w.zero_n, 0, (size_t)N * D::DX * sizeof(float), stream)); d_x_new = w.zero_n; }
  if (!d_e_new) { B3D_HIP_CHECK(hipMemsetAsync(w.zero_e, 0, (size_t)E * D::DE * sizeof(float), stream)); d_e_new = w.zero_e; }
  NodeBwdArgs nb;
  memset(&nb, 0, sizeof(nb));
  nb.N = N; nb.dst_ptr = g->dst_ptr; nb.dst_perm = g->dst_perm; nb.src_ptr = g->src_ptr; nb.src_perm = g->src_perm;
  nb.g_direct = d_x_new;
  nb.sH1 = w.nH1; nb.sH2 = w.nH2; nb.dM = w.dM; nb.GdH2 = w.GnH2; nb.GdH1 = w.GnH1; nb.wpack = w.wp_nbwd;
  B3D_TRY(launch_node_split<D>(mp_node_bwd_split_kernel<D>, "mp_node_bwd", nb, N, stream, B3D_K_NODE_BWD));
  EdgeBwdHArgs eb;
  memset(&eb, 0, sizeof(eb));
  eb.E = E; eb.src = g->src; eb.dst = g->dst; eb.dM = w.dM;
  eb.de_out = d_e_new; eb.de_in = w.de_in;
  eb.sH1 = w.sH1; eb.sH2 = w.sH2; eb.sF1 = w.sF1; eb.sP1 = w.sP1;
  eb.rmask = reinterpret_cast<const unsigned*>(w.rmask); eb.rmask2 = eb.rmask + edge_rows(E) * 16;
  eb.da_acc = w.da; eb.da_first = 1;
  eb.GdH1 = w.GdH1; eb.GdH2 = w.GdH2; eb.Gde = w.Gde; eb.GdF1 = w.GdF1; eb.GdP1 = w.GdP1;
  eb.wpack = w.wp_ebwd2;
  B3D_TRY(launch_es(es::edge_bwd_kernel<DB, true>, "edge_bwd", eb, E, stream, B3D_K_EDGE_BWD, ES::Bwd::LDS_BYTES));
  if (d_e) B3D_HIP_CHECK(hipMemcpyAsync(d_e, w.de_in, (size_t)E * D::DE * sizeof(float), hipMemcpyDeviceToDevice, stream));
  if (d_att) B3D_HIP_CHECK(hipMemcpyAsync(d_att, w.da, (size_t)E * D::DA * sizeof(float), hipMemcpyDeviceToDevice, stream));
  {  // per-node sums of the first-layer gradients, then (dx | dx0) = (node columns)^T . dT
    NodeGradProjArgs ga;
    memset(&ga, 0, sizeof(ga));
    ga.N = N; ga.dst_ptr = g->dst_ptr; ga.dst_perm = g->dst_perm; ga.src_ptr = g->src_ptr; ga.src_perm = g->src_perm;
    ga.GdH1 = w.GdH1; ga.GdF1 = w.GdF1; ga.GdP1 = w.GdP1; ga.dT = w.dT;
    const long tasks = (long)((N + 15) / 16) * ListSumGeom<DB>::TASKS;
    {
      ProfScope ps(B3D_K_NODE_BWD, stream);
      hipLaunchKernelGGL(node_listsum_kernel<DB>, dim3((unsigned)((tasks + 3) / 4)), dim3(256), 0, stream, ga);
    }
    B3D_TRY(launch_check("node_listsum_kernel"));
    NodeBwdGArgs ng;
    memset(&ng, 0, sizeof(ng));
    ng.N = N; ng.dT = w.dT; ng.gx = w.gx; ng.wpack = w.wp_gproj;
    B3D_TRY(set_lds(node_bwd_g_kernel<DB, false>, NodeBwdGLds<DB>::BYTES));
    {
      ProfScope ps(B3D_K_NODE_BWD, stream);
      hipLaunchKernelGGL((node_bwd_g_kernel<DB, false>), dim3((N + 15) / 16), dim3(kNodeBwdGWaves * 64), NodeBwdGLds<DB>::BYTES, stream, ng);
    }
    B3D_TRY(launch_check("node_bwd_g_kernel"));
    const unsigned gb = (unsigned)(((long)N * D::DX + 255) / 256);
    if (d_x) hipLaunchKernelGGL(copy_cols_kernel, dim3(gb), dim3(256), 0, stream, (const float*)w.gx, 2 * D::DX, d_x, D::DX, 0, N, D::DX);
    if (d_x0) hipLaunchKernelGGL(copy_cols_kernel, dim3(gb), dim3(256), 0, stream, (const float*)w.gx + D::DX, 2 * D::DX, d_x0, D::DX, 0, N, D::DX);
    B3D_TRY(launch_check("copy_cols_kernel"));
  }
  // ---- weight gradients: one cooperative launch over the ten Linear layers (the model's job plan, one layer variant) -------------
  {
    hipLaunchKernelGGL(iota_kernel, dim3(((E > N ? E : N) + 255) / 256), dim3(256), 0, stream, w.iota, E > N ? E : N);
    B3D_TRY(launch_check("iota_kernel"));
    B3D_HIP_CHECK(hipMemsetAsync(w.zrow, 0, 256 * sizeof(float), stream));
    WgBuilder wb;
    wb.begin(w.ws_table, kLayerTableCap, w.ws_task_job, kLayerTaskCap, w.iota, stream);
    const int rp = kStreamRowsPerTask, rn = kStreamNodeRowsPerTask;
    const int* src = g->src;
    const int* dst = g->dst;
    auto mat = [&](int ll, long rows, const float* gp, const int* gidx, int gstride, int gcol0, const float* ap, int astride, int awidth) {
      Col c[1] = {{ap, nullptr, 0, astride, 0, awidth}};
      wb.add_block(w.lin[ll], rows, 1, rp, gp, gidx, 0, gstride, gcol0, c, 1, true);
    };
    auto ncol = [&](int vl, int gcol0, const float* ap) {       // node columns of a first layer: G = a column block of dT
      Col c[1] = {{ap, nullptr, 0, D::DX, 0, 96}};
      wb.add_block(w.vlin[vl], N, 1, rn, w.dT, nullptr, 0, HC::GW, gcol0, c, 1, false);
    };
    wb.add_eu0_edge_columns(w.vlin[VL_EU0E], E, 1, rp, w.GdH1, 0, D::EH1, e, 0, D::DE, att);
    ncol(VL_EU0XI, HC::OA, x); ncol(VL_EU0XJ, HC::OB, x);
    mat(LL_EU1, E, w.GdH2, nullptr, D::EH2, 0, w.sH1, D::EH1, 256);
    mat(LL_EU2, E, w.Gde, nullptr, D::DE, 0, w.sH2, D::EH2, 128);
    {
      Col ce[1] = {{w.e_out, nullptr, 0, D::DE, 0, 64}};
      wb.add_block(w.vlin[VL_PA0E], E, 1, rp, w.GdP1, nullptr, 0, D::MH, 0, ce, 1, true);
      wb.add_block(w.vlin[VL_FU0E], E, 1, rp, w.GdF1, nullptr, 0, D::MH, 0, ce, 1, true);
    }
    ncol(VL_PA0X, HC::OP, x); ncol(VL_PA0X0, HC::OP, x0); ncol(VL_FU0X, HC::OF, x); ncol(VL_FU0X0, HC::OF, x0);
    mat(LL_PA1, E, w.dM, dst, D::NIN, 0, w.sP1, D::MH, 192);
    mat(LL_FU1, E, w.dM, src, D::NIN, D::DM, w.sF1, D::MH, 192);
    mat(LL_CF0, N, w.GnH1, nullptr, D::NH1, 0, w.M, D::NIN, 256);
    mat(LL_CF1, N, w.GnH2, nullptr, D::NH2, 0, w.nH1, D::NH1, 192);
    mat(LL_CF2, N, d_x_new, nullptr, D::DX, 0, w.nH2, D::NH2, 128);
    B3D_TRY(wb.launch(w.zrow, stream));
  }
  // ---- slabs -> parameter gradients ----
  RedArgs ra;
  ra.nentries = 0;
  const b3d_linear_grad* groups[] = {gr->edge_update, gr->create_past_msgs, gr->create_future_msgs, gr->combine_future_past};
  const int firstl[] = {LL_EU0, LL_PA0, LL_FU0, LL_CF0}, cnt[] = {3, 2, 2, 3};
  for (int gi = 0; gi < 4; ++gi)
    for (int i = 0; i < cnt[gi]; ++i) {
      const int ll = firstl[gi] + i;
      const b3d_linear_grad& dst_g = groups[gi][i];
      if (ll == LL_EU0 || ll == LL_PA0 || ll == LL_FU0) {       // first layers: one slab set per column block
        for (int v = 0; v < kLayerVl; ++v) {
          if (kVl[v].lin != kLayerLin[ll]) continue;
          RedEntry en = red_entry(w.vlin[v], dst_g.w ? dst_g.w + kVl[v].col0 : nullptr, kVl[v].bias ? dst_g.b : nullptr);
          en.ld = w.lin[ll].K;
          ra.e[ra.nentries++] = en;
          if (ra.nentries == kRedMaxEntries) { B3D_TRY(launch_reduce(ra, stream)); ra.nentries = 0; }
        }
        continue;
      }
      ra.e[ra.nentries++] = red_entry(w.lin[ll], dst_g.w, dst_g.b);
      if (ra.nentries == kRedMaxEntries) { B3D_TRY(launch_reduce(ra, stream)); ra.nentries = 0; }
    }
  B3D_TRY(launch_reduce(ra, stream));
  return B3D_OK;
}

#ifdef B3D_EXP_STAMPS
extern "C" int b3d_debug_stamps_clr(long long* host_dst) {
  return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(b3d::g_stamps), sizeof(long long) * 4 * 512 * 32) == hipSuccess ? 0 : 1;
}
#endif
