// Point-cloud feature stacks of the frozen LiDAR / radar encoders, eval mode:
//   conv1d(C,64,1)+BN+ReLU -> conv1d(64,128,1)+BN+ReLU -> conv1d(128,1024,1)+BN(+ReLU) -> max over the points
// (reference models/pointnet.py:9-57 STN3d, :111-165 PointNetfeat; models/radarnet.py:9-37 RadarNetfeat).  A conv1d
// with kernel 1 is a Linear applied to every point; in eval mode BatchNorm is an affine map and is folded into the
// Linear by the caller.  35.7 MMAC per LiDAR detection (both stacks) -- 96 % of forward_feat.
//
// One workgroup of 8 wavefronts takes 128 points (one LiDAR cloud, two radar clouds): each wavefront carries a
// 16-point tile through the three layers in registers (b3d_dev.hpp), the 128 -> 1024 layer streams its weights in
// 11 chunks and never materialises the [points, 1024] activation: every finished 16x16 output block is reduced to
// its per-feature maximum over the 16 points with four DPP row rotations, the 8 tile maxima meet in LDS once per
// cloud.  Optionally the input is multiplied by a per-cloud 3x3 matrix (PointNet's input transform, the `bmm` of
// pointnet.py:137) while it is loaded.  ReLU after the last layer commutes with the maximum and is applied once.
#include "b3d_common.hpp"
#include "b3d_dev.hpp"
#include "b3d_pack.hpp"
#include "b3d_launch.hpp"

namespace b3d {
namespace {

using PointSeq = LayerSeq<L<16, 64>, L<64, 128>, L<128, 1024>>;
constexpr int kPointFeat = 1024;
constexpr int kPointLds = kLdsBytes + 8 * kPointFeat * 4;      // weight ring + one row of tile maxima per wavefront

struct PointFeatArgs {
  const float* x;       // [B, C, P]
  const float* trans;   // [B, 3, 3] or nullptr
  int B, C, relu_last;
  float* out;           // [B, 1024]
  float* out_min;       // STATS only: per-cloud minimum, sum and sum of squares of the last layer's output
  float* out_sum;
  float* out_sq;
  const float* wpack;   // PointSeq images
};

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <class Op>
__device__ __forceinline__ float row_reduce16(float v, Op op) {
  v = op(v, dpp_f<0x128>(v));
  v = op(v, dpp_f<0x124>(v));
  v = op(v, dpp_f<0x122>(v));
  v = op(v, dpp_f<0x121>(v));
  return v;
}
// maximum over the 16 lanes of a DPP row (= the 16 points of a tile that share feature piece q); every lane gets it
__device__ __forceinline__ float row_max16(float v) {
  v = fmaxf(v, dpp_f<0x128>(v));      // row_ror:8
  v = fmaxf(v, dpp_f<0x124>(v));      // row_ror:4
  v = fmaxf(v, dpp_f<0x122>(v));      // row_ror:2
  v = fmaxf(v, dpp_f<0x121>(v));      // row_ror:1
  return v;
}

// STATS (train mode: the last BatchNorm uses the statistics of THIS batch, so it cannot be folded before its input
// exists): the last layer is evaluated raw (conv bias only) and every cloud's per-feature maximum, minimum, sum and
// sum of squares over its points are written; the caller derives mean / variance per feature from the sums and
// applies the now-known affine map to the maximum (positive scale) or the minimum (negative scale) -- the [points,
// 1024] activation is never stored.  Per weight chunk the 8 tile partials meet in a double-buffered LDS area and are
// combined behind the NEXT chunk's barrier (no extra barrier per chunk).
constexpr int kStatChunkFeat = chunk_rows(128, 1024, PointSeq::bf(2));   // features of one weight chunk of the 128 -> 1024 layer
template <int P, bool STATS>
__global__ __launch_bounds__(512, 1) void point_feat_kernel(const PointFeatArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int TPC = P / 16, CPW = 8 / TPC;                   // tiles per cloud, clouds per workgroup
  static_assert(P % 16 == 0 && 8 % TPC == 0, "a cloud is a whole number of tiles, a workgroup a whole number of clouds");
  WStreamT<512> ws;
  ws.init(a.wpack, smem);
  ws.template start<PointSeq>();
  float* xpart = smem + 2 * kWBufFloats;                       // [8 wavefronts][1024]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int m = lane & 15, q = lane >> 4;
  const int ngroups = (a.B + CPW - 1) / CPW;
  for (int g = blockIdx.x; g < ngroups; g += gridDim.x) {
    const bool more = g + (int)gridDim.x < ngroups;
    const int cloud = g * CPW + wave / TPC;
    const int cc = cloud < a.B ? cloud : a.B - 1;              // idle wavefronts of the last group recompute a valid cloud
    const int p = (wave % TPC) * 16 + m;
    v4f in[1] = {v4f{0.f, 0.f, 0.f, 0.f}};
    if (q == 0) {                                              // features 0..3 of the 16-wide padded input block
      const float* xp = a.x + ((long)cc * a.C) * P + p;
      float f[4] = {0.f, 0.f, 0.f, 0.f};
      for (int c = 0; c < a.C && c < 4; ++c) f[c] = xp[(long)c * P];
      if (a.trans) {                                           // x' = x^T . T  (row vector times the cloud's 3x3)
        const float* t = a.trans + (long)cc * 9;
        const float x0 = f[0], x1 = f[1], x2 = f[2];
        f[0] = x0 * t[0] + x1 * t[3] + x2 * t[6];
        f[1] = x0 * t[1] + x1 * t[4] + x2 * t[7];
        f[2] = x0 * t[2] + x1 * t[5] + x2 * t[8];
      }
      in[0] = v4f{f[0], f[1], f[2], f[3]};
    }
    wait_for(in[0]);
    v4f h1[4], h2[8];
    linear<PointSeq, 0, true>(ws, more, in, h1);
    linear<PointSeq, 1, true>(ws, more, h1, h2);
    if constexpr (STATS) {
      // xpart as [2 parities][4 quantities][8 wavefronts][96 features]
      constexpr int CB = kStatChunkFeat / 16;                  // output blocks per weight chunk
      auto fmax_ = [](float x, float y) { return fmaxf(x, y); };
      auto fmin_ = [](float x, float y) { return fminf(x, y); };
      auto fadd_ = [](float x, float y) { return x + y; };
      auto combine = [&](int chunk) {                          // all wavefronts: tile partials of `chunk` -> global
        const float* buf = xpart + (chunk & 1) * 4 * 8 * kStatChunkFeat;
        const int nfeat = min(kStatChunkFeat, kPointFeat - chunk * kStatChunkFeat);
        for (int f = threadIdx.x; f < CPW * nfeat; f += 512) {
          const int cl = f / nfeat, feat = f - cl * nfeat;
          const float* src = buf + (cl * TPC) * kStatChunkFeat + feat;
          float vmax = src[0], vmin = src[8 * kStatChunkFeat], vsum = src[16 * kStatChunkFeat], vsq = src[24 * kStatChunkFeat];
#pragma unroll
          for (int t = 1; t < TPC; ++t) {
            vmax = fmaxf(vmax, src[t * kStatChunkFeat]);
            vmin = fminf(vmin, src[(8 + t) * kStatChunkFeat]);
            vsum += src[(16 + t) * kStatChunkFeat];
            vsq += src[(24 + t) * kStatChunkFeat];
          }
          const int c = g * CPW + cl;
          if (c < a.B) {
            const long o = (long)c * kPointFeat + chunk * kStatChunkFeat + feat;
            a.out[o] = vmax; a.out_min[o] = vmin; a.out_sum[o] = vsum; a.out_sq[o] = vsq;
          }
        }
      };
      linear_emit<PointSeq, 2, false>(ws, more, h2, [&](int mb, v4f v) {
        const int chunk = mb / CB, lb = mb - chunk * CB;
        if (lb == 0 && chunk > 0) combine(chunk - 1);          // behind this chunk's barrier: every tile partial is there
        float* buf = xpart + (chunk & 1) * 4 * 8 * kStatChunkFeat + wave * kStatChunkFeat + lb * 16 + 4 * q;
        v4f t;
        t.x = row_reduce16(v.x, fmax_); t.y = row_reduce16(v.y, fmax_); t.z = row_reduce16(v.z, fmax_); t.w = row_reduce16(v.w, fmax_);
        if (m == 0) *reinterpret_cast<v4f*>(buf) = t;
        t.x = row_reduce16(v.x, fmin_); t.y = row_reduce16(v.y, fmin_); t.z = row_reduce16(v.z, fmin_); t.w = row_reduce16(v.w, fmin_);
        if (m == 0) *reinterpret_cast<v4f*>(buf + 8 * kStatChunkFeat) = t;
        t.x = row_reduce16(v.x, fadd_); t.y = row_reduce16(v.y, fadd_); t.z = row_reduce16(v.z, fadd_); t.w = row_reduce16(v.w, fadd_);
        if (m == 0) *reinterpret_cast<v4f*>(buf + 16 * kStatChunkFeat) = t;
        t.x = row_reduce16(v.x * v.x, fadd_); t.y = row_reduce16(v.y * v.y, fadd_); t.z = row_reduce16(v.z * v.z, fadd_);
        t.w = row_reduce16(v.w * v.w, fadd_);
        if (m == 0) *reinterpret_cast<v4f*>(buf + 24 * kStatChunkFeat) = t;
      });
      __syncthreads();
      combine((kPointFeat + kStatChunkFeat - 1) / kStatChunkFeat - 1);
      continue;                                                // next group: its first xpart write is 3 barriers away
    }
    float* mine = xpart + wave * kPointFeat;
    linear_emit<PointSeq, 2, false>(ws, more, h2, [&](int mb, v4f v) {
      v.x = row_max16(v.x); v.y = row_max16(v.y); v.z = row_max16(v.z); v.w = row_max16(v.w);
      if (m == 0) *reinterpret_cast<v4f*>(mine + mb * 16 + 4 * q) = v;
    });
    __syncthreads();
    for (int f = threadIdx.x; f < CPW * kPointFeat; f += 512) {
      const int cl = f / kPointFeat, feat = f % kPointFeat;
      const float* src = xpart + (cl * TPC) * kPointFeat + feat;
      float v = src[0];
#pragma unroll
      for (int t = 1; t < TPC; ++t) v = fmaxf(v, src[t * kPointFeat]);
      if (a.relu_last) v = fmaxf(v, 0.f);
      const int c = g * CPW + cl;
      if (c < a.B) a.out[(long)c * kPointFeat + feat] = v;
    }
    // the next group's first write to xpart sits behind two more weight-chunk barriers: no barrier needed here
  }
}

}  // namespace
}  // namespace b3d

using namespace b3d;

extern "C" size_t b3d_point_feat_workspace_bytes(void) { return (size_t)PointSeq::TOTAL_FLOATS * sizeof(float) + 256; }

static int point_feat_launch(const b3d_linear* conv, const float* x, const float* trans, int32_t B, int32_t C, int32_t P,
                             int32_t relu_last, void* workspace, size_t workspace_bytes, float* out, float* out_min,
                             float* out_sum, float* out_sq, hipStream_t stream);

extern "C" int b3d_point_feat(const b3d_linear* conv, const float* x, const float* trans, int32_t B, int32_t C, int32_t P,
                              int32_t relu_last, void* workspace, size_t workspace_bytes, float* out, b3d_stream stream_) {
  return point_feat_launch(conv, x, trans, B, C, P, relu_last, workspace, workspace_bytes, out, nullptr, nullptr, nullptr,
                           (hipStream_t)stream_);
}

extern "C" int b3d_point_feat_stats(const b3d_linear* conv, const float* x, const float* trans, int32_t B, int32_t C, int32_t P,
                                    void* workspace, size_t workspace_bytes, float* out_max, float* out_min, float* out_sum,
                                    float* out_sq, b3d_stream stream_) {
  B3D_REQUIRE(B == 0 || (out_min && out_sum && out_sq), "b3d_point_feat_stats: null output");
  return point_feat_launch(conv, x, trans, B, C, P, 0, workspace, workspace_bytes, out_max, out_min, out_sum, out_sq,
                           (hipStream_t)stream_);
}

static int point_feat_launch(const b3d_linear* conv, const float* x, const float* trans, int32_t B, int32_t C, int32_t P,
                             int32_t relu_last, void* workspace, size_t workspace_bytes, float* out, float* out_min,
                             float* out_sum, float* out_sq, hipStream_t stream) {
  B3D_REQUIRE(conv && workspace && (B == 0 || (x && out)), "b3d_point_feat: null argument");
  B3D_REQUIRE(conv[0].w && conv[1].w && conv[2].w && conv[0].b && conv[1].b && conv[2].b, "b3d_point_feat: null layer");
  B3D_REQUIRE(B >= 0 && C >= 1 && C <= 4 && (P == 64 || P == 128), "b3d_point_feat: B %d, C %d, P %d (C <= 4, P 64 or 128)", B, C, P);
  B3D_REQUIRE(!trans || C == 3, "b3d_point_feat: the input transform is 3x3");
  if (workspace_bytes < b3d_point_feat_workspace_bytes()) return fail(B3D_ERR_WORKSPACE, "b3d_point_feat: workspace too small");
  if (B == 0) return B3D_OK;
  float* wp = (float*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  PackDesc d[3];
  d[0] = pack_desc<PointSeq>(0, wp, (const float*)conv[0].w, (const float*)conv[0].b, 64, C, false);
  d[1] = pack_desc<PointSeq>(1, wp, (const float*)conv[1].w, (const float*)conv[1].b, 128, 64, false);
  d[2] = pack_desc<PointSeq>(2, wp, (const float*)conv[2].w, (const float*)conv[2].b, 1024, 128, false);
  B3D_TRY(pack_images(d, 3, stream));
  PointFeatArgs a;
  a.x = x; a.trans = trans; a.B = B; a.C = C; a.relu_last = relu_last; a.out = out; a.wpack = wp;
  a.out_min = out_min; a.out_sum = out_sum; a.out_sq = out_sq;
  const bool stats = out_min != nullptr;
  const int cpw = P == 128 ? 1 : 2;
  int groups = (B + cpw - 1) / cpw;
  if (groups > 1024) groups = 1024;                            // persistent: <= 4 groups per CU in flight order
#define B3D_POINT_LAUNCH(PP, ST)                                                                      \
  do {                                                                                                \
    B3D_TRY(set_lds(point_feat_kernel<PP, ST>, kPointLds));                                           \
    ProfScope ps(B3D_K_POINT_FEAT, stream);                                                           \
    hipLaunchKernelGGL((point_feat_kernel<PP, ST>), dim3(groups), dim3(512), kPointLds, stream, a);   \
  } while (0)
  if (P == 128 && stats) B3D_POINT_LAUNCH(128, true);
  else if (P == 128) B3D_POINT_LAUNCH(128, false);
  else if (stats) B3D_POINT_LAUNCH(64, true);
  else B3D_POINT_LAUNCH(64, false);
#undef B3D_POINT_LAUNCH
  return launch_check("point_feat_kernel");
}
