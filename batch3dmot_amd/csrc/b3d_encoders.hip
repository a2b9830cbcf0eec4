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
  const float* wpack;   // PointSeq images
};

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
// maximum over the 16 lanes of a DPP row (= the 16 points of a tile that share feature piece q); every lane gets it
__device__ __forceinline__ float row_max16(float v) {
  v = fmaxf(v, dpp_f<0x128>(v));      // row_ror:8
  v = fmaxf(v, dpp_f<0x124>(v));      // row_ror:4
  v = fmaxf(v, dpp_f<0x122>(v));      // row_ror:2
  v = fmaxf(v, dpp_f<0x121>(v));      // row_ror:1
  return v;
}

template <int P>
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

extern "C" int b3d_point_feat(const b3d_linear* conv, const float* x, const float* trans, int32_t B, int32_t C, int32_t P,
                              int32_t relu_last, void* workspace, size_t workspace_bytes, float* out, b3d_stream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
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
  const int cpw = P == 128 ? 1 : 2;
  int groups = (B + cpw - 1) / cpw;
  if (groups > 1024) groups = 1024;                            // persistent: <= 4 groups per CU in flight order
  if (P == 128) {
    B3D_TRY(set_lds(point_feat_kernel<128>, kPointLds));
    hipLaunchKernelGGL(point_feat_kernel<128>, dim3(groups), dim3(512), kPointLds, stream, a);
  } else {
    B3D_TRY(set_lds(point_feat_kernel<64>, kPointLds));
    hipLaunchKernelGGL(point_feat_kernel<64>, dim3(groups), dim3(512), kPointLds, stream, a);
  }
  return launch_check("point_feat_kernel");
}
