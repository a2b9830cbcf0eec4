// Point-cloud feature stacks of the frozen LiDAR / radar encoders, eval mode:
//   conv1d(C,64,1)+BN+ReLU -> conv1d(64,128,1)+BN+ReLU -> conv1d(128,1024,1)+BN(+ReLU) -> max over the points
// (reference models/pointnet.py:9-57 STN3d, :111-165 PointNetfeat; models/radarnet.py:9-37 RadarNetfeat).  A conv1d
// with kernel 1 is a Linear applied to every point; in eval mode BatchNorm is an affine map and is folded into the
// Linear by the caller.  35.7 MMAC per LiDAR detection (both stacks) -- 96 % of forward_feat.
//
// One workgroup of 8 wavefronts takes 256 points (two LiDAR clouds, four radar clouds): each wavefront carries TWO
// 16-point tiles through the three layers in registers, so that every weight fragment read from LDS feeds two tiles
// (the 16-row form reads 512 B of weights per MFMA -- half of the LDS array's rate at full MFMA rate -- and meets a
// workgroup barrier every 96 MFMAs).  The 128 -> 1024 layer streams its weights in 16 chunks and never materialises
// the [points, 1024] activation.  It is evaluated with the operands SWAPPED (points x features instead of features x
// points: same registers, the other MFMA argument), which leaves the 4 accumulator values of a lane on 4 POINTS of
// one feature: the per-feature maximum / minimum / sum / sum of squares over a wavefront's 32 points is 7 in-lane
// operations per quantity and two v_permlane{32,16}_swap steps that each finish two quantities at once, instead of
// sixteen 4-step DPP rotations per output block.  The epilogue of block k is issued behind the first MFMAs of block
// k + 1.  The clouds that do not fill a last round of 256-point groups go through the same code one tile per
// wavefront (128-point groups), so that the tail is spread over all CUs.  Optionally the input is multiplied by a
// per-cloud 3x3 matrix (PointNet's input transform, the `bmm` of pointnet.py:137) while it is loaded.  ReLU after the
// last layer commutes with the maximum and is applied once.
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
  int n_wide, n_narrow; // groups of 8 x 2 tiles, then groups of 8 x 1 tiles
};

// v_max_f32 / v_add_f32 on raw registers (fmaxf on a bit-cast value makes hipcc canonicalise both inputs first)
__device__ __forceinline__ float vmax_raw(float x, float y) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
  return r;
}
typedef unsigned u2v __attribute__((ext_vector_type(2)));
// lanes 0..31 hold quantity A of (feature = lane & 15, point quarter = lane >> 4), lanes 32..63 ... : given A and B per
// lane, returns op over the four lane quarters of A in lanes 0..31 and of B in lanes 32..63
template <bool MAX>
__device__ __forceinline__ float quarter_reduce2(float a, float b) {
  const u2v r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  const float x = __uint_as_float(r.x), y = __uint_as_float(r.y);
  const float e = MAX ? vmax_raw(x, y) : x + y;             // lanes 0..31: A of quarters {0,2} / {1,3}; 32..63: B
  const u2v s = __builtin_amdgcn_permlane16_swap(__float_as_uint(e), __float_as_uint(e), false, false);
  const float u = __uint_as_float(s.x), v = __uint_as_float(s.y);
  return MAX ? vmax_raw(u, v) : u + v;
}
__device__ __forceinline__ v4f bf_mfma6_pts(const Bf3& x, const Bf3& w, v4f acc) {   // rows = points, columns = features
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x.p2, w.p0, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x.p1, w.p1, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x.p0, w.p2, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x.p1, w.p0, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x.p0, w.p1, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x.p0, w.p0, acc, 0, 0, 0);
  return acc;
}

// STATS (train mode: the last BatchNorm uses the statistics of THIS batch, so it cannot be folded before its input
// exists): the last layer is evaluated raw (conv bias only) and every cloud's per-feature maximum, minimum, sum and
// sum of squares over its points are written; the caller derives mean / variance per feature from the sums and
// applies the now-known affine map to the maximum (positive scale) or the minimum (negative scale) -- the [points,
// 1024] activation is never stored.  Per weight chunk the wavefronts' partials meet in a double-buffered LDS area and
// are combined behind the NEXT chunk's barrier (no extra barrier per chunk).
constexpr int kStatChunkFeat = chunk_rows(128, 1024, PointSeq::bf(2));   // features of one weight chunk of the 128 -> 1024 layer
static_assert(PointSeq::layer_chunks(0) == 1 && PointSeq::layer_chunks(1) == 1 && !PointSeq::bf(0) && PointSeq::bf(1) && PointSeq::bf(2),
              "point_pass is written for a one-chunk fp32 first layer and bf16x3 images after it");

// One group of 8 * 16 * T points: T tiles per wavefront.
// STATS: 0 = eval (maximum only), 1 = maximum, minimum, sum, sum of squares, 2 = maximum, sum, sum of squares (the caller has
// folded sign(gamma) of the last BatchNorm into the last layer, b3d_point_stack_train: the minimum is never needed)
template <int P, int T, int STATS>
struct PointPass {
  static constexpr int WPC = P / (16 * T);                   // wavefronts per cloud
  static constexpr int CPW = 8 / WPC;                        // clouds per group
  static_assert(P % (16 * T) == 0 && WPC >= 1 && 8 % WPC == 0, "a wavefront's tiles lie in one cloud");
  static constexpr int CB = kStatChunkFeat / 16;             // output blocks per weight chunk
  static constexpr int S3 = row_stride(128, true);
  static constexpr int NCH3 = PointSeq::layer_chunks(2);
  static constexpr int F = kStatChunkFeat;

  // STATS: the wavefronts' partials of `chunk` -> global (all 512 threads)
  static __device__ __forceinline__ void combine(const PointFeatArgs& a, const float* xpart, int cloud0, int chunk) {
    const float* buf = xpart + (chunk & 1) * 4 * 8 * F;
    const int nfeat = min(F, kPointFeat - chunk * F);
    for (int f = threadIdx.x; f < CPW * nfeat; f += 512) {
      const int cl = f / nfeat, feat = f - cl * nfeat;
      const float* src = buf + (cl * WPC) * F + feat;
      float vmax = src[0], vmin = src[8 * F], vsum = src[16 * F], vsq = src[24 * F];
#pragma unroll
      for (int t = 1; t < WPC; ++t) {
        vmax = fmaxf(vmax, src[t * F]);
        if constexpr (STATS == 1) vmin = fminf(vmin, src[(8 + t) * F]);
        vsum += src[(16 + t) * F];
        vsq += src[(24 + t) * F];
      }
      const int c = cloud0 + cl;
      if (c < a.B) {
        const long o = (long)c * kPointFeat + chunk * F + feat;
        a.out[o] = vmax; a.out_sum[o] = vsum; a.out_sq[o] = vsq;
        if constexpr (STATS == 1) a.out_min[o] = vmin;
      }
    }
  }

  template <int CH>
  static __device__ __forceinline__ void chunk3(WStreamT<512>& ws, bool more, const PointFeatArgs& a, float* xpart, int cloud0,
                                                const Bf3 (&x3)[T][4]) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int m = lane & 15, q = lane >> 4;
    const float* w = ws.template acquire<PointSeq, 2 + CH>(more);
    if constexpr (STATS != 0 && CH > 0) combine(a, xpart, cloud0, CH - 1);   // behind this chunk's barrier: every partial is there
    constexpr int nb = (kPointFeat - CH * F < F ? kPointFeat - CH * F : F) / 16;
    const float* wrow = w + m * S3 + 4 * q;
    const float* wbias = w + m * S3 + bias_col(128, true);   // bias of feature m of a block
    // where this lane's finished values go: lanes 0..31 hold maximum | sum, lanes 32..63 minimum (negated) | sum of squares
    float* mm;
    float* ss = nullptr;
    if constexpr (STATS != 0) {
      mm = xpart + (CH & 1) * 4 * 8 * F + (lane >> 5) * 8 * F + wave * F + m;
      ss = mm + 16 * F;
    } else {
      mm = xpart + wave * kPointFeat + CH * F + m;
    }
    const unsigned flip = (STATS == 1 && lane >= 32) ? 0x80000000u : 0u;
    auto epilogue = [&](int lb, const v4f (&acc)[T]) {
      float mx = acc[0].x;
      mx = fmaxf(mx, acc[0].y); mx = fmaxf(mx, acc[0].z); mx = fmaxf(mx, acc[0].w);
#pragma unroll
      for (int t = 1; t < T; ++t) { mx = fmaxf(mx, acc[t].x); mx = fmaxf(mx, acc[t].y); mx = fmaxf(mx, acc[t].z); mx = fmaxf(mx, acc[t].w); }
      if constexpr (STATS != 0) {
        float mn = acc[0].x, sm = acc[0].x, sq = acc[0].x * acc[0].x;
        if constexpr (STATS == 1) { mn = fminf(mn, acc[0].y); mn = fminf(mn, acc[0].z); mn = fminf(mn, acc[0].w); }
        sm += acc[0].y; sm += acc[0].z; sm += acc[0].w;
        sq = fmaf(acc[0].y, acc[0].y, sq); sq = fmaf(acc[0].z, acc[0].z, sq); sq = fmaf(acc[0].w, acc[0].w, sq);
#pragma unroll
        for (int t = 1; t < T; ++t) {
          if constexpr (STATS == 1) { mn = fminf(mn, acc[t].x); mn = fminf(mn, acc[t].y); mn = fminf(mn, acc[t].z); mn = fminf(mn, acc[t].w); }
          sm += acc[t].x; sm += acc[t].y; sm += acc[t].z; sm += acc[t].w;
          sq = fmaf(acc[t].x, acc[t].x, sq); sq = fmaf(acc[t].y, acc[t].y, sq); sq = fmaf(acc[t].z, acc[t].z, sq); sq = fmaf(acc[t].w, acc[t].w, sq);
        }
        // (STATS == 2: lanes 32..63 park a second copy of the maximum in the plane the minimum would use; nobody reads it)
        const float r = STATS == 1 ? quarter_reduce2<true>(mx, -mn) : quarter_reduce2<true>(mx, mx);
        mm[lb * 16] = __uint_as_float(__float_as_uint(r) ^ flip);
        ss[lb * 16] = quarter_reduce2<false>(sm, sq);
      } else {
        mm[lb * 16] = quarter_reduce2<true>(mx, mx);
      }
    };
    Bf3 cur = bf_load<128>(wrow);
    float nbias = wbias[0];
    v4f prev[T];
#pragma unroll
    for (int lb = 0; lb < nb; ++lb) {
      v4f acc[T];
#pragma unroll
      for (int t = 0; t < T; ++t) acc[t] = v4f{nbias, nbias, nbias, nbias};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        Bf3 nxt = cur;
        if (c + 1 < 4) nxt = bf_load<128>(wrow + lb * 16 * S3 + 16 * (c + 1));
        else if (lb + 1 < nb) { nxt = bf_load<128>(wrow + (lb + 1) * 16 * S3); nbias = wbias[(lb + 1) * 16 * S3]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] = bf_mfma6_pts(x3[t][c], cur, acc[t]);
        __builtin_amdgcn_sched_barrier(0);
        if (c == 0 && lb > 0) epilogue(lb - 1, prev);        // the previous block's reductions run under these MFMAs
        cur = nxt;
      }
#pragma unroll
      for (int t = 0; t < T; ++t) prev[t] = acc[t];
    }
    epilogue(nb - 1, prev);
  }

  template <int... CH>
  static __device__ __forceinline__ void layer3(WStreamT<512>& ws, bool more, const PointFeatArgs& a, float* xpart, int cloud0,
                                                const Bf3 (&x3)[T][4], std::integer_sequence<int, CH...>) {
    (chunk3<CH>(ws, more, a, xpart, cloud0, x3), ...);
  }

  static __device__ __forceinline__ void run(WStreamT<512>& ws, bool more, const PointFeatArgs& a, float* xpart, int cloud0) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int m = lane & 15, q = lane >> 4;
    const int cloud = cloud0 + wave / WPC;
    const int cc = cloud < a.B ? cloud : a.B - 1;            // idle wavefronts of the last group recompute a valid cloud
    v4f in[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      in[t] = v4f{0.f, 0.f, 0.f, 0.f};
      if (q == 0) {                                          // features 0..3 of the 16-wide padded input block
        const int p = ((wave % WPC) * T + t) * 16 + m;
        const float* xp = a.x + ((long)cc * a.C) * P + p;
        float f[4] = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < a.C && c < 4; ++c) f[c] = xp[(long)c * P];
        if (a.trans) {                                       // x' = x^T . T  (row vector times the cloud's 3x3)
          const float* tr = a.trans + (long)cc * 9;
          const float x0 = f[0], x1 = f[1], x2 = f[2];
          f[0] = x0 * tr[0] + x1 * tr[3] + x2 * tr[6];
          f[1] = x0 * tr[1] + x1 * tr[4] + x2 * tr[7];
          f[2] = x0 * tr[2] + x1 * tr[5] + x2 * tr[8];
        }
        in[t] = v4f{f[0], f[1], f[2], f[3]};
      }
    }
#pragma unroll
    for (int t = 0; t < T; ++t) wait_for(in[t]);
    // C -> 64, exact fp32 MFMA
    v4f h1[T][4];
    {
      constexpr int S = row_stride(16, false);
      const float* w = ws.template acquire<PointSeq, 0>(more);
      const float* wrow = w + m * S + 4 * q;
      const float* wb = w + 4 * q * S + bias_col(16, false);
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        const v4f frag = *reinterpret_cast<const v4f*>(wrow + mb * 16 * S);
        const float* b = wb + mb * 16 * S;
        const v4f bias = v4f{b[0], b[S], b[2 * S], b[3 * S]};
#pragma unroll
        for (int t = 0; t < T; ++t) h1[t][mb] = relu4(mfma4(frag, in[t], bias));
      }
    }
    // 64 -> 128
    Bf3 x3[T][4];
    {
      constexpr int S = row_stride(64, true);
      Bf3 x2[T][2];
      const float* w = ws.template acquire<PointSeq, 1>(more);
#pragma unroll
      for (int t = 0; t < T; ++t) { x2[t][0] = bf_split(h1[t][0], h1[t][1]); x2[t][1] = bf_split(h1[t][2], h1[t][3]); }
      const float* wrow = w + m * S + 4 * q;
      const float* wb = w + 4 * q * S + bias_col(64, true);
      auto bias = [&](int mb) { const float* b = wb + mb * 16 * S; return v4f{b[0], b[S], b[2 * S], b[3 * S]}; };
      Bf3 cur = bf_load<64>(wrow);
      v4f nbias = bias(0);
      v4f h2[T][2];                                          // two blocks at a time -> one operand group of the next layer
#pragma unroll
      for (int mb = 0; mb < 8; ++mb) {
        v4f acc[T];
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] = nbias;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          Bf3 nxt = cur;
          if (c == 0) nxt = bf_load<64>(wrow + mb * 16 * S + 16);
          else if (mb + 1 < 8) { nxt = bf_load<64>(wrow + (mb + 1) * 16 * S); nbias = bias(mb + 1); }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t = 0; t < T; ++t) acc[t] = bf_mfma6(cur, x2[t][c], acc[t]);
          cur = nxt;
        }
#pragma unroll
        for (int t = 0; t < T; ++t) {
          h2[t][mb & 1] = relu4(acc[t]);
          if (mb & 1) x3[t][mb >> 1] = bf_split(h2[t][0], h2[t][1]);
        }
      }
    }
    layer3(ws, more, a, xpart, cloud0, x3, std::make_integer_sequence<int, NCH3>{});
    __syncthreads();
    if constexpr (STATS != 0) {
      combine(a, xpart, cloud0, NCH3 - 1);
    } else {
      for (int f = threadIdx.x; f < CPW * kPointFeat; f += 512) {
        const int cl = f / kPointFeat, feat = f % kPointFeat;
        const float* src = xpart + (cl * WPC) * kPointFeat + feat;
        float v = src[0];
#pragma unroll
        for (int t = 1; t < WPC; ++t) v = fmaxf(v, src[t * kPointFeat]);
        if (a.relu_last) v = relu1(v);
        const int c = cloud0 + cl;
        if (c < a.B) a.out[(long)c * kPointFeat + feat] = v;
      }
    }
    // the next group's first write to xpart sits behind three more weight-chunk barriers: no barrier needed here
  }
};

// items 0 .. n_wide-1: groups of two tiles per wavefront; then n_narrow groups of one tile per wavefront
template <int P, int STATS>
__global__ __launch_bounds__(512, 1) void point_feat_kernel(const PointFeatArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  using Wide = PointPass<P, 2, STATS>;
  using Narrow = PointPass<P, 1, STATS>;
  WStreamT<512> ws;
  ws.init(a.wpack, smem);
  ws.template start<PointSeq>();
  float* xpart = smem + 2 * kWBufFloats;
  const int items = a.n_wide + a.n_narrow;
  for (int it = blockIdx.x; it < items; it += gridDim.x) {
    const bool more = it + (int)gridDim.x < items;
    if (it < a.n_wide) Wide::run(ws, more, a, xpart, it * Wide::CPW);
    else Narrow::run(ws, more, a, xpart, a.n_wide * Wide::CPW + (it - a.n_wide) * Narrow::CPW);
  }
}

}  // namespace
}  // namespace b3d

using namespace b3d;

extern "C" size_t b3d_point_feat_workspace_bytes(void) { return (size_t)PointSeq::TOTAL_FLOATS * sizeof(float) + 256; }

static int point_feat_launch(const b3d_linear* conv, const float* x, const float* trans, int32_t B, int32_t C, int32_t P,
                             int32_t relu_last, void* workspace, size_t workspace_bytes, float* out, float* out_min,
                             float* out_sum, float* out_sq, hipStream_t stream, const float* sign3 = nullptr);

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

// out_sum without out_min: the three-quantity form (STATS 2); sign3 [1024] or nullptr: rows of the last layer's image (weights and
// bias) are negated where sign3 < 0
static int point_feat_launch(const b3d_linear* conv, const float* x, const float* trans, int32_t B, int32_t C, int32_t P,
                             int32_t relu_last, void* workspace, size_t workspace_bytes, float* out, float* out_min,
                             float* out_sum, float* out_sq, hipStream_t stream, const float* sign3) {
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
  d[2].row_sign = sign3;
  B3D_TRY(pack_images(d, 3, stream));
  PointFeatArgs a;
  a.x = x; a.trans = trans; a.B = B; a.C = C; a.relu_last = relu_last; a.out = out; a.wpack = wp;
  a.out_min = out_min; a.out_sum = out_sum; a.out_sq = out_sq;
  const int stats = out_min != nullptr ? 1 : (out_sum != nullptr ? 2 : 0);
  // 256-point groups; when the last round of them over the CUs would be less than half full, its clouds go as 128-point
  // groups instead (twice as many workgroups share the tail)
  const int cpw_wide = 256 / P, cpw_narrow = 128 / P;
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    return n;
  }();
  int n_wide = (B + cpw_wide - 1) / cpw_wide, n_narrow = 0;
  const int tail = n_wide % cus;
  if (tail > 0 && 2 * tail < cus) {
    n_wide -= tail;
    n_narrow = (B - n_wide * cpw_wide + cpw_narrow - 1) / cpw_narrow;
  }
  a.n_wide = n_wide; a.n_narrow = n_narrow;
  int groups = n_wide + n_narrow;
  if (groups > 4 * cus) groups = 4 * cus;                      // persistent beyond that: <= 4 workgroups per CU in dispatch order
#define B3D_POINT_LAUNCH(PP, ST)                                                                      \
  do {                                                                                                \
    B3D_TRY(set_lds(point_feat_kernel<PP, ST>, kPointLds));                                           \
    ProfScope ps(B3D_K_POINT_FEAT, stream);                                                           \
    hipLaunchKernelGGL((point_feat_kernel<PP, ST>), dim3(groups), dim3(512), kPointLds, stream, a);   \
  } while (0)
  if (P == 128 && stats == 1) B3D_POINT_LAUNCH(128, 1);
  else if (P == 128 && stats == 2) B3D_POINT_LAUNCH(128, 2);
  else if (P == 128) B3D_POINT_LAUNCH(128, 0);
  else if (stats == 1) B3D_POINT_LAUNCH(64, 1);
  else if (stats == 2) B3D_POINT_LAUNCH(64, 2);
  else B3D_POINT_LAUNCH(64, 0);
#undef B3D_POINT_LAUNCH
  return launch_check("point_feat_kernel");
}


// ---- first and second moments of a point stack's first two layer inputs (train-mode BatchNorm statistics) -----------------
// The pre-activation of a kernel-1 convolution is affine in its input, so its batch mean / variance follow from the
// input's mean and second-moment matrix over all B * P points (b3d_bn_fold_moments).  For the first layer the input is
// the (transformed) point itself, K = C <= 4: plain VALU sums.  For the second layer it is h1 = relu(W1' x + b1') [64]:
// h1 is recomputed per 16-point tile with one fp32 MFMA per 16 features in the points x features orientation, which
// leaves a lane's 4 accumulator values on 4 points of one feature -- exactly the operand layout of h1^T h1 (both MFMA
// arguments, k = points), so the 64 x 64 matrix accumulates in 64 registers per wavefront with no data movement.
namespace b3d {
namespace {

constexpr int kMomGrid = 256, kMomK = 64, kMomRow = kMomK * kMomK + kMomK;

struct MomArgs {
  const float* x;       // [B, C, P]
  const float* trans;   // [B, 3, 3] or nullptr
  int B, C;
  const float* w1;      // folded first layer [64, C], [64]; nullptr: moments of the input
  const float* b1;
  float* part;          // [kMomGrid][kMomRow] (h1) or [kMomGrid][20] (input: 16 products, 4 sums)
};

__device__ __forceinline__ void load_point(const MomArgs& a, int P, int cloud, int p, float (&f)[4]) {
  const float* xp = a.x + ((long)cloud * a.C) * P + p;
  f[0] = f[1] = f[2] = f[3] = 0.f;
  for (int c = 0; c < a.C && c < 4; ++c) f[c] = xp[(long)c * P];
  if (a.trans) {
    const float* tr = a.trans + (long)cloud * 9;
    const float x0 = f[0], x1 = f[1], x2 = f[2];
    f[0] = x0 * tr[0] + x1 * tr[3] + x2 * tr[6];
    f[1] = x0 * tr[1] + x1 * tr[4] + x2 * tr[7];
    f[2] = x0 * tr[2] + x1 * tr[5] + x2 * tr[8];
  }
}

// The same point in two halves for software pipelining: raw loads (unconditional: clamped channel, a stand-in address when there
// is no transform -- a branch or a loop around a load makes hipcc drain vmcnt(0) behind it) and the arithmetic that consumes them.
struct RawPoint { float x[4]; float tr[9]; };
__device__ __forceinline__ void load_point_raw(const MomArgs& a, int P, int cloud, int p, RawPoint& r) {
  const float* xp = a.x + ((long)cloud * a.C) * P + p;
  const int cmax = a.C - 1;
#pragma unroll
  for (int c = 0; c < 4; ++c) r.x[c] = xp[(long)(c < cmax ? c : cmax) * P];
  const float* tr = a.trans ? a.trans + (long)cloud * 9 : a.x;          // (a.x: 9 readable floats, values unused)
#pragma unroll
  for (int i = 0; i < 9; ++i) r.tr[i] = tr[i];
}
__device__ __forceinline__ void finish_point(const MomArgs& a, const RawPoint& r, float (&f)[4]) {
#pragma unroll
  for (int c = 0; c < 4; ++c) f[c] = c < a.C ? r.x[c] : 0.f;
  if (a.trans) {
    const float x0 = f[0], x1 = f[1], x2 = f[2];
    f[0] = x0 * r.tr[0] + x1 * r.tr[3] + x2 * r.tr[6];
    f[1] = x0 * r.tr[1] + x1 * r.tr[4] + x2 * r.tr[7];
    f[2] = x0 * r.tr[2] + x1 * r.tr[5] + x2 * r.tr[8];
  }
}

__global__ __launch_bounds__(256) void point_moments_in_kernel(const MomArgs a, int P) {
  __shared__ float red[4][20];
  float acc[20];
#pragma unroll
  for (int i = 0; i < 20; ++i) acc[i] = 0.f;
  const long n = (long)a.B * P;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float f[4];
    load_point(a, P, (int)(i / P), (int)(i % P), f);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[4 * r + c] = fmaf(f[r], f[c], acc[4 * r + c]);
      acc[16 + r] += f[r];
    }
  }
#pragma unroll
  for (int i = 0; i < 20; ++i) {
    float v = acc[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < 20) a.part[blockIdx.x * 20 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// Round 4: EIGHT wavefronts per workgroup (two per SIMD) and the next tile's point loaded under the current tile's 20 MFMAs --
// with four wavefronts and the load in front of each tile's first MFMA the launch was a chain of exposed global round trips
// (57 us for 270 k points whose MFMAs take ~5 us).
constexpr int kMomWaves = 8;
template <int P>
__global__ __launch_bounds__(kMomWaves * 64) void point_moments_h1_kernel(const MomArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];      // [4][kMomRow]: wavefronts w and w + 4 share an image
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = lane & 15, q = lane >> 4;
  float wv[4], bv[4];
#pragma unroll
  for (int bi = 0; bi < 4; ++bi) {                            // B operand of the first layer: W1'[16 bi + n][q]
    wv[bi] = q < a.C ? a.w1[(16 * bi + n) * a.C + q] : 0.f;
    bv[bi] = a.b1[16 * bi + n];
  }
  v4f sec[4][4];
  float sm[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) sec[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
  constexpr int TPC = P / 16;
  const long tiles = (long)a.B * TPC;
  const long stride = (long)gridDim.x * kMomWaves;
  long tile = (long)blockIdx.x * kMomWaves + wave;
  RawPoint raw;
  {
    const long t0 = tile < tiles ? tile : 0;
    load_point_raw(a, P, (int)(t0 / TPC), (int)(t0 % TPC) * 16 + n, raw);
  }
  for (; tile < tiles; tile += stride) {
    float f[4];
    finish_point(a, raw, f);
    const float xa = q == 0 ? f[0] : q == 1 ? f[1] : q == 2 ? f[2] : f[3];     // A operand: x[point n][channel q]
    const long nxt = tile + stride < tiles ? tile + stride : tile;             // (the last tile is loaded again: no branch around loads)
    load_point_raw(a, P, (int)(nxt / TPC), (int)(nxt % TPC) * 16 + n, raw);    // consumed at the top of the next iteration
    __builtin_amdgcn_sched_barrier(0);                                         // (hipcc otherwise sinks these loads below the MFMAs)
    v4f h[4];
#pragma unroll
    for (int bi = 0; bi < 4; ++bi) {
      h[bi] = relu4(__builtin_amdgcn_mfma_f32_16x16x4f32(xa, wv[bi], v4f{bv[bi], bv[bi], bv[bi], bv[bi]}, 0, 0, 0));
      sm[bi] += (h[bi].x + h[bi].y) + (h[bi].z + h[bi].w);
    }
#pragma unroll
    for (int bi = 0; bi < 4; ++bi)
#pragma unroll
      for (int bj = 0; bj < 4; ++bj) sec[bi][bj] = mfma4(h[bi], h[bj], sec[bi][bj]);
  }
  // Wavefronts w and w + 4 share an LDS image (66 KB per workgroup instead of 133 KB: the launch has to find room next to the
  // kernels of the other encoder streams): w < 4 stores, then w >= 4 adds its own values onto the same addresses.
  float* mine = smem + (wave & 3) * kMomRow;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if ((wave >> 2) == half) {
#pragma unroll
      for (int bi = 0; bi < 4; ++bi) {
#pragma unroll
        for (int bj = 0; bj < 4; ++bj) {                      // element [16 bi + 4 q + j][16 bj + n]
          float* d = mine + (16 * bi + 4 * q) * kMomK + 16 * bj + n;
          if (half == 0) {
            d[0] = sec[bi][bj].x; d[kMomK] = sec[bi][bj].y; d[2 * kMomK] = sec[bi][bj].z; d[3 * kMomK] = sec[bi][bj].w;
          } else {
            d[0] += sec[bi][bj].x; d[kMomK] += sec[bi][bj].y; d[2 * kMomK] += sec[bi][bj].z; d[3 * kMomK] += sec[bi][bj].w;
          }
        }
        float v = sm[bi];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (q == 0) {
          if (half == 0) mine[kMomK * kMomK + 16 * bi + n] = v;
          else mine[kMomK * kMomK + 16 * bi + n] += v;
        }
      }
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < kMomRow; i += kMomWaves * 64)
    a.part[(long)blockIdx.x * kMomRow + i] = (smem[i] + smem[kMomRow + i]) + (smem[2 * kMomRow + i] + smem[3 * kMomRow + i]);
}

// partials -> mu [K], second [K, K] = sums / count in float64 (row: KP*KP products then KP sums).  A workgroup owns 16
// columns; 64 threads per column split the rows (a thread per column walking all 256 rows was 52 us of load latency).
__global__ __launch_bounds__(1024) void point_moments_finish_kernel(const float* part, int rows, int KP, int K, long long count,
                                                                    double* mu, double* second) {
  __shared__ double red[64][17];
  const int col = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + col;
  const int row_len = KP * KP + KP;
  double s = 0.0;
  if (i < row_len)
    for (int r = rl; r < rows; r += 64) s += (double)part[(long)r * row_len + i];
  red[rl][col] = s;
  __syncthreads();
  if (rl != 0 || i >= row_len) return;
  for (int r = 1; r < 64; ++r) s += red[r][col];
  s /= (double)count;
  if (i < KP * KP) {
    const int rr = i / KP, cc = i % KP;
    if (rr < K && cc < K) second[rr * K + cc] = s;
  } else if (i - KP * KP < K) {
    mu[i - KP * KP] = s;
  }
}

}  // namespace
}  // namespace b3d

extern "C" size_t b3d_point_moments_workspace_bytes(void) { return (size_t)kMomGrid * kMomRow * sizeof(float) + 256; }

extern "C" int b3d_point_moments(const b3d_linear* fold1, const float* x, const float* trans, int32_t B, int32_t C, int32_t P,
                                 void* workspace, size_t workspace_bytes, double* mu, double* second, b3d_stream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  B3D_REQUIRE(x && workspace && mu && second, "b3d_point_moments: null argument");
  B3D_REQUIRE(B >= 1 && C >= 1 && C <= 4 && (P == 64 || P == 128), "b3d_point_moments: B %d, C %d, P %d (C <= 4, P 64 or 128)", B, C, P);
  B3D_REQUIRE(!trans || C == 3, "b3d_point_moments: the input transform is 3x3");
  B3D_REQUIRE(!fold1 || (fold1->w && fold1->b), "b3d_point_moments: null layer");
  if (workspace_bytes < b3d_point_moments_workspace_bytes()) return fail(B3D_ERR_WORKSPACE, "b3d_point_moments: workspace too small");
  MomArgs a;
  a.x = x; a.trans = trans; a.B = B; a.C = C;
  a.w1 = fold1 ? (const float*)fold1->w : nullptr;
  a.b1 = fold1 ? (const float*)fold1->b : nullptr;
  a.part = (float*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  const long long count = (long long)B * P;
  if (!fold1) {
    hipLaunchKernelGGL(point_moments_in_kernel, dim3(kMomGrid), dim3(256), 0, stream, a, (int)P);
    B3D_TRY(launch_check("point_moments_in_kernel"));
    hipLaunchKernelGGL(point_moments_finish_kernel, dim3(2), dim3(1024), 0, stream, a.part, kMomGrid, 4, (int)C, count, mu, second);
  } else {
    constexpr int lds = 4 * kMomRow * (int)sizeof(float);
    // ~4 tiles of 16 points per wavefront: a small batch (radar: 757 clouds x 64 points) is a third of the chip, not 256 workgroups
    // whose only work is writing a 64 x 64 partial each
    const long tiles = (long)B * (P / 16);
    long grid = (tiles + 4 * kMomWaves - 1) / (4 * kMomWaves);
    grid = grid < 16 ? 16 : grid > kMomGrid ? kMomGrid : grid;
    if (P == 128) {
      B3D_TRY(set_lds(point_moments_h1_kernel<128>, lds));
      hipLaunchKernelGGL(point_moments_h1_kernel<128>, dim3((unsigned)grid), dim3(kMomWaves * 64), lds, stream, a);
    } else {
      B3D_TRY(set_lds(point_moments_h1_kernel<64>, lds));
      hipLaunchKernelGGL(point_moments_h1_kernel<64>, dim3((unsigned)grid), dim3(kMomWaves * 64), lds, stream, a);
    }
    B3D_TRY(launch_check("point_moments_h1_kernel"));
    hipLaunchKernelGGL(point_moments_finish_kernel, dim3((kMomRow + 15) / 16), dim3(1024), 0, stream, a.part, (int)grid, kMomK, kMomK,
                       count, mu, second);
  }
  return launch_check("point_moments_finish_kernel");
}


// ---- train-mode BatchNorm bookkeeping of the point stacks, fused (the PyTorch form is ~25 tiny launches per layer) --------
namespace b3d {
namespace {

// running statistics as nn.BatchNorm1d updates them in a train-mode forward (unbiased variance, momentum or the
// cumulative average when momentum < 0, num_batches_tracked)
__device__ __forceinline__ void bn_track(float* running_mean, float* running_var, int o, double mean, double var_biased,
                                         long long count, float momentum, long long nbt_after) {
  if (!running_mean) return;
  const double mom = momentum >= 0.f ? (double)momentum : 1.0 / (double)nbt_after;
  const double unbiased = var_biased * ((double)count / (double)(count > 1 ? count - 1 : 1));
  running_mean[o] = (float)((1.0 - mom) * (double)running_mean[o] + mom * (double)(float)mean);
  running_var[o] = (float)((1.0 - mom) * (double)running_var[o] + mom * (double)(float)unbiased);
}

__global__ void bn_tick_kernel(long long* nbt) { *nbt += 1; }

// z = W h + b over all points: mean(z) = W mu + b, var(z)_o = W_o Cov W_o^T with Cov = second - mu mu^T (float64); then the
// BatchNorm of THIS batch folded into the layer: wf = W * scale, bf = b * scale + shift.
struct BnFoldArgs {
  const double* mu;       // [C]
  const double* second;   // [C, C] = E[h h^T]
  int C, O;
  const float *W, *b, *gamma, *beta;
  float *running_mean, *running_var;
  long long* nbt;
  float momentum, eps;
  long long count;
  float *wf, *bf;
};
__global__ __launch_bounds__(256) void bn_fold_moments_kernel(const BnFoldArgs a) {
  // one wavefront per output channel, 4 per workgroup; lane i holds row i of Cov . w
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int o = blockIdx.x * 4 + wave;
  if (o >= a.O) return;
  const float* w = a.W + (size_t)o * a.C;
  double mean_part = 0.0, var_part = 0.0;
  if (lane < a.C) {
    const double mi = a.mu[lane], wi = (double)w[lane];
    double t = 0.0;
    // eight independent (second, mu, w) triples per round trip, accumulated in column order (one dependent load per column took
    // 17 us for the 64-wide second layer of a point stack)
    int j = 0;
    for (; j + 8 <= a.C; j += 8) {
      double sv[8], mv[8], wv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { sv[u] = a.second[lane * a.C + j + u]; mv[u] = a.mu[j + u]; wv[u] = (double)w[j + u]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) t += (sv[u] - mi * mv[u]) * wv[u];
    }
    for (; j < a.C; ++j) t += (a.second[lane * a.C + j] - mi * a.mu[j]) * (double)w[j];
    mean_part = wi * mi;
    var_part = wi * t;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    mean_part += __shfl_xor(mean_part, off);
    var_part += __shfl_xor(var_part, off);
  }
  const double mean = mean_part + (double)a.b[o];
  const double var = var_part < 0.0 ? 0.0 : var_part;
  const float meanf = (float)mean, varf = (float)var;
  const float scale = a.gamma[o] / sqrtf(varf + a.eps);
  const float shift = a.beta[o] - meanf * scale;
  if (lane < a.C) a.wf[(size_t)o * a.C + lane] = w[lane] * scale;
  if (lane == 0) {
    a.bf[o] = a.b[o] * scale + shift;
    bn_track(a.running_mean, a.running_var, o, mean, var, a.count, a.momentum, (a.nbt && a.momentum < 0.f) ? *a.nbt + 1 : 1);
    // with a fixed momentum nobody reads the counter: it is advanced here; the cumulative average reads it in every
    // workgroup, there the host launches bn_tick_kernel behind this kernel
    if (o == 0 && a.nbt && a.momentum >= 0.f) *a.nbt += 1;
  }
}

// last layer: per-cloud max / min / sum / sum of squares of the raw conv output -> batch statistics -> y = BN(max or min)
constexpr int kBnMinMaxChunks = 64;     // row chunks of the statistics pass (a multiple of 8)
struct BnMinMaxArgs {
  const float *vmax, *vmin, *vsum, *vsq;   // [B, F]
  int B, F, relu;
  long long count;
  const float *gamma, *beta;
  float *running_mean, *running_var;
  long long* nbt;
  float momentum, eps;
  double* part;                            // [chunks][2][F]
  int chunks;
  float* y;                                // [B, F]
};
// (round 4: 64 row chunks instead of 32 and four independent row loads per iteration -- the two launches were 21 + 30 us of
// dependent L2 round trips on 128 workgroups for 26 MB of traffic)
__global__ __launch_bounds__(256) void bn_minmax_partial_kernel(const BnMinMaxArgs a) {
  const int f = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
  if (f >= a.F) return;
  const int per = (a.B + a.chunks - 1) / a.chunks;
  const int b0 = c * per, b1 = min(a.B, b0 + per);
  double s = 0.0, q = 0.0;
  int b = b0;
  for (; b + 4 <= b1; b += 4) {
    float vs[4], vq[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { vs[u] = a.vsum[(size_t)(b + u) * a.F + f]; vq[u] = a.vsq[(size_t)(b + u) * a.F + f]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) { s += (double)vs[u]; q += (double)vq[u]; }
  }
  for (; b < b1; ++b) { s += (double)a.vsum[(size_t)b * a.F + f]; q += (double)a.vsq[(size_t)b * a.F + f]; }
  a.part[((size_t)c * 2 + 0) * a.F + f] = s;
  a.part[((size_t)c * 2 + 1) * a.F + f] = q;
}
__global__ __launch_bounds__(256) void bn_minmax_apply_kernel(const BnMinMaxArgs a) {
  const int f = blockIdx.x * 256 + threadIdx.x;
  if (f >= a.F) return;
  double s = 0.0, q = 0.0;
  for (int c = 0; c < a.chunks; c += 8) {                    // (chunks is a multiple of 8: sixteen independent loads per round trip)
    double ps[8], pq[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { ps[u] = a.part[((size_t)(c + u) * 2 + 0) * a.F + f]; pq[u] = a.part[((size_t)(c + u) * 2 + 1) * a.F + f]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) { s += ps[u]; q += pq[u]; }
  }
  const double mean = s / (double)a.count;
  double var = q / (double)a.count - mean * mean;
  if (var < 0.0) var = 0.0;
  const double scale = (double)a.gamma[f] / sqrt(var + (double)a.eps);
  const double shift = (double)a.beta[f] - mean * scale;
  if (blockIdx.y == 0) {
    const long long nbt_after = (a.nbt && a.momentum < 0.f) ? *a.nbt + 1 : 1;
    bn_track(a.running_mean, a.running_var, f, (double)(float)mean, (double)(float)var, a.count, a.momentum, nbt_after);
    if (f == 0 && a.nbt && a.momentum >= 0.f) *a.nbt += 1;       // fixed momentum: nobody reads the counter (see bn_fold_moments_kernel)
  }
  const int per = (a.B + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = min(a.B, b0 + per);
  const float* const ext_of = scale > 0.0 ? a.vmax : a.vmin;  // max over the points of an increasing map, min of a decreasing one
  int b = b0;
  for (; b + 4 <= b1; b += 4) {
    float e[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) e[u] = ext_of[(size_t)(b + u) * a.F + f];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float v = (float)((double)e[u] * scale + shift);
      if (a.relu) v = relu1(v);
      a.y[(size_t)(b + u) * a.F + f] = v;
    }
  }
  for (; b < b1; ++b) {
    const size_t o = (size_t)b * a.F + f;
    float v = (float)((double)ext_of[o] * scale + shift);
    if (a.relu) v = relu1(v);
    a.y[o] = v;
  }
}

}  // namespace
}  // namespace b3d

extern "C" int b3d_bn_fold_moments(const double* mu, const double* second, int32_t C, const float* W, const float* b, int32_t O,
                                   const float* gamma, const float* beta, float* running_mean, float* running_var,
                                   int64_t* num_batches_tracked, float momentum, float eps, int64_t count, float* wf, float* bf,
                                   b3d_stream stream_) {
  B3D_REQUIRE(mu && second && W && b && gamma && beta && wf && bf, "b3d_bn_fold_moments: null argument");
  B3D_REQUIRE(C >= 1 && C <= 64 && O >= 1 && count >= 1, "b3d_bn_fold_moments: C %d (<= 64), O %d", (int)C, (int)O);
  B3D_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "b3d_bn_fold_moments: running statistics come as a pair");
  BnFoldArgs a{mu, second, C, O, W, b, gamma, beta, running_mean, running_var, (long long*)num_batches_tracked, momentum, eps,
               (long long)count, wf, bf};
  hipLaunchKernelGGL(bn_fold_moments_kernel, dim3((unsigned)((O + 3) / 4)), dim3(256), 0, (hipStream_t)stream_, a);
  B3D_TRY(launch_check("bn_fold_moments_kernel"));
  if (num_batches_tracked && momentum < 0.f) {
    hipLaunchKernelGGL(bn_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream_, (long long*)num_batches_tracked);
    B3D_TRY(launch_check("bn_tick_kernel"));
  }
  return B3D_OK;
}

extern "C" size_t b3d_bn_minmax_workspace_bytes(int32_t F) { return (size_t)kBnMinMaxChunks * 2 * (size_t)F * sizeof(double) + 256; }

extern "C" int b3d_bn_minmax_apply(const float* vmax, const float* vmin, const float* vsum, const float* vsq, int32_t B, int32_t F,
                                   int64_t count, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                   int64_t* num_batches_tracked, float momentum, float eps, int32_t relu, void* workspace,
                                   size_t workspace_bytes, float* y, b3d_stream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  B3D_REQUIRE(vmax && vmin && vsum && vsq && gamma && beta && workspace && y, "b3d_bn_minmax_apply: null argument");
  B3D_REQUIRE(B >= 1 && F >= 1 && count >= 1, "b3d_bn_minmax_apply: B %d, F %d", (int)B, (int)F);
  if (workspace_bytes < b3d_bn_minmax_workspace_bytes(F)) return fail(B3D_ERR_WORKSPACE, "b3d_bn_minmax_apply: workspace too small");
  BnMinMaxArgs a{vmax, vmin, vsum, vsq, B, F, relu, (long long)count, gamma, beta, running_mean, running_var,
                 (long long*)num_batches_tracked, momentum, eps, (double*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255), kBnMinMaxChunks, y};
  const unsigned fb = (unsigned)((F + 255) / 256);
  hipLaunchKernelGGL(bn_minmax_partial_kernel, dim3(fb, kBnMinMaxChunks), dim3(256), 0, stream, a);
  B3D_TRY(launch_check("bn_minmax_partial_kernel"));
  hipLaunchKernelGGL(bn_minmax_apply_kernel, dim3(fb, 128), dim3(256), 0, stream, a);
  B3D_TRY(launch_check("bn_minmax_apply_kernel"));
  if (num_batches_tracked && momentum < 0.f) {
    hipLaunchKernelGGL(bn_tick_kernel, dim3(1), dim3(1), 0, stream, (long long*)num_batches_tracked);
    B3D_TRY(launch_check("bn_tick_kernel"));
  }
  return B3D_OK;
}


// ---- the train-mode point stack as ONE call (b3d_point_stack_train) --------------------------------------------------------
// Round 5.  The sequence above (input moments -> finish -> fold -> h1 moments -> finish -> fold -> pack -> point_feat_stats ->
// minmax partial -> minmax apply: ten launches per stack, thirty per training step, each a chain of a few dependent L2 round trips)
// with
//   * the first layer's statistics finished and folded by the LAST workgroup of the input-moments launch (C <= 4: twenty sums and
//     64 channels of a 4 x 4 quadratic form -- no launch of their own);
//   * the 64 x 64 second moments of h1 on v_mfma_f32_16x16x32_bf16 (bf16x6, two 16-point tiles per instruction group: 96 matrix
//     instructions of 16 cycles per 32 points instead of 128 of 32 cycles on the exact-fp32 form);
//   * sign(gamma) of the last BatchNorm folded into the last convolution (exact), so that the running MAXIMUM serves either sign
//     of the scale: the minimum is neither computed nor stored;
//   * no "apply" pass: the call returns the per-cloud extremes and the batch's BatchNorm as a per-feature affine map
//     (out_scale >= 0, out_shift), which the consumer -- the first Linear of the head, b3d_fc_bn_forward -- applies while it stages
//     its input.
namespace b3d {
namespace {

constexpr int kStackTickets = 16;        // unsigned words at the head of the workspace: [0] input moments, [1 + fb] statistics

struct StackInArgs {
  MomArgs m;
  int P;
  const float *W1, *b1, *gamma, *beta;
  float *running_mean, *running_var;
  long long* nbt;
  float momentum, eps;
  long long count;
  float *wf, *bf;            // [64, C], [64]: the first layer with the batch's BatchNorm folded in
  unsigned* ticket;
};

__device__ __forceinline__ bool last_arriver(unsigned* ticket, unsigned expected, int* s_flag) {
  // MI355X_MICROARCH.md hand-off: every storing wave drains, workgroup barrier, one lane's agent-scope release in front of the
  // ticket; the last arriver acquires before it reads the others' partials
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = (t == expected - 1u) ? 1 : 0;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    *s_flag = last;
  }
  __syncthreads();
  return *s_flag != 0;
}

__global__ __launch_bounds__(256) void stack_moments_in_kernel(const StackInArgs a) {
  __shared__ float red[4][20];
  __shared__ double part[12][20];
  __shared__ double fin[20];
  __shared__ int s_last;
  float acc[20];
#pragma unroll
  for (int i = 0; i < 20; ++i) acc[i] = 0.f;
  const int P = a.P;
  const long n = (long)a.m.B * P;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float f[4];
    load_point(a.m, P, (int)(i / P), (int)(i % P), f);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[4 * r + c] = fmaf(f[r], f[c], acc[4 * r + c]);
      acc[16 + r] += f[r];
    }
  }
#pragma unroll
  for (int i = 0; i < 20; ++i) {
    float v = acc[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < 20) a.m.part[blockIdx.x * 20 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
  if (!last_arriver(a.ticket, gridDim.x, &s_last)) return;
  // ---- the last workgroup: sums over the workgroups' partials (float64, fixed order), then the first layer's BatchNorm ----
  const int rows = (int)gridDim.x;
  if (threadIdx.x < 240) {
    const int c = threadIdx.x % 20, g = threadIdx.x / 20;
    double s = 0.0;
    for (int r = g; r < rows; r += 12) s += (double)a.m.part[r * 20 + c];
    part[g][c] = s;
  }
  __syncthreads();
  if (threadIdx.x < 20) {
    double s = 0.0;
#pragma unroll
    for (int g = 0; g < 12; ++g) s += part[g][threadIdx.x];
    fin[threadIdx.x] = s / (double)a.count;            // [0..15] = E[x_r x_c], [16..19] = E[x_r]
  }
  __syncthreads();
  const long long nbt_before = a.nbt ? *a.nbt : 0;
  __syncthreads();
  if (threadIdx.x < 64) {
    const int o = threadIdx.x, C = a.m.C;
    const float* w = a.W1 + o * C;
    double mean = (double)a.b1[o], var = 0.0;
    for (int i = 0; i < C; ++i) {
      double t = 0.0;
      for (int j = 0; j < C; ++j) t += (fin[4 * i + j] - fin[16 + i] * fin[16 + j]) * (double)w[j];
      mean += (double)w[i] * fin[16 + i];
      var += (double)w[i] * t;
    }
    if (var < 0.0) var = 0.0;
    const float meanf = (float)mean, varf = (float)var;
    const float scale = a.gamma[o] / sqrtf(varf + a.eps);
    const float shift = a.beta[o] - meanf * scale;
    for (int i = 0; i < C; ++i) a.wf[o * C + i] = w[i] * scale;
    a.bf[o] = a.b1[o] * scale + shift;
    bn_track(a.running_mean, a.running_var, o, mean, var, a.count, a.momentum, nbt_before + 1);
  }
  if (threadIdx.x == 0) {
    if (a.nbt) *a.nbt = nbt_before + 1;
    *a.ticket = 0u;                                      // re-armed for the next call (stream order)
  }
}

// 64 x 64 second moments and the mean of h1 = relu(W1' x + b1') over all points, bf16x6.  A wavefront takes PAIRS of 16-point
// tiles: h1 of either tile comes out of one exact-fp32 MFMA per 16 features with a lane's four accumulator values on four POINTS
// of one feature; the two tiles' values of a lane are the eight k-slots of a v_mfma_f32_16x16x32_bf16 operand (k = points), the
// same fragment serving as either argument of h1^T h1.
template <int P>
__global__ __launch_bounds__(kMomWaves * 64) void stack_moments_h1_kernel(const MomArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];      // [4][kMomRow]: wavefronts w and w + 4 share an image
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = lane & 15, q = lane >> 4;
  float wv[4], bv[4];
#pragma unroll
  for (int bi = 0; bi < 4; ++bi) {
    wv[bi] = q < a.C ? a.w1[(16 * bi + n) * a.C + q] : 0.f;
    bv[bi] = a.b1[16 * bi + n];
  }
  v4f sec[4][4];
  float sm[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) sec[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
  constexpr int PPC = P / 32;                                  // tile pairs per cloud
  const long pairs = (long)a.B * PPC;
  const long stride = (long)gridDim.x * kMomWaves;
  long pr = (long)blockIdx.x * kMomWaves + wave;
  RawPoint ra, rb;
  {
    const long t0 = pr < pairs ? pr : 0;
    const int cloud = (int)(t0 / PPC), p0 = (int)(t0 % PPC) * 32 + n;
    load_point_raw(a, P, cloud, p0, ra);
    load_point_raw(a, P, cloud, p0 + 16, rb);
  }
  for (; pr < pairs; pr += stride) {
    float fa[4], fb[4];
    finish_point(a, ra, fa);
    finish_point(a, rb, fb);
    const float xa = q == 0 ? fa[0] : q == 1 ? fa[1] : q == 2 ? fa[2] : fa[3];
    const float xb = q == 0 ? fb[0] : q == 1 ? fb[1] : q == 2 ? fb[2] : fb[3];
    const long nxt = pr + stride < pairs ? pr + stride : pr;        // (the last pair is loaded again: no branch around loads)
    {
      const int cloud = (int)(nxt / PPC), p0 = (int)(nxt % PPC) * 32 + n;
      load_point_raw(a, P, cloud, p0, ra);
      load_point_raw(a, P, cloud, p0 + 16, rb);
    }
    __builtin_amdgcn_sched_barrier(0);
    Bf3 op[4];
#pragma unroll
    for (int bi = 0; bi < 4; ++bi) {
      const v4f bias = v4f{bv[bi], bv[bi], bv[bi], bv[bi]};
      const v4f ha = relu4(__builtin_amdgcn_mfma_f32_16x16x4f32(xa, wv[bi], bias, 0, 0, 0));
      const v4f hb = relu4(__builtin_amdgcn_mfma_f32_16x16x4f32(xb, wv[bi], bias, 0, 0, 0));
      sm[bi] += ((ha.x + ha.y) + (ha.z + ha.w)) + ((hb.x + hb.y) + (hb.z + hb.w));
      op[bi] = bf_split(ha, hb);
    }
#pragma unroll
    for (int bi = 0; bi < 4; ++bi)
#pragma unroll
      for (int bj = 0; bj < 4; ++bj) sec[bi][bj] = bf_mfma6(op[bi], op[bj], sec[bi][bj]);
  }
  float* mine = smem + (wave & 3) * kMomRow;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if ((wave >> 2) == half) {
#pragma unroll
      for (int bi = 0; bi < 4; ++bi) {
#pragma unroll
        for (int bj = 0; bj < 4; ++bj) {                      // element [16 bi + 4 q + j][16 bj + n]
          float* d = mine + (16 * bi + 4 * q) * kMomK + 16 * bj + n;
          if (half == 0) {
            d[0] = sec[bi][bj].x; d[kMomK] = sec[bi][bj].y; d[2 * kMomK] = sec[bi][bj].z; d[3 * kMomK] = sec[bi][bj].w;
          } else {
            d[0] += sec[bi][bj].x; d[kMomK] += sec[bi][bj].y; d[2 * kMomK] += sec[bi][bj].z; d[3 * kMomK] += sec[bi][bj].w;
          }
        }
        float v = sm[bi];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (q == 0) {
          if (half == 0) mine[kMomK * kMomK + 16 * bi + n] = v;
          else mine[kMomK * kMomK + 16 * bi + n] += v;
        }
      }
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < kMomRow; i += kMomWaves * 64)
    a.part[(long)blockIdx.x * kMomRow + i] = (smem[i] + smem[kMomRow + i]) + (smem[2 * kMomRow + i] + smem[3 * kMomRow + i]);
}

// Statistics of the last layer from the per-cloud sums (float64), one launch: row chunks -> partials; the LAST workgroup of each
// 256-feature block forms mean / variance, updates the running statistics and writes the batch's BatchNorm as an affine map of the
// sign-folded extreme: out_scale = |gamma| / sqrt(var + eps) >= 0, out_shift = beta - |gamma| mean' / sqrt(var + eps) with
// mean' = sign(gamma) mean (the statistics arrive for z' = sign(gamma) z; var(z') = var(z)).
struct StackStatArgs {
  const float *vsum, *vsq;                 // [B, F] of z'
  int B, F;
  long long count;
  const float *gamma, *beta;
  float *running_mean, *running_var;
  long long* nbt;
  float momentum, eps;
  double* part;                            // [chunks][2][F]
  int chunks;
  float *out_scale, *out_shift;            // [F]
  unsigned* ticket;                        // [1 + feature block] per-block arrival counters, [1 + fb count] = launch counter
};
__global__ __launch_bounds__(256) void stack_stats_kernel(const StackStatArgs a) {
  __shared__ int s_last;
  const int f = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
  const int per = (a.B + a.chunks - 1) / a.chunks;
  const int b0 = c * per, b1 = min(a.B, b0 + per);
  if (f < a.F) {
    double s = 0.0, q = 0.0;
    int b = b0;
    for (; b + 4 <= b1; b += 4) {
      float vs[4], vq[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { vs[u] = a.vsum[(size_t)(b + u) * a.F + f]; vq[u] = a.vsq[(size_t)(b + u) * a.F + f]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) { s += (double)vs[u]; q += (double)vq[u]; }
    }
    for (; b < b1; ++b) { s += (double)a.vsum[(size_t)b * a.F + f]; q += (double)a.vsq[(size_t)b * a.F + f]; }
    a.part[((size_t)c * 2 + 0) * a.F + f] = s;
    a.part[((size_t)c * 2 + 1) * a.F + f] = q;
  }
  if (!last_arriver(a.ticket + 1 + blockIdx.x, (unsigned)a.chunks, &s_last)) return;
  const long long nbt_before = a.nbt ? *a.nbt : 0;
  if (f < a.F) {
    double s = 0.0, q = 0.0;
    for (int cc = 0; cc < a.chunks; cc += 8) {                 // (chunks is a multiple of 8: sixteen independent loads per round trip)
      double ps[8], pq[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { ps[u] = a.part[((size_t)(cc + u) * 2 + 0) * a.F + f]; pq[u] = a.part[((size_t)(cc + u) * 2 + 1) * a.F + f]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) { s += ps[u]; q += pq[u]; }
    }
    const double meanp = s / (double)a.count;                  // of z' = sign(gamma) z
    double var = q / (double)a.count - meanp * meanp;
    if (var < 0.0) var = 0.0;
    const double g = (double)a.gamma[f];
    const double sgn = g < 0.0 ? -1.0 : 1.0;
    const double scale = (g < 0.0 ? -g : g) / sqrt(var + (double)a.eps);
    a.out_scale[f] = (float)scale;
    a.out_shift[f] = (float)((double)a.beta[f] - meanp * scale);
    bn_track(a.running_mean, a.running_var, f, (double)(float)(sgn * meanp), (double)(float)var, a.count, a.momentum, nbt_before + 1);
  }
  // the counter moves once, after every feature block has read it
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    a.ticket[1 + blockIdx.x] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(a.ticket + 1 + gridDim.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == gridDim.x - 1u) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      if (a.nbt) *a.nbt = nbt_before + 1;
      a.ticket[1 + gridDim.x] = 0u;
    }
  }
}

__global__ __launch_bounds__(256) void affine_kernel(const float* __restrict__ y, const float* __restrict__ scale,
                                                     const float* __restrict__ shift, long total, int N, int relu, float* __restrict__ out) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int n = (int)(i % N);
    const float v = fmaf(y[i], scale[n], shift[n]);
    out[i] = relu ? relu1(v) : v;
  }
}

struct StackWs {
  unsigned* ticket;          // [kStackTickets]: caller's, zero on entry, left zero
  float* mom_part;           // [kMomGrid][kMomRow]
  double *mu, *second;       // [64], [64 x 64]
  float *wf1, *bf1, *wf2, *bf2;
  float *vsum, *vsq;         // [B, 1024]
  double* stat_part;         // [kBnMinMaxChunks][2][1024]
  float* images;             // PointSeq
};
bool carve_stack(StackWs& w, void* ws, size_t bytes, int B, size_t* need) {
  Carver c(ws, bytes);
  w.mom_part = c.take<float>((size_t)kMomGrid * kMomRow);
  w.mu = c.take<double>(64);
  w.second = c.take<double>(64 * 64);
  w.wf1 = c.take<float>(64 * 4);
  w.bf1 = c.take<float>(64);
  w.wf2 = c.take<float>(128 * 64);
  w.bf2 = c.take<float>(128);
  w.vsum = c.take<float>((size_t)B * kPointFeat);
  w.vsq = c.take<float>((size_t)B * kPointFeat);
  w.stat_part = c.take<double>((size_t)kBnMinMaxChunks * 2 * kPointFeat);
  w.images = c.take<float>((size_t)PointSeq::TOTAL_FLOATS + 64);
  if (need) *need = c.off + 256;
  return c.ok();
}

}  // namespace
}  // namespace b3d

extern "C" size_t b3d_point_stack_train_workspace_bytes(int32_t B) {
  if (B < 0) B = 0;
  StackWs w{};
  size_t need = 0;
  carve_stack(w, nullptr, 0, B, &need);
  return need;
}

extern "C" int b3d_point_stack_train(const b3d_linear* conv, const b3d_batchnorm* bn, const float* x, const float* trans, int32_t B,
                                     int32_t C, int32_t P, void* tickets, void* workspace, size_t workspace_bytes, float* ext,
                                     float* out_scale, float* out_shift, b3d_stream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  B3D_REQUIRE(conv && bn && x && tickets && workspace && ext && out_scale && out_shift, "b3d_point_stack_train: null argument");
  B3D_REQUIRE(B >= 1 && (long long)B * P > 1 && C >= 1 && C <= 4 && (P == 64 || P == 128),
              "b3d_point_stack_train: B %d, C %d, P %d (more than one point, C <= 4, P 64 or 128)", (int)B, (int)C, (int)P);
  B3D_REQUIRE(!trans || C == 3, "b3d_point_stack_train: the input transform is 3x3");
  for (int i = 0; i < 3; ++i) {
    B3D_REQUIRE(conv[i].w && conv[i].b && bn[i].gamma && bn[i].beta, "b3d_point_stack_train: layer %d: null parameter", i);
    B3D_REQUIRE((bn[i].running_mean == nullptr) == (bn[i].running_var == nullptr), "b3d_point_stack_train: running statistics come as a pair");
  }
  StackWs w{};
  if (!carve_stack(w, workspace, workspace_bytes, B, nullptr)) return fail(B3D_ERR_WORKSPACE, "b3d_point_stack_train: workspace too small");
  w.ticket = (unsigned*)tickets;
  const long long count = (long long)B * P;
  // 1. input moments, their sums and the first layer's BatchNorm: one launch
  {
    StackInArgs a;
    memset(&a, 0, sizeof(a));
    a.m.x = x; a.m.trans = trans; a.m.B = B; a.m.C = C; a.m.part = w.mom_part;
    a.P = P;
    a.W1 = (const float*)conv[0].w; a.b1 = (const float*)conv[0].b; a.gamma = bn[0].gamma; a.beta = bn[0].beta;
    a.running_mean = bn[0].running_mean; a.running_var = bn[0].running_var; a.nbt = (long long*)bn[0].num_batches_tracked;
    a.momentum = bn[0].momentum; a.eps = bn[0].eps; a.count = count;
    a.wf = w.wf1; a.bf = w.bf1; a.ticket = w.ticket;
    long grid = ((long)B * P + 1023) / 1024;                   // >= 4 points per thread
    grid = grid < 1 ? 1 : grid > kMomGrid ? kMomGrid : grid;
    hipLaunchKernelGGL(stack_moments_in_kernel, dim3((unsigned)grid), dim3(256), 0, stream, a);
    B3D_TRY(launch_check("stack_moments_in_kernel"));
  }
  // 2. second moments of h1 (bf16x6), 3. their sums, 4. the second layer's BatchNorm
  {
    MomArgs a;
    a.x = x; a.trans = trans; a.B = B; a.C = C; a.w1 = w.wf1; a.b1 = w.bf1; a.part = w.mom_part;
    constexpr int lds = 4 * kMomRow * (int)sizeof(float);
    const long pairs = (long)B * (P / 32);
    long grid = (pairs + 2 * kMomWaves - 1) / (2 * kMomWaves);  // ~2 pairs (64 points) per wavefront at least
    grid = grid < 16 ? 16 : grid > kMomGrid ? kMomGrid : grid;
    if (P == 128) {
      B3D_TRY(set_lds(stack_moments_h1_kernel<128>, lds));
      hipLaunchKernelGGL(stack_moments_h1_kernel<128>, dim3((unsigned)grid), dim3(kMomWaves * 64), lds, stream, a);
    } else {
      B3D_TRY(set_lds(stack_moments_h1_kernel<64>, lds));
      hipLaunchKernelGGL(stack_moments_h1_kernel<64>, dim3((unsigned)grid), dim3(kMomWaves * 64), lds, stream, a);
    }
    B3D_TRY(launch_check("stack_moments_h1_kernel"));
    hipLaunchKernelGGL(point_moments_finish_kernel, dim3((kMomRow + 15) / 16), dim3(1024), 0, stream, (const float*)w.mom_part, (int)grid,
                       kMomK, kMomK, count, w.mu, w.second);
    B3D_TRY(launch_check("point_moments_finish_kernel"));
    B3D_TRY(b3d_bn_fold_moments(w.mu, w.second, 64, (const float*)conv[1].w, (const float*)conv[1].b, 128, bn[1].gamma, bn[1].beta,
                                bn[1].running_mean, bn[1].running_var, bn[1].num_batches_tracked, bn[1].momentum, bn[1].eps, count,
                                w.wf2, w.bf2, stream_));
  }
  // 5. images + the three layers over all points: per cloud the maximum, sum and sum of squares of z' = sign(gamma3) conv3(h2)
  {
    b3d_linear folded[3] = {{w.wf1, w.bf1}, {w.wf2, w.bf2}, {conv[2].w, conv[2].b}};
    B3D_TRY(point_feat_launch(folded, x, trans, B, C, P, 0, w.images, ((size_t)PointSeq::TOTAL_FLOATS + 64) * sizeof(float), ext, nullptr,
                              w.vsum, w.vsq, stream, bn[2].gamma));
  }
  // 6. the last layer's statistics -> out_scale / out_shift (+ running statistics)
  {
    StackStatArgs a;
    memset(&a, 0, sizeof(a));
    a.vsum = w.vsum; a.vsq = w.vsq; a.B = B; a.F = kPointFeat; a.count = count;
    a.gamma = bn[2].gamma; a.beta = bn[2].beta; a.running_mean = bn[2].running_mean; a.running_var = bn[2].running_var;
    a.nbt = (long long*)bn[2].num_batches_tracked; a.momentum = bn[2].momentum; a.eps = bn[2].eps;
    a.part = w.stat_part; a.chunks = kBnMinMaxChunks; a.out_scale = out_scale; a.out_shift = out_shift; a.ticket = w.ticket;
    static_assert(2 + kPointFeat / 256 <= kStackTickets, "ticket words");
    hipLaunchKernelGGL(stack_stats_kernel, dim3(kPointFeat / 256, kBnMinMaxChunks), dim3(256), 0, stream, a);
    B3D_TRY(launch_check("stack_stats_kernel"));
  }
  return B3D_OK;
}

extern "C" int b3d_affine(const float* y, const float* scale, const float* shift, int32_t B, int32_t N, int32_t relu, float* out,
                          b3d_stream stream_) {
  B3D_REQUIRE(B >= 0 && N > 0, "b3d_affine: bad shape");
  if (B == 0) return B3D_OK;
  B3D_REQUIRE(y && scale && shift && out, "b3d_affine: null argument");
  const long total = (long)B * N;
  long blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(affine_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, y, scale, shift, total, N, (int)relu, out);
  return launch_check("affine_kernel");
}
