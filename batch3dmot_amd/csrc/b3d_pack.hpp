// Weight image packing (see b3d_dev.hpp "packed weight image geometry").
#pragma once
#include <hip/hip_runtime.h>

namespace b3d {

struct PackDesc {
  const float* w;    // forward weight [out, in] row-major
  const float* b;    // bias [out] or nullptr (never used for transposed images)
  float* dst;        // image base (chunked layout)
  int N, K;          // true rows / cols OF THE IMAGE (transposed: N = in, K = out of the forward layer)
  int NP, KP;        // padded to multiples of 16
  int transposed;    // 1: image of W^T (data-gradient operand); 2 / 3: not an image -- fill N ints at dst with
                     // 0, 1, 2, ... / with zeros (index and zero rows of the streaming weight gradient)
};

constexpr int kPackMax = 64;      // 3 KB of kernel arguments: one launch packs a whole model's images
struct PackArgs {
  int n;
  PackDesc d[kPackMax];
};

int pack_images(const PackDesc* descs, int n, hipStream_t stream);

// Helper: descriptor for layer LI of a LayerSeq placed at `base`.
template <class Seq>
inline PackDesc pack_desc(int li, float* base, const float* w, const float* b, int N, int K, bool transposed) {
  PackDesc d;
  d.w = w; d.b = transposed ? nullptr : b;
  d.dst = base + Seq::layer_off(li);
  d.N = N; d.K = K; d.NP = Seq::np(li); d.KP = Seq::kp(li);
  d.transposed = transposed ? 1 : 0;
  return d;
}

inline PackDesc fill_desc(void* dst, int count, bool iota) {
  PackDesc d;
  d.w = nullptr; d.b = nullptr; d.dst = (float*)dst;
  d.N = count; d.K = 0; d.NP = 0; d.KP = 0;
  d.transposed = iota ? 2 : 3;
  return d;
}

}  // namespace b3d
