// Weight image packing (see b3d_dev.hpp "packed weight image geometry").
#pragma once
#include <hip/hip_runtime.h>
#include "b3d_dev.hpp"

namespace b3d {

struct PackDesc {
  const float* w;    // forward weight [out, in] row-major
  const float* b;    // bias [out] or nullptr (never used for transposed images)
  float* dst;        // image base (chunked layout)
  int N, K;          // true rows / cols OF THE IMAGE (transposed: N = in, K = out of the forward layer)
  int NP, KP;        // padded to multiples of 16
  int transposed;    // 1: image of W^T (data-gradient operand); 2 / 3: not an image -- fill N ints at dst with
                     // 0, 1, 2, ... / with zeros (index and zero rows of the streaming weight gradient)
  // A descriptor may cover only image rows [row0, row0 + nrows) (several slices of different weight
  // matrices stacked into one image) and read a column slice of its source: element (r, c) of the
  // slice is w[r * ld + c] (transposed: w[c * ld + r]).  N counts the slice's valid rows.
  int ld;            // leading dimension of the source matrix
  int row0, nrows;   // image rows covered (multiples of 16)
  int col0;          // >= 0 with `partial`: the slice lands at image columns [col0, col0 + K)
  int partial;       // 1: write only the slice's nrows x K cells (w == nullptr: zeros); other cells belong to other
                     //    descriptors of the same image
  int bf;            // image format of the layer (b3d_dev.hpp): 0 fp32, 1 bf16x3
  const float* row_sign;   // [N] or nullptr: row r of the image (weights and bias; not for transposed images) is multiplied by
                           // -1 where row_sign[r] < 0 (exact: the train-mode point stacks fold sign(gamma) of the LAST BatchNorm
                           // into the last convolution, so that one running maximum serves either sign of the scale)
};

constexpr int kPackMax = 120;     // 8.5 KB of kernel arguments (the kernarg segment is plain memory on AMD): one launch packs a whole model's images
struct PackArgs {
  int n;
  PackDesc d[kPackMax];
};
static_assert(sizeof(PackArgs) <= 16384, "kernel argument block");

int pack_images(const PackDesc* descs, int n, hipStream_t stream);

// Helper: descriptor for layer LI of a LayerSeq placed at `base`.
template <class Seq>
inline PackDesc pack_desc(int li, float* base, const float* w, const float* b, int N, int K, bool transposed) {
  PackDesc d;
  d.w = w; d.b = transposed ? nullptr : b;
  d.dst = base + Seq::layer_off(li);
  d.N = N; d.K = K; d.NP = Seq::np(li); d.KP = Seq::kp(li);
  d.transposed = transposed ? 1 : 0;
  d.ld = transposed ? N : K;
  d.row0 = 0; d.nrows = d.NP;
  d.col0 = 0; d.partial = 0;
  d.bf = Seq::bf(li) ? 1 : 0;
  d.row_sign = nullptr;
  return d;
}

// Slice descriptor: rows [row0, row0 + nrows) of layer LI's image <- the [N, K] slice of a forward
// weight starting at `w` with leading dimension ld (transposed: the slice is [K, N] in the source).
template <class Seq>
inline PackDesc pack_slice(int li, float* base, const float* w, const float* b, int N, int K, int ld, int row0, int nrows,
                           bool transposed) {
  PackDesc d = pack_desc<Seq>(li, base, w, b, N, K, transposed);
  d.ld = ld; d.row0 = row0; d.nrows = nrows;
  return d;
}

// Block of an image assembled from several weight slices: rows [row0, row0 + nrows), columns
// [col0, col0 + K); w == nullptr writes zeros.
template <class Seq>
inline PackDesc pack_block(int li, float* base, const float* w, int N, int K, int ld, int row0, int nrows, int col0,
                           bool transposed) {
  PackDesc d = pack_slice<Seq>(li, base, w, nullptr, N, K, ld, row0, nrows, transposed);
  d.col0 = col0; d.partial = 1;
  return d;
}

// Fragment-stream images (b3d_estream.hpp): one descriptor per layer of a sequence.
struct FragDesc {
  const float* w;      // element (n, k) of the layer = w[n * ld + k] (transposed: w[k * ld + n])
  const float* b;      // bias [N] or nullptr
  void* steps;         // first step of the layer inside the image
  float* bias;         // the layer's slot in the image's bias table (always written: zeros without a bias)
  int N, K, ld, transposed;
};
constexpr int kFragMax = 24;
struct FragArgs { int n; FragDesc d[kFragMax]; };
int pack_frags(const FragDesc* descs, int n, hipStream_t stream);
template <class S>
inline FragDesc frag_desc(int li, float* image, const float* w, const float* b, int ld, bool transposed) {
  FragDesc d;
  d.w = w; d.b = b;
  d.steps = reinterpret_cast<char*>(image) + (size_t)S::first_step(li) * 6144;
  d.bias = reinterpret_cast<float*>(reinterpret_cast<char*>(image) + S::WEIGHT_BYTES) + S::bias_off(li);
  d.N = S::n(li); d.K = S::k(li); d.ld = ld; d.transposed = transposed ? 1 : 0;
  return d;
}

inline PackDesc fill_desc(void* dst, int count, bool iota) {
  PackDesc d;
  d.w = nullptr; d.b = nullptr; d.dst = (float*)dst;
  d.N = count; d.K = 0; d.NP = 0; d.KP = 0;
  d.transposed = iota ? 2 : 3;
  d.ld = 0; d.row0 = 0; d.nrows = 0; d.col0 = 0; d.partial = 0; d.bf = 0; d.row_sign = nullptr;
  return d;
}

}  // namespace b3d
