// Microbenchmark + layout check: one Linear layer on a 16-row wave tile, Y^T = W . X^T + b, as
//   (a) v_mfma_f32_16x16x4_f32 (exact fp32, the library's first form), and
//   (b) "bf16x6": every fp32 operand split EXACTLY into three bf16 pieces (8 + 8 + 8 significand bits, by
//       truncation: x = x0 + x1 + x2), the product formed from the six piece products whose weight is >= 2^-24
//       (x0w0, x0w1, x1w0, x0w2, x1w1, x2w0) on v_mfma_f32_16x16x32_bf16 with fp32 accumulation: fp32-class
//       accuracy at 16 / 6 = 2.67x the fp32 MFMA rate.
// Two layers are chained (64 -> 128 -> ReLU -> 32) so that the accumulator -> next operand hand-over (no lane
// movement: the k index of the next layer is PERMUTED consistently in the weight image) is checked too.
//   hipcc --offload-arch=gfx950 -O3 bf16x6_linear.hip -o bf16x6_linear && ./bf16x6_linear
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

// ---- image geometry --------------------------------------------------------------------------------
// fp32 image: row stride K + 8 dwords (conflict-free ds_read_b128), bias at column K.
// bf16x3 image: row = [piece 0: K/2 dwords][piece 1][piece 2][bias][pad]: stride 3K/2 + 8 dwords.
//   inside a piece, the 32 features of group c sit at dwords 16 c .. 16 c + 15 in the order the B operand is held:
//   position 8 g + j  <->  feature 32 c + 16 (j >> 2) + 4 g + (j & 3)      (g = 0..3 lane quarter, j = 0..7)
__host__ __device__ constexpr int stride32(int K) { return K + 8; }
__host__ __device__ constexpr int stride16(int K) { return 3 * K / 2 + 8; }
__host__ __device__ constexpr int perm_pos(int f) { return 8 * ((f & 15) >> 2) + 4 * (f >> 4) + (f & 3); }   // f in [0,32)

// ---- bf16x6 ------------------------------------------------------------------------------------------
// exact three-way split of 8 fp32 values (two layout-L blocks of one row) into the B-operand fragments
__device__ __forceinline__ void split3(const v4f a, const v4f b, bf8& p0, bf8& p1, bf8& p2) {
  const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  unsigned h[8], m[8], l[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const unsigned xb = __float_as_uint(x[i]);
    h[i] = xb & 0xffff0000u;
    const float r1 = x[i] - __uint_as_float(h[i]);           // exact: <= 16 significant bits
    m[i] = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(m[i]);             // exact: <= 8 significant bits
    l[i] = __float_as_uint(r2);
  }
  u4 q0, q1, q2;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    q0[d] = (h[2 * d] >> 16) | (h[2 * d + 1] & 0xffff0000u);
    q1[d] = (m[2 * d] >> 16) | (m[2 * d + 1] & 0xffff0000u);
    q2[d] = (l[2 * d] >> 16) | (l[2 * d + 1] & 0xffff0000u);
  }
  p0 = __builtin_bit_cast(bf8, q0);
  p1 = __builtin_bit_cast(bf8, q1);
  p2 = __builtin_bit_cast(bf8, q2);
}

struct Frag3 { bf8 w0, w1, w2; };
template <int K>
__device__ __forceinline__ Frag3 load3(const unsigned* p) {
  Frag3 f;
  f.w0 = __builtin_bit_cast(bf8, *reinterpret_cast<const u4*>(p));
  f.w1 = __builtin_bit_cast(bf8, *reinterpret_cast<const u4*>(p + K / 2));
  f.w2 = __builtin_bit_cast(bf8, *reinterpret_cast<const u4*>(p + K));
  return f;
}
__device__ __forceinline__ v4f mfma6(const Frag3& f, const bf8 x0, const bf8 x1, const bf8 x2, v4f acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.w0, x2, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.w1, x1, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.w2, x0, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.w0, x1, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.w1, x0, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.w0, x0, acc, 0, 0, 0);
  return acc;
}

// software-pipelined: the fragments (and bias) of step t+1 are read from LDS before the 6 MFMAs of step t are issued
template <int K, int N, bool RELU>
__device__ __forceinline__ void linear_bf6(const unsigned* __restrict__ w, const v4f* __restrict__ in, v4f* __restrict__ out) {
  constexpr int KG = K / 32, NB = N / 16, S = stride16(K);
  const int lane = threadIdx.x & 63, m = lane & 15, g = lane >> 4;
  bf8 x0[KG], x1[KG], x2[KG];
#pragma unroll
  for (int c = 0; c < KG; ++c) split3(in[2 * c], in[2 * c + 1], x0[c], x1[c], x2[c]);
  const unsigned* wrow = w + m * S + 4 * g;
  const unsigned* wbias = w + 4 * g * S + 3 * K / 2;
  auto bias = [&](int mb) {
    const unsigned* wb = wbias + 16 * mb * S;
    return v4f{__uint_as_float(wb[0]), __uint_as_float(wb[S]), __uint_as_float(wb[2 * S]), __uint_as_float(wb[3 * S])};
  };
  Frag3 cur = load3<K>(wrow);
  v4f nb = bias(0);
#pragma unroll
  for (int mb = 0; mb < NB; ++mb) {
    v4f acc = nb;
#pragma unroll
    for (int c = 0; c < KG; ++c) {
      Frag3 nxt = cur;
      if (c + 1 < KG) nxt = load3<K>(wrow + 16 * mb * S + 16 * (c + 1));
      else if (mb + 1 < NB) { nxt = load3<K>(wrow + 16 * (mb + 1) * S); nb = bias(mb + 1); }
      __builtin_amdgcn_sched_barrier(0);
      acc = mfma6(cur, x0[c], x1[c], x2[c], acc);
      cur = nxt;
    }
    if (RELU) { acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f); }
    out[mb] = acc;
  }
}

template <int K, int N, bool RELU>
__device__ __forceinline__ void linear_f32(const float* __restrict__ w, const v4f* __restrict__ in, v4f* __restrict__ out) {
  constexpr int KB = K / 16, NB = N / 16, S = stride32(K);
  static_assert(NB % 2 == 0, "pairs of output blocks");
  const int lane = threadIdx.x & 63, m = lane & 15, q = lane >> 4;
  const float* wrow = w + m * S + 4 * q;
  auto frag = [&](int mb, int kb) { return *reinterpret_cast<const v4f*>(wrow + 16 * mb * S + 16 * kb); };
  auto bias = [&](int mb) { const float* wb = w + (16 * mb + 4 * q) * S + K; return v4f{wb[0], wb[S], wb[2 * S], wb[3 * S]}; };
  v4f fa0 = frag(0, 0), fa1 = frag(1, 0), nb0 = bias(0), nb1 = bias(1);
#pragma unroll
  for (int mb = 0; mb < NB; mb += 2) {
    v4f acc0 = nb0, acc1 = nb1;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      v4f na0 = fa0, na1 = fa1;
      if (kb + 1 < KB) { na0 = frag(mb, kb + 1); na1 = frag(mb + 1, kb + 1); }
      else if (mb + 2 < NB) { na0 = frag(mb + 2, 0); na1 = frag(mb + 3, 0); nb0 = bias(mb + 2); nb1 = bias(mb + 3); }
      __builtin_amdgcn_sched_barrier(0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa0.x, in[kb].x, acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa0.y, in[kb].y, acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa0.z, in[kb].z, acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa0.w, in[kb].w, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa1.x, in[kb].x, acc1, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa1.y, in[kb].y, acc1, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa1.z, in[kb].z, acc1, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa1.w, in[kb].w, acc1, 0, 0, 0);
      fa0 = na0; fa1 = na1;
    }
    if (RELU) {
      acc0.x = fmaxf(acc0.x, 0.f); acc0.y = fmaxf(acc0.y, 0.f); acc0.z = fmaxf(acc0.z, 0.f); acc0.w = fmaxf(acc0.w, 0.f);
      acc1.x = fmaxf(acc1.x, 0.f); acc1.y = fmaxf(acc1.y, 0.f); acc1.z = fmaxf(acc1.z, 0.f); acc1.w = fmaxf(acc1.w, 0.f);
    }
    out[mb] = acc0; out[mb + 1] = acc1;
  }
}

constexpr int K0 = 64, N0 = 128, N1 = 32;

// correctness: one wavefront, two chained layers, x [16, K0] -> y [16, N1]
template <bool BF6>
__global__ __launch_bounds__(64) void check_kernel(const float* __restrict__ img0, const float* __restrict__ img1, int n0, int n1,
                                                   const float* __restrict__ x, float* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int i = threadIdx.x; i < n0; i += 64) lds[i] = img0[i];
  for (int i = threadIdx.x; i < n1; i += 64) lds[n0 + i] = img1[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  v4f in[K0 / 16], h[N0 / 16], o[N1 / 16];
#pragma unroll
  for (int b = 0; b < K0 / 16; ++b) in[b] = *reinterpret_cast<const v4f*>(x + r * K0 + 16 * b + 4 * q);
  if constexpr (BF6) {
    linear_bf6<K0, N0, true>(reinterpret_cast<const unsigned*>(lds), in, h);
    linear_bf6<N0, N1, false>(reinterpret_cast<const unsigned*>(lds + n0), h, o);
  } else {
    linear_f32<K0, N0, true>(lds, in, h);
    linear_f32<N0, N1, false>(lds + n0, h, o);
  }
#pragma unroll
  for (int b = 0; b < N1 / 16; ++b) *reinterpret_cast<v4f*>(y + r * N1 + 16 * b + 4 * q) = o[b];
}

// throughput: 8 wavefronts per workgroup, one workgroup per CU, a 256 -> 256 layer applied REPS times (output fed back)
constexpr int TK = 128;
template <bool BF6>
__global__ __launch_bounds__(512, 2) void rate_kernel(const float* __restrict__ img, int n, int reps, float* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int i = threadIdx.x; i < n; i += 512) lds[i] = img[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  v4f a[TK / 16], b[TK / 16];
#pragma unroll
  for (int i = 0; i < TK / 16; ++i) a[i] = v4f{0.001f * lane, 0.002f * i, 0.5f, -0.25f};
  for (int r = 0; r < reps; ++r) {
    if constexpr (BF6) {
      linear_bf6<TK, TK, true>(reinterpret_cast<const unsigned*>(lds), a, b);
      linear_bf6<TK, TK, true>(reinterpret_cast<const unsigned*>(lds), b, a);
    } else {
      linear_f32<TK, TK, true>(lds, a, b);
      linear_f32<TK, TK, true>(lds, b, a);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < TK / 16; ++i) s += a[i].x + a[i].y + a[i].z + a[i].w;
  y[blockIdx.x * 512 + threadIdx.x] = s;
}

// ---- host ---------------------------------------------------------------------------------------------
static void pack32(const std::vector<float>& W, const std::vector<float>& b, int N, int K, std::vector<float>& img) {
  const int S = stride32(K);
  img.assign((size_t)N * S, 0.f);
  for (int r = 0; r < N; ++r) {
    for (int c = 0; c < K; ++c) img[(size_t)r * S + c] = W[(size_t)r * K + c];
    img[(size_t)r * S + K] = b[r];
  }
}
static void pack16(const std::vector<float>& W, const std::vector<float>& b, int N, int K, std::vector<float>& img) {
  const int S = stride16(K);
  std::vector<unsigned> u((size_t)N * S, 0u);
  for (int r = 0; r < N; ++r) {
    for (int c = 0; c < K; ++c) {
      float x = W[(size_t)r * K + c];
      unsigned xb; memcpy(&xb, &x, 4);
      unsigned h = xb & 0xffff0000u; float hf; memcpy(&hf, &h, 4);
      float r1 = x - hf; unsigned r1b; memcpy(&r1b, &r1, 4);
      unsigned m = r1b & 0xffff0000u; float mf; memcpy(&mf, &m, 4);
      float r2 = r1 - mf; unsigned l; memcpy(&l, &r2, 4);
      const unsigned piece[3] = {h >> 16, m >> 16, l >> 16};
      const int pos = 32 * (c / 32) + perm_pos(c % 32);
      for (int p = 0; p < 3; ++p) {
        unsigned& d = u[(size_t)r * S + p * (K / 2) + pos / 2];
        d |= piece[p] << (16 * (pos & 1));
      }
    }
    memcpy(&u[(size_t)r * S + 3 * K / 2], &b[r], 4);
  }
  img.resize(u.size());
  memcpy(img.data(), u.data(), u.size() * 4);
}

int main() {
  srand(3);
  auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
  std::vector<float> W0(N0 * K0), b0(N0), W1(N1 * N0), b1(N1), x(16 * K0);
  for (auto& v : W0) v = rnd() * 0.3f;
  for (auto& v : b0) v = rnd() * 0.1f;
  for (auto& v : W1) v = rnd() * 0.3f;
  for (auto& v : b1) v = rnd() * 0.1f;
  for (auto& v : x) v = rnd() * 2.f;
  std::vector<double> ref(16 * N1);
  for (int r = 0; r < 16; ++r) {
    std::vector<double> h(N0);
    for (int n = 0; n < N0; ++n) {
      double s = b0[n];
      for (int k = 0; k < K0; ++k) s += (double)W0[n * K0 + k] * x[r * K0 + k];
      h[n] = s > 0 ? s : 0;
    }
    for (int n = 0; n < N1; ++n) {
      double s = b1[n];
      for (int k = 0; k < N0; ++k) s += (double)W1[n * N0 + k] * h[k];
      ref[r * N1 + n] = s;
    }
  }
  float *dx, *dy;
  hipMalloc(&dx, x.size() * 4); hipMalloc(&dy, 16 * N1 * 4);
  hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; ++mode) {
    std::vector<float> i0, i1;
    if (mode) { pack16(W0, b0, N0, K0, i0); pack16(W1, b1, N1, N0, i1); }
    else { pack32(W0, b0, N0, K0, i0); pack32(W1, b1, N1, N0, i1); }
    float *d0, *d1;
    hipMalloc(&d0, i0.size() * 4); hipMalloc(&d1, i1.size() * 4);
    hipMemcpy(d0, i0.data(), i0.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d1, i1.data(), i1.size() * 4, hipMemcpyHostToDevice);
    const int lds = (int)(i0.size() + i1.size()) * 4;
    if (mode) hipLaunchKernelGGL(check_kernel<true>, dim3(1), dim3(64), lds, 0, d0, d1, (int)i0.size(), (int)i1.size(), dx, dy);
    else hipLaunchKernelGGL(check_kernel<false>, dim3(1), dim3(64), lds, 0, d0, d1, (int)i0.size(), (int)i1.size(), dx, dy);
    std::vector<float> y(16 * N1);
    hipMemcpy(y.data(), dy, y.size() * 4, hipMemcpyDeviceToHost);
    double err = 0, mx = 0;
    for (size_t i = 0; i < y.size(); ++i) { err = fmax(err, fabs(y[i] - ref[i])); mx = fmax(mx, fabs(ref[i])); }
    printf("%-7s two chained layers: max abs err %.3e, relative to max |y| %.3e  (%s)\n", mode ? "bf16x6" : "fp32", err, err / mx,
           hipGetErrorString(hipGetLastError()));
    hipFree(d0); hipFree(d1);
  }
  // ---- rate ----
  std::vector<float> W(TK * TK), b(TK);
  for (auto& v : W) v = rnd() * 0.05f;
  for (auto& v : b) v = rnd() * 0.01f;
  float* dout; hipMalloc(&dout, 256 * 512 * 4);
  for (int mode = 0; mode < 2; ++mode) {
    std::vector<float> img;
    if (mode) pack16(W, b, TK, TK, img); else pack32(W, b, TK, TK, img);
    float* dimg; hipMalloc(&dimg, img.size() * 4);
    hipMemcpy(dimg, img.data(), img.size() * 4, hipMemcpyHostToDevice);
    const int lds = (int)img.size() * 4, reps = 200;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&] {
      if (mode) {
        hipFuncSetAttribute((const void*)rate_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(rate_kernel<true>, dim3(256), dim3(512), lds, 0, dimg, (int)img.size(), reps, dout);
      } else {
        hipFuncSetAttribute((const void*)rate_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(rate_kernel<false>, dim3(256), dim3(512), lds, 0, dimg, (int)img.size(), reps, dout);
      }
    };
    launch();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.0 * TK * TK * 16 * 8 * 256 * 2.0 * reps;
    printf("%-7s 128x128 layer, 8 waves x 256 workgroups (LDS %d KB): %.3f ms -> %.1f TFLOP/s (fp32-equivalent)  (%s)\n", mode ? "bf16x6" : "fp32", lds >> 10, ms,
           flop / (ms * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
    hipFree(dimg);
  }
  return 0;
}
