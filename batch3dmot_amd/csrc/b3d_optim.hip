// Adam step of the training loop (reference train.py:106-109: torch.optim.Adam with L2 weight
// decay) over ONE contiguous fp32 parameter buffer: one launch instead of the ~8 multi-tensor
// launches (and ~0.4 ms of host time) of the eager optimizer.
#include "b3d_common.hpp"

namespace b3d {
namespace {

struct AdamArgs {
  float* p;
  const float* g;
  float* m;
  float* v;
  long n;
  float beta1, beta2, one_minus_beta1, one_minus_beta2, eps, weight_decay;
  float step_size;          // lr / (1 - beta1^t)
  float bc2_sqrt;           // sqrt(1 - beta2^t)
};

// torch/optim/adam.py (_single_tensor_adam / _multi_tensor_adam, non-capturable, amsgrad=False,
// maximize=False), operation for operation:
//   g' = g + wd p;  m = lerp(m, g', 1-b1);  v = b2 v + (1-b2) g' g';
//   p -= step_size * m / (sqrt(v) / bc2_sqrt + eps)
__global__ __launch_bounds__(256) void adam_kernel(const AdamArgs a) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  float p = a.p[i];
  float g = a.g[i];
  if (a.weight_decay != 0.f) g = __fmaf_rn(p, a.weight_decay, g);
  float m = a.m[i], v = a.v[i];
  m = __fmaf_rn(a.one_minus_beta1, g - m, m);
  v = __fmaf_rn(a.one_minus_beta2 * g, g, v * a.beta2);
  const float denom = __fsqrt_rn(v) / a.bc2_sqrt + a.eps;
  p = p - a.step_size * (m / denom);
  a.p[i] = p; a.m[i] = m; a.v[i] = v;
}

// Step counter on the device (for hipGraph-captured training steps: a captured launch replays its arguments,
// so the bias corrections cannot come from a host integer).
__global__ __launch_bounds__(256) void adam_dev_kernel(AdamArgs a, const long long* __restrict__ step, float lr) {
  // the bias corrections once per workgroup (two float64 pow per THREAD were most of this launch: 21 us for 1.3 M parameters)
  __shared__ float s_corr[2];
  if (threadIdx.x == 0) {
    const double t = (double)(*step + 1);
    const double bc1 = 1.0 - pow((double)a.beta1, t), bc2 = 1.0 - pow((double)a.beta2, t);
    s_corr[0] = (float)((double)lr / bc1);
    s_corr[1] = (float)sqrt(bc2);
  }
  __syncthreads();
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const float step_size = s_corr[0], bc2_sqrt = s_corr[1];
  float p = a.p[i];
  float g = a.g[i];
  if (a.weight_decay != 0.f) g = __fmaf_rn(p, a.weight_decay, g);
  float m = a.m[i], v = a.v[i];
  m = __fmaf_rn(a.one_minus_beta1, g - m, m);
  v = __fmaf_rn(a.one_minus_beta2 * g, g, v * a.beta2);
  const float denom = __fsqrt_rn(v) / bc2_sqrt + a.eps;
  p = p - step_size * (m / denom);
  a.p[i] = p; a.m[i] = m; a.v[i] = v;
}
__global__ void adam_tick_kernel(long long* step) { *step += 1; }

}  // namespace
}  // namespace b3d

using namespace b3d;

extern "C" int b3d_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                             float beta1, float beta2, float eps, float weight_decay, int64_t step,
                             b3d_stream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  B3D_REQUIRE(param && grad && exp_avg && exp_avg_sq, "b3d_adam_step: null argument");
  B3D_REQUIRE(n >= 0 && step >= 1, "b3d_adam_step: n %lld, step %lld", (long long)n, (long long)step);
  if (n == 0) return B3D_OK;
  AdamArgs a;
  a.p = param; a.g = grad; a.m = exp_avg; a.v = exp_avg_sq; a.n = (long)n;
  a.beta1 = beta1; a.beta2 = beta2; a.one_minus_beta1 = 1.f - beta1; a.one_minus_beta2 = 1.f - beta2;
  a.eps = eps; a.weight_decay = weight_decay;
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  a.step_size = (float)((double)lr / bc1);
  a.bc2_sqrt = (float)sqrt(bc2);
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a);
  B3D_HIP_CHECK(hipGetLastError());
  return B3D_OK;
}

extern "C" int b3d_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                                 float beta1, float beta2, float eps, float weight_decay, int64_t* step_dev,
                                 b3d_stream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  B3D_REQUIRE(param && grad && exp_avg && exp_avg_sq && step_dev, "b3d_adam_step_dev: null argument");
  B3D_REQUIRE(n >= 0, "b3d_adam_step_dev: n %lld", (long long)n);
  AdamArgs a;
  a.p = param; a.g = grad; a.m = exp_avg; a.v = exp_avg_sq; a.n = (long)n;
  a.beta1 = beta1; a.beta2 = beta2; a.one_minus_beta1 = 1.f - beta1; a.one_minus_beta2 = 1.f - beta2;
  a.eps = eps; a.weight_decay = weight_decay; a.step_size = 0.f; a.bc2_sqrt = 0.f;
  if (n > 0) {
    hipLaunchKernelGGL(adam_dev_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a, (const long long*)step_dev, lr);
    B3D_HIP_CHECK(hipGetLastError());
  }
  hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, stream, (long long*)step_dev);
  B3D_HIP_CHECK(hipGetLastError());
  return B3D_OK;
}
