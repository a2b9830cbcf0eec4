// Microbenchmark: GPU time per kernel of a chain of dependent small kernels (stream order vs hipGraph replay).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void tiny(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
__global__ void touch(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.f; }

static double run(const char* name, int n, bool graph, int grid, int wg, float* buf, int elems, unsigned flags) {
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto body = [&]() {
    for (int i = 0; i < n; ++i) {
      if (flags) hipExtLaunchKernelGGL(touch, dim3(grid), dim3(wg), 0, s, nullptr, nullptr, flags, buf, elems);
      else if (grid == 1) hipLaunchKernelGGL(tiny, dim3(1), dim3(wg), 0, s, buf);
      else hipLaunchKernelGGL(touch, dim3(grid), dim3(wg), 0, s, buf, elems);
    }
  };
  float ms = 0;
  if (graph) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    body();
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int r = 0; r < 5; ++r) hipGraphLaunch(ge, s);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  } else {
    body(); hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    body();
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  printf("%-34s %-6s grid %6d x %4d: %6.2f us per kernel\n", name, graph ? "graph" : "stream", grid, wg, ms * 1e3 / n);
  return ms;
}

int main() {
  float* buf; hipMalloc(&buf, 64 << 20); hipMemset(buf, 0, 64 << 20);
  for (int g = 0; g < 2; ++g) {
    run("1 thread", 1000, g, 1, 64, buf, 0, 0);
    run("188 x 512 (node kernel shape)", 1000, g, 188, 512, buf, 188 * 512, 0);
    run("1940 x 256, 2 MB touched", 1000, g, 1940, 256, buf, 1940 * 256, 0);
    run("8 MB touched", 500, g, 8192, 256, buf, 8192 * 256, 0);
  }
  run("188 x 512 any-order launch", 1000, false, 188, 512, buf, 188 * 512, hipExtAnyOrderLaunch);
  return 0;
}
