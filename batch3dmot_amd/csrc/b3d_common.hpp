// Host-side helpers shared by the translation units of libb3d_hip.so.
#pragma once
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "../../include/b3d.h"

namespace b3d {

// ReLU as torch computes it: NaN stays NaN.  v_max_f32 returns the OTHER operand for a quiet NaN, so fmaxf(NaN, 0) = 0
// would turn a diverged activation back into a finite one; one compare + one select per element instead.
__device__ __forceinline__ float relu1(float x) { return x < 0.f ? 0.f : x; }

// thread-local error string returned by b3d_last_error(); the only mutable global state.
char* last_error_buf();
int fail(int code, const char* fmt, ...);

#define B3D_HIP_CHECK(expr)                                                            \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess)                                                              \
      return ::b3d::fail(B3D_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                         __FILE__, __LINE__);                                          \
  } while (0)

#define B3D_TRY(expr)          \
  do {                         \
    int _r = (expr);           \
    if (_r != B3D_OK) return _r; \
  } while (0)

#define B3D_REQUIRE(cond, ...) \
  do {                         \
    if (!(cond)) return ::b3d::fail(B3D_ERR_ARG, __VA_ARGS__); \
  } while (0)

inline int launch_check(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(B3D_ERR_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
  return B3D_OK;
}

// Number of workgroups for a row-tiled kernel: one 128-row tile per workgroup, capped so that
// very large inputs grid-stride (256 CUs, at most a few resident workgroups each).
inline int grid_for_tiles(long rows, int tile_rows, int cap = 2048) {
  long t = (rows + tile_rows - 1) / tile_rows;
  if (t < 1) t = 1;
  if (const char* ev = getenv("B3D_GRID_CAP")) {          // testing aid: force the grid-stride (several tiles per workgroup) path
    const int v = atoi(ev);
    if (v >= 1) cap = v;
  }
  return (int)(t < cap ? t : cap);
}

// Kernel-family timers (b3d_prof_*): records an event pair around a launch when enabled.
bool prof_on(int family);
void prof_begin(int family, hipStream_t stream);
void prof_end(hipStream_t stream);
// roctx ranges named after the kernel family around the same launches (b3d_prof_markers / B3D_ROCTX=1): rocprofv3 --marker-trace
// timelines become self-describing.  Host-side ranges: inside a hipGraph capture they bracket the capture, not the replay.
bool marker_on();
void marker_push(int family);
void marker_pop();
int marker_enable(int on);
struct ProfScope {
  hipStream_t s;
  bool on, mk;
  ProfScope(int family, hipStream_t stream) : s(stream), on(prof_on(family)), mk(marker_on()) {
    if (mk) marker_push(family);
    if (on) prof_begin(family, s);
  }
  ~ProfScope() {
    if (on) prof_end(s);
    if (mk) marker_pop();
  }
};

// Raise a kernel's dynamic-LDS limit once per (kernel, device): hipFuncSetAttribute costs ~2 us of host time
// and a training step launches ~45 kernels.
int set_lds_cached(const void* kernel, int bytes);

// Library-owned side streams (one set per device, created on first use, never destroyed): work
// that has no consumer on the caller's stream until later (the discarded k-NN + GAT block, weight
// gradients) is forked onto them and joined back before the entry point returns, so that from the
// caller's point of view everything is still ordered on the stream it passed.
struct Side {
  hipStream_t s;
  hipEvent_t ev_fork, ev_join;
};
int side_get(int idx, Side** out);             // idx < kSideStreams
int side_fork(hipStream_t main, Side* sd);     // side waits for all work enqueued on main so far
int side_join(Side* sd, hipStream_t main);     // main waits for all work enqueued on side so far
constexpr int kSideStreams = 2;

// Bump allocator over a caller-provided workspace (the library allocates nothing).
struct Carver {
  char* base;
  size_t off, cap;
  bool dry;     // size query: only count
  Carver(void* p, size_t bytes) : base((char*)p), off(0), cap(bytes), dry(p == nullptr) {}
  template <class T>
  T* take(size_t n) {
    size_t a = (off + 255) & ~(size_t)255;
    off = a + n * sizeof(T);
    if (dry) return nullptr;
    return (off <= cap) ? (T*)(base + a) : nullptr;
  }
  bool ok() const { return dry || off <= cap; }
};

}  // namespace b3d
