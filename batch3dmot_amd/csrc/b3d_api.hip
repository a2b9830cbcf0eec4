// Version / error plumbing of the C ABI.
#include "b3d_common.hpp"

#include <mutex>
#include <vector>

namespace b3d {

char* last_error_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(last_error_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

// ---- kernel-family timers ------------------------------------------------------------------------
namespace {
struct ProfState {
  std::mutex mu;
  bool on = false;
  std::vector<hipEvent_t> pool;          // event pairs: 2*i, 2*i+1
  std::vector<int> fam;                  // family of pair i
  size_t used = 0;                       // pairs handed out since reset
  int open = -1;
};
ProfState& prof() { static ProfState p; return p; }
}  // namespace

bool prof_on() { return prof().on; }

void prof_begin(int family, hipStream_t stream) {
  ProfState& p = prof();
  std::lock_guard<std::mutex> lk(p.mu);
  if (!p.on) return;
  if (p.used * 2 + 2 > p.pool.size()) {
    for (int i = 0; i < 512; ++i) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) return;
      p.pool.push_back(e);
    }
  }
  p.fam.resize(p.pool.size() / 2);
  p.fam[p.used] = family;
  p.open = (int)p.used;
  (void)hipEventRecord(p.pool[2 * p.used], stream);
}

void prof_end(hipStream_t stream) {
  ProfState& p = prof();
  std::lock_guard<std::mutex> lk(p.mu);
  if (p.open < 0) return;
  (void)hipEventRecord(p.pool[2 * p.open + 1], stream);
  p.used = (size_t)p.open + 1;
  p.open = -1;
}

}  // namespace b3d

extern "C" int b3d_prof_enable(int on) {
  auto& p = b3d::prof();
  std::lock_guard<std::mutex> lk(p.mu);
  p.on = on != 0;
  return B3D_OK;
}
extern "C" int b3d_prof_reset(void) {
  auto& p = b3d::prof();
  std::lock_guard<std::mutex> lk(p.mu);
  p.used = 0;
  p.open = -1;
  return B3D_OK;
}
extern "C" int b3d_prof_read(int family, double* total_ms, int* launches) {
  auto& p = b3d::prof();
  std::lock_guard<std::mutex> lk(p.mu);
  if (!total_ms || !launches) return b3d::fail(B3D_ERR_ARG, "b3d_prof_read: null output");
  double tot = 0.0;
  int n = 0;
  for (size_t i = 0; i < p.used; ++i) {
    if (p.fam[i] != family) continue;
    if (hipEventSynchronize(p.pool[2 * i + 1]) != hipSuccess) return b3d::fail(B3D_ERR_HIP, "event sync failed");
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.pool[2 * i], p.pool[2 * i + 1]) != hipSuccess) return b3d::fail(B3D_ERR_HIP, "event elapsed failed");
    tot += ms;
    ++n;
  }
  *total_ms = tot;
  *launches = n;
  return B3D_OK;
}

extern "C" int b3d_version(void) { return 100; }   // 0.1.0
extern "C" const char* b3d_last_error(void) { return b3d::last_error_buf(); }
