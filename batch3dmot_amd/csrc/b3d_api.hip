// Version / error plumbing of the C ABI.
#include "b3d_common.hpp"

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

}  // namespace b3d

extern "C" int b3d_version(void) { return 100; }   // 0.1.0
extern "C" const char* b3d_last_error(void) { return b3d::last_error_buf(); }
