// Error reporting + version for libeat_hip.so (no mutable global state: the error text is
// thread-local).
#include "eat_common.h"

namespace eat {
char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}
}  // namespace eat

extern "C" int eat_version(void) { return 100; }
extern "C" const char* eat_last_error_string(void) { return eat::err_buf(); }
