// Error plumbing + version for libts_hip.so.
#include "ts_common.hpp"

namespace ts {

char* err_buf() {
  static thread_local char buf[512] = "ok";
  return buf;
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

}  // namespace ts

extern "C" int ts_version(void) { return TS_ABI_VERSION; }

extern "C" const char* ts_last_error_string(void) { return ts::err_buf(); }
