// Version and error reporting of liblsrecon (see include/lsrecon.h).
#include "common.hpp"

namespace lsr {

char* error_buffer() {
  static thread_local char buf[512] = {0};
  return buf;
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 512, fmt, ap);
  va_end(ap);
  return code;
}

}  // namespace lsr

extern "C" int lsr_version(void) { return LSR_VERSION; }

extern "C" const char* lsr_last_error(void) { return lsr::error_buffer(); }
