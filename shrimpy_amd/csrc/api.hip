// Version and error reporting of liblsrecon (see include/lsrecon.h).
#include "common.hpp"
#include "host_parallel.hpp"

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

// Does this CPU have the FMA3 instructions the host twins' translation units are compiled for?  (Asked from here:
// this file is built for the baseline x86-64.)
bool host_fma_ok() {
  static const bool ok = __builtin_cpu_supports("fma") != 0;
  return ok;
}

}  // namespace lsr

extern "C" int lsr_version(void) { return LSR_VERSION; }

extern "C" const char* lsr_last_error(void) { return lsr::error_buffer(); }

// The stamp of the sources this binary was built from (csrc/Makefile: SHA16).  A build outside the Makefile has none.
#ifndef LSR_SOURCE_SHA16
#define LSR_SOURCE_SHA16 "unstamped"
#endif
extern "C" const char* lsr_source_sha16(void) { return LSR_SOURCE_SHA16; }

// Page-locked host memory of exactly `bytes` bytes, allocated by the HIP runtime (hipHostMalloc: memory the
// driver owns and maps for every device), for the staging slots of shrimpy_amd/staging.py.  Not
// hipHostRegister on an ordinary allocation: that pins the pages through the kernel's user-pointer path,
// where any later change of the CPU mapping (huge-page collapse, NUMA balancing, compaction) evicts and
// restores the process's GPU queues under copies in flight.
extern "C" int lsr_pinned_alloc(int64_t bytes, void** out) {
  LSR_REQUIRE_PTR(out);
  *out = nullptr;
  LSR_REQUIRE(bytes > 0, LSR_E_ARG, "pinned allocation of %lld bytes", (long long)bytes);
  void* ptr = nullptr;
  const hipError_t e = hipHostMalloc(&ptr, static_cast<size_t>(bytes), hipHostMallocPortable);
  if (e != hipSuccess) {
    (void)hipGetLastError();   // not a sticky state: the caller may fall back to pageable slots
    return lsr::fail(static_cast<int>(e), "hipHostMalloc(%lld bytes): %s", (long long)bytes, hipGetErrorString(e));
  }
  *out = ptr;
  return LSR_OK;
}

extern "C" int lsr_pinned_free(void* ptr) {
  if (ptr == nullptr) return LSR_OK;
  const hipError_t e = hipHostFree(ptr);
  if (e != hipSuccess) return lsr::fail(static_cast<int>(e), "hipHostFree: %s", hipGetErrorString(e));
  return LSR_OK;
}
