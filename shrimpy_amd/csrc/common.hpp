// Shared host-side helpers of liblsrecon (gfx950 only; no other target is supported).
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/lsrecon.h"

namespace lsr {

// Thread-local message behind lsr_last_error().
char* error_buffer();
int fail(int code, const char* fmt, ...);

inline int launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(static_cast<int>(e), "%s: %s", what, hipGetErrorString(e));
  return LSR_OK;
}

inline hipStream_t as_stream(lsr_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Raise a kernel's dynamic-LDS ceiling above the 64 KB default.  The attribute belongs to the kernel's
// code object on ONE device, so it is set once per (kernel, device): `done` is that kernel's bit mask of
// devices already served (one static per call site; devices >= 64 are simply set every time).
// Returns LSR_OK or the HIP error (message in lsr_last_error()).
inline int allow_dynamic_lds(const void* kernel, int bytes, std::atomic<uint64_t>& done, const char* what) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return fail(static_cast<int>(e), "%s: hipGetDevice: %s", what, hipGetErrorString(e));
  const uint64_t bit = dev < 64 ? uint64_t(1) << dev : 0;
  if (bit && (done.load(std::memory_order_acquire) & bit)) return LSR_OK;
  e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess)
    return fail(static_cast<int>(e), "%s: device %d refuses %d bytes of dynamic LDS: %s", what, dev, bytes,
                hipGetErrorString(e));
  if (bit) done.fetch_or(bit, std::memory_order_release);
  return LSR_OK;
}

// Does a workgroup of the current device get `bytes` of LDS?  (gfx950: 160 KB per CU.)  Without a device to
// ask -- the build container -- the answer is yes: the *_supported functions then speak for the lengths only.
inline bool lds_fits(size_t bytes) {
  int dev = 0, cap = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return true; }
  if (hipDeviceGetAttribute(&cap, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) {
    (void)hipGetLastError();
    return true;
  }
  return static_cast<size_t>(cap) >= bytes;
}

constexpr int kWave = 64;  // CDNA4 wavefront

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace lsr

#define LSR_REQUIRE_PTR(p)                                                  \
  do {                                                                      \
    if ((p) == nullptr) return lsr::fail(LSR_E_NULL, "%s is NULL", #p);     \
  } while (0)

#define LSR_REQUIRE(cond, code, ...)                       \
  do {                                                     \
    if (!(cond)) return lsr::fail((code), __VA_ARGS__);    \
  } while (0)

// ---- device-side exact fp64 helpers (no FMA contraction: results must match scipy's C) ----
#if defined(__HIPCC__)
namespace lsr {

__device__ __forceinline__ double dmul(double a, double b) { return __dmul_rn(a, b); }
__device__ __forceinline__ double dadd(double a, double b) { return __dadd_rn(a, b); }

// Coordinate of one input axis for output index (zo, yo, xo), evaluated in scipy's order:
// ((zo*m0 + yo*m1) + xo*m2) + shift, every product and sum rounded separately.
__device__ __forceinline__ double affine_coord(double zo, double yo, double xo, double m0,
                                               double m1, double m2, double shift) {
  double c = dmul(zo, m0);
  c = dadd(c, dmul(yo, m1));
  c = dadd(c, dmul(xo, m2));
  return dadd(c, shift);
}

}  // namespace lsr
#endif
