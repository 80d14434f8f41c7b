// Shared host-side helpers of liblsrecon (gfx950 only; no other target is supported).
#pragma once

#include <hip/hip_runtime.h>

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
