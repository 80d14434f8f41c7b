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
  if (e != hipSuccess) {
    (void)hipGetLastError();   // not a sticky state: callers fall back to another kernel and then ask launch_status()
    return fail(static_cast<int>(e), "%s: device %d refuses %d bytes of dynamic LDS: %s", what, dev, bytes,
                hipGetErrorString(e));
  }
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

// What this library talks about: extents in [1, 2^30) and fewer than 2^48 voxels per volume, element counts below 2^48,
// row strides below 2^31 and plane strides below 2^32 elements -- so that every product of extents, strides and small
// factors in the host-side planning (tile counts, byte sizes, workgroup grids) stays far inside int64.  An entry point
// checks its arguments against these before it computes anything with them (tools/fuzz_device_args.py under UBSan).
constexpr int64_t kMaxExtent = int64_t(1) << 30;
constexpr int64_t kMaxVoxels = int64_t(1) << 48;
inline bool volume_in_range(int64_t Z, int64_t Y, int64_t X) {
  if (Z <= 0 || Y <= 0 || X <= 0 || Z >= kMaxExtent || Y >= kMaxExtent || X >= kMaxExtent) return false;
  return Z * Y <= kMaxVoxels / X;
}
inline bool strides_in_range(int64_t pitch, int64_t plane) {
  return pitch >= 0 && pitch < (int64_t(1) << 31) && plane >= 0 && plane < (int64_t(1) << 32);
}

}  // namespace lsr

#define LSR_REQUIRE_PTR(p)                                                  \
  do {                                                                      \
    if ((p) == nullptr) return lsr::fail(LSR_E_NULL, "%s is NULL", #p);     \
  } while (0)

#define LSR_REQUIRE(cond, code, ...)                       \
  do {                                                     \
    if (!(cond)) return lsr::fail((code), __VA_ARGS__);    \
  } while (0)

#define LSR_REQUIRE_VOLUME(Z, Y, X)                                                                                   \
  LSR_REQUIRE(lsr::volume_in_range((Z), (Y), (X)), LSR_E_UNSUPPORTED,                                                 \
              "shape (%lld,%lld,%lld): every extent must be below 2^30 and the volume below 2^48 voxels", (long long)(Z), \
              (long long)(Y), (long long)(X))
#define LSR_REQUIRE_COUNT(n) \
  LSR_REQUIRE((n) < lsr::kMaxVoxels, LSR_E_UNSUPPORTED, "%lld elements: the limit is 2^48", (long long)(n))
#define LSR_REQUIRE_STRIDES(pitch, plane)                                                                      \
  LSR_REQUIRE(lsr::strides_in_range((pitch), (plane)), LSR_E_UNSUPPORTED,                                      \
              "strides (%lld, %lld): a row stride must be in [0, 2^31), a plane stride in [0, 2^32) elements", \
              (long long)(pitch), (long long)(plane))

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
