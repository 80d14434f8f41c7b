// Host twins of the DynaTrack estimator entry points of estimators.hip (no kernel in this file): the same
// signatures with HOST pointers and the same arithmetic, for the boxes where the reference's tracker itself runs
// on the CPU (`torch.device("cuda" if torch.cuda.is_available() else "cpu")`, shrimpy/dynatrack/tracking.py:1054;
// its CI has no GPU, shrimpy/tests/conftest.py:11-17).  Product code: nothing here calls or reads oracle/.
//
//   lsr_minmax_f32_cpu             <->  lsr_minmax_f32             img.min(), img.max()          (:533-535, :583)
//   lsr_histogram_f32_cpu          <->  lsr_histogram_f32          torch.histc                   (:465, :586)
//   lsr_weighted_centroid_f32_cpu  <->  lsr_weighted_centroid_f32  _intensity_center_of_mass     (:596-649)
//   lsr_mask_centroid_f32_cpu      <->  lsr_mask_centroid_f32      _center_of_mass(img > thr)    (:545-569)
//   lsr_blur_reflect_f32_cpu       <->  lsr_blur_reflect_f32       one axis of _gaussian_blur_3d (:386-422)
//   lsr_match_shape_f32_cpu        <->  lsr_match_shape_f32        _match_shape                  (:266-306)
//   lsr_cross_power_c64_cpu / _into_c64_cpu, lsr_peak_abs_shifted_f32_cpu
//                                  <->  the element-wise steps of _phase_cross_corr             (:309-378)
//
// What "the same" means: min / max, histogram bins (the float32 expression of torch.histc), the shape map, the cross
// power and the peak (first maximum in fftshift order) are identical values; the blur is the kernels' FMA chain in
// ascending tap order -- bit-equal; the centroid sums are fp64 like the kernels' but added in row order (the
// kernels reduce in a tree): equal to the last bits of a double, not bit for bit.  `scratch` and `stream` are
// ignored (kept so the signatures are identical).  Threads: host_parallel.hpp.

#include <cmath>
#include <cstring>
#include <vector>

#include "common.hpp"
#include "host_parallel.hpp"

namespace {

using lsr::parallel_ranges;
using lsr::parallel_ranges_indexed;

constexpr int kMaxBins = 4096;   // as estimators.hip
constexpr int kBlurMaxR = 64;    // as estimators.hip
constexpr int kMaxWorkers = 1024;

int check_volume(const float* in, int64_t Z, int64_t Y, int64_t X) {
  LSR_REQUIRE_PTR(in);
  LSR_REQUIRE(Z > 0 && Y > 0 && X > 0, LSR_E_SHAPE, "shape (%lld,%lld,%lld) must be positive", (long long)Z, (long long)Y,
              (long long)X);
  LSR_REQUIRE_VOLUME(Z, Y, X);
  LSR_REQUIRE(Z < (int64_t(1) << 30) && Y < (int64_t(1) << 30) && X < (int64_t(1) << 30), LSR_E_UNSUPPORTED,
              "a dimension exceeds 2^30");
  return LSR_OK;
}

inline int64_t reflect(int64_t i, int64_t n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  return i;
}

inline int64_t match_index(int64_t o, int64_t ni, int64_t no) {
  if (no > ni) return reflect(o - (no - ni) / 2, ni);
  return o + (ni - no) / 2;
}

// sums {w, w z, w y, w x} over the volume; MASK: w = v > param, else w = max(v - param, 0) in float32
template <bool MASK>
int centroid_cpu(const float* in, int64_t Z, int64_t Y, int64_t X, float param, double* out4) {
  if (int rc = check_volume(in, Z, Y, X)) return rc;
  LSR_REQUIRE_PTR(out4);
  const int64_t rows = Z * Y;
  double part[4 * kMaxWorkers];   // (one slot per range: at most lsr_get_host_threads() <= 1024 of them)
  const int used = parallel_ranges_indexed(rows, [&](int k, int64_t first, int64_t last) {
    double sw = 0.0, sz = 0.0, sy = 0.0, sx = 0.0;
    for (int64_t r = first; r < last; ++r) {
      const float* row = in + r * X;
      double w_row = 0.0, wx_row = 0.0;
      for (int64_t x = 0; x < X; ++x) {
        const float v = row[x];
        const float wf = MASK ? (v > param ? 1.0f : 0.0f) : std::fmax(v - param, 0.0f);
        const double w = static_cast<double>(wf);
        w_row += w;
        wx_row += w * static_cast<double>(x);
      }
      const int64_t z = r / Y, y = r - z * Y;
      sw += w_row;
      sz += w_row * static_cast<double>(z);
      sy += w_row * static_cast<double>(y);
      sx += wx_row;
    }
    double* p = part + 4 * k;
    p[0] = sw; p[1] = sz; p[2] = sy; p[3] = sx;
  });
  // the ranges' sums are added in range order: the result depends on the thread count, not on scheduling
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  for (int k = 0; k < used; ++k)
    for (int c = 0; c < 4; ++c) s[c] += part[static_cast<size_t>(4 * k + c)];
  for (int c = 0; c < 4; ++c) out4[c] = s[c];
  return LSR_OK;
}

}  // namespace

extern "C" int lsr_minmax_f32_cpu(const float* in, int64_t n, float* out2, void*, lsr_stream_t) {
  LSR_REQUIRE_HOST_FMA();
  LSR_REQUIRE_PTR(in);
  LSR_REQUIRE_PTR(out2);
  LSR_REQUIRE(n > 0, LSR_E_SHAPE, "n = %lld must be positive", (long long)n);
  LSR_REQUIRE_COUNT(n);
  float part[2 * kMaxWorkers];
  const int used = parallel_ranges_indexed(n, [&](int k, int64_t first, int64_t last) {
    float lo = INFINITY, hi = -INFINITY;
    for (int64_t i = first; i < last; ++i) {   // fminf / fmaxf: a NaN sample is skipped, as in the kernel
      lo = std::fmin(lo, in[i]);
      hi = std::fmax(hi, in[i]);
    }
    part[static_cast<size_t>(2 * k)] = lo;
    part[static_cast<size_t>(2 * k + 1)] = hi;
  });
  float lo = INFINITY, hi = -INFINITY;
  for (int k = 0; k < used; ++k) {
    lo = std::fmin(lo, part[static_cast<size_t>(2 * k)]);
    hi = std::fmax(hi, part[static_cast<size_t>(2 * k + 1)]);
  }
  out2[0] = lo;
  out2[1] = hi;
  return LSR_OK;
}

extern "C" int lsr_histogram_f32_cpu(const float* in, int64_t n, float vmin, float vmax, int nbins, unsigned* counts,
                                     lsr_stream_t) {
  LSR_REQUIRE_HOST_FMA();
  LSR_REQUIRE_PTR(in);
  LSR_REQUIRE_PTR(counts);
  LSR_REQUIRE(n > 0, LSR_E_SHAPE, "n = %lld must be positive", (long long)n);
  LSR_REQUIRE_COUNT(n);
  LSR_REQUIRE(n < (int64_t(1) << 32), LSR_E_UNSUPPORTED, "n = %lld: the bins count in 32 bits, histogram the volume in pieces",
              (long long)n);
  LSR_REQUIRE(nbins >= 1 && nbins <= kMaxBins, LSR_E_ARG, "nbins %d outside [1, %d]", nbins, kMaxBins);
  LSR_REQUIRE(vmax > vmin, LSR_E_ARG, "histogram range [%g, %g] is empty", vmin, vmax);
  std::memset(counts, 0, sizeof(unsigned) * static_cast<size_t>(nbins));
  const float range = vmax - vmin, fb = static_cast<float>(nbins);
  std::atomic<bool> failed{false};
  std::atomic<unsigned>* shared = reinterpret_cast<std::atomic<unsigned>*>(counts);
  static_assert(sizeof(std::atomic<unsigned>) == sizeof(unsigned), "plain counters behind the atomics");
  parallel_ranges(n, [&](int64_t first, int64_t last) {
    std::vector<unsigned> local(static_cast<size_t>(nbins), 0u);
    for (int64_t i = first; i < last; ++i) {
      const float v = in[i];
      if (v >= vmin && v <= vmax) {   // torch.histc: outside the range (and NaN) is dropped
        int bin = static_cast<int>((v - vmin) * fb / range);
        bin = bin >= nbins ? nbins - 1 : bin;
        ++local[static_cast<size_t>(bin)];
      }
    }
    for (int b = 0; b < nbins; ++b)
      if (local[static_cast<size_t>(b)]) shared[b].fetch_add(local[static_cast<size_t>(b)], std::memory_order_relaxed);
  }, failed);
  LSR_REQUIRE(!failed.load(), LSR_E_ARG, "lsr_histogram_f32_cpu: out of memory for a worker's bins");
  return LSR_OK;
}

extern "C" int lsr_weighted_centroid_f32_cpu(const float* in, int64_t Z, int64_t Y, int64_t X, float background,
                                             double* out4, void*, lsr_stream_t) {
  LSR_REQUIRE_HOST_FMA();
  return centroid_cpu<false>(in, Z, Y, X, background, out4);
}

extern "C" int lsr_mask_centroid_f32_cpu(const float* in, int64_t Z, int64_t Y, int64_t X, float threshold, double* out4,
                                         void*, lsr_stream_t) {
  LSR_REQUIRE_HOST_FMA();
  return centroid_cpu<true>(in, Z, Y, X, threshold, out4);
}

extern "C" int lsr_blur_reflect_f32_cpu(const float* in, float* out, int64_t Z, int64_t Y, int64_t X, int axis,
                                        const float* taps, int radius, float sub, float div, lsr_stream_t) {
  LSR_REQUIRE_HOST_FMA();
  if (int rc = check_volume(in, Z, Y, X)) return rc;
  LSR_REQUIRE_PTR(out);
  LSR_REQUIRE_PTR(taps);
  LSR_REQUIRE(in != out, LSR_E_ARG, "out must not alias in");
  LSR_REQUIRE(axis >= 0 && axis <= 2, LSR_E_ARG, "axis %d outside [0, 2]", axis);
  const int64_t L = axis == 0 ? Z : (axis == 1 ? Y : X);
  LSR_REQUIRE(radius >= 0 && radius <= kBlurMaxR, LSR_E_UNSUPPORTED, "radius %d outside [0, %d]", radius, kBlurMaxR);
  LSR_REQUIRE(radius < L, LSR_E_ARG, "reflect padding needs radius %d < axis length %lld", radius, (long long)L);
  const int64_t outer = axis == 0 ? 1 : (axis == 1 ? Z : Z * Y);
  const int64_t inner = axis == 0 ? Y * X : (axis == 1 ? X : 1);
  const int ntaps = 2 * radius + 1;
  const bool map = div != 0.0f;
  auto value = [&](float v) { return map ? (v - sub) / div : v; };
  // work items = (outer, position) pairs; every output is the FMA chain over its taps in ascending order
  std::atomic<bool> failed{false};
  parallel_ranges(outer * L, [&](int64_t first, int64_t last) {
    std::vector<float> acc(static_cast<size_t>(inner));
    for (int64_t item = first; item < last; ++item) {
      const int64_t o = item / L, a = item - o * L;
      const float* base = in + o * L * inner;
      if (inner == 1) {
        float c = 0.0f;
        for (int t = 0; t < ntaps; ++t) c = std::fmaf(taps[t], value(base[reflect(a + t - radius, L)]), c);
        out[o * L + a] = c;
        continue;
      }
      std::fill(acc.begin(), acc.end(), 0.0f);
      for (int t = 0; t < ntaps; ++t) {
        const float w = taps[t];
        const float* src = base + reflect(a + t - radius, L) * inner;
        float* ac = acc.data();
        for (int64_t i = 0; i < inner; ++i) ac[i] = std::fmaf(w, value(src[i]), ac[i]);
      }
      std::memcpy(out + (o * L + a) * inner, acc.data(), sizeof(float) * static_cast<size_t>(inner));
    }
  }, failed);
  LSR_REQUIRE(!failed.load(), LSR_E_ARG, "lsr_blur_reflect_f32_cpu: out of memory for a worker's row buffer");
  return LSR_OK;
}

extern "C" int lsr_match_shape_f32_cpu(const float* in, int64_t Zi, int64_t Yi, int64_t Xi, float* out, int64_t Zo,
                                       int64_t Yo, int64_t Xo, lsr_stream_t) {
  LSR_REQUIRE_HOST_FMA();
  if (int rc = check_volume(in, Zi, Yi, Xi)) return rc;
  LSR_REQUIRE_PTR(out);
  if (int rc = check_volume(out, Zo, Yo, Xo)) return rc;
  // reflect padding needs pad < dim on each side (F.pad's rule, tracking.py:290-300)
  const int64_t ni[3] = {Zi, Yi, Xi}, no[3] = {Zo, Yo, Xo};
  for (int k = 0; k < 3; ++k)
    LSR_REQUIRE(no[k] <= ni[k] || (no[k] - ni[k] + 1) / 2 < ni[k], LSR_E_ARG,
                "axis %d: cannot reflect-pad %lld samples to %lld", k, (long long)ni[k], (long long)no[k]);
  parallel_ranges(Zo * Yo, [&](int64_t first, int64_t last) {
    for (int64_t r = first; r < last; ++r) {
      const int64_t zo = r / Yo, yo = r - zo * Yo;
      const float* src = in + (match_index(zo, Zi, Zo) * Yi + match_index(yo, Yi, Yo)) * Xi;
      float* dst = out + r * Xo;
      for (int64_t xo = 0; xo < Xo; ++xo) dst[xo] = src[match_index(xo, Xi, Xo)];
    }
  });
  return LSR_OK;
}

namespace {
// prod = a * conj(b): (ar + i ai)(br - i bi), the products and sums torch's complex multiply performs
template <bool INTO_B>
int cross_power_cpu(float* a, float* b, int64_t n) {
  LSR_REQUIRE_PTR(a);
  LSR_REQUIRE_PTR(b);
  LSR_REQUIRE(n > 0, LSR_E_SHAPE, "n = %lld must be positive", (long long)n);
  LSR_REQUIRE_COUNT(n);
  parallel_ranges(n, [&](int64_t first, int64_t last) {
    for (int64_t i = first; i < last; ++i) {
      const float ar = a[2 * i], ai = a[2 * i + 1], br = b[2 * i], bi = b[2 * i + 1];
      const float re = ar * br + ai * bi, im = ai * br - ar * bi;
      float* dst = INTO_B ? b : a;
      dst[2 * i] = re;
      dst[2 * i + 1] = im;
    }
  });
  return LSR_OK;
}
}  // namespace

extern "C" int lsr_cross_power_c64_cpu(float* a, const float* b, int64_t n, lsr_stream_t) {
  LSR_REQUIRE_HOST_FMA();
  return cross_power_cpu<false>(a, const_cast<float*>(b), n);
}

extern "C" int lsr_cross_power_into_c64_cpu(const float* a, float* b, int64_t n, lsr_stream_t) {
  LSR_REQUIRE_HOST_FMA();
  return cross_power_cpu<true>(const_cast<float*>(a), b, n);
}

extern "C" int lsr_peak_abs_shifted_f32_cpu(const float* in, int64_t Z, int64_t Y, int64_t X, long long* out_index, void*,
                                            lsr_stream_t) {
  LSR_REQUIRE_HOST_FMA();
  if (int rc = check_volume(in, Z, Y, X)) return rc;
  LSR_REQUIRE_PTR(out_index);
  // the largest |v|; ties: the smallest flat index in fftshift order (torch.argmax returns the first maximum)
  struct Best { float v; unsigned long long i; };
  Best part[kMaxWorkers];
  auto merge = [](Best& b, float v, unsigned long long i) {
    if (v > b.v || (v == b.v && i < b.i)) { b.v = v; b.i = i; }
  };
  const int used = parallel_ranges_indexed(Z * Y, [&](int k, int64_t first, int64_t last) {
    Best best{-1.0f, ~0ull};
    for (int64_t r = first; r < last; ++r) {
      const int64_t z = r / Y, y = r - z * Y;
      const int64_t zs = (z + Z / 2) % Z, ys = (y + Y / 2) % Y;   // fftshift: index i -> (i + n/2) % n
      const unsigned long long base = (static_cast<unsigned long long>(zs) * Y + ys) * X;
      const float* row = in + r * X;
      for (int64_t x = 0; x < X; ++x) {
        const float v = std::fabs(row[x]);
        if (v >= best.v) merge(best, v, base + static_cast<unsigned long long>((x + X / 2) % X));
      }
    }
    part[static_cast<size_t>(k)] = best;
  });
  Best best{-1.0f, ~0ull};
  for (int k = 0; k < used; ++k) merge(best, part[static_cast<size_t>(k)].v, part[static_cast<size_t>(k)].i);
  out_index[0] = static_cast<long long>(best.i);
  return LSR_OK;
}
