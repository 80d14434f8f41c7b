// Host twins of the hot-path entry points (no kernel in this file): the same signatures with HOST pointers,
// the same argument checks, the same arithmetic in the same order -- so the results equal the device
// kernels' bit for bit -- for the boxes where the reference itself falls back to the CPU
// (`torch.device("cuda" if torch.cuda.is_available() else "cpu")`, shrimpy/preprocessing.py:78-82; its CI has
// no GPU, shrimpy/tests/conftest.py:11-17) and for BASELINE config 1 ("deskew-only via the CPU path").
// This is product code: it does not call, link or read anything under oracle/.
//
//   lsr_deskew_f32_cpu / lsr_deskew_u16_cpu   <->  lsr_deskew_f32 / lsr_deskew_u16     (deskew.hip)
//   lsr_affine_f32_cpu                        <->  lsr_affine_f32                      (affine.hip)
//   lsr_average_slices_f32_cpu                <->  lsr_average_slices_f32              (deskew.hip)
//   lsr_correlate_sep_f32_cpu                 <->  lsr_correlate_sep_f32               (correlate.hip)
//   lsr_correlate_dense_f32_cpu               <->  lsr_correlate_dense_f32             (correlate.hip)
//   lsr_rl_dense_f32_cpu                      <->  lsr_rl_dense_f32                    (correlate.hip)
//   lsr_flatfield_pattern_f32_cpu / _u16_cpu  <->  lsr_flatfield_pattern_f32 / _u16    (flatfield.hip)
//   lsr_flatfield_apply_f32_cpu / _u16_cpu    <->  lsr_flatfield_apply_f32 / _u16      (flatfield.hip)
//
// Arithmetic (what "the same" means):
//   resamplers -- coordinates ((zo*m0 + yo*m1) + xo*m2) + shift, weights w0 = 1 - f, w1 = 1 - w0 and the
//     8-corner sum ((v*wz)*wy)*wx accumulated z-major in fp64, every operation rounded on its own (the TU is
//     built with -ffp-contract=off), the result rounded to f32 once: scipy.ndimage.affine_transform(order=1);
//   averaging  -- ((d0 + d1) + ...) / n in f32, the last group edge-padded;
//   stencils   -- f32 FMA chains in tap order (separable: x, then y, then z; dense: z-major), zeros outside the
//     volume, the Richardson-Lucy epilogues of correlate.hip.
// The `stream` argument is ignored (kept so the signatures are identical).  Threads: plain std::thread over
// output planes, at most lsr_set_host_threads(n) of them (default 1) -- no OpenMP runtime enters the process
// (the reference's warning about torch plus a second OpenMP, shrimpy/tests/conftest.py:11-17).

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <mutex>
#include <new>
#include <system_error>
#include <thread>
#include <vector>

#include "common.hpp"
#include "correlate_common.hpp"
#include "host_parallel.hpp"

namespace {

using lsr::parallel_ranges;
std::atomic<int>& g_threads = lsr::g_host_threads;
constexpr int kMaxAvg = 16;    // as deskew.hip
// the stencil kernels hold 15 taps per axis (31 along z, correlate_z.hip); larger dense PSFs run in the Fourier
// domain on the device (deconvolve_fft.py, up to 129 taps per axis) -- the twins' loops take any count
constexpr int kMaxTaps = 129;
constexpr int kMaxZTaps = 129;

// acc[x] = fma(w, v(x + shift), acc[x]) over a whole row, v = row[...] inside [0, X) and 0 outside it (row == nullptr:
// a row of zeros) -- the FMA with 0 is executed, as the kernels execute it; the middle part is a plain packed loop.
inline void row_fma(float* __restrict__ acc, float w, const float* __restrict__ row, int64_t shift, int64_t X) {
  int64_t lo = 0, hi = 0;
  if (row != nullptr) {
    lo = shift < 0 ? (-shift < X ? -shift : X) : 0;
    hi = shift > 0 ? (X - shift > lo ? X - shift : lo) : X;
    if (hi < lo) hi = lo;
  }
  for (int64_t x = 0; x < lo; ++x) acc[x] = std::fmaf(w, 0.0f, acc[x]);
  for (int64_t x = lo; x < hi; ++x) acc[x] = std::fmaf(w, row[x + shift], acc[x]);
  for (int64_t x = hi; x < X; ++x) acc[x] = std::fmaf(w, 0.0f, acc[x]);
}

struct AxisTap {
  int64_t i0, i1;
  double w0, w1;
  bool out0, out1;
};

template <bool GRID>
inline bool axis_tap(double c, int64_t n, AxisTap& t) {
  if (!GRID && (c < 0.0 || c > static_cast<double>(n - 1))) return false;
  const double fl = std::floor(c);
  const double f = c - fl;
  t.w0 = 1.0 - f;
  t.w1 = 1.0 - t.w0;
  if (!GRID) {
    t.i0 = static_cast<int64_t>(fl);
    t.i1 = t.i0 + 1 < n ? t.i0 + 1 : n - 1;
    t.out0 = t.out1 = false;
  } else {
    const double lo = fl < -2.0 ? -2.0 : fl;
    const int64_t start = static_cast<int64_t>(lo > static_cast<double>(n) + 1.0 ? static_cast<double>(n) + 1.0 : lo);
    t.out0 = start < 0 || start >= n;
    t.out1 = start + 1 < 0 || start + 1 >= n;
    t.i0 = start < 0 ? 0 : (start > n - 1 ? n - 1 : start);
    t.i1 = start + 1 < 0 ? 0 : (start + 1 > n - 1 ? n - 1 : start + 1);
  }
  return true;
}

inline double coord(double zo, double yo, double xo, const double* row) {
  double c = zo * row[0];
  c = c + yo * row[1];
  c = c + xo * row[2];
  return c + row[3];
}

template <typename T, bool GRID>
inline float sample(const T* in, int64_t Z, int64_t Y, int64_t X, const double M[12], double zo, double yo, double xo,
                    float cval) {
  AxisTap tz, ty, tx;
  if (!axis_tap<GRID>(coord(zo, yo, xo, M), Z, tz) || !axis_tap<GRID>(coord(zo, yo, xo, M + 4), Y, ty) ||
      !axis_tap<GRID>(coord(zo, yo, xo, M + 8), X, tx))
    return cval;
  const double cv = static_cast<double>(cval);
  double t = 0.0;
  for (int a = 0; a < 2; ++a) {
    const int64_t oz = (a ? tz.i1 : tz.i0) * Y * X;
    const double wz = a ? tz.w1 : tz.w0;
    const bool bz = a ? tz.out1 : tz.out0;
    for (int b = 0; b < 2; ++b) {
      const int64_t oy = oz + (b ? ty.i1 : ty.i0) * X;
      const double wy = b ? ty.w1 : ty.w0;
      const bool by = b ? ty.out1 : ty.out0;
      for (int c = 0; c < 2; ++c) {
        double v = static_cast<double>(in[oy + (c ? tx.i1 : tx.i0)]);
        if (GRID && (bz || by || (c ? tx.out1 : tx.out0))) v = cv;
        v = v * wz;
        v = v * wy;
        v = v * (c ? tx.w1 : tx.w0);
        t = t + v;
      }
    }
  }
  return static_cast<float>(t);
}

bool is_integer(double v) { return v == static_cast<double>(static_cast<int64_t>(v)); }

int check_matrix(const double M[12]) {
  LSR_REQUIRE_PTR(M);
  for (int i = 0; i < 12; ++i) LSR_REQUIRE(M[i] == M[i] && M[i] - M[i] == 0.0, LSR_E_ARG, "M[%d] is not finite", i);
  return LSR_OK;
}

template <typename T>
int deskew_cpu(const T* in, int64_t Z, int64_t Y, int64_t X, float* out, int64_t Zo, int64_t Yo, int64_t Xo,
               int64_t out_pitch, int64_t out_plane, int64_t Zd, const double M[12], int avg_n, float cval = 0.0f) {
  LSR_REQUIRE_PTR(in);
  LSR_REQUIRE_PTR(out);
  if (int rc = check_matrix(M)) return rc;
  LSR_REQUIRE(Z > 0 && Y > 0 && X > 0, LSR_E_SHAPE, "raw shape (%lld,%lld,%lld) must be positive", (long long)Z, (long long)Y,
              (long long)X);
  LSR_REQUIRE_VOLUME(Z, Y, X);
  LSR_REQUIRE(Zo > 0 && Yo > 0 && Xo > 0 && Zd > 0, LSR_E_SHAPE, "output shape (%lld,%lld,%lld) / Zd %lld must be positive",
              (long long)Zo, (long long)Yo, (long long)Xo, (long long)Zd);
  LSR_REQUIRE_VOLUME(Zo, Yo, Xo);
  LSR_REQUIRE_VOLUME(Zd, Yo, Xo);
  LSR_REQUIRE(avg_n >= 1 && avg_n <= kMaxAvg, LSR_E_ARG, "avg_n %d outside [1,%d]", avg_n, kMaxAvg);
  LSR_REQUIRE(Zo == lsr::ceil_div(Zd, avg_n), LSR_E_SHAPE, "Zo %lld != ceil(Zd %lld / avg_n %d)", (long long)Zo,
              (long long)Zd, avg_n);
  LSR_REQUIRE(out_pitch >= Xo && out_plane >= Yo * out_pitch, LSR_E_SHAPE,
              "output strides (%lld, %lld) are smaller than the output plane (%lld x %lld)", (long long)out_pitch,
              (long long)out_plane, (long long)Yo, (long long)Xo);
  const bool structured = M[1] == 0.0 && (M[4] == 1.0 || M[4] == -1.0) && M[5] == 0.0 && M[6] == 0.0 && is_integer(M[7]) &&
                          M[8] == 0.0 && (M[9] == 1.0 || M[9] == -1.0) && M[10] == 0.0 && is_integer(M[11]);
  LSR_REQUIRE(structured, LSR_E_UNSUPPORTED,
              "matrix is not a deskew shear (rows 1,2 must be signed unit axes with integer offsets, M[0][1] == 0): use "
              "lsr_affine_f32_cpu + lsr_average_slices_f32_cpu");
  // The shear's structure (deskew.hip uses the same): z_in depends on (zd, xo) only, y_in on zd only, x_in on yo only,
  // and y_in, x_in are whole numbers -- so of scipy's eight corners two carry weight (the z neighbours at (y_in, x_in))
  // and the other six add zeros.  Per deskewed plane the z taps are tabulated once along xo and reused for every yo;
  // a voxel is  float( (0.0 + double(v0) * w0) + double(v1) * w1 ), the generic resampler's own sum without its
  // zero terms (finite inputs: a zero-weight corner holding inf / NaN would poison scipy's sum and the generic twin's,
  // not the kernel's and not this one).  Then the average of avg_n planes in f32, as the kernel accumulates it.
  std::vector<int64_t> xin_v(static_cast<size_t>(Yo));
  for (int64_t yo = 0; yo < Yo; ++yo) {     // x_in(yo), or -1 where it leaves the stack
    const double c = coord(0.0, static_cast<double>(yo), 0.0, M + 8);
    xin_v[static_cast<size_t>(yo)] = (c < 0.0 || c > static_cast<double>(X - 1)) ? -1 : static_cast<int64_t>(std::floor(c));
  }
  const int64_t* const xin = xin_v.data();
  const int64_t plane = Y * X;
  std::atomic<bool> failed{false};
  parallel_ranges(Zo, [&](int64_t z_first, int64_t z_last) {
    std::vector<int64_t> o0_v(static_cast<size_t>(Xo)), o1_v(static_cast<size_t>(Xo));
    std::vector<double> w0_v(static_cast<size_t>(Xo)), w1_v(static_cast<size_t>(Xo));
    std::vector<float> d_v(static_cast<size_t>(Xo));
    int64_t* const o0 = o0_v.data();
    int64_t* const o1 = o1_v.data();
    double* const w0 = w0_v.data();
    double* const w1 = w1_v.data();
    float* const d = d_v.data();
    for (int64_t zo = z_first; zo < z_last; ++zo) {
      for (int k = 0; k < avg_n; ++k) {
        const int64_t zd = zo * avg_n + k < Zd - 1 ? zo * avg_n + k : Zd - 1;
        const double cy = coord(static_cast<double>(zd), 0.0, 0.0, M + 4);
        const bool y_ok = !(cy < 0.0 || cy > static_cast<double>(Y - 1));
        const int64_t y_in = y_ok ? static_cast<int64_t>(std::floor(cy)) : 0;
        for (int64_t xo = 0; xo < Xo; ++xo) {      // the z taps of this deskewed plane (o0 < 0: outside the stack)
          AxisTap tz;
          if (y_ok && axis_tap<false>(coord(static_cast<double>(zd), 0.0, static_cast<double>(xo), M), Z, tz)) {
            o0[xo] = tz.i0 * plane + y_in * X;
            o1[xo] = tz.i1 * plane + y_in * X;
            w0[xo] = tz.w0;
            w1[xo] = tz.w1;
          } else {
            o0[xo] = -1;
          }
        }
        for (int64_t yo = 0; yo < Yo; ++yo) {
          float* const row = out + zo * out_plane + yo * out_pitch;
          const int64_t xi = xin[yo];
          if (xi < 0) {
            for (int64_t xo = 0; xo < Xo; ++xo) d[xo] = cval;
          } else {
            const T* const col = in + xi;
            for (int64_t xo = 0; xo < Xo; ++xo) {
              if (o0[xo] < 0) {
                d[xo] = cval;
                continue;
              }
              double t = 0.0 + static_cast<double>(col[o0[xo]]) * w0[xo];
              t = t + static_cast<double>(col[o1[xo]]) * w1[xo];
              d[xo] = static_cast<float>(t);
            }
          }
          if (k == 0) {             // (avg_n == 1: the sample itself, no division)
            for (int64_t xo = 0; xo < Xo; ++xo) row[xo] = d[xo];
          } else if (k + 1 < avg_n) {
            for (int64_t xo = 0; xo < Xo; ++xo) row[xo] = row[xo] + d[xo];
          } else {
            const float n = static_cast<float>(avg_n);
            for (int64_t xo = 0; xo < Xo; ++xo) row[xo] = (row[xo] + d[xo]) / n;
          }
        }
      }
    }
  }, failed);
  if (failed.load()) return lsr::fail(LSR_E_ARG, "out of host memory for the per-thread tap tables");
  return LSR_OK;
}

int check_corr(const float* in, float* out, const float* aux, int64_t Z, int64_t Y, int64_t X, int pz, int py, int px,
               int epilogue) {
  LSR_REQUIRE_PTR(in);
  LSR_REQUIRE_PTR(out);
  LSR_REQUIRE(Z > 0 && Y > 0 && X > 0, LSR_E_SHAPE, "shape (%lld,%lld,%lld) must be positive", (long long)Z, (long long)Y,
              (long long)X);
  LSR_REQUIRE_VOLUME(Z, Y, X);
  LSR_REQUIRE(pz >= 1 && py >= 1 && px >= 1 && (pz & 1) && (py & 1) && (px & 1) && pz <= kMaxZTaps && py <= kMaxTaps &&
                  px <= kMaxTaps,
              LSR_E_UNSUPPORTED, "PSF taps (%d,%d,%d) must be odd, <= %d in plane and <= %d along z", pz, py, px, kMaxTaps,
              kMaxZTaps);
  LSR_REQUIRE(epilogue == LSR_EPI_NONE || epilogue == LSR_EPI_RATIO || epilogue == LSR_EPI_UPDATE, LSR_E_ARG,
              "unknown epilogue %d", epilogue);
  if (epilogue != LSR_EPI_NONE) LSR_REQUIRE_PTR(aux);
  LSR_REQUIRE(in != out, LSR_E_ARG, "out must not alias in");
  return LSR_OK;
}

// H^T 1 of the dense form, as correlate.hip evaluates it from the prefix-sum table
double dense_norm(const double* P, int pz, int py, int px, int64_t Z, int64_t Y, int64_t X, int64_t z, int64_t y, int64_t x) {
  const int cz = pz / 2, cy = py / 2, cx = px / 2;
  auto lo = [](int64_t v) { return static_cast<int>(v > 0 ? v : 0); };
  auto hi = [](int64_t n, int64_t v) { return static_cast<int>(v < n ? v : n); };
  const int a0 = lo(cz - z), a1 = hi(pz, Z - z + cz), b0 = lo(cy - y), b1 = hi(py, Y - y + cy), c0 = lo(cx - x),
            c1 = hi(px, X - x + cx);
  const int sb = px + 1, sa = (py + 1) * sb;
  if (a0 == 0 && a1 == pz && b0 == 0 && b1 == py && c0 == 0 && c1 == px) return P[pz * sa + py * sb + px];
  return ((P[a1 * sa + b1 * sb + c1] - P[a0 * sa + b1 * sb + c1]) - (P[a1 * sa + b0 * sb + c1] - P[a0 * sa + b0 * sb + c1])) -
         ((P[a1 * sa + b1 * sb + c0] - P[a0 * sa + b1 * sb + c0]) - (P[a1 * sa + b0 * sb + c0] - P[a0 * sa + b0 * sb + c0]));
}

}  // namespace

extern "C" int lsr_set_host_threads(int n) {
  LSR_REQUIRE(n >= 1 && n <= 1024, LSR_E_ARG, "host threads %d outside [1, 1024]", n);
  g_threads.store(n, std::memory_order_relaxed);
  return LSR_OK;
}

extern "C" int lsr_get_host_threads(void) { return g_threads.load(std::memory_order_relaxed); }

extern "C" int lsr_deskew_f32_cpu(const float* in, int64_t Z, int64_t Y, int64_t X, float* out, int64_t Zo, int64_t Yo,
                                  int64_t Xo, int64_t out_pitch, int64_t out_plane, int64_t Zd, const double M[12],
                                  int avg_n, lsr_stream_t) {
  LSR_REQUIRE_HOST_FMA();
  return deskew_cpu(in, Z, Y, X, out, Zo, Yo, Xo, out_pitch, out_plane, Zd, M, avg_n);
}

// ... with the value outside the stack (scipy's cval; `cval` is a HOST scalar here, NULL = 0): "constant" border only
extern "C" int lsr_deskew_cval_cpu(const void* in, int in_u16, int64_t Z, int64_t Y, int64_t X, float* out, int64_t Zo,
                                   int64_t Yo, int64_t Xo, int64_t out_pitch, int64_t out_plane, int64_t Zd,
                                   const double M[12], int avg_n, int mode, const float* flat_pattern,
                                   const float* flat_mean, const float* cval, lsr_stream_t) {
  LSR_REQUIRE_HOST_FMA();
  LSR_REQUIRE(mode == LSR_MODE_CONSTANT, LSR_E_UNSUPPORTED, "the host deskew twin has the \"constant\" border only");
  LSR_REQUIRE(flat_pattern == nullptr && flat_mean == nullptr, LSR_E_UNSUPPORTED, "the host deskew twin takes a corrected stack");
  const float cv = cval ? cval[0] : 0.0f;
  if (in_u16) return deskew_cpu(static_cast<const uint16_t*>(in), Z, Y, X, out, Zo, Yo, Xo, out_pitch, out_plane, Zd, M, avg_n, cv);
  return deskew_cpu(static_cast<const float*>(in), Z, Y, X, out, Zo, Yo, Xo, out_pitch, out_plane, Zd, M, avg_n, cv);
}

extern "C" int lsr_deskew_u16_cpu(const uint16_t* in, int64_t Z, int64_t Y, int64_t X, float* out, int64_t Zo, int64_t Yo,
                                  int64_t Xo, int64_t out_pitch, int64_t out_plane, int64_t Zd, const double M[12],
                                  int avg_n, lsr_stream_t) {
  LSR_REQUIRE_HOST_FMA();
  return deskew_cpu(in, Z, Y, X, out, Zo, Yo, Xo, out_pitch, out_plane, Zd, M, avg_n);
}

extern "C" int lsr_affine_f32_cpu(const float* in, int64_t Zi, int64_t Yi, int64_t Xi, float* out, int64_t Zo, int64_t Yo,
                                  int64_t Xo, const double M[12], float cval, int mode, lsr_stream_t) {
  LSR_REQUIRE_HOST_FMA();
  LSR_REQUIRE_PTR(in);
  LSR_REQUIRE_PTR(out);
  if (int rc = check_matrix(M)) return rc;
  LSR_REQUIRE(Zi > 0 && Yi > 0 && Xi > 0 && Zo > 0 && Yo > 0 && Xo > 0, LSR_E_SHAPE, "shapes must be positive");
  LSR_REQUIRE_VOLUME(Zi, Yi, Xi);
  LSR_REQUIRE_VOLUME(Zo, Yo, Xo);
  LSR_REQUIRE(mode == LSR_MODE_CONSTANT || mode == LSR_MODE_GRID_CONSTANT, LSR_E_ARG,
              "mode %d: LSR_MODE_CONSTANT or LSR_MODE_GRID_CONSTANT (the f32-interpolation flag has no host twin: "
              "the host arithmetic is always scipy's fp64)", mode);
  const bool grid = mode == LSR_MODE_GRID_CONSTANT;
  parallel_ranges(Zo, [&](int64_t z_first, int64_t z_last) {
    for (int64_t zo = z_first; zo < z_last; ++zo)
      for (int64_t yo = 0; yo < Yo; ++yo) {
        float* row = out + (zo * Yo + yo) * Xo;
        for (int64_t xo = 0; xo < Xo; ++xo)
          row[xo] = grid ? sample<float, true>(in, Zi, Yi, Xi, M, double(zo), double(yo), double(xo), cval)
                         : sample<float, false>(in, Zi, Yi, Xi, M, double(zo), double(yo), double(xo), cval);
      }
  });
  return LSR_OK;
}

extern "C" int lsr_average_slices_f32_cpu(const float* in, int64_t Zd, int64_t Y, int64_t X, float* out, int64_t Zo, int avg_n,
                                          lsr_stream_t) {
  LSR_REQUIRE_HOST_FMA();
  LSR_REQUIRE_PTR(in);
  LSR_REQUIRE_PTR(out);
  LSR_REQUIRE(Zd > 0 && Y > 0 && X > 0 && Zo > 0, LSR_E_SHAPE, "shape must be positive");
  LSR_REQUIRE_VOLUME(Zd, Y, X);
  LSR_REQUIRE_VOLUME(Zo, Y, X);
  LSR_REQUIRE(avg_n >= 1 && avg_n <= kMaxAvg, LSR_E_ARG, "avg_n %d outside [1,%d]", avg_n, kMaxAvg);
  LSR_REQUIRE(Zo == lsr::ceil_div(Zd, avg_n), LSR_E_SHAPE, "Zo %lld != ceil(Zd %lld / avg_n %d)", (long long)Zo,
              (long long)Zd, avg_n);
  const int64_t plane = Y * X;
  parallel_ranges(Zo, [&](int64_t z_first, int64_t z_last) {
    for (int64_t zo = z_first; zo < z_last; ++zo)
      for (int64_t r = 0; r < plane; ++r) {
        float acc = 0.0f;
        for (int k = 0; k < avg_n; ++k) {
          const int64_t zd = zo * avg_n + k < Zd - 1 ? zo * avg_n + k : Zd - 1;
          const float d = in[zd * plane + r];
          acc = k == 0 ? d : acc + d;
        }
        out[zo * plane + r] = avg_n > 1 ? acc / static_cast<float>(avg_n) : acc;
      }
  });
  return LSR_OK;
}

// The reduction scalars of an UPDATE pass (correlate_common.hpp: flux, change, total), as the kernels add them: a
// range's rows in f64 here, one locked add per range.
namespace {
struct HostStats {
  double flux = 0.0, change = 0.0, total = 0.0;
  void add(float x_old, float xu, float x_new) {
    flux += xu;
    change += std::fabs(static_cast<double>(x_new) - static_cast<double>(x_old));
    total += x_new;
  }
  void flush(double* dst, std::mutex& m) const {
    std::lock_guard<std::mutex> g(m);
    dst[0] += flux; dst[1] += change; dst[2] += total;
  }
};
}  // namespace

extern "C" int lsr_correlate_sep_f32_cpu(const float* in, float* out, const float* aux, int64_t Z, int64_t Y, int64_t X,
                                         const float* wz, int pz, const float* wy, int py, const float* wx, int px,
                                         int epilogue, float eps, const float* nz, const float* ny, const float* nx,
                                         lsr_stream_t stream) {
  return lsr_correlate_sep_stats_f32_cpu(in, out, aux, Z, Y, X, wz, pz, wy, py, wx, px, epilogue, eps, nz, ny, nx, nullptr,
                                         stream);
}

extern "C" int lsr_correlate_sep_stats_f32_cpu(const float* in, float* out, const float* aux, int64_t Z, int64_t Y, int64_t X,
                                               const float* wz, int pz, const float* wy, int py, const float* wx, int px,
                                               int epilogue, float eps, const float* nz, const float* ny, const float* nx,
                                               double* stats, lsr_stream_t) {
  LSR_REQUIRE_HOST_FMA();
  if (int rc = check_corr(in, out, aux, Z, Y, X, pz, py, px, epilogue)) return rc;
  LSR_REQUIRE_PTR(wz);
  LSR_REQUIRE_PTR(wy);
  LSR_REQUIRE_PTR(wx);
  if (epilogue == LSR_EPI_UPDATE) {
    LSR_REQUIRE_PTR(nz);
    LSR_REQUIRE_PTR(ny);
    LSR_REQUIRE_PTR(nx);
  }
  const int64_t plane = Y * X;
  const int cz = pz / 2, cy = py / 2, cx = px / 2;
  // in-plane passes of every plane first (x then y, f32 FMA chains from 0), then the z chain + epilogue.
  // Every voxel sees exactly the kernels' operations in the kernels' order -- a tap that falls outside the volume is
  // an FMA with 0, not a skipped step -- but the loops run taps outermost and x innermost over whole rows, so that
  // the compiler turns each into packed FMAs (the per-voxel tap loop was a serial dependency chain: 4x slower than
  // scipy's correlate1d on one core; this form is faster than it).
  std::atomic<bool> failed{false};
  std::vector<float> filtered;
  try {
    filtered.resize(static_cast<size_t>(Z * plane));
  } catch (const std::bad_alloc&) {
    return lsr::fail(LSR_E_ARG, "out of host memory for a %lld-voxel intermediate", (long long)(Z * plane));
  }
  parallel_ranges(Z, [&](int64_t z_first, int64_t z_last) {
    std::vector<float> rows_v(static_cast<size_t>(plane));
    float* const rows = rows_v.data();
    for (int64_t z = z_first; z < z_last; ++z) {
      const float* const src = in + z * plane;
      for (int64_t y = 0; y < Y; ++y) {
        const float* __restrict__ srow = src + y * X;
        float* __restrict__ acc = rows + y * X;
        for (int64_t x = 0; x < X; ++x) acc[x] = 0.0f;
        for (int c = 0; c < px; ++c) row_fma(acc, wx[c], srow, c - cx, X);
      }
      float* const dst = filtered.data() + z * plane;
      for (int64_t y = 0; y < Y; ++y) {
        float* __restrict__ acc = dst + y * X;
        for (int64_t x = 0; x < X; ++x) acc[x] = 0.0f;
        for (int b = 0; b < py; ++b) {
          const float w = wy[b];
          const int64_t gy = y + b - cy;
          if (gy >= 0 && gy < Y) {
            const float* __restrict__ r = rows + gy * X;
            for (int64_t x = 0; x < X; ++x) acc[x] = std::fmaf(w, r[x], acc[x]);
          } else {
            for (int64_t x = 0; x < X; ++x) acc[x] = std::fmaf(w, 0.0f, acc[x]);
          }
        }
      }
    }
  }, failed);
  if (failed.load()) return lsr::fail(LSR_E_ARG, "out of host memory for the per-thread row buffers");
  std::mutex stats_lock;
  parallel_ranges(Z * Y, [&](int64_t r_first, int64_t r_last) {
    std::vector<float> c_v(static_cast<size_t>(X));
    float* __restrict__ c = c_v.data();
    HostStats st;
    for (int64_t zy = r_first; zy < r_last; ++zy) {
      const int64_t z = zy / Y, y = zy - z * Y;
      // the march of correlate.hip: the first plane's term is a plain product, the others FMAs onto it
      for (int a = 0; a < pz; ++a) {
        const int64_t zi = z + a - cz;
        const float w = wz[a];
        const bool inside = zi >= 0 && zi < Z;
        const float* __restrict__ p = filtered.data() + (inside ? zi : 0) * plane + y * X;
        if (a == 0) {
          if (inside) for (int64_t x = 0; x < X; ++x) c[x] = w * p[x];
          else for (int64_t x = 0; x < X; ++x) c[x] = w * 0.0f;
        } else if (inside) {
          for (int64_t x = 0; x < X; ++x) c[x] = std::fmaf(w, p[x], c[x]);
        } else {
          for (int64_t x = 0; x < X; ++x) c[x] = std::fmaf(w, 0.0f, c[x]);
        }
      }
      const int64_t o = zy * X;
      float* __restrict__ dst = out + o;
      if (epilogue == LSR_EPI_RATIO) {
        const float* __restrict__ ax = aux + o;
        for (int64_t x = 0; x < X; ++x) dst[x] = ax[x] / (c[x] + eps);
      } else if (epilogue == LSR_EPI_UPDATE) {
        const float* __restrict__ ax = aux + o;
        const float nzy = nz[z] * ny[y];
        if (stats == nullptr) {
          for (int64_t x = 0; x < X; ++x) dst[x] = ax[x] * c[x] / (nzy * nx[x]);
        } else {   // (dst may be aux itself: read, then write)
          for (int64_t x = 0; x < X; ++x) {
            const float a = ax[x], ac = a * c[x], v = ac / (nzy * nx[x]);
            dst[x] = v;
            st.add(a, ac, v);
          }
        }
      } else {
        for (int64_t x = 0; x < X; ++x) dst[x] = c[x];
      }
    }
    if (stats != nullptr && epilogue == LSR_EPI_UPDATE) st.flush(stats, stats_lock);
  }, failed);
  if (failed.load()) return lsr::fail(LSR_E_ARG, "out of host memory for the per-thread row buffers");
  return LSR_OK;
}

extern "C" int lsr_correlate_dense_f32_cpu(const float* in, float* out, const float* aux, int64_t Z, int64_t Y, int64_t X,
                                           const float* w, int pz, int py, int px, int epilogue, float eps,
                                           const double* norm_table, lsr_stream_t stream) {
  return lsr_correlate_dense_stats_f32_cpu(in, out, aux, Z, Y, X, w, pz, py, px, epilogue, eps, norm_table, nullptr, stream);
}

extern "C" int lsr_correlate_dense_stats_f32_cpu(const float* in, float* out, const float* aux, int64_t Z, int64_t Y,
                                                 int64_t X, const float* w, int pz, int py, int px, int epilogue, float eps,
                                                 const double* norm_table, double* stats, lsr_stream_t) {
  LSR_REQUIRE_HOST_FMA();
  if (int rc = check_corr(in, out, aux, Z, Y, X, pz, py, px, epilogue)) return rc;
  LSR_REQUIRE_PTR(w);
  if (epilogue == LSR_EPI_UPDATE) LSR_REQUIRE_PTR(norm_table);
  const int64_t plane = Y * X;
  const int cz = pz / 2, cy = py / 2, cx = px / 2;
  // per output row: taps outermost (planes in z order, within a plane y-major, then x: the order the march accumulates
  // in), x innermost -- every voxel's chain is the same sequence of FMAs, a row at a time (packed FMAs on the host)
  std::atomic<bool> failed{false};
  std::mutex stats_lock;
  parallel_ranges(Z * Y, [&](int64_t r_first, int64_t r_last) {
    std::vector<float> c_v(static_cast<size_t>(X));
    float* __restrict__ c = c_v.data();
    HostStats st;
    for (int64_t zy = r_first; zy < r_last; ++zy) {
      const int64_t z = zy / Y, y = zy - z * Y;
      for (int64_t x = 0; x < X; ++x) c[x] = 0.0f;
      for (int a = 0; a < pz; ++a) {
        const int64_t zi = z + a - cz;
        if (zi < 0 || zi >= Z) continue;      // (a whole plane of zeros leaves the chain unchanged)
        for (int b = 0; b < py; ++b) {
          const int64_t gy = y + b - cy;
          const float* row = gy >= 0 && gy < Y ? in + zi * plane + gy * X : nullptr;
          for (int k = 0; k < px; ++k) row_fma(c, w[(a * py + b) * px + k], row, k - cx, X);
        }
      }
      const int64_t o = zy * X;
      float* __restrict__ dst = out + o;
      if (epilogue == LSR_EPI_RATIO) {
        for (int64_t x = 0; x < X; ++x) dst[x] = aux[o + x] / (c[x] + eps);
      } else if (epilogue == LSR_EPI_UPDATE) {
        for (int64_t x = 0; x < X; ++x) {
          const float a = aux[o + x], ac = a * c[x];
          const float v = ac / static_cast<float>(dense_norm(norm_table, pz, py, px, Z, Y, X, z, y, x));
          dst[x] = v;
          if (stats != nullptr) st.add(a, ac, v);
        }
      } else {
        for (int64_t x = 0; x < X; ++x) dst[x] = c[x];
      }
    }
    if (stats != nullptr && epilogue == LSR_EPI_UPDATE) st.flush(stats, stats_lock);
  }, failed);
  if (failed.load()) return lsr::fail(LSR_E_ARG, "out of host memory for the per-thread row buffers");
  return LSR_OK;
}

// The whole loop of lsr_rl_dense_f32 on host memory: `iters` x { ratio = y / (H x + eps); x <- x * H^T ratio / H^T 1 },
// x updated in place, `ratio` scratch.
extern "C" int lsr_rl_dense_f32_cpu(const float* y, float* x, float* ratio, int64_t Z, int64_t Y, int64_t X, const float* psf,
                                    const float* psf_flipped, int pz, int py, int px, const double* norm_table, int iters,
                                    float eps, lsr_stream_t stream) {
  return lsr_rl_dense_stats_f32_cpu(y, x, ratio, Z, Y, X, psf, psf_flipped, pz, py, px, norm_table, iters, eps, nullptr, stream);
}

extern "C" int lsr_rl_dense_stats_f32_cpu(const float* y, float* x, float* ratio, int64_t Z, int64_t Y, int64_t X,
                                          const float* psf, const float* psf_flipped, int pz, int py, int px,
                                          const double* norm_table, int iters, float eps, double* stats,
                                          lsr_stream_t stream) {
  LSR_REQUIRE_HOST_FMA();
  LSR_REQUIRE_PTR(y);
  LSR_REQUIRE_PTR(x);
  LSR_REQUIRE_PTR(ratio);
  LSR_REQUIRE(iters >= 0, LSR_E_ARG, "iters %d must be >= 0", iters);
  LSR_REQUIRE(ratio != x && ratio != y && x != y, LSR_E_ARG, "y, x and ratio must be distinct");
  if (stats != nullptr)
    for (int i = 0; i < lsr::kRlStats * iters; ++i) stats[i] = 0.0;
  for (int it = 0; it < iters; ++it) {
    int rc = lsr_correlate_dense_f32_cpu(x, ratio, y, Z, Y, X, psf_flipped, pz, py, px, LSR_EPI_RATIO, eps, nullptr, stream);
    if (rc) return rc;
    rc = lsr_correlate_dense_stats_f32_cpu(ratio, x, x, Z, Y, X, psf, pz, py, px, LSR_EPI_UPDATE, eps, norm_table,
                                           stats ? stats + lsr::kRlStats * it : nullptr, stream);
    if (rc) return rc;
  }
  return LSR_OK;
}

// ---- bright-field flat-field (shrimpy/preprocessing.py:385-404): pattern = volume.quantile(0.5, dim=0), its
// mean, out = in / pattern * mean.  The median as flatfield.hip forms it: the two middle order statistics a <= b,
// torch's lerp b - (b - a) * 0.5 for an even count, a for an odd one, NaN if the column holds one; the mean in f64.

namespace {

template <typename T>
int flat_pattern_cpu(const T* in, int64_t Z, int64_t Y, int64_t X, float* pattern, float* mean_out) {
  LSR_REQUIRE_PTR(in);
  LSR_REQUIRE_PTR(pattern);
  LSR_REQUIRE_PTR(mean_out);
  LSR_REQUIRE(Z > 0 && Y > 0 && X > 0, LSR_E_SHAPE, "shape (%lld,%lld,%lld) must be positive", (long long)Z, (long long)Y,
              (long long)X);
  LSR_REQUIRE_VOLUME(Z, Y, X);
  LSR_REQUIRE(Z < 65536, LSR_E_UNSUPPORTED, "Z = %lld: the median kernel counts in 16 bits", (long long)Z);
  const int64_t plane = Y * X;
  std::atomic<bool> failed{false};
  parallel_ranges(plane, [&](int64_t first, int64_t last) {
    std::vector<float> col(static_cast<size_t>(Z));
    for (int64_t i = first; i < last; ++i) {
      bool nan = false;
      for (int64_t z = 0; z < Z; ++z) {
        const float v = static_cast<float>(in[z * plane + i]);
        col[static_cast<size_t>(z)] = v;
        nan |= v != v;
      }
      if (nan) {
        pattern[i] = std::nanf("");
        continue;
      }
      const int64_t hi = Z / 2;
      std::nth_element(col.begin(), col.begin() + hi, col.end());
      const float b = col[static_cast<size_t>(hi)];
      if (Z & 1) {
        pattern[i] = b;
      } else {
        const float a = *std::max_element(col.begin(), col.begin() + hi);
        pattern[i] = b - (b - a) * 0.5f;
      }
    }
  }, failed);
  if (failed.load()) return lsr::fail(LSR_E_ARG, "out of host memory for the per-thread column buffers");
  double sum = 0.0;
  for (int64_t i = 0; i < plane; ++i) sum += static_cast<double>(pattern[i]);
  mean_out[0] = static_cast<float>(sum / static_cast<double>(plane));
  return LSR_OK;
}

template <typename T>
int flat_apply_cpu(const T* in, const float* pattern, const float* mean, float* out, int64_t Z, int64_t Y, int64_t X) {
  LSR_REQUIRE_PTR(in);
  LSR_REQUIRE_PTR(pattern);
  LSR_REQUIRE_PTR(mean);
  LSR_REQUIRE_PTR(out);
  LSR_REQUIRE(Z > 0 && Y > 0 && X > 0, LSR_E_SHAPE, "shape (%lld,%lld,%lld) must be positive", (long long)Z, (long long)Y,
              (long long)X);
  LSR_REQUIRE_VOLUME(Z, Y, X);
  const int64_t plane = Y * X;
  const float m = mean[0];
  parallel_ranges(Z, [&](int64_t z_first, int64_t z_last) {
    for (int64_t z = z_first; z < z_last; ++z)
      for (int64_t i = 0; i < plane; ++i) out[z * plane + i] = static_cast<float>(in[z * plane + i]) / pattern[i] * m;
  });
  return LSR_OK;
}

}  // namespace

extern "C" int lsr_flatfield_pattern_f32_cpu(const float* in, int64_t Z, int64_t Y, int64_t X, float* pattern, float* mean_out,
                                             void*, lsr_stream_t) {
  LSR_REQUIRE_HOST_FMA();
  return flat_pattern_cpu(in, Z, Y, X, pattern, mean_out);
}
extern "C" int lsr_flatfield_pattern_u16_cpu(const uint16_t* in, int64_t Z, int64_t Y, int64_t X, float* pattern,
                                             float* mean_out, void*, lsr_stream_t) {
  LSR_REQUIRE_HOST_FMA();
  return flat_pattern_cpu(in, Z, Y, X, pattern, mean_out);
}
extern "C" int lsr_flatfield_apply_f32_cpu(const float* in, const float* pattern, const float* mean_dev, float* out, int64_t Z,
                                           int64_t Y, int64_t X, lsr_stream_t) {
  LSR_REQUIRE_HOST_FMA();
  return flat_apply_cpu(in, pattern, mean_dev, out, Z, Y, X);
}
extern "C" int lsr_flatfield_apply_u16_cpu(const uint16_t* in, const float* pattern, const float* mean_dev, float* out,
                                           int64_t Z, int64_t Y, int64_t X, lsr_stream_t) {
  LSR_REQUIRE_HOST_FMA();
  return flat_apply_cpu(in, pattern, mean_dev, out, Z, Y, X);
}
