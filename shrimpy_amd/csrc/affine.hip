// General order-1 (trilinear) affine resample for gfx950.
//
// Registration apply of the north-star (label-free <-> fluorescence): the reference has no
// symbol for it (docs/data_structure.md:58-62); the CPU path it stands in for is
// scipy.ndimage.affine_transform(order=1, mode="constant"|"grid-constant").  Also the
// general-matrix fallback of deskew (any 3x4 map), with lsr_average_slices_f32.
//
// Memory-bound 8-tap gather: lanes run along the output-fastest axis (coalesced 256-B stores);
// for the near-identity maps of registration consecutive lanes read consecutive input
// addresses, so the 8 taps of a wave touch ~4 rows x 2 planes of 256-B segments that the
// vector L1 / XCD L2 serve after the first touch.  Algorithmic bytes: 4*N_src + 4*N_out.
//
// Arithmetic: coordinates, weights and the 8-corner sum in fp64, in scipy's operation order
// (see common.hpp), result rounded to f32 once -> bit-identical to the CPU oracle for finite
// inputs.  The fp64 cost (~60 DP ops / voxel) is comparable to the HBM time; see DESIGN.md.

#include "common.hpp"

namespace {

constexpr int kThreads = 256;
constexpr int kTileX = 64;
constexpr int kTileY = 4;

struct AffineArgs {
  const float* in;
  float* out;
  int64_t Zi, Yi, Xi;
  int64_t Zo, Yo, Xo;
  double m[12];
  float cval;
  int mode;
  int64_t tiles_x, tiles_y;
};

struct AxisTap {
  int64_t i0, i1;   // clamped neighbour indices
  double w0, w1;    // scipy weights: w0 = 1 - f, w1 = 1 - w0
  bool out0, out1;  // grid-constant: neighbour is outside the volume -> cval
};

// Returns false if (mode constant) the coordinate is outside [0, n-1] -> whole sample is cval.
__device__ __forceinline__ bool axis_tap(double c, int64_t n, int mode, AxisTap& t) {
  if (mode == LSR_MODE_CONSTANT && (c < 0.0 || c > static_cast<double>(n - 1))) return false;
  const double fl = floor(c);
  const double f = c - fl;
  t.w0 = 1.0 - f;
  t.w1 = 1.0 - t.w0;
  // indices only matter while a neighbour can be inside; clamp far-away coordinates first
  const double lim = static_cast<double>(n) + 1.0;
  const int64_t start = static_cast<int64_t>(fmin(fmax(fl, -2.0), lim));
  t.out0 = (start < 0) || (start >= n);
  t.out1 = (start + 1 < 0) || (start + 1 >= n);
  t.i0 = min(max(start, int64_t(0)), n - 1);
  t.i1 = min(max(start + 1, int64_t(0)), n - 1);
  return true;
}

__global__ __launch_bounds__(kThreads) void affine_kernel(AffineArgs p) {
  int64_t bid = blockIdx.x;
  const int64_t tx = bid % p.tiles_x;
  bid /= p.tiles_x;
  const int64_t ty = bid % p.tiles_y;
  const int64_t zo = bid / p.tiles_y;
  const int64_t xo = tx * kTileX + (threadIdx.x & 63);
  const int64_t yo = ty * kTileY + (threadIdx.x >> 6);
  if (xo >= p.Xo || yo >= p.Yo) return;

  const double zd = static_cast<double>(zo), yd = static_cast<double>(yo),
               xd = static_cast<double>(xo);
  const double cz = lsr::affine_coord(zd, yd, xd, p.m[0], p.m[1], p.m[2], p.m[3]);
  const double cy = lsr::affine_coord(zd, yd, xd, p.m[4], p.m[5], p.m[6], p.m[7]);
  const double cx = lsr::affine_coord(zd, yd, xd, p.m[8], p.m[9], p.m[10], p.m[11]);

  AxisTap tz, ty_, tx_;
  float result = p.cval;
  if (axis_tap(cz, p.Zi, p.mode, tz) && axis_tap(cy, p.Yi, p.mode, ty_) &&
      axis_tap(cx, p.Xi, p.mode, tx_)) {
    const bool grid = p.mode == LSR_MODE_GRID_CONSTANT;
    const double cv = static_cast<double>(p.cval);
    const int64_t sz = p.Yi * p.Xi;
    double t = 0.0;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int64_t iz = a ? tz.i1 : tz.i0;
      const double wz = a ? tz.w1 : tz.w0;
      const bool oz = a ? tz.out1 : tz.out0;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int64_t iy = b ? ty_.i1 : ty_.i0;
        const double wy = b ? ty_.w1 : ty_.w0;
        const bool oy = b ? ty_.out1 : ty_.out0;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int64_t ix = c ? tx_.i1 : tx_.i0;
          const double wx = c ? tx_.w1 : tx_.w0;
          const bool ox = c ? tx_.out1 : tx_.out0;
          double coeff;
          if (grid && (oz || oy || ox)) {
            coeff = cv;
          } else {
            coeff = static_cast<double>(p.in[iz * sz + iy * p.Xi + ix]);
          }
          coeff = lsr::dmul(coeff, wz);
          coeff = lsr::dmul(coeff, wy);
          coeff = lsr::dmul(coeff, wx);
          t = lsr::dadd(t, coeff);
        }
      }
    }
    result = static_cast<float>(t);
  }
  p.out[(zo * p.Yo + yo) * p.Xo + xo] = result;
}

}  // namespace

extern "C" int lsr_affine_f32(const float* in, int64_t Zi, int64_t Yi, int64_t Xi, float* out,
                              int64_t Zo, int64_t Yo, int64_t Xo, const double M[12], float cval,
                              int mode, lsr_stream_t stream) {
  LSR_REQUIRE_PTR(in);
  LSR_REQUIRE_PTR(out);
  LSR_REQUIRE_PTR(M);
  LSR_REQUIRE(Zi > 0 && Yi > 0 && Xi > 0, LSR_E_SHAPE,
              "input shape (%lld,%lld,%lld) must be positive", (long long)Zi, (long long)Yi,
              (long long)Xi);
  LSR_REQUIRE(Zo > 0 && Yo > 0 && Xo > 0, LSR_E_SHAPE,
              "output shape (%lld,%lld,%lld) must be positive", (long long)Zo, (long long)Yo,
              (long long)Xo);
  LSR_REQUIRE(mode == LSR_MODE_CONSTANT || mode == LSR_MODE_GRID_CONSTANT, LSR_E_ARG,
              "unknown border mode %d", mode);
  for (int i = 0; i < 12; ++i)
    LSR_REQUIRE(M[i] == M[i] && M[i] - M[i] == 0.0, LSR_E_ARG, "M[%d] is not finite", i);

  AffineArgs p;
  p.in = in;
  p.out = out;
  p.Zi = Zi; p.Yi = Yi; p.Xi = Xi;
  p.Zo = Zo; p.Yo = Yo; p.Xo = Xo;
  for (int i = 0; i < 12; ++i) p.m[i] = M[i];
  p.cval = cval;
  p.mode = mode;
  p.tiles_x = lsr::ceil_div(Xo, kTileX);
  p.tiles_y = lsr::ceil_div(Yo, kTileY);
  const int64_t blocks = p.tiles_x * p.tiles_y * Zo;
  LSR_REQUIRE(blocks < (int64_t(1) << 31), LSR_E_SHAPE, "grid of %lld workgroups is too large",
              (long long)blocks);
  hipLaunchKernelGGL(affine_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0,
                     lsr::as_stream(stream), p);
  return lsr::launch_status("lsr_affine_f32");
}
