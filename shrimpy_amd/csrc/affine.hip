// General order-1 (trilinear) affine resample for gfx950.
//
// Registration apply of the north-star (label-free <-> fluorescence): the reference has no
// symbol for it (docs/data_structure.md:58-62); the CPU path it stands in for is
// scipy.ndimage.affine_transform(order=1, mode="constant"|"grid-constant").  Also the
// general-matrix fallback of deskew (any 3x4 map), with lsr_average_slices_f32.
//
// 8-tap gather: lanes run along the output-fastest axis (coalesced 256-B stores); for the
// near-identity maps of registration consecutive lanes read consecutive input addresses, so the 8
// taps of a wave touch ~4 rows x 2 planes of 256-B segments that the vector L1 / XCD L2 serve after
// the first touch.  Algorithmic bytes: 4*N_src + 4*N_out.
//
// Arithmetic: coordinates, weights and the 8-corner sum in fp64, in scipy's operation order
// (see common.hpp), result rounded to f32 once -> bit-identical to the CPU oracle for finite
// inputs.  That makes the kernel fp64-VALU-bound (~60 DP ops per voxel ~ the HBM time at the
// 78 TFLOP/s fp64 vector rate), so everything else is kept cheap: a thread produces 4 voxels of
// one output row (the (zo, yo) part of each coordinate is computed once per row -- scipy's sum
// order ((zo*m0 + yo*m1) + xo*m2) + shift makes that prefix exact to hoist), indices are 32-bit.

#include "common.hpp"

namespace {

constexpr int kThreads = 256;
constexpr int kLanesX = 64;
constexpr int kRows = kThreads / kLanesX;  // 4 output rows per workgroup
constexpr int kPerThread = 4;              // voxels per thread along x, stride 64
constexpr int kTileX = kLanesX * kPerThread;

struct AffineArgs {
  const float* in;
  float* out;
  int Zi, Yi, Xi;
  unsigned pitch, plane;   // source strides in floats (dense: Xi, Yi * Xi)
  int64_t opitch, oplane;  // output strides in floats (dense: Xo, Yo * Xo)
  int Zo, Yo, Xo;
  double m[12];
  float cval;
  int mode;
  int tiles_x, tiles_y;
};

struct AxisTap {
  int i0, i1;       // clamped neighbour indices
  double w0, w1;    // scipy weights: w0 = 1 - f, w1 = 1 - w0
  double f;         // fractional part
  bool out0, out1;  // grid-constant: neighbour is outside the volume -> cval
};

// Returns false if (mode constant) the coordinate is outside [0, n-1] -> whole sample is cval.
template <bool GRID>
__device__ __forceinline__ bool axis_tap(double c, int n, AxisTap& t) {
  if (!GRID && (c < 0.0 || c > static_cast<double>(n - 1))) return false;
  const double fl = floor(c);
  const double f = c - fl;
  t.f = f;
  t.w0 = 1.0 - f;
  t.w1 = 1.0 - t.w0;
  if constexpr (!GRID) {
    // 0 <= c <= n-1: floor(c) is a valid index; only the upper neighbour can leave the volume
    // (c == n-1 exactly, where its weight is 0)
    t.i0 = static_cast<int>(fl);
    t.i1 = min(t.i0 + 1, n - 1);
    t.out0 = t.out1 = false;
  } else {
    // indices only matter while a neighbour can be inside; clamp far-away coordinates first
    const int start = static_cast<int>(fmin(fmax(fl, -2.0), static_cast<double>(n) + 1.0));
    t.out0 = start < 0 || start >= n;
    t.out1 = start + 1 < 0 || start + 1 >= n;
    t.i0 = min(max(start, 0), n - 1);
    t.i1 = min(max(start + 1, 0), n - 1);
  }
  return true;
}

typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));  // 8-byte load, 4-byte aligned

// mode "constant": both x neighbours of a row in ONE 8-byte load (half the gather instructions;
// the texture-address unit, not HBM or the ALUs, bounds this kernel).  When the upper neighbour
// is the clamped one (coordinate exactly on the last column, weight 0) the pair is read one
// element lower and both values come from its upper half.
__device__ __forceinline__ void load_x_pair(const float* row, int i0, int n, float& v0, float& v1) {
  const int base = min(i0, n - 2);
  const f32x2u pr = *reinterpret_cast<const f32x2u*>(row + base);
  v0 = base == i0 ? pr.x : pr.y;
  v1 = pr.y;
}

template <bool GRID, bool F32>
__global__ __launch_bounds__(kThreads) void affine_kernel(AffineArgs p) {
  int bid = blockIdx.x;
  const int tx = bid % p.tiles_x;
  bid /= p.tiles_x;
  const int ty = bid % p.tiles_y;
  const int zo = bid / p.tiles_y;
  const int yo = ty * kRows + static_cast<int>(threadIdx.x >> 6);
  if (yo >= p.Yo) return;
  const int x_first = tx * kTileX + static_cast<int>(threadIdx.x & 63);

  // row part of each coordinate: (zo*m0 + yo*m1), scipy's first two terms
  const double zd = static_cast<double>(zo), yd = static_cast<double>(yo);
  const double rz = lsr::dadd(lsr::dmul(zd, p.m[0]), lsr::dmul(yd, p.m[1]));
  const double ry = lsr::dadd(lsr::dmul(zd, p.m[4]), lsr::dmul(yd, p.m[5]));
  const double rx = lsr::dadd(lsr::dmul(zd, p.m[8]), lsr::dmul(yd, p.m[9]));
  // element indices fit 32 bits unsigned (host check); one 64-bit add per load
  const unsigned sz = p.plane;
  const double cv = static_cast<double>(p.cval);
  float* orow = p.out + static_cast<int64_t>(zo) * p.oplane + static_cast<int64_t>(yo) * p.opitch;

#pragma unroll
  for (int k = 0; k < kPerThread; ++k) {
    const int xo = x_first + k * kLanesX;
    if (xo >= p.Xo) break;
    const double xd = static_cast<double>(xo);
    const double cz = lsr::dadd(lsr::dadd(rz, lsr::dmul(xd, p.m[2])), p.m[3]);
    const double cy = lsr::dadd(lsr::dadd(ry, lsr::dmul(xd, p.m[6])), p.m[7]);
    const double cx = lsr::dadd(lsr::dadd(rx, lsr::dmul(xd, p.m[10])), p.m[11]);

    AxisTap tz, ty_, tx_;
    float result = p.cval;
    if (axis_tap<GRID>(cz, p.Zi, tz) && axis_tap<GRID>(cy, p.Yi, ty_) &&
        axis_tap<GRID>(cx, p.Xi, tx_)) {
      if constexpr (F32) {
        // LSR_MODE_F32_INTERP: fp64 coordinates (border decisions unchanged), f32 weights and
        // FMAs -- not bit-identical to scipy (~1e-6 relative), HBM-bound instead of fp64-bound
        const float wz1 = static_cast<float>(tz.f), wy1 = static_cast<float>(ty_.f),
                    wx1 = static_cast<float>(tx_.f);  // (w1 = 1 - (1 - f) differs from f by <= 1 ulp of fp64)
        float v[2][2][2];
        if constexpr (!GRID) {
          if (p.Xi >= 2) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
              for (int b = 0; b < 2; ++b) {
                const unsigned o = static_cast<unsigned>(a ? tz.i1 : tz.i0) * sz +
                                   static_cast<unsigned>(b ? ty_.i1 : ty_.i0) * p.pitch;
                load_x_pair(p.in + o, tx_.i0, p.Xi, v[a][b][0], v[a][b][1]);
              }
          } else {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
              for (int b = 0; b < 2; ++b)
                v[a][b][0] = v[a][b][1] = p.in[static_cast<unsigned>(a ? tz.i1 : tz.i0) * sz +
                                               static_cast<unsigned>(b ? ty_.i1 : ty_.i0) * p.pitch];
          }
        } else
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              const unsigned o = static_cast<unsigned>(a ? tz.i1 : tz.i0) * sz +
                                 static_cast<unsigned>(b ? ty_.i1 : ty_.i0) * p.pitch +
                                 static_cast<unsigned>(c ? tx_.i1 : tx_.i0);
              float val = p.in[o];
              if (GRID && ((a ? tz.out1 : tz.out0) || (b ? ty_.out1 : ty_.out0) || (c ? tx_.out1 : tx_.out0)))
                val = p.cval;
              v[a][b][c] = val;
            }
        float r2[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) r2[a][b] = fmaf(wx1, v[a][b][1] - v[a][b][0], v[a][b][0]);
        const float r10 = fmaf(wy1, r2[0][1] - r2[0][0], r2[0][0]);
        const float r11 = fmaf(wy1, r2[1][1] - r2[1][0], r2[1][0]);
        result = fmaf(wz1, r11 - r10, r10);
      } else {
      double t = 0.0;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const unsigned oz = static_cast<unsigned>(a ? tz.i1 : tz.i0) * sz;
        const double wz = a ? tz.w1 : tz.w0;
        const bool bz = a ? tz.out1 : tz.out0;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const unsigned oy = oz + static_cast<unsigned>(b ? ty_.i1 : ty_.i0) * p.pitch;
          const double wy = b ? ty_.w1 : ty_.w0;
          const bool by = b ? ty_.out1 : ty_.out0;
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const double wx = c ? tx_.w1 : tx_.w0;
            const bool bx = c ? tx_.out1 : tx_.out0;
            double coeff = static_cast<double>(p.in[oy + static_cast<unsigned>(c ? tx_.i1 : tx_.i0)]);
            if (GRID && (bz || by || bx)) coeff = cv;
            coeff = lsr::dmul(coeff, wz);
            coeff = lsr::dmul(coeff, wy);
            coeff = lsr::dmul(coeff, wx);
            t = lsr::dadd(t, coeff);
          }
        }
      }
      result = static_cast<float>(t);
      }
    }
    orow[xo] = result;
  }
}

}  // namespace

namespace lsr {
// affine_planar.hip: z-decoupled maps, either border rule; false = not applicable
// (pitch / plane: source strides in floats; dense = Xi, Yi * Xi)
bool launch_affine_planar(const float* in, int64_t Zi, int64_t Yi, int64_t Xi, int64_t pitch, int64_t plane,
                          float* out, int64_t Zo, int64_t Yo, int64_t Xo, int64_t opitch, int64_t oplane,
                          const double M[12], float cval, bool f32, bool grid, hipStream_t s);
bool affine_planar_geometry(int64_t Yi, int64_t Xi, int64_t pitch, const double M[12], int* box_y, int* box_x,
                            int* slots, int64_t* lds_bytes, int* tile = nullptr, int waves = 8);
// affine_box.hip: any map whose per-block source box fits in LDS (z-coupled maps included),
// either border rule; false = not applicable
bool launch_affine_box(const float* in, int64_t Zi, int64_t Yi, int64_t Xi, int64_t pitch, int64_t plane, float* out,
                       int64_t Zo, int64_t Yo, int64_t Xo, int64_t opitch, int64_t oplane, const double M[12], float cval,
                       bool f32, bool grid, hipStream_t s);
bool affine_box_geometry(int64_t Zi, int64_t Yi, int64_t Xi, int64_t pitch, int64_t plane, const double M[12],
                         int* box_z, int* box_y, int* box_x, int64_t* lds_bytes);
bool affine_box_shape(int64_t Zi, int64_t Yi, int64_t Xi, const double M[12], int out6[6]);
}  // namespace lsr

extern "C" int lsr_affine_kernel_choice(int64_t Yi, int64_t Xi, const double M[12], int mode) {
  if (M == nullptr) return 0;
  int by, bx, sl;
  int64_t lds;
  const int border = mode & ~LSR_MODE_F32_INTERP;
  return (border == LSR_MODE_CONSTANT || border == LSR_MODE_GRID_CONSTANT) &&
                 lsr::affine_planar_geometry(Yi, Xi, Xi, M, &by, &bx, &sl, &lds)
             ? 1
             : 0;
}

extern "C" int lsr_affine_path(int64_t Zi, int64_t Yi, int64_t Xi, const double M[12], int mode) {
  if (M == nullptr) return 0;
  const int border = mode & ~LSR_MODE_F32_INTERP;
  if (border != LSR_MODE_CONSTANT && border != LSR_MODE_GRID_CONSTANT) return 0;
  int a, b, c;
  int64_t lds;
  if (!lsr::volume_in_range(Zi, Yi, Xi)) return 0;
  if (lsr::affine_planar_geometry(Yi, Xi, Xi, M, &a, &b, &c, &lds)) return 1;
  if (lsr::affine_box_geometry(Zi, Yi, Xi, Xi, Yi * Xi, M, &a, &b, &c, &lds)) return 2;
  return 0;
}

// the same question for a source with padded rows (lsr_affine_pitched_f32)
extern "C" int lsr_affine_path_pitched(int64_t Zi, int64_t Yi, int64_t Xi, int64_t in_pitch, int64_t in_plane,
                                       const double M[12], int mode) {
  if (M == nullptr) return 0;
  const int border = mode & ~LSR_MODE_F32_INTERP;
  if (border != LSR_MODE_CONSTANT && border != LSR_MODE_GRID_CONSTANT) return 0;
  int a, b, c;
  int64_t lds;
  if (!lsr::volume_in_range(Zi, Yi, Xi) || !lsr::strides_in_range(in_pitch, in_plane)) return 0;
  if (in_plane % 4 == 0 && lsr::affine_planar_geometry(Yi, Xi, in_pitch, M, &a, &b, &c, &lds)) return 1;
  if (lsr::affine_box_geometry(Zi, Yi, Xi, in_pitch, in_plane, M, &a, &b, &c, &lds)) return 2;
  return 0;
}

extern "C" int lsr_affine_box_shape(int64_t Zi, int64_t Yi, int64_t Xi, const double M[12], int* out6) {
  if (M == nullptr || out6 == nullptr) return 0;
  return lsr::affine_box_shape(Zi, Yi, Xi, M, out6) ? 1 : 0;
}

namespace {
int affine_impl(const char* what, const float* in, int64_t Zi, int64_t Yi, int64_t Xi, int64_t pitch, int64_t plane,
                float* out, int64_t Zo, int64_t Yo, int64_t Xo, int64_t opitch, int64_t oplane, const double M[12],
                float cval, int mode, lsr_stream_t stream);
}

extern "C" int lsr_affine_f32(const float* in, int64_t Zi, int64_t Yi, int64_t Xi, float* out,
                              int64_t Zo, int64_t Yo, int64_t Xo, const double M[12], float cval,
                              int mode, lsr_stream_t stream) {
  LSR_REQUIRE(Zi > 0 && Yi > 0 && Xi > 0 && Zo > 0 && Yo > 0 && Xo > 0, LSR_E_SHAPE,
              "shapes (%lld,%lld,%lld) -> (%lld,%lld,%lld) must be positive", (long long)Zi, (long long)Yi, (long long)Xi,
              (long long)Zo, (long long)Yo, (long long)Xo);
  LSR_REQUIRE_VOLUME(Zi, Yi, Xi);
  LSR_REQUIRE_VOLUME(Zo, Yo, Xo);
  return affine_impl("lsr_affine_f32", in, Zi, Yi, Xi, Xi, Yi * Xi, out, Zo, Yo, Xo, Xo, Yo * Xo, M, cval, mode, stream);
}

extern "C" int lsr_affine_pitched_f32(const float* in, int64_t Zi, int64_t Yi, int64_t Xi, int64_t in_pitch,
                                      int64_t in_plane, float* out, int64_t Zo, int64_t Yo, int64_t Xo,
                                      int64_t out_pitch, int64_t out_plane, const double M[12], float cval, int mode,
                                      lsr_stream_t stream) {
  LSR_REQUIRE_VOLUME(Zi, Yi, Xi);
  LSR_REQUIRE_VOLUME(Zo, Yo, Xo);
  LSR_REQUIRE_STRIDES(in_pitch, in_plane);
  LSR_REQUIRE_STRIDES(out_pitch, out_plane);
  LSR_REQUIRE(in_pitch >= Xi && in_plane >= Yi * in_pitch, LSR_E_SHAPE,
              "source strides (%lld, %lld) are smaller than a (%lld x %lld) plane", (long long)in_pitch,
              (long long)in_plane, (long long)Yi, (long long)Xi);
  LSR_REQUIRE(out_pitch >= Xo && out_plane >= Yo * out_pitch, LSR_E_SHAPE,
              "output strides (%lld, %lld) are smaller than a (%lld x %lld) plane", (long long)out_pitch,
              (long long)out_plane, (long long)Yo, (long long)Xo);
  return affine_impl("lsr_affine_pitched_f32", in, Zi, Yi, Xi, in_pitch, in_plane, out, Zo, Yo, Xo, out_pitch, out_plane,
                     M, cval, mode, stream);
}

namespace {
int affine_impl(const char* what, const float* in, int64_t Zi, int64_t Yi, int64_t Xi, int64_t pitch, int64_t plane,
                float* out, int64_t Zo, int64_t Yo, int64_t Xo, int64_t opitch, int64_t oplane, const double M[12],
                float cval, int mode, lsr_stream_t stream) {
  LSR_REQUIRE_PTR(in);
  LSR_REQUIRE_PTR(out);
  LSR_REQUIRE_PTR(M);
  LSR_REQUIRE(Zi > 0 && Yi > 0 && Xi > 0, LSR_E_SHAPE,
              "input shape (%lld,%lld,%lld) must be positive", (long long)Zi, (long long)Yi,
              (long long)Xi);
  LSR_REQUIRE_VOLUME(Zi, Yi, Xi);
  LSR_REQUIRE(Zo > 0 && Yo > 0 && Xo > 0, LSR_E_SHAPE,
              "output shape (%lld,%lld,%lld) must be positive", (long long)Zo, (long long)Yo,
              (long long)Xo);
  LSR_REQUIRE_VOLUME(Zo, Yo, Xo);
  const int64_t lim = int64_t(1) << 30;
  LSR_REQUIRE(Zi < lim && Yi < lim && Xi < lim && Zo < lim && Yo < lim && Xo < lim,
              LSR_E_UNSUPPORTED, "a dimension exceeds 2^30");
  LSR_REQUIRE(pitch < lim, LSR_E_UNSUPPORTED, "a source row of %lld floats exceeds 2^30", (long long)pitch);
  const bool f32 = (mode & LSR_MODE_F32_INTERP) != 0;
  mode &= ~LSR_MODE_F32_INTERP;
  LSR_REQUIRE(mode == LSR_MODE_CONSTANT || mode == LSR_MODE_GRID_CONSTANT, LSR_E_ARG,
              "unknown border mode %d", mode);
  for (int i = 0; i < 12; ++i)
    LSR_REQUIRE(M[i] == M[i] && M[i] - M[i] == 0.0, LSR_E_ARG, "M[%d] is not finite", i);

  // the LDS-staged kernels (either border rule): z-decoupled maps, then any map whose source box fits LDS
  if (in != out &&
      (lsr::launch_affine_planar(in, Zi, Yi, Xi, pitch, plane, out, Zo, Yo, Xo, opitch, oplane, M, cval, f32,
                                 mode == LSR_MODE_GRID_CONSTANT, lsr::as_stream(stream)) ||
       lsr::launch_affine_box(in, Zi, Yi, Xi, pitch, plane, out, Zo, Yo, Xo, opitch, oplane, M, cval, f32,
                              mode == LSR_MODE_GRID_CONSTANT, lsr::as_stream(stream))))
    return lsr::launch_status(what);

  // (the LDS-staged kernels address a volume plane by plane with 64-bit bases; the gather kernel below does not)
  LSR_REQUIRE(Zi * plane <= (int64_t(1) << 32), LSR_E_UNSUPPORTED,
              "the moving volume spans more than 2^32 elements and this map does not fit the LDS-staged kernels "
              "(the gather kernel indexes elements in 32 bits): resample it in z slabs");
  AffineArgs p;
  p.in = in;
  p.out = out;
  p.Zi = static_cast<int>(Zi); p.Yi = static_cast<int>(Yi); p.Xi = static_cast<int>(Xi);
  p.pitch = static_cast<unsigned>(pitch); p.plane = static_cast<unsigned>(plane);
  p.opitch = opitch; p.oplane = oplane;
  p.Zo = static_cast<int>(Zo); p.Yo = static_cast<int>(Yo); p.Xo = static_cast<int>(Xo);
  for (int i = 0; i < 12; ++i) p.m[i] = M[i];
  p.cval = cval;
  p.mode = mode;
  p.tiles_x = static_cast<int>(lsr::ceil_div(Xo, kTileX));
  p.tiles_y = static_cast<int>(lsr::ceil_div(Yo, kRows));
  const int64_t blocks = int64_t(p.tiles_x) * p.tiles_y * Zo;
  LSR_REQUIRE(blocks < (int64_t(1) << 31), LSR_E_SHAPE, "grid of %lld workgroups is too large",
              (long long)blocks);
  const dim3 grid(static_cast<unsigned>(blocks)), block(kThreads);
  hipStream_t s = lsr::as_stream(stream);
  if (mode == LSR_MODE_GRID_CONSTANT) {
    if (f32) hipLaunchKernelGGL((affine_kernel<true, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((affine_kernel<true, false>), grid, block, 0, s, p);
  } else {
    if (f32) hipLaunchKernelGGL((affine_kernel<false, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((affine_kernel<false, false>), grid, block, 0, s, p);
  }
  return lsr::launch_status(what);
}
}  // namespace
