// Oblique-plane deskew for gfx950, slice averaging fused in.
//
// Replaces biahub.deskew.fast_deskew_zyx as called at shrimpy/preprocessing.py:408-413
// (reference root /root/reference). Geometry (matrix rows, output shape) is computed on the
// host (shrimpy_amd/geometry.py); this file only evaluates it.
//
// The deskew map interpolates along ONE input axis only:
//     z_in = a*zd + b*xo + c      (fractional; zd = pre-average output plane, xo = scan axis)
//     y_in = sy*zd + oy           (integer: one tilt row per output plane)
//     x_in = sx*yo + ox           (integer: raw X <-> output Y', 1:1)
// so an output plane (Y', X') is a transposed, 1-D-resampled copy of the input slice
// in[:, y_in, :].  The output-fastest axis X' walks the input's SLOWEST axis, hence a
// workgroup stages an input slab (z-range x 64 contiguous x) in LDS with coalesced 256-B row
// reads, and emits 256-B coalesced stores along X' -- a resampling transpose through LDS.
//
// HBM traffic per launch: 4*N_in (each raw voxel read ~once; neighbouring X' tiles share
// 1-2 slab rows) + 4*N_out.  fp64 work is ~10 DP ops per pre-average voxel, far below the
// HBM time, so the arithmetic follows scipy.ndimage exactly (see common.hpp).

#include "common.hpp"

#include <cstdlib>
#include <type_traits>

namespace {

constexpr int kTileY = 64;    // output Y' per workgroup == contiguous raw-x run (256 B rows)
constexpr int kTileX = 64;    // output X' per workgroup (one wave-width of coalesced stores)
constexpr int kSlabRows = 68; // LDS rows (input z) a workgroup can stage
constexpr int kPitch = kTileY + 1;  // +1 float: lanes walk z at ~b rows/lane -> distinct banks
constexpr int kThreads = 256;
constexpr int kRowsPerThread = kTileY * kTileX / kThreads;  // 16 output points per thread
constexpr int kMaxAvg = 4096;  // sanity bound only

struct DeskewArgs {
  const void* in;           // raw stack: float32, or (U16) uint16 camera counts
  float* out;
  int64_t Z, Y, X;          // raw
  int64_t Zo, Yo, Xo, Zd;   // output, and pre-average depth
  int64_t out_pitch, out_plane;  // output strides in floats (dense: Xo, Yo*Xo)
  double a, b, c;           // z_in = a*zd + b*xo + c
  int64_t oy, ox;
  int sy, sx;
  int avg_n;
  int grid;                 // border rule: 0 = scipy "constant", 1 = "grid-constant" (blend towards 0 across the z border)
  int tile_x;               // X' handled per workgroup (<= kTileX, chosen so the slab fits)
  int64_t tiles_x, tiles_y; // workgroup grid, flattened: x fastest, then y, then zo
  int64_t blocks;           // tiles_x * tiles_y * Zo
  int xcd_swizzle;          // 1: XCD-contiguous tile order (the grid is padded to a multiple of 8)
  const float* flat_pattern;  // FLAT: (Y, X) per-pixel median over Z (flatfield.hip)
  const float* flat_mean;     // FLAT: its mean, a device scalar
  const float* cval;          // the value outside the stack: a device scalar (e.g. the stack's minimum, lsr_minmax_*); NULL = 0
};

// FLAT fuses the bright-field flat-field correction into the staging pass: every raw sample
// becomes in / pattern[y][x] * mean (same operations, same order as the separate apply kernel, so
// the result is bit-identical to flat-field followed by deskew) and the corrected volume is never
// written to HBM.
// U16: the raw stack is the camera's uint16 counts; they become float32 in the staging pass (exact),
// so the 8.6 GB float copy of the stack never exists -- half the HBM read, half the PCIe upload.
template <bool FLAT, bool U16 = false>
__global__ __launch_bounds__(kThreads) void deskew_kernel(DeskewArgs p) {
  using raw_t = std::conditional_t<U16, unsigned short, float>;
  __shared__ float slab[kSlabRows * kPitch];
  __shared__ float zero_row[kPitch];   // a row of the fill value (zero unless the caller names another: scipy's cval)
  const float cv = p.cval ? p.cval[0] : 0.0f;
  if (threadIdx.x < kPitch) zero_row[threadIdx.x] = cv;   // visible after the first barrier below

  const int tid = threadIdx.x;
  // Workgroups b, b + 8, ... run on one XCD: each XCD takes a contiguous run of the tile order (x fastest), so that
  // neighbouring X' tiles -- whose 256-byte store runs share a cache line wherever the output row pitch is not a
  // multiple of 32 floats (a dense (171, 2048, 2270) volume: every row) -- complete that line in ONE L2 instead of
  // evicting two partial lines to HBM, and the slab rows they share are fetched once.  p.xcd_swizzle = 0: as numbered.
  int64_t bid = blockIdx.x;
  if (p.xcd_swizzle) {
    const int64_t per = (p.blocks + 7) >> 3;
    bid = (bid & 7) * per + (bid >> 3);
    if (bid >= p.blocks) return;
  }
  const int64_t tx = bid % p.tiles_x;
  bid /= p.tiles_x;
  const int64_t ty = bid % p.tiles_y;
  const int64_t zo = bid / p.tiles_y;

  const int64_t xo0 = tx * p.tile_x;
  const int64_t yo0 = ty * kTileY;
  const int n_xo = static_cast<int>(min(static_cast<int64_t>(p.tile_x), p.Xo - xo0));
  const int n_yo = static_cast<int>(min(static_cast<int64_t>(kTileY), p.Yo - yo0));

  // compute-phase mapping: lane -> X' (coalesced stores), 4 waves stride over Y'
  const int li = tid & 63;
  const int lj0 = tid >> 6;
  const double xo_d = static_cast<double>(xo0 + li);

  float acc[kRowsPerThread];
#pragma unroll
  for (int m = 0; m < kRowsPerThread; ++m) acc[m] = 0.0f;

  for (int k = 0; k < p.avg_n; ++k) {
    const int64_t zd = min(zo * p.avg_n + k, p.Zd - 1);  // edge padding of the remainder
    const int64_t y_in = p.sy * zd + p.oy;
    const bool row_ok = (y_in >= 0) && (y_in < p.Y);

    // z_in at both ends of the tile (monotonic in xo): the slab's z-range.
    const double zd_d = static_cast<double>(zd);
    const double z_first =
        lsr::affine_coord(zd_d, 0.0, static_cast<double>(xo0), p.a, 0.0, p.b, p.c);
    const double z_last =
        lsr::affine_coord(zd_d, 0.0, static_cast<double>(xo0 + n_xo - 1), p.a, 0.0, p.b, p.c);
    const double z_lo_d = floor(fmin(z_first, z_last));
    const double z_hi_d = floor(fmax(z_first, z_last)) + 1.0;
    // clamp to the volume (as doubles first: they may be far outside int range)
    const double zmax_d = static_cast<double>(p.Z - 1);
    const bool any_z = (z_hi_d >= 0.0) && (z_lo_d <= zmax_d);
    const int64_t z_lo = static_cast<int64_t>(fmin(fmax(z_lo_d, 0.0), zmax_d));
    const int64_t z_hi = static_cast<int64_t>(fmin(fmax(z_hi_d, 0.0), zmax_d));
    const int n_rows = any_z ? static_cast<int>(z_hi - z_lo + 1) : 0;  // <= kSlabRows by host

    if (k > 0) __syncthreads();  // previous slab fully consumed
    if (row_ok && n_rows > 0) {
      // stage: slab[zl][j] = in[z_lo+zl][y_in][sx*(yo0+j)+ox]; lanes along raw x (coalesced)
      const int col = tid & 63;          // position inside the contiguous 64-float x run
      const int j = (p.sx > 0) ? col : (kTileY - 1 - col);
      const int64_t x_in = p.sx * (yo0 + j) + p.ox;
      const bool col_ok = (j < n_yo) && (x_in >= 0) && (x_in < p.X);
      // Loads are unconditional (lanes past the tile read a clamped, valid address and drop the
      // value) and issued in batches before their first use: under a per-lane condition hipcc
      // waits for every load before issuing the next, which left this kernel latency-bound
      // (81 % of wave cycles waiting at 3.5 TB/s).
      const int64_t x_safe = min(max(x_in, static_cast<int64_t>(0)), p.X - 1);
      const raw_t* src = static_cast<const raw_t*>(p.in) + (z_lo * p.Y + y_in) * p.X + x_safe;
      const int64_t z_stride = p.Y * p.X;
      float pat = 1.0f, mean = 1.0f;
      if constexpr (FLAT) {
        pat = p.flat_pattern[y_in * p.X + x_safe];
        mean = p.flat_mean[0];
      }
      constexpr int kStep = kThreads / 64;   // slab rows per pass over the workgroup
      constexpr int kBatch = 6;              // loads in flight per thread
      for (int zl0 = tid >> 6; zl0 < n_rows; zl0 += kStep * kBatch) {
        float v[kBatch];
#pragma unroll
        for (int i = 0; i < kBatch; ++i) v[i] = static_cast<float>(src[min(zl0 + i * kStep, n_rows - 1) * z_stride]);
#pragma unroll
        for (int i = 0; i < kBatch; ++i) {
          const int zl = zl0 + i * kStep;
          if (zl < n_rows) {
            float w = v[i];
            if constexpr (FLAT) w = w / pat * mean;
            slab[zl * kPitch + j] = col_ok ? w : cv;
          }
        }
      }
    }
    __syncthreads();

    // sample: scipy order-1 along z only (the y/x weights are exactly 1 and 0)
    const double z_in = lsr::affine_coord(zd_d, 0.0, xo_d, p.a, 0.0, p.b, p.c);
    // "constant": a coordinate outside [0, Z-1] gives 0.  "grid-constant": the volume continues as
    // zeros, so a coordinate in (-1, 0) or (Z-1, Z) still blends its one inside neighbour with 0
    // (scipy sums cval * weight for the outside tap: +0 here).  y_in / x_in are integers: a row or
    // column outside the volume is 0 under both rules.
    const bool ok = row_ok && (li < n_xo) &&
                    (p.grid ? (z_in > -1.0 && z_in < zmax_d + 1.0) : (!(z_in < 0.0) && !(z_in > zmax_d)));
    if (ok) {
      const double zf = floor(z_in);
      const double f = z_in - zf;
      const double w0 = 1.0 - f;
      const double w1 = 1.0 - w0;
      const int64_t z0 = static_cast<int64_t>(zf);              // -1 .. Z-1
      const int r0 = static_cast<int>(max(z0, static_cast<int64_t>(0)) - z_lo);
      const int r1 = static_cast<int>(min(z0 + 1, p.Z - 1) - z_lo);
      // a neighbour past the end of the scan ("grid-constant" only) reads the row of zeros
      const float* s0 = z0 < 0 ? zero_row : slab + r0 * kPitch;
      const float* s1 = (p.grid && z0 + 1 > p.Z - 1) ? zero_row : slab + r1 * kPitch;
#pragma unroll
      for (int m = 0; m < kRowsPerThread; ++m) {
        const int j = lj0 + 4 * m;
        double t = lsr::dadd(0.0, lsr::dmul(static_cast<double>(s0[j]), w0));
        t = lsr::dadd(t, lsr::dmul(static_cast<double>(s1[j]), w1));
        const float d = static_cast<float>(t);
        acc[m] = (k == 0) ? d : (acc[m] + d);
      }
    } else {
      // outside the stack: the fill value, added like any other sample of the group (scipy returns cval there)
#pragma unroll
      for (int m = 0; m < kRowsPerThread; ++m) acc[m] = (k == 0) ? cv : (acc[m] + cv);
    }
  }

  if (li < n_xo) {
    const float denom = static_cast<float>(p.avg_n);
    float* dst = p.out + zo * p.out_plane + yo0 * p.out_pitch + xo0 + li;
#pragma unroll
    for (int m = 0; m < kRowsPerThread; ++m) {
      const int j = lj0 + 4 * m;
      if (j < n_yo) dst[static_cast<int64_t>(j) * p.out_pitch] = (p.avg_n > 1) ? acc[m] / denom : acc[m];
    }
  }
}

__global__ __launch_bounds__(256) void average_slices_kernel(const float* __restrict__ in,
                                                             float* __restrict__ out, int64_t Zd,
                                                             int64_t plane, int64_t Zo,
                                                             int avg_n) {
  const int64_t total = Zo * plane;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += stride) {
    const int64_t zo = i / plane;
    const int64_t r = i - zo * plane;
    float acc = 0.0f;
    for (int k = 0; k < avg_n; ++k) {
      const int64_t zd = min(zo * avg_n + k, Zd - 1);
      const float d = in[zd * plane + r];
      acc = (k == 0) ? d : (acc + d);
    }
    out[i] = (avg_n > 1) ? acc / static_cast<float>(avg_n) : acc;
  }
}

bool is_integer(double v) { return v == static_cast<double>(static_cast<int64_t>(v)); }

}  // namespace

namespace {

int deskew_impl(const char* what, const void* in, bool u16, int64_t Z, int64_t Y, int64_t X, float* out,
                int64_t Zo, int64_t Yo, int64_t Xo, int64_t out_pitch, int64_t out_plane,
                int64_t Zd, const double M[12], int avg_n, const float* flat_pattern,
                const float* flat_mean, lsr_stream_t stream, int mode = LSR_MODE_CONSTANT, const float* cval = nullptr) {
  LSR_REQUIRE_PTR(in);
  LSR_REQUIRE(mode == LSR_MODE_CONSTANT || mode == LSR_MODE_GRID_CONSTANT, LSR_E_ARG,
              "mode %d: LSR_MODE_CONSTANT or LSR_MODE_GRID_CONSTANT", mode);
  LSR_REQUIRE_PTR(out);
  LSR_REQUIRE_PTR(M);
  LSR_REQUIRE(Z > 0 && Y > 0 && X > 0, LSR_E_SHAPE, "raw shape (%lld,%lld,%lld) must be positive",
              (long long)Z, (long long)Y, (long long)X);
  LSR_REQUIRE_VOLUME(Z, Y, X);
  LSR_REQUIRE(Zo > 0 && Yo > 0 && Xo > 0 && Zd > 0, LSR_E_SHAPE,
              "output shape (%lld,%lld,%lld) / Zd %lld must be positive", (long long)Zo,
              (long long)Yo, (long long)Xo, (long long)Zd);
  LSR_REQUIRE_VOLUME(Zo, Yo, Xo);
  LSR_REQUIRE_VOLUME(Zd, Yo, Xo);
  LSR_REQUIRE(avg_n >= 1 && avg_n <= kMaxAvg, LSR_E_ARG, "avg_n %d outside [1,%d]", avg_n,
              kMaxAvg);
  LSR_REQUIRE(Zo == lsr::ceil_div(Zd, avg_n), LSR_E_SHAPE,
              "Zo %lld != ceil(Zd %lld / avg_n %d)", (long long)Zo, (long long)Zd, avg_n);
  LSR_REQUIRE_STRIDES(out_pitch, out_plane);
  LSR_REQUIRE(out_pitch >= Xo && out_plane >= Yo * out_pitch, LSR_E_SHAPE,
              "output strides (%lld, %lld) are smaller than the output plane (%lld x %lld)",
              (long long)out_pitch, (long long)out_plane, (long long)Yo, (long long)Xo);
  for (int i = 0; i < 12; ++i)
    LSR_REQUIRE(M[i] == M[i] && M[i] - M[i] == 0.0, LSR_E_ARG, "M[%d] is not finite", i);

  // deskew structure: only z_in is fractional
  const bool structured = M[1] == 0.0 && (M[4] == 1.0 || M[4] == -1.0) && M[5] == 0.0 &&
                          M[6] == 0.0 && is_integer(M[7]) && M[8] == 0.0 &&
                          (M[9] == 1.0 || M[9] == -1.0) && M[10] == 0.0 && is_integer(M[11]);
  LSR_REQUIRE(structured, LSR_E_UNSUPPORTED,
              "matrix is not a deskew shear (rows 1,2 must be signed unit axes with integer "
              "offsets, M[0][1] == 0): use lsr_affine_f32 + lsr_average_slices_f32");

  DeskewArgs p;
  p.in = in;
  p.out = out;
  p.Z = Z; p.Y = Y; p.X = X;
  p.Zo = Zo; p.Yo = Yo; p.Xo = Xo; p.Zd = Zd;
  p.out_pitch = out_pitch; p.out_plane = out_plane;
  p.a = M[0]; p.b = M[2]; p.c = M[3];
  p.sy = static_cast<int>(M[4]); p.oy = static_cast<int64_t>(M[7]);
  p.sx = static_cast<int>(M[9]); p.ox = static_cast<int64_t>(M[11]);
  p.avg_n = avg_n;
  p.grid = mode == LSR_MODE_GRID_CONSTANT ? 1 : 0;

  // slab rows needed by a tile of w X' values: floor span of |b|*(w-1) plus the +1 neighbour
  const double ab = p.b < 0 ? -p.b : p.b;
  int tile_x = kTileX;
  while (tile_x > 1 && ab * (tile_x - 1) + 3.0 > static_cast<double>(kSlabRows)) tile_x >>= 1;
  LSR_REQUIRE(ab * (tile_x - 1) + 3.0 <= static_cast<double>(kSlabRows), LSR_E_UNSUPPORTED,
              "scan step per output voxel |b| = %g is too large for the LDS slab", ab);
  p.tile_x = tile_x;
  p.tiles_x = lsr::ceil_div(Xo, tile_x);
  p.tiles_y = lsr::ceil_div(Yo, kTileY);
  const int64_t blocks = p.tiles_x * p.tiles_y * Zo;
  LSR_REQUIRE(blocks < (int64_t(1) << 31), LSR_E_SHAPE, "grid of %lld workgroups is too large",
              (long long)blocks);

  p.flat_pattern = flat_pattern;
  p.flat_mean = flat_mean;
  p.cval = cval;
  p.blocks = blocks;
  p.xcd_swizzle = 1;
  if (const char* e = std::getenv("LSR_DESKEW_SWIZZLE")) p.xcd_swizzle = e[0] != '0';   // measurement override
  const dim3 grid(static_cast<unsigned>(p.xcd_swizzle ? 8 * ((blocks + 7) / 8) : blocks)), block(kThreads);
  hipStream_t s = lsr::as_stream(stream);
  if (u16 && flat_pattern != nullptr) hipLaunchKernelGGL((deskew_kernel<true, true>), grid, block, 0, s, p);
  else if (u16) hipLaunchKernelGGL((deskew_kernel<false, true>), grid, block, 0, s, p);
  else if (flat_pattern != nullptr) hipLaunchKernelGGL((deskew_kernel<true, false>), grid, block, 0, s, p);
  else hipLaunchKernelGGL((deskew_kernel<false, false>), grid, block, 0, s, p);
  return lsr::launch_status(what);
}

}  // namespace

extern "C" int lsr_deskew_f32(const float* in, int64_t Z, int64_t Y, int64_t X, float* out,
                              int64_t Zo, int64_t Yo, int64_t Xo, int64_t out_pitch,
                              int64_t out_plane, int64_t Zd, const double M[12], int avg_n,
                              lsr_stream_t stream) {
  return deskew_impl("lsr_deskew_f32", in, false, Z, Y, X, out, Zo, Yo, Xo, out_pitch, out_plane, Zd, M,
                     avg_n, nullptr, nullptr, stream);
}

extern "C" int lsr_deskew_u16(const uint16_t* in, int64_t Z, int64_t Y, int64_t X, float* out,
                              int64_t Zo, int64_t Yo, int64_t Xo, int64_t out_pitch,
                              int64_t out_plane, int64_t Zd, const double M[12], int avg_n,
                              lsr_stream_t stream) {
  return deskew_impl("lsr_deskew_u16", in, true, Z, Y, X, out, Zo, Yo, Xo, out_pitch, out_plane, Zd, M,
                     avg_n, nullptr, nullptr, stream);
}

extern "C" int lsr_deskew_flat_u16(const uint16_t* in, int64_t Z, int64_t Y, int64_t X, float* out,
                                   int64_t Zo, int64_t Yo, int64_t Xo, int64_t out_pitch,
                                   int64_t out_plane, int64_t Zd, const double M[12], int avg_n,
                                   const float* flat_pattern, const float* flat_mean,
                                   lsr_stream_t stream) {
  LSR_REQUIRE_PTR(flat_pattern);
  LSR_REQUIRE_PTR(flat_mean);
  return deskew_impl("lsr_deskew_flat_u16", in, true, Z, Y, X, out, Zo, Yo, Xo, out_pitch, out_plane, Zd,
                     M, avg_n, flat_pattern, flat_mean, stream);
}

extern "C" int lsr_deskew_flat_f32(const float* in, int64_t Z, int64_t Y, int64_t X, float* out,
                                   int64_t Zo, int64_t Yo, int64_t Xo, int64_t out_pitch,
                                   int64_t out_plane, int64_t Zd, const double M[12], int avg_n,
                                   const float* flat_pattern, const float* flat_mean,
                                   lsr_stream_t stream) {
  LSR_REQUIRE_PTR(flat_pattern);
  LSR_REQUIRE_PTR(flat_mean);
  return deskew_impl("lsr_deskew_flat_f32", in, false, Z, Y, X, out, Zo, Yo, Xo, out_pitch, out_plane, Zd,
                     M, avg_n, flat_pattern, flat_mean, stream);
}

extern "C" int lsr_deskew_border(const void* in, int in_u16, int64_t Z, int64_t Y, int64_t X, float* out,
                                 int64_t Zo, int64_t Yo, int64_t Xo, int64_t out_pitch, int64_t out_plane,
                                 int64_t Zd, const double M[12], int avg_n, int mode, const float* flat_pattern,
                                 const float* flat_mean, lsr_stream_t stream) {
  LSR_REQUIRE((flat_pattern == nullptr) == (flat_mean == nullptr), LSR_E_NULL,
              "flat_pattern and flat_mean come together (both NULL: no flat-field correction)");
  return deskew_impl("lsr_deskew_border", in, in_u16 != 0, Z, Y, X, out, Zo, Yo, Xo, out_pitch, out_plane, Zd, M,
                     avg_n, flat_pattern, flat_mean, stream, mode);
}

extern "C" int lsr_deskew_cval(const void* in, int in_u16, int64_t Z, int64_t Y, int64_t X, float* out,
                               int64_t Zo, int64_t Yo, int64_t Xo, int64_t out_pitch, int64_t out_plane,
                               int64_t Zd, const double M[12], int avg_n, int mode, const float* flat_pattern,
                               const float* flat_mean, const float* cval, lsr_stream_t stream) {
  LSR_REQUIRE((flat_pattern == nullptr) == (flat_mean == nullptr), LSR_E_NULL,
              "flat_pattern and flat_mean come together (both NULL: no flat-field correction)");
  return deskew_impl("lsr_deskew_cval", in, in_u16 != 0, Z, Y, X, out, Zo, Yo, Xo, out_pitch, out_plane, Zd, M,
                     avg_n, flat_pattern, flat_mean, stream, mode, cval);
}

extern "C" int lsr_average_slices_f32(const float* in, int64_t Zd, int64_t Y, int64_t X,
                                      float* out, int64_t Zo, int avg_n, lsr_stream_t stream) {
  LSR_REQUIRE_PTR(in);
  LSR_REQUIRE_PTR(out);
  LSR_REQUIRE(Zd > 0 && Y > 0 && X > 0 && Zo > 0, LSR_E_SHAPE, "shape must be positive");
  LSR_REQUIRE_VOLUME(Zd, Y, X);
  LSR_REQUIRE_VOLUME(Zo, Y, X);
  LSR_REQUIRE(avg_n >= 1 && avg_n <= kMaxAvg, LSR_E_ARG, "avg_n %d outside [1,%d]", avg_n,
              kMaxAvg);
  LSR_REQUIRE(Zo == lsr::ceil_div(Zd, avg_n), LSR_E_SHAPE, "Zo %lld != ceil(Zd %lld / avg_n %d)",
              (long long)Zo, (long long)Zd, avg_n);
  const int64_t plane = Y * X;
  const int64_t total = Zo * plane;
  const int64_t blocks = lsr::ceil_div(total, 256) < 8192 ? lsr::ceil_div(total, 256) : 8192;
  hipLaunchKernelGGL(average_slices_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0,
                     lsr::as_stream(stream), in, out, Zd, plane, Zo, avg_n);
  return lsr::launch_status("lsr_average_slices_f32");
}
