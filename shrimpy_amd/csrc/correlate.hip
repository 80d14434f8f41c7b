// 3-D PSF correlation with fused Richardson-Lucy epilogues for gfx950.
//
// North-star: "each RL iteration is a 3D PSF stencil with LDS tiling"; the reference has no RL
// symbol (docs/data_structure.md:58-62), the CPU path is an explicit loop over
// scipy.ndimage.convolve / correlate (oracle/cpu_ref.py:richardson_lucy).
//
// Structure: 2.5-D streaming.  A workgroup owns a (32 x 64) column of (y, x) and marches along
// z.  Every input plane is staged ONCE in LDS with its in-plane halo, filtered in-plane, and
// its contribution is scattered into PZ per-thread register accumulators -- one per pending
// output plane -- so there is no z-halo re-read and no z traffic through LDS.  When an output
// plane has received all PZ contributions it is finished by the fused epilogue:
//     RATIO : out = y / (c + eps)           (first half of an RL iteration)
//     UPDATE: out = x * c / (H^T 1)         (second half; H^T 1 evaluated analytically)
// Algorithmic HBM bytes per voxel per launch: 4 (in) + 4 (aux) + 4 (out) = 12.
//
// Separable PSFs (rank-1: wz x wy x wx) take pz+py+px FMAs per voxel: HBM-bound.
// Dense PSFs take pz*py*px FMAs per voxel: fp32-VALU-bound beyond ~120 taps.  No MFMA.

#include <cstdlib>

#include "common.hpp"
#include "correlate_common.hpp"

namespace {

constexpr int kThreads = 256;
constexpr int kTileY = 32;
constexpr int kTileX = 64;
constexpr int kRun = kTileY / (kThreads / 64);  // 8 consecutive y per thread, one x column
constexpr int kMaxTaps = 15;                    // per axis
constexpr int kARows = kTileY + kMaxTaps - 1;   // staged plane rows incl. halo
constexpr int kAPitch = kTileX + kMaxTaps - 1 + 2;
constexpr int kBPitch = kTileX;

using lsr::CorrArgs;

// H^T 1 for the dense form: sum of the taps whose sample lies inside the volume, from the
// inclusive prefix-sum table P[a][b][c] = sum_{a'<a,b'<b,c'<c} w.
__device__ double dense_norm(const CorrArgs& p, int64_t z, int64_t y, int64_t x) {
  const int cz = p.pz / 2, cy = p.py / 2, cx = p.px / 2;
  const int a0 = static_cast<int>(max(int64_t(0), cz - z));
  const int a1 = static_cast<int>(min(int64_t(p.pz), p.Z - z + cz));
  const int b0 = static_cast<int>(max(int64_t(0), cy - y));
  const int b1 = static_cast<int>(min(int64_t(p.py), p.Y - y + cy));
  const int c0 = static_cast<int>(max(int64_t(0), cx - x));
  const int c1 = static_cast<int>(min(int64_t(p.px), p.X - x + cx));
  const int sb = p.px + 1, sa = (p.py + 1) * sb;
  const double* P = p.norm_table;
  if (a0 == 0 && a1 == p.pz && b0 == 0 && b1 == p.py && c0 == 0 && c1 == p.px)
    return P[p.pz * sa + p.py * sb + p.px];  // interior: every tap lands inside
  return ((P[a1 * sa + b1 * sb + c1] - P[a0 * sa + b1 * sb + c1]) -
          (P[a1 * sa + b0 * sb + c1] - P[a0 * sa + b0 * sb + c1])) -
         ((P[a1 * sa + b1 * sb + c0] - P[a0 * sa + b1 * sb + c0]) -
          (P[a1 * sa + b0 * sb + c0] - P[a0 * sa + b0 * sb + c0]));
}

template <int PZ, bool SEP>
__global__ __launch_bounds__(kThreads) void correlate_march_kernel(CorrArgs p) {
  __shared__ float bufA[kARows * kAPitch];
  __shared__ float bufB[SEP ? kARows * kBPitch : 1];

  const int tid = threadIdx.x;
  const int lx = tid & 63;
  const int yq = tid >> 6;

  int64_t bid = blockIdx.x;
  const int64_t tx = bid % p.tiles_x;
  bid /= p.tiles_x;
  const int64_t ty = bid % p.tiles_y;
  const int64_t zc = bid / p.tiles_y;

  const int64_t x0 = tx * kTileX;
  const int64_t y0 = ty * kTileY;
  const int64_t zb = zc * p.z_chunk;
  const int64_t ze = min(zb + p.z_chunk, p.Z);

  const int py = p.py, px = p.px;
  const int cz = PZ / 2, cy = py / 2, cx = px / 2;
  const int a_rows = kTileY + py - 1;
  const int a_cols = kTileX + px - 1;
  const int64_t plane = p.Y * p.X;

  // pending output planes: acc[m][j] <-> z_out = zi - cz + j after input plane zi is absorbed
  float acc[kRun][PZ];
#pragma unroll
  for (int m = 0; m < kRun; ++m)
#pragma unroll
    for (int j = 0; j < PZ; ++j) acc[m][j] = 0.0f;

  float wzr[PZ];
  if constexpr (SEP) {
#pragma unroll
    for (int a = 0; a < PZ; ++a) wzr[a] = p.wz[a];
  }

  lsr::RlStats st;   // LSR_EPI_UPDATE with p.stats: the launch's reduction scalars (correlate_common.hpp)
  const int64_t zi_begin = max(zb - cz, int64_t(0));
  const int64_t zi_end = ze + cz;  // exclusive
  for (int64_t zi = zi_begin; zi < zi_end; ++zi) {
    const bool have_plane = zi < p.Z;
    float pl[kRun];
#pragma unroll
    for (int m = 0; m < kRun; ++m) pl[m] = 0.0f;

    if (have_plane) {
      __syncthreads();  // previous plane's readers of bufA (and bufB) are done
      // ---- stage plane zi with its (y, x) halo; zero outside the volume ----
      const float* src = p.in + zi * plane;
      for (int e = tid; e < a_rows * a_cols; e += kThreads) {
        const int r = e / a_cols;
        const int c = e - r * a_cols;
        const int64_t gy = y0 + r - cy;
        const int64_t gx = x0 + c - cx;
        float v = 0.0f;
        if (gy >= 0 && gy < p.Y && gx >= 0 && gx < p.X) v = src[gy * p.X + gx];
        bufA[r * kAPitch + c] = v;
      }
      __syncthreads();

      if constexpr (SEP) {
        // ---- x pass: bufB[r][x] = sum_c wx[c] * bufA[r][x + c] ----
        for (int r = yq; r < a_rows; r += kThreads / 64) {
          const float* row = bufA + r * kAPitch + lx;
          float s = 0.0f;
          for (int c = 0; c < px; ++c) s = fmaf(p.wx[c], row[c], s);
          bufB[r * kBPitch + lx] = s;
        }
        __syncthreads();
        // ---- y pass: pl[m] = sum_b wy[b] * bufB[yq*8 + m + b][x] ----
        const float* col = bufB + (yq * kRun) * kBPitch + lx;
        for (int b = 0; b < py; ++b) {
          const float wb = p.wy[b];
#pragma unroll
          for (int m = 0; m < kRun; ++m) pl[m] = fmaf(wb, col[(m + b) * kBPitch], pl[m]);
        }
      }
    }

    if constexpr (SEP) {
      // ---- z: shift the pending planes and add this plane's contribution in one FMA ----
#pragma unroll
      for (int m = 0; m < kRun; ++m) {
#pragma unroll
        for (int j = 0; j < PZ - 1; ++j) acc[m][j] = fmaf(wzr[PZ - 1 - j], pl[m], acc[m][j + 1]);
        acc[m][PZ - 1] = wzr[0] * pl[m];
      }
    } else {
#pragma unroll
      for (int m = 0; m < kRun; ++m) {
#pragma unroll
        for (int j = 0; j < PZ - 1; ++j) acc[m][j] = acc[m][j + 1];
        acc[m][PZ - 1] = 0.0f;
      }
      if (have_plane) {
        const float* base = bufA + (yq * kRun) * kAPitch + lx;
        const int tap_plane = py * px;
        for (int b = 0; b < py; ++b) {
          for (int c = 0; c < px; ++c) {
            float wv[PZ];
#pragma unroll
            for (int a = 0; a < PZ; ++a) wv[a] = p.w[a * tap_plane + b * px + c];
#pragma unroll
            for (int m = 0; m < kRun; ++m) {
              const float v = base[(m + b) * kAPitch + c];
#pragma unroll
              for (int j = 0; j < PZ; ++j) acc[m][j] = fmaf(wv[PZ - 1 - j], v, acc[m][j]);
            }
          }
        }
      }
    }

    // ---- epilogue: output plane z_out has now seen all PZ input planes ----
    const int64_t z_out = zi - cz;
    if (z_out >= zb) {
      const int64_t gx = x0 + lx;
      if (gx < p.X) {
#pragma unroll
        for (int m = 0; m < kRun; ++m) {
          const int64_t gy = y0 + yq * kRun + m;
          if (gy < p.Y) {
            const int64_t o = z_out * plane + gy * p.X + gx;
            const float c = acc[m][0];
            float r;
            if (p.epilogue == LSR_EPI_RATIO) {
              r = p.aux[o] / (c + p.eps);
            } else if (p.epilogue == LSR_EPI_UPDATE) {
              float nrm;
              if constexpr (SEP) {
                nrm = p.nz[z_out] * p.ny[gy] * p.nx[gx];
              } else {
                nrm = static_cast<float>(dense_norm(p, z_out, gy, gx));
              }
              const float a = p.aux[o], ac = a * c;
              r = ac / nrm;
              if (p.stats) st.add(a, ac, r);
            } else {
              r = c;
            }
            p.out[o] = r;
          }
        }
      }
    }
  }
  if (p.stats) {   // (kernel-uniform)
    __syncthreads();
    lsr::rl_stats_flush<kThreads / 64>(st, bufA, p.stats);
  }
}

template <bool SEP>
int launch_correlate(const CorrArgs& p, hipStream_t s) {
  const int64_t blocks = p.tiles_x * p.tiles_y * lsr::ceil_div(p.Z, p.z_chunk);
  if (blocks >= (int64_t(1) << 31))
    return lsr::fail(LSR_E_SHAPE, "grid of %lld workgroups is too large", (long long)blocks);
  const dim3 grid(static_cast<unsigned>(blocks)), block(kThreads);
  switch (p.pz) {
#define LSR_CASE(N)                                                                    \
  case N:                                                                              \
    hipLaunchKernelGGL((correlate_march_kernel<N, SEP>), grid, block, 0, s, p);        \
    break;
    LSR_CASE(1)
    LSR_CASE(3)
    LSR_CASE(5)
    LSR_CASE(7)
    LSR_CASE(9)
    LSR_CASE(11)
    LSR_CASE(13)
    LSR_CASE(15)
#undef LSR_CASE
    default:
      return lsr::fail(LSR_E_UNSUPPORTED, "pz = %d: taps per axis must be odd and <= %d", p.pz,
                       kMaxTaps);
  }
  return lsr::launch_status(SEP ? "lsr_correlate_sep_f32" : "lsr_correlate_dense_f32");
}

int check_common(const float* in, float* out, const float* aux, int64_t Z, int64_t Y, int64_t X,
                 int pz, int py, int px, int epilogue) {
  LSR_REQUIRE_PTR(in);
  LSR_REQUIRE_PTR(out);
  LSR_REQUIRE(Z > 0 && Y > 0 && X > 0, LSR_E_SHAPE, "shape (%lld,%lld,%lld) must be positive",
              (long long)Z, (long long)Y, (long long)X);
  LSR_REQUIRE_VOLUME(Z, Y, X);
  LSR_REQUIRE(pz >= 1 && py >= 1 && px >= 1 && (pz & 1) && (py & 1) && (px & 1) &&
                  pz <= kMaxTaps && py <= kMaxTaps && px <= kMaxTaps,
              LSR_E_UNSUPPORTED, "PSF taps (%d,%d,%d) must be odd and <= %d per axis", pz, py, px,
              kMaxTaps);
  LSR_REQUIRE(epilogue == LSR_EPI_NONE || epilogue == LSR_EPI_RATIO || epilogue == LSR_EPI_UPDATE,
              LSR_E_ARG, "unknown epilogue %d", epilogue);
  if (epilogue != LSR_EPI_NONE) LSR_REQUIRE_PTR(aux);
  LSR_REQUIRE(in != out, LSR_E_ARG, "out must not alias in");
  return LSR_OK;
}

// Split z only as far as needed to fill the chip: every chunk re-reads pz-1 halo planes.
int64_t pick_z_chunk(int64_t Z, int64_t tiles_xy, int pz) {
  const int64_t want = 256 * 8;  // workgroups for ~8 per CU
  int64_t chunks = lsr::ceil_div(want, tiles_xy);
  if (chunks < 1) chunks = 1;
  int64_t chunk = lsr::ceil_div(Z, chunks);
  const int64_t min_chunk = 8 * static_cast<int64_t>(pz);  // keep halo overhead <= ~12 %
  if (chunk < min_chunk) chunk = min_chunk;
  if (chunk > Z) chunk = Z;
  return chunk;
}

}  // namespace

extern "C" int lsr_correlate_sep_stats_f32(const float* in, float* out, const float* aux, int64_t Z, int64_t Y, int64_t X, const float* wz, int pz, const float* wy, int py, const float* wx, int px, int epilogue, float eps, const float* nz, const float* ny, const float* nx, double* stats, lsr_stream_t stream) {
  // Dense volumes without a halo: the generic bounds-checked kernel (correct, not tuned).
  if (int rc = check_common(in, out, aux, Z, Y, X, pz, py, px, epilogue)) return rc;
  LSR_REQUIRE_PTR(wz);
  LSR_REQUIRE_PTR(wy);
  LSR_REQUIRE_PTR(wx);
  if (epilogue == LSR_EPI_UPDATE) {
    LSR_REQUIRE_PTR(nz);
    LSR_REQUIRE_PTR(ny);
    LSR_REQUIRE_PTR(nx);
  }
  CorrArgs p{};
  p.in = in; p.out = out; p.aux = aux;
  p.Z = Z; p.Y = Y; p.X = X;
  p.wz = wz; p.wy = wy; p.wx = wx;
  p.pz = pz; p.py = py; p.px = px;
  p.epilogue = epilogue; p.eps = eps;
  p.nz = nz; p.ny = ny; p.nx = nx;
  p.stats = epilogue == LSR_EPI_UPDATE ? stats : nullptr;
  p.tiles_x = lsr::ceil_div(X, kTileX);
  p.tiles_y = lsr::ceil_div(Y, kTileY);
  p.z_chunk = pick_z_chunk(Z, p.tiles_x * p.tiles_y, pz);
  return launch_correlate<true>(p, lsr::as_stream(stream));
}

extern "C" int lsr_correlate_sep_f32(const float* in, float* out, const float* aux, int64_t Z, int64_t Y, int64_t X, const float* wz, int pz, const float* wy, int py, const float* wx, int px, int epilogue, float eps, const float* nz, const float* ny, const float* nx, lsr_stream_t stream) {
  return lsr_correlate_sep_stats_f32(in, out, aux, Z, Y, X, wz, pz, wy, py, wx, px, epilogue, eps, nz, ny, nx, nullptr, stream);
}

namespace {

// compiled tap counts of the tuned kernel: 3,5,...,15 along z, square 3..15 in plane; smaller
// PSFs are centred in the next size up (zero taps elsewhere)
void sep_compiled_taps(int pz, int py, int px, int* PZ, int* PYX) {
  *PZ = lsr::sep_round_taps(pz);
  *PYX = lsr::sep_round_taps(py > px ? py : px);
}

// The RL loops own their scalars: stats[3 * iters] is zeroed on the stream before the first launch adds to it.
int zero_stats(double* stats, int iters, lsr_stream_t stream) {
  if (stats == nullptr || iters <= 0) return LSR_OK;
  const hipError_t e = hipMemsetAsync(stats, 0, sizeof(double) * lsr::kRlStats * static_cast<size_t>(iters), lsr::as_stream(stream));
  if (e != hipSuccess) return lsr::fail(static_cast<int>(e), "hipMemsetAsync(stats): %s", hipGetErrorString(e));
  return LSR_OK;
}

int check_taps(int pz, int py, int px) {
  LSR_REQUIRE(pz >= 1 && py >= 1 && px >= 1 && (pz & 1) && (py & 1) && (px & 1) &&
                  pz <= kMaxTaps && py <= kMaxTaps && px <= kMaxTaps,
              LSR_E_UNSUPPORTED, "PSF taps (%d,%d,%d) must be odd and <= %d per axis", pz, py, px,
              kMaxTaps);
  return LSR_OK;
}

}  // namespace

extern "C" int lsr_sep_padded_shape(int64_t Y, int64_t X, int pz, int py, int px,
                                    int64_t shape[4]) {
  LSR_REQUIRE_PTR(shape);
  LSR_REQUIRE(Y > 0 && X > 0, LSR_E_SHAPE, "plane (%lld,%lld) must be positive", (long long)Y,
              (long long)X);
  LSR_REQUIRE_VOLUME(1, Y, X);
  if (int rc = check_taps(pz, py, px)) return rc;
  int PZ, PYX;
  sep_compiled_taps(pz, py, px, &PZ, &PYX);
  // the volume must satisfy every tuned kernel: dense (32 x 64 tiles), separable (32/24 x 128) and
  // the fused RL iteration (same tiles, twice the halo: two stencils back to back)
  auto up = [](int64_t v, int64_t m) { return lsr::ceil_div(v, m) * m; };
  const int64_t rows_a = up(Y, 32), rows_b = up(Y, 24);
  const int64_t halo = 2 * (PYX / 2);
  shape[0] = (rows_a > rows_b ? rows_a : rows_b) + 2 * halo;              // rows
  const int64_t cols_a = (lsr::ceil_div(X, lsr::kSepTileX) - 1) * lsr::kSepTileX + lsr::sep_stage_cols(PYX);
  const int64_t cols_b =
      (lsr::ceil_div(X, lsr::kSepWideTileX) - 1) * lsr::kSepWideTileX + lsr::sep_wide_stage_cols(PYX);
  const int64_t cols_c = PYX / 2 + up(X, lsr::kSepWideTileX) + lsr::fused_window_halo(PYX);
  int64_t cols = cols_a > cols_b ? cols_a : cols_b;
  if (cols_c > cols) cols = cols_c;
  // pitch: a multiple of 32 floats (128-B lines) covering the last tile's staged window
  shape[1] = up(lsr::kSepOriginCol - PYX / 2 + cols, 32);
  shape[2] = halo;                                                        // row of logical y = 0
  shape[3] = lsr::kSepOriginCol;                                          // col of logical x = 0
  return LSR_OK;
}

extern "C" int lsr_correlate_sep_strided_stats_f32(const float* in, int64_t in_pitch, int64_t in_plane, const float* aux, int64_t aux_pitch, int64_t aux_plane, float* out, int64_t out_pitch, int64_t out_plane, int64_t Z, int64_t Y, int64_t X, const float* wz, int pz, const float* wy, int py, const float* wx, int px, int epilogue, float eps, const float* nz, const float* ny, const float* nx, double* stats, lsr_stream_t stream) {
  LSR_REQUIRE_PTR(in);
  LSR_REQUIRE_PTR(out);
  LSR_REQUIRE_PTR(wz);
  LSR_REQUIRE_PTR(wy);
  LSR_REQUIRE_PTR(wx);
  LSR_REQUIRE(Z > 0 && Y > 0 && X > 0, LSR_E_SHAPE, "shape (%lld,%lld,%lld) must be positive",
              (long long)Z, (long long)Y, (long long)X);
  LSR_REQUIRE_VOLUME(Z, Y, X);
  if (int rc = check_taps(pz, py, px)) return rc;
  LSR_REQUIRE(epilogue == LSR_EPI_NONE || epilogue == LSR_EPI_RATIO || epilogue == LSR_EPI_UPDATE,
              LSR_E_ARG, "unknown epilogue %d", epilogue);
  if (epilogue != LSR_EPI_NONE) LSR_REQUIRE_PTR(aux);
  if (epilogue == LSR_EPI_UPDATE) {
    LSR_REQUIRE_PTR(nz);
    LSR_REQUIRE_PTR(ny);
    LSR_REQUIRE_PTR(nx);
  }
  LSR_REQUIRE(in != out, LSR_E_ARG, "out must not alias in");
  int PZ, PYX;
  sep_compiled_taps(pz, py, px, &PZ, &PYX);
  int64_t need[4];
  lsr_sep_padded_shape(Y, X, pz, py, px, need);
  LSR_REQUIRE_STRIDES(in_pitch, in_plane);
  LSR_REQUIRE(in_pitch >= need[1] && in_plane >= need[0] * in_pitch, LSR_E_SHAPE,
              "in strides (%lld,%lld) are smaller than the padded shape (%lld rows x %lld) that "
              "lsr_sep_padded_shape asks for",
              (long long)in_pitch, (long long)in_plane, (long long)need[0], (long long)need[1]);
  LSR_REQUIRE(in_pitch % 4 == 0 && in_plane % 4 == 0, LSR_E_ARG,
              "pitch and plane stride of the padded input must be multiples of 4 floats");
  const int64_t lim = int64_t(1) << 30;
  LSR_REQUIRE(in_plane < lim && aux_plane < lim && out_plane < lim && Z < lim, LSR_E_UNSUPPORTED,
              "plane strides exceed the kernel's 32-bit in-plane offsets");
  LSR_REQUIRE(out_pitch >= X && (epilogue == LSR_EPI_NONE || aux_pitch >= X), LSR_E_SHAPE,
              "aux/out pitch smaller than X");

  lsr::SepArgs p{};
  p.in = in; p.aux = aux; p.out = out;
  p.in_plane = in_plane; p.aux_plane = aux_plane; p.out_plane = out_plane;
  p.in_pitch = static_cast<int>(in_pitch);
  p.aux_pitch = static_cast<int>(aux_pitch);
  p.out_pitch = static_cast<int>(out_pitch);
  p.Z = static_cast<int>(Z); p.Y = static_cast<int>(Y); p.X = static_cast<int>(X);
  p.wz = wz; p.wy = wy; p.wx = wx;
  p.pz = pz; p.py = py; p.px = px;
  p.epilogue = epilogue; p.eps = eps;
  p.nz = nz; p.ny = ny; p.nx = nx;
  p.stats = epilogue == LSR_EPI_UPDATE ? stats : nullptr;
  p.tiles_x = static_cast<int>(lsr::ceil_div(X, lsr::kSepWideTileX));
  p.tiles_y = static_cast<int>(lsr::ceil_div(Y, lsr::sep_wide_tile_y(PZ)));
  p.z_chunk = static_cast<int>(pick_z_chunk(Z, int64_t(p.tiles_x) * p.tiles_y, PZ));
  const int64_t blocks64 = int64_t(p.tiles_x) * p.tiles_y * lsr::ceil_div(Z, p.z_chunk);
  LSR_REQUIRE(blocks64 < (int64_t(1) << 31), LSR_E_SHAPE, "grid of %lld workgroups is too large",
              (long long)blocks64);
  const unsigned blocks = static_cast<unsigned>(blocks64);
  hipStream_t s = lsr::as_stream(stream);
  bool ok = false;
  switch (PZ) {
    case 3: ok = lsr::launch_sep_pz3(PYX, p, blocks, s); break;
    case 5: ok = lsr::launch_sep_pz5(PYX, p, blocks, s); break;
    case 7: ok = lsr::launch_sep_pz7(PYX, p, blocks, s); break;
    case 9: ok = lsr::launch_sep_pz9(PYX, p, blocks, s); break;
    case 11: ok = lsr::launch_sep_pz11(PYX, p, blocks, s); break;
    case 13: ok = lsr::launch_sep_pz13(PYX, p, blocks, s); break;
    case 15: ok = lsr::launch_sep_pz15(PYX, p, blocks, s); break;
    default: break;
  }
  LSR_REQUIRE(ok, LSR_E_UNSUPPORTED, "no separable specialisation for taps (%d,%d,%d)", pz, py, px);
  return lsr::launch_status("lsr_correlate_sep_strided_f32");
}

extern "C" int lsr_correlate_sep_strided_f32(const float* in, int64_t in_pitch, int64_t in_plane, const float* aux, int64_t aux_pitch, int64_t aux_plane, float* out, int64_t out_pitch, int64_t out_plane, int64_t Z, int64_t Y, int64_t X, const float* wz, int pz, const float* wy, int py, const float* wx, int px, int epilogue, float eps, const float* nz, const float* ny, const float* nx, lsr_stream_t stream) {
  return lsr_correlate_sep_strided_stats_f32(in, in_pitch, in_plane, aux, aux_pitch, aux_plane, out, out_pitch, out_plane, Z, Y, X, wz, pz, wy, py, wx, px, epilogue, eps, nz, ny, nx, nullptr, stream);
}

extern "C" int lsr_correlate_dense_stats_f32(const float* in, float* out, const float* aux, int64_t Z, int64_t Y, int64_t X, const float* w, int pz, int py, int px, int epilogue, float eps, const double* norm_table, double* stats, lsr_stream_t stream) {
  if (int rc = check_common(in, out, aux, Z, Y, X, pz, py, px, epilogue)) return rc;
  LSR_REQUIRE_PTR(w);
  if (epilogue == LSR_EPI_UPDATE) LSR_REQUIRE_PTR(norm_table);
  CorrArgs p{};
  p.in = in; p.out = out; p.aux = aux;
  p.Z = Z; p.Y = Y; p.X = X;
  p.w = w;
  p.pz = pz; p.py = py; p.px = px;
  p.epilogue = epilogue; p.eps = eps;
  p.norm_table = norm_table;
  p.stats = epilogue == LSR_EPI_UPDATE ? stats : nullptr;
  p.tiles_x = lsr::ceil_div(X, kTileX);
  p.tiles_y = lsr::ceil_div(Y, kTileY);
  p.z_chunk = pick_z_chunk(Z, p.tiles_x * p.tiles_y, pz);
  return launch_correlate<false>(p, lsr::as_stream(stream));
}

extern "C" int lsr_correlate_dense_f32(const float* in, float* out, const float* aux, int64_t Z, int64_t Y, int64_t X, const float* w, int pz, int py, int px, int epilogue, float eps, const double* norm_table, lsr_stream_t stream) {
  return lsr_correlate_dense_stats_f32(in, out, aux, Z, Y, X, w, pz, py, px, epilogue, eps, norm_table, nullptr, stream);
}

extern "C" int lsr_rl_sep_stats_f32(const float* y, int64_t y_pitch, int64_t y_plane, int init_from_y, float* x_pad, float* ratio_pad, float* x_out, int64_t Z, int64_t Y, int64_t X, const float* kz, const float* kz_flipped, int pz, const float* ky, const float* ky_flipped, int py, const float* kx, const float* kx_flipped, int px, const float* nz, const float* ny, const float* nx, int iters, float eps, double* stats, lsr_stream_t stream) {
  LSR_REQUIRE_PTR(y);
  LSR_REQUIRE_PTR(x_pad);
  LSR_REQUIRE_PTR(ratio_pad);
  LSR_REQUIRE(iters >= 1, LSR_E_ARG, "iters %d must be >= 1", iters);
  LSR_REQUIRE(ratio_pad != x_pad, LSR_E_ARG, "x_pad and ratio_pad must be distinct");
  int64_t ps[4];
  if (int rc = lsr_sep_padded_shape(Y, X, pz, py, px, ps)) return rc;
  const int64_t pitch = ps[1], plane = ps[0] * ps[1];
  const int64_t origin = ps[2] * pitch + ps[3];
  float* xl = x_pad + origin;        // logical (0,0,0) inside the padded volumes
  float* rl = ratio_pad + origin;
  if (int rc = zero_stats(stats, iters, stream)) return rc;
  for (int it = 0; it < iters; ++it) {
    // ratio = y / (H x + eps);  H x = convolve(x, psf) = correlate(x, flipped psf).
    // With init_from_y the first iteration reads x = y straight from the (padded) y volume.
    const bool from_y = init_from_y && it == 0;
    const float* xin = from_y ? y : xl;
    const int64_t xin_pitch = from_y ? y_pitch : pitch, xin_plane = from_y ? y_plane : plane;
    int rc = lsr_correlate_sep_strided_f32(xin, xin_pitch, xin_plane, y, y_pitch, y_plane, rl, pitch,
                                           plane, Z, Y, X, kz_flipped, pz, ky_flipped, py,
                                           kx_flipped, px, LSR_EPI_RATIO, eps, nullptr, nullptr,
                                           nullptr, stream);
    if (rc) return rc;
    // x <- x * H^T ratio / H^T 1;  H^T r = correlate(r, psf).  The last update may go straight
    // to the dense result.
    const bool last = it + 1 == iters && x_out != nullptr;
    rc = lsr_correlate_sep_strided_stats_f32(rl, pitch, plane, xin, xin_pitch, xin_plane, last ? x_out : xl,
                                             last ? X : pitch, last ? Y * X : plane, Z, Y, X, kz, pz, ky,
                                             py, kx, px, LSR_EPI_UPDATE, eps, nz, ny, nx,
                                             stats ? stats + lsr::kRlStats * it : nullptr, stream);
    if (rc) return rc;
  }
  return LSR_OK;
}

extern "C" int lsr_rl_sep_f32(const float* y, int64_t y_pitch, int64_t y_plane, int init_from_y, float* x_pad, float* ratio_pad, float* x_out, int64_t Z, int64_t Y, int64_t X, const float* kz, const float* kz_flipped, int pz, const float* ky, const float* ky_flipped, int py, const float* kx, const float* kx_flipped, int px, const float* nz, const float* ny, const float* nx, int iters, float eps, lsr_stream_t stream) {
  return lsr_rl_sep_stats_f32(y, y_pitch, y_plane, init_from_y, x_pad, ratio_pad, x_out, Z, Y, X, kz, kz_flipped, pz, ky, ky_flipped, py, kx, kx_flipped, px, nz, ny, nx, iters, eps, nullptr, stream);
}

namespace {

int device_cu_count() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop{};
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

// Work split of the fused kernel (one workgroup per CU; a piece of a tile column costs
// 2 * (PZ - 1) halo planes): whole columns for as many full dispatch rounds as the tiles give,
// the remaining tiles cut along z so that they fill one more round.
void plan_fused_split(int64_t tiles_xy, int64_t Z, int PZ, int* n_full, int* pieces, int* z_chunk, int per_cu = 1) {
  const int64_t cus = device_cu_count() * per_cu;   // workgroups resident at once
  const int64_t min_chunk = 2 * (PZ - 1) > 8 ? 2 * (PZ - 1) : 8;  // halo no more than the payload
  int64_t full = tiles_xy / cus * cus;
  int64_t rest = tiles_xy - full;
  if (full == 0) {  // fewer tiles than CUs: cut every column
    rest = tiles_xy;
  }
  int64_t k = rest > 0 ? cus / rest : 1;
  if (k < 1) k = 1;
  int64_t chunk = lsr::ceil_div(Z, k);
  if (chunk < min_chunk) chunk = min_chunk < Z ? min_chunk : Z;
  *n_full = static_cast<int>(full);
  *z_chunk = static_cast<int>(chunk);
  *pieces = static_cast<int>(lsr::ceil_div(Z, chunk));
}

}  // namespace

extern "C" int lsr_rl_sep_fused_supported(int pz, int py, int px) {
  if (pz < 1 || py < 1 || px < 1 || !(pz & 1) || !(py & 1) || !(px & 1)) return 0;
  const int PZ = lsr::sep_round_taps(pz), PYX = lsr::sep_round_taps(py > px ? py : px);
  return lsr::fused_compiled(PZ, PYX) ? 1 : 0;
}

extern "C" int lsr_rl_sep_fused_taps_count(void) { return 6 * 16; }

extern "C" int lsr_rl_sep_fused_prepare_taps(const float* kz_host, int pz, const float* ky_host, int py,
                                             const float* kx_host, int px, float* taps_host) {
  LSR_REQUIRE_PTR(kz_host);
  LSR_REQUIRE_PTR(ky_host);
  LSR_REQUIRE_PTR(kx_host);
  LSR_REQUIRE_PTR(taps_host);
  if (int rc = check_taps(pz, py, px)) return rc;
  LSR_REQUIRE(lsr_rl_sep_fused_supported(pz, py, px), LSR_E_UNSUPPORTED,
              "no fused RL specialisation for taps (%d,%d,%d) (up to %d x %d x %d, not 15 z taps with "
              "11+ in-plane taps): use lsr_rl_sep_f32",
              pz, py, px, lsr::kFusedMaxPZ, lsr::kFusedMaxPYX, lsr::kFusedMaxPYX);
  int PZ, PYX;
  sep_compiled_taps(pz, py, px, &PZ, &PYX);
  // rows: stage 1 (H = correlation with the flipped PSF) x, y, z; stage 2 (H^T, the PSF) x, y, z;
  // each centred in its compiled extent, zeros elsewhere
  const float* src[3] = {kx_host, ky_host, kz_host};
  const int n[3] = {px, py, pz}, N[3] = {PYX, PYX, PZ};
  for (int i = 0; i < 96; ++i) taps_host[i] = 0.0f;
  for (int a = 0; a < 3; ++a) {
    const int off = (N[a] - n[a]) / 2;
    for (int i = 0; i < n[a]; ++i) {
      taps_host[a * 16 + off + i] = src[a][n[a] - 1 - i];
      taps_host[(3 + a) * 16 + off + i] = src[a][i];
    }
  }
  return LSR_OK;
}

#ifdef LSR_FUSED_PROBE_TIME
static unsigned long long* g_fused_probe = nullptr;
// diagnostic build only: device buffer of 4 * grid uint64 that every fused launch overwrites
extern "C" void lsr_debug_set_fused_probe(void* dev) { g_fused_probe = static_cast<unsigned long long*>(dev); }
#endif

extern "C" int lsr_rl_sep_fused_stats_f32(const float* y, int64_t y_pitch, int64_t y_plane, int init_from_y, float* x_a, float* x_b, float* x_out, int64_t Z, int64_t Y, int64_t X, const float* taps, int pz, int py, int px, const float* nz, const float* ny, const float* nx, int iters, float eps, double* stats, lsr_stream_t stream) {
  LSR_REQUIRE_PTR(y);
  LSR_REQUIRE_PTR(x_a);
  LSR_REQUIRE_PTR(x_b);
  LSR_REQUIRE_PTR(taps);
  LSR_REQUIRE_PTR(nz); LSR_REQUIRE_PTR(ny); LSR_REQUIRE_PTR(nx);
  LSR_REQUIRE(Z > 0 && Y > 0 && X > 0, LSR_E_SHAPE, "shape (%lld,%lld,%lld) must be positive",
              (long long)Z, (long long)Y, (long long)X);
  LSR_REQUIRE_VOLUME(Z, Y, X);
  LSR_REQUIRE(iters >= 1, LSR_E_ARG, "iters %d must be >= 1", iters);
  LSR_REQUIRE(x_a != x_b, LSR_E_ARG, "x_a and x_b must be distinct");
  if (int rc = check_taps(pz, py, px)) return rc;
  LSR_REQUIRE(lsr_rl_sep_fused_supported(pz, py, px), LSR_E_UNSUPPORTED,
              "no fused RL specialisation for taps (%d,%d,%d) (up to %d x %d x %d, not 15 z taps with "
              "11+ in-plane taps): use lsr_rl_sep_f32",
              pz, py, px, lsr::kFusedMaxPZ, lsr::kFusedMaxPYX, lsr::kFusedMaxPYX);
  int PZ, PYX;
  sep_compiled_taps(pz, py, px, &PZ, &PYX);
  int64_t ps[4];
  if (int rc = lsr_sep_padded_shape(Y, X, pz, py, px, ps)) return rc;
  const int64_t pitch = ps[1], plane = ps[0] * ps[1];
  const int64_t origin = ps[2] * pitch + ps[3];
  LSR_REQUIRE_STRIDES(y_pitch, y_plane);
  LSR_REQUIRE(y_pitch >= pitch && y_plane >= ps[0] * y_pitch, LSR_E_SHAPE,
              "y strides (%lld,%lld) are smaller than the padded shape (%lld rows x %lld) that "
              "lsr_sep_padded_shape asks for: y must be a zero-haloed padded volume",
              (long long)y_pitch, (long long)y_plane, (long long)ps[0], (long long)pitch);
  LSR_REQUIRE(y_pitch % 4 == 0 && y_plane % 4 == 0, LSR_E_ARG,
              "pitch and plane stride of the padded y must be multiples of 4 floats");
  const int64_t lim = int64_t(1) << 29;
  LSR_REQUIRE(plane < lim && y_plane < lim && Y * X < lim && Z < lim, LSR_E_UNSUPPORTED,
              "plane strides exceed the kernel's 32-bit in-plane offsets");

  lsr::FusedArgs p{};
#ifdef LSR_FUSED_PROBE_TIME
  p.probe = g_fused_probe;
#endif
  p.y = y; p.y_pitch = static_cast<int>(y_pitch); p.y_plane = y_plane;
  p.Z = static_cast<int>(Z); p.Y = static_cast<int>(Y); p.X = static_cast<int>(X);
  p.taps = taps; p.eps = eps;
  p.nz = nz; p.ny = ny; p.nx = nx;
  p.tiles_x = static_cast<int>(lsr::ceil_div(X, lsr::kSepWideTileX));
  p.tiles_y = static_cast<int>(lsr::ceil_div(Y, 8 * lsr::fused_run(PZ, PYX)));
  const int64_t tiles_xy = int64_t(p.tiles_x) * p.tiles_y;
  plan_fused_split(tiles_xy, Z, PZ, &p.n_full, &p.pieces, &p.z_chunk);
  const int64_t blocks64 = p.n_full + (tiles_xy - p.n_full) * p.pieces;
  LSR_REQUIRE(blocks64 < (int64_t(1) << 31), LSR_E_SHAPE, "grid of %lld workgroups is too large",
              (long long)blocks64);
  const unsigned blocks = static_cast<unsigned>(blocks64);
  hipStream_t s = lsr::as_stream(stream);

  float* bufs[2] = {x_a + origin, x_b + origin};  // logical (0,0,0) of the two working volumes
  if (int rc = zero_stats(stats, iters, stream)) return rc;
  for (int it = 0; it < iters; ++it) {
    const bool from_y = init_from_y && it == 0;
    const bool last = it + 1 == iters && x_out != nullptr;
    p.x = from_y ? y : bufs[it & 1];
    p.pitch = static_cast<int>(from_y ? y_pitch : pitch);
    p.plane = from_y ? y_plane : plane;
    p.out = last ? x_out : bufs[(it + 1) & 1];
    p.out_pitch = static_cast<int>(last ? X : pitch);
    p.out_plane = last ? Y * X : plane;
    p.mask_out = last ? 1 : 0;
    p.stats = stats ? stats + lsr::kRlStats * it : nullptr;
    bool ok = false;
    switch (PZ) {
      case 3: ok = lsr::launch_fused_pz3(PYX, p, blocks, s); break;
      case 5: ok = lsr::launch_fused_pz5(PYX, p, blocks, s); break;
      case 7: ok = lsr::launch_fused_pz7(PYX, p, blocks, s); break;
      case 9: ok = lsr::launch_fused_pz9(PYX, p, blocks, s); break;
      case 11: ok = lsr::launch_fused_pz11(PYX, p, blocks, s); break;
      case 13: ok = lsr::launch_fused_pz13(PYX, p, blocks, s); break;
      case 15: ok = lsr::launch_fused_pz15(PYX, p, blocks, s); break;
      default: break;
    }
    LSR_REQUIRE(ok, LSR_E_UNSUPPORTED, "no fused specialisation for taps (%d,%d,%d)", pz, py, px);
    if (int rc = lsr::launch_status("lsr_rl_sep_fused_f32")) return rc;
  }
  return LSR_OK;
}

extern "C" int lsr_rl_sep_fused_f32(const float* y, int64_t y_pitch, int64_t y_plane, int init_from_y, float* x_a, float* x_b, float* x_out, int64_t Z, int64_t Y, int64_t X, const float* taps, int pz, int py, int px, const float* nz, const float* ny, const float* nx, int iters, float eps, lsr_stream_t stream) {
  return lsr_rl_sep_fused_stats_f32(y, y_pitch, y_plane, init_from_y, x_a, x_b, x_out, Z, Y, X, taps, pz, py, px, nz, ny, nx, iters, eps, nullptr, stream);
}

// ---- fused iteration for psf = ky (x) kzx (rl_fused_ysep.hip) --------------------------------------------

extern "C" int lsr_rl_ysep_fused_supported(int pz, int py, int px) {
  if (pz < 1 || py < 1 || px < 1 || !(pz & 1) || !(py & 1) || !(px & 1)) return 0;
  const int PZ = lsr::sep_round_taps(pz), PYX = lsr::sep_round_taps(py > px ? py : px);
  return PZ <= lsr::kYsepMaxPZ && PYX <= lsr::kYsepMaxPYX ? 1 : 0;
}

extern "C" int lsr_rl_ysep_fused_taps_count(void) { return 2 * lsr::kYsepTapStage; }

extern "C" int lsr_rl_ysep_fused_prepare_taps(const float* ky_host, int py, const float* kzx_host, int pz, int px,
                                              float* taps_host) {
  LSR_REQUIRE_PTR(ky_host);
  LSR_REQUIRE_PTR(kzx_host);
  LSR_REQUIRE_PTR(taps_host);
  if (int rc = check_taps(pz, py, px)) return rc;
  LSR_REQUIRE(lsr_rl_ysep_fused_supported(pz, py, px), LSR_E_UNSUPPORTED,
              "no fused ky (x) kzx specialisation for taps (%d,%d,%d) (up to %d z taps, %d in-plane): use "
              "lsr_correlate_zxy_padded_f32", pz, py, px, lsr::kYsepMaxPZ, lsr::kYsepMaxPYX);
  const int PZ = lsr::sep_round_taps(pz), PYX = lsr::sep_round_taps(py > px ? py : px);
  const int oz = (PZ - pz) / 2, oy = (PYX - py) / 2, ox = (PYX - px) / 2;
  for (int i = 0; i < 2 * lsr::kYsepTapStage; ++i) taps_host[i] = 0.0f;
  // two stages of kYsepTapStage floats: stage 1 (H = correlation with the reversed PSF), stage 2 (H^T, the PSF itself);
  // the (z, x) taps of column offset c at [kYsepTapGroup * c + j], j = PZ - 1 - a (the order a staged plane feeds the
  // pending planes in) -- one 64-byte group per column offset, fetched by the kernel with one or two scalar loads --
  // the y taps at kYsepTapY + b; each centred in its compiled extent, zeros elsewhere
  for (int stage = 0; stage < 2; ++stage) {
    float* t = taps_host + lsr::kYsepTapStage * stage;
    for (int a = 0; a < pz; ++a)
      for (int c = 0; c < px; ++c) {
        const float v = stage == 0 ? kzx_host[(pz - 1 - a) * px + (px - 1 - c)] : kzx_host[a * px + c];
        t[(c + ox) * lsr::kYsepTapGroup + (PZ - 1 - (a + oz))] = v;
      }
    for (int b = 0; b < py; ++b) t[lsr::kYsepTapY + b + oy] = stage == 0 ? ky_host[py - 1 - b] : ky_host[b];
  }
  return LSR_OK;
}

extern "C" int lsr_rl_ysep_fused_stats_f32(const float* y, int64_t y_pitch, int64_t y_plane, int init_from_y, float* x_a, float* x_b, float* x_out, int64_t Z, int64_t Y, int64_t X, const float* taps, int pz, int py, int px, const double* norm_table, float norm_full, int iters, float eps, double* stats, lsr_stream_t stream) {
  LSR_REQUIRE_PTR(y);
  LSR_REQUIRE_PTR(x_a);
  LSR_REQUIRE_PTR(x_b);
  LSR_REQUIRE_PTR(taps);
  LSR_REQUIRE_PTR(norm_table);
  LSR_REQUIRE(Z > 0 && Y > 0 && X > 0, LSR_E_SHAPE, "shape (%lld,%lld,%lld) must be positive", (long long)Z, (long long)Y,
              (long long)X);
  LSR_REQUIRE_VOLUME(Z, Y, X);
  LSR_REQUIRE(iters >= 1, LSR_E_ARG, "iters %d must be >= 1", iters);
  LSR_REQUIRE(x_a != x_b, LSR_E_ARG, "x_a and x_b must be distinct");
  if (int rc = check_taps(pz, py, px)) return rc;
  LSR_REQUIRE(lsr_rl_ysep_fused_supported(pz, py, px), LSR_E_UNSUPPORTED,
              "no fused ky (x) kzx specialisation for taps (%d,%d,%d) (up to %d z taps, %d in-plane): use "
              "lsr_correlate_zxy_padded_f32", pz, py, px, lsr::kYsepMaxPZ, lsr::kYsepMaxPYX);
  const int PZ = lsr::sep_round_taps(pz), PYX = lsr::sep_round_taps(py > px ? py : px);
  int64_t ps[4];
  if (int rc = lsr_sep_padded_shape(Y, X, pz, py, px, ps)) return rc;
  const int64_t pitch = ps[1], plane = ps[0] * ps[1];
  const int64_t origin = ps[2] * pitch + ps[3];
  LSR_REQUIRE_STRIDES(y_pitch, y_plane);
  LSR_REQUIRE(y_pitch >= pitch && y_plane >= ps[0] * y_pitch, LSR_E_SHAPE,
              "y strides (%lld,%lld) are smaller than the padded shape (%lld rows x %lld) that lsr_sep_padded_shape asks "
              "for: y must be a zero-haloed padded volume", (long long)y_pitch, (long long)y_plane, (long long)ps[0],
              (long long)pitch);
  LSR_REQUIRE(y_pitch % 4 == 0 && y_plane % 4 == 0, LSR_E_ARG,
              "pitch and plane stride of the padded y must be multiples of 4 floats");
  const int64_t lim = int64_t(1) << 29;
  LSR_REQUIRE(plane < lim && y_plane < lim && Y * X < lim && Z < lim, LSR_E_UNSUPPORTED,
              "plane strides exceed the kernel's 32-bit in-plane offsets");

  lsr::YsepArgs p{};
  p.y = y; p.y_pitch = static_cast<int>(y_pitch); p.y_plane = y_plane;
  p.Z = static_cast<int>(Z); p.Y = static_cast<int>(Y); p.X = static_cast<int>(X);
  p.taps = taps; p.eps = eps;
  p.pz = pz; p.py = py; p.px = px;
  p.norm_table = norm_table; p.norm_full = norm_full;
  // one shape: 512 threads on 32 x 128 tiles, one workgroup per CU.  (Rounds 3-4 also had 256-thread workgroups on
  // 32 x 64 tiles, two per CU: once the border normalisation stopped setting the launch time it was the slower one --
  // 4.98 against 4.63 ms per iteration on config 2, its halo is 1.72 x the tile against 1.55 x -- and is gone.)
  p.narrow = 0;
  p.tiles_x = static_cast<int>(lsr::ceil_div(X, p.narrow ? 64 : lsr::kSepWideTileX));
  p.tiles_y = static_cast<int>(lsr::ceil_div(Y, lsr::ysep_tile_rows(PZ, PYX)));
  const int64_t tiles_xy = int64_t(p.tiles_x) * p.tiles_y;
  plan_fused_split(tiles_xy, Z, PZ, &p.n_full, &p.pieces, &p.z_chunk, p.narrow ? 2 : 1);
  const int64_t blocks64 = p.n_full + (tiles_xy - p.n_full) * p.pieces;
  LSR_REQUIRE(blocks64 < (int64_t(1) << 31), LSR_E_SHAPE, "grid of %lld workgroups is too large", (long long)blocks64);
  const unsigned blocks = static_cast<unsigned>(blocks64);
  hipStream_t s = lsr::as_stream(stream);

  float* bufs[2] = {x_a + origin, x_b + origin};  // logical (0,0,0) of the two working volumes
  if (int rc = zero_stats(stats, iters, stream)) return rc;
  for (int it = 0; it < iters; ++it) {
    const bool from_y = init_from_y && it == 0;
    const bool last = it + 1 == iters && x_out != nullptr;
    p.x = from_y ? y : bufs[it & 1];
    p.pitch = static_cast<int>(from_y ? y_pitch : pitch);
    p.plane = from_y ? y_plane : plane;
    p.out = last ? x_out : bufs[(it + 1) & 1];
    p.out_pitch = static_cast<int>(last ? X : pitch);
    p.out_plane = last ? Y * X : plane;
    p.stats = stats ? stats + lsr::kRlStats * it : nullptr;
    bool ok = false;
    switch (PZ) {
      case 3: ok = lsr::launch_ysep_pz3(PYX, p, blocks, s); break;
      case 5: ok = lsr::launch_ysep_pz5(PYX, p, blocks, s); break;
      case 7: ok = lsr::launch_ysep_pz7(PYX, p, blocks, s); break;
      case 9: ok = lsr::launch_ysep_pz9(PYX, p, blocks, s); break;
      case 11: ok = lsr::launch_ysep_pz11(PYX, p, blocks, s); break;
      default: break;
    }
    LSR_REQUIRE(ok, LSR_E_UNSUPPORTED, "no fused ky (x) kzx specialisation for taps (%d,%d,%d)", pz, py, px);
    if (int rc = lsr::launch_status("lsr_rl_ysep_fused_f32")) return rc;
  }
  return LSR_OK;
}

extern "C" int lsr_rl_ysep_fused_f32(const float* y, int64_t y_pitch, int64_t y_plane, int init_from_y, float* x_a, float* x_b, float* x_out, int64_t Z, int64_t Y, int64_t X, const float* taps, int pz, int py, int px, const double* norm_table, float norm_full, int iters, float eps, lsr_stream_t stream) {
  return lsr_rl_ysep_fused_stats_f32(y, y_pitch, y_plane, init_from_y, x_a, x_b, x_out, Z, Y, X, taps, pz, py, px, norm_table, norm_full, iters, eps, nullptr, stream);
}

extern "C" int lsr_rl_dense_stats_f32(const float* y, float* x, float* ratio, int64_t Z, int64_t Y, int64_t X, const float* psf, const float* psf_flipped, int pz, int py, int px, const double* norm_table, int iters, float eps, double* stats, lsr_stream_t stream) {
  LSR_REQUIRE_PTR(y);
  LSR_REQUIRE_PTR(x);
  LSR_REQUIRE_PTR(ratio);
  LSR_REQUIRE(iters >= 0, LSR_E_ARG, "iters %d must be >= 0", iters);
  LSR_REQUIRE(ratio != x && ratio != y && x != y, LSR_E_ARG, "y, x and ratio must be distinct");
  if (int rc = zero_stats(stats, iters, stream)) return rc;
  for (int it = 0; it < iters; ++it) {
    int rc = lsr_correlate_dense_f32(x, ratio, y, Z, Y, X, psf_flipped, pz, py, px, LSR_EPI_RATIO,
                                     eps, nullptr, stream);
    if (rc) return rc;
    rc = lsr_correlate_dense_stats_f32(ratio, x, x, Z, Y, X, psf, pz, py, px, LSR_EPI_UPDATE, eps,
                                       norm_table, stats ? stats + lsr::kRlStats * it : nullptr, stream);
    if (rc) return rc;
  }
  return LSR_OK;
}

extern "C" int lsr_rl_dense_f32(const float* y, float* x, float* ratio, int64_t Z, int64_t Y, int64_t X, const float* psf, const float* psf_flipped, int pz, int py, int px, const double* norm_table, int iters, float eps, lsr_stream_t stream) {
  return lsr_rl_dense_stats_f32(y, x, ratio, Z, Y, X, psf, psf_flipped, pz, py, px, norm_table, iters, eps, nullptr, stream);
}

// ------------------------------------------------------------------------------------------
// Tuned dense path (correlate_dense.hip): padded input, taps prepared once by the host.
// ------------------------------------------------------------------------------------------

namespace {

bool dense_compiled_taps(int pz, int py, int px, int* PZ, int* PYX) {
  const int a = lsr::sep_round_taps(pz);
  const int b = lsr::sep_round_taps(py > px ? py : px);
  if (a > 11 || b > 9) return false;
  *PZ = a;
  *PYX = b;
  return true;
}

}  // namespace

extern "C" int lsr_dense_taps_count(int pz, int py, int px) {
  if (int rc = check_taps(pz, py, px)) return rc;
  int PZ, PYX;
  LSR_REQUIRE(dense_compiled_taps(pz, py, px, &PZ, &PYX), LSR_E_UNSUPPORTED,
              "the tuned dense kernel covers pz <= 11 and py, px <= 9 (got %d,%d,%d): use "
              "lsr_correlate_dense_f32",
              pz, py, px);
  return 2 * PZ * PYX * PYX;  // every tap twice: {w, w} pairs for the packed FMAs
}

extern "C" int lsr_dense_prepare_taps(const float* psf_host, int pz, int py, int px, int flip,
                                      float* taps_host) {
  LSR_REQUIRE_PTR(psf_host);
  LSR_REQUIRE_PTR(taps_host);
  const int n = lsr_dense_taps_count(pz, py, px);
  if (n < 0) return n;
  int PZ, PYX;
  dense_compiled_taps(pz, py, px, &PZ, &PYX);
  for (int i = 0; i < n; ++i) taps_host[i] = 0.0f;
  const int oz = (PZ - pz) / 2, oy = (PYX - py) / 2, ox = (PYX - px) / 2;
  for (int a = 0; a < pz; ++a)
    for (int b = 0; b < py; ++b)
      for (int c = 0; c < px; ++c) {
        // correlation tap (a, b, c); flip = the convolution H x = correlate with the reversed PSF
        const float v = flip ? psf_host[((pz - 1 - a) * py + (py - 1 - b)) * px + (px - 1 - c)]
                             : psf_host[(a * py + b) * px + c];
        const int A = a + oz, B = b + oy, C = c + ox;
        const int t = (C * PYX + B) * PZ + (PZ - 1 - A);  // [c][b][j], j = PZ-1-a
        taps_host[2 * t] = taps_host[2 * t + 1] = v;
      }
  return LSR_OK;
}

namespace {
// mode 0 / 1: lsr_correlate_dense_padded_f32 (1 = the caller's PSF has a single y tap);
// mode 2: lsr_correlate_zxy_padded_f32 (taps = the (z, x) stencil in the layout of the full PSF, ky = y taps)
int dense_padded_launch(const char* what, int mode, const float* ky,
    const float* in, int64_t in_pitch, int64_t in_plane, const float* aux, int64_t aux_pitch,
    int64_t aux_plane, float* out, int64_t out_pitch, int64_t out_plane, int64_t Z, int64_t Y,
    int64_t X, const float* taps, int pz, int py, int px, int epilogue, float eps,
    const double* norm_table, float norm_full, double* stats, lsr_stream_t stream) {
  LSR_REQUIRE_PTR(in);
  LSR_REQUIRE_PTR(out);
  LSR_REQUIRE_PTR(taps);
  LSR_REQUIRE(Z > 0 && Y > 0 && X > 0, LSR_E_SHAPE, "shape (%lld,%lld,%lld) must be positive",
              (long long)Z, (long long)Y, (long long)X);
  LSR_REQUIRE_VOLUME(Z, Y, X);
  if (int rc = check_taps(pz, py, px)) return rc;
  int PZ, PYX;
  LSR_REQUIRE(dense_compiled_taps(pz, py, px, &PZ, &PYX), LSR_E_UNSUPPORTED,
              "the tuned dense kernel covers pz <= 11 and py, px <= 9 (got %d,%d,%d)", pz, py, px);
  LSR_REQUIRE(epilogue == LSR_EPI_NONE || epilogue == LSR_EPI_RATIO || epilogue == LSR_EPI_UPDATE ||
                  epilogue == LSR_EPI_SCALE,
              LSR_E_ARG, "unknown epilogue %d", epilogue);
  const bool has_aux = epilogue == LSR_EPI_RATIO || epilogue == LSR_EPI_UPDATE;
  if (has_aux) LSR_REQUIRE_PTR(aux);
  if (epilogue == LSR_EPI_UPDATE || epilogue == LSR_EPI_SCALE) LSR_REQUIRE_PTR(norm_table);
  LSR_REQUIRE(in != out, LSR_E_ARG, "out must not alias in");
  int64_t need[4];
  lsr_sep_padded_shape(Y, X, pz, py, px, need);
  LSR_REQUIRE_STRIDES(in_pitch, in_plane);
  LSR_REQUIRE(in_pitch >= need[1] && in_plane >= need[0] * in_pitch, LSR_E_SHAPE,
              "in strides (%lld,%lld) are smaller than the padded shape (%lld rows x %lld)",
              (long long)in_pitch, (long long)in_plane, (long long)need[0], (long long)need[1]);
  LSR_REQUIRE(in_pitch % 4 == 0 && in_plane % 4 == 0, LSR_E_ARG,
              "pitch and plane stride of the padded input must be multiples of 4 floats");
  const int64_t lim = int64_t(1) << 30;
  LSR_REQUIRE(in_plane < lim && aux_plane < lim && out_plane < lim && Z < lim, LSR_E_UNSUPPORTED,
              "plane strides exceed the kernel's 32-bit in-plane offsets");
  LSR_REQUIRE(out_pitch >= X && (!has_aux || aux_pitch >= X), LSR_E_SHAPE,
              "aux/out pitch smaller than X");

  lsr::DenseArgs p{};
  p.ysep = mode == 2 ? 2 : (py == 1 ? 1 : 0);
  p.ky = ky;
  if (mode == 2) {
    LSR_REQUIRE_PTR(ky);
    LSR_REQUIRE(epilogue == LSR_EPI_RATIO || epilogue == LSR_EPI_UPDATE, LSR_E_ARG,
                "the one-launch ky (x) kzx correlation carries the RL epilogues only (got %d)", epilogue);
  }
  p.in = in; p.aux = aux; p.out = out;
  p.in_plane = in_plane; p.aux_plane = aux_plane; p.out_plane = out_plane;
  p.in_pitch = static_cast<int>(in_pitch);
  p.aux_pitch = static_cast<int>(aux_pitch);
  p.out_pitch = static_cast<int>(out_pitch);
  p.Z = static_cast<int>(Z); p.Y = static_cast<int>(Y); p.X = static_cast<int>(X);
  p.pz = pz; p.py = py; p.px = px;
  p.epilogue = epilogue; p.eps = eps;
  p.norm_full = norm_full;
  p.norm_table = norm_table;
  p.stats = epilogue == LSR_EPI_UPDATE ? stats : nullptr;
  p.taps = taps;
  p.tiles_x = static_cast<int>(lsr::ceil_div(X, lsr::kSepTileX));
  p.tiles_y = static_cast<int>(lsr::ceil_div(Y, lsr::kSepTileY));
  p.z_chunk = static_cast<int>(pick_z_chunk(Z, int64_t(p.tiles_x) * p.tiles_y, PZ));
  const int64_t blocks64 = int64_t(p.tiles_x) * p.tiles_y * lsr::ceil_div(Z, p.z_chunk);
  LSR_REQUIRE(blocks64 < (int64_t(1) << 31), LSR_E_SHAPE, "grid of %lld workgroups is too large",
              (long long)blocks64);
  const unsigned blocks = static_cast<unsigned>(blocks64);
  hipStream_t s = lsr::as_stream(stream);
  bool ok = false;
  switch (PZ) {
    case 3: ok = lsr::launch_dense_pz3(PYX, p, blocks, s); break;
    case 5: ok = lsr::launch_dense_pz5(PYX, p, blocks, s); break;
    case 7: ok = lsr::launch_dense_pz7(PYX, p, blocks, s); break;
    case 9: ok = lsr::launch_dense_pz9(PYX, p, blocks, s); break;
    case 11: ok = lsr::launch_dense_pz11(PYX, p, blocks, s); break;
    default: break;
  }
  LSR_REQUIRE(ok, LSR_E_UNSUPPORTED, "no dense specialisation for taps (%d,%d,%d)", pz, py, px);
  return lsr::launch_status(what);
}
}  // namespace

extern "C" int lsr_correlate_dense_padded_stats_f32(const float* in, int64_t in_pitch, int64_t in_plane, const float* aux, int64_t aux_pitch, int64_t aux_plane, float* out, int64_t out_pitch, int64_t out_plane, int64_t Z, int64_t Y, int64_t X, const float* taps, int pz, int py, int px, int epilogue, float eps, const double* norm_table, float norm_full, double* stats, lsr_stream_t stream) {
  return dense_padded_launch("lsr_correlate_dense_padded_f32", 0, nullptr, in, in_pitch, in_plane, aux, aux_pitch,
                             aux_plane, out, out_pitch, out_plane, Z, Y, X, taps, pz, py, px, epilogue, eps,
                             norm_table, norm_full, stats, stream);
}

extern "C" int lsr_correlate_dense_padded_f32(const float* in, int64_t in_pitch, int64_t in_plane, const float* aux, int64_t aux_pitch, int64_t aux_plane, float* out, int64_t out_pitch, int64_t out_plane, int64_t Z, int64_t Y, int64_t X, const float* taps, int pz, int py, int px, int epilogue, float eps, const double* norm_table, float norm_full, lsr_stream_t stream) {
  return lsr_correlate_dense_padded_stats_f32(in, in_pitch, in_plane, aux, aux_pitch, aux_plane, out, out_pitch, out_plane, Z, Y, X, taps, pz, py, px, epilogue, eps, norm_table, norm_full, nullptr, stream);
}

extern "C" int lsr_correlate_zxy_padded_stats_f32(const float* in, int64_t in_pitch, int64_t in_plane, const float* aux, int64_t aux_pitch, int64_t aux_plane, float* out, int64_t out_pitch, int64_t out_plane, int64_t Z, int64_t Y, int64_t X, const float* taps_zx, const float* ky, int pz, int py, int px, int epilogue, float eps, const double* norm_table, float norm_full, double* stats, lsr_stream_t stream) {
  return dense_padded_launch("lsr_correlate_zxy_padded_f32", 2, ky, in, in_pitch, in_plane, aux, aux_pitch,
                             aux_plane, out, out_pitch, out_plane, Z, Y, X, taps_zx, pz, py, px, epilogue, eps,
                             norm_table, norm_full, stats, stream);
}

extern "C" int lsr_correlate_zxy_padded_f32(const float* in, int64_t in_pitch, int64_t in_plane, const float* aux, int64_t aux_pitch, int64_t aux_plane, float* out, int64_t out_pitch, int64_t out_plane, int64_t Z, int64_t Y, int64_t X, const float* taps_zx, const float* ky, int pz, int py, int px, int epilogue, float eps, const double* norm_table, float norm_full, lsr_stream_t stream) {
  return lsr_correlate_zxy_padded_stats_f32(in, in_pitch, in_plane, aux, aux_pitch, aux_plane, out, out_pitch, out_plane, Z, Y, X, taps_zx, ky, pz, py, px, epilogue, eps, norm_table, norm_full, nullptr, stream);
}

extern "C" int lsr_rl_dense_padded_stats_f32(const float* y, int64_t y_pitch, int64_t y_plane, int init_from_y, float* x_pad, float* ratio_pad, float* x_out, int64_t Z, int64_t Y, int64_t X, const float* taps, const float* taps_flipped, int pz, int py, int px, const double* norm_table, float norm_full, int iters, float eps, double* stats, lsr_stream_t stream) {
  LSR_REQUIRE_PTR(y);
  LSR_REQUIRE_PTR(x_pad);
  LSR_REQUIRE_PTR(ratio_pad);
  LSR_REQUIRE(iters >= 1, LSR_E_ARG, "iters %d must be >= 1", iters);
  LSR_REQUIRE(ratio_pad != x_pad, LSR_E_ARG, "x_pad and ratio_pad must be distinct");
  int64_t ps[4];
  if (int rc = lsr_sep_padded_shape(Y, X, pz, py, px, ps)) return rc;
  const int64_t pitch = ps[1], plane = ps[0] * ps[1];
  const int64_t origin = ps[2] * pitch + ps[3];
  float* xl = x_pad + origin;
  float* rl = ratio_pad + origin;
  if (int rc = zero_stats(stats, iters, stream)) return rc;
  for (int it = 0; it < iters; ++it) {
    const bool from_y = init_from_y && it == 0;
    const float* xin = from_y ? y : xl;
    const int64_t xin_pitch = from_y ? y_pitch : pitch, xin_plane = from_y ? y_plane : plane;
    int rc = lsr_correlate_dense_padded_f32(xin, xin_pitch, xin_plane, y, y_pitch, y_plane, rl, pitch,
                                            plane, Z, Y, X, taps_flipped, pz, py, px, LSR_EPI_RATIO,
                                            eps, nullptr, 0.0f, stream);
    if (rc) return rc;
    const bool last = it + 1 == iters && x_out != nullptr;
    rc = lsr_correlate_dense_padded_stats_f32(rl, pitch, plane, xin, xin_pitch, xin_plane, last ? x_out : xl,
                                              last ? X : pitch, last ? Y * X : plane, Z, Y, X, taps, pz,
                                              py, px, LSR_EPI_UPDATE, eps, norm_table, norm_full,
                                              stats ? stats + lsr::kRlStats * it : nullptr, stream);
    if (rc) return rc;
  }
  return LSR_OK;
}

extern "C" int lsr_rl_dense_padded_f32(const float* y, int64_t y_pitch, int64_t y_plane, int init_from_y, float* x_pad, float* ratio_pad, float* x_out, int64_t Z, int64_t Y, int64_t X, const float* taps, const float* taps_flipped, int pz, int py, int px, const double* norm_table, float norm_full, int iters, float eps, lsr_stream_t stream) {
  return lsr_rl_dense_padded_stats_f32(y, y_pitch, y_plane, init_from_y, x_pad, ratio_pad, x_out, Z, Y, X, taps, taps_flipped, pz, py, px, norm_table, norm_full, iters, eps, nullptr, stream);
}
