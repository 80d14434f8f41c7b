// The x leg of the tracker's cross-correlation, both ways, as one kernel each (reference
// shrimpy/dynatrack/tracking.py:266-378).  With rocFFT the forward x leg was four passes over the volume --
// reflect-pad / crop to the FFT grid (_match_shape), the half-length complex transform, the real-to-complex
// post pass, the transpose to y-contiguous -- and the way back four more: transpose, complex-to-real pre
// pass, transform, peak search.  A row of the grid (2304 reals = 1152 complex) fits in LDS eight times
// over, so here a workgroup takes eight neighbouring rows (fixed z, eight y) and
//
//   lsr_rfft_rows_t_c64:   gathers them from the un-padded source with the reflect / crop index map, packs
//                          (even, odd) samples as complex, transforms in LDS (fft_lds.hpp), applies the
//                          real-to-complex post step and stores X[k] TRANSPOSED, out[z][k][y]: 64-byte runs;
//   lsr_irfft_rows_peak:   loads such a tile, applies the complex-to-real pre step, transforms back, and
//                          reduces |corr| with its fftshift-ed flat index to one candidate per workgroup --
//                          the correlation volume is never written.
//
// The same two kernels carry the x leg of a Richardson-Lucy iteration done in the Fourier domain
// (shrimpy_amd/deconvolve_fft.py: PSFs beyond the stencil kernels' extents, e.g. measured bead PSFs):
//
//   lsr_rfft_rows_zero_t_c64:  the forward kernel with the source placed at the grid's origin and ZERO padding
//                              behind it (linear, not circular, convolution); tiles of padding rows store zeros, the
//                              planes behind the source are left unwritten (the z leg takes them as zeros);
//   lsr_irfft_rows_rl_f32:     the inverse kernel with a Richardson-Lucy epilogue instead of the peak search: the
//                              rows are scaled, cropped to the volume and leave as ratio = y / (H x + eps) or as
//                              x_new = x * H^T(ratio) / H^T 1 (+ the iteration's reduction scalars); the convolved
//                              volume is never written.
//
// Unnormalised, like hipFFT: forward X[k] = sum_n x[n] exp(-2 pi i k n / X); the inverse carries a factor
// 2 X / 2 = X (irrelevant to an argmax, and the same for every voxel).
// Lengths: X a multiple of 4 with X / 2 5-smooth and <= 2048 (lsr_rfft_rows_supported); others keep rocFFT.

#include "correlate_common.hpp"
#include "fft_lds.hpp"

namespace {

using namespace lsr_fft;

constexpr int kThreads = 512;
constexpr int kRows = 8;                      // y rows per workgroup: 64-byte runs in the transposed layout
constexpr int kPerRow = kThreads / kRows;     // 64 threads share a row's butterflies
constexpr int kMaxM = 2048;                   // longest half-length: 8 * 2049 * 8 B + 8 KB of twiddles = 139 KB of LDS

struct RowsArgs {
  const float* in;        // forward: un-padded source [Zi][Yi][Xi]
  int Zi, Yi, Xi;
  float2* spec;           // [Z][XC][Y] (forward: written; inverse: read)
  int Z, Y, X, M, XC;     // FFT grid, M = X / 2, XC = M + 1
  const float2* tw_half;  // [M / 2]  exp(-2 pi i k / M)
  const float2* tw_x;     // [M + 1]  exp(-2 pi i k / X)
  Factors f;
  float* pval;            // inverse: one candidate per workgroup
  unsigned long long* pidx;
  // Richardson-Lucy epilogue (irfft_rows_rl_kernel)
  const float* aux;       // ratio epilogue: y; update epilogue: x  ((Zo, Yo, Xo) float32)
  float* out;             // (Zo, Yo, Xo) float32; may be aux (every voxel is read and written by one thread)
  int Zo, Yo, Xo;         // the volume: the grid's first Zo x Yo x Xo points
  float scale, eps;       // 1 / (Z Y X) (the transforms are unnormalised); eps of the ratio
  int pz, py, px;         // PSF extents (geometry of the normalisation table)
  const double* norm_table;   // (pz+1)(py+1)(px+1) prefix sums of the PSF (border voxels)
  float norm_full;        // sum of all taps (interior voxels)
  double* stats;          // update epilogue: three running sums of the iteration (correlate_common.hpp) or NULL
};

__device__ __forceinline__ int reflect_index(int i, int n) {
  if (n == 1) return 0;
  const int period = 2 * (n - 1);
  i = i % period;
  if (i < 0) i += period;
  return i < n ? i : period - i;
}
// _match_shape (tracking.py:266-306): reflect-pad (left = d // 2) or centre-crop (start = d // 2), per axis
__device__ __forceinline__ int match_index(int o, int ni, int no) {
  if (no > ni) return reflect_index(o - (no - ni) / 2, ni);
  return o + (ni - no) / 2;
}

struct Tile {
  float2* buf;       // [kRows][pitch]
  float2* tw;        // [M / 2]
  int pitch;
};

__device__ __forceinline__ Tile carve(float2* smem, int M) {
  Tile t;
  t.pitch = M + 1;
  t.buf = smem;
  t.tw = smem + kRows * t.pitch;
  return t;
}

// exp(-2 pi i k / M) from the half table: w^(k + M/2) = -w^k
__device__ __forceinline__ float2 tw_m(const float2* tw, int half, int i) {
  const bool hi = i >= half;
  const float2 v = tw[hi ? i - half : i];
  return hi ? float2{-v.x, -v.y} : v;
}

// Workgroups b, b + 8, ... land on the same XCD (round-robin dispatch).  Neighbouring y tiles share every 128-byte
// line of the transposed spectrum (64-byte runs each): each XCD takes a CONTIGUOUS run of the tile order, so that the
// two halves of a line meet in one L2 instead of being fetched from HBM by two.  Grid = 8 * ceil(tiles / 8).
__device__ __forceinline__ int xcd_tile(int n_tiles, int block) {
  const int per = (n_tiles + 7) >> 3;
  const int t = (block & 7) * per + (block >> 3);
  return t < n_tiles ? t : -1;
}
__host__ __device__ inline unsigned xcd_grid(int64_t n_tiles) { return static_cast<unsigned>(8 * ((n_tiles + 7) / 8)); }

// A tile of zeros in the transposed spectrum (rows y0 .. y0 + nrows - 1 of plane z).
__device__ __forceinline__ void store_zero_tile(const RowsArgs& p, int z, int y0, int nrows, int tid) {
  const int r = tid & (kRows - 1), k0 = tid / kRows;
  if (r < nrows) {
    float2* out = p.spec + static_cast<int64_t>(z) * p.XC * p.Y + y0 + r;
    for (int k = k0; k <= p.M; k += kThreads / kRows) out[static_cast<int64_t>(k) * p.Y] = float2{0.0f, 0.0f};
  }
}

// real-to-complex post step of the transformed tile, stored transposed: X[k] = E[k] + w_X^k O[k],
//   E = (Z[k] + conj(Z[M - k])) / 2,  O = -i (Z[k] - conj(Z[M - k])) / 2,  Z[M] = Z[0]
__device__ __forceinline__ void r2c_post_store(const RowsArgs& p, const Tile& t, int z, int y0, int nrows, int tid) {
  const int M = p.M;
  const int r = tid & (kRows - 1), k0 = tid / kRows;
  if (r < nrows) {
    const float2* row = t.buf + r * t.pitch;
    float2* out = p.spec + static_cast<int64_t>(z) * p.XC * p.Y + y0 + r;
    const float2* twx = p.tw_x;
    const int64_t ystride = p.Y;
    batched_loop<8>(k0, M + 1, kThreads / kRows, [twx](int k) { return twx[k]; },
                    [row, out, M, ystride](int k, float2 w) {
                      const float2 a = row[k == M ? 0 : k], b = cconj(row[k == 0 ? 0 : M - k]);
                      const float2 e = float2{0.5f * (a.x + b.x), 0.5f * (a.y + b.y)};
                      const float2 o = mul_mi(float2{0.5f * (a.x - b.x), 0.5f * (a.y - b.y)});
                      out[static_cast<int64_t>(k) * ystride] = cadd(e, cmul(w, o));
                    });
  }
}

// ZERO: the source sits at the grid's origin, zeros behind it (no index map)
template <bool ZERO>
__global__ __launch_bounds__(kThreads) void rfft_rows_kernel(RowsArgs p) {
  extern __shared__ float2 smem[];
  const int M = p.M, half = M / 2;
  const Tile t = carve(smem, M);
  const int tid = threadIdx.x;
  const int tiles_y = (p.Y + kRows - 1) / kRows;
  int z, y0;
  if constexpr (ZERO) {
    // tiles that hold source rows first, in XCD runs; then the tiles of pure padding (they only store zeros), so
    // that no XCD's run is mostly padding
    const int ty_src = (p.Yi + kRows - 1) / kRows, n_src = p.Zi * ty_src;
    const int first_pad = static_cast<int>(xcd_grid(n_src));
    if (static_cast<int>(blockIdx.x) < first_pad) {
      const int tile = xcd_tile(n_src, blockIdx.x);
      if (tile < 0) return;
      z = tile / ty_src;
      y0 = (tile - z * ty_src) * kRows;
    } else {
      const int q = static_cast<int>(blockIdx.x) - first_pad, beside = tiles_y - ty_src, n_beside = p.Zi * beside;
      if (q >= n_beside) return;   // (the planes behind the source are not written: lsr_spectrum_multiply_z_c64's z_valid)
      z = q / beside;
      y0 = (ty_src + q - z * beside) * kRows;
    }
  } else {
    const int tile = xcd_tile(p.Z * tiles_y, blockIdx.x);
    if (tile < 0) return;
    z = tile / tiles_y;
    y0 = (tile - z * tiles_y) * kRows;
  }
  const int nrows = min(kRows, p.Y - y0);

  if constexpr (ZERO) {
    if (z >= p.Zi || y0 >= p.Yi) {   // (uniform) a tile of padding: its spectrum is zero
      store_zero_tile(p, z, y0, nrows, tid);
      return;
    }
  }
  for (int k = tid; k < half; k += kThreads) t.tw[k] = p.tw_half[k];
  if constexpr (ZERO) {
    const int r = tid / kPerRow, lane = tid & (kPerRow - 1);
    float2* row = t.buf + r * t.pitch;
    if (r < nrows && y0 + r < p.Yi) {
      const float* src = p.in + (static_cast<int64_t>(z) * p.Yi + y0 + r) * p.Xi;
      const int last = p.Xi - 1, n_src = (p.Xi + 1) / 2;      // packed pairs that touch the source
      batched_loop<8>(lane, n_src, kPerRow,
                      [src, last](int m) { return float2{src[2 * m], src[min(2 * m + 1, last)]}; },
                      [row, last](int m, float2 v) { row[m] = float2{v.x, 2 * m + 1 <= last ? v.y : 0.0f}; });
      for (int m = lane + ((n_src - lane + kPerRow - 1) / kPerRow) * kPerRow; m < M; m += kPerRow) row[m] = float2{0.0f, 0.0f};
    } else {
      for (int m = lane; m < M; m += kPerRow) row[m] = float2{0.0f, 0.0f};
    }
  } else {  // gather: row r of the tile <- source row (match(z), match(y0 + r)), columns through match(x); (even, odd) packed
    const int r = tid / kPerRow, lane = tid & (kPerRow - 1);
    float2* row = t.buf + r * t.pitch;
    if (r < nrows) {
      const float* src = p.in + (static_cast<int64_t>(match_index(z, p.Zi, p.Z)) * p.Yi + match_index(y0 + r, p.Yi, p.Y)) * p.Xi;
      const int shift = p.X > p.Xi ? -((p.X - p.Xi) / 2) : (p.Xi - p.X) / 2;   // source column of grid column 0 (before reflection)
      const int xi = p.Xi;
      batched_loop<8>(lane, M, kPerRow,
                      [src, shift, xi](int m) {
                        const int x0 = 2 * m + shift, x1 = x0 + 1;
                        const bool inside = x0 >= 0 && x1 < xi;
                        return float2{src[inside ? x0 : reflect_index(x0, xi)], src[inside ? x1 : reflect_index(x1, xi)]};
                      },
                      [row](int m, float2 v) { row[m] = v; });
    } else {
      for (int m = lane; m < M; m += kPerRow) row[m] = float2{0.0f, 0.0f};
    }
  }
  __syncthreads();

  const float2* twl = t.tw;
  transform<kMaxM, kPerRow>(t.buf + (tid / kPerRow) * t.pitch, M, p.f, [twl, half](int i) { return tw_m(twl, half, i); },
                            tid & (kPerRow - 1));

  __syncthreads();            // the post step below reads rows across wavefronts
  r2c_post_store(p, t, z, y0, nrows, tid);
}

// Tile of the spectrum for the inverse kernels: X[k], k = 0 .. M, of eight neighbouring y (64-byte runs); rows at or
// past `y_end` are zeros.
__device__ __forceinline__ void load_spectrum_tile(const RowsArgs& p, const Tile& t, int z, int y0, int y_end, int tid) {
  const int M = p.M;
  const int r = tid & (kRows - 1), k0 = tid / kRows;
  float2* row = t.buf + r * t.pitch;
  if (y0 + r < y_end) {
    const float2* in = p.spec + static_cast<int64_t>(z) * p.XC * p.Y + y0 + r;
    const int64_t ystride = p.Y;
    batched_loop<10>(k0, M + 1, kThreads / kRows, [in, ystride](int k) { return in[static_cast<int64_t>(k) * ystride]; },
                     [row](int k, float2 v) { row[k] = v; });
  } else {
    for (int k = k0; k <= M; k += kThreads / kRows) row[k] = float2{0.0f, 0.0f};
  }
}

// complex-to-real pre step, pairs (m, M - m) by one thread; conjugated on the way for the conj-FFT-conj inverse:
//   Zt[m] = (X[m] + conj(X[M - m])) + i conj(w_X^m) (X[m] - conj(X[M - m]))        (= 2 x the packed signal's spectrum)
__device__ __forceinline__ void c2r_pre_step(const RowsArgs& p, const Tile& t, int tid) {
  const int M = p.M, half = M / 2;
  const int r = tid / kPerRow, lane = tid & (kPerRow - 1);
  float2* row = t.buf + r * t.pitch;
  const float2* twx = p.tw_x;
  struct Pair { float2 a, b; };
  batched_loop<5>(lane, half + 1, kPerRow, [twx, M](int m) { return Pair{twx[m], twx[M - m]}; },
                  [row, M](int m, Pair w) {
                    const int mm = M - m;                       // partner; m == 0 pairs with X[M], m == M / 2 with itself
                    const float2 xa = row[m], xb = row[mm];
                    const float2 wa = cconj(w.a), wb = cconj(w.b);
                    const float2 za = cadd(cadd(xa, cconj(xb)), mul_i(cmul(wa, csub(xa, cconj(xb)))));
                    const float2 zb = cadd(cadd(xb, cconj(xa)), mul_i(cmul(wb, csub(xb, cconj(xa)))));
                    row[m] = cconj(za);
                    if (m != 0 && mm != m) row[mm] = cconj(zb);
                  });
}

__device__ __forceinline__ void peak_merge(float& v, unsigned long long& i, float v2, unsigned long long i2) {
  if (v2 > v || (v2 == v && i2 < i)) { v = v2; i = i2; }
}

__global__ __launch_bounds__(kThreads) void irfft_rows_peak_kernel(RowsArgs p) {
  extern __shared__ float2 smem[];
  const int M = p.M, half = M / 2;
  const Tile t = carve(smem, M);
  // the reduction at the end reuses the tile's memory (6 KB of static LDS would cost the second workgroup per CU)
  unsigned long long* const s_i = reinterpret_cast<unsigned long long*>(smem);
  float* const s_v = reinterpret_cast<float*>(smem + kThreads);
  const int tid = threadIdx.x;
  const int tiles_y = (p.Y + kRows - 1) / kRows;
  const int z = blockIdx.x / tiles_y, y0 = (blockIdx.x - z * tiles_y) * kRows;
  const int nrows = min(kRows, p.Y - y0);

  for (int k = tid; k < half; k += kThreads) t.tw[k] = p.tw_half[k];
  load_spectrum_tile(p, t, z, y0, p.Y, tid);
  __syncthreads();
  c2r_pre_step(p, t, tid);
  __syncthreads();

  const float2* twl = t.tw;
  transform<kMaxM, kPerRow>(t.buf + (tid / kPerRow) * t.pitch, M, p.f, [twl, half](int i) { return tw_m(twl, half, i); },
                            tid & (kPerRow - 1));

  // conj(FFT(conj(Zt)))[m] = x[2m] + i x[2m + 1]; the candidate of every value is its fftshift-ed flat index
  float best = -1.0f;
  unsigned long long best_i = ~0ull;
  {
    const int r = tid / kPerRow, lane = tid & (kPerRow - 1);
    if (r < nrows) {
      const float2* row = t.buf + r * t.pitch;
      const int y = y0 + r;
      const int zs = (z + p.Z / 2) % p.Z, ys = (y + p.Y / 2) % p.Y;
      const unsigned long long base = (static_cast<unsigned long long>(zs) * p.Y + ys) * p.X;
      for (int m = lane; m < M; m += kPerRow) {
        const float2 v = row[m];
        const float a = fabsf(v.x), b = fabsf(v.y);      // conj does not change the moduli
        if (a >= best) peak_merge(best, best_i, a, base + static_cast<unsigned>((2 * m + p.X / 2) % p.X));
        if (b >= best) peak_merge(best, best_i, b, base + static_cast<unsigned>((2 * m + 1 + p.X / 2) % p.X));
      }
    }
  }
  __syncthreads();            // every row has been scanned: the tile's memory is free
  s_v[tid] = best;
  s_i[tid] = best_i;
  __syncthreads();
  for (int w = kThreads / 2; w > 0; w >>= 1) {
    if (tid < w) peak_merge(s_v[tid], s_i[tid], s_v[tid + w], s_i[tid + w]);
    __syncthreads();
  }
  if (tid == 0) {
    p.pval[blockIdx.x] = s_v[0];
    p.pidx[blockIdx.x] = s_i[0];
  }
}

// H^T 1 at a border voxel from the PSF's prefix sums (the taps that stay inside the volume): as dense_norm of
// correlate_dense.hip
__device__ __forceinline__ float rl_border_norm(const RowsArgs& p, int z, int y, int x) {
  const double* P = p.norm_table;
  const int cz = p.pz / 2, cy = p.py / 2, cx = p.px / 2;
  const int a0 = max(0, cz - z), a1 = min(p.pz, p.Zo - z + cz);
  const int b0 = max(0, cy - y), b1 = min(p.py, p.Yo - y + cy);
  const int c0 = max(0, cx - x), c1 = min(p.px, p.Xo - x + cx);
  const int sb = p.px + 1, sa = (p.py + 1) * sb;
  return static_cast<float>(((P[a1 * sa + b1 * sb + c1] - P[a0 * sa + b1 * sb + c1]) -
                             (P[a1 * sa + b0 * sb + c1] - P[a0 * sa + b0 * sb + c1])) -
                            ((P[a1 * sa + b1 * sb + c0] - P[a0 * sa + b1 * sb + c0]) -
                             (P[a1 * sa + b0 * sb + c0] - P[a0 * sa + b0 * sb + c0])));
}

// The inverse x leg with a Richardson-Lucy epilogue.  v = scale * (the complex-to-real inverse of the tile's rows),
// i.e. H x or H^T(ratio) on the volume's first Zo x Yo x Xo grid points:
//   EPI = LSR_EPI_RATIO:   out = aux / (max(v, 0) + eps)          (aux = y; a float32 transform can leave a tiny
//                                                                  negative where H x is 0: clamped, H x >= 0)
//   EPI = LSR_EPI_UPDATE:  out = aux * v / H^T 1                  (aux = x; STATS: the iteration's three sums)
// Tiles past the volume (the padding planes and rows of the grid) are not even loaded.
// CHAIN: the epilogue's output row goes straight back through the FORWARD x leg (zero padding, transform, post step)
// and replaces the tile of the spectrum it came from: what the next convolution of the iteration starts from.  The
// ratio then never exists in memory at all (p.out may be NULL), x_new is stored once and not read again by a separate
// forward launch; the tiles of pure padding, which the unchained kernel does not even launch, store zeros.
template <int EPI, bool STATS, bool CHAIN = false>
__global__ __launch_bounds__(kThreads) void irfft_rows_rl_kernel(RowsArgs p) {
  extern __shared__ float2 smem[];
  const int M = p.M, half = M / 2;
  const Tile t = carve(smem, M);
  const int tid = threadIdx.x;
  // only the tiles that hold rows of the volume are transformed (the padding planes and rows are never read)
  const int ty_out = (p.Yo + kRows - 1) / kRows, n_out = p.Zo * ty_out;
  int z, y0;
  if (!CHAIN || static_cast<int>(blockIdx.x) < static_cast<int>(xcd_grid(n_out))) {
    const int tile = xcd_tile(n_out, blockIdx.x);
    if (tile < 0) return;
    z = tile / ty_out;
    y0 = (tile - z * ty_out) * kRows;
  } else {   // CHAIN: a tile of padding beside or behind the volume (dealt as in rfft_rows_kernel<true>)
    const int tiles_y = (p.Y + kRows - 1) / kRows;
    const int q = static_cast<int>(blockIdx.x) - static_cast<int>(xcd_grid(n_out)), beside = tiles_y - ty_out;
    const int n_beside = p.Zo * beside;
    if (q < n_beside) {           // (the planes behind the volume are not written: z_valid of the z leg)
      z = q / beside;
      y0 = (ty_out + q - z * beside) * kRows;
      store_zero_tile(p, z, y0, min(kRows, p.Y - y0), tid);
    }
    return;
  }
  const int nrows = min(kRows, p.Yo - y0);

  for (int k = tid; k < half; k += kThreads) t.tw[k] = p.tw_half[k];
  load_spectrum_tile(p, t, z, y0, p.Y, tid);
  __syncthreads();
  c2r_pre_step(p, t, tid);
  __syncthreads();

  // UPDATE: H^T 1 of the border voxels this lane will write, looked up BEFORE the transform so that the table's loads
  // (eight dependent float64 reads per voxel) are long back when the epilogue wants them.  A lane's m = lane + 64 k
  // meets the left border (x < cx <= 64) only at k = 0 and the right border (the last cx <= 64 columns: at most 33
  // values of m) at most once; every other voxel of the row shares one value.  (Looked up inside the epilogue loop,
  // the 19 % of wave-iterations that touch a border each waited a microsecond: 3.75 against 2.98 ms for the ratio form.)
  float n_left[2] = {1.0f, 1.0f}, n_right[2] = {1.0f, 1.0f}, n_row = 1.0f;
  int m_right = -1;
  if constexpr (EPI == LSR_EPI_UPDATE) {
    const int r = tid / kPerRow, lane = tid & (kPerRow - 1);
    if (r < nrows) {
      const int y = y0 + r;
      const int cz = p.pz / 2, cy = p.py / 2, cx = p.px / 2;
      const bool zy_inside = z >= cz && z < p.Zo - cz && y >= cy && y < p.Yo - cy;
      n_row = zy_inside ? p.norm_full : rl_border_norm(p, z, y, min(cx, p.Xo - 1));
      const int m_lo = max(p.Xo - cx, 0) >> 1, m_hi = (p.Xo - 1) >> 1;
      m_right = m_lo + ((lane - m_lo) & (kPerRow - 1));
      if (m_right > m_hi) m_right = -1;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int xl = 2 * lane + h, xr = 2 * m_right + h;
        n_left[h] = (xl < cx && xl < p.Xo) ? rl_border_norm(p, z, y, xl) : n_row;
        n_right[h] = (m_right >= 0 && xr >= p.Xo - cx && xr < p.Xo) ? rl_border_norm(p, z, y, xr) : n_row;
      }
    }
  }

  const float2* twl = t.tw;
  transform<kMaxM, kPerRow>(t.buf + (tid / kPerRow) * t.pitch, M, p.f, [twl, half](int i) { return tw_m(twl, half, i); },
                            tid & (kPerRow - 1));

  // conj(FFT(conj(Zt)))[m] = v[2m] + i v[2m + 1]: the row holds its conjugate
  lsr::RlStats stats;
  {
    const int r = tid / kPerRow, lane = tid & (kPerRow - 1);
    if (r < nrows) {
      float2* row = t.buf + r * t.pitch;
      const int y = y0 + r;
      const int64_t base = (static_cast<int64_t>(z) * p.Yo + y) * p.Xo;
      const float* aux = p.aux + base;
      float* out = p.out == nullptr ? nullptr : p.out + base;
      const int cx = p.px / 2;
      const int last = p.Xo - 1;
      batched_loop<6>(lane, (p.Xo + 1) / 2, kPerRow,
                      [aux, last](int m) { return float2{aux[2 * m], aux[min(2 * m + 1, last)]}; },
                      [&](int m, float2 av) {
        const float2 c = row[m];
        const float v[2] = {c.x * p.scale, -c.y * p.scale};
        const float a2[2] = {av.x, av.y};
        float res[2] = {0.0f, 0.0f};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int x = 2 * m + h;
          if (x < p.Xo) {
            const float a = a2[h];
            if constexpr (EPI == LSR_EPI_RATIO) {
              res[h] = a / (fmaxf(v[h], 0.0f) + p.eps);
              if (!CHAIN || out != nullptr) out[x] = res[h];
            } else {
              const float nrm = (m == lane && x < cx) ? n_left[h] : ((m == m_right && x >= p.Xo - cx) ? n_right[h] : n_row);
              const float xu = a * v[h];
              res[h] = xu / nrm;
              out[x] = res[h];
              if constexpr (STATS) stats.add(a, xu, res[h]);
            }
          }
        }
        if constexpr (CHAIN) row[m] = float2{res[0], res[1]};     // the packed (even, odd) samples of the forward leg
      });
      if constexpr (CHAIN) {   // zero padding behind the row
        const int n_src = (p.Xo + 1) / 2;
        for (int m = lane + ((n_src - lane + kPerRow - 1) / kPerRow) * kPerRow; m < M; m += kPerRow) row[m] = float2{0.0f, 0.0f};
      }
    } else if constexpr (CHAIN) {   // a row of the tile past the volume: padding
      float2* row = t.buf + r * t.pitch;
      for (int m = lane; m < M; m += kPerRow) row[m] = float2{0.0f, 0.0f};
    }
  }
  if constexpr (CHAIN) {
    __syncthreads();
    transform<kMaxM, kPerRow>(t.buf + (tid / kPerRow) * t.pitch, M, p.f, [twl, half](int i) { return tw_m(twl, half, i); },
                              tid & (kPerRow - 1));
    __syncthreads();
    r2c_post_store(p, t, z, y0, min(kRows, p.Y - y0), tid);
  }
  if constexpr (STATS) lsr::rl_stats_flush<kThreads / 64>(stats, reinterpret_cast<float*>(smem), p.stats);
}

__global__ __launch_bounds__(256) void rows_peak_final_kernel(const float* __restrict__ pval,
                                                             const unsigned long long* __restrict__ pidx, int64_t nb,
                                                             long long* __restrict__ out) {
  __shared__ float s_v[256];
  __shared__ unsigned long long s_i[256];
  float best = -1.0f;
  unsigned long long best_i = ~0ull;
  for (int64_t i = threadIdx.x; i < nb; i += 256) peak_merge(best, best_i, pval[i], pidx[i]);
  s_v[threadIdx.x] = best;
  s_i[threadIdx.x] = best_i;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (static_cast<int>(threadIdx.x) < w) peak_merge(s_v[threadIdx.x], s_i[threadIdx.x], s_v[threadIdx.x + w], s_i[threadIdx.x + w]);
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = static_cast<long long>(s_i[0]);
}

int fill(RowsArgs& p, int64_t Z, int64_t Y, int64_t X, const float* tw_half, const float* tw_x) {
  LSR_REQUIRE(Z > 0 && Y > 0 && X > 0, LSR_E_SHAPE, "grid (%lld,%lld,%lld) must be positive", (long long)Z, (long long)Y,
              (long long)X);
  LSR_REQUIRE_VOLUME(Z, Y, X);
  LSR_REQUIRE(lsr_rfft_rows_supported(X), LSR_E_UNSUPPORTED,
              "row length %lld: a multiple of 4 whose half is 5-smooth and at most %d", (long long)X, kMaxM);
  LSR_REQUIRE_PTR(tw_half);
  LSR_REQUIRE_PTR(tw_x);
  const int64_t lim = int64_t(1) << 30;
  LSR_REQUIRE(Z < lim && Y < lim, LSR_E_UNSUPPORTED, "a dimension exceeds 2^30");
  p.Z = static_cast<int>(Z); p.Y = static_cast<int>(Y); p.X = static_cast<int>(X);
  p.M = p.X / 2; p.XC = p.M + 1;
  p.tw_half = reinterpret_cast<const float2*>(tw_half);
  p.tw_x = reinterpret_cast<const float2*>(tw_x);
  LSR_REQUIRE(factorize(p.M, &p.f), LSR_E_UNSUPPORTED, "row length %lld has too many factors", (long long)X);
  LSR_REQUIRE(Z * lsr::ceil_div(Y, kRows) < (int64_t(1) << 31), LSR_E_SHAPE, "grid of workgroups is too large");
  return LSR_OK;
}

// the tile and the half twiddle table; at least the 6 KB the final reduction of the inverse kernel lays over it
size_t lds_bytes(int M) {
  const size_t tile = (static_cast<size_t>(kRows) * (M + 1) + M / 2) * sizeof(float2);
  const size_t reduction = static_cast<size_t>(kThreads) * (sizeof(unsigned long long) + sizeof(float));
  return tile > reduction ? tile : reduction;
}

template <typename K>
int allow_lds(K kernel, std::atomic<uint64_t>& done, const char* what) {
  return lsr::allow_dynamic_lds(reinterpret_cast<const void*>(kernel), static_cast<int>(lds_bytes(kMaxM)), done, what);
}

}  // namespace

// 1 when the row kernels take a transform of length n along x (n % 4 == 0, n / 2 5-smooth and <= 2048)
extern "C" int lsr_rfft_rows_supported(int64_t n) {
  if (n < 8 || n % 4 != 0 || n / 2 > kMaxM) return 0;
  int64_t m = n / 2;
  for (int f : {2, 3, 5})
    while (m % f == 0) m /= f;
  return m == 1 && lsr::lds_fits(lds_bytes(static_cast<int>(n / 2)));
}

extern "C" int64_t lsr_rfft_rows_scratch_bytes(int64_t Z, int64_t Y) {
  if (!lsr::volume_in_range(Z, Y, 1)) return -1;
  return Z * lsr::ceil_div(Y, kRows) * 16;   // one (value, index) candidate per workgroup of lsr_irfft_rows_peak
}

extern "C" int lsr_rfft_rows_t_c64(const float* in, int64_t Zi, int64_t Yi, int64_t Xi, float* spec, int64_t Z, int64_t Y,
                                   int64_t X, const float* tw_half, const float* tw_x, lsr_stream_t stream) {
  LSR_REQUIRE_PTR(in);
  LSR_REQUIRE_PTR(spec);
  LSR_REQUIRE(Zi > 0 && Yi > 0 && Xi > 0, LSR_E_SHAPE, "source shape (%lld,%lld,%lld) must be positive", (long long)Zi,
              (long long)Yi, (long long)Xi);
  LSR_REQUIRE_VOLUME(Zi, Yi, Xi);
  const int64_t lim = int64_t(1) << 30;
  LSR_REQUIRE(Zi < lim && Yi < lim && Xi < lim, LSR_E_UNSUPPORTED, "a dimension exceeds 2^30");
  // F.pad "reflect" needs pad < size on every padded axis (torch raises otherwise)
  LSR_REQUIRE((Z <= Zi || Z - Zi < 2 * Zi - 1) && (Y <= Yi || Y - Yi < 2 * Yi - 1) && (X <= Xi || X - Xi < 2 * Xi - 1),
              LSR_E_ARG, "reflect padding from (%lld,%lld,%lld) to (%lld,%lld,%lld) exceeds the source size", (long long)Zi,
              (long long)Yi, (long long)Xi, (long long)Z, (long long)Y, (long long)X);
  RowsArgs p{};
  if (int rc = fill(p, Z, Y, X, tw_half, tw_x)) return rc;
  p.in = in;
  p.Zi = static_cast<int>(Zi); p.Yi = static_cast<int>(Yi); p.Xi = static_cast<int>(Xi);
  p.spec = reinterpret_cast<float2*>(spec);
  static std::atomic<uint64_t> lds_allowed{0};
  if (int rc = allow_lds(rfft_rows_kernel<false>, lds_allowed, "lsr_rfft_rows_t_c64")) return rc;
  const unsigned blocks = xcd_grid(Z * lsr::ceil_div(Y, kRows));
  hipLaunchKernelGGL(rfft_rows_kernel<false>, dim3(blocks), dim3(kThreads), lds_bytes(p.M), lsr::as_stream(stream), p);
  return lsr::launch_status("lsr_rfft_rows_t_c64");
}

extern "C" int lsr_irfft_rows_peak(const float* spec, int64_t Z, int64_t Y, int64_t X, const float* tw_half,
                                   const float* tw_x, long long* out_index, void* scratch, lsr_stream_t stream) {
  LSR_REQUIRE_PTR(spec);
  LSR_REQUIRE_PTR(out_index);
  LSR_REQUIRE_PTR(scratch);
  RowsArgs p{};
  if (int rc = fill(p, Z, Y, X, tw_half, tw_x)) return rc;
  p.spec = const_cast<float2*>(reinterpret_cast<const float2*>(spec));
  const int64_t nb = Z * lsr::ceil_div(Y, kRows);
  p.pidx = static_cast<unsigned long long*>(scratch);
  p.pval = reinterpret_cast<float*>(static_cast<char*>(scratch) + 8 * nb);
  static std::atomic<uint64_t> lds_allowed{0};
  if (int rc = allow_lds(irfft_rows_peak_kernel, lds_allowed, "lsr_irfft_rows_peak")) return rc;
  hipStream_t s = lsr::as_stream(stream);
  hipLaunchKernelGGL(irfft_rows_peak_kernel, dim3(static_cast<unsigned>(nb)), dim3(kThreads), lds_bytes(p.M), s, p);
  hipLaunchKernelGGL(rows_peak_final_kernel, dim3(1), dim3(256), 0, s, p.pval, p.pidx, nb, out_index);
  return lsr::launch_status("lsr_irfft_rows_peak");
}

extern "C" int lsr_rfft_rows_zero_t_c64(const float* in, int64_t Zi, int64_t Yi, int64_t Xi, float* spec, int64_t Z,
                                        int64_t Y, int64_t X, const float* tw_half, const float* tw_x,
                                        lsr_stream_t stream) {
  LSR_REQUIRE_PTR(in);
  LSR_REQUIRE_PTR(spec);
  LSR_REQUIRE(Zi > 0 && Yi > 0 && Xi > 0, LSR_E_SHAPE, "source shape (%lld,%lld,%lld) must be positive", (long long)Zi,
              (long long)Yi, (long long)Xi);
  LSR_REQUIRE_VOLUME(Zi, Yi, Xi);
  LSR_REQUIRE(Zi <= Z && Yi <= Y && Xi <= X, LSR_E_SHAPE, "the source (%lld,%lld,%lld) must fit the grid (%lld,%lld,%lld)",
              (long long)Zi, (long long)Yi, (long long)Xi, (long long)Z, (long long)Y, (long long)X);
  RowsArgs p{};
  if (int rc = fill(p, Z, Y, X, tw_half, tw_x)) return rc;
  p.in = in;
  p.Zi = static_cast<int>(Zi); p.Yi = static_cast<int>(Yi); p.Xi = static_cast<int>(Xi);
  p.spec = reinterpret_cast<float2*>(spec);
  static std::atomic<uint64_t> lds_allowed{0};
  if (int rc = allow_lds(rfft_rows_kernel<true>, lds_allowed, "lsr_rfft_rows_zero_t_c64")) return rc;
  // tiles with source rows, then the tiles of padding rows beside them; nothing for the planes behind the source
  const int64_t src_tiles = Zi * lsr::ceil_div(Yi, kRows), beside = Zi * (lsr::ceil_div(Y, kRows) - lsr::ceil_div(Yi, kRows));
  const unsigned blocks = xcd_grid(src_tiles) + static_cast<unsigned>(beside);
  hipLaunchKernelGGL(rfft_rows_kernel<true>, dim3(blocks), dim3(kThreads), lds_bytes(p.M), lsr::as_stream(stream), p);
  return lsr::launch_status("lsr_rfft_rows_zero_t_c64");
}

extern "C" int lsr_irfft_rows_rl_f32(const float* spec, int64_t Z, int64_t Y, int64_t X, const float* tw_half,
                                     const float* tw_x, int epilogue, const float* aux, float* out, int64_t Zo, int64_t Yo,
                                     int64_t Xo, float scale, float eps, int pz, int py, int px, const double* norm_table,
                                     float norm_full, double* stats, lsr_stream_t stream) {
  LSR_REQUIRE_PTR(spec);
  LSR_REQUIRE_PTR(aux);
  LSR_REQUIRE_PTR(out);
  LSR_REQUIRE(epilogue == LSR_EPI_RATIO || epilogue == LSR_EPI_UPDATE, LSR_E_ARG,
              "epilogue must be LSR_EPI_RATIO or LSR_EPI_UPDATE, got %d", epilogue);
  LSR_REQUIRE(Zo > 0 && Yo > 0 && Xo > 0 && Zo <= Z && Yo <= Y && Xo <= X, LSR_E_SHAPE,
              "the volume (%lld,%lld,%lld) must be positive and fit the grid (%lld,%lld,%lld)", (long long)Zo, (long long)Yo,
              (long long)Xo, (long long)Z, (long long)Y, (long long)X);
  LSR_REQUIRE(scale > 0.0f && eps > 0.0f, LSR_E_ARG, "scale and eps must be positive");
  RowsArgs p{};
  if (int rc = fill(p, Z, Y, X, tw_half, tw_x)) return rc;
  if (epilogue == LSR_EPI_UPDATE) {
    LSR_REQUIRE_PTR(norm_table);
    LSR_REQUIRE(pz > 0 && py > 0 && px > 0 && pz % 2 == 1 && py % 2 == 1 && px % 2 == 1 && pz < 4096 && py < 4096 &&
                    px < 4096, LSR_E_ARG, "PSF extents (%d,%d,%d) must be odd and positive", pz, py, px);
    LSR_REQUIRE(norm_full > 0.0f, LSR_E_ARG, "norm_full must be positive");
  }
  p.spec = const_cast<float2*>(reinterpret_cast<const float2*>(spec));
  p.aux = aux; p.out = out;
  p.Zo = static_cast<int>(Zo); p.Yo = static_cast<int>(Yo); p.Xo = static_cast<int>(Xo);
  p.scale = scale; p.eps = eps;
  p.pz = pz; p.py = py; p.px = px;
  p.norm_table = norm_table; p.norm_full = norm_full; p.stats = stats;
  const unsigned blocks = xcd_grid(Zo * lsr::ceil_div(Yo, kRows));
  hipStream_t s = lsr::as_stream(stream);
  static std::atomic<uint64_t> a0{0}, a1{0}, a2{0};
  if (epilogue == LSR_EPI_RATIO) {
    if (int rc = allow_lds(irfft_rows_rl_kernel<LSR_EPI_RATIO, false>, a0, "lsr_irfft_rows_rl_f32")) return rc;
    hipLaunchKernelGGL((irfft_rows_rl_kernel<LSR_EPI_RATIO, false>), dim3(blocks), dim3(kThreads), lds_bytes(p.M), s, p);
  } else if (stats == nullptr) {
    if (int rc = allow_lds(irfft_rows_rl_kernel<LSR_EPI_UPDATE, false>, a1, "lsr_irfft_rows_rl_f32")) return rc;
    hipLaunchKernelGGL((irfft_rows_rl_kernel<LSR_EPI_UPDATE, false>), dim3(blocks), dim3(kThreads), lds_bytes(p.M), s, p);
  } else {
    if (int rc = allow_lds(irfft_rows_rl_kernel<LSR_EPI_UPDATE, true>, a2, "lsr_irfft_rows_rl_f32")) return rc;
    hipLaunchKernelGGL((irfft_rows_rl_kernel<LSR_EPI_UPDATE, true>), dim3(blocks), dim3(kThreads), lds_bytes(p.M), s, p);
  }
  return lsr::launch_status("lsr_irfft_rows_rl_f32");
}

// As lsr_irfft_rows_rl_f32, CHAINED into the forward x leg of the next convolution: the epilogue's output (zero-padded
// to the grid) is transformed and stored back over `spec`, which afterwards holds what lsr_rfft_rows_zero_t_c64 would
// have produced from `out`.  LSR_EPI_RATIO: `out` may be NULL (the ratio is then never written).
extern "C" int lsr_rl_rows_chain_f32(float* spec, int64_t Z, int64_t Y, int64_t X, const float* tw_half, const float* tw_x,
                                     int epilogue, const float* aux, float* out, int64_t Zo, int64_t Yo, int64_t Xo,
                                     float scale, float eps, int pz, int py, int px, const double* norm_table,
                                     float norm_full, double* stats, lsr_stream_t stream) {
  LSR_REQUIRE_PTR(spec);
  LSR_REQUIRE_PTR(aux);
  LSR_REQUIRE(epilogue == LSR_EPI_RATIO || epilogue == LSR_EPI_UPDATE, LSR_E_ARG,
              "epilogue must be LSR_EPI_RATIO or LSR_EPI_UPDATE, got %d", epilogue);
  if (epilogue == LSR_EPI_UPDATE) LSR_REQUIRE_PTR(out);
  LSR_REQUIRE(Zo > 0 && Yo > 0 && Xo > 0 && Zo <= Z && Yo <= Y && Xo <= X, LSR_E_SHAPE,
              "the volume (%lld,%lld,%lld) must be positive and fit the grid (%lld,%lld,%lld)", (long long)Zo, (long long)Yo,
              (long long)Xo, (long long)Z, (long long)Y, (long long)X);
  LSR_REQUIRE(scale > 0.0f && eps > 0.0f, LSR_E_ARG, "scale and eps must be positive");
  RowsArgs p{};
  if (int rc = fill(p, Z, Y, X, tw_half, tw_x)) return rc;
  if (epilogue == LSR_EPI_UPDATE) {
    LSR_REQUIRE_PTR(norm_table);
    LSR_REQUIRE(pz > 0 && py > 0 && px > 0 && pz % 2 == 1 && py % 2 == 1 && px % 2 == 1 && pz < 4096 && py < 4096 &&
                    px < 4096, LSR_E_ARG, "PSF extents (%d,%d,%d) must be odd and positive", pz, py, px);
    LSR_REQUIRE(norm_full > 0.0f, LSR_E_ARG, "norm_full must be positive");
  }
  p.spec = reinterpret_cast<float2*>(spec);
  p.aux = aux; p.out = out;
  p.Zo = static_cast<int>(Zo); p.Yo = static_cast<int>(Yo); p.Xo = static_cast<int>(Xo);
  p.scale = scale; p.eps = eps;
  p.pz = pz; p.py = py; p.px = px;
  p.norm_table = norm_table; p.norm_full = norm_full; p.stats = stats;
  const int64_t out_tiles = Zo * lsr::ceil_div(Yo, kRows), beside = Zo * (lsr::ceil_div(Y, kRows) - lsr::ceil_div(Yo, kRows));
  const unsigned blocks = xcd_grid(out_tiles) + static_cast<unsigned>(beside);
  hipStream_t s = lsr::as_stream(stream);
  static std::atomic<uint64_t> a0{0}, a1{0}, a2{0};
  if (epilogue == LSR_EPI_RATIO) {
    if (int rc = allow_lds(irfft_rows_rl_kernel<LSR_EPI_RATIO, false, true>, a0, "lsr_rl_rows_chain_f32")) return rc;
    hipLaunchKernelGGL((irfft_rows_rl_kernel<LSR_EPI_RATIO, false, true>), dim3(blocks), dim3(kThreads), lds_bytes(p.M), s, p);
  } else if (stats == nullptr) {
    if (int rc = allow_lds(irfft_rows_rl_kernel<LSR_EPI_UPDATE, false, true>, a1, "lsr_rl_rows_chain_f32")) return rc;
    hipLaunchKernelGGL((irfft_rows_rl_kernel<LSR_EPI_UPDATE, false, true>), dim3(blocks), dim3(kThreads), lds_bytes(p.M), s, p);
  } else {
    if (int rc = allow_lds(irfft_rows_rl_kernel<LSR_EPI_UPDATE, true, true>, a2, "lsr_rl_rows_chain_f32")) return rc;
    hipLaunchKernelGGL((irfft_rows_rl_kernel<LSR_EPI_UPDATE, true, true>), dim3(blocks), dim3(kThreads), lds_bytes(p.M), s, p);
  }
  return lsr::launch_status("lsr_rl_rows_chain_f32");
}
