// Reductions and filters behind the DynaTrack shift estimators that consume the deskewed volume
// (reference shrimpy/dynatrack/tracking.py, root /root/reference):
//
//   lsr_minmax_f32             img.min(), img.max()                      (:533-535, :460-463, :583)
//   lsr_histogram_f32          torch.histc(img, bins, min, max)          (:465, :586)
//   lsr_weighted_centroid_f32  _intensity_center_of_mass                 (:596-649)
//   lsr_mask_centroid_f32      _center_of_mass(img > threshold)          (:545-569, :540)
//   lsr_blur_reflect_f32       one axis of _gaussian_blur_3d             (:386-422; F.pad reflect + conv3d)
//   lsr_match_shape_f32, lsr_cross_power_c64, lsr_peak_abs_shifted_f32
//                              the element-wise steps of _phase_cross_corr (:266-378)
//
// All of them stream the volume once at HBM speed; the sums are accumulated in fp64 and reduced in a
// fixed order (two launches, no float atomics), so results do not depend on scheduling.

#include "common.hpp"

namespace {

constexpr int kBlocks = 2048;   // partial results of the two-stage reductions
constexpr int kThreads = 256;

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Grid-stride walk over a flat array: 16-byte loads, four of them in flight per thread, where the
// pointer is 16-byte aligned (torch allocations are); the tail and unaligned arrays go scalar.
template <typename F>
__device__ __forceinline__ void for_each_flat(const float* __restrict__ in, int64_t n, F&& f) {
  const int64_t tid = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  const int64_t nthreads = static_cast<int64_t>(gridDim.x) * kThreads;
  int64_t done = 0;
  if ((reinterpret_cast<uintptr_t>(in) & 15) == 0) {
    const int64_t n4 = n >> 2;
    const f32x4* in4 = reinterpret_cast<const f32x4*>(in);
    int64_t i = tid;
    for (; i + 3 * nthreads < n4; i += 4 * nthreads) {
      const f32x4 a = in4[i], b = in4[i + nthreads], c = in4[i + 2 * nthreads], d = in4[i + 3 * nthreads];
      f(a.x); f(a.y); f(a.z); f(a.w); f(b.x); f(b.y); f(b.z); f(b.w);
      f(c.x); f(c.y); f(c.z); f(c.w); f(d.x); f(d.y); f(d.z); f(d.w);
    }
    for (; i < n4; i += nthreads) {
      const f32x4 a = in4[i];
      f(a.x); f(a.y); f(a.z); f(a.w);
    }
    done = n4 << 2;
  }
  for (int64_t i = done + tid; i < n; i += nthreads) f(in[i]);
}

// ---------------------------------------------------------------------------------- min / max
__global__ __launch_bounds__(kThreads) void minmax_partial_kernel(const float* __restrict__ in, int64_t n,
                                                                  float* __restrict__ partial) {
  __shared__ float s_lo[kThreads], s_hi[kThreads];
  float lo = INFINITY, hi = -INFINITY;
  for_each_flat(in, n, [&](float v) {
    lo = fminf(lo, v);
    hi = fmaxf(hi, v);
  });
  s_lo[threadIdx.x] = lo;
  s_hi[threadIdx.x] = hi;
  __syncthreads();
  for (int w = kThreads / 2; w > 0; w >>= 1) {
    if (static_cast<int>(threadIdx.x) < w) {
      s_lo[threadIdx.x] = fminf(s_lo[threadIdx.x], s_lo[threadIdx.x + w]);
      s_hi[threadIdx.x] = fmaxf(s_hi[threadIdx.x], s_hi[threadIdx.x + w]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = s_lo[0];
    partial[2 * blockIdx.x + 1] = s_hi[0];
  }
}
// ... of uint16 camera counts (the fill value of a deskew with cval = "min"): 16-byte loads, eight counts each
__global__ __launch_bounds__(kThreads) void minmax_u16_partial_kernel(const unsigned short* __restrict__ in, int64_t n,
                                                                      float* __restrict__ partial) {
  __shared__ float s_lo[kThreads], s_hi[kThreads];
  unsigned lo = 0xFFFFu, hi = 0u;
  const int64_t tid = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  const int64_t nthreads = static_cast<int64_t>(gridDim.x) * kThreads;
  int64_t done = 0;
  if ((reinterpret_cast<uintptr_t>(in) & 15) == 0) {
    const int64_t n8 = n >> 3;
    const uint4* in8 = reinterpret_cast<const uint4*>(in);
    for (int64_t i = tid; i < n8; i += nthreads) {
      const uint4 v = in8[i];
      const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned a = w[k] & 0xFFFFu, b = w[k] >> 16;
        lo = min(lo, min(a, b));
        hi = max(hi, max(a, b));
      }
    }
    done = n8 << 3;
  }
  for (int64_t i = done + tid; i < n; i += nthreads) {
    const unsigned a = in[i];
    lo = min(lo, a);
    hi = max(hi, a);
  }
  s_lo[threadIdx.x] = static_cast<float>(lo);
  s_hi[threadIdx.x] = static_cast<float>(hi);
  __syncthreads();
  for (int w = kThreads / 2; w > 0; w >>= 1) {
    if (static_cast<int>(threadIdx.x) < w) {
      s_lo[threadIdx.x] = fminf(s_lo[threadIdx.x], s_lo[threadIdx.x + w]);
      s_hi[threadIdx.x] = fmaxf(s_hi[threadIdx.x], s_hi[threadIdx.x + w]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = s_lo[0];
    partial[2 * blockIdx.x + 1] = s_hi[0];
  }
}
__global__ __launch_bounds__(kThreads) void minmax_final_kernel(const float* __restrict__ partial, int nb,
                                                                float* __restrict__ out) {
  __shared__ float s_lo[kThreads], s_hi[kThreads];
  float lo = INFINITY, hi = -INFINITY;
  for (int i = threadIdx.x; i < nb; i += kThreads) {
    lo = fminf(lo, partial[2 * i]);
    hi = fmaxf(hi, partial[2 * i + 1]);
  }
  s_lo[threadIdx.x] = lo;
  s_hi[threadIdx.x] = hi;
  __syncthreads();
  for (int w = kThreads / 2; w > 0; w >>= 1) {
    if (static_cast<int>(threadIdx.x) < w) {
      s_lo[threadIdx.x] = fminf(s_lo[threadIdx.x], s_lo[threadIdx.x + w]);
      s_hi[threadIdx.x] = fmaxf(s_hi[threadIdx.x], s_hi[threadIdx.x + w]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = s_lo[0];
    out[1] = s_hi[0];
  }
}

// ---------------------------------------------------------------------------------- histogram
// torch.histc: bin = (int)((x - min) * nbins / (max - min)) in float32, x == max goes to the last
// bin, x outside [min, max] (and NaN) is dropped.
constexpr int kMaxBins = 4096;
__global__ __launch_bounds__(kThreads) void histogram_kernel(const float* __restrict__ in, int64_t n,
                                                             float vmin, float vmax, int nbins,
                                                             unsigned* __restrict__ counts) {
  extern __shared__ unsigned s_hist[];
  for (int b = threadIdx.x; b < nbins; b += kThreads) s_hist[b] = 0;
  __syncthreads();
  const float range = vmax - vmin, fb = static_cast<float>(nbins);
  for_each_flat(in, n, [&](float v) {
    if (v >= vmin && v <= vmax) {
      int bin = static_cast<int>((v - vmin) * fb / range);
      bin = bin >= nbins ? nbins - 1 : bin;
      atomicAdd(&s_hist[bin], 1u);
    }
  });
  __syncthreads();
  for (int b = threadIdx.x; b < nbins; b += kThreads)
    if (s_hist[b]) atomicAdd(&counts[b], s_hist[b]);  // integer adds: order-independent
}

// ---------------------------------------------------------------------------------- centroids
// One workgroup walks rows (z, y); a thread accumulates w and w * x over its x positions; the row
// totals feed sum(w), sum(w z), sum(w y), sum(w x) in fp64.  MASK: w = v > threshold, else
// w = max(v - background, 0) evaluated in f32 like the reference.
template <bool MASK>
__global__ __launch_bounds__(kThreads) void centroid_partial_kernel(const float* __restrict__ in, int Z, int Y,
                                                                    int X, float param,
                                                                    double* __restrict__ partial) {
  __shared__ double red[4][kThreads];
  double sw = 0.0, sz = 0.0, sy = 0.0, sx = 0.0;
  const int64_t rows = static_cast<int64_t>(Z) * Y;
  auto weight = [&](float v) {
    if constexpr (MASK) return v > param ? 1.0f : 0.0f;
    else return fmaxf(v - param, 0.0f);
  };
  // four rows per step: four independent loads in flight per thread
  for (int64_t r0 = static_cast<int64_t>(blockIdx.x) * 4; r0 < rows; r0 += static_cast<int64_t>(gridDim.x) * 4) {
    double w_row[4] = {0.0, 0.0, 0.0, 0.0}, wx_row[4] = {0.0, 0.0, 0.0, 0.0};
    const int nr = static_cast<int>(min(static_cast<int64_t>(4), rows - r0));
    for (int x = threadIdx.x; x < X; x += kThreads) {
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = in[(r0 + min(k, nr - 1)) * X + x];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const double w = static_cast<double>(weight(v[k]));
        w_row[k] += w;
        wx_row[k] += w * static_cast<double>(x);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k < nr) {
        const int64_t r = r0 + k;
        const int z = static_cast<int>(r / Y), y = static_cast<int>(r - static_cast<int64_t>(z) * Y);
        sw += w_row[k];
        sz += w_row[k] * static_cast<double>(z);
        sy += w_row[k] * static_cast<double>(y);
        sx += wx_row[k];
      }
    }
  }
  red[0][threadIdx.x] = sw; red[1][threadIdx.x] = sz; red[2][threadIdx.x] = sy; red[3][threadIdx.x] = sx;
  __syncthreads();
  for (int w = kThreads / 2; w > 0; w >>= 1) {
    if (static_cast<int>(threadIdx.x) < w)
#pragma unroll
      for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x < 4) partial[4 * blockIdx.x + threadIdx.x] = red[threadIdx.x][0];
}
__global__ __launch_bounds__(kThreads) void sum4_final_kernel(const double* __restrict__ partial, int nb,
                                                              double* __restrict__ out) {
  __shared__ double red[4][kThreads];
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < nb; i += kThreads)
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] += partial[4 * i + k];
#pragma unroll
  for (int k = 0; k < 4; ++k) red[k][threadIdx.x] = s[k];
  __syncthreads();
  for (int w = kThreads / 2; w > 0; w >>= 1) {
    if (static_cast<int>(threadIdx.x) < w)
#pragma unroll
      for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x < 4) out[threadIdx.x] = red[threadIdx.x][0];
}

// ---------------------------------------------------------------------------------- reflect blur
// out = correlate1d(in, taps) along one axis with F.pad(mode="reflect") borders (index -k -> k,
// n-1+k -> n-1-k; needs radius < n).  The volume is viewed as (outer, L, inner): the filtered axis
// has length L and stride `inner`.  A workgroup stages (SEG + 2r) x 64 values -- SEG positions of
// the axis plus halo, 64 consecutive `inner` elements -- in LDS; a thread owns OPT consecutive
// positions of one inner column and walks the taps eight at a time over a register window of
// OPT + 7 staged values, so an LDS value is read once per eight taps instead of once per tap
// (the per-tap form was LDS-issue-bound at 2 reads per FMA).  Taps are applied in ascending order,
// one FMA each, whatever the blocking.  For the contiguous axis (inner == 1) lanes run along the
// axis itself: four consecutive outputs per thread, 16-byte LDS reads.
// Optional input map (v - sub) / div fused into staging (the reference rescales to [0, 1] first).
constexpr int kBlurMaxR = 64;
constexpr int kTapBlock = 8;

struct BlurArgs {
  const float* in;
  float* out;
  const float* taps;  // device, 2r+1 floats
  int64_t outer, L, inner;
  int r;
  float sub, div;     // div == 0: no map
};

__device__ __forceinline__ int reflect(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  return i;
}

template <int OPT>  // outputs per thread; segment = 4 * OPT positions
__global__ __launch_bounds__(kThreads) void blur_strided_kernel(BlurArgs p) {
  constexpr int kSeg = (kThreads / 64) * OPT;
  extern __shared__ float tile[];  // [(kSeg + 2r + kTapBlock)][64]; the last rows are never used in an FMA
  __shared__ float s_taps[2 * kBlurMaxR + 1];
  const int r = p.r, ntaps = 2 * r + 1;
  const int64_t inner_tiles = (p.inner + 63) / 64, seg_tiles = (p.L + kSeg - 1) / kSeg;
  int64_t bid = blockIdx.x;
  const int64_t it = bid % inner_tiles;
  bid /= inner_tiles;
  const int64_t st = bid % seg_tiles;
  const int64_t o = bid / seg_tiles;
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int64_t i0 = it * 64 + lane;
  const bool col_ok = i0 < p.inner;
  const int64_t a0 = st * kSeg;
  // positions of this segment that exist, plus halo: reflect() is only valid within r of the axis
  const int n_out = static_cast<int>(min(static_cast<int64_t>(kSeg), p.L - a0));
  const int rows = n_out + 2 * r;
  const float* base = p.in + o * p.L * p.inner + (col_ok ? i0 : 0);
  for (int t = threadIdx.x; t < ntaps; t += kThreads) s_taps[t] = p.taps[t];
  // unconditional loads (columns past the volume read column 0 and are never stored), eight in
  // flight per thread
  constexpr int kStep = kThreads / 64, kBatch = 8;
  for (int row0 = grp; row0 < rows; row0 += kStep * kBatch) {
    float v[kBatch];
#pragma unroll
    for (int i = 0; i < kBatch; ++i) {
      const int a = reflect(static_cast<int>(a0) + min(row0 + i * kStep, rows - 1) - r, static_cast<int>(p.L));
      v[i] = base[static_cast<int64_t>(a) * p.inner];
    }
#pragma unroll
    for (int i = 0; i < kBatch; ++i) {
      const int row = row0 + i * kStep;
      if (row < rows) tile[row * 64 + lane] = p.div != 0.0f ? (v[i] - p.sub) / p.div : v[i];
    }
  }
  __syncthreads();
  const int k0 = grp * OPT;
  if (k0 >= n_out) return;
  float acc[OPT];
#pragma unroll
  for (int k = 0; k < OPT; ++k) acc[k] = 0.0f;
  const float* col = tile + k0 * 64 + lane;
  for (int t0 = 0; t0 < ntaps; t0 += kTapBlock) {
    float win[OPT + kTapBlock - 1];
#pragma unroll
    for (int j = 0; j < OPT + kTapBlock - 1; ++j) win[j] = col[(t0 + j) * 64];
    if (t0 + kTapBlock <= ntaps) {
#pragma unroll
      for (int tt = 0; tt < kTapBlock; ++tt) {
        const float w = s_taps[t0 + tt];
#pragma unroll
        for (int k = 0; k < OPT; ++k) acc[k] = fmaf(w, win[k + tt], acc[k]);
      }
    } else {
#pragma unroll
      for (int tt = 0; tt < kTapBlock; ++tt) {
        if (t0 + tt < ntaps) {  // uniform
          const float w = s_taps[t0 + tt];
#pragma unroll
          for (int k = 0; k < OPT; ++k) acc[k] = fmaf(w, win[k + tt], acc[k]);
        }
      }
    }
  }
  if (col_ok) {
    float* obase = p.out + o * p.L * p.inner + i0;
#pragma unroll
    for (int k = 0; k < OPT; ++k)
      if (k0 + k < n_out) obase[(a0 + k0 + k) * p.inner] = acc[k];
  }
}

// The same tiling with two adjacent columns per lane: 128-column tiles (512-byte rows), 8-byte LDS
// reads and packed FMAs (v_pk_fma_f32), for the long kernels where the arithmetic is what bounds
// the pass (41 taps at sigma = 5).  Each column sees its taps in the same order, one FMA each.
// Needs an even `inner` and 8-byte aligned arrays (the two columns travel as one float2).
typedef float bf32x2 __attribute__((ext_vector_type(2)));
template <int OPT>
__global__ __launch_bounds__(kThreads) void blur_strided_pk_kernel(BlurArgs p) {
  constexpr int kSeg = (kThreads / 64) * OPT;
  extern __shared__ __attribute__((aligned(8))) float tile[];  // [(kSeg + 2r + kTapBlock)][128]
  __shared__ float s_taps[2 * kBlurMaxR + 1];
  bf32x2* tile2 = reinterpret_cast<bf32x2*>(tile);             // [rows][64] pairs
  const int r = p.r, ntaps = 2 * r + 1;
  const int64_t inner_tiles = (p.inner + 127) / 128, seg_tiles = (p.L + kSeg - 1) / kSeg;
  int64_t bid = blockIdx.x;
  const int64_t it = bid % inner_tiles;
  bid /= inner_tiles;
  const int64_t st = bid % seg_tiles;
  const int64_t o = bid / seg_tiles;
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int64_t i0 = it * 128 + 2 * lane;
  const bool col_ok = i0 < p.inner;  // inner is even: both columns or none
  const int64_t a0 = st * kSeg;
  const int n_out = static_cast<int>(min(static_cast<int64_t>(kSeg), p.L - a0));
  const int rows = n_out + 2 * r;
  const float* base = p.in + o * p.L * p.inner + (col_ok ? i0 : 0);
  for (int t = threadIdx.x; t < ntaps; t += kThreads) s_taps[t] = p.taps[t];
  constexpr int kStep = kThreads / 64, kBatch = 8;
  for (int row0 = grp; row0 < rows; row0 += kStep * kBatch) {
    bf32x2 v[kBatch];
#pragma unroll
    for (int i = 0; i < kBatch; ++i) {
      const int a = reflect(static_cast<int>(a0) + min(row0 + i * kStep, rows - 1) - r, static_cast<int>(p.L));
      v[i] = *reinterpret_cast<const bf32x2*>(base + static_cast<int64_t>(a) * p.inner);
    }
#pragma unroll
    for (int i = 0; i < kBatch; ++i) {
      const int row = row0 + i * kStep;
      if (row < rows) {
        bf32x2 w = v[i];
        if (p.div != 0.0f) { w.x = (w.x - p.sub) / p.div; w.y = (w.y - p.sub) / p.div; }
        tile2[row * 64 + lane] = w;
      }
    }
  }
  __syncthreads();
  const int k0 = grp * OPT;
  if (k0 >= n_out) return;
  bf32x2 acc[OPT];
#pragma unroll
  for (int k = 0; k < OPT; ++k) acc[k] = bf32x2{0.0f, 0.0f};
  const bf32x2* col = tile2 + k0 * 64 + lane;
  for (int t0 = 0; t0 < ntaps; t0 += kTapBlock) {
    bf32x2 win[OPT + kTapBlock - 1];
#pragma unroll
    for (int j = 0; j < OPT + kTapBlock - 1; ++j) win[j] = col[(t0 + j) * 64];
#pragma unroll
    for (int tt = 0; tt < kTapBlock; ++tt) {
      if (t0 + tt < ntaps) {  // uniform; false only in the last block
        const float w = s_taps[t0 + tt];
        const bf32x2 w2 = {w, w};
#pragma unroll
        for (int k = 0; k < OPT; ++k) acc[k] = __builtin_elementwise_fma(w2, win[k + tt], acc[k]);
      }
    }
  }
  if (col_ok) {
    float* obase = p.out + o * p.L * p.inner + i0;
#pragma unroll
    for (int k = 0; k < OPT; ++k)
      if (k0 + k < n_out) *reinterpret_cast<bf32x2*>(obase + (a0 + k0 + k) * p.inner) = acc[k];
  }
}

// Marching form of the strided blur for radii whose ring fits LDS: a workgroup owns 256 consecutive
// `inner` elements (1 KB rows) of one `outer` index and walks a segment of the axis 16 positions at
// a time, keeping the last 2r + 16 rows in an LDS ring -- every source row is read once (the tiled
// form above re-reads 2r rows per 128) and in 1 KB pieces instead of 256-byte ones.  Same taps in
// the same order, one FMA each: bit-identical to the tiled form.
constexpr int kMarchRows = 16;
constexpr int kMarchCols = kThreads;
// up to 40 KB of ring: four workgroups per CU.  Beyond that the tiled form is faster (measured at
// r = 20: 3.0 ms against 2.1-2.5 ms per axis; at r = 8: 1.54-1.64 ms against 1.80-2.02 ms)
constexpr int kMarchMaxR = 12;
struct BlurMarchArgs {
  BlurArgs b;
  int seg_len;   // positions per workgroup along the axis
  int segs;      // segments per column strip
};
__global__ __launch_bounds__(kThreads) void blur_march_kernel(BlurMarchArgs q) {
  const BlurArgs& p = q.b;
  extern __shared__ float ring[];  // [2r + 16][256]
  __shared__ float s_taps[2 * kBlurMaxR + 1];
  const int r = p.r, ntaps = 2 * r + 1, RR = 2 * r + kMarchRows;
  const int64_t strips = (p.inner + kMarchCols - 1) / kMarchCols;
  int64_t bid = blockIdx.x;
  const int seg = static_cast<int>(bid % q.segs);
  bid /= q.segs;
  const int64_t strip = bid % strips;
  const int64_t o = bid / strips;
  const int L = static_cast<int>(p.L);
  const int a_begin = seg * q.seg_len, a_end = min(a_begin + q.seg_len, L);
  const int64_t col = strip * kMarchCols + threadIdx.x;
  const bool col_ok = col < p.inner;
  const float* base = p.in + o * p.L * p.inner + (col_ok ? col : p.inner - 1);
  float* obase = p.out + o * p.L * p.inner + col;
  for (int t = threadIdx.x; t < ntaps; t += kThreads) s_taps[t] = p.taps[t];
  float* mine = ring + threadIdx.x;
  auto put = [&](int j, float v) {  // source position j (may lie outside [0, L): reflected by the loader)
    int slot = (j + r) % RR;        // j >= -r always
    mine[slot * kMarchCols] = p.div != 0.0f ? (v - p.sub) / p.div : v;
  };
  // fill: positions a_begin - r .. a_begin + r - 1 (the first step adds the next 16)
  for (int j0 = a_begin - r; j0 < a_begin + r; j0 += 8) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = base[static_cast<int64_t>(reflect(min(j0 + i, a_begin + r - 1), L)) * p.inner];
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (j0 + i < a_begin + r) put(j0 + i, v[i]);
  }
  for (int a = a_begin; a < a_end; a += kMarchRows) {
    // rows a + r .. a + r + 15 replace the oldest 16 (their readers finished before the barrier
    // that ended the previous step)
    float v[kMarchRows];
    const int last = min(a_end - 1 + r, L - 1 + r);  // nothing beyond is needed; reflect() needs j <= L-1+r
#pragma unroll
    for (int i = 0; i < kMarchRows; ++i)
      v[i] = base[static_cast<int64_t>(reflect(min(a + r + i, last), L)) * p.inner];
#pragma unroll
    for (int i = 0; i < kMarchRows; ++i) put(a + r + i, v[i]);
    __syncthreads();
    float acc[kMarchRows];
#pragma unroll
    for (int k = 0; k < kMarchRows; ++k) acc[k] = 0.0f;
    int slot0 = (a - r + r) % RR;  // ring slot of source position a - r (tap 0 of output a)
    for (int t0 = 0; t0 < ntaps; t0 += kTapBlock) {
      float win[kMarchRows + kTapBlock - 1];
      int slot = slot0;
#pragma unroll
      for (int j = 0; j < kMarchRows + kTapBlock - 1; ++j) {
        win[j] = mine[slot * kMarchCols];
        slot = slot + 1 == RR ? 0 : slot + 1;
      }
#pragma unroll
      for (int tt = 0; tt < kTapBlock; ++tt) {
        if (t0 + tt < ntaps) {  // uniform; false only in the last block
          const float w = s_taps[t0 + tt];
#pragma unroll
          for (int k = 0; k < kMarchRows; ++k) acc[k] = fmaf(w, win[k + tt], acc[k]);
        }
      }
      slot0 += kTapBlock;
      if (slot0 >= RR) slot0 -= RR;
    }
    if (col_ok) {
#pragma unroll
      for (int k = 0; k < kMarchRows; ++k)
        if (a + k < a_end) obase[static_cast<int64_t>(a + k) * p.inner] = acc[k];
    }
    __syncthreads();
  }
}

constexpr int kRowSeg = 4 * kThreads;  // outputs per row segment of the contiguous-axis kernel
__global__ __launch_bounds__(kThreads) void blur_contiguous_kernel(BlurArgs p) {
  extern __shared__ __attribute__((aligned(16))) float tile[];  // [kRowSeg + 2r + kTapBlock + 4]
  __shared__ float s_taps[2 * kBlurMaxR + 1];
  const int r = p.r, ntaps = 2 * r + 1;
  const int64_t seg_tiles = (p.L + kRowSeg - 1) / kRowSeg;
  const int64_t items = p.outer * seg_tiles;  // (row, segment) pairs; a workgroup walks several
  for (int t = threadIdx.x; t < ntaps; t += kThreads) s_taps[t] = p.taps[t];
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  for (int64_t item = blockIdx.x; item < items; item += gridDim.x) {
    const int64_t st = item % seg_tiles, row = item / seg_tiles;
    const int64_t a0 = st * kRowSeg;
    const float* base = p.in + row * p.L;
    const int n_out = static_cast<int>(min(static_cast<int64_t>(kRowSeg), p.L - a0));
    __syncthreads();  // the previous item's reads are done (and the taps are in place)
    const int n_stage = n_out + 2 * r;
    for (int t0 = threadIdx.x; t0 < n_stage; t0 += kThreads * 5) {  // five loads in flight per thread
      float v[5];
#pragma unroll
      for (int i = 0; i < 5; ++i)
        v[i] = base[reflect(static_cast<int>(a0) + min(t0 + i * kThreads, n_stage - 1) - r, static_cast<int>(p.L))];
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const int t = t0 + i * kThreads;
        if (t < n_stage) tile[t] = p.div != 0.0f ? (v[i] - p.sub) / p.div : v[i];
      }
    }
    __syncthreads();
    const int k0 = 4 * threadIdx.x;  // four consecutive outputs
    if (k0 < n_out) {
      float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      for (int t0 = 0; t0 < ntaps; t0 += kTapBlock) {
        float win[12];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(tile + k0 + t0 + 4 * q);
          win[4 * q] = v.x; win[4 * q + 1] = v.y; win[4 * q + 2] = v.z; win[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int tt = 0; tt < kTapBlock; ++tt) {
          if (t0 + tt < ntaps) {  // uniform; false only in the last block
            const float w = s_taps[t0 + tt];
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = fmaf(w, win[k + tt], acc[k]);
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (a0 + k0 + k < p.L) p.out[row * p.L + a0 + k0 + k] = acc[k];
    }
  }
}

// ---------------------------------------------------------------------------------- phase cross-correlation
// The element-wise steps around the two FFTs of _phase_cross_corr (tracking.py:309-378); the FFTs
// themselves are library calls (rocFFT through torch.fft, like a plain library GEMM).

// _match_shape (:266-306): per axis, pad to the FFT length with F.pad "reflect" (left = d // 2) or
// crop the centre (start = d // 2).
struct MatchArgs {
  const float* in;
  float* out;
  int Zi, Yi, Xi, Zo, Yo, Xo;
};
__device__ __forceinline__ int match_index(int o, int ni, int no) {
  if (no > ni) return reflect(o - (no - ni) / 2, ni);
  return o + (ni - no) / 2;
}
__global__ __launch_bounds__(kThreads) void match_shape_kernel(MatchArgs p) {
  const int64_t rows = static_cast<int64_t>(p.Zo) * p.Yo;
  for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
    const int zo = static_cast<int>(r / p.Yo), yo = static_cast<int>(r - static_cast<int64_t>(zo) * p.Yo);
    const float* src = p.in + (static_cast<int64_t>(match_index(zo, p.Zi, p.Zo)) * p.Yi + match_index(yo, p.Yi, p.Yo)) * p.Xi;
    float* dst = p.out + r * p.Xo;
    for (int xo = threadIdx.x; xo < p.Xo; xo += kThreads) dst[xo] = src[match_index(xo, p.Xi, p.Xo)];
  }
}

// prod = f1 * conj(f2) (complex64 as float pairs), written over f1 (INTO_B = false) or over f2
// (INTO_B = true: f1, e.g. a cached spectrum of the reference volume, stays intact)
template <bool INTO_B>
__global__ __launch_bounds__(kThreads) void cross_power_kernel(float* __restrict__ a, float* __restrict__ b,
                                                               int64_t n) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kThreads;
  f32x2* a2 = reinterpret_cast<f32x2*>(a);
  f32x2* b2 = reinterpret_cast<f32x2*>(b);
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n; i += stride) {
    const f32x2 va = a2[i], vb = b2[i];
    // (ar + i ai)(br - i bi), the products and sums torch's complex multiply performs
    const f32x2 r = {va.x * vb.x + va.y * vb.y, va.y * vb.x - va.x * vb.y};
    (INTO_B ? b2 : a2)[i] = r;
  }
}

// dst[a][c][b] = src[a][b][c] for 8-byte elements (complex64): the layout changes between the per-axis
// transforms of the 3-D FFT (dynatrack._Fft3 runs one contiguous batched 1-D transform per axis; rocFFT's
// own 3-D plan spends more time in its transposes and in torch's input clone than in the butterflies).
// 64 x 64 tiles through LDS: 512-byte runs on both sides.  Also the plain 2-D transpose (A = 1).
constexpr int kTr = 64;
struct TransposeArgs {
  const double* src;     // 8-byte payloads, moved as doubles (never interpreted)
  double* dst;
  int64_t A, B, C;
  int64_t tiles_b, tiles_c;
};
__global__ __launch_bounds__(kThreads) void transpose_c64_kernel(TransposeArgs p) {
  __shared__ double tile[kTr][kTr + 1];
  int64_t t = blockIdx.x;
  const int64_t tc = t % p.tiles_c;
  t /= p.tiles_c;
  const int64_t tb = t % p.tiles_b;
  const int64_t a = t / p.tiles_b;
  const int64_t b0 = tb * kTr, c0 = tc * kTr;
  const int lane = threadIdx.x & 63, row0 = threadIdx.x >> 6;     // 4 rows per pass, 16 passes
  constexpr int kPer = kTr / (kThreads / 64);
  // all 16 loads of a thread are issued before the first use: addresses are clamped into the array
  // (the values of out-of-range positions are never stored), so no load sits under a condition
  const double* src = p.src + a * p.B * p.C;
  const int64_t col = min(c0 + lane, p.C - 1);
  double v[kPer];
#pragma unroll
  for (int i = 0; i < kPer; ++i) v[i] = src[min(b0 + row0 + 4 * i, p.B - 1) * p.C + col];
#pragma unroll
  for (int i = 0; i < kPer; ++i) tile[row0 + 4 * i][lane] = v[i];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kPer; ++i) v[i] = tile[lane][row0 + 4 * i];
  double* dst = p.dst + (a * p.C + c0) * p.B + b0 + lane;
  if (b0 + lane < p.B) {
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      const int r = row0 + 4 * i;
      if (c0 + r < p.C) dst[static_cast<int64_t>(r) * p.B] = v[i];
    }
  }
}

// argmax(fftshift(|corr|)) without materialising either: the largest |v|, ties resolved by the
// smallest flat index in fftshift order (torch.argmax returns the first maximum).
struct PeakArgs {
  const float* in;
  int Z, Y, X;
  float* pval;                   // partial maxima
  unsigned long long* pidx;      // their shifted flat indices
};
__device__ __forceinline__ void peak_merge(float& v, unsigned long long& i, float v2, unsigned long long i2) {
  if (v2 > v || (v2 == v && i2 < i)) { v = v2; i = i2; }
}
__global__ __launch_bounds__(kThreads) void peak_partial_kernel(PeakArgs p) {
  __shared__ float s_v[kThreads];
  __shared__ unsigned long long s_i[kThreads];
  float best = -1.0f;
  unsigned long long best_i = ~0ull;
  const int64_t rows = static_cast<int64_t>(p.Z) * p.Y;
  const bool vec = (p.X & 3) == 0 && (reinterpret_cast<uintptr_t>(p.in) & 15) == 0;   // rows are 16-byte aligned
  for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
    const int z = static_cast<int>(r / p.Y), y = static_cast<int>(r - static_cast<int64_t>(z) * p.Y);
    const int zs = (z + p.Z / 2) % p.Z, ys = (y + p.Y / 2) % p.Y;   // fftshift: index i -> (i + n/2) % n
    const float* row = p.in + r * p.X;
    const unsigned long long base = (static_cast<unsigned long long>(zs) * p.Y + ys) * p.X;
    // the shifted index is only worked out for a value that can replace the running maximum
    if (vec) {
      const float4* row4 = reinterpret_cast<const float4*>(row);
      for (int x4 = threadIdx.x; x4 < p.X / 4; x4 += kThreads) {
        const float4 f = row4[x4];
        const float v[4] = {fabsf(f.x), fabsf(f.y), fabsf(f.z), fabsf(f.w)};
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (v[k] >= best) peak_merge(best, best_i, v[k], base + static_cast<unsigned>((4 * x4 + k + p.X / 2) % p.X));
      }
    } else {
      for (int x = threadIdx.x; x < p.X; x += kThreads) {
        const float v = fabsf(row[x]);
        if (v >= best) peak_merge(best, best_i, v, base + static_cast<unsigned>((x + p.X / 2) % p.X));
      }
    }
  }
  s_v[threadIdx.x] = best;
  s_i[threadIdx.x] = best_i;
  __syncthreads();
  for (int w = kThreads / 2; w > 0; w >>= 1) {
    if (static_cast<int>(threadIdx.x) < w) peak_merge(s_v[threadIdx.x], s_i[threadIdx.x], s_v[threadIdx.x + w], s_i[threadIdx.x + w]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    p.pval[blockIdx.x] = s_v[0];
    p.pidx[blockIdx.x] = s_i[0];
  }
}
__global__ __launch_bounds__(kThreads) void peak_final_kernel(const float* __restrict__ pval,
                                                              const unsigned long long* __restrict__ pidx, int nb,
                                                              long long* __restrict__ out) {
  __shared__ float s_v[kThreads];
  __shared__ unsigned long long s_i[kThreads];
  float best = -1.0f;
  unsigned long long best_i = ~0ull;
  for (int i = threadIdx.x; i < nb; i += kThreads) peak_merge(best, best_i, pval[i], pidx[i]);
  s_v[threadIdx.x] = best;
  s_i[threadIdx.x] = best_i;
  __syncthreads();
  for (int w = kThreads / 2; w > 0; w >>= 1) {
    if (static_cast<int>(threadIdx.x) < w) peak_merge(s_v[threadIdx.x], s_i[threadIdx.x], s_v[threadIdx.x + w], s_i[threadIdx.x + w]);
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = static_cast<long long>(s_i[0]);
}

int check_volume(const float* in, int64_t Z, int64_t Y, int64_t X) {
  LSR_REQUIRE_PTR(in);
  LSR_REQUIRE(Z > 0 && Y > 0 && X > 0, LSR_E_SHAPE, "shape (%lld,%lld,%lld) must be positive",
              (long long)Z, (long long)Y, (long long)X);
  LSR_REQUIRE_VOLUME(Z, Y, X);
  LSR_REQUIRE(Z < (int64_t(1) << 30) && Y < (int64_t(1) << 30) && X < (int64_t(1) << 30), LSR_E_UNSUPPORTED,
              "a dimension exceeds 2^30");
  return LSR_OK;
}

int grid_for(int64_t work_items) {
  const int64_t b = lsr::ceil_div(work_items, static_cast<int64_t>(kThreads));
  return static_cast<int>(b < kBlocks ? (b < 1 ? 1 : b) : kBlocks);
}

}  // namespace

extern "C" int lsr_reduce_scratch_bytes(void) { return kBlocks * 4 * static_cast<int>(sizeof(double)); }

extern "C" int lsr_minmax_f32(const float* in, int64_t n, float* out2, void* scratch, lsr_stream_t stream) {
  LSR_REQUIRE_PTR(in);
  LSR_REQUIRE_PTR(out2);
  LSR_REQUIRE_PTR(scratch);
  LSR_REQUIRE(n > 0, LSR_E_SHAPE, "n = %lld must be positive", (long long)n);
  LSR_REQUIRE_COUNT(n);
  const int nb = grid_for(n);
  hipStream_t s = lsr::as_stream(stream);
  float* partial = static_cast<float*>(scratch);
  hipLaunchKernelGGL(minmax_partial_kernel, dim3(nb), dim3(kThreads), 0, s, in, n, partial);
  hipLaunchKernelGGL(minmax_final_kernel, dim3(1), dim3(kThreads), 0, s, partial, nb, out2);
  return lsr::launch_status("lsr_minmax_f32");
}

extern "C" int lsr_minmax_u16(const uint16_t* in, int64_t n, float* out2, void* scratch, lsr_stream_t stream) {
  LSR_REQUIRE_PTR(in);
  LSR_REQUIRE_PTR(out2);
  LSR_REQUIRE_PTR(scratch);
  LSR_REQUIRE(n > 0, LSR_E_SHAPE, "n = %lld must be positive", (long long)n);
  LSR_REQUIRE_COUNT(n);
  const int nb = grid_for((n + 1) / 2);
  hipStream_t s = lsr::as_stream(stream);
  float* partial = static_cast<float*>(scratch);
  hipLaunchKernelGGL(minmax_u16_partial_kernel, dim3(nb), dim3(kThreads), 0, s, in, n, partial);
  hipLaunchKernelGGL(minmax_final_kernel, dim3(1), dim3(kThreads), 0, s, partial, nb, out2);
  return lsr::launch_status("lsr_minmax_u16");
}

extern "C" int lsr_histogram_f32(const float* in, int64_t n, float vmin, float vmax, int nbins,
                                 unsigned* counts, lsr_stream_t stream) {
  LSR_REQUIRE_PTR(in);
  LSR_REQUIRE_PTR(counts);
  LSR_REQUIRE(n > 0, LSR_E_SHAPE, "n = %lld must be positive", (long long)n);
  LSR_REQUIRE_COUNT(n);
  // (one bin may take every sample: 2^32 of them would wrap its counter -- callers add up pieces)
  LSR_REQUIRE(n < (int64_t(1) << 32), LSR_E_UNSUPPORTED, "n = %lld: the bins count in 32 bits, histogram the volume in pieces",
              (long long)n);
  LSR_REQUIRE(nbins >= 1 && nbins <= kMaxBins, LSR_E_ARG, "nbins %d outside [1, %d]", nbins, kMaxBins);
  LSR_REQUIRE(vmax > vmin, LSR_E_ARG, "histogram range [%g, %g] is empty", vmin, vmax);
  hipStream_t s = lsr::as_stream(stream);
  if (hipMemsetAsync(counts, 0, sizeof(unsigned) * nbins, s) != hipSuccess)
    return lsr::launch_status("lsr_histogram_f32");
  hipLaunchKernelGGL(histogram_kernel, dim3(grid_for(n)), dim3(kThreads), sizeof(unsigned) * nbins, s, in,
                     n, vmin, vmax, nbins, counts);
  return lsr::launch_status("lsr_histogram_f32");
}

namespace {
template <bool MASK>
int centroid(const char* what, const float* in, int64_t Z, int64_t Y, int64_t X, float param, double* out4,
             void* scratch, lsr_stream_t stream) {
  if (int rc = check_volume(in, Z, Y, X)) return rc;
  LSR_REQUIRE_PTR(out4);
  LSR_REQUIRE_PTR(scratch);
  const int64_t rows = Z * Y, groups = lsr::ceil_div(rows, int64_t(4));
  const int nb = static_cast<int>(groups < kBlocks ? groups : kBlocks);
  hipStream_t s = lsr::as_stream(stream);
  double* partial = static_cast<double*>(scratch);
  hipLaunchKernelGGL(centroid_partial_kernel<MASK>, dim3(nb), dim3(kThreads), 0, s, in, static_cast<int>(Z),
                     static_cast<int>(Y), static_cast<int>(X), param, partial);
  hipLaunchKernelGGL(sum4_final_kernel, dim3(1), dim3(kThreads), 0, s, partial, nb, out4);
  return lsr::launch_status(what);
}
}  // namespace

extern "C" int lsr_weighted_centroid_f32(const float* in, int64_t Z, int64_t Y, int64_t X, float background,
                                         double* out4, void* scratch, lsr_stream_t stream) {
  return centroid<false>("lsr_weighted_centroid_f32", in, Z, Y, X, background, out4, scratch, stream);
}

extern "C" int lsr_mask_centroid_f32(const float* in, int64_t Z, int64_t Y, int64_t X, float threshold,
                                     double* out4, void* scratch, lsr_stream_t stream) {
  return centroid<true>("lsr_mask_centroid_f32", in, Z, Y, X, threshold, out4, scratch, stream);
}

extern "C" int lsr_blur_reflect_f32(const float* in, float* out, int64_t Z, int64_t Y, int64_t X, int axis,
                                    const float* taps, int radius, float sub, float div,
                                    lsr_stream_t stream) {
  if (int rc = check_volume(in, Z, Y, X)) return rc;
  LSR_REQUIRE_PTR(out);
  LSR_REQUIRE_PTR(taps);
  LSR_REQUIRE(in != out, LSR_E_ARG, "out must not alias in");
  LSR_REQUIRE(axis >= 0 && axis <= 2, LSR_E_ARG, "axis %d must be 0 (z), 1 (y) or 2 (x)", axis);
  LSR_REQUIRE(radius >= 0 && radius <= kBlurMaxR, LSR_E_UNSUPPORTED, "radius %d outside [0, %d]", radius,
              kBlurMaxR);
  const int64_t dims[3] = {Z, Y, X};
  LSR_REQUIRE(radius < dims[axis], LSR_E_ARG, "reflect padding needs radius %d < axis length %lld", radius,
              (long long)dims[axis]);
  BlurArgs p;
  p.in = in; p.out = out; p.taps = taps; p.r = radius; p.sub = sub; p.div = div;
  p.L = dims[axis];
  p.outer = axis == 0 ? 1 : (axis == 1 ? Z : Z * Y);
  p.inner = axis == 0 ? Y * X : (axis == 1 ? X : 1);
  hipStream_t s = lsr::as_stream(stream);
  if (axis == 2) {
    const int64_t items = p.outer * lsr::ceil_div(p.L, static_cast<int64_t>(kRowSeg));
    const int64_t blocks = items < 16384 ? items : 16384;
    hipLaunchKernelGGL(blur_contiguous_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kThreads),
                       sizeof(float) * (kRowSeg + 2 * radius + kTapBlock + 4), s, p);
  } else if (radius <= kMarchMaxR && p.L >= 2 * kMarchRows) {
    BlurMarchArgs q;
    q.b = p;
    q.seg_len = p.L > 768 ? 512 : static_cast<int>(p.L);
    q.segs = static_cast<int>(lsr::ceil_div(p.L, static_cast<int64_t>(q.seg_len)));
    const int64_t blocks = p.outer * lsr::ceil_div(p.inner, static_cast<int64_t>(kMarchCols)) * q.segs;
    LSR_REQUIRE(blocks < (int64_t(1) << 31), LSR_E_SHAPE, "grid of %lld workgroups is too large",
                (long long)blocks);
    hipLaunchKernelGGL(blur_march_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kThreads),
                       sizeof(float) * (2 * radius + kMarchRows) * kMarchCols, s, q);
  } else if (radius > kMarchMaxR && p.inner % 2 == 0 && p.L > 32 &&
             ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 7) == 0 &&
             (64 + 2 * radius + kTapBlock) * 128 * sizeof(float) <= 65536) {
    // long kernels: two columns per lane, packed FMAs
    const int64_t blocks = p.outer * lsr::ceil_div(p.L, int64_t(64)) * lsr::ceil_div(p.inner, int64_t(128));
    LSR_REQUIRE(blocks < (int64_t(1) << 31), LSR_E_SHAPE, "grid of %lld workgroups is too large",
                (long long)blocks);
    hipLaunchKernelGGL(blur_strided_pk_kernel<16>, dim3(static_cast<unsigned>(blocks)), dim3(kThreads),
                       sizeof(float) * (64 + 2 * radius + kTapBlock) * 128, s, p);
  } else {
    // 128-position segments (less halo per output) while the tile stays within 64 KB of LDS
    const bool wide = (128 + 2 * radius + kTapBlock) * 64 * sizeof(float) <= 65536 && p.L > 64;
    const int seg = wide ? 128 : 64;
    const int64_t blocks = p.outer * lsr::ceil_div(p.L, static_cast<int64_t>(seg)) * lsr::ceil_div(p.inner, int64_t(64));
    LSR_REQUIRE(blocks < (int64_t(1) << 31), LSR_E_SHAPE, "grid of %lld workgroups is too large",
                (long long)blocks);
    const size_t lds = sizeof(float) * (seg + 2 * radius + kTapBlock) * 64;
    if (wide)
      hipLaunchKernelGGL(blur_strided_kernel<32>, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), lds, s, p);
    else
      hipLaunchKernelGGL(blur_strided_kernel<16>, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), lds, s, p);
  }
  return lsr::launch_status("lsr_blur_reflect_f32");
}

extern "C" int lsr_match_shape_f32(const float* in, int64_t Zi, int64_t Yi, int64_t Xi, float* out, int64_t Zo,
                                   int64_t Yo, int64_t Xo, lsr_stream_t stream) {
  if (int rc = check_volume(in, Zi, Yi, Xi)) return rc;
  if (int rc = check_volume(out, Zo, Yo, Xo)) return rc;
  LSR_REQUIRE(in != out, LSR_E_ARG, "out must not alias in");
  const int64_t si[3] = {Zi, Yi, Xi}, so[3] = {Zo, Yo, Xo};
  for (int a = 0; a < 3; ++a)
    LSR_REQUIRE(so[a] <= si[a] || (so[a] - si[a] + 1) / 2 < si[a], LSR_E_ARG,
                "axis %d: reflect padding %lld -> %lld needs a pad smaller than the axis", a,
                (long long)si[a], (long long)so[a]);
  MatchArgs p{in, out, static_cast<int>(Zi), static_cast<int>(Yi), static_cast<int>(Xi), static_cast<int>(Zo),
              static_cast<int>(Yo), static_cast<int>(Xo)};
  const int64_t rows = Zo * Yo;
  hipLaunchKernelGGL(match_shape_kernel, dim3(static_cast<unsigned>(rows < 65536 ? rows : 65536)), dim3(kThreads), 0,
                     lsr::as_stream(stream), p);
  return lsr::launch_status("lsr_match_shape_f32");
}

extern "C" int lsr_cross_power_c64(float* a, const float* b, int64_t n, lsr_stream_t stream) {
  LSR_REQUIRE_PTR(a);
  LSR_REQUIRE_PTR(b);
  LSR_REQUIRE(n > 0, LSR_E_SHAPE, "n = %lld must be positive", (long long)n);
  LSR_REQUIRE_COUNT(n);
  LSR_REQUIRE((reinterpret_cast<uintptr_t>(a) & 7) == 0 && (reinterpret_cast<uintptr_t>(b) & 7) == 0, LSR_E_ARG,
              "complex64 arrays must be 8-byte aligned");
  hipLaunchKernelGGL(cross_power_kernel<false>, dim3(grid_for(n) * 4), dim3(kThreads), 0, lsr::as_stream(stream), a,
                     const_cast<float*>(b), n);
  return lsr::launch_status("lsr_cross_power_c64");
}

extern "C" int lsr_transpose_last2_c64(const float* src, float* dst, int64_t A, int64_t B, int64_t C,
                                       lsr_stream_t stream) {
  LSR_REQUIRE_PTR(src);
  LSR_REQUIRE_PTR(dst);
  LSR_REQUIRE(src != dst, LSR_E_ARG, "the transpose works out of place");
  LSR_REQUIRE(A > 0 && B > 0 && C > 0, LSR_E_SHAPE, "shape (%lld,%lld,%lld) must be positive", (long long)A, (long long)B,
              (long long)C);
  LSR_REQUIRE_VOLUME(A, B, C);
  LSR_REQUIRE((reinterpret_cast<uintptr_t>(src) & 7) == 0 && (reinterpret_cast<uintptr_t>(dst) & 7) == 0, LSR_E_ARG,
              "complex64 arrays must be 8-byte aligned");
  TransposeArgs p;
  p.src = reinterpret_cast<const double*>(src);
  p.dst = reinterpret_cast<double*>(dst);
  p.A = A; p.B = B; p.C = C;
  p.tiles_b = lsr::ceil_div(B, kTr);
  p.tiles_c = lsr::ceil_div(C, kTr);
  const int64_t blocks = A * p.tiles_b * p.tiles_c;
  LSR_REQUIRE(blocks < (int64_t(1) << 31), LSR_E_SHAPE, "grid of %lld workgroups is too large", (long long)blocks);
  hipLaunchKernelGGL(transpose_c64_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0, lsr::as_stream(stream), p);
  return lsr::launch_status("lsr_transpose_last2_c64");
}

extern "C" int lsr_cross_power_into_c64(const float* a, float* b, int64_t n, lsr_stream_t stream) {
  LSR_REQUIRE_PTR(a);
  LSR_REQUIRE_PTR(b);
  LSR_REQUIRE(n > 0, LSR_E_SHAPE, "n = %lld must be positive", (long long)n);
  LSR_REQUIRE_COUNT(n);
  LSR_REQUIRE((reinterpret_cast<uintptr_t>(a) & 7) == 0 && (reinterpret_cast<uintptr_t>(b) & 7) == 0, LSR_E_ARG,
              "complex64 arrays must be 8-byte aligned");
  hipLaunchKernelGGL(cross_power_kernel<true>, dim3(grid_for(n) * 4), dim3(kThreads), 0, lsr::as_stream(stream),
                     const_cast<float*>(a), b, n);
  return lsr::launch_status("lsr_cross_power_into_c64");
}

extern "C" int lsr_peak_abs_shifted_f32(const float* in, int64_t Z, int64_t Y, int64_t X, long long* out_index,
                                        void* scratch, lsr_stream_t stream) {
  if (int rc = check_volume(in, Z, Y, X)) return rc;
  LSR_REQUIRE_PTR(out_index);
  LSR_REQUIRE_PTR(scratch);
  const int64_t rows = Z * Y;
  const int nb = static_cast<int>(rows < kBlocks ? rows : kBlocks);
  PeakArgs p;
  p.in = in; p.Z = static_cast<int>(Z); p.Y = static_cast<int>(Y); p.X = static_cast<int>(X);
  p.pidx = static_cast<unsigned long long*>(scratch);                       // kBlocks * 8 bytes
  p.pval = reinterpret_cast<float*>(static_cast<char*>(scratch) + sizeof(unsigned long long) * kBlocks);
  hipStream_t s = lsr::as_stream(stream);
  hipLaunchKernelGGL(peak_partial_kernel, dim3(nb), dim3(kThreads), 0, s, p);
  hipLaunchKernelGGL(peak_final_kernel, dim3(1), dim3(kThreads), 0, s, p.pval, p.pidx, nb, out_index);
  return lsr::launch_status("lsr_peak_abs_shifted_f32");
}
