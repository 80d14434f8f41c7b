// Bright-field flat-field correction for gfx950: per-pixel median over Z, then
// out = in / median * mean(median).
//
// Replaces _LabelfreePreprocessor._flat_field_BF (shrimpy/preprocessing.py:385-404, reference root
// /root/reference):  static_pattern = volume.quantile(0.5, dim=0);  volume / static_pattern *
// static_pattern.mean().  The reference calls its earlier numpy version "a large per-FOV CPU
// bottleneck"; torch.quantile sorts every (y, x) column.
//
// Median = exact order statistics by RADIX SELECT, streaming: a 512-thread workgroup owns 128
// consecutive pixels (512-byte row runs) and passes over their Z samples at most four times, one
// key byte per pass.  Pass d histograms byte d of the (order-preserving) keys that share the
// selected prefix -- 256 bins x 128 pixels of 16-bit counters packed in pairs = 64 KB of LDS,
// updated with ds_add_u32, bank = pixel, so the adds never conflict -- and one thread per pixel
// then walks its histogram to the bin holding the wanted rank.  For an even Z the two middle ranks
// k-1, k share a prefix until, at some byte, they fall into different bins; from there on rank k-1
// is the MAXIMUM of its bin and rank k the MINIMUM of its own, so the next pass is a min/max
// reduction and the pixel is done.  No pass is ever added: <= 4 reads of the volume (2 for stacks
// of integer camera counts, whose keys are 16-bit integers -- see the kernel), nothing else
// is written but the (Y, X) pattern.  Results are the exact middle elements; the interpolation
// between them is torch's: lerp(a, b, 0.5) = b - (b - a) * 0.5 in f32.
//
// Algorithmic HBM bytes: 4 * N_in per pass, <= 4 passes (median) + 4 * N_in + 4 * N_in (apply,
// when it is not fused into the deskew kernel's staging, deskew.hip).

#include "common.hpp"

#include <type_traits>

namespace {

constexpr int kCols = 128;     // pixels per workgroup
constexpr int kZGroups = 4;    // z phases per pixel (waves 2g, 2g+1 take z = g mod 4)
constexpr int kThreads = kCols * kZGroups;
constexpr int kUnroll = 16;    // loads in flight per thread
constexpr int kReduceBlocks = 256;

enum Mode : int { kShared = 0, kSplit = 1, kDone = 2 };

__device__ __forceinline__ unsigned key_of(float v) {
  const unsigned u = __float_as_uint(v);
  return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);  // unsigned order == float order
}
__device__ __forceinline__ float value_of(unsigned k) {
  return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu));
}

struct MedianArgs {
  const void* in;   // float32, or (U16) the camera's uint16 counts
  float* pattern;
  int Z;
  int64_t plane;  // Y * X
};

template <bool U16>
__global__ __launch_bounds__(kThreads) void flat_median_kernel(MedianArgs p) {
  using raw_t = std::conditional_t<U16, unsigned short, float>;
  __shared__ unsigned hist[128 * kCols];  // [bin >> 1][pixel], two 16-bit counters per word
  __shared__ unsigned s_prefix0[kCols], s_prefix1[kCols];  // selected key prefixes (rank k0 / k1)
  __shared__ unsigned s_lo[kCols], s_hi[kCols];            // split mode: min of bin 1, max of bin 0
  __shared__ int s_r0[kCols], s_r1[kCols];                 // ranks inside the selected bin
  __shared__ int s_mode[kCols];
  __shared__ int s_nan[kCols];

  const int tid = threadIdx.x;
  const int c = tid & (kCols - 1);
  const int g = tid / kCols;
  const int64_t col0 = static_cast<int64_t>(blockIdx.x) * kCols;
  const int64_t col = min(col0 + c, p.plane - 1);  // clamped: loads stay in bounds
  const raw_t* src = static_cast<const raw_t*>(p.in) + col;
  const int Z = p.Z;

  // Camera stacks are integer counts below 65536 stored as f32: their keys are 16-bit integers
  // and the select needs two passes, not four.  The workgroup tries that when its first samples
  // look like counts and falls back to the float keys (from scratch) should any sample of its 128
  // pixels turn out not to be one.
  auto is_count = [](float v) { return v >= 0.0f && v < 65536.0f && v == truncf(v); };
  bool int_mode = U16 || __syncthreads_and(is_count(static_cast<float>(src[static_cast<int64_t>(min(g, Z - 1)) * p.plane])) ? 1 : 0) != 0;

  auto reset_state = [&]() {
    if (tid < kCols) {
      s_prefix0[tid] = 0;
      s_prefix1[tid] = 0;
      s_r0[tid] = (Z - 1) / 2;  // 0-based ranks of the two middle elements (equal for odd Z)
      s_r1[tid] = Z / 2;
      s_mode[tid] = kShared;
      s_nan[tid] = 0;
    }
  };
  reset_state();

  for (int level = int_mode ? 2 : 0; level < 4; ++level) {
    const bool first_pass = level == (int_mode ? 2 : 0);
    const int shift = 24 - 8 * level;
    for (int i = tid; i < 128 * kCols; i += kThreads) hist[i] = 0;
    if (tid < kCols) {
      s_lo[tid] = 0xffffffffu;
      s_hi[tid] = 0;
    }
    __syncthreads();
    const int mode = s_mode[c];
    const unsigned pre0 = s_prefix0[c], pre1 = s_prefix1[c];
    // The streaming loop is VALU-bound (every sample: key, prefix test, bin, LDS add), so it is
    // specialised per pass kind, walks the column by pointer increments and keeps the bounds
    // check out of the full batches.
    bool saw_nan = false, not_count = false;
    unsigned vmax = 0, vmin = 0xffffffffu;
    // first pass: the top key byte (sign + 7 exponent bits) of a pixel hardly changes along z, so
    // runs of equal bins are counted in registers and cost one LDS add each (the LDS atomics, not
    // HBM, bound this kernel: one ds_add per sample was ~8 ms at config 2)
    unsigned run_bin = 0xffffffffu, run_len = 0;
    auto flush_run = [&]() {
      if (run_len) atomicAdd(&hist[(run_bin >> 1) * kCols + c], run_len << (16 * (run_bin & 1)));
    };
    auto visit = [&](float v, auto kind) {
      constexpr int KIND = decltype(kind)::value;  // 0: first pass, 1: later histogram pass, 2: min/max pass
      const unsigned key = int_mode ? static_cast<unsigned>(v) : key_of(v);
      if constexpr (KIND == 0) {
        saw_nan |= v != v;
        not_count |= !is_count(v);
        const unsigned bin = (key >> shift) & 255u;  // (the higher bytes are all zero / all wanted)
        if (bin == run_bin) {
          ++run_len;
        } else {
          flush_run();
          run_bin = bin;
          run_len = 1;
        }
      } else if constexpr (KIND == 1) {
        if ((key >> (shift + 8)) == pre0) {
          const unsigned bin = (key >> shift) & 255u;
          atomicAdd(&hist[(bin >> 1) * kCols + c], 1u << (16 * (bin & 1)));
        }
      } else {
        const unsigned top = key >> (shift + 8);
        if (top == pre0) vmax = max(vmax, key);
        if (top == pre1) vmin = min(vmin, key);
      }
    };
    auto stream = [&](auto kind) {
      const int64_t step = static_cast<int64_t>(kZGroups) * p.plane;
      const raw_t* q = src + static_cast<int64_t>(g) * p.plane;
      int z = g;
      for (; z + (kUnroll - 1) * kZGroups < Z; z += kUnroll * kZGroups) {  // full batches
        float v[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) v[u] = static_cast<float>(q[u * step]);
        q += kUnroll * step;
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) visit(v[u], kind);
      }
      for (; z < Z; z += kZGroups, q += step) visit(static_cast<float>(*q), kind);
    };
    if (mode == kShared) {
      if (first_pass) {
        stream(std::integral_constant<int, 0>{});
        flush_run();
      } else {
        stream(std::integral_constant<int, 1>{});
      }
    } else if (mode == kSplit) {
      stream(std::integral_constant<int, 2>{});
      atomicMax(&s_hi[c], vmax);
      atomicMin(&s_lo[c], vmin);
    }
    if (first_pass && !int_mode && saw_nan) s_nan[c] = 1;
    if (first_pass && int_mode) {
      if (__syncthreads_or(not_count ? 1 : 0)) {  // not a stack of counts after all: float keys, from scratch
        int_mode = false;
        reset_state();
        level = -1;
        continue;
      }
    } else {
      __syncthreads();
    }
    if (tid < kCols && s_mode[tid] != kDone) {
      if (s_mode[tid] == kSplit) {
        s_prefix0[tid] = s_hi[tid];  // full keys now
        s_prefix1[tid] = s_lo[tid];
        s_mode[tid] = kDone;
      } else {
        // walk the histogram: bins of rank r0 and rank r1
        int r0 = s_r0[tid], r1 = s_r1[tid];
        int b0 = -1, b1 = -1, cum = 0;
        for (int w = 0; w < 128 && b1 < 0; ++w) {
          const unsigned word = hist[w * kCols + tid];
          const int n_lo = static_cast<int>(word & 0xffffu), n_hi = static_cast<int>(word >> 16);
          if (b0 < 0 && r0 < cum + n_lo) { b0 = 2 * w; r0 -= cum; }
          if (b1 < 0 && r1 < cum + n_lo) { b1 = 2 * w; r1 -= cum; }
          cum += n_lo;
          if (b0 < 0 && r0 < cum + n_hi) { b0 = 2 * w + 1; r0 -= cum; }
          if (b1 < 0 && r1 < cum + n_hi) { b1 = 2 * w + 1; r1 -= cum; }
          cum += n_hi;
        }
        const unsigned base = s_prefix0[tid] << 8;
        s_prefix0[tid] = base | static_cast<unsigned>(b0);
        s_prefix1[tid] = base | static_cast<unsigned>(b1);
        s_r0[tid] = r0;
        s_r1[tid] = r1;
        if (level == 3) s_mode[tid] = kDone;           // the bins are whole keys
        else if (b0 != b1) s_mode[tid] = kSplit;       // finish with one min / max pass
      }
    }
    __syncthreads();
  }

  if (tid < kCols && col0 + tid < p.plane) {
    const float a = int_mode ? static_cast<float>(s_prefix0[tid]) : value_of(s_prefix0[tid]);
    const float b = int_mode ? static_cast<float>(s_prefix1[tid]) : value_of(s_prefix1[tid]);
    // torch.quantile(0.5): lerp(a, b, w) with w = 0.5 for an even count (the branch of
    // at::native::lerp for |w| >= 0.5), w = 0 for an odd one
    float med = (Z & 1) ? a : b - (b - a) * 0.5f;
    if (s_nan[tid]) med = __uint_as_float(0x7fc00000u);  // quantile propagates NaN
    p.pattern[col0 + tid] = med;
  }
}

// mean(pattern): deterministic two-stage reduction in f64
__global__ __launch_bounds__(256) void flat_sum_kernel(const float* __restrict__ pattern, int64_t n,
                                                       double* __restrict__ partial) {
  __shared__ double red[256];
  double s = 0.0;
  const int64_t per = (n + gridDim.x - 1) / gridDim.x;
  const int64_t lo = per * blockIdx.x, hi = min(lo + per, n);
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) s += static_cast<double>(pattern[i]);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (static_cast<int>(threadIdx.x) < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void flat_mean_kernel(const double* __restrict__ partial, int nb,
                                                        int64_t n, float* __restrict__ mean_out) {
  __shared__ double red[256];
  red[threadIdx.x] = static_cast<int>(threadIdx.x) < nb ? partial[threadIdx.x] : 0.0;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (static_cast<int>(threadIdx.x) < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) mean_out[0] = static_cast<float>(red[0] / static_cast<double>(n));
}

// out = in / pattern * mean, four voxels per thread
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <typename raw_t>
__global__ __launch_bounds__(256) void flat_apply_any_kernel(const raw_t* __restrict__ in,
                                                             const float* __restrict__ pattern,
                                                             const float* __restrict__ mean_dev,
                                                             float* __restrict__ out, int64_t plane,
                                                             int64_t total) {
  const float mean = mean_dev[0];
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += stride)
    out[i] = static_cast<float>(in[i]) / pattern[i % plane] * mean;
}

__global__ __launch_bounds__(256) void flat_apply_kernel(const float* __restrict__ in,
                                                         const float* __restrict__ pattern,
                                                         const float* __restrict__ mean_dev,
                                                         float* __restrict__ out, int64_t plane,
                                                         int64_t total) {
  const float mean = mean_dev[0];
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x * 4;
  for (int64_t i = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) * 4; i < total;
       i += stride) {
    if (i + 3 < total && (plane & 3) == 0) {  // rows of a plane stay 16-byte aligned
      const f32x4 v = *reinterpret_cast<const f32x4*>(in + i);
      const f32x4 q = *reinterpret_cast<const f32x4*>(pattern + i % plane);
      f32x4 r;
      r.x = v.x / q.x * mean; r.y = v.y / q.y * mean; r.z = v.z / q.z * mean; r.w = v.w / q.w * mean;
      *reinterpret_cast<f32x4*>(out + i) = r;
    } else {
      for (int64_t j = i; j < min(i + 4, total); ++j) out[j] = in[j] / pattern[j % plane] * mean;
    }
  }
}

}  // namespace

extern "C" int lsr_flatfield_scratch_bytes(void) { return kReduceBlocks * static_cast<int>(sizeof(double)); }

namespace {
int flatfield_pattern(const char* what, const void* in, bool u16, int64_t Z, int64_t Y, int64_t X,
                      float* pattern, float* mean_out, void* scratch, lsr_stream_t stream) {
  LSR_REQUIRE_PTR(in);
  LSR_REQUIRE_PTR(pattern);
  LSR_REQUIRE_PTR(mean_out);
  LSR_REQUIRE_PTR(scratch);
  LSR_REQUIRE(Z > 0 && Y > 0 && X > 0, LSR_E_SHAPE, "shape (%lld,%lld,%lld) must be positive",
              (long long)Z, (long long)Y, (long long)X);
  LSR_REQUIRE_VOLUME(Z, Y, X);
  LSR_REQUIRE(Z < 65536, LSR_E_UNSUPPORTED, "Z = %lld: the per-pixel histograms count in 16 bits",
              (long long)Z);
  const int64_t plane = Y * X;
  const int64_t tiles = lsr::ceil_div(plane, static_cast<int64_t>(kCols));
  LSR_REQUIRE(tiles < (int64_t(1) << 31), LSR_E_SHAPE, "plane of %lld pixels is too large",
              (long long)plane);
  hipStream_t s = lsr::as_stream(stream);
  MedianArgs p{in, pattern, static_cast<int>(Z), plane};
  if (u16) hipLaunchKernelGGL(flat_median_kernel<true>, dim3(static_cast<unsigned>(tiles)), dim3(kThreads), 0, s, p);
  else hipLaunchKernelGGL(flat_median_kernel<false>, dim3(static_cast<unsigned>(tiles)), dim3(kThreads), 0, s, p);
  double* partial = static_cast<double*>(scratch);
  const int nb = static_cast<int>(plane < kReduceBlocks ? plane : kReduceBlocks);
  hipLaunchKernelGGL(flat_sum_kernel, dim3(nb), dim3(256), 0, s, pattern, plane, partial);
  hipLaunchKernelGGL(flat_mean_kernel, dim3(1), dim3(256), 0, s, partial, nb, plane, mean_out);
  return lsr::launch_status(what);
}
}  // namespace

extern "C" int lsr_flatfield_pattern_f32(const float* in, int64_t Z, int64_t Y, int64_t X,
                                         float* pattern, float* mean_out, void* scratch,
                                         lsr_stream_t stream) {
  return flatfield_pattern("lsr_flatfield_pattern_f32", in, false, Z, Y, X, pattern, mean_out, scratch, stream);
}

extern "C" int lsr_flatfield_pattern_u16(const uint16_t* in, int64_t Z, int64_t Y, int64_t X,
                                         float* pattern, float* mean_out, void* scratch,
                                         lsr_stream_t stream) {
  return flatfield_pattern("lsr_flatfield_pattern_u16", in, true, Z, Y, X, pattern, mean_out, scratch, stream);
}

extern "C" int lsr_flatfield_apply_u16(const uint16_t* in, const float* pattern, const float* mean_dev,
                                       float* out, int64_t Z, int64_t Y, int64_t X, lsr_stream_t stream) {
  LSR_REQUIRE_PTR(in);
  LSR_REQUIRE_PTR(pattern);
  LSR_REQUIRE_PTR(mean_dev);
  LSR_REQUIRE_PTR(out);
  LSR_REQUIRE(Z > 0 && Y > 0 && X > 0, LSR_E_SHAPE, "shape (%lld,%lld,%lld) must be positive",
              (long long)Z, (long long)Y, (long long)X);
  LSR_REQUIRE_VOLUME(Z, Y, X);
  const int64_t plane = Y * X, total = Z * plane;
  int64_t blocks = lsr::ceil_div(total, static_cast<int64_t>(256));
  if (blocks > 256 * 64) blocks = 256 * 64;
  hipLaunchKernelGGL(flat_apply_any_kernel<unsigned short>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0,
                     lsr::as_stream(stream), in, pattern, mean_dev, out, plane, total);
  return lsr::launch_status("lsr_flatfield_apply_u16");
}

extern "C" int lsr_flatfield_apply_f32(const float* in, const float* pattern, const float* mean_dev,
                                       float* out, int64_t Z, int64_t Y, int64_t X,
                                       lsr_stream_t stream) {
  LSR_REQUIRE_PTR(in);
  LSR_REQUIRE_PTR(pattern);
  LSR_REQUIRE_PTR(mean_dev);
  LSR_REQUIRE_PTR(out);
  LSR_REQUIRE(Z > 0 && Y > 0 && X > 0, LSR_E_SHAPE, "shape (%lld,%lld,%lld) must be positive",
              (long long)Z, (long long)Y, (long long)X);
  LSR_REQUIRE_VOLUME(Z, Y, X);
  const int64_t plane = Y * X, total = Z * plane;
  int64_t blocks = lsr::ceil_div(total, static_cast<int64_t>(256 * 4));
  if (blocks > 256 * 64) blocks = 256 * 64;
  hipLaunchKernelGGL(flat_apply_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0,
                     lsr::as_stream(stream), in, pattern, mean_dev, out, plane, total);
  return lsr::launch_status("lsr_flatfield_apply_f32");
}
