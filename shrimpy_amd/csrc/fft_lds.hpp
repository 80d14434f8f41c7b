// Complex FFTs of LDS-resident sequences: Stockham autosort passes of radix 2 / 3 / 4 / 5, in place
// (a thread reads all inputs of its butterflies into registers before the barrier and writes all outputs
// after it).  Shared by the kernels that transform short or contiguous axes themselves instead of handing
// them to rocFFT together with the passes around them (zcorr.hip, rfft_rows.hip).  Forward sign; an inverse
// is conj(FFT(conj(v))).
#pragma once

#include "common.hpp"

namespace lsr_fft {

constexpr int kMaxFactors = 12;

struct Factors {
  int n;                           // number of passes
  int radix[kMaxFactors];          // each 2, 3, 4 or 5
  float inv_stride[kMaxFactors];   // 1 / s of each pass (s = product of the radices before it)
};

// radix-4 passes first, then what is left of the twos, threes and fives; false if n is not 5-smooth
inline bool factorize(int64_t n, Factors* f) {
  if (n < 1) return false;
  int nf = 0;
  while (n % 4 == 0 && nf < kMaxFactors) { f->radix[nf++] = 4; n /= 4; }
  for (int r : {2, 3, 5})
    while (n % r == 0 && nf < kMaxFactors) { f->radix[nf++] = r; n /= r; }
  if (n != 1) return false;
  f->n = nf;
  for (int i = 0, stride = 1; i < nf; ++i) {
    f->inv_stride[i] = 1.0f / static_cast<float>(stride);
    stride *= f->radix[i];
  }
  return true;
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return float2{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return float2{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return float2{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ float2 cconj(float2 a) { return float2{a.x, -a.y}; }
__device__ __forceinline__ float2 mul_mi(float2 a) { return float2{a.y, -a.x}; }   // a * (-i)
__device__ __forceinline__ float2 mul_i(float2 a) { return float2{-a.y, a.x}; }    // a * i

// A strided loop whose iterations begin with a global load: U loads are issued before the first one is used.  Written
// as a plain loop, hipcc waits for each iteration's load before it issues the next -- ONE load in flight per wave, and
// the kernels around these transforms spent most of their time in exactly that (round 4: s_waitcnt vmcnt(0) inside every
// tile-load, twiddle and epilogue loop of rfft_rows.hip / zcorr.hip).  `load(j)` must be valid for every j the loop
// visits; the unused slots of the last batch re-read its last index.
template <int U, typename L, typename S>
__device__ __forceinline__ void batched_loop(int first, int end, int step, L&& load, S&& use) {
  for (int i = first; i < end; i += U * step) {
    decltype(load(0)) v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = i + u * step;
      v[u] = load(j < end ? j : end - 1);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = i + u * step;
      if (j < end) use(j, v[u]);
    }
  }
}

// DFT of R points, forward sign
template <int R>
__device__ __forceinline__ void dft(float2 (&a)[R]) {
  if constexpr (R == 2) {
    const float2 t = a[0];
    a[0] = cadd(t, a[1]);
    a[1] = csub(t, a[1]);
  } else if constexpr (R == 3) {
    constexpr float c = -0.5f, s = -0.86602540378443864676f;   // exp(-2 pi i / 3) = c + i s
    const float2 t1 = cadd(a[1], a[2]), t2 = csub(a[1], a[2]);
    const float2 m = float2{a[0].x + c * t1.x, a[0].y + c * t1.y};
    const float2 r = float2{-s * t2.y, s * t2.x};               // i s t2
    a[0] = cadd(a[0], t1);
    a[1] = cadd(m, r);
    a[2] = csub(m, r);
  } else if constexpr (R == 4) {
    const float2 s0 = cadd(a[0], a[2]), d0 = csub(a[0], a[2]);
    const float2 s1 = cadd(a[1], a[3]), d1 = mul_mi(csub(a[1], a[3]));
    a[0] = cadd(s0, s1);
    a[1] = cadd(d0, d1);
    a[2] = csub(s0, s1);
    a[3] = csub(d0, d1);
  } else {
    static_assert(R == 5, "radix");
    constexpr float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;    // cos(2 pi k / 5)
    constexpr float s1 = -0.95105651629515357212f, s2 = -0.58778525229247312917f;   // -sin(2 pi k / 5)
    const float2 t1 = cadd(a[1], a[4]), t2 = cadd(a[2], a[3]);
    const float2 t3 = csub(a[1], a[4]), t4 = csub(a[2], a[3]);
    const float2 m1 = float2{a[0].x + c1 * t1.x + c2 * t2.x, a[0].y + c1 * t1.y + c2 * t2.y};
    const float2 m2 = float2{a[0].x + c2 * t1.x + c1 * t2.x, a[0].y + c2 * t1.y + c1 * t2.y};
    // i * (s1 t3 + s2 t4), i * (s2 t3 - s1 t4)
    const float2 u1 = float2{-(s1 * t3.y + s2 * t4.y), s1 * t3.x + s2 * t4.x};
    const float2 u2 = float2{-(s2 * t3.y - s1 * t4.y), s2 * t3.x - s1 * t4.x};
    a[0] = cadd(a[0], cadd(t1, t2));
    a[1] = cadd(m1, u1);
    a[4] = csub(m1, u1);
    a[2] = cadd(m2, u2);
    a[3] = csub(m2, u2);
  }
}

// Ordering between a pass's reads and its writes, and between passes.  The PERCOL threads that share a
// sequence always sit in ONE wavefront (PERCOL divides 64 and sequences are dealt to consecutive threads), a
// wavefront's LDS instructions execute in order, and nobody else touches the sequence during a transform:
// a wavefront-scope fence (it only keeps the compiler from moving LDS accesses across it and waits for the
// wave's own outstanding ones) replaces the workgroup barrier -- sixteen to twenty-four barriers per
// transform pair were what these kernels spent most of their cycles in (SQ counters: waves issuing 20-30 %
// of the time, 60 % idle).
//
// The fence is sequentially consistent: pass k's stores followed by pass k + 1's loads BY OTHER LANES is a
// store -> load ordering, which acquire/release alone does not promise; and the wave barrier keeps the
// compiler from scheduling any lane's code across the point (it emits no instruction on a wave64 target).
// Both rest on the 64-lane wavefront: a wave32 build would split a sequence over two waves.
// (ROCm 7 dropped the __AMDGCN_WAVEFRONT_SIZE macro; gfx950 has no wave32 mode, so the target is the check.)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "the LDS transforms assume 64-lane wavefronts: build for gfx950 only"
#endif
__device__ __forceinline__ void sequence_sync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// One Stockham pass (decimation in frequency) over the sequence `seq` of length N, shared by PERCOL
// threads of one wavefront (thread t takes butterflies t, t + PERCOL, ...).
//   x[q + s (p + m k)]  ->  y[q + s (R p + j)] = (sum_k x_k w_R^{jk}) * w_n^{p j},   n = R m, 0 <= p < m, 0 <= q < s
// `tw(i)` = exp(-2 pi i / N * i), 0 <= i < N.
template <int R, int MAXN, int PERCOL, typename TW>
__device__ __forceinline__ void stockham_pass(float2* seq, int N, int n, int s, float inv_s, TW&& tw, int t) {
  static_assert(PERCOL <= 64 && 64 % PERCOL == 0, "the threads of a sequence must share a wavefront");
  const int m = n / R;
  const int per_seq = N / R;                       // butterflies per sequence (m * s)
  constexpr int kMaxBf = (MAXN / R + PERCOL - 1) / PERCOL;
  float2 a[kMaxBf][R];
#pragma unroll
  for (int i = 0; i < kMaxBf; ++i) {
    const int r = t + i * PERCOL;
    if (r < per_seq) {
      // r / s and r % s: exact in float for r <= 1024, s <= 1024 (the quotient is at least 0.5 / s off an integer)
      const int p = static_cast<int>((static_cast<float>(r) + 0.5f) * inv_s), q = r - p * s;
      const float2* x = seq + q + s * p;
#pragma unroll
      for (int k = 0; k < R; ++k) a[i][k] = x[s * m * k];
    }
  }
  sequence_sync();
#pragma unroll
  for (int i = 0; i < kMaxBf; ++i) {
    const int r = t + i * PERCOL;
    if (r < per_seq) {
      const int p = static_cast<int>((static_cast<float>(r) + 0.5f) * inv_s), q = r - p * s;
      dft<R>(a[i]);
      float2* y = seq + q + s * R * p;
      y[0] = a[i][0];
      const int step = p * s;                       // w_n^p = w_N^(p s); step * j < N
#pragma unroll
      for (int j = 1; j < R; ++j) y[s * j] = cmul(a[i][j], tw(step * j));
    }
  }
  sequence_sync();
}

template <int MAXN, int PERCOL, typename TW>
__device__ __forceinline__ void transform(float2* seq, int N, const Factors& f, TW&& tw, int t) {
  int n = N, s = 1;
  for (int i = 0; i < f.n; ++i) {
    const int r = f.radix[i];
    const float inv_s = f.inv_stride[i];
    if (r == 4) stockham_pass<4, MAXN, PERCOL>(seq, N, n, s, inv_s, tw, t);
    else if (r == 2) stockham_pass<2, MAXN, PERCOL>(seq, N, n, s, inv_s, tw, t);
    else if (r == 3) stockham_pass<3, MAXN, PERCOL>(seq, N, n, s, inv_s, tw, t);
    else stockham_pass<5, MAXN, PERCOL>(seq, N, n, s, inv_s, tw, t);
    n /= r;
    s *= r;
  }
}

}  // namespace lsr_fft
