// The z leg of the tracker's cross-correlation in one pass (reference
// shrimpy/dynatrack/tracking.py:309-378: corr = irfftn(rfftn(ref) * conj(rfftn(mov)))).
//
// shrimpy_amd/fft3.py runs the 3-D transforms axis by axis.  Around the cross power the moving volume's
// spectrum went: transpose to z-contiguous, forward transform along z, product with the reference's
// spectrum, inverse transform along z, transpose back -- five passes over 3.4 GB each (8.7 of the 22 ms a
// cached-reference correlation took).  The z axis of a deskewed volume is short (171 -> 180, 86 -> 90,
// 67 -> 72: tracking.py:248-263 pads to 5-smooth lengths), so a whole column fits in LDS many times over:
// this kernel loads a tile of 16 neighbouring y columns x all z straight from the [Z][XC][Y] layout the y
// transform leaves (128-byte runs), transforms every column along z in LDS (Stockham autosort, radix
// 2 / 3 / 4 / 5 passes from a host-made factor list), multiplies by the conjugate-arranged reference
// spectrum (kept fully transformed in [XC][Y][Z], contiguous along z), transforms back and stores the tile
// where it came from.  One read of each operand, one write: 10 GB instead of 34 GB.
//
// out = Z * IFFT_z( F1 * conj( FFT_z(g) ) ), unnormalised like the rest of fft3 (the consumer takes an
// argmax).  The inverse is done with the forward machinery: IFFT(v) = conj(FFT(conj(v))) / Z.

#include "fft_lds.hpp"

#include <cstdlib>

namespace {

using namespace lsr_fft;

constexpr int kThreads = 256;
constexpr int kCols = 16;          // y columns per workgroup: 128-byte runs of complex64 (32 columns x 512 threads: 3.50 against 3.32 ms)
constexpr int kPerCol = kThreads / kCols;   // threads that share a column's butterflies
constexpr int kMaxN = 256;         // longest z transform: kCols * (kMaxN + 1) * 8 bytes = 33 KB of LDS

struct ZcorrArgs {
  const float2* f1;      // [XC][Y][N]  reference spectrum, fully transformed
  float2* g;             // [N][XC][Y]  moving spectrum after the x and y transforms; overwritten
  const float2* tw;      // [N] exp(-2 pi i k / N)
  int N, Y, XC;
  Factors f;
  float inv_n;                     // 1 / N
  int mode;                        // 0: f1 * conj(G) (cross-correlation), 1: f1 * G, 2: conj(f1) * G (spectrum products)
  int z_valid, z_keep;             // planes [z_valid, N) of g are zeros (not read); only planes [0, z_keep) are stored
};

__global__ __launch_bounds__(kThreads) void zcorr_kernel(ZcorrArgs p) {
  extern __shared__ float2 smem[];
  const int N = p.N, pitch = N + 1;
  float2* buf = smem;                 // [kCols][pitch]
  float2* tw = smem + kCols * pitch;  // [N]
  const int tid = threadIdx.x;
  const int tiles_y = (p.Y + kCols - 1) / kCols;
  const int xc = blockIdx.x / tiles_y, y0 = (blockIdx.x - xc * tiles_y) * kCols;
  const int ncols = min(kCols, p.Y - y0);

  for (int k = tid; k < N; k += kThreads) tw[k] = p.tw[k];
  // tile of g: rows z, kCols consecutive y (columns past Y: zeros, never stored)
  const int c = tid & (kCols - 1), z0 = tid / kCols;
  const int64_t row = static_cast<int64_t>(p.XC) * p.Y;
  const float2* g = p.g + static_cast<int64_t>(xc) * p.Y + y0 + c;
  // (batched: eleven dependent load -> LDS-store rounds per thread otherwise, one load in flight per wave)
  if (c < ncols) {
    float2* col = buf + c * pitch;
    batched_loop<8>(z0, p.z_valid, kThreads / kCols, [g, row](int z) { return g[z * row]; }, [col](int z, float2 v) { col[z] = v; });
    for (int z = z0 + ((p.z_valid - z0 + kThreads / kCols - 1) / (kThreads / kCols)) * (kThreads / kCols); z < N; z += kThreads / kCols)
      col[z] = float2{0.0f, 0.0f};      // zero padding behind the volume: known, not read
  } else {
    for (int z = z0; z < N; z += kThreads / kCols) buf[c * pitch + z] = float2{0.0f, 0.0f};
  }
  __syncthreads();

  float2* const my_col = buf + (tid / kPerCol) * pitch;      // the column this thread works on in the passes
  const int t = tid & (kPerCol - 1);
  const float2* f1 = p.f1 + (static_cast<int64_t>(xc) * p.Y + y0) * N;
  // round 0: G = FFT_z(g), then v = conj(F1 * conj(G)) = conj(F1) * G;  round 1: FFT(v) = conj(N * IFFT(F1 conj(G)))
  // (requesting the reference's values before the first transform and holding them in registers was tried:
  // 153 VGPRs, one workgroup per CU, 6.4 ms instead of 3.5)
  // Every column belongs to the sixteen threads that transform it (one quarter of a wavefront): the product
  // with the reference is done by those threads too, so nothing between the load and the store needs a barrier.
  const int my = tid / kPerCol;
  for (int round = 0; round < 2; ++round) {
    transform<kMaxN, kPerCol>(my_col, N, p.f, [tw](int i) { return tw[i]; }, t);
    if (round == 0) {
      if (my < ncols) {
        const float2* ref = f1 + static_cast<int64_t>(my) * N;
        // (what the second transform needs is conj(v) of the product v wanted: conj(FFT(conj(v))) = N IFFT(v))
        const int mode = p.mode;
        batched_loop<8>(t, N, kPerCol, [ref](int k) { return ref[k]; }, [my_col, mode](int k, float2 f) {
          if (mode == 0) my_col[k] = cmul(cconj(f), my_col[k]);            // v = f1 conj(G)
          else if (mode == 1) my_col[k] = cconj(cmul(f, my_col[k]));       // v = f1 G
          else my_col[k] = cmul(f, cconj(my_col[k]));                      // v = conj(f1) G
        });
      }
      sequence_sync();
    }
  }
  __syncthreads();            // the store below takes columns across wavefronts

  float2* out = p.g + static_cast<int64_t>(xc) * p.Y + y0 + c;
  if (c < ncols)
    for (int z = z0; z < p.z_keep; z += kThreads / kCols) out[z * row] = cconj(buf[c * pitch + z]);
}

}  // namespace

// 1 when lsr_cross_correlate_z_c64 handles a z transform of length n (5-smooth, <= 256)
extern "C" int lsr_cross_correlate_z_supported(int64_t n) {
  if (n < 2 || n > kMaxN) return 0;
  const size_t lds = (static_cast<size_t>(kCols) * (n + 1) + n) * sizeof(float2);
  for (int f : {2, 3, 5})
    while (n % f == 0) n /= f;
  return n == 1 && lsr::lds_fits(lds);
}

namespace {
int z_leg(const float* f1, float* g, const float* twiddles, int64_t N, int64_t Y, int64_t XC, int mode, int64_t z_valid,
          int64_t z_keep, lsr_stream_t stream);
}

extern "C" int lsr_cross_correlate_z_c64(const float* f1, float* g, const float* twiddles, int64_t N, int64_t Y, int64_t XC,
                                         lsr_stream_t stream) {
  return z_leg(f1, g, twiddles, N, Y, XC, 0, N, N, stream);
}

// g <- N * IFFT_z( f1 * FFT_z(g) )  (conj_f1 = 0)  or  N * IFFT_z( conj(f1) * FFT_z(g) )  (conj_f1 = 1): the z leg of a
// convolution / correlation with the volume behind f1 done in the Fourier domain (deconvolve_fft.py); layouts as
// lsr_cross_correlate_z_c64.
// z_valid: planes [z_valid, N) of g are the zero padding behind the volume -- taken as zeros, never read (the forward x
// and y legs need not produce them); z_keep: only planes [0, z_keep) of the result are stored (the inverse y and x legs
// read no others).  N, N = everything.
extern "C" int lsr_spectrum_multiply_z_c64(const float* f1, float* g, const float* twiddles, int64_t N, int64_t Y,
                                           int64_t XC, int conj_f1, int64_t z_valid, int64_t z_keep, lsr_stream_t stream) {
  LSR_REQUIRE(conj_f1 == 0 || conj_f1 == 1, LSR_E_ARG, "conj_f1 must be 0 or 1, got %d", conj_f1);
  LSR_REQUIRE(z_valid >= 1 && z_valid <= N && z_keep >= 1 && z_keep <= N, LSR_E_ARG,
              "z_valid %lld and z_keep %lld must lie in [1, N = %lld]", (long long)z_valid, (long long)z_keep, (long long)N);
  return z_leg(f1, g, twiddles, N, Y, XC, conj_f1 ? 2 : 1, z_valid, z_keep, stream);
}

namespace {
int z_leg(const float* f1, float* g, const float* twiddles, int64_t N, int64_t Y, int64_t XC, int mode, int64_t z_valid,
          int64_t z_keep, lsr_stream_t stream) {
  LSR_REQUIRE_PTR(f1);
  LSR_REQUIRE_PTR(g);
  LSR_REQUIRE_PTR(twiddles);
  LSR_REQUIRE(N > 0 && Y > 0 && XC > 0, LSR_E_SHAPE, "shape (%lld,%lld,%lld) must be positive", (long long)N, (long long)Y,
              (long long)XC);
  LSR_REQUIRE_VOLUME(N, Y, XC);
  LSR_REQUIRE(lsr_cross_correlate_z_supported(N), LSR_E_UNSUPPORTED,
              "z length %lld: this kernel transforms 5-smooth lengths from 2 to %d", (long long)N, kMaxN);
  LSR_REQUIRE((reinterpret_cast<uintptr_t>(f1) & 7) == 0 && (reinterpret_cast<uintptr_t>(g) & 7) == 0 &&
                  (reinterpret_cast<uintptr_t>(twiddles) & 7) == 0, LSR_E_ARG, "complex64 arrays must be 8-byte aligned");
  ZcorrArgs p;
  p.f1 = reinterpret_cast<const float2*>(f1);
  p.g = reinterpret_cast<float2*>(g);
  p.tw = reinterpret_cast<const float2*>(twiddles);
  p.N = static_cast<int>(N); p.Y = static_cast<int>(Y); p.XC = static_cast<int>(XC);
  LSR_REQUIRE(factorize(N, &p.f), LSR_E_UNSUPPORTED, "z length %lld has more than %d factors", (long long)N, kMaxFactors);
  if (const char* e = std::getenv("LSR_ZCORR_ORDER")) {   // measurement override: the radices in another order, e.g. "5,3,3,4"
    int f[kMaxFactors], k = 0;
    long long prod = 1;
    for (const char* c = e; *c && k < kMaxFactors;) {
      const int v = std::atoi(c);
      if (v != 2 && v != 3 && v != 4 && v != 5) { k = 0; break; }
      f[k++] = v;
      prod *= v;
      while (*c && *c != ',') ++c;
      if (*c == ',') ++c;
    }
    if (k > 0 && prod == N) {
      p.f.n = k;
      for (int i = 0, stride = 1; i < k; ++i) {
        p.f.radix[i] = f[i];
        p.f.inv_stride[i] = 1.0f / static_cast<float>(stride);
        stride *= f[i];
      }
    }
  }
  p.inv_n = 1.0f / static_cast<float>(p.N);
  p.mode = mode;
  p.z_valid = static_cast<int>(z_valid); p.z_keep = static_cast<int>(z_keep);
  const int64_t blocks = XC * lsr::ceil_div(Y, kCols);
  LSR_REQUIRE(blocks < (int64_t(1) << 31), LSR_E_SHAPE, "grid of %lld workgroups is too large", (long long)blocks);
  const size_t lds = (static_cast<size_t>(kCols) * (p.N + 1) + p.N) * sizeof(float2);
  static std::atomic<uint64_t> lds_allowed{0};
  if (int rc = lsr::allow_dynamic_lds(reinterpret_cast<const void*>(zcorr_kernel),
                                      static_cast<int>((kCols * (kMaxN + 1) + kMaxN) * sizeof(float2)), lds_allowed,
                                      "lsr_cross_correlate_z_c64"))
    return rc;
  hipLaunchKernelGGL(zcorr_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), lds, lsr::as_stream(stream), p);
  return lsr::launch_status("lsr_cross_correlate_z_c64");
}
}  // namespace
