// 1-D correlation along z with up to 31 taps and the Richardson-Lucy epilogues: the z half of a separable PSF whose axial
// extent exceeds the 15 taps the tiled kernels hold in registers next to their in-plane passes.  A measured axial PSF on
// 0.17 um z-voxels easily spans +-10 planes (missing 5); such a PSF runs each correlation as two launches --
// the in-plane factors through correlate_sep.hip with one z tap, then this kernel:
//
//     c[z, y, x] = sum_a w[a] * in[z + a - pz/2, y, x]       (0 outside the volume)
//     out        = epilogue(c, aux)                          (LSR_EPI_NONE / RATIO / UPDATE as everywhere)
//
// It is a pure stream: a thread owns four consecutive x of one row and marches along z, keeping the PZ pending outputs
// in registers (acc[j] <-> z_out = zi - pz/2 + j after plane zi is absorbed; the shift is folded into the FMAs), one
// 16-byte load and one 16-byte store per plane, the next plane requested before this one's FMAs.  12 algorithmic bytes
// per voxel (in, aux, out); no LDS, no barrier.  `in` needs no halo (bounds are tested: the planes a chunk reads beyond
// the volume are zeros) and any strides; rows may be padded (pitch >= X).
//
// No reference code: docs/data_structure.md:58-62 ("algorithms for deconvolution ... are being developed").

#include "common.hpp"
#include "correlate_common.hpp"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxZTaps = 31;

struct ZArgs {
  const float* in;
  const float* aux;
  float* out;
  int64_t in_plane, aux_plane, out_plane;
  int in_pitch, aux_pitch, out_pitch;
  int Z, Y, X;
  const float* wz;   // pz taps (device)
  int pz;
  float eps;
  const float* nz;   // UPDATE: norm = nz[z] * ny[y] * nx[x]
  const float* ny;
  const float* nx;
  int z_chunk;
  int x_groups;      // ceil(X / 4)
  double* stats;     // UPDATE: the launch's RL scalars (correlate_common.hpp) or NULL
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PZ, int EPI>
__global__ __launch_bounds__(kThreads) void correlate_z_kernel(ZArgs p) {
  constexpr int CZ = PZ / 2;
  const int g = blockIdx.x * kThreads + threadIdx.x;   // group of four x
  const int y = blockIdx.y;
  const int zb = blockIdx.z * p.z_chunk, ze = min(zb + p.z_chunk, p.Z);
  const int x = 4 * g;
  const bool live = g < p.x_groups;
  const int n_valid = live ? min(4, p.X - x) : 0;      // a ragged last group: lanes past X neither load nor store
  lsr::RlStats st;
  if (live) {
    float w[PZ];
#pragma unroll
    for (int a = 0; a < PZ; ++a) w[a] = p.wz[a];
    const float* in = p.in + static_cast<int64_t>(y) * p.in_pitch + x;
    const float* aux = EPI == LSR_EPI_NONE ? nullptr : p.aux + static_cast<int64_t>(y) * p.aux_pitch + x;
    float* out = p.out + static_cast<int64_t>(y) * p.out_pitch + x;
    // rows of a 16-byte-aligned, 4-multiple-pitch volume take vector accesses; anything else goes element by element
    auto aligned16 = [&](const float* base, int64_t plane) {
      return n_valid == 4 && ((reinterpret_cast<uintptr_t>(base) | (static_cast<uintptr_t>(plane) * 4)) & 15) == 0;
    };
    const bool vec_in = aligned16(in, p.in_plane), vec_out = aligned16(out, p.out_plane);
    const bool vec_aux = EPI != LSR_EPI_NONE && aligned16(aux, p.aux_plane);
    auto load4 = [&](const float* base, bool vec) {
      f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
      if (vec) return *reinterpret_cast<const f32x4*>(base);
      if (n_valid > 0) v.x = base[0];
      if (n_valid > 1) v.y = base[1];
      if (n_valid > 2) v.z = base[2];
      if (n_valid > 3) v.w = base[3];
      return v;
    };
    f32x4 rxy = {1.0f, 1.0f, 1.0f, 1.0f};
    if constexpr (EPI == LSR_EPI_UPDATE) {
      const float nyv = p.ny[y];
      rxy = f32x4{nyv * p.nx[min(x, p.X - 1)], nyv * p.nx[min(x + 1, p.X - 1)], nyv * p.nx[min(x + 2, p.X - 1)],
                  nyv * p.nx[min(x + 3, p.X - 1)]};
    }
    f32x4 acc[PZ];
#pragma unroll
    for (int j = 0; j < PZ; ++j) acc[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const int zi0 = max(zb - CZ, 0), zi1 = ze + CZ;   // planes [zi0, zi1): those >= Z are zeros
    f32x4 next = zi0 < p.Z ? load4(in + static_cast<int64_t>(zi0) * p.in_plane, vec_in) : f32x4{0, 0, 0, 0};
    // the planes before zi0 a chunk would need are either outside the volume (zeros: nothing to add) or, for chunks
    // that start inside, absorbed here without producing output
    for (int zi = max(zb - CZ, 0); zi < zi1; ++zi) {
      const f32x4 v = next;
      if (zi + 1 < zi1) next = zi + 1 < p.Z ? load4(in + static_cast<int64_t>(zi + 1) * p.in_plane, vec_in) : f32x4{0, 0, 0, 0};
      // acc[j] <-> output plane zi - CZ + j; tap index a = zi - z_out + CZ = PZ - 1 - j
#pragma unroll
      for (int j = 0; j < PZ - 1; ++j) acc[j] = __builtin_elementwise_fma(f32x4{w[PZ - 1 - j], w[PZ - 1 - j], w[PZ - 1 - j], w[PZ - 1 - j]}, v, acc[j + 1]);
      acc[PZ - 1] = f32x4{w[0], w[0], w[0], w[0]} * v;
      const int zo = zi - CZ;
      if (zo >= zb && zo < ze) {
        f32x4 c = acc[0], r;
        if constexpr (EPI == LSR_EPI_RATIO) {
          const f32x4 a = load4(aux + static_cast<int64_t>(zo) * p.aux_plane, vec_aux);
          r = a / (c + f32x4{p.eps, p.eps, p.eps, p.eps});
        } else if constexpr (EPI == LSR_EPI_UPDATE) {
          const f32x4 a = load4(aux + static_cast<int64_t>(zo) * p.aux_plane, vec_aux);
          const float nzv = p.nz[zo];
          const f32x4 ac = a * c;
          r = ac / (f32x4{nzv, nzv, nzv, nzv} * rxy);
          if (p.stats) {
            if (n_valid > 0) st.add(a.x, ac.x, r.x);
            if (n_valid > 1) st.add(a.y, ac.y, r.y);
            if (n_valid > 2) st.add(a.z, ac.z, r.z);
            if (n_valid > 3) st.add(a.w, ac.w, r.w);
          }
        } else {
          r = c;
        }
        float* o = out + static_cast<int64_t>(zo) * p.out_plane;
        if (vec_out) {
          *reinterpret_cast<f32x4*>(o) = r;
          continue;
        }
        if (n_valid > 0) o[0] = r.x;
        if (n_valid > 1) o[1] = r.y;
        if (n_valid > 2) o[2] = r.z;
        if (n_valid > 3) o[3] = r.w;
      }
    }
    // (a chunk that starts inside the volume begins at plane zb - CZ with zero sums: output plane zo >= zb needs planes
    // zo - CZ .. zo + CZ, none of them earlier)
  }
  if constexpr (EPI == LSR_EPI_UPDATE) {
    if (p.stats) {   // (kernel-uniform)
      __shared__ float scratch[3 * (kThreads / 64)];
      lsr::rl_stats_flush<kThreads / 64>(st, scratch, p.stats);
    }
  }
}

template <int PZ>
void launch_pz(const ZArgs& p, int epilogue, dim3 grid, hipStream_t s) {
  switch (epilogue) {
    case LSR_EPI_RATIO: hipLaunchKernelGGL((correlate_z_kernel<PZ, LSR_EPI_RATIO>), grid, dim3(kThreads), 0, s, p); break;
    case LSR_EPI_UPDATE: hipLaunchKernelGGL((correlate_z_kernel<PZ, LSR_EPI_UPDATE>), grid, dim3(kThreads), 0, s, p); break;
    default: hipLaunchKernelGGL((correlate_z_kernel<PZ, LSR_EPI_NONE>), grid, dim3(kThreads), 0, s, p); break;
  }
}

}  // namespace

extern "C" int lsr_correlate_z_max_taps(void) { return kMaxZTaps; }

extern "C" int lsr_correlate_z_f32(const float* in, int64_t in_pitch, int64_t in_plane, const float* aux, int64_t aux_pitch,
                                   int64_t aux_plane, float* out, int64_t out_pitch, int64_t out_plane, int64_t Z, int64_t Y,
                                   int64_t X, const float* wz, int pz, int epilogue, float eps, const float* nz,
                                   const float* ny, const float* nx, double* stats, lsr_stream_t stream) {
  LSR_REQUIRE_PTR(in);
  LSR_REQUIRE_PTR(out);
  LSR_REQUIRE_PTR(wz);
  LSR_REQUIRE(Z > 0 && Y > 0 && X > 0, LSR_E_SHAPE, "shape (%lld,%lld,%lld) must be positive", (long long)Z, (long long)Y,
              (long long)X);
  LSR_REQUIRE_VOLUME(Z, Y, X);
  LSR_REQUIRE(pz >= 1 && (pz & 1) && pz <= kMaxZTaps, LSR_E_UNSUPPORTED, "z taps %d must be odd and <= %d", pz, kMaxZTaps);
  LSR_REQUIRE(epilogue == LSR_EPI_NONE || epilogue == LSR_EPI_RATIO || epilogue == LSR_EPI_UPDATE, LSR_E_ARG,
              "unknown epilogue %d", epilogue);
  if (epilogue != LSR_EPI_NONE) LSR_REQUIRE_PTR(aux);
  if (epilogue == LSR_EPI_UPDATE) {
    LSR_REQUIRE_PTR(nz);
    LSR_REQUIRE_PTR(ny);
    LSR_REQUIRE_PTR(nx);
  }
  LSR_REQUIRE(in != out, LSR_E_ARG, "out must not alias in");
  LSR_REQUIRE_STRIDES(in_pitch, in_plane);
  LSR_REQUIRE_STRIDES(out_pitch, out_plane);
  if (epilogue != LSR_EPI_NONE) LSR_REQUIRE_STRIDES(aux_pitch, aux_plane);
  LSR_REQUIRE(in_pitch >= X && out_pitch >= X && (epilogue == LSR_EPI_NONE || aux_pitch >= X), LSR_E_SHAPE,
              "a row stride is smaller than X");
  LSR_REQUIRE(Y < 65536, LSR_E_UNSUPPORTED, "Y = %lld: this kernel's grid takes fewer than 65536 rows", (long long)Y);

  ZArgs p{};
  p.in = in; p.aux = aux; p.out = out;
  p.in_plane = in_plane; p.aux_plane = aux_plane; p.out_plane = out_plane;
  p.in_pitch = static_cast<int>(in_pitch); p.aux_pitch = static_cast<int>(aux_pitch); p.out_pitch = static_cast<int>(out_pitch);
  p.Z = static_cast<int>(Z); p.Y = static_cast<int>(Y); p.X = static_cast<int>(X);
  p.wz = wz; p.pz = pz; p.eps = eps;
  p.nz = nz; p.ny = ny; p.nx = nx;
  p.stats = epilogue == LSR_EPI_UPDATE ? stats : nullptr;
  p.x_groups = static_cast<int>(lsr::ceil_div(X, 4));
  // enough workgroups to fill the chip, as few z chunks as that takes: a chunk re-reads pz - 1 planes
  const int64_t per_plane = lsr::ceil_div(p.x_groups, kThreads) * Y;
  int64_t chunks = lsr::ceil_div(int64_t(256) * 8, per_plane);
  if (chunks < 1) chunks = 1;
  int64_t chunk = lsr::ceil_div(Z, chunks);
  if (chunk < 4 * int64_t(pz)) chunk = 4 * int64_t(pz);
  if (chunk > Z) chunk = Z;
  p.z_chunk = static_cast<int>(chunk);
  const int64_t gz = lsr::ceil_div(Z, chunk);
  LSR_REQUIRE(gz < 65536, LSR_E_SHAPE, "grid of %lld z chunks is too large", (long long)gz);
  const dim3 grid(static_cast<unsigned>(lsr::ceil_div(p.x_groups, kThreads)), static_cast<unsigned>(Y), static_cast<unsigned>(gz));
  hipStream_t s = lsr::as_stream(stream);
  switch (pz) {
#define LSR_Z(N) case N: launch_pz<N>(p, epilogue, grid, s); break;
    LSR_Z(1) LSR_Z(3) LSR_Z(5) LSR_Z(7) LSR_Z(9) LSR_Z(11) LSR_Z(13) LSR_Z(15) LSR_Z(17) LSR_Z(19) LSR_Z(21) LSR_Z(23)
    LSR_Z(25) LSR_Z(27) LSR_Z(29) LSR_Z(31)
#undef LSR_Z
    default: return lsr::fail(LSR_E_UNSUPPORTED, "z taps %d", pz);
  }
  return lsr::launch_status("lsr_correlate_z_f32");
}
