// Normal equations of one Gauss-Newton step of intensity-based affine registration -- the
// "estimate" half of the label-free <-> light-sheet registration (the apply half
// is affine*.hip).  The reference has no code for it (docs/data_structure.md:58-62); the model is the
// standard additive Lucas-Kanade step with a linear intensity map, on the same conventions as the
// apply kernels (3x4 matrix, target index -> moving coordinate, trilinear interpolation):
//
//     r(x)  = gain * M(A x~) + offset - T(x)                 x over a strided grid of target voxels (a stride per axis)
//     J(x)  = [ gain * dM/dz * x~ , gain * dM/dy * x~ , gain * dM/dx * x~ , M , 1 ]      (14 columns)
//     H     = sum J J^T  (upper triangle, 105),   b = sum J r  (14),   sse = sum r^2,   n
//
// x~ = ((zo - cz) / S, (yo - cy) / S, (xo - cx) / S, 1): centred, scaled target coordinates, so that
// the 14 x 14 system is well conditioned; the host maps the solution back to voxel units.
// A voxel contributes when its moving coordinate lies in [0, n - 1) on every axis (all eight taps
// inside the moving volume).
//
// 121 fp64 sums per thread live in registers for the whole grid-stride walk (one wave per SIMD: the
// 512-entry unified VGPR / AGPR file holds them), are reduced across the wave by DPP shuffles and
// written per workgroup; the host adds the 256 partial rows in a fixed order -- no float atomics,
// results do not depend on scheduling.  The taps are plain global gathers: this runs a few dozen
// times per registration on subsampled grids, not once per volume.

#include "common.hpp"

namespace {

constexpr int kThreads = 256;
constexpr int kParams = 14;
constexpr int kH = kParams * (kParams + 1) / 2;   // 105
constexpr int kOut = kH + kParams + 2;            // + b, sse, n

struct NormalArgs {
  const float* moving;
  const float* target;
  int Zi, Yi, Xi;
  int Zo, Yo, Xo;
  double m[12];
  double gain, offset;
  int sz, sy, sx;        // sampling stride per axis
  int nz, ny, nx;        // sampled grid: indices 0, stride, 2 stride, ... below Zo / Yo / Xo
  double cz, cy, cx, inv_s;
  double* partial;       // [gridDim.x][kOut]
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__global__ __launch_bounds__(kThreads, 1) void affine_normal_kernel(NormalArgs p) {
  double acc[kOut];
#pragma unroll
  for (int i = 0; i < kOut; ++i) acc[i] = 0.0;

  const int64_t n_samples = static_cast<int64_t>(p.nz) * p.ny * p.nx;
  const int64_t plane_i = static_cast<int64_t>(p.Yi) * p.Xi;
  for (int64_t s = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; s < n_samples;
       s += static_cast<int64_t>(gridDim.x) * kThreads) {
    const int ix = static_cast<int>(s % p.nx);
    const int64_t t = s / p.nx;
    const int iy = static_cast<int>(t % p.ny), iz = static_cast<int>(t / p.ny);
    const int zo = iz * p.sz, yo = iy * p.sy, xo = ix * p.sx;
    const double zd = zo, yd = yo, xd = xo;
    const double cz = p.m[0] * zd + p.m[1] * yd + p.m[2] * xd + p.m[3];
    const double cy = p.m[4] * zd + p.m[5] * yd + p.m[6] * xd + p.m[7];
    const double cx = p.m[8] * zd + p.m[9] * yd + p.m[10] * xd + p.m[11];
    if (!(cz >= 0.0 && cz < p.Zi - 1 && cy >= 0.0 && cy < p.Yi - 1 && cx >= 0.0 && cx < p.Xi - 1)) continue;
    const int jz = static_cast<int>(cz), jy = static_cast<int>(cy), jx = static_cast<int>(cx);
    const double fz = cz - jz, fy = cy - jy, fx = cx - jx;
    const float* base = p.moving + jz * plane_i + static_cast<int64_t>(jy) * p.Xi + jx;
    const double v000 = base[0], v001 = base[1], v010 = base[p.Xi], v011 = base[p.Xi + 1];
    const double v100 = base[plane_i], v101 = base[plane_i + 1], v110 = base[plane_i + p.Xi],
                 v111 = base[plane_i + p.Xi + 1];
    // interpolant and its gradient: lerp along x, then y, then z
    const double a00 = v000 + fx * (v001 - v000), a01 = v010 + fx * (v011 - v010);
    const double a10 = v100 + fx * (v101 - v100), a11 = v110 + fx * (v111 - v110);
    const double b0 = a00 + fy * (a01 - a00), b1 = a10 + fy * (a11 - a10);
    const double mval = b0 + fz * (b1 - b0);
    const double gz = b1 - b0;
    const double gy = (a01 - a00) + fz * ((a11 - a10) - (a01 - a00));
    const double d00 = v001 - v000, d01 = v011 - v010, d10 = v101 - v100, d11 = v111 - v110;
    const double e0 = d00 + fy * (d01 - d00), e1 = d10 + fy * (d11 - d10);
    const double gx = e0 + fz * (e1 - e0);
    const double tv = p.target[(static_cast<int64_t>(zo) * p.Yo + yo) * p.Xo + xo];
    const double r = p.gain * mval + p.offset - tv;
    const double xt[4] = {(zd - p.cz) * p.inv_s, (yd - p.cy) * p.inv_s, (xd - p.cx) * p.inv_s, 1.0};
    double J[kParams];
    const double g[3] = {p.gain * gz, p.gain * gy, p.gain * gx};
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) J[4 * a + b] = g[a] * xt[b];
    J[12] = mval;
    J[13] = 1.0;
    int k = 0;
#pragma unroll
    for (int i = 0; i < kParams; ++i)
#pragma unroll
      for (int j = i; j < kParams; ++j) {
        acc[k] = fma(J[i], J[j], acc[k]);
        ++k;
      }
#pragma unroll
    for (int i = 0; i < kParams; ++i) acc[kH + i] = fma(J[i], r, acc[kH + i]);
    acc[kH + kParams] = fma(r, r, acc[kH + kParams]);
    acc[kH + kParams + 1] += 1.0;
  }

  // wave sums, then the workgroup's four waves in wave order
  __shared__ double s_part[kThreads / 64][kOut];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < kOut; ++i) {
    const double v = wave_sum(acc[i]);
    if (lane == 0) s_part[wave][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < kOut) {
    double v = 0.0;
#pragma unroll
    for (int w = 0; w < kThreads / 64; ++w) v += s_part[w][threadIdx.x];
    p.partial[static_cast<int64_t>(blockIdx.x) * kOut + threadIdx.x] = v;
  }
}

}  // namespace

extern "C" int lsr_affine_normal_size(void) { return kOut; }
extern "C" int lsr_affine_normal_blocks(void) { return 256; }

extern "C" int lsr_affine_normal_equations_f32(const float* moving, int64_t Zi, int64_t Yi, int64_t Xi,
                                               const float* target, int64_t Zo, int64_t Yo, int64_t Xo,
                                               const double M[12], double gain, double offset,
                                               const int stride[3], const double centre[3], double scale,
                                               double* partial,
                                               lsr_stream_t stream) {
  LSR_REQUIRE_PTR(moving);
  LSR_REQUIRE_PTR(target);
  LSR_REQUIRE_PTR(M);
  LSR_REQUIRE_PTR(centre);
  LSR_REQUIRE_PTR(partial);
  LSR_REQUIRE(Zi >= 2 && Yi >= 2 && Xi >= 2, LSR_E_SHAPE, "moving shape (%lld,%lld,%lld): every axis needs two samples",
              (long long)Zi, (long long)Yi, (long long)Xi);
  LSR_REQUIRE(Zo > 0 && Yo > 0 && Xo > 0, LSR_E_SHAPE, "target shape (%lld,%lld,%lld) must be positive", (long long)Zo,
              (long long)Yo, (long long)Xo);
  LSR_REQUIRE_VOLUME(Zo, Yo, Xo);
  const int64_t lim = int64_t(1) << 30;
  LSR_REQUIRE(Zi < lim && Yi < lim && Xi < lim && Zo < lim && Yo < lim && Xo < lim, LSR_E_UNSUPPORTED,
              "a dimension exceeds 2^30");
  LSR_REQUIRE_PTR(stride);
  LSR_REQUIRE(stride[0] >= 1 && stride[1] >= 1 && stride[2] >= 1, LSR_E_ARG, "strides must be >= 1, got (%d,%d,%d)",
              stride[0], stride[1], stride[2]);
  LSR_REQUIRE(scale > 0.0, LSR_E_ARG, "scale must be positive");
  for (int i = 0; i < 12; ++i) LSR_REQUIRE(M[i] == M[i] && M[i] - M[i] == 0.0, LSR_E_ARG, "M[%d] is not finite", i);
  NormalArgs p;
  p.moving = moving; p.target = target;
  p.Zi = static_cast<int>(Zi); p.Yi = static_cast<int>(Yi); p.Xi = static_cast<int>(Xi);
  p.Zo = static_cast<int>(Zo); p.Yo = static_cast<int>(Yo); p.Xo = static_cast<int>(Xo);
  for (int i = 0; i < 12; ++i) p.m[i] = M[i];
  p.gain = gain; p.offset = offset;
  p.sz = stride[0]; p.sy = stride[1]; p.sx = stride[2];
  p.nz = static_cast<int>(lsr::ceil_div(Zo, stride[0]));
  p.ny = static_cast<int>(lsr::ceil_div(Yo, stride[1]));
  p.nx = static_cast<int>(lsr::ceil_div(Xo, stride[2]));
  p.cz = centre[0]; p.cy = centre[1]; p.cx = centre[2];
  p.inv_s = 1.0 / scale;
  p.partial = partial;
  hipLaunchKernelGGL(affine_normal_kernel, dim3(256), dim3(kThreads), 0, lsr::as_stream(stream), p);
  return lsr::launch_status("lsr_affine_normal_equations_f32");
}
