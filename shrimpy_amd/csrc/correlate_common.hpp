// Argument blocks shared by the correlation kernels (correlate.hip, correlate_sep.hip).
#pragma once

#include "common.hpp"

namespace lsr {

// Generic kernels (dense volumes, zero padding evaluated with bounds checks).
struct CorrArgs {
  const float* in;
  float* out;
  const float* aux;
  int64_t Z, Y, X;
  const float* wz;  // separable factors (device), pz / py / px taps
  const float* wy;
  const float* wx;
  const float* w;   // dense taps (device), C-order (pz, py, px)
  int pz, py, px;
  int epilogue;
  float eps;
  const float* nz;  // separable norm factors (Z, Y, X floats)
  const float* ny;
  const float* nx;
  const double* norm_table;  // dense: (pz+1)(py+1)(px+1) prefix sums
  int64_t tiles_x, tiles_y;
  int64_t z_chunk;  // output planes per workgroup along z
  double* stats;    // LSR_EPI_UPDATE: three running sums of this launch (rl_stats below) or NULL
};

// Tuned separable kernel: every volume is strided, `in` additionally carries a zero halo so that
// no load needs a bounds check.  Pointers address the LOGICAL element (0,0,0).
struct SepArgs {
  const float* in;
  const float* aux;
  float* out;
  int64_t in_plane, aux_plane, out_plane;  // z strides (floats)
  int in_pitch, aux_pitch, out_pitch;      // y strides (floats)
  int Z, Y, X;
  const float* wz;
  const float* wy;
  const float* wx;
  int pz, py, px;
  int epilogue;
  float eps;
  const float* nz;
  const float* ny;
  const float* nx;
  int tiles_x, tiles_y;
  int z_chunk;
  double* stats;    // LSR_EPI_UPDATE: three running sums of this launch (rl_stats below) or NULL
};

// Tuned dense kernel: like SepArgs.  `taps` is a small DEVICE array prepared by
// lsr_dense_prepare_taps: layout [c][b][j] over the COMPILED extents (PZ, PYX, PYX), j = PZ-1-a
// (z reversed: the order in which a staged plane feeds the pending output planes), the caller's
// taps centred, zeros elsewhere.  The kernel reads it through the constant address space, i.e.
// the scalar cache: every FMA takes its weight from an SGPR.
struct DenseArgs {
  const float* in;
  const float* aux;
  float* out;
  int64_t in_plane, aux_plane, out_plane;
  int in_pitch, aux_pitch, out_pitch;
  int Z, Y, X;
  int pz, py, px;            // the caller's PSF extents (norm table geometry)
  int epilogue;
  float eps;
  float norm_full;           // sum of all taps (interior voxels)
  const double* norm_table;  // (pz+1)(py+1)(px+1) prefix sums (border voxels)
  int tiles_x, tiles_y;
  int z_chunk;
  const float* taps;
  int ysep;                  // 1: a single y tap -- the (z, x)-stencil specialisation; 2: ky (x) kzx in one launch
  const float* ky;           // ysep == 2: the py y taps (device)
  double* stats;             // LSR_EPI_UPDATE: three running sums of this launch (rl_stats below) or NULL
};

// Tile geometry of the tuned kernels, needed by the host to size the halo (lsr_sep_padded_shape).
// Dense kernel: 32 x 64 tiles.  Separable kernel: kSepWideTileY x kSepWideTileX.
constexpr int kSepTileY = 32;
constexpr int kSepTileX = 64;
constexpr int kSepCols = 2;                       // 64-column groups per separable tile
constexpr int kSepWideTileX = 64 * kSepCols;      // 128
// rows per thread of the separable kernel (8 waves -> 8*run tile rows): 4 while the z accumulators
// (PZ x 2*run registers) fit the 256-VGPR budget of one 512-thread workgroup per CU, else 3
constexpr int sep_run(int PZ) { return PZ <= 9 ? 4 : 3; }
constexpr int sep_wide_tile_y(int PZ) { return 8 * sep_run(PZ); }
// Column of logical x = 0 inside a padded row: a whole 128-byte line, so that every tile's output
// run (and its `aux` run, when that volume is padded too) starts on a cache line.  Measured:
// RL launch 2.93 ms -> 2.18 ms against an origin at column PX/2 (line-straddling stores).
constexpr int kSepOriginCol = 32;
inline int sep_round_taps(int n) { return n < 3 ? 3 : n; }  // compiled: 3, 5, ..., 15
// floats the dense kernel stages per row: 64 + PX - 1 rounded up to a multiple of 4, at least 72
inline int sep_stage_cols(int PX) {
  int c = (kSepTileX + PX - 1 + 3) / 4 * 4;
  return c < 72 ? 72 : c;
}
// floats the separable kernel stages per row: its last x-pass item reads ceil((4+PX-1)/4) 16-byte
// pieces starting at column tile_x - 4
// Fused Richardson-Lucy iteration (rl_fused_sep.hip): x, y and x_new are padded volumes that share
// one geometry (pitch, plane); pointers address the LOGICAL element (0,0,0).
struct FusedArgs {
  const float* x;
  const float* y;
  float* out;
  int64_t plane, y_plane, out_plane;  // z strides (floats)
  int pitch, y_pitch, out_pitch;      // y strides (floats)
  int Z, Y, X;
  const float* taps;  // device block from lsr_rl_sep_fused_prepare_taps (6 rows of 16 floats)
  float eps;
  int mask_out;       // 1: `out` is dense, stores are masked to the volume; 0: padded, unmasked
  const float* nz;
  const float* ny;
  const float* nx;
  int tiles_x, tiles_y;
  int n_full;   // tiles [0, n_full) are whole z columns, one workgroup each (dispatched first)
  int pieces;   // every other tile: `pieces` workgroups of z_chunk planes
  int z_chunk;
  double* stats;  // three running sums of this iteration (rl_stats below) or NULL: the <.., STATS = false> kernel
#ifdef LSR_FUSED_PROBE_TIME
  unsigned long long* probe;  // diagnostic build: 4 timestamps per workgroup (tools/fused_drift.py)
#endif
};
// columns staged left and right of a 128-column tile: twice the PSF radius, rounded up to 16 bytes
constexpr int fused_window_halo(int PYX) { return 4 * ((2 * (PYX / 2) + 3) / 4); }
// tile rows / 8 of the fused kernel: as large as the accumulators (PZ planes of both stencils, 256
// VGPRs per thread) and the staged windows (160 KB of LDS) allow
constexpr int fused_run(int PZ, int PYX) {
  if (PZ >= 11 && PYX >= 11) return 2;
  const int by_z = PZ <= 9 ? 4 : (PZ <= 11 ? 3 : 2);
  const int by_yx = PYX <= 9 ? 4 : (PYX <= 11 ? 3 : 2);
  return by_z < by_yx ? by_z : by_yx;
}
constexpr int kFusedMaxPZ = 15, kFusedMaxPYX = 15;
// ... except 15 z taps with 11+ in-plane taps: the accumulators spill even on the smallest tile
// (and a spill costs a full vmcnt drain per plane); those PSFs take the two-launch kernels
constexpr bool fused_compiled(int PZ, int PYX) {
  return PZ <= kFusedMaxPZ && PYX <= kFusedMaxPYX && !(PZ >= 15 && PYX >= 11);
}

// Fused Richardson-Lucy iteration for psf = ky (x) kzx (rl_fused_ysep.hip): volumes as FusedArgs; the taps are a
// 256-float device block from lsr_rl_ysep_fused_prepare_taps, the normalisation that of the dense kernels.
struct YsepArgs {
  const float* x;
  const float* y;
  float* out;
  int64_t plane, y_plane, out_plane;  // z strides (floats)
  int pitch, y_pitch, out_pitch;      // y strides (floats)
  int Z, Y, X;
  const float* taps;
  float eps;
  int pz, py, px;                     // the caller's PSF extents (geometry of the norm table)
  const double* norm_table;           // (pz+1)(py+1)(px+1) prefix sums of the full PSF
  float norm_full;                    // sum of all taps (interior voxels)
  int tiles_x, tiles_y;
  int n_full, pieces, z_chunk;        // work split, as FusedArgs
  double* stats;                      // three running sums of this iteration (rl_stats below) or NULL
  int narrow;                         // 1: 256-thread workgroups on 32 x 64 tiles (two per CU); 0: 512 threads, 32 x 128
};
// tile rows: the accumulators of both stencils (PZ planes each) must fit 256 VGPRs per thread
// (9 x 9 in-plane taps on 32 rows would be six pairs per thread in stage 1: 256 VGPRs and a spill)
constexpr int ysep_tile_rows(int PZ, int PYX) { return PZ <= 9 && !(PZ == 9 && PYX == 9) ? 32 : 24; }
constexpr int kYsepMaxPZ = 11, kYsepMaxPYX = 9;
// its tap block (lsr_rl_ysep_fused_prepare_taps): per stage kYsepMaxPYX groups of 16 floats ((z, x) taps of one column
// offset), then the y taps
constexpr int kYsepTapGroup = 16, kYsepTapY = kYsepTapGroup * kYsepMaxPYX, kYsepTapStage = kYsepTapY + 16;

// ---- Richardson-Lucy reduction scalars (VERDICT r3 row g; north-star "wavefront reductions for the ratio /
// normalisation") -----------------------------------------------------------------------------------------------------
// Every kernel that finishes an iteration (the fused launches, any LSR_EPI_UPDATE epilogue) can add three sums over the
// voxels it writes to a caller-supplied double[3]:
//   [0] flux    sum x * H^T(ratio)   (= sum x_new * H^T 1: what the update conserves, -> sum y as eps -> 0)
//   [1] change  sum |x_new - x|      (the update norm: / [2] it is the relative change a caller stops on)
//   [2] total   sum x_new
// The kernels have x * u in hand before the division by H^T 1, so the flux costs one add per voxel and no extra byte.
// Per thread the sums run in f32 over the voxels it owns (a few hundred to ~1400, rounding errors independent from
// thread to thread), then: DPP wave reduction -> LDS -> three double adds per workgroup -> ONE atomic per workgroup and
// sum (global_atomic_add_f64; the order of the workgroups' doubles is the only run-to-run freedom, ~1e-16 relative).
constexpr int kRlStats = 3;
#if defined(__HIPCC__)
struct RlStats {
  float flux = 0.0f, change = 0.0f, total = 0.0f;
  __device__ __forceinline__ void add(float x_old, float xu, float x_new) {
    flux += xu;
    change += __builtin_fabsf(x_new - x_old);
    total += x_new;
  }
  __device__ __forceinline__ void pin() { asm volatile("" : "+v"(flux), "+v"(change), "+v"(total)); }
};
// The hand-pipelined kernels end with loads in flight into registers whose values nobody will read (the prefetches of
// planes past the last one).  To the compiler those registers are dead after their last use, and the kernels' final
// `s_waitcnt vmcnt(0)` is a volatile asm, which orders other volatile asms but no plain arithmetic: code that follows
// the loop -- the reduction's lane arithmetic -- may be scheduled above the wait INTO such a register, and the load
// then lands on top of it (round 4: one instance in forty-nine returned sums that were off by 1e-3, differently every
// run).  Naming every in-flight destination in an (empty) volatile asm BEHIND the wait keeps it allocated up to there.
template <typename V>
__device__ __forceinline__ void keep_until_here(V& v) { asm volatile("" : "+v"(v)); }
template <typename V, int K>
__device__ __forceinline__ void keep_until_here(V (&a)[K]) {
#pragma unroll
  for (int i = 0; i < K; ++i) asm volatile("" : "+v"(a[i]));
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// All threads of the workgroup call this once, after their last use of `lds` (>= 3 * NWAVES floats, any content).
template <int NWAVES>
__device__ __forceinline__ void rl_stats_flush(const RlStats& s, float* lds, double* dst) {
  const float a = wave_sum(s.flux), b = wave_sum(s.change), c = wave_sum(s.total);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();   // every wave is done with the buffers this scratch may alias
  if (lane == 0) {
    lds[3 * wave] = a;
    lds[3 * wave + 1] = b;
    lds[3 * wave + 2] = c;
  }
  __syncthreads();
  if (threadIdx.x < kRlStats) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < NWAVES; ++w) t += static_cast<double>(lds[3 * w + threadIdx.x]);
    unsafeAtomicAdd(dst + threadIdx.x, t);
  }
}
#endif

inline int sep_wide_stage_cols(int PX) { return kSepWideTileX - 4 + 4 * ((4 + PX - 1 + 3) / 4); }

// correlate_sep.hip, compiled once per PZ (-DLSR_SEP_PZ=n).  `pyx` is the (square) in-plane tap
// count; false = no such specialisation.
#define LSR_DECL_SEP(n) \
  bool launch_sep_pz##n(int pyx, const SepArgs& p, unsigned blocks, hipStream_t s);
LSR_DECL_SEP(3)
LSR_DECL_SEP(5)
LSR_DECL_SEP(7)
LSR_DECL_SEP(9)
LSR_DECL_SEP(11)
LSR_DECL_SEP(13)
LSR_DECL_SEP(15)
#undef LSR_DECL_SEP

// rl_fused_sep.hip, compiled once per PZ (-DLSR_FUSED_PZ=n); pyx in {3,...,15}.
#define LSR_DECL_FUSED(n) \
  bool launch_fused_pz##n(int pyx, const FusedArgs& p, unsigned blocks, hipStream_t s);
LSR_DECL_FUSED(3)
LSR_DECL_FUSED(5)
LSR_DECL_FUSED(7)
LSR_DECL_FUSED(9)
LSR_DECL_FUSED(11)
LSR_DECL_FUSED(13)
LSR_DECL_FUSED(15)
#undef LSR_DECL_FUSED

// rl_fused_ysep.hip, compiled once per PZ (-DLSR_YSEP_PZ=n); pyx in {3,5,7,9}.
#define LSR_DECL_YSEP(n) \
  bool launch_ysep_pz##n(int pyx, const YsepArgs& p, unsigned blocks, hipStream_t s);
LSR_DECL_YSEP(3)
LSR_DECL_YSEP(5)
LSR_DECL_YSEP(7)
LSR_DECL_YSEP(9)
LSR_DECL_YSEP(11)
#undef LSR_DECL_YSEP

// correlate_dense.hip, compiled once per PZ (-DLSR_DENSE_PZ=n); pyx in {3,5,7,9}, PZ*pyx*pyx <= 900.
#define LSR_DECL_DENSE(n) \
  bool launch_dense_pz##n(int pyx, const DenseArgs& p, unsigned blocks, hipStream_t s);
LSR_DECL_DENSE(3)
LSR_DECL_DENSE(5)
LSR_DECL_DENSE(7)
LSR_DECL_DENSE(9)
LSR_DECL_DENSE(11)
#undef LSR_DECL_DENSE

}  // namespace lsr
