// Dense (non-separable) 3-D PSF correlation with fused Richardson-Lucy epilogues -- the
// fp32-VALU-bound kernel (pz*py*px FMAs per voxel; a 9x7x7 PSF is 441 taps).
// One translation unit per PZ (-DLSR_DENSE_PZ=n).
//
// Same skeleton as correlate_sep.hip: zero-haloed padded input, a 512-thread workgroup owns a
// 32 x 64 (y, x) column and marches along z, planes are fetched two iterations ahead into
// registers and committed to a double-buffered LDS window, loads and waits are managed by hand
// (see the header of correlate_sep.hip for why), one workgroup barrier per plane.
//
// The compute differs: every staged plane contributes to PZ pending output planes at once.
// A thread owns one column and 4 rows; for each in-plane tap (b, c) it holds the 4+PYX-1 column
// values in registers (conflict-free ds_read_b32) and issues PZ x 4 FMAs whose weight operand is
// an SGPR: the taps are read through the CONSTANT address space (scalar cache) -- the one way a
// kernel that also stores to global memory gets scalar loads -- one (b, c) group ahead of its use.
// An opaque per-plane offset (always 0) keeps hipcc from hoisting all pz*py*px loads out of the
// plane loop, which would spill ~1800 SGPRs.  The whole tap nest is unrolled (1764 FMAs per
// thread and plane for 9x7x7).
//
// Roofline: 2*pz*py*px flop per voxel per launch against the 157 TFLOP/s fp32 vector peak;
// algorithmic HBM bytes stay 12 per voxel per launch (far below the VALU time beyond ~120 taps).

#include "common.hpp"
#include "correlate_common.hpp"

#ifndef LSR_DENSE_PZ
#error "compile with -DLSR_DENSE_PZ=<odd tap count along z>"
#endif

namespace {

using lsr::DenseArgs;

constexpr int kTY = lsr::kSepTileY;      // 32
constexpr int kTX = lsr::kSepTileX;      // 64
constexpr int kRun = 4;                  // rows per thread
constexpr int kWaves = kTY / kRun;       // 8
constexpr int kThreads = 64 * kWaves;    // 512
constexpr int kBand = 8;

template <int PYX>
struct Tile {
  static constexpr int AR = kTY + PYX - 1;
  static constexpr int AC = kTX + PYX - 1;
  static constexpr int PA0 = (AC + 3) / 4 * 4;
  static constexpr int PA = PA0 < 72 ? 72 : PA0;   // must equal lsr::sep_stage_cols(PYX)
  static constexpr int CH = PA / 4;
  static constexpr int NCH = AR * CH;
  static constexpr int SL = (NCH + kThreads - 1) / kThreads;
  static constexpr int ASZ = AR * PA;
  static_assert(SL == 2, "the hand-counted waits assume two staging loads per thread");
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float fast_rcp(float d) {
  float r = __builtin_amdgcn_rcpf(d);
  return fmaf(fmaf(-d, r, 1.0f), r, r);
}
// A wave-uniform pointer the register allocator must keep in SGPRs (under SGPR pressure hipcc
// otherwise hands the "s" operand of the asm loads a VGPR pair, which does not assemble).
__device__ __forceinline__ const float* uniform_ptr(const float* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v));
  const unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v >> 32));
  return reinterpret_cast<const float*>((static_cast<unsigned long long>(hi) << 32) | lo);
}
__device__ __forceinline__ void gload_x4(f32x4& dst, const float* sbase, int voff_bytes) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(dst) : "v"(voff_bytes), "s"(sbase) : "memory");
}
__device__ __forceinline__ void gload_x1(float& dst, const float* sbase, int voff_bytes) {
  asm volatile("global_load_dword %0, %1, %2" : "+v"(dst) : "v"(voff_bytes), "s"(sbase) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_loads(f32x4& a, f32x4& b) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_loads(float (&a)[kRun]) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "n"(N) : "memory");
}
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

// H^T 1 at (z, y, x): the sum of the taps whose sample lies inside the volume, from the prefix-sum
// table P[a][b][c] = sum_{a'<a, b'<b, c'<c} w of the CALLER's pz x py x px PSF.
// (P: the table, staged in LDS by the UPDATE kernel -- eight dependent global loads per border
// voxel made the UPDATE launch 22 % slower than the RATIO launch)
__device__ float dense_norm(const DenseArgs& p, const double* P, int z, int y, int x) {
  const int cz = p.pz / 2, cy = p.py / 2, cx = p.px / 2;
  const int a0 = max(0, cz - z), a1 = min(p.pz, p.Z - z + cz);
  const int b0 = max(0, cy - y), b1 = min(p.py, p.Y - y + cy);
  const int c0 = max(0, cx - x), c1 = min(p.px, p.X - x + cx);
  const int sb = p.px + 1, sa = (p.py + 1) * sb;
  return static_cast<float>(((P[a1 * sa + b1 * sb + c1] - P[a0 * sa + b1 * sb + c1]) -
                             (P[a1 * sa + b0 * sb + c1] - P[a0 * sa + b0 * sb + c1])) -
                            ((P[a1 * sa + b1 * sb + c0] - P[a0 * sa + b1 * sb + c0]) -
                             (P[a1 * sa + b0 * sb + c0] - P[a0 * sa + b0 * sb + c0])));
}

// MODE 0: the full PZ x PYX x PYX stencil.
// MODE 1: the caller's PSF has one tap along y (a (z, x) stencil: the dense half of a PSF that
//         separates along y, see deconvolve.py) -- only the centre row of the compiled PYX x PYX
//         footprint is visited, PZ * PYX FMAs per voxel instead of PZ * PYX * PYX.
// MODE 2: the whole PSF ky (x) kzx in one launch: every wave first filters its own four rows of the
//         staged plane along y (PYX taps, LDS -> LDS, rows only that wave reads: no barrier), then
//         runs the (z, x) stencil of MODE 1 on the filtered rows.  PZ * PYX + PYX FMAs per voxel.
// STATS (LSR_EPI_UPDATE only): the launch's Richardson-Lucy scalars are summed in the epilogue and added to p.stats
// (correlate_common.hpp: RlStats) -- its own instantiation: a per-point test of p.stats in the epilogue, and three more
// live registers, cost the plain launch 12 % (2.43 -> 2.73 ms on config 2's grid, round 4).
template <int PZ, int PYX, int EPI, int MODE, bool STATS = false>
__global__ __launch_bounds__(kThreads) void correlate_dense_kernel(DenseArgs p) {
  using T = Tile<PYX>;
  __shared__ f32x4 bufA4[2 * T::ASZ / 4];
  constexpr bool kAux = EPI == LSR_EPI_RATIO || EPI == LSR_EPI_UPDATE;
  constexpr bool kNorm = EPI == LSR_EPI_UPDATE || EPI == LSR_EPI_SCALE;
  constexpr int NA = kAux ? kRun : 0;  // aux loads per iteration
  // UPDATE / SCALE: the (pz+1)(py+1)(px+1) prefix sums of the PSF, for the border normalisation
  constexpr int kNormTable = kNorm ? 12 * 10 * 10 : 1;
  __shared__ double s_norm[kNormTable];
  constexpr bool YS = MODE != 0;
  __shared__ float bufB[MODE == 2 ? kTY * T::PA : 1];  // y-filtered rows of the plane being absorbed
  if constexpr (kNorm) {
    const int n = (p.pz + 1) * (p.py + 1) * (p.px + 1);
    for (int i = threadIdx.x; i < n; i += kThreads) s_norm[i] = p.norm_table[i];
    __syncthreads();
  }

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // XCD-aware, band-swizzled tile order (see correlate_sep.hip)
  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x;
    const int per = nblk / 8, rem = nblk % 8;
    const int xcd = bid % 8, idx = bid / 8;
    bid = xcd * per + (xcd < rem ? xcd : rem) + idx;
  }
  const int tiles_xy = p.tiles_x * p.tiles_y;
  const int zc = bid / tiles_xy;
  const int lin = bid - zc * tiles_xy;
  const int band = lin / (p.tiles_x * kBand);
  const int lb = lin - band * (p.tiles_x * kBand);
  const int band_h = min(kBand, p.tiles_y - band * kBand);
  const int tx = lb / band_h;
  const int ty = band * kBand + (lb - tx * band_h);

  const int Z = p.Z, Y = p.Y, X = p.X;
  const int x0 = tx * kTX, y0 = ty * kTY;
  const int zb = zc * p.z_chunk;
  const int ze = min(zb + p.z_chunk, Z);
  constexpr int cz = PZ / 2, cyx = PYX / 2;

  // ---- staging (identical to the separable kernel)
  const float* const in_tile = p.in + (static_cast<int64_t>(y0 - cyx) * p.in_pitch + (x0 - cyx));
  int s_voff[T::SL], s_loff[T::SL];
#pragma unroll
  for (int k = 0; k < T::SL; ++k) {
    const int e = min(tid + k * kThreads, T::NCH - 1);
    const int r = e / T::CH, c = e - r * T::CH;
    s_voff[k] = (r * p.in_pitch + 4 * c) * 4;
    s_loff[k] = e;
  }
  f32x4 st0[T::SL] = {}, st1[T::SL] = {};   // (initialised: in/out operands of the asm loads)

  // ---- compute / epilogue geometry: column `lane`, rows 4*wave .. 4*wave+3
  const int acol = (wave * kRun) * T::PA + lane;
  const int gx_out = x0 + lane;
  const bool xok = gx_out < X;
  const int gxc = min(gx_out, X - 1);
  const int gy_out0 = y0 + wave * kRun;
  int a_voff[kRun], o_off[kRun];
  bool ok[kRun];
#pragma unroll
  for (int m = 0; m < kRun; ++m) {
    const int gy = gy_out0 + m;
    ok[m] = xok && gy < Y;
    a_voff[m] = (min(gy, Y - 1) * p.aux_pitch + gxc) * 4;
    o_off[m] = min(gy, Y - 1) * p.out_pitch + gxc;
  }
  // in-plane interior test for the norm of the UPDATE epilogue (the caller's PSF extents)
  bool yx_inside[kRun];
  {
    const int ry = p.py / 2, rx = p.px / 2;
#pragma unroll
    for (int m = 0; m < kRun; ++m) {
      const int gy = gy_out0 + m;
      yx_inside[m] = gy >= ry && gy < Y - ry && gx_out >= rx && gx_out < X - rx;
    }
  }

  typedef const float __attribute__((address_space(4))) cfloat;  // constant AS: scalar loads
  const cfloat* const taps_base = (const cfloat*)p.taps;
  float kyv[PYX];  // MODE 2: the caller's py y taps, centred in the compiled extent
#pragma unroll
  for (int b = 0; b < PYX; ++b) kyv[b] = 0.0f;
  if constexpr (MODE == 2) {
    const cfloat* const ky = (const cfloat*)p.ky;
    const int off = (PYX - p.py) / 2;
#pragma unroll
    for (int b = 0; b < PYX; ++b)
      if (b >= off && b < off + p.py) kyv[b] = ky[b - off];
  }

  // pending output planes as packed pairs (rows 2q, 2q+1 of the thread's column): v_pk_fma_f32
  static_assert(kRun % 2 == 0, "rows are paired");
  f32x2 acc[PZ][kRun / 2];
#pragma unroll
  for (int j = 0; j < PZ; ++j)
#pragma unroll
    for (int q = 0; q < kRun / 2; ++q) acc[j][q] = f32x2{0.0f, 0.0f};
  float aux0[kRun], aux1[kRun];
#pragma unroll
  for (int m = 0; m < kRun; ++m) aux0[m] = aux1[m] = 0.0f;

  lsr::RlStats stats;   // UPDATE with p.stats: the launch's reduction scalars (correlate_common.hpp)
  const int zi_begin = max(zb - cz, 0);
  const int zi_end = ze + cz;

  auto fetch = [&](int zplane, f32x4 (&st)[T::SL]) __attribute__((always_inline)) {
    const float* src = uniform_ptr(in_tile + static_cast<int64_t>(min(max(zplane, 0), Z - 1)) * p.in_plane);
    gload_x4(st[0], src, s_voff[0]);
    gload_x4(st[1], src, s_voff[1]);
  };
  auto fetch_aux = [&](int zout, float (&aux)[kRun]) __attribute__((always_inline)) {
    if constexpr (kAux) {
      const float* a = uniform_ptr(p.aux + static_cast<int64_t>(min(max(zout, 0), Z - 1)) * p.aux_plane);
#pragma unroll
      for (int m = 0; m < kRun; ++m) gload_x1(aux[m], a, a_voff[m]);
    }
  };

  // Iteration zi consumes plane zi+1 (committed one iteration earlier into A[par^1]).
  auto iteration = [&](const int zi, const int par, f32x4 (&st)[T::SL], float (&aux_use)[kRun],
                       float (&aux_load)[kRun]) __attribute__((always_inline)) {
    f32x4* A_commit = bufA4 + par * (T::ASZ / 4);  // plane zi+2
    const float* A_c = reinterpret_cast<const float*>(bufA4 + (par ^ 1) * (T::ASZ / 4)) + acol;

    wait_loads<2 + 2 * NA>(st[0], st[1]);
    A_commit[s_loff[0]] = st[0];
    A_commit[s_loff[1]] = st[1];
    __builtin_amdgcn_sched_barrier(0);
    fetch(zi + 4, st);
    fetch_aux(zi + 2 - cz, aux_load);

    const int zcur = zi + 1;
    if (zcur >= zi_begin && zcur < zi_end) {  // wave-uniform
      // Pending planes move up by one (acc[j] <-> z_out = zcur - cz + j) -- folded into the first
      // tap group's FMAs (acc[j] = w * v + acc[j+1], ascending j), so no register moves.
      if (zcur < Z) {
        if constexpr (MODE == 2) {
          // y pass of this wave's rows: columns lane and (for the x halo) lane + 64
          const float* Ay = reinterpret_cast<const float*>(bufA4 + (par ^ 1) * (T::ASZ / 4)) + (wave * kRun) * T::PA;
          float* Bw = bufB + (wave * kRun) * T::PA;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int col = lane + 64 * h;
            if (h == 0 || col < T::AC) {
              float cy[kRun + PYX - 1];
#pragma unroll
              for (int j = 0; j < kRun + PYX - 1; ++j) cy[j] = Ay[j * T::PA + col];
#pragma unroll
              for (int m = 0; m < kRun; ++m) {
                float sy = kyv[0] * cy[m];
#pragma unroll
                for (int b = 1; b < PYX; ++b) sy = fmaf(kyv[b], cy[m + b], sy);
                Bw[m * T::PA + col] = sy;
              }
            }
          }
        }
        // (LDS operations of one wave execute in order: the reads below see the rows just written)
        const float* const S_c = MODE == 2 ? bufB + acol : A_c;   // MODE 2: row j of S is output row j
        int opaque = 0;
        asm volatile("" : "+s"(opaque));  // loop-variant for the optimiser, always 0
        const cfloat* taps = taps_base + opaque;
        // every tap is stored twice: an aligned SGPR pair {w, w} is the packed FMA's scalar
        // operand as it is (no s_mov to realign odd taps)
        // two SGPR sets (the next group's taps load while this group's FMAs issue) while 2 * 2 * PZ
        // scalars fit; one set from 11 z taps up
        constexpr int NBUF = PZ <= 9 ? 2 : 1;
        f32x2 wq[NBUF][PZ];
#pragma unroll
        for (int j = 0; j < PZ; ++j) {
          constexpr int t0 = YS ? cyx * PZ : 0;  // first visited group: (c = 0, b = centre) or (0, 0)
          wq[0][j] = f32x2{taps[2 * (t0 + j)], taps[2 * (t0 + j) + 1]};
        }
#pragma unroll
        for (int c = 0; c < PYX; ++c) {
          // column values: all kRun + PYX - 1 rows the y taps reach, or just the output rows
          constexpr int R0 = YS ? cyx : 0, NR = YS ? kRun : kRun + PYX - 1;
          constexpr int RS = MODE == 2 ? 0 : R0;   // first row to read in the source plane
          float cv[NR];
#pragma unroll
          for (int j = 0; j < NR; ++j) cv[j] = S_c[(j + RS) * T::PA + c];
          constexpr int B0 = YS ? cyx : 0, B1 = YS ? cyx + 1 : PYX;
#pragma unroll
          for (int b = B0; b < B1; ++b) {
            constexpr int G = YS ? PYX : PYX * PYX;             // tap groups this kernel visits
            const int g = YS ? c : c * PYX + b;                 // ... and this one's place among them
            auto tap_group = [](int gi) { return YS ? gi * PYX + cyx : gi; };  // its index in the tap block
            if constexpr (NBUF == 2) {  // taps of the NEXT group -> the other SGPR set
              if (g + 1 < G) {
#pragma unroll
                for (int j = 0; j < PZ; ++j)
                  wq[(g + 1) & 1][j] = f32x2{taps[2 * (tap_group(g + 1) * PZ + j)], taps[2 * (tap_group(g + 1) * PZ + j) + 1]};
              }
            } else if (g > 0) {
#pragma unroll
              for (int j = 0; j < PZ; ++j)
                wq[0][j] = f32x2{taps[2 * (tap_group(g) * PZ + j)], taps[2 * (tap_group(g) * PZ + j) + 1]};
            }
#pragma unroll
            for (int q = 0; q < kRun / 2; ++q) {
              const f32x2 v = f32x2{cv[2 * q + b - R0], cv[2 * q + 1 + b - R0]};
              if (g == 0) {
#pragma unroll
                for (int j = 0; j < PZ - 1; ++j) acc[j][q] = __builtin_elementwise_fma(wq[0][j], v, acc[j + 1][q]);
                acc[PZ - 1][q] = wq[0][PZ - 1] * v;
              } else {
#pragma unroll
                for (int j = 0; j < PZ; ++j) acc[j][q] = __builtin_elementwise_fma(wq[g & (NBUF - 1)][j], v, acc[j][q]);
              }
            }
          }
        }
      } else {  // a plane past the volume contributes nothing: shift only
#pragma unroll
        for (int j = 0; j < PZ - 1; ++j)
#pragma unroll
          for (int q = 0; q < kRun / 2; ++q) acc[j][q] = acc[j + 1][q];
#pragma unroll
        for (int q = 0; q < kRun / 2; ++q) acc[PZ - 1][q] = f32x2{0.0f, 0.0f};
      }

      const int z_out = zcur - cz;
      auto acc0 = [&](int m) { return (m & 1) ? acc[0][m >> 1].y : acc[0][m >> 1].x; };
      if (z_out >= zb) {  // wave-uniform
        float* o = p.out + static_cast<int64_t>(z_out) * p.out_plane;
        if constexpr (kAux) wait_loads<2 + NA>(aux_use);
        if constexpr (EPI == LSR_EPI_RATIO) {
#pragma unroll
          for (int m = 0; m < kRun; ++m)
            if (ok[m]) o[o_off[m]] = aux_use[m] * fast_rcp(acc0(m) + p.eps);
        } else if constexpr (kNorm) {
          const int rz = p.pz / 2;
          const bool z_inside = z_out >= rz && z_out < Z - rz;
#pragma unroll
          for (int m = 0; m < kRun; ++m) {
            if (ok[m]) {
              const float nrm = (z_inside && yx_inside[m]) ? p.norm_full
                                                           : dense_norm(p, s_norm, z_out, gy_out0 + m, gx_out);
              if constexpr (EPI == LSR_EPI_UPDATE) {
                const float xu = aux_use[m] * acc0(m);
                const float v = xu * fast_rcp(nrm);
                o[o_off[m]] = v;
                if constexpr (STATS) stats.add(aux_use[m], xu, v);
              } else
                o[o_off[m]] = acc0(m) * fast_rcp(nrm);
            }
          }
        } else {
#pragma unroll
          for (int m = 0; m < kRun; ++m)
            if (ok[m]) o[o_off[m]] = acc0(m);
        }
      }
    }
    lds_barrier();
  };

  const int zs = (zi_begin - 2) & ~1;
  fetch(zs + 2, st0);
  fetch_aux(zs - cz, aux1);      // placeholder: keeps the load count of an iteration pair
  fetch(zs + 3, st1);
  fetch_aux(zs + 1 - cz, aux0);  // output plane of iteration zs
  for (int zi = zs; zi + 1 < zi_end; zi += 2) {
    iteration(zi, 0, st0, aux0, aux1);
    iteration(zi + 1, 1, st1, aux1, aux0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (EPI == LSR_EPI_UPDATE && STATS) {
    lsr::keep_until_here(st0);   // (in-flight prefetches: correlate_common.hpp, keep_until_here)
    lsr::keep_until_here(st1);
    lsr::keep_until_here(aux0);
    lsr::keep_until_here(aux1);
    stats.pin();
    lsr::rl_stats_flush<kWaves>(stats, reinterpret_cast<float*>(bufA4), p.stats);
  }
}

template <int PZ, int PYX>
bool launch_one(const DenseArgs& p, dim3 grid, hipStream_t s) {
  {
    const dim3 block(kThreads);
    if (p.ysep == 2) {  // ky (x) kzx in one launch (RL ratio / update)
      switch (p.epilogue) {
        case LSR_EPI_RATIO:
          hipLaunchKernelGGL((correlate_dense_kernel<PZ, PYX, LSR_EPI_RATIO, 2>), grid, block, 0, s, p);
          return true;
        case LSR_EPI_UPDATE:
          if (p.stats != nullptr) hipLaunchKernelGGL((correlate_dense_kernel<PZ, PYX, LSR_EPI_UPDATE, 2, true>), grid, block, 0, s, p);
          else hipLaunchKernelGGL((correlate_dense_kernel<PZ, PYX, LSR_EPI_UPDATE, 2>), grid, block, 0, s, p);
          return true;
        default:
          return false;
      }
    }
    if (p.ysep == 1) {  // a (z, x) stencil: plain or normalised output (the y factor runs as its own pass)
      switch (p.epilogue) {
        case LSR_EPI_NONE:
          hipLaunchKernelGGL((correlate_dense_kernel<PZ, PYX, LSR_EPI_NONE, 1>), grid, block, 0, s, p);
          return true;
        case LSR_EPI_SCALE:
          hipLaunchKernelGGL((correlate_dense_kernel<PZ, PYX, LSR_EPI_SCALE, 1>), grid, block, 0, s, p);
          return true;
        default:
          break;  // RATIO / UPDATE of a one-row PSF: the general kernel below (zero taps included)
      }
    }
    switch (p.epilogue) {
      case LSR_EPI_NONE:
        hipLaunchKernelGGL((correlate_dense_kernel<PZ, PYX, LSR_EPI_NONE, 0>), grid, block, 0, s, p);
        return true;
      case LSR_EPI_RATIO:
        hipLaunchKernelGGL((correlate_dense_kernel<PZ, PYX, LSR_EPI_RATIO, 0>), grid, block, 0, s, p);
        return true;
      case LSR_EPI_UPDATE:
        if (p.stats != nullptr) hipLaunchKernelGGL((correlate_dense_kernel<PZ, PYX, LSR_EPI_UPDATE, 0, true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((correlate_dense_kernel<PZ, PYX, LSR_EPI_UPDATE, 0>), grid, block, 0, s, p);
        return true;
      case LSR_EPI_SCALE:
        hipLaunchKernelGGL((correlate_dense_kernel<PZ, PYX, LSR_EPI_SCALE, 0>), grid, block, 0, s, p);
        return true;
      default:
        return false;
    }
  }
}

}  // namespace

namespace lsr {

#define LSR_CAT2(a, b) a##b
#define LSR_CAT(a, b) LSR_CAT2(a, b)
bool LSR_CAT(launch_dense_pz, LSR_DENSE_PZ)(int pyx, const DenseArgs& p, unsigned blocks, hipStream_t s) {
  constexpr int PZ = LSR_DENSE_PZ;
  const dim3 grid(blocks);
  switch (pyx) {
    case 3: return launch_one<PZ, 3>(p, grid, s);
    case 5: return launch_one<PZ, 5>(p, grid, s);
    case 7: return launch_one<PZ, 7>(p, grid, s);
    case 9: return launch_one<PZ, 9>(p, grid, s);
    default: return false;
  }
}

}  // namespace lsr
