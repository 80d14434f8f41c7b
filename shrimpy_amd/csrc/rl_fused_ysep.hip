// One Richardson-Lucy iteration in ONE launch for PSFs that separate along y only, psf = ky (x) kzx:
//
//     x_new = x * H^T( y / (H x + eps) ) / (H^T 1)
//
// The PSF of an oblique light sheet is tilted in the (z, x) plane and Gaussian along y -- SURVEY.md 8(d)'s
// secondary PSF, "rotated 30 deg about Y" -- so it has no rank-1 form, but it factors into a y kernel and a
// dense (z, x) stencil.  The two-launch form (correlate_dense.hip, MODE 2: RATIO, then UPDATE) moves 24
// bytes per voxel and iteration through HBM; this kernel moves 12 algorithmic bytes (x, y -> x_new): the
// ratio volume never leaves the CU.  It is rl_fused_sep.hip with the passes of this PSF family:
//
//   stage 1  t1 = ky~ * x  (y pass, LDS -> LDS) on the staged plane, then c = kzx~ * t1 on the tile grown by
//            the in-plane radius C: every staged plane feeds the PZ pending ratio planes at once, PZ * PYX
//            FMAs per point from a window of 2C+1 neighbouring columns;  ratio = y * rcp(c + eps), zero
//            outside the volume                                                    -> LDS only
//   stage 2  t2 = ky * ratio, u = kzx * t2 on the tile;  x_new = x * u * rcp(H^T 1)     -> HBM
//
// (~ = reversed taps.)  H^T 1 at border voxels comes from the full PSF's prefix-sum table (staged in LDS),
// interior voxels use the tap sum -- exactly what the two-launch UPDATE does.  The arithmetic of every
// voxel (y chain first: product, then FMAs in tap order; (z, x) chain plane by plane in z, tap by tap in x;
// rcp with one Newton step) is the two-launch kernels', so both forms return bit-identical volumes.
//
// Two shapes of the same kernel (template <NW, NCG>): 8 waves owning a 32 x 128 tile (a thread: 4 rows of two
// 64-column groups, packed pairs = the two groups of a row; one workgroup per CU), and 4 waves owning a
// 32 x 64 tile (a thread: 8 rows of one column, packed pairs = two neighbouring rows; ~78 KB of LDS, so TWO
// workgroups share a CU and one's LDS-bound y passes and barriers overlap the other's FMAs).
//
// Volumes, work split, the LDS ring of staged x planes (global_load_lds_dwordx4), the hand-counted
// s_waitcnt vmcnt scheme and the per-iteration schedule (two workgroup barriers) are rl_fused_sep.hip's;
// see its header.  Roofline: 2 * (PZ * PYX + PYX) FMAs per voxel and iteration on ~1.2x the points for
// stage 1 -- fp32-VALU and HBM time are of the same order, neither hides the other
// completely.  Algorithmic HBM bytes: 12 per voxel and iteration.

#include "common.hpp"
#include "correlate_common.hpp"

#ifndef LSR_YSEP_PZ
#error "compile with -DLSR_YSEP_PZ=<odd tap count along z>"
#endif

namespace {

using lsr::YsepArgs;

constexpr int kBand = 8;
constexpr int kRing = 3;

constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }

// NW waves per workgroup, NCG column groups (of 64) per thread; RPW = tile rows per wave.
template <int PZ, int PYX, int NW, int NCG>
struct Geo {
  static constexpr int NT = 64 * NW;                      // threads
  static constexpr int TXW = 64 * NCG;                    // tile width
  static constexpr int C = PYX / 2, CZ = PZ / 2;
  static constexpr int TY = lsr::ysep_tile_rows(PZ, PYX);      // tile rows
  static constexpr int RPW = TY / NW;                     // stage-2 rows per wave
  static constexpr int NP2 = RPW * NCG / 2;               // stage-2 packed pairs per thread
  static constexpr int WL = lsr::fused_window_halo(PYX);  // staged columns left/right of the tile
  static constexpr int AR = TY + 4 * C;                   // staged rows
  static constexpr int PA = TXW + 2 * WL;                 // staged columns = pitch of A and of B1
  static constexpr int CH = PA / 4;                       // 16-byte chunks per staged row
  static constexpr int NCH = AR * CH;
  static constexpr int SL = cdiv(NCH, NT);                // glds per thread and plane
  static constexpr int ASZ = cdiv(NCH, 64) * 64 * 4;      // floats per ring slot (whole waves of chunks)
  static constexpr int R1 = TY + 2 * C;                   // stage-1 (ratio) rows
  // Stage 1 computes the ratio on the grown tile: R1 rows of TXW + E columns.  A thread holds NP1 packed pairs; a pair is
  // (row, lane) and (row, lane + 64) of one row -- except the LAST pair of the waves that own one row fewer than the
  // others: it holds two neighbouring EDGE columns (TXW + 2 k, TXW + 2 k + 1) of some row.  The rows are dealt so that
  // every wave runs the same NP1 pairs: waves [0, NFULL) own NP1 rows, waves [NFULL, NW) own NP1 - 1 rows and 64 edge
  // pairs each.  (Rounds 3-4 gave every wave ceil(R1 / NW) rows -- two of them padding for the last wave at 9 x 7 x 7 --
  // and the edge columns to the first waves as an extra scalar chain: the waves the barrier waits for did 5 pairs + 63
  // scalar FMAs per plane, the others 5 or 3 useful pairs.)
  static constexpr int NP1 = R1 % NW == 0 ? R1 / NW + 1 : cdiv(R1, NW);   // stage-1 packed pairs per thread
  static constexpr int NFULL = R1 - NW * (NP1 - 1);       // waves with NP1 rows (the others: NP1 - 1 rows + an edge pair)
  static constexpr int NIT1 = R1 * CH;                    // chunks of t1 (zero fill)
  static constexpr int XIT1 = cdiv(NIT1, NT);
  static constexpr int B1SZ = R1 * PA;
  static constexpr int SH1 = WL - 2 * C;                  // B1 column of ratio column 0's first x tap
  static constexpr int E = 2 * C;                         // ratio columns beyond the 64-lane groups
  static constexpr int HE = E / 2;                        // edge pairs per ratio row
  static constexpr int NEP = R1 * HE;                     // edge pairs of the tile
  static constexpr int PR = TXW + 8;                      // pitch of R and B2 (ratio columns 0 .. TXW + 2C - 1)
  static constexpr int CH2 = PR / 4;
  static constexpr int RSZ = R1 * PR;
  static constexpr int NIT2 = TY * CH2;
  static constexpr int XIT2 = cdiv(NIT2, NT);
  static constexpr int B2SZ = TY * PR;
  static constexpr int RY = 3;                            // output rows per y-pass item
  static constexpr int NG1 = cdiv(R1, RY) * CH;           // y1 items
  static constexpr int NG2 = cdiv(TY, RY) * CH2;          // y2 items
  // LDS map (floats): ring | B1 | R | B2 | dump
  static constexpr int OFF_B1 = kRing * ASZ;
  static constexpr int OFF_R = OFF_B1 + B1SZ;
  static constexpr int OFF_B2 = OFF_R + RSZ;
  static constexpr int OFF_DUMP = OFF_B2 + B2SZ;          // 1 KB: where glds of waves past the window land
  static constexpr int TOTAL = OFF_DUMP + 256;
  static constexpr int NY = 2 * NP1;                      // y loads per thread and iteration
  static constexpr int NXC = 2 * NP2;                     // x (centre) loads
  static constexpr int NTAP = PYX * PZ;                   // (z, x) taps of one stage
  static_assert(TY % NW == 0 && NCG == 2, "rows per wave; pairs are the two column groups of a row");
  static_assert(NFULL >= 0 && NFULL <= NW && (NW - NFULL) * 64 >= NEP && E % 2 == 0, "the edge pairs fit the short waves");
  static_assert(TOTAL * 4 <= 160 * 1024, "LDS per workgroup");
  static_assert(2 * C <= WL && WL <= lsr::kSepOriginCol && 2 * C <= 8, "halo columns");
  static_assert(SL + 2 * (NY + NXC) <= 63, "vmcnt is a 6-bit counter");
  static_assert(OFF_B1 % 4 == 0 && OFF_R % 4 == 0 && OFF_B2 % 4 == 0, "aligned buffers");
  static_assert(PZ <= lsr::kYsepTapGroup && PYX <= lsr::kYsepMaxPYX, "the tap block's groups");
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float fast_rcp(float d) {
  float r = __builtin_amdgcn_rcpf(d);
  return fmaf(fmaf(-d, r, 1.0f), r, r);
}
__device__ __forceinline__ f32x2 splat(float a) { return f32x2{a, a}; }
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 fast_rcp2(f32x2 d) {
  const f32x2 r = f32x2{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  return pk_fma(pk_fma(-d, r, splat(1.0f)), r, r);
}

// ---- hand-managed memory operations (as rl_fused_sep.hip) ----------------------------------------
template <int IMM>
__device__ __forceinline__ void gload(float& dst, const float* sbase, int voff) {
  asm volatile("global_load_dword %0, %1, %2 offset:%3" : "+v"(dst) : "v"(voff), "s"(sbase), "n"(IMM) : "memory");
}
template <int IMM>
__device__ __forceinline__ void gstore(float* sbase, int voff, float v) {
  asm volatile("global_store_dword %0, %1, %2 offset:%3 nt" : : "v"(voff), "v"(v), "s"(sbase), "n"(IMM) : "memory");
}
__device__ __forceinline__ void glds_x4(const float* sbase, int voff, unsigned lds_byte_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               :
               : "v"(voff), "s"(sbase), "s"(lds_byte_addr)
               : "memory");  // (m0 is a reserved register: hipcc sets it right at each of its own uses)
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}
template <int K>
__device__ __forceinline__ void tie(float (&a)[K]) {
#pragma unroll
  for (int i = 0; i < K; ++i) asm volatile("" : "+v"(a[i]));
}
// The FMAs of a tap group have no side effect, and their results are next used a whole plane later: left
// alone, the optimiser sinks them to the end of the loop body -- away from the taps, which were fetched in
// place (volatile) and would all have to stay live until then.  Passing an accumulator through an empty
// volatile asm pins its computation where it is written.
__device__ __forceinline__ void pin(f32x2& a) { asm volatile("" : "+v"(a)); }
__device__ __forceinline__ void pin(float& a) { asm volatile("" : "+v"(a)); }
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

// H^T 1 at (z, y, x) from the prefix-sum table of the caller's pz x py x px PSF (as correlate_dense.hip)
__device__ float dense_norm(const YsepArgs& p, const double* P, int z, int y, int x) {
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

// dst[r][g] = sum_b w[b] * src[r + b][g] for rows r < rows and 16-byte chunks g < chunks (both arrays `chunks`
// wide); item i -> (row group i / chunks, chunk i % chunks), `items` = ceil(rows / RY) * chunks of them.
// Per output the chain is w[0] * v, then FMAs in tap order.
template <int RY, int PYX, int NT>
__device__ __forceinline__ void ypass(const f32x4* src, f32x4* dst, int chunks, int rows, int items, const float (&w)[PYX],
                                      int tid) {
  for (int i = tid; i < items; i += NT) {   // (one round, or two for the widest windows)
    const int rg = i / chunks, g = i - rg * chunks;
    const int r0 = rg * RY;
    const f32x4* in = src + r0 * chunks + g;
    f32x4 a[RY + PYX - 1];
#pragma unroll
    for (int k = 0; k < RY + PYX - 1; ++k) a[k] = in[k * chunks];   // (rows past the last one: stale LDS, results dropped)
#pragma unroll
    for (int r = 0; r < RY; ++r) {
      f32x2 lo = splat(w[0]) * f32x2{a[r].x, a[r].y}, hi = splat(w[0]) * f32x2{a[r].z, a[r].w};
#pragma unroll
      for (int b = 1; b < PYX; ++b) {
        lo = pk_fma(splat(w[b]), f32x2{a[r + b].x, a[r + b].y}, lo);
        hi = pk_fma(splat(w[b]), f32x2{a[r + b].z, a[r + b].w}, hi);
      }
      if (r0 + r < rows) dst[(r0 + r) * chunks + g] = f32x4{lo.x, lo.y, hi.x, hi.y};
    }
  }
}

// Where the two halves of a thread's packed pair i sit: NCG == 2: row i, column groups 0 and 1; NCG == 1: rows
// 2 i and 2 i + 1 of the one column group.  (row of half h, its column offset in floats)
template <int NCG>
__device__ __forceinline__ constexpr int pair_row(int i, int h) { return NCG == 2 ? i : 2 * i + h; }
template <int NCG>
__device__ __forceinline__ constexpr int pair_col(int h) { return NCG == 2 ? 64 * h : 0; }

// STATS: the iteration's reduction scalars (correlate_common.hpp: RlStats) are summed in the epilogue and added to p.stats.
template <int PZ, int PYX, int NW, int NCG, bool STATS>
__global__ __launch_bounds__(64 * NW, 2) void rl_fused_ysep_kernel(YsepArgs p) {   // (2 waves per SIMD: <= 256 VGPRs)
  using T = Geo<PZ, PYX, NW, NCG>;
  constexpr int C = T::C, CZ = T::CZ, TY = T::TY, NT = T::NT, TXW = T::TXW;
  constexpr int NP1 = T::NP1, NP2 = T::NP2, RPW = T::RPW, LP = T::NP1 - 1;
  constexpr int NY = T::NY, NXC = T::NXC, SL = T::SL;
  __shared__ f32x4 smem4[T::TOTAL / 4 + 1];
  float* const smem = reinterpret_cast<float*>(smem4);
  f32x4* const B1_4 = smem4 + T::OFF_B1 / 4;
  const float* const B1 = smem + T::OFF_B1;
  float* const Rw = smem + T::OFF_R;
  const f32x4* const R_4 = smem4 + T::OFF_R / 4;
  f32x4* const B2_4 = smem4 + T::OFF_B2 / 4;
  const float* const B2 = smem + T::OFF_B2;
  const unsigned lds_base =
      static_cast<unsigned>(reinterpret_cast<size_t>((__attribute__((address_space(3))) char*)smem4));

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // work items, longest first, XCD-contiguous, tiles in bands of 8 rows (rl_fused_sep.hip)
  auto xcd_contiguous = [](int b, int n) {
    const int per = n / 8, rem = n % 8;
    const int xcd = b % 8, idx = b / 8;
    return xcd * per + (xcd < rem ? xcd : rem) + idx;
  };
  const int Z = p.Z, Y = p.Y, X = p.X;
  int lin, zb, ze;
  if (static_cast<int>(blockIdx.x) < p.n_full) {
    lin = xcd_contiguous(blockIdx.x, p.n_full);
    zb = 0;
    ze = Z;
  } else {
    const int t = xcd_contiguous(blockIdx.x - p.n_full, gridDim.x - p.n_full);
    const int col = t / p.pieces;
    lin = p.n_full + col;
    zb = (t - col * p.pieces) * p.z_chunk;
    ze = min(zb + p.z_chunk, Z);
  }
  const int band = lin / (p.tiles_x * kBand);
  const int lb = lin - band * (p.tiles_x * kBand);
  const int band_h = min(kBand, p.tiles_y - band * kBand);
  const int tx = lb / band_h;
  const int ty = band * kBand + (lb - tx * band_h);
  const int x0 = tx * TXW, y0 = ty * TY;

  // taps: two stages (stage 1 = reversed taps, stage 2 = the PSF's) of kYsepTapStage floats: the (z, x) taps of column
  // offset c at [16 c + j], j = PZ - 1 - a, the y taps at kYsepTapY.  They are read through the constant address space,
  // i.e. by SCALAR loads (s_load_dwordx8 + x1 per column offset), one column offset ahead of the FMAs that use them: no
  // VALU slot is spent on a tap (round 3 held them across the lanes of four VGPRs and pulled every tap out with a
  // v_readlane: 140 of the ~1200 vector instructions per plane).  `opaque` (always 0) makes the address loop-variant, so
  // that the loads stay inside the plane loop -- hoisted, the 2 * PZ * PYX taps would have to live in SGPRs.
  typedef const float __attribute__((address_space(4))) cfloat;
  const cfloat* const taps_c = (const cfloat*)p.taps;
  int opaque = 0;

  // ---- staging (glds): chunk e = tid + NT k of the (AR x PA) window whose first element is (y0 - 2C, x0 - WL)
  const float* const x_tile = p.x + (static_cast<int64_t>(y0 - 2 * C) * p.pitch + (x0 - T::WL));
  // (byte offsets inside the plane, held across the loop: round 3 recomputed them per plane to save SL registers; since
  // the edge columns moved into the last pair there is room, and the divisions were ~30 vector instructions per plane)
  // (... except the largest instance, 11 x 9 x 9: its accumulators leave no room, tools/asm_hazards.py found it
  // spilling; it keeps recomputing)
  constexpr bool HOLD = !(PZ >= 11 && PYX >= 9);
  int tid_v = tid;
  auto s_voff_of = [&](int k) {
    const int e = min(tid_v + k * NT, T::NCH - 1);
    const int r = e / T::CH, c = e - r * T::CH;
    return (r * p.pitch + 4 * c) * 4;
  };
  int s_voff_r[HOLD ? SL : 1];
  if constexpr (HOLD) {
#pragma unroll
    for (int k = 0; k < SL; ++k) s_voff_r[k] = s_voff_of(k);
  }
  auto s_voff = [&](int k) { return HOLD ? s_voff_r[HOLD ? k : 0] : s_voff_of(k); };
  // ---- stage-1 points: the wave's ratio rows r1_row0 .. (NP1 or NP1 - 1 of them), ratio columns lane and lane + 64.
  // Ratio row r <-> tile row r - C; ratio column rho <-> tile column rho - C <-> window column rho - C + WL.
  const bool short_wave = wave >= T::NFULL;                         // scalar: NP1 - 1 rows and an edge pair
  const int r1_row0 = short_wave ? T::NFULL * NP1 + (wave - T::NFULL) * LP : wave * NP1;   // scalar
  const int t1_col = r1_row0 * T::PA + lane + T::SH1;               // B1 float index of the first x tap of (row0, lane)
  const int r_col = r1_row0 * T::PR + lane;                         // R float index of the same point
  const bool interior = x0 - C >= 0 && x0 + TXW + C <= X && y0 - C >= 0 && y0 + TY + C <= Y;
  // the LAST pair of a thread: (row r1_row0 + LP, columns lane / lane + 64) in the full waves, an edge pair in the
  // short ones -- so its two halves are addressed through per-lane registers: (row << 16) | ratio column of each half;
  // the B1 / R / y offsets are derived from them where they are used
  int lp_rc[2];
  {
    const int et = min(max((wave - T::NFULL) * 64 + lane, 0), T::NEP - 1);   // (threads past the last edge pair repeat it)
    const int er = et / T::HE, ec = TXW + 2 * (et - er * T::HE);
    lp_rc[0] = short_wave ? (er << 16) | ec : ((r1_row0 + LP) << 16) | lane;
    lp_rc[1] = short_wave ? (er << 16) | (ec + 1) : ((r1_row0 + LP) << 16) | (lane + 64);
  }
  auto lp_t1_of = [&](int h) { return (lp_rc[h] >> 16) * T::PA + (lp_rc[h] & 0xffff) + T::SH1; };   // B1 index of its first x tap
  auto lp_r_of = [&](int h) { return (lp_rc[h] >> 16) * T::PR + (lp_rc[h] & 0xffff); };               // R index
  auto lp_voff_of = [&](int h) { return ((lp_rc[h] >> 16) * p.y_pitch + (lp_rc[h] & 0xffff)) * 4; };  // from the y window's first element
  int lp_t1_r[2] = {0, 0}, lp_r_r[2] = {0, 0}, lp_voff_r[2] = {0, 0};
  if constexpr (HOLD) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      lp_t1_r[h] = lp_t1_of(h);
      lp_r_r[h] = lp_r_of(h);
      lp_voff_r[h] = lp_voff_of(h);
    }
  }
  auto lp_t1 = [&](int h) { return HOLD ? lp_t1_r[h] : lp_t1_of(h); };
  auto lp_r = [&](int h) { return HOLD ? lp_r_r[h] : lp_r_of(h); };
  auto lp_voff = [&](int h) { return HOLD ? lp_voff_r[h] : lp_voff_of(h); };
  // ---- stage-2 points: the wave's tile rows wave * RPW .. + RPW - 1, tile columns lane (+ 64)
  const int t2_col = (wave * RPW) * T::PR + lane;
  // rows of a wave differ by a wave-uniform stride: one lane offset register per stream, the row term goes
  // into the scalar base of each load / store
  const int lane_off = lane * 4;
  // in-plane interior test of the epilogue's normalisation (the caller's PSF extents)
  const int ry = p.py / 2, rx = p.px / 2, rz = p.pz / 2;
  // every point of the tile has all its in-plane taps inside the volume (wave-uniform: the fast path of the norm)
  const bool tile_norm_interior = x0 >= rx && x0 + TXW <= X - rx && y0 >= ry && y0 + TY <= Y - ry;
  const bool okc[2] = {x0 + lane < X, x0 + lane + pair_col<NCG>(1) < X};

  f32x2 acc1[PZ][NP1], acc2[PZ][NP2];
#pragma unroll
  for (int j = 0; j < PZ; ++j) {
#pragma unroll
    for (int i = 0; i < NP1; ++i) acc1[j][i] = splat(0.0f);
#pragma unroll
    for (int i = 0; i < NP2; ++i) acc2[j][i] = splat(0.0f);
  }
  lsr::RlStats st;
  // 1 / (H^T 1) of the thread's 2 * NP2 output points on plane o.  Almost every point of almost every plane is interior
  // (all taps land inside the volume: the tap sum); the rest take the prefix-sum table (global memory, a few hundred bytes
  // that stay in cache) -- one rolled loop, so that its eight reads and their branch exist once in the code, not once
  // per point.
  auto plane_norms = [&](int o, float (&rn)[2 * NP2]) __attribute__((always_inline)) {
    const float rfull = fast_rcp(p.norm_full);
#pragma unroll
    for (int i = 0; i < 2 * NP2; ++i) rn[i] = rfull;
    if (!(tile_norm_interior && o >= rz && o < Z - rz)) {   // wave-uniform
#pragma unroll 1
      for (int k = 0; k < 2 * NP2; ++k) {
        const int i = k >> 1, h = k & 1;
        const int row = NCG == 2 ? i : 2 * i + h, col = NCG == 2 ? 64 * h : 0;
        const int gy = min(y0 + wave * RPW + row, Y - 1), gx = min(x0 + lane + col, X - 1);
        const bool inside = o >= rz && o < Z - rz && gy >= ry && gy < Y - ry && gx >= rx && gx < X - rx;
        const float r = fast_rcp(inside ? p.norm_full : dense_norm(p, p.norm_table, o, gy, gx));
#pragma unroll
        for (int t = 0; t < 2 * NP2; ++t) rn[t] = k == t ? r : rn[t];
      }
    }
  };
  // The table walk costs eight dependent global reads per point and a full drain of the load pipeline -- per PLANE in
  // round 3, in every tile that touches the volume's border: 14 % of config 2's tiles ran ~5 x slower than the others
  // and set the launch time (4.0 ms with all arithmetic removed, against 2.4 ms for the separable kernel's skeleton).
  // H^T 1 depends on z only within the PSF's z radius of the volume's ends, so the norms of a plane with all z taps
  // inside are computed ONCE per workgroup, here, with the same formula (bit-identical values), and the walk runs for
  // the first and last `rz` planes only.
  float rn_mid[2 * NP2];
  plane_norms(rz < Z - rz ? rz : 0, rn_mid);   // (a volume thinner than the PSF has no such plane: rn_mid unused)
  float yv[2 * NP1], xc[2 * NP2];   // [2 i + h] = half h of pair i
#pragma unroll
  for (int i = 0; i < 2 * NP1; ++i) yv[i] = 0.0f;
#pragma unroll
  for (int i = 0; i < 2 * NP2; ++i) xc[i] = 0.0f;
  __builtin_amdgcn_sched_barrier(0);

  auto clampz = [&](int z) { return min(max(z, 0), Z - 1); };
  auto issue_glds = [&](int plane, int slot) {  // SL loads
    const float* src = x_tile + static_cast<int64_t>(clampz(plane)) * p.plane;
    const unsigned dst = lds_base + (slot * T::ASZ + wave * 64 * 4) * 4;
#pragma unroll
    for (int k = 0; k < SL; ++k) {
      const bool live = wave * 64 + k * NT < T::NCH;  // wave-uniform
      glds_x4(src, s_voff(k), live ? dst + k * NT * 16 : lds_base + T::OFF_DUMP * 4);
    }
  };
  const float* const xc_tile = p.x + (static_cast<int64_t>(y0) * p.pitch + x0);
  const float* const y_tile = p.y + (static_cast<int64_t>(y0 - C) * p.y_pitch + (x0 - C));
  float* const o_tile = p.out + (static_cast<int64_t>(y0) * p.out_pitch + x0);
  auto issue_xc = [&](int o) {  // NXC loads: x at the output points of plane o
    const float* base = xc_tile + (static_cast<int64_t>(clampz(o)) * p.plane + static_cast<int64_t>(wave * RPW) * p.pitch);
#pragma unroll
    for (int i = 0; i < NP2; ++i) {
      gload<0>(xc[2 * i], base + pair_row<NCG>(i, 0) * p.pitch, lane_off);
      gload<4 * pair_col<NCG>(1)>(xc[2 * i + 1], base + pair_row<NCG>(i, 1) * p.pitch, lane_off);
    }
  };
  auto issue_y = [&](int q) {  // NY loads: y at the ratio points of plane q
    const float* base = y_tile + static_cast<int64_t>(clampz(q)) * p.y_plane;
#pragma unroll
    for (int i = 0; i < LP; ++i) {
      gload<0>(yv[2 * i], base + (r1_row0 + i) * p.y_pitch, lane_off);   // (scalar rows)
      gload<4 * 64>(yv[2 * i + 1], base + (r1_row0 + i) * p.y_pitch, lane_off);
    }
    gload<0>(yv[2 * LP], base, lp_voff(0));
    gload<0>(yv[2 * LP + 1], base, lp_voff(1));
  };

  const int q_lo = max(zb - CZ, 0), q_hi = min(ze - 1 + CZ, Z - 1);  // ratio planes that matter
  const int p_lo = max(q_lo - CZ, 0);
  const int p_hi = ze + 2 * CZ;  // inclusive: the iteration that completes output plane ze - 1

  int slot = 0;
  issue_glds(p_lo, 0);
  issue_xc(p_lo - 1 - 2 * CZ);
  issue_y(p_lo - CZ);
  issue_glds(p_lo + 1, 1);
  issue_xc(p_lo - 1 - 2 * CZ);
  issue_y(p_lo - CZ);
  wait_vm<SL + 2 * (NXC + NY)>();  // this wave's part of plane p_lo has landed
  lds_barrier();

  for (int pz = p_lo; pz <= p_hi; ++pz) {
    asm volatile("" : "+s"(opaque));   // loop-variant for the optimiser: the tap loads stay in the loop
    const cfloat* const taps1 = taps_c + opaque;                       // stage 1
    const cfloat* const taps2 = taps_c + lsr::kYsepTapStage + opaque;  // stage 2
    if constexpr (!HOLD) {   // loop-variant for the optimiser: no hoisting of the offsets derived from these
      asm volatile("" : "+v"(lp_rc[0]), "+v"(lp_rc[1]));
      asm volatile("" : "+v"(tid_v));
    }
    const int qr = pz - 1 - CZ;      // ratio plane in R (written by the previous iteration)
    const int o = qr - CZ;           // output plane completed by this iteration
    const int q = pz - CZ;           // ratio plane completed by this iteration
    const bool x_live = pz < Z && pz <= q_hi + CZ;
    const bool r_live = qr >= q_lo && qr <= q_hi;
    const int slot2 = slot >= 1 ? slot - 1 : 2;           // (slot + 2) % 3

    // ---------------- phase A: stage the plane after next; the two y passes ----------------
    issue_glds(pz + 2, slot2);
    // y passes, LDS -> LDS: an item is one 16-byte column chunk of RY consecutive output rows -- RY + 2C input
    // chunks instead of RY * (2C + 1) (these passes are what the LDS pipe spends the phase on)
#ifdef LSR_YSEP_PROBE_NOYPASS   // measurement build (results wrong): the two LDS -> LDS y passes are skipped
    if (false) {
#else
    if (x_live) {
#endif
      float w1y[PYX];
#pragma unroll
      for (int b = 0; b < PYX; ++b) w1y[b] = taps1[lsr::kYsepTapY + b];
      ypass<T::RY, PYX, NT>(smem4 + slot * (T::ASZ / 4), B1_4, T::CH, T::R1, T::NG1, w1y, tid);
    } else {  // a plane outside the volume: zeros
#pragma unroll
      for (int k = 0; k < T::XIT1; ++k)
        if (k + 1 < T::XIT1 || tid + k * NT < T::NIT1) B1_4[tid + k * NT] = f32x4{0, 0, 0, 0};
    }
#ifdef LSR_YSEP_PROBE_NOYPASS
    if (false) {
#else
    if (r_live) {
#endif
      float w2y[PYX];
#pragma unroll
      for (int b = 0; b < PYX; ++b) w2y[b] = taps2[lsr::kYsepTapY + b];
      ypass<T::RY, PYX, NT>(R_4, B2_4, T::CH2, TY, T::NG2, w2y, tid);
    } else {
#pragma unroll
      for (int k = 0; k < T::XIT2; ++k)
        if (k + 1 < T::XIT2 || tid + k * NT < T::NIT2) B2_4[tid + k * NT] = f32x4{0, 0, 0, 0};
    }
    lds_barrier();

    // ---------------- phase B ----------------
    // stage 2: absorb t2 of ratio plane qr into the pending output planes, finish output plane o
    {
      // column c of the thread's pairs one tap group ahead of its FMAs; a scheduling fence per group keeps
      // the PZ taps of ONE group in SGPRs (all PZ * PYX at once spill)
      const float* base = B2 + t2_col;
      auto ld = [&](int i, int c) {
        return f32x2{base[pair_row<NCG>(i, 0) * T::PR + c], base[pair_row<NCG>(i, 1) * T::PR + pair_col<NCG>(1) + c]};
      };
      f32x2 vb[2][NP2];   // two register sets, indexed by the (compile-time) parity of c: no copies
      float wq[2][PZ];    // ... and two SGPR sets of taps
#pragma unroll
      for (int i = 0; i < NP2; ++i) vb[0][i] = ld(i, 0);
#pragma unroll
      for (int j = 0; j < PZ; ++j) wq[0][j] = taps2[j];
#ifdef LSR_YSEP_PROBE_NONEST2   // measurement build (results wrong): one column offset of the stage-2 nest instead of PYX
      constexpr int kCols2 = 1;
#else
      constexpr int kCols2 = PYX;
#endif
#pragma unroll
      for (int c = 0; c < kCols2; ++c) {
        f32x2 (&v)[NP2] = vb[c & 1];
        if (c + 1 < PYX) {
#pragma unroll
          for (int i = 0; i < NP2; ++i) vb[(c + 1) & 1][i] = ld(i, c + 1);
#pragma unroll
          for (int j = 0; j < PZ; ++j) wq[(c + 1) & 1][j] = taps2[lsr::kYsepTapGroup * (c + 1) + j];
        }
#pragma unroll
        for (int j = 0; j < PZ; ++j) {
          const float ws = wq[c & 1][j];
          const f32x2 w = splat(ws);
#pragma unroll
          for (int i = 0; i < NP2; ++i) {
            if (c == 0) acc2[j][i] = j + 1 < PZ ? pk_fma(w, v[i], acc2[j + 1][i]) : w * v[i];
            else acc2[j][i] = pk_fma(w, v[i], acc2[j][i]);
            pin(acc2[j][i]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // x(o): requested by the previous iteration; issued since: its y loads, this iteration's glds
      wait_vm<NY + SL>();
      tie(xc);
      if (o >= zb && o < ze) {
        float* obase = o_tile + (static_cast<int64_t>(o) * p.out_plane + static_cast<int64_t>(wave * RPW) * p.out_pitch);
        float rn[2 * NP2];
        if (o >= rz && o < Z - rz) {   // all z taps inside: the norms computed once, above
#pragma unroll
          for (int i = 0; i < 2 * NP2; ++i) rn[i] = rn_mid[i];
        } else {
          plane_norms(o, rn);
        }
#pragma unroll
        for (int i = 0; i < NP2; ++i) {
          // (x * u) * rcp(H^T 1): the two-launch UPDATE's order
          const float xu0 = xc[2 * i] * acc2[0][i].x, xu1 = xc[2 * i + 1] * acc2[0][i].y;
          const float v0 = xu0 * rn[2 * i], v1 = xu1 * rn[2 * i + 1];
          if (y0 + wave * RPW + pair_row<NCG>(i, 0) < Y && okc[0]) {   // (row test wave-uniform)
            gstore<0>(obase + pair_row<NCG>(i, 0) * p.out_pitch, lane_off, v0);
            if constexpr (STATS) st.add(xc[2 * i], xu0, v0);
          }
          if (y0 + wave * RPW + pair_row<NCG>(i, 1) < Y && okc[NCG == 2 ? 1 : 0]) {
            gstore<4 * pair_col<NCG>(1)>(obase + pair_row<NCG>(i, 1) * p.out_pitch, lane_off, v1);
            if constexpr (STATS) st.add(xc[2 * i + 1], xu1, v1);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);  // the refill reuses xc
      issue_xc(o + 1);
    }
    // stage 1: absorb t1 of x plane pz into the pending ratio planes, finish ratio plane q
    {
      const float* base = B1 + t1_col;
      const float* const lp0 = B1 + lp_t1(0);
      const float* const lp1 = B1 + lp_t1(1);
      auto ld = [&](int i, int c) {
        if (i == LP) return f32x2{lp0[c], lp1[c]};
        return f32x2{base[i * T::PA + c], base[i * T::PA + 64 + c]};
      };
      f32x2 vb[2][NP1];
#pragma unroll
      for (int i = 0; i < NP1; ++i) vb[0][i] = ld(i, 0);
      float wq[2][PZ];
#pragma unroll
      for (int j = 0; j < PZ; ++j) wq[0][j] = taps1[j];
#ifdef LSR_YSEP_PROBE_NONEST1   // ... of the stage-1 nest
      constexpr int kCols1 = 1;
#else
      constexpr int kCols1 = PYX;
#endif
#pragma unroll
      for (int c = 0; c < kCols1; ++c) {
        f32x2 (&v)[NP1] = vb[c & 1];
        if (c + 1 < PYX) {
#pragma unroll
          for (int i = 0; i < NP1; ++i) vb[(c + 1) & 1][i] = ld(i, c + 1);
#pragma unroll
          for (int j = 0; j < PZ; ++j) wq[(c + 1) & 1][j] = taps1[lsr::kYsepTapGroup * (c + 1) + j];
        }
#pragma unroll
        for (int j = 0; j < PZ; ++j) {
          const float ws = wq[c & 1][j];
          const f32x2 w = splat(ws);
#pragma unroll
          for (int i = 0; i < NP1; ++i) {
            if (c == 0) acc1[j][i] = j + 1 < PZ ? pk_fma(w, v[i], acc1[j + 1][i]) : w * v[i];
            else acc1[j][i] = pk_fma(w, v[i], acc1[j][i]);
            pin(acc1[j][i]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // y(q): requested by the previous iteration; issued since: this iteration's glds and x loads
      wait_vm<SL + NXC>();
      tie(yv);
      const bool q_in = q >= q_lo && q <= q_hi;  // wave-uniform; planes outside are zero
      if (q_in && interior) {
#pragma unroll
        for (int i = 0; i < NP1; ++i) {
          const f32x2 r = f32x2{yv[2 * i], yv[2 * i + 1]} * fast_rcp2(acc1[0][i] + splat(p.eps));
          if (i == LP) {
            Rw[lp_r(0)] = r.x;
            Rw[lp_r(1)] = r.y;
          } else {
            Rw[r_col + i * T::PR] = r.x;
            Rw[r_col + i * T::PR + 64] = r.y;
          }
        }
      } else {
        // border tiles and planes outside the volume: ratio is zero wherever its point is outside
        const int gx0 = x0 + lane - C;
        const bool inc[2] = {q_in && gx0 >= 0 && gx0 < X, q_in && gx0 + 64 >= 0 && gx0 + 64 < X};
#pragma unroll
        for (int i = 0; i < NP1; ++i) {
          const f32x2 r = f32x2{yv[2 * i], yv[2 * i + 1]} * fast_rcp2(acc1[0][i] + splat(p.eps));
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            if (i == LP) {   // the per-lane pair: its own row and column
              const int gy = y0 + (lp_rc[h] >> 16) - C, gx = x0 + (lp_rc[h] & 0xffff) - C;
              Rw[lp_r(h)] = (q_in && gy >= 0 && gy < Y && gx >= 0 && gx < X) ? (h ? r.y : r.x) : 0.0f;
            } else {
              const int gy = y0 + r1_row0 + i - C;
              const bool row_in = gy >= 0 && gy < Y;  // wave-uniform
              Rw[r_col + i * T::PR + 64 * h] = (row_in && inc[h]) ? (h ? r.y : r.x) : 0.0f;
            }
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);  // the refill reuses yv
      issue_y(q + 1);
    }
    // plane pz+1 (requested one iteration ago); issued since: x, y loads of the previous iteration, this
    // iteration's glds, x and y loads
    wait_vm<SL + 2 * (NXC + NY)>();
    lds_barrier();
    slot = slot == 2 ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing may land after the wave has ended
  if constexpr (STATS) {
    lsr::keep_until_here(xc);    // (in-flight prefetches: correlate_common.hpp, keep_until_here)
    lsr::keep_until_here(yv);
    st.pin();
    lsr::rl_stats_flush<NW>(st, smem + T::OFF_B2, p.stats);
  }
}

template <int PZ, int PYX>
bool launch_one(const YsepArgs& p, dim3 grid, hipStream_t s) {
  if (p.narrow) return false;   // (the 256-thread shape <PZ, PYX, 4, 1> is no longer instantiated)
  if (p.stats != nullptr) hipLaunchKernelGGL((rl_fused_ysep_kernel<PZ, PYX, 8, 2, true>), grid, dim3(512), 0, s, p);
  else hipLaunchKernelGGL((rl_fused_ysep_kernel<PZ, PYX, 8, 2, false>), grid, dim3(512), 0, s, p);
  return true;
}

}  // namespace

namespace lsr {

#define LSR_CAT2(a, b) a##b
#define LSR_CAT(a, b) LSR_CAT2(a, b)
bool LSR_CAT(launch_ysep_pz, LSR_YSEP_PZ)(int pyx, const YsepArgs& p, unsigned blocks, hipStream_t s) {
  constexpr int PZ = LSR_YSEP_PZ;
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
