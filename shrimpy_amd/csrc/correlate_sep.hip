// Separable (rank-1) 3-D PSF correlation with fused Richardson-Lucy epilogues -- the HBM-bound
// headline kernel.  Compile-time tap counts; one translation unit per PZ (-DLSR_SEP_PZ=n).
//
// The input volume carries a ZERO HALO (lsr_sep_padded_shape): zero padding of the correlation is
// real memory, so every load in this kernel is unconditional, in bounds and 16-byte aligned.
//
// A 512-thread workgroup (8 waves) owns a 32 (y) x 64 (x) column and marches along z, one
// workgroup barrier per plane.  Iteration zi:
//   commit   : plane zi+2, fetched two iterations ago into registers, -> LDS A[zi&1] (a linear
//              16-B-per-thread copy of the (32+PY-1) x 72 window).
//   prefetch : plane zi+4 -> the registers just freed (global_load_dwordx4, 288-B row runs), and
//              the `aux` values of the NEXT iteration's output plane.  Two planes per workgroup
//              are always in flight; nothing waits on them inside this iteration.
//   x pass   : plane zi+1, A[(zi+1)&1] -> B[(zi+1)&1].  A thread owns (row, 4 consecutive x); its
//              4+PX-1 inputs come from 3..5 ds_read_b128.  Lanes are assigned to items by the
//              HARDWARE's b128 lane groups, so each group reads one contiguous 256-B row run:
//              conflict-free for any pitch.  B's pitch (80) keeps the b128 writes conflict-free.
//   y pass   : plane zi from B[zi&1]; thread = one column x 4 rows (4+PY-1 ds_read_b32).
//   z        : PZ pending output planes per point in registers; one FMA per pending plane both
//              shifts the window and absorbs this plane.
//   epilogue : completed plane zi-PZ/2: ratio = y*rcp(c+eps) or x*c*rcp(H^T 1), strided store.
//
// THE MEMORY PIPELINE IS MANAGED BY HAND.  With loads and stores pending on the same counter,
// hipcc (ROCm 7.2) treats vmcnt as out of order and turns every wait for a load into vmcnt(0) --
// a full drain, store acknowledgements included, once per plane (measured in round 1: it capped
// the kernel at 47 % of the HBM peak).  So the loads are inline asm (invisible to the compiler's
// wait insertion), each consumer is preceded by a hand-counted `s_waitcnt vmcnt(N)` tied to the
// destination registers, and the barrier is a raw s_barrier.  vmcnt retires in issue order on
// gfx950 (loads and stores alike), so N = the number of LOADS issued after the wanted one is a
// safe bound: younger stores only make the wait return a little later, never too early.
//
// Algorithmic HBM bytes: 12 per voxel per launch (in + aux + out).  Real traffic adds the in-plane
// halo of `in` ((32+PY-1)*72/(32*64) = 1.34x for 7x7), largely L2 hits thanks to the XCD-aware
// tile order, and PZ-1 planes per z-chunk.

#include "common.hpp"
#include "correlate_common.hpp"

#ifndef LSR_SEP_PZ
#error "compile with -DLSR_SEP_PZ=<odd tap count along z>"
#endif

namespace {

using lsr::SepArgs;

constexpr int kTY = lsr::kSepTileY;      // 32
constexpr int kTX = lsr::kSepTileX;      // 64
constexpr int kRun = 4;                  // consecutive y per thread in the y / z passes
constexpr int kWaves = kTY / kRun;       // 8
constexpr int kThreads = 64 * kWaves;    // 512
constexpr int kBand = 8;                 // tile rows per band of the tile walk (see the kernel)

template <int PY, int PX>
struct Tile {
  static constexpr int AR = kTY + PY - 1;                        // input rows of a tile
  static constexpr int AC = kTX + PX - 1;                        // input cols of a tile
  static constexpr int PA0 = (AC + 3) / 4 * 4;
  static constexpr int PA = PA0 < 72 ? 72 : PA0;                 // staged cols == LDS pitch of A
  static constexpr int CH = PA / 4;                              // 16-B chunks per staged row
  static constexpr int NCH = AR * CH;                            // chunks per plane
  static constexpr int SL = (NCH + kThreads - 1) / kThreads;     // chunks per thread
  static constexpr int WIN = 4 + PX - 1;                         // inputs of an x-pass item
  static constexpr int NPIECE = (WIN + 3) / 4;                   // ds_read_b128 per item
  static constexpr int PB = 80;                                  // LDS pitch of B (16 mod 32)
  static constexpr int ASZ = AR * PA;
  static constexpr int BSZ = AR * PB;
  static_assert(4 * 15 + 4 * NPIECE <= PA, "x-pass reads stay inside a staged row");
  static_assert(AR <= 2 * kTY, "at most two x-pass items per thread");
  static_assert(SL == 2, "the hand-counted waits assume two staging loads per thread");
};

typedef float f32x4 __attribute__((ext_vector_type(4)));  // native 16-byte vector (one VGPR quad)

__device__ __forceinline__ float fast_rcp(float d) {
  float r = __builtin_amdgcn_rcpf(d);    // v_rcp_f32, 1 ulp
  return fmaf(fmaf(-d, r, 1.0f), r, r);  // + one Newton step
}

// ---- hand-managed global loads: scalar base + unsigned 32-bit byte offset per lane -------------
__device__ __forceinline__ void gload_x4(f32x4& dst, const float* sbase, int voff_bytes) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff_bytes), "s"(sbase) : "memory");
}
__device__ __forceinline__ void gload_x1(float& dst, const float* sbase, int voff_bytes) {
  asm volatile("global_load_dword %0, %1, %2" : "=v"(dst) : "v"(voff_bytes), "s"(sbase) : "memory");
}
// Wait until at most N vector-memory operations are outstanding; the registers are passed
// through so that every later use depends on this statement.
template <int N>
__device__ __forceinline__ void wait_loads(f32x4& a, f32x4& b) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_loads(float (&a)[kRun], float& b) {
  asm volatile("s_waitcnt vmcnt(%5)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b) : "n"(N) : "memory");
}
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's LDS writes have landed
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

// ds_read_b128 is serviced in four fixed 16-lane groups: {0-3,12-15,20-27}, {4-11,16-19,28-31} and
// the same +32 (MI355X_MICROARCH.md, LDS).  Returns the group (0..3) and the rank inside it (0..15).
__device__ __forceinline__ void b128_lane_group(int lane, int& group, int& rank) {
  const int m = lane & 31;
  int gs, r;
  if (m < 4) { gs = 0; r = m; }
  else if (m < 12) { gs = 1; r = m - 4; }
  else if (m < 16) { gs = 0; r = m - 8; }
  else if (m < 20) { gs = 1; r = m - 8; }
  else if (m < 28) { gs = 0; r = m - 12; }
  else { gs = 1; r = m - 16; }
  group = 2 * (lane >> 5) + gs;
  rank = r;
}

template <int PZ, int PY, int PX, int EPI>
__global__ __launch_bounds__(kThreads) void correlate_sep_kernel(SepArgs p) {
  using T = Tile<PY, PX>;
  // LDS is addressed in 16-byte units wherever b128 accesses are wanted, so their alignment is
  // part of the type (hipcc otherwise splits them into ds_read2_b32 / ds_read2_b64)
  __shared__ f32x4 bufA4[2 * T::ASZ / 4];
  __shared__ f32x4 bufB4[2 * T::BSZ / 4];
  float* const bufB = reinterpret_cast<float*>(bufB4);

  // loads per iteration besides the two staging loads: aux rows (+ nz for the update)
  constexpr int NA = EPI == LSR_EPI_NONE ? 0 : (EPI == LSR_EPI_UPDATE ? kRun + 1 : kRun);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar

  // XCD-aware tile order: workgroups b, b+8, ... share an XCD (round-robin dispatch), so give
  // each XCD a contiguous run of tiles: x-neighbours then share their halo columns in one L2.
  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x;
    const int per = nblk / 8, rem = nblk % 8;
    const int xcd = bid % 8, idx = bid / 8;
    bid = xcd * per + (xcd < rem ? xcd : rem) + idx;  // XCD k owns per + (k < rem) tiles
  }
  // ... and walk the (y, x) tile grid in bands of 8 tile rows, column-major inside a band: any
  // 64 consecutive tiles (= the workgroups an XCD keeps resident) then form an 8 x 8 patch, so
  // halo rows AND halo columns are shared inside one L2.
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
  constexpr int cz = PZ / 2, cy = PY / 2, cx = PX / 2;

  // taps: wave-uniform -> SGPRs.  The caller's pz/py/px taps sit centred in PZ/PY/PX.
  float wz[PZ], wy[PY], wx[PX];
  {
    const int oz = (PZ - p.pz) / 2, oy = (PY - p.py) / 2, ox = (PX - p.px) / 2;
#pragma unroll
    for (int i = 0; i < PZ; ++i) wz[i] = (i >= oz && i < oz + p.pz) ? p.wz[i - oz] : 0.0f;
#pragma unroll
    for (int i = 0; i < PY; ++i) wy[i] = (i >= oy && i < oy + p.py) ? p.wy[i - oy] : 0.0f;
#pragma unroll
    for (int i = 0; i < PX; ++i) wx[i] = (i >= ox && i < ox + p.px) ? p.wx[i - ox] : 0.0f;
  }

  // ---- staging: chunk e = tid + 512*k of the (AR x PA) window; the LDS image is linear in e.
  // Threads past the last chunk re-fetch and re-write the last one (benign): nothing conditional.
  // Global side: scalar base = the window's first element, per-lane byte offset >= 0.
  const float* const in_tile = p.in + (static_cast<int64_t>(y0 - cy) * p.in_pitch + (x0 - cx));
  int s_voff[T::SL];  // byte offset from in_tile (+ z * plane)
  int s_loff[T::SL];  // 16-byte chunk index inside an A buffer
#pragma unroll
  for (int k = 0; k < T::SL; ++k) {
    const int e = min(tid + k * kThreads, T::NCH - 1);
    const int r = e / T::CH, c = e - r * T::CH;
    s_voff[k] = (r * p.in_pitch + 4 * c) * 4;
    s_loff[k] = e;
  }
  f32x4 st0[T::SL], st1[T::SL];  // plane q is staged in set q & 1

  // ---- x pass: lane -> (row group, x group) by hardware b128 lane group
  int xg, xq;
  b128_lane_group(lane, xg, xq);
  const int xrow0 = 4 * wave + xg;                 // item 0: rows 0..31
  const bool has1 = kTY + xrow0 < T::AR;           // item 1: rows 32.. (only some lanes)
  const int xa = xrow0 * (T::PA / 4) + xq;         // 16-B units; + kTY*PA/4 for item 1
  const int xb = xrow0 * (T::PB / 4) + xq;         // 16-B units; + kTY*PB/4 for item 1

  // ---- y / z pass and epilogue: thread owns column `lane`, rows 4*wave .. 4*wave+3
  const int ycol = (wave * kRun) * T::PB + lane;
  const int gx_out = x0 + lane;
  const bool xok = gx_out < X;
  const int gxc = min(gx_out, X - 1);              // clamped: aux loads are always in bounds
  const int gy_out0 = y0 + wave * kRun;            // scalar
  int a_voff[kRun], o_off[kRun];
  bool ok[kRun];
#pragma unroll
  for (int m = 0; m < kRun; ++m) {
    const int gy = gy_out0 + m;
    ok[m] = xok && gy < Y;
    a_voff[m] = (min(gy, Y - 1) * p.aux_pitch + gxc) * 4;
    o_off[m] = min(gy, Y - 1) * p.out_pitch + gxc;
  }
  float rnyx[kRun];  // reciprocal of the in-plane part of H^T 1 (UPDATE epilogue)
#pragma unroll
  for (int m = 0; m < kRun; ++m) rnyx[m] = 0.0f;
  if constexpr (EPI == LSR_EPI_UPDATE) {
    const float nxv = p.nx[gxc];
#pragma unroll
    for (int m = 0; m < kRun; ++m) rnyx[m] = fast_rcp(p.ny[min(gy_out0 + m, Y - 1)] * nxv);
  }

  // pending output planes: acc[j][m] <-> z_out = zi - cz + j once plane zi is absorbed
  float acc[PZ][kRun];
#pragma unroll
  for (int j = 0; j < PZ; ++j)
#pragma unroll
    for (int m = 0; m < kRun; ++m) acc[j][m] = 0.0f;
  // aux (and nz) of output plane zi - cz live in set zi & 1, requested one iteration ahead
  float aux0[kRun], aux1[kRun];
#pragma unroll
  for (int m = 0; m < kRun; ++m) aux0[m] = aux1[m] = 0.0f;
  float nzv0 = 1.0f, nzv1 = 1.0f;

  const int zi_begin = max(zb - cz, 0);
  const int zi_end = ze + cz;  // exclusive; planes >= Z contribute zeros

  auto fetch = [&](int zplane, f32x4 (&st)[T::SL]) {  // 2 loads
    const float* src = in_tile + static_cast<int64_t>(min(max(zplane, 0), Z - 1)) * p.in_plane;
    gload_x4(st[0], src, s_voff[0]);
    gload_x4(st[1], src, s_voff[1]);
  };
  auto fetch_aux = [&](int zout, float (&aux)[kRun], float& nzv) {  // NA loads
    if constexpr (EPI != LSR_EPI_NONE) {
      const int zc_ = min(max(zout, 0), Z - 1);
      const float* a = p.aux + static_cast<int64_t>(zc_) * p.aux_plane;
#pragma unroll
      for (int m = 0; m < kRun; ++m) gload_x1(aux[m], a, a_voff[m]);
      if constexpr (EPI == LSR_EPI_UPDATE) gload_x1(nzv, p.nz + zc_, 0);
    }
  };

  // One iteration; `par` = zi & 1 is a literal at both call sites.
  auto iteration = [&](const int zi, const int par, f32x4 (&st)[T::SL], float (&aux_use)[kRun],
                       float& nz_use, float (&aux_load)[kRun], float& nz_load) {
    f32x4* A_commit = bufA4 + par * (T::ASZ / 4);            // plane zi+2
    const f32x4* A_x = bufA4 + (par ^ 1) * (T::ASZ / 4);     // plane zi+1
    f32x4* B_x = bufB4 + (par ^ 1) * (T::BSZ / 4);           // plane zi+1
    const float* B_y = bufB + par * T::BSZ;                  // plane zi

    // commit plane zi+2.  Its loads were issued two iterations ago; issued since: aux (NA),
    // the other staging set (2), aux (NA).
    wait_loads<2 + 2 * NA>(st[0], st[1]);
    A_commit[s_loff[0]] = st[0];
    A_commit[s_loff[1]] = st[1];
    __builtin_amdgcn_sched_barrier(0);  // the refill reuses these registers: keep it behind
    fetch(zi + 4, st);
    fetch_aux(zi + 1 - cz, aux_load, nz_load);

    // x pass of plane zi+1
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      if (it == 0 || has1) {
        const f32x4* src = A_x + xa + it * (kTY * T::PA / 4);
        float w[4 * T::NPIECE];
#pragma unroll
        for (int i = 0; i < T::NPIECE; ++i) {
          const f32x4 v = src[i];
          w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
        }
        f32x4 o;
        o.x = wx[0] * w[0];
        o.y = wx[0] * w[1];
        o.z = wx[0] * w[2];
        o.w = wx[0] * w[3];
#pragma unroll
        for (int c = 1; c < PX; ++c) {
          o.x = fmaf(wx[c], w[c], o.x);
          o.y = fmaf(wx[c], w[c + 1], o.y);
          o.z = fmaf(wx[c], w[c + 2], o.z);
          o.w = fmaf(wx[c], w[c + 3], o.w);
        }
        B_x[xb + it * (kTY * T::PB / 4)] = o;
      }
    }

    if (zi >= zi_begin && zi < zi_end) {  // wave-uniform
      float pl[kRun];
#pragma unroll
      for (int m = 0; m < kRun; ++m) pl[m] = 0.0f;
      if (zi < Z) {
        const float* col = B_y + ycol;
        float cv[kRun + PY - 1];
#pragma unroll
        for (int j = 0; j < kRun + PY - 1; ++j) cv[j] = col[j * T::PB];
#pragma unroll
        for (int m = 0; m < kRun; ++m) {
          float s = wy[0] * cv[m];
#pragma unroll
          for (int b = 1; b < PY; ++b) s = fmaf(wy[b], cv[m + b], s);
          pl[m] = s;
        }
      }
      // z: shift the pending planes and absorb this plane in the same FMA
#pragma unroll
      for (int j = 0; j < PZ - 1; ++j)
#pragma unroll
        for (int m = 0; m < kRun; ++m) acc[j][m] = fmaf(wz[PZ - 1 - j], pl[m], acc[j + 1][m]);
#pragma unroll
      for (int m = 0; m < kRun; ++m) acc[PZ - 1][m] = wz[0] * pl[m];

      const int z_out = zi - cz;
      if (z_out >= zb) {  // wave-uniform
        float* o = p.out + static_cast<int64_t>(z_out) * p.out_plane;  // scalar
        if constexpr (EPI != LSR_EPI_NONE) {
          // aux of this plane was requested in the previous iteration; issued since: this
          // iteration's staging loads (2) and aux (NA)
          wait_loads<2 + NA>(aux_use, nz_use);
        }
        if constexpr (EPI == LSR_EPI_RATIO) {
#pragma unroll
          for (int m = 0; m < kRun; ++m)
            if (ok[m]) o[o_off[m]] = aux_use[m] * fast_rcp(acc[0][m] + p.eps);
        } else if constexpr (EPI == LSR_EPI_UPDATE) {
          const float rz = fast_rcp(nz_use);
#pragma unroll
          for (int m = 0; m < kRun; ++m)
            if (ok[m]) o[o_off[m]] = aux_use[m] * acc[0][m] * (rz * rnyx[m]);
        } else {
#pragma unroll
          for (int m = 0; m < kRun; ++m)
            if (ok[m]) o[o_off[m]] = acc[0][m];
        }
      }
    }
    lds_barrier();  // A[par], B[par^1] complete; A[par^1], B[par] free
  };

  // Start two planes early at an even index: the extra leading iterations only move (valid,
  // clamped) data through the pipeline, nothing they produce is consumed.  The prologue issues
  // the same load sequence an iteration pair would, so the hand-counted waits hold from the start.
  const int zs = (zi_begin - 2) & ~1;
  fetch(zs + 2, st0);
  fetch_aux(zs - 1 - cz, aux1, nzv1);
  fetch(zs + 3, st1);
  fetch_aux(zs - cz, aux0, nzv0);
  for (int zi = zs; zi < zi_end; zi += 2) {
    iteration(zi, 0, st0, aux0, nzv0, aux1, nzv1);
    iteration(zi + 1, 1, st1, aux1, nzv1, aux0, nzv0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing may land after the wave has ended
}

template <int PZ, int PYX>
bool launch_one(const SepArgs& p, dim3 grid, hipStream_t s) {
  const dim3 block(kThreads);
  switch (p.epilogue) {
    case LSR_EPI_NONE:
      hipLaunchKernelGGL((correlate_sep_kernel<PZ, PYX, PYX, LSR_EPI_NONE>), grid, block, 0, s, p);
      return true;
    case LSR_EPI_RATIO:
      hipLaunchKernelGGL((correlate_sep_kernel<PZ, PYX, PYX, LSR_EPI_RATIO>), grid, block, 0, s, p);
      return true;
    case LSR_EPI_UPDATE:
      hipLaunchKernelGGL((correlate_sep_kernel<PZ, PYX, PYX, LSR_EPI_UPDATE>), grid, block, 0, s, p);
      return true;
    default:
      return false;
  }
}

}  // namespace

namespace lsr {

// One definition per translation unit: -DLSR_SEP_PZ=3 ... 15.
#define LSR_CAT2(a, b) a##b
#define LSR_CAT(a, b) LSR_CAT2(a, b)
bool LSR_CAT(launch_sep_pz, LSR_SEP_PZ)(int pyx, const SepArgs& p, unsigned blocks, hipStream_t s) {
  constexpr int PZ = LSR_SEP_PZ;
  const dim3 grid(blocks);
  switch (pyx) {
    case 3: return launch_one<PZ, 3>(p, grid, s);
    case 5: return launch_one<PZ, 5>(p, grid, s);
    case 7: return launch_one<PZ, 7>(p, grid, s);
    case 9: return launch_one<PZ, 9>(p, grid, s);
    case 11: return launch_one<PZ, 11>(p, grid, s);
    case 13: return launch_one<PZ, 13>(p, grid, s);
    case 15: return launch_one<PZ, 15>(p, grid, s);
    default: return false;
  }
}

}  // namespace lsr
