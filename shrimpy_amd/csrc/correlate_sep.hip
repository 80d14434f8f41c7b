// Separable (rank-1) 3-D PSF correlation with fused Richardson-Lucy epilogues -- the HBM-bound
// headline kernel.  Compile-time tap counts; one translation unit per PZ (-DLSR_SEP_PZ=n).
//
// The input volume carries a ZERO HALO (lsr_sep_padded_shape): zero padding of the correlation is
// real memory, so every load in this kernel is unconditional and in bounds.
//
// A 512-thread workgroup (8 waves) owns a 8*kRun (y) x 64*kCols (x) column and marches along z, one
// workgroup barrier per plane.  Iteration zi:
//   commit   : plane zi+2, fetched two iterations ago into registers, -> LDS A[zi&1] (a linear
//              16-B-per-thread copy of the (8*kRun+PY-1) x (64*kCols+8) window).
//   prefetch : plane zi+4 -> the registers just freed (global_load_dwordx4, whole row runs), and
//              the `aux` values of the NEXT iteration's output plane.  Two planes per workgroup
//              are always in flight; nothing waits on them inside this iteration.
//   x pass   : plane zi+1, A[(zi+1)&1] -> B[(zi+1)&1].  A thread owns (row, 4 consecutive x); its
//              4+PX-1 inputs come from 3..5 ds_read_b128.  Lanes are assigned to items by the
//              HARDWARE's b128 lane groups, so each group reads one contiguous 256-B row run:
//              conflict-free for any pitch.  B's pitch (16 mod 32) keeps the b128 writes
//              conflict-free.
//   y pass   : plane zi from B[zi&1]; thread = kCols columns x kRun rows (kRun+PY-1 ds_read_b32 each).
//   z        : PZ pending output planes per point in registers; one FMA per pending plane both
//              shifts the window and absorbs this plane.
//   epilogue : completed plane zi-PZ/2: ratio = y*rcp(c+eps) or x*c*rcp(H^T 1), strided store.
//
// kCols = 2, kRun = 4 (a 32 x 128 tile, 85 KB of LDS for 9x7x7, one workgroup per CU; kRun = 3
// from PZ = 11 up, where the z accumulators would otherwise spill): the in-plane window of a tile then spans 6
// cache lines per row for 4 lines of output instead of 8 for 4 with two 64-wide tiles.  Neighbour
// workgroups ask for their shared halo lines at the same moment and L2 does not merge those
// misses (measured: 1.41x reads vs algorithmic with 64-wide tiles), so fewer, wider tiles is the
// way to cut the re-fetch; occupancy (1 vs 2 workgroups per CU) measured no difference.
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
// Algorithmic HBM bytes: 12 per voxel per launch (in + aux + out).

#include "common.hpp"
#include "correlate_common.hpp"

#ifndef LSR_SEP_PZ
#error "compile with -DLSR_SEP_PZ=<odd tap count along z>"
#endif

namespace {

using lsr::SepArgs;

constexpr int kCols = lsr::kSepCols;     // 64-wide column groups per workgroup (2)
constexpr int kRun = lsr::sep_run(LSR_SEP_PZ);  // consecutive y per thread in the y / z passes
constexpr int kTY = 8 * kRun;           // 8 waves * kRun rows (32 or 24)
constexpr int kTX = lsr::kSepWideTileX; // 64 * kCols (128)
constexpr int kPts = kRun * kCols;      // output points per thread
constexpr int kWaves = 8;
constexpr int kThreads = 64 * kWaves;    // 512
constexpr int kBand = 8;                 // tile rows per band of the tile walk (see the kernel)

template <int PY, int PX>
struct Tile {
  static constexpr int AR = kTY + PY - 1;                        // input rows of a tile
  static constexpr int WIN = 4 + PX - 1;                         // inputs of an x-pass item
  static constexpr int NPIECE = (WIN + 3) / 4;                   // ds_read_b128 per item
  static constexpr int PA = kTX - 4 + 4 * NPIECE;                // staged cols == LDS pitch of A
  static constexpr int CH = PA / 4;                              // 16-B chunks per staged row
  static constexpr int NCH = AR * CH;                            // chunks per plane
  static constexpr int SL = (NCH + kThreads - 1) / kThreads;     // chunks per thread
  static constexpr int PB = kTX + 16;                            // LDS pitch of B (16 mod 32)
  static constexpr int ASZ = AR * PA;
  static constexpr int BSZ = AR * PB;
  static constexpr int RS = AR * kCols;                          // (row, 64-col segment) pairs
  static constexpr int XIT = (RS + 4 * kWaves - 1) / (4 * kWaves);  // x-pass items per thread
  static_assert(PA >= kTX + PX - 1, "staged window covers the halo");
  static_assert(SL >= 2 && SL <= 4, "the hand-written waits cover two to four staging loads");
  static_assert(2 * (ASZ + BSZ) * 4 <= 160 * 1024, "LDS per workgroup");
};

typedef float f32x4 __attribute__((ext_vector_type(4)));  // native 16-byte vector (one VGPR quad)

__device__ __forceinline__ float fast_rcp(float d) {
  float r = __builtin_amdgcn_rcpf(d);    // v_rcp_f32, 1 ulp
  return fmaf(fmaf(-d, r, 1.0f), r, r);  // + one Newton step
}

// ---- hand-managed global loads: scalar base + unsigned 32-bit byte offset per lane -------------
// The destination is an IN/OUT operand ("+v"), as in rl_fused_sep.hip: a register that is loaded again before its value
// was read -- the prologue's placeholder loads, the prefetches of the planes past the last one -- must stay the same
// physical register while the older load is in flight.  As a pure output ("=v") the older value is dead to the compiler
// and the register free between the two loads: round 4 found the <13, 9, 9> instance computing LDS offsets in such a
// register, and the in-flight load landing on top of them (results that differed from run to run).
__device__ __forceinline__ void gload_x4(f32x4& dst, const float* sbase, int voff_bytes) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(dst) : "v"(voff_bytes), "s"(sbase) : "memory");
}
__device__ __forceinline__ void gload_x1(float& dst, const float* sbase, int voff_bytes) {
  asm volatile("global_load_dword %0, %1, %2" : "+v"(dst) : "v"(voff_bytes), "s"(sbase) : "memory");
}
// Wait until at most N vector-memory operations are outstanding; the registers are passed
// through so that every later use depends on this statement.
template <int N, int SL>
__device__ __forceinline__ void wait_stage(f32x4 (&st)[SL]) {
  if constexpr (SL == 2) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(st[0]), "+v"(st[1]) : "n"(N) : "memory");
  } else if constexpr (SL == 3) {
    asm volatile("s_waitcnt vmcnt(%3)" : "+v"(st[0]), "+v"(st[1]), "+v"(st[2]) : "n"(N) : "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(st[0]), "+v"(st[1]), "+v"(st[2]), "+v"(st[3]) : "n"(N) : "memory");
  }
}
template <int N>
__device__ __forceinline__ void wait_aux(float (&a)[kPts], float& b) {
  if constexpr (kPts == 4) {
    asm volatile("s_waitcnt vmcnt(%5)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b) : "n"(N) : "memory");
  } else if constexpr (kPts == 6) {
    asm volatile("s_waitcnt vmcnt(%7)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(b)
                 : "n"(N)
                 : "memory");
  } else {
    static_assert(kPts == 8, "one or two columns per thread");
    asm volatile("s_waitcnt vmcnt(%9)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]),
                   "+v"(a[7]), "+v"(b)
                 : "n"(N)
                 : "memory");
  }
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

  // loads per iteration besides the staging loads: aux values (+ nz for the update)
#ifdef LSR_PROBE_NOAUX
  constexpr int NA = 0;
#else
  constexpr int NA = EPI == LSR_EPI_NONE ? 0 : (EPI == LSR_EPI_UPDATE ? kPts + 1 : kPts);
#endif
  constexpr int SL = T::SL;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar

  // XCD-aware tile order: workgroups b, b+8, ... share an XCD (round-robin dispatch), so give
  // each XCD a contiguous run of tiles ...
  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x;
    const int per = nblk / 8, rem = nblk % 8;
    const int xcd = bid % 8, idx = bid / 8;
    bid = xcd * per + (xcd < rem ? xcd : rem) + idx;  // XCD k owns per + (k < rem) tiles
  }
  // ... and walk the (y, x) tile grid in bands of 8 tile rows, column-major inside a band, so
  // that the workgroups an XCD keeps resident form a compact patch.
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
  int s_voff[SL];  // byte offset from in_tile (+ z * plane)
  int s_loff[SL];  // 16-byte chunk index inside an A buffer
#pragma unroll
  for (int k = 0; k < SL; ++k) {
    const int e = min(tid + k * kThreads, T::NCH - 1);
    const int r = e / T::CH, c = e - r * T::CH;
    s_voff[k] = (r * p.in_pitch + 4 * c) * 4;
    s_loff[k] = e;
  }
  f32x4 st0[SL] = {}, st1[SL] = {};  // plane q is staged in set q & 1 (initialised: in/out operands of the loads)

  // ---- x pass: lane -> (row-segment, x group) by hardware b128 lane group; a row-segment is one
  // 64-column run of one staged row, item `it` of this thread is row-segment xrs0 + 32*it
  int xg, xq;
  b128_lane_group(lane, xg, xq);
  const int xrs0 = 4 * wave + xg;

  // ---- y / z pass and epilogue: thread owns columns lane + 64*col, rows 4*wave .. 4*wave+3
  const int ycol = (wave * kRun) * T::PB + lane;
  const int gy_out0 = y0 + wave * kRun;  // scalar
  int a_voff[kPts], o_off[kPts];
  bool ok[kPts];
#pragma unroll
  for (int col = 0; col < kCols; ++col) {
    const int gx = x0 + lane + 64 * col;
    const int gxc = min(gx, X - 1);  // clamped: aux loads are always in bounds
#pragma unroll
    for (int m = 0; m < kRun; ++m) {
      const int gy = gy_out0 + m;
      ok[col * kRun + m] = gx < X && gy < Y;
      a_voff[col * kRun + m] = (min(gy, Y - 1) * p.aux_pitch + gxc) * 4;
      o_off[col * kRun + m] = min(gy, Y - 1) * p.out_pitch + gxc;
    }
  }
  // reciprocals of the in-plane factors of H^T 1 (UPDATE epilogue); the product order
  // (rz * rny) * rnx is shared with rl_fused_sep.hip, whose results are bit-identical
  float rny[kRun], rnx[kCols];
#pragma unroll
  for (int m = 0; m < kRun; ++m) rny[m] = 0.0f;
#pragma unroll
  for (int col = 0; col < kCols; ++col) rnx[col] = 0.0f;
  if constexpr (EPI == LSR_EPI_UPDATE) {
#pragma unroll
    for (int col = 0; col < kCols; ++col) rnx[col] = fast_rcp(p.nx[min(x0 + lane + 64 * col, X - 1)]);
#pragma unroll
    for (int m = 0; m < kRun; ++m) rny[m] = fast_rcp(p.ny[min(gy_out0 + m, Y - 1)]);
  }

  // pending output planes: acc[j][i] <-> z_out = zi - cz + j once plane zi is absorbed
  float acc[PZ][kPts];
#pragma unroll
  for (int j = 0; j < PZ; ++j)
#pragma unroll
    for (int i = 0; i < kPts; ++i) acc[j][i] = 0.0f;
  // aux (and nz) of output plane zi - cz live in set zi & 1, requested one iteration ahead
  float aux0[kPts], aux1[kPts];
#pragma unroll
  for (int i = 0; i < kPts; ++i) aux0[i] = aux1[i] = 0.0f;
  float nzv0 = 1.0f, nzv1 = 1.0f;

  const int zi_begin = max(zb - cz, 0);
  const int zi_end = ze + cz;  // exclusive; planes >= Z contribute zeros
  lsr::RlStats stats;   // UPDATE with p.stats: the launch's reduction scalars (correlate_common.hpp)

  auto fetch = [&](int zplane, f32x4 (&st)[SL]) __attribute__((always_inline)) {  // SL loads
    const float* src = in_tile + static_cast<int64_t>(min(max(zplane, 0), Z - 1)) * p.in_plane;
#pragma unroll
    for (int k = 0; k < SL; ++k) gload_x4(st[k], src, s_voff[k]);
  };
  auto fetch_aux = [&](int zout, float (&aux)[kPts], float& nzv) __attribute__((always_inline)) {  // NA loads
    if constexpr (EPI != LSR_EPI_NONE && NA != 0) {
      const int zc_ = min(max(zout, 0), Z - 1);
      const float* a = p.aux + static_cast<int64_t>(zc_) * p.aux_plane;
#pragma unroll
      for (int i = 0; i < kPts; ++i) gload_x1(aux[i], a, a_voff[i]);
      if constexpr (EPI == LSR_EPI_UPDATE) gload_x1(nzv, p.nz + zc_, 0);
    }
  };

  // One iteration; `par` = zi & 1 is a literal at both call sites.
  auto iteration = [&](const int zi, const int par, f32x4 (&st)[SL], float (&aux_use)[kPts],
                       float& nz_use, float (&aux_load)[kPts], float& nz_load) __attribute__((always_inline)) {
    f32x4* A_commit = bufA4 + par * (T::ASZ / 4);            // plane zi+2
    const f32x4* A_x = bufA4 + (par ^ 1) * (T::ASZ / 4);     // plane zi+1
    f32x4* B_x = bufB4 + (par ^ 1) * (T::BSZ / 4);           // plane zi+1
    const float* B_y = bufB + par * T::BSZ;                  // plane zi

    // commit plane zi+2.  Its loads were issued two iterations ago; issued since: aux (NA),
    // the other staging set (SL), aux (NA).
    wait_stage<SL + 2 * NA>(st);
#pragma unroll
    for (int k = 0; k < SL; ++k) A_commit[s_loff[k]] = st[k];
    __builtin_amdgcn_sched_barrier(0);  // the refill reuses these registers: keep it behind
    fetch(zi + 4, st);
    fetch_aux(zi + 1 - cz, aux_load, nz_load);

    // x pass of plane zi+1
#ifndef LSR_SEP_PROBE
#pragma unroll
    for (int it = 0; it < T::XIT; ++it) {
      const int rs = xrs0 + it * (4 * kWaves);
      if (it + 1 < T::XIT || rs < T::RS) {
        const int row = rs / kCols, seg = rs - row * kCols;
        const f32x4* src = A_x + row * (T::PA / 4) + seg * 16 + xq;
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
        B_x[row * (T::PB / 4) + seg * 16 + xq] = o;
      }
    }
#else
    (void)A_x; (void)B_x;
#endif

    if (zi >= zi_begin && zi < zi_end) {  // wave-uniform
      float pl[kPts];
#pragma unroll
      for (int i = 0; i < kPts; ++i) pl[i] = 0.0f;
#ifdef LSR_SEP_PROBE
      (void)B_y;
#pragma unroll
      for (int i = 0; i < kPts; ++i) pl[i] = reinterpret_cast<const float*>(A_x)[tid + 512 * i];
      if (false) {
#else
      if (zi < Z) {
#endif
#pragma unroll
        for (int col = 0; col < kCols; ++col) {
          const float* colp = B_y + ycol + 64 * col;
          float cv[kRun + PY - 1];
#pragma unroll
          for (int j = 0; j < kRun + PY - 1; ++j) cv[j] = colp[j * T::PB];
#pragma unroll
          for (int m = 0; m < kRun; ++m) {
            float s = wy[0] * cv[m];
#pragma unroll
            for (int b = 1; b < PY; ++b) s = fmaf(wy[b], cv[m + b], s);
            pl[col * kRun + m] = s;
          }
        }
      }
      // z: shift the pending planes and absorb this plane in the same FMA
#ifdef LSR_SEP_PROBE
#pragma unroll
      for (int i = 0; i < kPts; ++i) acc[0][i] = pl[i];
#else
#pragma unroll
      for (int j = 0; j < PZ - 1; ++j)
#pragma unroll
        for (int i = 0; i < kPts; ++i) acc[j][i] = fmaf(wz[PZ - 1 - j], pl[i], acc[j + 1][i]);
#pragma unroll
      for (int i = 0; i < kPts; ++i) acc[PZ - 1][i] = wz[0] * pl[i];
#endif

      const int z_out = zi - cz;
#ifdef LSR_PROBE_NOSTORE
      if (z_out >= zb && acc[0][0] == 12345.678f) {
#else
      if (z_out >= zb) {  // wave-uniform
#endif
        float* o = p.out + static_cast<int64_t>(z_out) * p.out_plane;  // scalar
        if constexpr (EPI != LSR_EPI_NONE && NA != 0) {
          // aux of this plane was requested in the previous iteration; issued since: this
          // iteration's staging loads (SL) and aux (NA)
          wait_aux<SL + NA>(aux_use, nz_use);
        }
        if constexpr (EPI == LSR_EPI_RATIO) {
#pragma unroll
          for (int i = 0; i < kPts; ++i)
            if (ok[i]) o[o_off[i]] = aux_use[i] * fast_rcp(acc[0][i] + p.eps);
        } else if constexpr (EPI == LSR_EPI_UPDATE) {
          const float rz = fast_rcp(nz_use);
#pragma unroll
          for (int i = 0; i < kPts; ++i)
            if (ok[i]) {
              const float xu = aux_use[i] * acc[0][i];
              const float v = xu * ((rz * rny[i % kRun]) * rnx[i / kRun]);
              o[o_off[i]] = v;
              if (p.stats) stats.add(aux_use[i], xu, v);   // (kernel-uniform)
            }
        } else {
#pragma unroll
          for (int i = 0; i < kPts; ++i)
            if (ok[i]) o[o_off[i]] = acc[0][i];
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
  if constexpr (EPI == LSR_EPI_UPDATE) {
    lsr::keep_until_here(st0);   // (in-flight prefetches: correlate_common.hpp, keep_until_here)
    lsr::keep_until_here(st1);
    lsr::keep_until_here(aux0);
    lsr::keep_until_here(aux1);
    lsr::keep_until_here(nzv0);
    lsr::keep_until_here(nzv1);
    stats.pin();
    if (p.stats) lsr::rl_stats_flush<kWaves>(stats, reinterpret_cast<float*>(bufB4), p.stats);
  }
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
