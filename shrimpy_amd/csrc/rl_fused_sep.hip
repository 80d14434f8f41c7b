// One Richardson-Lucy iteration in ONE launch for separable PSFs:
//
//     x_new = x * H^T( y / (H x + eps) ) / (H^T 1)
//
// The two-launch form (correlate_sep.hip: RATIO, then UPDATE) moves 24 bytes per voxel and
// iteration through HBM (x, y -> ratio;  ratio, x -> x_new); this kernel moves 12 algorithmic bytes
// (x, y -> x_new) -- the ratio volume never leaves the CU.  It is the same z-marching 2.5-D
// stencil, run twice back to back inside the workgroup:
//
//   stage 1  c = H x on the tile grown by the in-plane PSF radius C   ((TY+2C) x (128+2C) points)
//            ratio = y * rcp(c + eps), zero outside the volume          -> LDS only
//   stage 2  u = H^T ratio on the tile (TY x 128), x_new = x * u * rcp(H^T 1) -> HBM
//
// All three volumes (x in, y, x out) are padded volumes (lsr_sep_padded_shape) with a ZERO halo of
// 2C rows / 32 columns, so every load is unconditional and in bounds, and "zero outside the
// volume" is real memory for stage 1.  Stage 2's zero padding is the explicit mask on `ratio`.
// Arithmetic per voxel (operation order, FMA placement, rcp refinement) is that of the two-launch
// kernels, so both paths return bit-identical volumes (tests/test_gpu_parity.py checks equality).
//
// A 512-thread workgroup (8 waves, one workgroup per CU) owns a TY x 128 column of output and
// marches along z.  Iteration p (two workgroup barriers):
//
//   phase A   glds: x plane p+2 -> LDS ring slot (p+2)%3   (global_load_lds_dwordx4, no registers)
//             x1 pass: x plane p,        A[p%3] -> B1      (4 outputs per item, ds_read_b128)
//             x2 pass: ratio plane p-1-CZ, R    -> B2
//   phase B   y2/z2 pass: B2 -> stage-2 accumulators; output plane o = p-1-2CZ is complete:
//             epilogue and store; then request x(o+1), nz(o+1) for the next epilogue
//             y1/z1 pass: B1 -> stage-1 accumulators; ratio plane q = p-CZ is complete -> R;
//             then request y(q+1) for the next ratio
//             wait for this wave's glds of plane p+1, barrier
//
// Memory pipeline by hand, as in correlate_sep.hip: global loads are inline asm, every consumer is
// preceded by a hand-counted `s_waitcnt vmcnt(N)`, N = the number of LOADS this wave issued after
// the wanted one (MI355X_MICROARCH.md: vector-memory operations retire in issue order, so younger
// stores can only delay the wait).  Every iteration issues the same loads whether or not its
// planes are inside the volume (addresses are clamped), which keeps N a compile-time constant.
//
// Algorithmic HBM bytes: 12 per voxel and iteration.

#include "common.hpp"
#include "correlate_common.hpp"

#ifndef LSR_FUSED_PZ
#error "compile with -DLSR_FUSED_PZ=<odd tap count along z>"
#endif

namespace {

using lsr::FusedArgs;

constexpr int kTX = lsr::kSepWideTileX;  // 128
constexpr int kWaves = 8;
constexpr int kThreads = 64 * kWaves;
constexpr int kBand = 8;
constexpr int kRing = 3;                 // LDS ring slots of staged x planes

constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }

template <int PZ, int PYX, int RUN>
struct Geo {
  static constexpr int C = PYX / 2, CZ = PZ / 2;
  static constexpr int TY = 8 * RUN;
  static constexpr int WL = lsr::fused_window_halo(PYX);  // staged columns left/right of the tile
  static constexpr int AR = TY + 4 * C;                   // staged rows
  static constexpr int PA = kTX + 2 * WL;                 // staged columns = pitch of A
  static constexpr int CH = PA / 4;                       // 16-byte chunks per staged row
  static constexpr int NCH = AR * CH;
  static constexpr int SL = cdiv(NCH, kThreads);          // glds per thread and plane
  static constexpr int ASZ = cdiv(NCH, 64) * 64 * 4;      // floats per ring slot (whole waves of chunks)
  static constexpr int NP = cdiv(4 + 2 * C, 4);           // ds_read_b128 per x-pass item
  static constexpr int G1 = (kTX + WL) / 4;               // x1 items per row
  static constexpr int PB1 = 4 * G1;                      // pitch of B1 (linear in the item index)
  static constexpr int NIT1 = AR * G1;
  static constexpr int XIT1 = cdiv(NIT1, kThreads);
  static constexpr int R1 = TY + 2 * C;                   // stage-1 (ratio) rows
  static constexpr int RUN1 = cdiv(R1, 8);                // ratio rows per thread
  static constexpr int E = PB1 - kTX;                     // ratio columns beyond the two 64-lane groups
  static constexpr int NE = R1 * E;                       // "edge" points of the ratio plane
  static constexpr int EP = cdiv(NE, kThreads);           // edge points per thread
  static constexpr int SH = WL - 2 * C;                   // B1 column -> ratio column shift (0 or 2)
  static constexpr int PR = 4 * (kTX / 4 - 1 + NP);       // pitch of R
  static constexpr int RSZ = 4 + 8 * RUN1 * PR;           // one leading chunk absorbs columns < 0
  static constexpr int NIT2 = R1 * (kTX / 4);
  static constexpr int XIT2 = cdiv(NIT2, kThreads);
  static constexpr int B1SZ = AR * PB1;
  static constexpr int B2SZ = R1 * kTX;
  // LDS map (floats): ring | B1 | R | B2.  B1's wasted rows read into R, never past the end.
  static constexpr int OFF_B1 = kRing * ASZ;
  static constexpr int OFF_R = OFF_B1 + B1SZ;
  static constexpr int OFF_B2 = OFF_R + RSZ;
  static constexpr int OFF_DUMP = OFF_B2 + B2SZ;          // 1 KB: where glds of waves past the window land
  static constexpr int OFF_RNY = OFF_DUMP + 256;          // 1 / ny of the tile rows
  static constexpr int TOTAL = OFF_RNY + 32;
  static constexpr int NY = 2 * RUN1 + EP;                // y loads per thread and iteration
#ifdef LSR_FUSED_PROBE_NOXC
  static constexpr int NXC = 0;  // diagnostic build: the x re-read is dropped (results wrong)
#else
  static constexpr int NXC = 2 * RUN + 1;                 // x (centre) + nz loads
#endif
  static_assert(TOTAL * 4 <= 160 * 1024, "LDS per workgroup");
  static_assert(PR == PB1, "B1 and R share a pitch");
  static_assert(2 * C <= WL && WL <= lsr::kSepOriginCol, "halo columns");
  static_assert(SL + 2 * (NY + NXC) <= 63, "vmcnt is a 6-bit counter");
  static_assert(OFF_B1 % 4 == 0 && OFF_R % 4 == 0 && OFF_B2 % 4 == 0, "16-byte aligned buffers");
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
// Two floats in an even-aligned register pair: the operand of the packed fp32 instructions
// (v_pk_fma_f32 / v_pk_mul_f32: two lanes' worth of work per issue slot).  Each component is an
// ordinary IEEE operation, so packing never changes a result.  The pairs are chosen by hand -- the
// two column groups of a thread, which ds_read2 delivers in adjacent registers -- because hipcc's
// own pairing (rows of one column) costs two v_mov per packed instruction.
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float fast_rcp(float d) {
  float r = __builtin_amdgcn_rcpf(d);
  return fmaf(fmaf(-d, r, 1.0f), r, r);
}
__device__ __forceinline__ f32x2 splat(float a) { return f32x2{a, a}; }
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 fast_rcp2(f32x2 d) {  // fast_rcp on both components
  const f32x2 r = f32x2{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  return pk_fma(pk_fma(-d, r, splat(1.0f)), r, r);
}
// One x-pass item: 4 consecutive outputs from 4 + PX - 1 inputs held in NP 16-byte pieces.
template <int PX, int NP>
__device__ __forceinline__ f32x4 xpass_item(const f32x4* src, const float (&wx)[PX]) {
  float w[4 * NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const f32x4 v = src[i];
    w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
  }
  f32x2 o01 = splat(wx[0]) * f32x2{w[0], w[1]};
  f32x2 o23 = splat(wx[0]) * f32x2{w[2], w[3]};
#pragma unroll
  for (int c = 1; c < PX; ++c) {
    o01 = pk_fma(splat(wx[c]), f32x2{w[c], w[c + 1]}, o01);
    o23 = pk_fma(splat(wx[c]), f32x2{w[c + 2], w[c + 3]}, o23);
  }
  return f32x4{o01.x, o01.y, o23.x, o23.y};
}

// ---- hand-managed memory operations: scalar base + unsigned 32-bit byte offset per lane -------
// The destination is an in/out operand: a register that is loaded again before its value was used
// (the prologue does that) must stay the same physical register while the older load is in flight.
template <int IMM>
__device__ __forceinline__ void gload(float& dst, const float* sbase, int voff) {
  asm volatile("global_load_dword %0, %1, %2 offset:%3" : "+v"(dst) : "v"(voff), "s"(sbase), "n"(IMM) : "memory");
}
// Stores are non-temporal: x_new is not read again before the next launch, and keeping it out of
// the way leaves more of L2 / MALL to the x planes that ARE read again nine planes later
// (measured 2.56 -> 2.50 ms per launch; `nt` on the y loads instead made it slower, 2.66 ms).
#ifndef LSR_FUSED_STORE_POLICY
#define LSR_FUSED_STORE_POLICY "nt"   // probes (round 2): "sc1", "nt sc1", "sc0 sc1" -- see DESIGN.md section 4.3
#endif
#ifndef LSR_FUSED_GLDS_POLICY
#define LSR_FUSED_GLDS_POLICY ""      // probe: "nt" on the LDS-DMA of the x window
#endif
template <int IMM>
__device__ __forceinline__ void gstore(float* sbase, int voff, float v) {
  asm volatile("global_store_dword %0, %1, %2 offset:%3 " LSR_FUSED_STORE_POLICY
               :
               : "v"(voff), "v"(v), "s"(sbase), "n"(IMM)
               : "memory");
}
// LDS-DMA: 16 bytes per lane, LDS address = m0 + 16 * lane.  One wait state between the write of
// m0 and the load (s_nop).
__device__ __forceinline__ void glds_x4(const float* sbase, int voff, unsigned lds_byte_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 " LSR_FUSED_GLDS_POLICY
               :
               : "v"(voff), "s"(sbase), "s"(lds_byte_addr)
               : "memory");  // (m0 is a reserved register: hipcc sets it right at each of its own uses)
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}
// After a wait: pass the loaded registers through an (empty) volatile asm, so that every later use
// depends on a statement the compiler keeps behind the wait.
template <int K>
__device__ __forceinline__ void tie(float (&a)[K]) {
#pragma unroll
  for (int i = 0; i < K; ++i) asm volatile("" : "+v"(a[i]));
}
__device__ __forceinline__ void tie(float& a) { asm volatile("" : "+v"(a)); }
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

// STATS: the iteration's reduction scalars (correlate_common.hpp: RlStats) are summed in the epilogue and added to
// p.stats[0..2] -- a separate instantiation, so that the kernel without them is the one it always was.
template <int PZ, int PYX, int RUN, bool STATS>
__global__ __launch_bounds__(kThreads) void rl_fused_sep_kernel(FusedArgs p) {
  using T = Geo<PZ, PYX, RUN>;
  constexpr int C = T::C, CZ = T::CZ, TY = T::TY, RUN1 = T::RUN1, EP = T::EP;
  constexpr int NP = T::NP, NY = T::NY, NXC = T::NXC, SL = T::SL;
  __shared__ f32x4 smem4[T::TOTAL / 4];
  float* const smem = reinterpret_cast<float*>(smem4);
  f32x4* const B1_4 = smem4 + T::OFF_B1 / 4;
  const float* const B1 = smem + T::OFF_B1;
  float* const Rw = smem + T::OFF_R + 4;            // ratio column 0 of row 0
  const f32x4* const R_4 = smem4 + T::OFF_R / 4 + 1;
  f32x4* const B2_4 = smem4 + T::OFF_B2 / 4;
  const float* const B2 = smem + T::OFF_B2;
  const unsigned lds_base =
      static_cast<unsigned>(reinterpret_cast<size_t>((__attribute__((address_space(3))) char*)smem4));

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // Work items, longest first: tiles [0, n_full) are whole z columns; the remaining tiles are cut
  // into `pieces` z chunks each, so that the last dispatch round is full too (one workgroup per
  // CU: with 4.5 columns per CU either half the chip idles for a round, or every column pays the
  // 2 * (PZ - 1) halo planes twice).  Workgroups that run side by side stay on neighbouring tiles
  // AND on the same planes: their halo reads then hit L2 / MALL -- an even split of the tile-major
  // plane sequence over the CUs (tried) loses that and was 17 % slower.
  // XCD-aware order inside each class: workgroups b, b+8, ... share an XCD (round-robin dispatch),
  // so every XCD gets a contiguous run of tiles.
  auto xcd_contiguous = [](int b, int n) {
#ifdef LSR_FUSED_PLAIN_ORDER   // probe build: blocks in launch order, one per XCD in turn
    return b;
#endif
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
  // tiles in bands of 8 tile rows, column-major inside a band (compact patches per XCD)
  const int band = lin / (p.tiles_x * kBand);
  const int lb = lin - band * (p.tiles_x * kBand);
  const int band_h = min(kBand, p.tiles_y - band * kBand);
#if defined(LSR_FUSED_PLAIN_ORDER) && LSR_FUSED_PLAIN_ORDER == 2   // probe build: row-major tiles
  const int tx = lin % p.tiles_x;
  const int ty = lin / p.tiles_x;
  (void)lb; (void)band_h;
#else
  const int tx = lb / band_h;
  const int ty = band * kBand + (lb - tx * band_h);
#endif
  const int x0 = tx * kTX, y0 = ty * TY;
#ifdef LSR_FUSED_LAG
  if ((tx + ty) & 1) {
#pragma unroll
    for (int i = 0; i < LSR_FUSED_LAG; ++i) __builtin_amdgcn_s_sleep(127);
  }
#endif

  // taps: a device block prepared by lsr_rl_sep_fused_prepare_taps -- six rows of 16 floats
  // (stage 1 = flipped PSF: x, y, z; stage 2 = PSF: x, y, z), centred in the compiled extents, zero
  // elsewhere.  They live across the lanes of two VGPRs; each pass pulls its taps into SGPRs with
  // v_readlane right before it runs.  (All 46 in SGPRs for the whole loop do not fit next to the
  // addressing state -- hipcc then spills them to VGPR lanes itself, and more besides; fetching
  // them per pass from the scalar cache instead costs an exposed s_load round trip per pass.)
  float tv0 = p.taps[lane], tv1 = p.taps[64 + (lane & 31)];
  auto load_taps = [&](int row, auto& w) {
#pragma unroll
    for (int i = 0; i < static_cast<int>(sizeof(w) / sizeof(float)); ++i) {
      const int f = row * 16 + i;
      w[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(f < 64 ? tv0 : tv1), f & 63));
    }
  };

  // ---- staging (glds): chunk e = tid + 512 k of the (AR x PA) window whose first element is
  // (y0 - 2C, x0 - WL); the LDS image of a ring slot is linear in e.
  const float* const x_tile = p.x + (static_cast<int64_t>(y0 - 2 * C) * p.pitch + (x0 - T::WL));
  int s_voff[SL];
#pragma unroll
  for (int k = 0; k < SL; ++k) {
    const int e = min(tid + k * kThreads, T::NCH - 1);
    const int r = e / T::CH, c = e - r * T::CH;
    s_voff[k] = (r * p.pitch + 4 * c) * 4;
  }
  // ---- x1 items: i = tid + 512 k -> (row, g); A chunk = row * CH + g, B1 chunk = i
  int a1_chunk[T::XIT1];
#pragma unroll
  for (int k = 0; k < T::XIT1; ++k) {
    const int i = min(tid + k * kThreads, T::NIT1 - 1);
    const int row = i / T::G1;
    a1_chunk[k] = i + row * (T::CH - T::G1);
  }
  // ---- stage-1 points.  Main: ratio columns b = lane + 64 cg (B1 column index), ratio rows
  // wave * RUN1 + m.  Ratio row r <-> tile row r - C; B1 column b <-> tile column b + C - WL.
  const int r1_row0 = wave * RUN1;                            // scalar
  const int y1_col = r1_row0 * T::PB1 + lane;                 // B1 float index of (row0, lane)
  const int r_col = r1_row0 * T::PR + lane - T::SH;           // R float index of the same point
  // a tile whose grown (ratio) region lies inside the volume needs no in-plane masks (most tiles)
  const bool interior = x0 - C >= 0 && x0 + kTX + C <= X && y0 - C >= 0 && y0 + TY + C <= Y;
  // edge points: t = tid + 512 e -> (row t / E, column 128 + t % E)
  int e_b1[EP], e_voff[EP];  // (the R index of an edge point is e_b1 - SH: B1 and R share a pitch)
  int e_rc[EP];              // (row << 16) | column of the edge point in the ratio region
#pragma unroll
  for (int e = 0; e < EP; ++e) {
    const int t = min(tid + e * kThreads, T::NE - 1);
    const int er = t / T::E, ec = t - er * T::E;
    e_b1[e] = er * T::PB1 + kTX + ec;
    e_rc[e] = (er << 16) | (kTX + ec);
    e_voff[e] = (er * p.y_pitch + kTX + ec) * 4;   // from the y window's first element
  }
  // ---- stage-2 points: tile columns lane + 64 cg, tile rows wave * RUN + m
  const int y2_col = (wave * RUN) * kTX + lane;
  // per-row byte offsets of the aux loads and the stores (row term included, so that one scalar
  // base per stream and plane is all the address arithmetic an iteration does)
  int xc_voff[RUN], o_voff[RUN], y_voff[RUN1];
#pragma unroll
  for (int m = 0; m < RUN; ++m) {
    xc_voff[m] = ((wave * RUN + m) * p.pitch + lane) * 4;
    o_voff[m] = ((wave * RUN + m) * p.out_pitch + lane) * 4;
  }
#pragma unroll
  for (int m = 0; m < RUN1; ++m) y_voff[m] = (min(r1_row0 + m, T::R1 - 1) * p.y_pitch + lane) * 4;
  // reciprocal in-plane norm factors: 1/nx per column in registers, 1/ny per tile row in LDS
  float rnx[2];
#pragma unroll
  for (int cg = 0; cg < 2; ++cg) rnx[cg] = fast_rcp(p.nx[min(x0 + lane + 64 * cg, X - 1)]);
  float* const rny_lds = smem + T::OFF_RNY;
  if (tid < TY) rny_lds[tid] = fast_rcp(p.ny[min(y0 + tid, Y - 1)]);

  // pending planes; a pair = (column group 0, column group 1) of one row
  f32x2 acc1[PZ][RUN1], acc2[PZ][RUN];
  float acc1e[PZ][EP];
#pragma unroll
  for (int j = 0; j < PZ; ++j) {
#pragma unroll
    for (int i = 0; i < RUN1; ++i) acc1[j][i] = splat(0.0f);
#pragma unroll
    for (int i = 0; i < EP; ++i) acc1e[j][i] = 0.0f;
#pragma unroll
    for (int i = 0; i < RUN; ++i) acc2[j][i] = splat(0.0f);
  }
  float yv[2 * RUN1], ye[EP], xc[2 * RUN], nzv = 1.0f;  // [m] = group 0, [RUN(1) + m] = group 1
#pragma unroll
  for (int i = 0; i < 2 * RUN1; ++i) yv[i] = 0.0f;
#pragma unroll
  for (int i = 0; i < EP; ++i) ye[i] = 0.0f;
#pragma unroll
  for (int i = 0; i < 2 * RUN; ++i) xc[i] = 0.0f;
  const f32x2 rnx2 = f32x2{rnx[0], rnx[1]};
  // the iteration's scalars (STATS): packed running sums -- a pair = the thread's two column groups, like everything else
  // here -- so that a row pair costs five instructions (two v_pk_add, one v_pk_add with a negated operand, two v_add |.|)
  f32x2 st_flux = splat(0.0f), st_total = splat(0.0f);
  float st_change = 0.0f;
  auto st_add = [&](f32x2 x_old, f32x2 xu, f32x2 v) {
    st_flux += xu;
    st_total += v;
    const f32x2 d = v - x_old;
    st_change += __builtin_fabsf(d.x);
    st_change += __builtin_fabsf(d.y);
  };
  __builtin_amdgcn_sched_barrier(0);  // setup loads (taps, norms) are consumed above this line

  auto clampz = [&](int z) { return min(max(z, 0), Z - 1); };
  auto issue_glds = [&](int plane, int slot) {  // SL loads
    const float* src = x_tile + static_cast<int64_t>(clampz(plane)) * p.plane;
    const unsigned dst = lds_base + (slot * T::ASZ + wave * 64 * 4) * 4;
#pragma unroll
    for (int k = 0; k < SL; ++k) {
      // every wave issues SL loads (constant vmcnt bookkeeping); a wave whose chunks lie past the
      // window sends them to the dump area
      const bool live = wave * 64 + k * kThreads < T::NCH;  // wave-uniform
      glds_x4(src, s_voff[k], live ? dst + k * kThreads * 16 : lds_base + T::OFF_DUMP * 4);
    }
  };
  const float* const xc_tile = p.x + (static_cast<int64_t>(y0) * p.pitch + x0);
  const float* const y_tile = p.y + (static_cast<int64_t>(y0 - C) * p.y_pitch + (x0 + C - T::WL));
  float* const o_tile = p.out + (static_cast<int64_t>(y0) * p.out_pitch + x0);
  auto issue_xc = [&](int o) {  // NXC loads: x at the output points of plane o, and nz[o]
#ifdef LSR_FUSED_PROBE_NOXC
    return;
#endif
    const int oc = clampz(o);
    const float* base = xc_tile + static_cast<int64_t>(oc) * p.plane;
#pragma unroll
    for (int m = 0; m < RUN; ++m) {
      gload<0>(xc[m], base, xc_voff[m]);
      gload<256>(xc[RUN + m], base, xc_voff[m]);
    }
    gload<0>(nzv, p.nz + oc, 0);
  };
  auto issue_y = [&](int q) {  // NY loads: y at the ratio points of plane q
    const float* base = y_tile + static_cast<int64_t>(clampz(q)) * p.y_plane;
#pragma unroll
    for (int m = 0; m < RUN1; ++m) {
      gload<0>(yv[m], base, y_voff[m]);
      gload<256>(yv[RUN1 + m], base, y_voff[m]);
    }
#pragma unroll
    for (int e = 0; e < EP; ++e) gload<0>(ye[e], base, e_voff[e]);
  };

  // planes: x plane p feeds ratio planes p-CZ .. p+CZ; ratio plane q feeds outputs q-CZ .. q+CZ
  const int q_lo = max(zb - CZ, 0), q_hi = min(ze - 1 + CZ, Z - 1);  // ratio planes that matter
  const int p_lo = max(q_lo - CZ, 0);
  const int p_hi = ze + 2 * CZ;  // inclusive: the iteration that completes output plane ze - 1

  // prologue: the same load sequence two iterations would issue
  int slot = 0;  // ring slot of plane p
  issue_glds(p_lo, 0);
  issue_xc(p_lo - 1 - 2 * CZ);
  issue_y(p_lo - CZ);
  issue_glds(p_lo + 1, 1);
  issue_xc(p_lo - 1 - 2 * CZ);
  issue_y(p_lo - CZ);
  wait_vm<SL + 2 * (NXC + NY)>();  // this wave's part of plane p_lo has landed
  lds_barrier();

#ifdef LSR_FUSED_PROBE_TIME
  if (p.probe && tid == 0) p.probe[4 * blockIdx.x] = wall_clock64();
#endif
  for (int pz = p_lo; pz <= p_hi; ++pz) {
    asm volatile("" : "+v"(tv0), "+v"(tv1));  // loop-variant for the optimiser: no hoisting of the taps
#ifdef LSR_FUSED_PROBE_TIME
    if (p.probe && tid == 0 && (pz == p_lo + 40 || pz == p_lo + 120))
      p.probe[4 * blockIdx.x + (pz == p_lo + 40 ? 1 : 2)] = wall_clock64();
#endif
    const int qr = pz - 1 - CZ;      // ratio plane in R (written by the previous iteration)
    const int o = qr - CZ;           // output plane completed by this iteration
    const int q = pz - CZ;           // ratio plane completed by this iteration
    const bool x_live = pz < Z && pz <= q_hi + CZ;        // x plane pz exists and is needed
    const bool r_live = qr >= q_lo && qr <= q_hi;         // ratio plane qr is non-zero
    const int slot2 = slot >= 1 ? slot - 1 : 2;           // (slot + 2) % 3

    // ---------------- phase A ----------------
    issue_glds(pz + 2, slot2);
#ifdef LSR_FUSED_PROBE_NOCOMPUTE
    if (false) {
#else
    if (x_live) {
#endif
      float w1x[PYX];
      load_taps(0, w1x);
      const f32x4* A_4 = smem4 + slot * (T::ASZ / 4);
#pragma unroll
      for (int k = 0; k < T::XIT1; ++k)
        if (k + 1 < T::XIT1 || tid + k * kThreads < T::NIT1)
          B1_4[tid + k * kThreads] = xpass_item<PYX, NP>(A_4 + a1_chunk[k], w1x);
    } else {  // a plane outside the volume: zeros (a handful of iterations per workgroup)
#pragma unroll
      for (int k = 0; k < T::XIT1; ++k)
        if (k + 1 < T::XIT1 || tid + k * kThreads < T::NIT1) B1_4[tid + k * kThreads] = f32x4{0, 0, 0, 0};
    }
#ifdef LSR_FUSED_PROBE_NOCOMPUTE
    if (false) {
#else
    if (r_live) {
#endif
      float w2x[PYX];
      load_taps(3, w2x);
#pragma unroll
      for (int k = 0; k < T::XIT2; ++k) {
        const int j = tid + k * kThreads;
        if (k + 1 < T::XIT2 || j < T::NIT2)
          B2_4[j] = xpass_item<PYX, NP>(R_4 + (j + (j >> 5) * (T::PR / 4 - kTX / 4)), w2x);
      }
    } else {
#pragma unroll
      for (int k = 0; k < T::XIT2; ++k)
        if (k + 1 < T::XIT2 || tid + k * kThreads < T::NIT2) B2_4[tid + k * kThreads] = f32x4{0, 0, 0, 0};
    }
    lds_barrier();

    // ---------------- phase B ----------------
    // stage 2: absorb ratio plane qr, finish output plane o
    {
      float w2y[PYX], w2z[PZ];
      load_taps(4, w2y);
      load_taps(5, w2z);
      f32x2 pl[RUN];
#ifdef LSR_FUSED_PROBE_NOCOMPUTE
#pragma unroll
      for (int m = 0; m < RUN; ++m) pl[m] = splat(B2[y2_col + m]);
      if (false)
#endif
      {
        const float* colp = B2 + y2_col;
        f32x2 cv[RUN + 2 * C];
#pragma unroll
        for (int j = 0; j < RUN + 2 * C; ++j) cv[j] = f32x2{colp[j * kTX], colp[j * kTX + 64]};
#pragma unroll
        for (int m = 0; m < RUN; ++m) {
          f32x2 s2 = splat(w2y[0]) * cv[m];
#pragma unroll
          for (int b = 1; b < PYX; ++b) s2 = pk_fma(splat(w2y[b]), cv[m + b], s2);
          pl[m] = s2;
        }
      }
#pragma unroll
      for (int j = 0; j < PZ - 1; ++j)
#pragma unroll
        for (int i = 0; i < RUN; ++i) acc2[j][i] = pk_fma(splat(w2z[PZ - 1 - j]), pl[i], acc2[j + 1][i]);
#pragma unroll
      for (int i = 0; i < RUN; ++i) acc2[PZ - 1][i] = splat(w2z[0]) * pl[i];

      // x(o), nz(o): requested by the previous iteration; issued since: its y loads, this
      // iteration's glds
      wait_vm<NY + SL>();
      tie(xc);
      tie(nzv);
#ifdef LSR_FUSED_PROBE_NOSTORE
      if (o >= zb && o < ze && nzv == 12345.678f) {
#else
      if (o >= zb && o < ze) {
#endif
        float* obase = o_tile + static_cast<int64_t>(o) * p.out_plane;
        const float rz = fast_rcp(nzv);
        if (!p.mask_out) {
          // a padded destination: rows and columns past the volume exist, and what lands there is
          // x * u / n with x = 0 from the zero halo
#pragma unroll
          for (int m = 0; m < RUN; ++m) {
            const f32x2 xo = f32x2{xc[m], xc[RUN + m]};
            const f32x2 xu = xo * acc2[0][m];
            const f32x2 v = xu * (splat(rz * rny_lds[wave * RUN + m]) * rnx2);
            gstore<0>(obase, o_voff[m], v.x);
            gstore<256>(obase, o_voff[m], v.y);
            if constexpr (STATS) st_add(xo, xu, v);   // (points past the volume: x = 0 from the halo, all three terms 0)
          }
        } else {  // the dense result of the last iteration: masked to the volume
          const bool ok0 = x0 + lane < X, ok1 = x0 + lane + 64 < X;
#pragma unroll
          for (int m = 0; m < RUN; ++m) {
            if (y0 + wave * RUN + m < Y) {  // wave-uniform
              const f32x2 xo = f32x2{xc[m], xc[RUN + m]};
              const f32x2 xu = xo * acc2[0][m];
              const f32x2 v = xu * (splat(rz * rny_lds[wave * RUN + m]) * rnx2);
              if (ok0) gstore<0>(obase, o_voff[m], v.x);
              if (ok1) gstore<256>(obase, o_voff[m], v.y);
              if constexpr (STATS) {   // columns past the volume hold x = 0 (zero halo): they add nothing
                st_add(xo, xu, v);
              }
            }
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);  // the refill reuses xc / nzv
      issue_xc(o + 1);
    }
    // stage 1: absorb x plane pz, finish ratio plane q
    {
      float w1y[PYX], w1z[PZ];
      load_taps(1, w1y);
      load_taps(2, w1z);
      f32x2 pl[RUN1];
      float ple[EP];
#ifdef LSR_FUSED_PROBE_NOCOMPUTE
#pragma unroll
      for (int m = 0; m < RUN1; ++m) pl[m] = splat(B1[y1_col + m]);
      if (false)
#endif
      {
        const float* colp = B1 + y1_col;
        f32x2 cv[RUN1 + 2 * C];
#pragma unroll
        for (int j = 0; j < RUN1 + 2 * C; ++j) cv[j] = f32x2{colp[j * T::PB1], colp[j * T::PB1 + 64]};
#pragma unroll
        for (int m = 0; m < RUN1; ++m) {
          f32x2 s2 = splat(w1y[0]) * cv[m];
#pragma unroll
          for (int b = 1; b < PYX; ++b) s2 = pk_fma(splat(w1y[b]), cv[m + b], s2);
          pl[m] = s2;
        }
      }
#pragma unroll
      for (int e = 0; e < EP; ++e) {
        const float* colp = B1 + e_b1[e];
#ifdef LSR_FUSED_PROBE_NOCOMPUTE
        ple[e] = colp[0];
        continue;
#endif
        float s1 = w1y[0] * colp[0];
#pragma unroll
        for (int b = 1; b < PYX; ++b) s1 = fmaf(w1y[b], colp[b * T::PB1], s1);
        ple[e] = s1;
      }
#pragma unroll
      for (int j = 0; j < PZ - 1; ++j) {
#pragma unroll
        for (int i = 0; i < RUN1; ++i) acc1[j][i] = pk_fma(splat(w1z[PZ - 1 - j]), pl[i], acc1[j + 1][i]);
#pragma unroll
        for (int e = 0; e < EP; ++e) acc1e[j][e] = fmaf(w1z[PZ - 1 - j], ple[e], acc1e[j + 1][e]);
      }
#pragma unroll
      for (int i = 0; i < RUN1; ++i) acc1[PZ - 1][i] = splat(w1z[0]) * pl[i];
#pragma unroll
      for (int e = 0; e < EP; ++e) acc1e[PZ - 1][e] = w1z[0] * ple[e];

      // y(q): requested by the previous iteration; issued since: this iteration's glds and x loads
      wait_vm<SL + NXC>();
      tie(yv);
      tie(ye);
      const bool q_in = q >= q_lo && q <= q_hi;  // wave-uniform; planes outside are zero
      if (q_in && interior) {
        // (rows >= R1 of the last wave and the SH leading columns are padding of R: never read)
#pragma unroll
        for (int m = 0; m < RUN1; ++m) {
          const f32x2 r = f32x2{yv[m], yv[RUN1 + m]} * fast_rcp2(acc1[0][m] + splat(p.eps));
          Rw[r_col + m * T::PR] = r.x;
          Rw[r_col + m * T::PR + 64] = r.y;
        }
#pragma unroll
        for (int e = 0; e < EP; ++e)
          if (e + 1 < EP || tid + e * kThreads < T::NE)
            Rw[e_b1[e] - T::SH] = ye[e] * fast_rcp(acc1e[0][e] + p.eps);
      } else {
        // border tiles and planes outside the volume: ratio is zero wherever its point is outside
        const int gx0 = x0 + lane + C - T::WL;
        const bool in0 = q_in && gx0 >= 0 && gx0 < X, in1 = q_in && gx0 + 64 >= 0 && gx0 + 64 < X;
#pragma unroll
        for (int m = 0; m < RUN1; ++m) {
          const int gy = y0 + r1_row0 + m - C;
          const bool row_in = gy >= 0 && gy < Y && r1_row0 + m < T::R1;  // wave-uniform
          const f32x2 r = f32x2{yv[m], yv[RUN1 + m]} * fast_rcp2(acc1[0][m] + splat(p.eps));
          Rw[r_col + m * T::PR] = (row_in && in0) ? r.x : 0.0f;
          Rw[r_col + m * T::PR + 64] = (row_in && in1) ? r.y : 0.0f;
        }
#pragma unroll
        for (int e = 0; e < EP; ++e) {
          if (e + 1 < EP || tid + e * kThreads < T::NE) {
            const int gy = y0 + (e_rc[e] >> 16) - C, gx = x0 + (e_rc[e] & 0xffff) + C - T::WL;
            const float r = ye[e] * fast_rcp(acc1e[0][e] + p.eps);
            Rw[e_b1[e] - T::SH] = (q_in && gy >= 0 && gy < Y && gx >= 0 && gx < X) ? r : 0.0f;
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);  // the refill reuses yv / ye
      issue_y(q + 1);
    }
    // plane pz+1 (requested one iteration ago); issued since: x, y loads of the previous
    // iteration, this iteration's glds, x and y loads
    wait_vm<SL + 2 * (NXC + NY)>();
    lds_barrier();
    slot = slot == 2 ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing may land after the wave has ended
  if constexpr (STATS) {
    // the registers the last prefetches land in stay allocated up to here, and the sums start from here (behind the wait)
    lsr::keep_until_here(xc);
    lsr::keep_until_here(yv);
    lsr::keep_until_here(ye);
    lsr::keep_until_here(nzv);
    asm volatile("" : "+v"(st_flux), "+v"(st_total), "+v"(st_change));
    lsr::RlStats st;
    st.flux = st_flux.x + st_flux.y;
    st.change = st_change;
    st.total = st_total.x + st_total.y;
    lsr::rl_stats_flush<kWaves>(st, smem + T::OFF_B2, p.stats);
  }
#ifdef LSR_FUSED_PROBE_TIME
  if (p.probe && tid == 0) p.probe[4 * blockIdx.x + 3] = wall_clock64();
#endif
}

template <int PZ, int PYX>
bool launch_one(const FusedArgs& p, dim3 grid, hipStream_t s) {
  if constexpr (lsr::fused_compiled(PZ, PYX)) {
    constexpr int RUN = lsr::fused_run(PZ, PYX);
    if (p.stats != nullptr) hipLaunchKernelGGL((rl_fused_sep_kernel<PZ, PYX, RUN, true>), grid, dim3(kThreads), 0, s, p);
    else hipLaunchKernelGGL((rl_fused_sep_kernel<PZ, PYX, RUN, false>), grid, dim3(kThreads), 0, s, p);
    return true;
  } else {
    return false;
  }
}

}  // namespace

namespace lsr {

#define LSR_CAT2(a, b) a##b
#define LSR_CAT(a, b) LSR_CAT2(a, b)
bool LSR_CAT(launch_fused_pz, LSR_FUSED_PZ)(int pyx, const FusedArgs& p, unsigned blocks, hipStream_t s) {
  constexpr int PZ = LSR_FUSED_PZ;
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
