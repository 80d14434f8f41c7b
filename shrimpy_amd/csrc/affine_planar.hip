// Order-1 affine resample for maps whose z axis is decoupled from the plane -- the registration
// case (label-free <-> fluorescence on one instrument): rotation / scale / shear / translation
// inside (y, x), scale + translation along z:
//
//        | a  0  0  tz |        z_in depends on zo only,
//   M =  | 0  b  c  ty |        (y_in, x_in) depend on (yo, xo) only.
//        | 0  d  e  tx |
//
// Replaces the same scipy.ndimage.affine_transform(order=1, mode="constant" | "grid-constant") call as
// affine.hip (which keeps the matrices neither this kernel nor affine_box.hip takes), bit for bit.
//
// Why a second kernel: the general one spends ~170 instructions per voxel, half of them fp64
// coordinate / weight arithmetic, and gathers its 8 taps through the texture-address path
// (0.25 of the HBM peak at config 3).  With the decoupled structure
//   * the in-plane taps (LDS index, 4 weights) of an output pixel are the same for EVERY plane: a
//     thread computes them once for its 8 pixels and keeps them in registers;
//   * the z taps (2 source planes, 2 weights) of an output plane are wave-uniform scalars;
// so a 512-thread workgroup takes a 32 x 128 pixel tile and marches along zo.  Source planes come
// by LDS-DMA (global_load_lds_dwordx4, whole rows of the tile's source box) into a ring of 3 or 4
// slots, one output plane ahead; the 8 taps are LDS reads; what is left per voxel is scipy's own
// 8 x ((v * wz) * wy) * wx + accumulate in fp64 (or 7 f32 FMAs with exact = False).
//
// Algorithmic HBM bytes: 4 * N_src + 4 * N_out.

#include "common.hpp"

#include <cstdio>
#include <cstdlib>

namespace {

// Tiles: (8 NR) x (64 NC) pixels, NR * NC per thread (columns lane + 64 c, rows wave + 8 r).  32 x 128 (NR = 4, NC = 2) is
// the shape of the registrations between two modalities of one instrument; maps that DECIMATE in the plane (a 2 x
// coarser target grid) have source boxes of 160+ KB at that size and used to fall to the gather
// kernel: they take the largest of 32 x 64, 16 x 128, 16 x 64 whose ring fits (lsr::affine_planar_geometry).
constexpr int kWaves = 8;         // the default workgroup: 8 waves; 4 for maps with large source boxes (launch_affine_planar)
struct TileShape { int nr, nc; };
constexpr TileShape kTiles[] = {{4, 2}, {4, 1}, {2, 2}, {2, 1}};

struct PlanarArgs {
  const float* in;
  float* out;
  int Zi, Yi, Xi;
  int pitch;             // source row stride in floats (a multiple of 4, >= Xi rounded up to 4; columns
                         // [Xi, pitch) hold finite values -- they only ever meet weight 0)
  int64_t plane;         // source plane stride in floats (a multiple of 4)
  int64_t opitch, oplane;  // output strides in floats (dense: Xo, Yo * Xo)
  int Zo, Yo, Xo;
  double a, tz;          // z_in = zo * a + tz
  double b, c, ty;       // y_in = (yo * b + xo * c) + ty
  double d, e, tx;       // x_in = (yo * d + xo * e) + tx
  float cval;
  int box_y, box_x;      // LDS box of a source plane (rows, floats per row: a multiple of 4)
  int slots;             // ring depth: 3 (|a| <= 1) or 4
  int tiles_x, tiles_y;
  int z_chunk;           // output planes per workgroup
  int py_n, px_n;        // patches per plane chunk
  int patch_h, patch_w;  // tiles per patch (rows, columns); patch_h * patch_w = 64 = the workgroups an XCD holds
  int per_xcd;           // workgroups per XCD: workgroups b, b + 8, ... run on one XCD
};

// Exact mode: the workgroups an XCD runs side by side form a patch of tiles marching over the same planes
// (see the kernel's block -> tile mapping).
constexpr int kPatchSize = 64;

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds_x4(const float* sbase, int voff, unsigned lds_byte_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               :
               : "v"(voff), "s"(sbase), "s"(lds_byte_addr)
               : "memory");  // (m0 is a RESERVED register to hipcc: naming it as a clobber is refused with a
                             // warning; the compiler re-loads m0 right before each of its own uses instead)
}
// One output value = exactly ONE vector-memory operation, whatever the optimiser thinks of its neighbours:
// the counted wait of the plane loop (`s_waitcnt vmcnt(kPts)`) is only right while every plane issues at
// least kPts stores behind its DMAs.
__device__ __forceinline__ void store_one(float* addr, float v) {
  asm volatile("global_store_dword %0, %1, off" : : "v"(addr), "v"(v) : "memory");
}

// scipy's coordinate ((zo*m0 + yo*m1) + xo*m2) + shift with zo*m0 == 0 (exact: adding +-0 is the
// identity), i.e. (yo*m1 + xo*m2) + shift
__device__ __forceinline__ double plane_coord(double yo, double xo, double m1, double m2, double shift) {
  return lsr::dadd(lsr::dadd(lsr::dmul(yo, m1), lsr::dmul(xo, m2)), shift);
}

// GRID: scipy's mode="grid-constant" -- the volume is continued with cval, so a tap outside it contributes
// cval times its weight and samples up to one voxel outside still blend (mode="constant" drops the whole
// sample).  The box then starts at source index -1 at the earliest: row / column 0 of the box may stand for
// index -1 (a duplicate of index 0 in LDS, replaced by cval through the tap's flag).
// NW waves per workgroup (8, or 4: half-height tiles, twice the workgroups per CU)
template <bool F32, bool GRID, int NR, int NC, int NW>
__global__ __launch_bounds__(64 * NW, 4) void affine_planar_kernel(PlanarArgs p) {  // (<= 128 VGPRs: four waves per SIMD)
  constexpr int kThreads = 64 * NW;
  constexpr int kTY = NW * NR, kTX = 64 * NC, kPts = NR * NC;
  extern __shared__ f32x4 smem4[];
  const float* const smem = reinterpret_cast<const float*>(smem4);
  const unsigned lds_base =
      static_cast<unsigned>(reinterpret_cast<size_t>((__attribute__((address_space(3))) char*)smem4));

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tx, ty, zc;
  if constexpr (F32) {
    // the HBM-bound mode: consecutive workgroups (one per XCD in turn) take consecutive tiles of a row, so
    // the eight XCDs stream neighbouring addresses.  Patches fetch fewer bytes here too (5.1 against 5.8 GB)
    // but ran 6-20 % slower whatever their shape.
    int bid = blockIdx.x;
    tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    ty = bid % p.tiles_y;
    zc = bid / p.tiles_y;
  } else {
    // the fp64-issue-bound mode: every XCD gets a contiguous run of the patch-major order (chunk, patch row,
    // patch column | tile in patch); its 64 resident workgroups march over the same planes and the halo rows
    // and columns their windows share are L2 hits (HBM read 4.4 GB for a 4.3 GB source, 5.8 GB before)
    const int linear = (static_cast<int>(blockIdx.x) & 7) * p.per_xcd + (static_cast<int>(blockIdx.x) >> 3);
    const int inner = linear & (kPatchSize - 1);
    int patch = linear / kPatchSize;
    const int px = patch % p.px_n;
    patch /= p.px_n;
    const int py = patch % p.py_n;
    zc = patch / p.py_n;
    tx = px * p.patch_w + inner % p.patch_w;
    ty = py * p.patch_h + inner / p.patch_w;
  }
  if (tx >= p.tiles_x || ty >= p.tiles_y || zc * p.z_chunk >= p.Zo) return;   // ragged patches, padded grid
  const int x0 = tx * kTX, y0 = ty * kTY;
  const int zo_begin = zc * p.z_chunk, zo_end = min(zo_begin + p.z_chunk, p.Zo);

  // ---- source box of the tile.  The coordinate expression is monotone in each index (every rounded
  // product and sum is), so its minimum over the tile is the value at the corner that takes the low
  // index where the coefficient is >= 0 and the high one where it is negative -- evaluated with the
  // pixels' own expression it IS the smallest coordinate any pixel computes: no slack row or column.
  const int y1 = min(y0 + kTY, p.Yo) - 1, x1 = min(x0 + kTX, p.Xo) - 1;
  const double cy_min = plane_coord(static_cast<double>(p.b < 0.0 ? y1 : y0), static_cast<double>(p.c < 0.0 ? x1 : x0),
                                    p.b, p.c, p.ty);
  const double cx_min = plane_coord(static_cast<double>(p.d < 0.0 ? y1 : y0), static_cast<double>(p.e < 0.0 ? x1 : x0),
                                    p.d, p.e, p.tx);
  constexpr double kLow = GRID ? -1.0 : 0.0;     // first source index a tap may carry
  const int ylo = static_cast<int>(fmin(fmax(floor(cy_min), kLow), static_cast<double>(p.Yi - 1)));
  const int xlo = static_cast<int>(fmin(fmax(floor(cx_min), kLow), static_cast<double>(p.Xi - 1))) & ~3;   // (-1 -> -4)
  const int box_x = p.box_x, box_y = p.box_y;
  const int slot_floats = (box_y * box_x + 255) & ~255;  // whole waves of 16-byte chunks

  // ---- per-pixel in-plane taps, once ------------------------------------------------------
  // The upper neighbour of a tap is always read one element / one row further on, also when it lies
  // past the volume (coordinate exactly on the last index): its weight is then exactly 0 and what the
  // staging put there is a duplicate of in-volume data -- the same zero product (finite inputs).  So
  // the four in-plane taps of a source plane are two ds_read2_b32 at a fixed row stride.
  int idx00[kPts];        // LDS byte offset of (iy0, ix0) inside a slot
  bool ok[kPts];          // pixel inside the output and (mode constant) its coordinate inside the moving plane
  unsigned outside[kPts]; // GRID: taps that lie outside the plane -- bit 0: row iy0, 1: row iy0 + 1, 2: column ix0, 3: ix0 + 1
  double wy0[kPts], wy1[kPts], wx0[kPts], wx1[kPts];
#pragma unroll
  for (int i = 0; i < kPts; ++i) {
    const int yo = y0 + wave + NW * (i / NC), xo = x0 + lane + 64 * (i % NC);
    const double cy = plane_coord(static_cast<double>(yo), static_cast<double>(xo), p.b, p.c, p.ty);
    const double cx = plane_coord(static_cast<double>(yo), static_cast<double>(xo), p.d, p.e, p.tx);
    const bool in_output = yo < p.Yo && xo < p.Xo;
    const bool inside = in_output && !(cy < 0.0) && !(cy > static_cast<double>(p.Yi - 1)) &&
                        !(cx < 0.0) && !(cx > static_cast<double>(p.Xi - 1));
    const double fy = floor(cy), fx = floor(cx);
    const double ry = cy - fy, rx = cx - fx;
    wy0[i] = 1.0 - ry; wy1[i] = 1.0 - wy0[i];
    wx0[i] = 1.0 - rx; wx1[i] = 1.0 - wx0[i];
    int iy0 = 0, ix0 = 0;
    outside[i] = 0;
    if constexpr (GRID) {
      // lower neighbours clamped to [-2, n + 1] (beyond that nothing of the sample is inside and the
      // indices do not matter); the box covers the taps of every sample that touches the plane
      const int sy = static_cast<int>(fmin(fmax(fy, -2.0), static_cast<double>(p.Yi) + 1.0));
      const int sx = static_cast<int>(fmin(fmax(fx, -2.0), static_cast<double>(p.Xi) + 1.0));
      outside[i] = (sy < 0 || sy >= p.Yi ? 1u : 0u) | (sy + 1 < 0 || sy + 1 >= p.Yi ? 2u : 0u) |
                   (sx < 0 || sx >= p.Xi ? 4u : 0u) | (sx + 1 < 0 || sx + 1 >= p.Xi ? 8u : 0u);
      iy0 = min(max(sy - ylo, 0), box_y - 2);
      ix0 = min(max(sx - xlo, 0), box_x - 2);
    } else if (inside) {
      // the box covers every inside pixel's taps by construction (host sizes it from |b|,|c|,|d|,|e|)
      iy0 = min(max(static_cast<int>(fy) - ylo, 0), box_y - 2);
      ix0 = min(max(static_cast<int>(fx) - xlo, 0), box_x - 2);
    }
    idx00[i] = (iy0 * box_x + ix0) * 4;
    ok[i] = GRID ? in_output : inside;
  }
  const bool tile_full = y0 + kTY <= p.Yo && x0 + kTX <= p.Xo;   // every thread stores all its pixels
  bool in_out[NR], col_out[NC];
#pragma unroll
  for (int r = 0; r < NR; ++r) in_out[r] = y0 + wave + NW * r < p.Yo;
#pragma unroll
  for (int c = 0; c < NC; ++c) col_out[c] = x0 + lane + 64 * c < p.Xo;
  if constexpr (!GRID) {
    // mode "constant": a tile none of whose pixels maps into the moving plane is cval on every plane -- no staging, no
    // arithmetic.  (A target grid larger than the moving volume's footprint -- keep_overhang, a decimating map kept at
    // the source's shape -- is mostly such tiles; they used to run the whole pipeline and drop its results.)
    bool any_inside = false;
#pragma unroll
    for (int i = 0; i < kPts; ++i) any_inside |= ok[i];
    if (!__syncthreads_or(any_inside)) {
      for (int zo = zo_begin; zo < zo_end; ++zo) {
        float* orow = p.out + static_cast<int64_t>(zo) * p.oplane + static_cast<int64_t>(y0) * p.opitch + x0;
#pragma unroll
        for (int i = 0; i < kPts; ++i)
          if (in_out[i / NC] && col_out[i % NC])
            store_one(orow + static_cast<int64_t>(wave + NW * (i / NC)) * p.opitch + lane + 64 * (i % NC), p.cval);
      }
      return;
    }
  }

  // ---- staging: 16-byte chunks of the box, lane-linear in LDS -------------------------------
  const int chunks_x = box_x >> 2;
  const int n_chunks = box_y * chunks_x;
  constexpr int kMaxLoads = 8;  // host keeps n_chunks <= 8 * threads
  int s_voff[kMaxLoads];
#pragma unroll
  for (int k = 0; k < kMaxLoads; ++k) {
    const int e = min(tid + k * kThreads, n_chunks - 1);
    const int r = e / chunks_x, c4 = e - r * chunks_x;
    const int gy = min(max(ylo + r, 0), p.Yi - 1);    // rows / columns past the volume: duplicates
    const int gx = min(max(xlo + 4 * c4, 0), ((p.Xi + 3) & ~3) - 4);
    s_voff[k] = (gy * p.pitch + gx) * 4;
  }
  const int n_loads = (n_chunks + kThreads - 1) / kThreads;  // scalar
  auto issue_plane = [&](int zs, int slot) {
    const float* src = p.in + static_cast<int64_t>(zs) * p.plane;
    const unsigned dst = lds_base + (slot * slot_floats + wave * 64 * 4) * 4;
#pragma unroll
    for (int k = 0; k < kMaxLoads; ++k)
      if (k < n_loads && wave * 64 + k * kThreads < n_chunks) glds_x4(src, s_voff[k], dst + k * kThreads * 16);
  };

  // ring bookkeeping (scalars): slot s holds source plane resident[s]
  int resident[4] = {-1, -1, -1, -1};
  const int slots = p.slots;
  auto slot_of = [&](int zs) { return zs % slots; };
  // source planes and weights of output plane zo; false (mode constant only): the plane is cval.  GRID:
  // `zout` bit 0 / 1 = the lower / upper source plane lies outside the volume (z0 / z1 are then clamped
  // stand-ins whose values are replaced by cval)
  auto z_taps = [&](int zo, int& z0, int& z1, double& wz0, double& wz1, unsigned& zout) {
    zout = 0;
    const double cz = lsr::dadd(lsr::dmul(static_cast<double>(zo), p.a), p.tz);
    if (!GRID && (cz < 0.0 || cz > static_cast<double>(p.Zi - 1))) return false;
    const double fz = floor(cz), rz = cz - fz;
    wz0 = 1.0 - rz;
    wz1 = 1.0 - wz0;
    if constexpr (GRID) {
      const int sz = static_cast<int>(fmin(fmax(fz, -2.0), static_cast<double>(p.Zi) + 1.0));
      zout = (sz < 0 || sz >= p.Zi ? 1u : 0u) | (sz + 1 < 0 || sz + 1 >= p.Zi ? 2u : 0u);
      z0 = min(max(sz, 0), p.Zi - 1);
      z1 = min(max(sz + 1, 0), p.Zi - 1);
    } else {
      z0 = static_cast<int>(fz);
      z1 = min(z0 + 1, p.Zi - 1);
    }
    return true;
  };
  auto request = [&](int zs) {  // make plane zs resident (issue its DMA if it is not)
    const int s = slot_of(zs);
    bool have = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) have |= (k == s && resident[k] == zs);
    if (!have) {
      issue_plane(zs, s);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (k == s) resident[k] = zs;
    }
  };

  {  // prologue: the first output plane's sources
    int z0, z1;
    double w0, w1;
    unsigned zf;
    if (zo_begin < zo_end && z_taps(zo_begin, z0, z1, w0, w1, zf)) {
      request(z0);
      request(z1);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (no stores behind the prologue's DMAs yet: wait for all)
  for (int zo = zo_begin; zo < zo_end; ++zo) {
    int z0 = 0, z1 = 0;
    double wz0 = 0.0, wz1 = 0.0;
    unsigned zout = 0;
    const bool z_in = z_taps(zo, z0, z1, wz0, wz1, zout);
    // this plane's sources were requested one iteration ago: wait, publish.  Vector-memory operations
    // retire in issue order and stores share the counter: behind those DMAs this wave issued only the
    // previous plane's stores -- exactly kPts of them in a tile that lies wholly inside the output -- so
    // waiting for "all but kPts" waits for the DMAs and lets the stores drain on their own (waiting
    // for 0 stalled every plane on its predecessor's HBM writes).
    if (tile_full) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(kPts) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // next plane's sources: their slots differ from this plane's (|a| <= 1.5, ring of 3 or 4)
    {
      int n0, n1;
      double u0, u1;
      unsigned nf;
      if (zo + 1 < zo_end && z_taps(zo + 1, n0, n1, u0, u1, nf)) {
        request(n0);
        request(n1);
      }
    }
    float* orow = p.out + static_cast<int64_t>(zo) * p.oplane + static_cast<int64_t>(y0) * p.opitch + x0;
    const char* s0 = reinterpret_cast<const char*>(smem + slot_of(z0) * slot_floats);
    const char* s1 = reinterpret_cast<const char*>(smem + slot_of(z1) * slot_floats);
    typedef float f32x2 __attribute__((ext_vector_type(2), aligned(4)));
    const int row_b = box_x * 4;
    // G pixels at a time: all their LDS reads first, then the arithmetic -- nothing in between is
    // conditional, so the pixels overlap (an `if (inside)` per pixel made hipcc run them one by one,
    // each waiting for its own reads)
    constexpr int G = kPts < (F32 ? 4 : 2) ? kPts : (F32 ? 4 : 2);   // (the fp64 corner sums of four pixels at once do not fit 128 VGPRs)
    static_assert(kPts % G == 0, "pixels per thread come in whole groups");
#pragma unroll
    for (int h = 0; h < kPts; h += G) {
      f32x2 v[G][4];
#pragma unroll
      for (int k = 0; k < G; ++k) {
        const int o = idx00[h + k];
        v[k][0] = *reinterpret_cast<const f32x2*>(s0 + o);
        v[k][1] = *reinterpret_cast<const f32x2*>(s0 + o + row_b);
        v[k][2] = *reinterpret_cast<const f32x2*>(s1 + o);
        v[k][3] = *reinterpret_cast<const f32x2*>(s1 + o + row_b);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (GRID) {   // taps outside the volume carry cval (flags: in-plane per pixel, z per plane)
#pragma unroll
        for (int k = 0; k < G; ++k) {
          const unsigned f = outside[h + k];
          const bool r0 = f & 1u, r1 = f & 2u, c0 = f & 4u, c1 = f & 8u, p0 = zout & 1u, p1 = zout & 2u;
          v[k][0].x = (p0 || r0 || c0) ? p.cval : v[k][0].x;
          v[k][0].y = (p0 || r0 || c1) ? p.cval : v[k][0].y;
          v[k][1].x = (p0 || r1 || c0) ? p.cval : v[k][1].x;
          v[k][1].y = (p0 || r1 || c1) ? p.cval : v[k][1].y;
          v[k][2].x = (p1 || r0 || c0) ? p.cval : v[k][2].x;
          v[k][2].y = (p1 || r0 || c1) ? p.cval : v[k][2].y;
          v[k][3].x = (p1 || r1 || c0) ? p.cval : v[k][3].x;
          v[k][3].y = (p1 || r1 || c1) ? p.cval : v[k][3].y;
        }
      }
      float res[G];
#pragma unroll
      for (int k = 0; k < G; ++k) {
        const int i = h + k;
        float result;
        if constexpr (F32) {
          const float fx = static_cast<float>(wx1[i]), fy = static_cast<float>(wy1[i]),
                      fz = static_cast<float>(wz1);
          const float a0 = fmaf(fx, v[k][0].y - v[k][0].x, v[k][0].x), a1 = fmaf(fx, v[k][1].y - v[k][1].x, v[k][1].x);
          const float b0 = fmaf(fx, v[k][2].y - v[k][2].x, v[k][2].x), b1 = fmaf(fx, v[k][3].y - v[k][3].x, v[k][3].x);
          const float c0 = fmaf(fy, a1 - a0, a0), c1 = fmaf(fy, b1 - b0, b0);
          result = fmaf(fz, c1 - c0, c0);
        } else {
          // scipy's corner order and product order: ((v * wz) * wy) * wx, summed in sequence
          double t = 0.0;
          auto corner = [&](float val, double wz, double wy, double wx) {
            t = lsr::dadd(t, lsr::dmul(lsr::dmul(lsr::dmul(static_cast<double>(val), wz), wy), wx));
          };
          corner(v[k][0].x, wz0, wy0[i], wx0[i]);
          corner(v[k][0].y, wz0, wy0[i], wx1[i]);
          corner(v[k][1].x, wz0, wy1[i], wx0[i]);
          corner(v[k][1].y, wz0, wy1[i], wx1[i]);
          corner(v[k][2].x, wz1, wy0[i], wx0[i]);
          corner(v[k][2].y, wz1, wy0[i], wx1[i]);
          corner(v[k][3].x, wz1, wy1[i], wx0[i]);
          corner(v[k][3].y, wz1, wy1[i], wx1[i]);
          result = static_cast<float>(t);
        }
        res[k] = (z_in && ok[i]) ? result : p.cval;
        asm volatile("" : "+v"(res[k]));   // (keeps hipcc from sinking the pixel into its store's condition)
      }
#pragma unroll
      for (int k = 0; k < G; ++k) {
        const int i = h + k;
        if (in_out[i / NC] && col_out[i % NC])
          store_one(orow + static_cast<int64_t>(wave + NW * (i / NC)) * p.opitch + lane + 64 * (i % NC), res[k]);
      }
    }
    // (no barrier here: the next iteration's DMAs are issued behind its own barrier, which every
    // wave reaches only after these reads)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace

namespace lsr {

// Returns true if the planar kernel took the launch; false = not applicable, use the general one.
// Geometry of the planar path for this matrix and moving volume; false = not applicable.
bool affine_planar_geometry(int64_t Yi, int64_t Xi, int64_t pitch, const double M[12], int* box_y_out, int* box_x_out,
                            int* slots_out, int64_t* lds_bytes_out, int* tile_out, int waves) {
  if (!volume_in_range(1, Yi, Xi) || !strides_in_range(pitch, 0)) return false;
  if (M[1] != 0.0 || M[2] != 0.0 || M[4] != 0.0 || M[8] != 0.0) return false;
  const double a = M[0] < 0 ? -M[0] : M[0];
  // rows start on 16-byte boundaries (LDS-DMA moves 16-byte chunks) and hold whole chunks up to the last column
  if (a > 1.5 || pitch % 4 != 0 || pitch < ((Xi + 3) & ~int64_t(3)) || Xi < 8 || Yi < 2) return false;
  auto ab = [](double v) { return v < 0 ? -v : v; };
  if (Yi * pitch * 4 >= (int64_t(1) << 31)) return false;  // 32-bit in-plane byte offsets
  const int slots = a <= 1.0 ? 3 : 4;
  // the largest tile whose ring fits.  Source box of a tile: span of floor() over the tile (<= floor(extent) + 2; the
  // 1e-6 absorbs the different summation order on the device) + 1 for the upper neighbour [+ 3 + 3: 16-byte alignment
  // of the first column, rows rounded up to whole chunks]
  for (int t = 0; t < static_cast<int>(sizeof(kTiles) / sizeof(kTiles[0])); ++t) {
    const int ty = waves * kTiles[t].nr, tx = 64 * kTiles[t].nc;
    const double ey = ab(M[5]) * (ty - 1) + ab(M[6]) * (tx - 1), ex = ab(M[9]) * (ty - 1) + ab(M[10]) * (tx - 1);
    if (!(ey < 4096.0) || !(ex < 4096.0)) continue;
    const int box_y = static_cast<int>(ey + 1e-6) + 3;
    const int box_x = (static_cast<int>(ex + 1e-6) + 3 + 3 + 3) & ~3;
    const int64_t lds_bytes = int64_t(slots) * ((int64_t(box_y) * box_x + 255) & ~int64_t(255)) * 4;
    if (lds_bytes > 150 * 1024 || int64_t(box_y) * (box_x / 4) > 8 * 64 * waves) continue;
    *box_y_out = box_y; *box_x_out = box_x; *slots_out = slots; *lds_bytes_out = lds_bytes;
    if (tile_out != nullptr) *tile_out = t;
    return true;
  }
  return false;
}

bool launch_affine_planar(const float* in, int64_t Zi, int64_t Yi, int64_t Xi, int64_t pitch, int64_t plane,
                          float* out, int64_t Zo, int64_t Yo, int64_t Xo, int64_t opitch, int64_t oplane,
                          const double M[12], float cval, bool f32, bool grid, hipStream_t s) {
  int box_y, box_x, slots, tile = 0;
  int64_t lds_bytes;
  if (plane % 4 != 0 || (reinterpret_cast<uintptr_t>(in) & 15) != 0) return false;
  int waves = kWaves;
  if (!affine_planar_geometry(Yi, Xi, pitch, M, &box_y, &box_x, &slots, &lds_bytes, &tile, waves)) return false;
  // Maps whose source box is much larger than the tile because they ROTATE in the plane (10 degrees and more) run 3-16 %
  // faster on half-height tiles with four waves (twice the workgroups per CU, their per-plane barriers independent);
  // near-identity maps (config 3: box 1.3 x the tile) lose 2-6 % there (round 4, profiles/r04_sweep_geometry.jsonl).
  {
    const int64_t tile_px = int64_t(waves * kTiles[tile].nr) * (64 * kTiles[tile].nc);
    // (with fewer tiles than CUs the z chunks already spread the work; halving the tiles there lost 11-28 %)
    const int64_t tiles8 = ceil_div(Yo, int64_t(waves * kTiles[tile].nr)) * ceil_div(Xo, int64_t(64 * kTiles[tile].nc));
    // (rotations only: a 2x decimation gains 5-13 % on a full-size target and loses 10-25 % on a half-size one)
    auto mag = [](double v) { return v < 0 ? -v : v; };
    bool four = int64_t(box_y) * box_x * 10 > tile_px * 16 && tiles8 >= 256 && (mag(M[6]) > 0.1 || mag(M[9]) > 0.1);
    if (const char* e = std::getenv("LSR_PLANAR_WAVES")) four = std::atoi(e) == 4;      // measurement override
    int by, bx, sl, tl = 0;
    int64_t lb;
    if (four && affine_planar_geometry(Yi, Xi, pitch, M, &by, &bx, &sl, &lb, &tl, 4)) {
      waves = 4; box_y = by; box_x = bx; slots = sl; lds_bytes = lb; tile = tl;
    }
  }
  const int kTY = waves * kTiles[tile].nr, kTX = 64 * kTiles[tile].nc;

  PlanarArgs p;
  p.in = in; p.out = out;
  p.Zi = static_cast<int>(Zi); p.Yi = static_cast<int>(Yi); p.Xi = static_cast<int>(Xi);
  p.pitch = static_cast<int>(pitch); p.plane = plane;
  p.opitch = opitch; p.oplane = oplane;
  p.Zo = static_cast<int>(Zo); p.Yo = static_cast<int>(Yo); p.Xo = static_cast<int>(Xo);
  p.a = M[0]; p.tz = M[3];
  p.b = M[5]; p.c = M[6]; p.ty = M[7];
  p.d = M[9]; p.e = M[10]; p.tx = M[11];
  p.cval = cval;
  p.box_y = box_y; p.box_x = box_x; p.slots = slots;
  p.tiles_x = static_cast<int>(ceil_div(Xo, kTX));
  p.tiles_y = static_cast<int>(ceil_div(Yo, kTY));
  // z split: enough workgroups for ~4 per CU, chunks of at least 16 planes (each chunk refetches
  // one source plane)
  const int64_t tiles = int64_t(p.tiles_x) * p.tiles_y;
  int64_t chunks = ceil_div(256 * 4, tiles);
  if (chunks < 1) chunks = 1;
  int64_t chunk = ceil_div(Zo, chunks);
  if (chunk < 16) chunk = 16;
  if (chunk > Zo) chunk = Zo;
  p.z_chunk = static_cast<int>(chunk);
  p.patch_h = 8; p.patch_w = 8;
  if (const char* e = std::getenv("LSR_PLANAR_PATCH")) {   // measurement override: "h,w" with h * w = 64
    int h = 0, w = 0;
    if (std::sscanf(e, "%d,%d", &h, &w) == 2 && h > 0 && w > 0 && h * w == kPatchSize) { p.patch_h = h; p.patch_w = w; }
  }
  p.py_n = static_cast<int>(ceil_div(p.tiles_y, p.patch_h));
  p.px_n = static_cast<int>(ceil_div(p.tiles_x, p.patch_w));
  const int64_t padded = int64_t(p.py_n) * p.px_n * ceil_div(Zo, chunk) * kPatchSize;
  if (padded >= (int64_t(1) << 30)) return false;
  p.per_xcd = static_cast<int>(ceil_div(padded, 8));
  const int64_t blocks = f32 ? tiles * ceil_div(Zo, chunk) : int64_t(p.per_xcd) * 8;
  static std::atomic<uint64_t> lds_allowed[32] = {};
  using Kernel = void (*)(PlanarArgs);
#define LSR_PLANAR_TILE(NR, NC, NW)                                                                            \
  {affine_planar_kernel<false, false, NR, NC, NW>, affine_planar_kernel<true, false, NR, NC, NW>,              \
   affine_planar_kernel<false, true, NR, NC, NW>, affine_planar_kernel<true, true, NR, NC, NW>}
  static const Kernel kernels[2][4][4] = {
      {LSR_PLANAR_TILE(4, 2, 8), LSR_PLANAR_TILE(4, 1, 8), LSR_PLANAR_TILE(2, 2, 8), LSR_PLANAR_TILE(2, 1, 8)},
      {LSR_PLANAR_TILE(4, 2, 4), LSR_PLANAR_TILE(4, 1, 4), LSR_PLANAR_TILE(2, 2, 4), LSR_PLANAR_TILE(2, 1, 4)}};
#undef LSR_PLANAR_TILE                                          // [waves == 4][tile][2 * grid + f32], tiles as kTiles
  const int w4 = waves == 4 ? 1 : 0;
  const Kernel kernel = kernels[w4][tile][2 * grid + f32];
  if (lsr::allow_dynamic_lds(reinterpret_cast<const void*>(kernel), 150 * 1024,
                             lds_allowed[16 * w4 + 4 * tile + 2 * grid + f32], "affine_planar_kernel") != LSR_OK)
    return false;   // the caller runs the gather kernel
  hipLaunchKernelGGL(kernel, dim3(static_cast<unsigned>(blocks)), dim3(64 * waves),
                     static_cast<size_t>(lds_bytes), s, p);
  return true;
}

}  // namespace lsr
