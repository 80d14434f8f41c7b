// Order-1 affine resample for ANY 3x4 map whose source footprint per block fits in LDS -- in
// particular the maps that couple z with the plane (a label-free <-> light-sheet registration of an
// oblique system is tilted: z_in depends on (yo, xo), and (y_in, x_in) on zo), which
// affine_planar.hip cannot take and which used to fall to the 8-tap global gather of affine.hip
// (0.24-0.29 of the HBM peak: ~170 instructions per voxel and gathers through the texture path).
//
// Same scipy.ndimage.affine_transform(order=1, mode="constant") arithmetic, bit for bit (exact) or
// with f32 interpolation (LSR_MODE_F32_INTERP): coordinates ((zo*m0 + yo*m1) + xo*m2) + shift in
// fp64, the border rule on them, weights w0 = 1 - f, w1 = 1 - w0, corners summed in scipy's order.
//
// Structure: a 256-thread workgroup owns a TZ x TY x TX block of 4096 OUTPUT voxels (the host picks
// the shape that makes the source box smallest for this matrix: a tilt about y wants a short TX, a
// tilt about x a short TY).  The source voxels the block can touch lie in the bounding box of its 8
// corners' coordinates -- the map is linear and every rounding in the coordinate expression is
// monotone, so the corner values, evaluated with the same expression, ARE the extremes: no slack.
// The workgroup stages that box into LDS with LDS-DMA (global_load_lds_dwordx4: whole
// 16-byte-aligned row pieces, no staging registers), waits once, and then every tap is an LDS read.
// Four workgroups fit on a CU when the box is under 39 KB (near-identity maps: ~28 KB), so three
// compute while the fourth's box is in flight; there is no ring and no per-plane barrier.
// (Rounds 2-3 ran blocks of 8192 voxels on 512 threads, two per CU: 2-6 % slower in exact mode and up to
// 12 % in f32 mode on a rotated + tilted map, equal on a pure tilt -- round 4, profiles/r04_affine_kernels.jsonl.)
//
// A thread owns 16 / TZ output pixels (yo, xo) and walks them along zo: yo*m1 and xo*m2 are per
// pixel, zo*m0 per plane -- scipy's sum order makes those prefixes exact to hoist -- so a voxel costs
// 9 fp64 adds for its coordinates, then v_fract / v_cvt for the taps.  The voxel body is branch-free
// (out-of-range voxels read clamped taps and drop the result), so hipcc interleaves the independent
// voxels of a thread and the fp64 / LDS latencies overlap.
//
// Workgroups are numbered so that the ones resident on an XCD together form a 4 x 4 x 4 patch of
// blocks: neighbouring boxes overlap by their halo, and the overlap is then an L2 hit, not a second
// HBM read.
//
// Algorithmic HBM bytes: 4 * N_src + 4 * N_out.

#include "common.hpp"

#include <algorithm>
#include <cstdlib>

namespace {

constexpr int kThreads = 256;
constexpr int kVoxelsPerThread = 16;         // blocks of 4096 output voxels
constexpr int kMaxLoads = 20;                // 16-byte chunks per thread: boxes up to 160 KB
constexpr int kPatch = 4;                    // blocks per patch edge

struct BoxArgs {
  const float* in;
  float* out;
  int Zi, Yi, Xi;
  unsigned pitch, plane;   // source strides in floats (multiples of 4; columns [Xi, pitch) finite: they only meet weight 0)
  unsigned opitch;         // output row stride in floats (host: Yo * opitch < 2^31)
  int64_t oplane;          // output plane stride in floats
  int Zo, Yo, Xo;
  double m[12];
  float cval;
  int ty, tx, tx_shift;      // block rows / columns (powers of two; TZ is the template parameter)
  int box_z, box_y, box_x;   // staged box (planes, rows, floats per row: a multiple of 4)
  float inv_cx, inv_by;      // 1 / (box_x / 4), 1 / box_y  (index -> (plane, row, chunk) without a divide)
  int tz_n, ty_n, tx_n;      // blocks per axis
  int linear;                // 1: patch-major order as numbered (f32 mode); 0: a contiguous run of it per XCD (exact mode)
  int pz_n, py_n, px_n;      // patches per axis
  int per_xcd;               // blocks per XCD (padded grid / 8)
  int probe;                 // diagnostics (-DLSR_BOX_PROBES, env LSR_BOX_PROBE): 1 = no staging, 3 = staging + stores only
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds_x4(const float* sbase, unsigned voff, unsigned lds_byte_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               :
               : "v"(voff), "s"(sbase), "s"(lds_byte_addr)
               : "memory");  // (m0 is a RESERVED register to hipcc: naming it as a clobber is refused with a
                             // warning; the compiler re-loads m0 right before each of its own uses instead)
}

// extremes of one coordinate over the block.  The coordinate expression is monotone in each index
// (every product and every rounded sum is), so the minimum sits at the corner that takes, per axis,
// the low index where the coefficient is >= 0 and the high index where it is negative -- evaluated
// with the voxels' own expression, it IS the smallest value any voxel of the block computes.
__device__ __forceinline__ void corner_range(const double* m, const int lo[3], const int hi[3], double& cmin,
                                             double& cmax) {
  double a[3], b[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const bool up = !(m[k] < 0.0);
    a[k] = static_cast<double>(up ? lo[k] : hi[k]);
    b[k] = static_cast<double>(up ? hi[k] : lo[k]);
  }
  cmin = lsr::affine_coord(a[0], a[1], a[2], m[0], m[1], m[2], m[3]);
  cmax = lsr::affine_coord(b[0], b[1], b[2], m[0], m[1], m[2], m[3]);
}

// One axis of one voxel: is the coordinate inside [0, n - 1] (scipy's test, on the fp64 value), its
// lower tap and its fraction.  GRID (mode "grid-constant": the volume continued with cval): the lower tap
// clamped to [-2, n + 1] and which of the two taps lie outside the volume (bit 0: lower, bit 1: upper).
struct Tap {
  int i;        // floor(c) for an inside coordinate (trunc(c) = floor(c) for c >= 0)
  double f;     // c - floor(c), exact (v_fract_f64)
  bool inside;
  unsigned out; // GRID only
};
template <bool GRID, bool INTERIOR = false>
__device__ __forceinline__ Tap axis_tap(double c, double last) {
  Tap t;
  t.f = __builtin_amdgcn_fract(c);
  if constexpr (INTERIOR) {   // the block's corner coordinates all lie in [0, n - 1): nothing to test
    t.i = static_cast<int>(c);
    t.inside = true;
    t.out = 0;
  } else if constexpr (GRID) {
    const int i = static_cast<int>(fmin(fmax(floor(c), -2.0), last + 2.0));
    const int n = static_cast<int>(last) + 1;
    t.i = i;
    t.out = (static_cast<unsigned>(i < 0) | static_cast<unsigned>(i >= n)) |
            ((static_cast<unsigned>(i + 1 < 0) | static_cast<unsigned>(i + 1 >= n)) << 1);
    t.inside = true;
  } else {
    t.i = static_cast<int>(c);   // v_cvt_i32_f64: towards zero, saturating
    // (bitwise on purpose: short-circuit evaluation would turn every voxel into its own exec-masked
    // region and keep the independent voxels of a thread from overlapping)
    t.inside = static_cast<bool>(static_cast<int>(!(c < 0.0)) & static_cast<int>(!(c > last)));
    t.out = 0;
  }
  return t;
}

// One block of the launch: its output origin, the origin of its source box, and whether it exists
// (the patch grid is padded) / sees anything of the moving volume.
struct Blk {
  int z0, y0, x0;
  int zlo, ylo, xlo;
  bool valid, blind;
  bool interior;   // every coordinate of the block lies in [0, n - 1) on its axis: no tap leaves the volume
};

// Block g of the XCD-ordered list: patch-major, z fastest inside a 4 x 4 x 4 patch.
template <int TZ, bool GRID>
__device__ __forceinline__ Blk locate(const BoxArgs& p, int g) {
  Blk b;
  const int patch = g >> 6, inner = g & 63;
  const int pz = patch % p.pz_n, py = (patch / p.pz_n) % p.py_n, px = patch / (p.pz_n * p.py_n);
  const int bz = pz * kPatch + (inner & 3), by = py * kPatch + ((inner >> 2) & 3), bx = px * kPatch + (inner >> 4);
  b.valid = bz < p.tz_n && by < p.ty_n && bx < p.tx_n;   // (padding of the patch grid; wave-uniform)
  b.z0 = bz * TZ; b.y0 = by * p.ty; b.x0 = bx * p.tx;
  const int lo[3] = {b.z0, b.y0, b.x0};
  const int hi[3] = {min(b.z0 + TZ, p.Zo) - 1, min(b.y0 + p.ty, p.Yo) - 1, min(b.x0 + p.tx, p.Xo) - 1};
  double zmin, zmax, ymin, ymax, xmin, xmax;
  corner_range(p.m + 0, lo, hi, zmin, zmax);
  corner_range(p.m + 4, lo, hi, ymin, ymax);
  corner_range(p.m + 8, lo, hi, xmin, xmax);
  const double Zl = static_cast<double>(p.Zi - 1), Yl = static_cast<double>(p.Yi - 1), Xl = static_cast<double>(p.Xi - 1);
  // GRID: a tap may carry source index -1 (the box's first plane / row / column then stands for it: a
  // duplicate of index 0 in LDS, replaced by cval through the tap's flag), and a coordinate up to one
  // voxel outside still touches the volume
  constexpr double kLow = GRID ? -1.0 : 0.0;
  b.zlo = static_cast<int>(fmin(fmax(floor(zmin), kLow), Zl));
  b.ylo = static_cast<int>(fmin(fmax(floor(ymin), kLow), Yl));
  b.xlo = static_cast<int>(fmin(fmax(floor(xmin), kLow), Xl)) & ~3;   // (-1 -> -4)
  // a block that sees nothing of the volume: every voxel is cval (GRID: a weighted sum of cvals), nothing to stage
  b.blind = GRID ? (!(zmax > -1.0) || !(zmin < Zl + 1.0) || !(ymax > -1.0) || !(ymin < Yl + 1.0) || !(xmax > -1.0) ||
                    !(xmin < Xl + 1.0))
                 : (zmax < 0.0 || zmin > Zl || ymax < 0.0 || ymin > Yl || xmax < 0.0 || xmin > Xl);
  b.interior = zmin >= 0.0 && zmax < Zl && ymin >= 0.0 && ymax < Yl && xmin >= 0.0 && xmax < Xl;
  return b;
}

// Issue the LDS-DMA of a block's source box: chunk e = tid + NT k, LDS image linear in e.
template <int NT>
__device__ __forceinline__ void stage(const BoxArgs& p, const Blk& b, unsigned lds_byte_base, int tid, int wave) {
  const int box_x = p.box_x, box_y = p.box_y, box_z = p.box_z;
  const int chunks_x = box_x >> 2;
  const int n_chunks = box_z * box_y * chunks_x;
  const unsigned plane_i = p.plane;
  // 32-bit byte offsets are taken from the box's first source plane (host: box_z planes < 4 GiB)
  const float* const src = p.in + static_cast<int64_t>(max(b.zlo, 0)) * plane_i;
  const int n_loads = (n_chunks + NT - 1) / NT;
  for (int k = 0; k < n_loads; ++k) {
    const int e = min(tid + k * NT, n_chunks - 1);
    // e -> (plane, row, chunk); the reciprocals are exact enough for e < 2^20 with the fix-up
    int row = static_cast<int>((static_cast<float>(e) + 0.5f) * p.inv_cx);
    row -= (row * chunks_x > e);
    row += ((row + 1) * chunks_x <= e);
    const int c4 = e - row * chunks_x;
    int pl = static_cast<int>((static_cast<float>(row) + 0.5f) * p.inv_by);
    pl -= (pl * box_y > row);
    pl += ((pl + 1) * box_y <= row);
    const int r = row - pl * box_y;
    // planes / rows / columns past the volume (or, GRID, before it): duplicates of the nearest inside one
    const int zfirst = max(b.zlo, 0);
    const unsigned gz = static_cast<unsigned>(min(max(b.zlo + pl, 0), p.Zi - 1) - zfirst);
    const unsigned gy = static_cast<unsigned>(min(max(b.ylo + r, 0), p.Yi - 1));
    const unsigned gx = static_cast<unsigned>(min(max(b.xlo + 4 * c4, 0), ((p.Xi + 3) & ~3) - 4));
    const unsigned voff = (gz * plane_i + gy * p.pitch + gx) * 4u;
    if (wave * 64 + k * NT < n_chunks)   // wave-uniform: whole waves of chunks
      glds_x4(src, voff, __builtin_amdgcn_readfirstlane(lds_byte_base + static_cast<unsigned>((k * NT + wave * 64) * 16)));
  }
}

// DEP: bit k set = source coordinate k (z, y, x) depends on zo.  A coordinate that does not is
// worked out once per pixel instead of once per voxel (a tilt about y leaves y_in free of zo, a tilt
// about x leaves x_in).
template <bool F32, int TZ, int DEP, int NT, bool GRID, bool INTERIOR = false>
__device__ __forceinline__ void compute(const BoxArgs& p, const Blk& b, const float* smem, int tid, int probe) {
  constexpr int P = kVoxelsPerThread / TZ;   // output pixels per thread
  static_assert(P >= 1 && P * TZ == kVoxelsPerThread, "block shape");
  if constexpr (!INTERIOR) {
    // Most blocks of a registration never see the border: both taps of every axis are inside the volume, the two
    // border rules are the same arithmetic there, and the test-free walk does it (workgroup-uniform branch).
    // grid-constant: the clamp to [-2, n + 1], the six flags and the eight selects per voxel had cost 1.0 of 4.3 ms
    // on a tilted map.  constant: the six fp64 compares of the inside test go too -- worth 0-2 % (round 4: 3.30 ms
    // either way at config-3 size; the walk is not bound by how many fp64 instructions it issues).
    if (b.interior) {
      compute<F32, TZ, DEP, NT, false, true>(p, b, smem, tid, probe);
      return;
    }
  }
  const int z0 = b.z0, y0 = b.y0, x0 = b.x0, zlo = b.zlo, ylo = b.ylo, xlo = b.xlo;
  const int box_x = p.box_x, box_y = p.box_y, box_z = p.box_z;
  const int plane_floats = box_y * box_x;
  const double Zl = static_cast<double>(p.Zi - 1), Yl = static_cast<double>(p.Yi - 1), Xl = static_cast<double>(p.Xi - 1);

  // ---- per-pixel constants -------------------------------------------------------------------
  // pixel q = tid + NT j of the block's TY x TX plane: consecutive lanes are consecutive xo
  unsigned pix[P];   // yo * opitch + xo: the output offset inside a plane (host: Yo * opitch < 2^31)
  double tzy[P], tyy[P], txy[P], tzx[P], tyx[P], txx[P];
  bool ok[P];
#pragma unroll
  for (int j = 0; j < P; ++j) {
    const int q = tid + j * NT;
    const int yo = y0 + (q >> p.tx_shift), xo = x0 + (q & (p.tx - 1));
    ok[j] = yo < p.Yo && xo < p.Xo;
    pix[j] = static_cast<unsigned>(yo) * p.opitch + static_cast<unsigned>(xo);
    const double yd = static_cast<double>(yo), xd = static_cast<double>(xo);
    tzy[j] = lsr::dmul(yd, p.m[1]); tyy[j] = lsr::dmul(yd, p.m[5]); txy[j] = lsr::dmul(yd, p.m[9]);
    tzx[j] = lsr::dmul(xd, p.m[2]); tyx[j] = lsr::dmul(xd, p.m[6]); txx[j] = lsr::dmul(xd, p.m[10]);
  }

  // coordinates that do not depend on zo: their taps, once per pixel (zo * 0 is an exact zero, so
  // any plane's expression gives the value every plane computes)
  Tap hz[P], hy[P], hx[P];
  {
    const double zd = static_cast<double>(z0);
#pragma unroll
    for (int j = 0; j < P; ++j) {
      hz[j] = axis_tap<GRID, INTERIOR>(lsr::dadd(lsr::dadd(lsr::dadd(lsr::dmul(zd, p.m[0]), tzy[j]), tzx[j]), p.m[3]), Zl);
      hy[j] = axis_tap<GRID, INTERIOR>(lsr::dadd(lsr::dadd(lsr::dadd(lsr::dmul(zd, p.m[4]), tyy[j]), tyx[j]), p.m[7]), Yl);
      hx[j] = axis_tap<GRID, INTERIOR>(lsr::dadd(lsr::dadd(lsr::dadd(lsr::dmul(zd, p.m[8]), txy[j]), txx[j]), p.m[11]), Xl);
    }
  }

  const bool live = !b.blind;
  // The upper neighbour of a tap is always read one element / row / plane further on, also when it
  // lies past the volume (coordinate exactly on the last index): its weight is then exactly 0, and
  // what the staging put there is a duplicate of in-volume data, so the product is the same zero the
  // lower neighbour would give (finite inputs).  That keeps the eight taps at fixed strides:
  // four ds_read2_b32 with offsets (0, 1).
  const int row_b = box_x * 4, plane_b = plane_floats * 4;
  const int o_max = (box_z * plane_floats - plane_floats - box_x - 2) * 4;   // clamp for dropped voxels
  const char* const smem_b = reinterpret_cast<const char*>(smem);
  typedef float f32x2 __attribute__((ext_vector_type(2), aligned(4)));
  // G = 4 voxels at a time (P pixels x U planes): their coordinate chains, LDS reads and
  // interpolations are independent, and nothing in between is conditional, so they overlap
  constexpr int U = P >= 4 ? 1 : 4 / P;
  constexpr int G = P * U;
  const int nz = min(TZ, p.Zo - z0);   // uniform
  if (probe == 3) {   // diagnostics: staging + stores only
    for (int dz = 0; dz < nz; ++dz)
#pragma unroll
      for (int j = 0; j < P; ++j)
        if (ok[j]) p.out[static_cast<int64_t>(z0 + dz) * p.oplane + pix[j]] = smem[tid];
    return;
  }
  for (int dz = 0; dz < nz; dz += U) {
    Tap az[G], ay[G], ax[G];
    f32x2 v[G][4];
    bool inside[G];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int zo = min(z0 + dz + u, p.Zo - 1);   // (a plane past the volume repeats the last one; not stored)
      const double zd = static_cast<double>(zo);
      const double tzz = lsr::dmul(zd, p.m[0]), tyz = lsr::dmul(zd, p.m[4]), txz = lsr::dmul(zd, p.m[8]);
#pragma unroll
      for (int j = 0; j < P; ++j) {
        const int g = u * P + j;
        // scipy's order: ((zo*m0 + yo*m1) + xo*m2) + shift
        if constexpr (DEP & 1) az[g] = axis_tap<GRID, INTERIOR>(lsr::dadd(lsr::dadd(lsr::dadd(tzz, tzy[j]), tzx[j]), p.m[3]), Zl);
        else az[g] = hz[j];
        if constexpr (DEP & 2) ay[g] = axis_tap<GRID, INTERIOR>(lsr::dadd(lsr::dadd(lsr::dadd(tyz, tyy[j]), tyx[j]), p.m[7]), Yl);
        else ay[g] = hy[j];
        if constexpr (DEP & 4) ax[g] = axis_tap<GRID, INTERIOR>(lsr::dadd(lsr::dadd(lsr::dadd(txz, txy[j]), txx[j]), p.m[11]), Xl);
        else ax[g] = hx[j];
        inside[g] = (GRID || INTERIOR) ? true
                         : static_cast<bool>(static_cast<int>(live) & static_cast<int>(az[g].inside) &
                                             static_cast<int>(ay[g].inside) & static_cast<int>(ax[g].inside));
        // the box covers every inside voxel's taps by construction; the clamp keeps an out-of-range
        // voxel (whose result is dropped) inside the LDS image
        int o = (__mul24(az[g].i - zlo, plane_floats) + __mul24(ay[g].i - ylo, box_x) + (ax[g].i - xlo)) * 4;
        o = min(max(o, 0), o_max);
        v[g][0] = *reinterpret_cast<const f32x2*>(smem_b + o);
        v[g][1] = *reinterpret_cast<const f32x2*>(smem_b + o + row_b);
        v[g][2] = *reinterpret_cast<const f32x2*>(smem_b + o + plane_b);
        v[g][3] = *reinterpret_cast<const f32x2*>(smem_b + o + plane_b + row_b);
      }
    }
    // every LDS read of the group is issued before the first interpolation waits for one
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (GRID) {   // taps outside the volume carry cval (a blind block: all of them, whatever LDS holds)
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const unsigned zf = live ? az[g].out : 3u, yf = ay[g].out, xf = ax[g].out;
        const bool p0 = zf & 1u, p1 = zf & 2u, r0 = yf & 1u, r1 = yf & 2u, c0 = xf & 1u, c1 = xf & 2u;
        v[g][0].x = (p0 || r0 || c0) ? p.cval : v[g][0].x;
        v[g][0].y = (p0 || r0 || c1) ? p.cval : v[g][0].y;
        v[g][1].x = (p0 || r1 || c0) ? p.cval : v[g][1].x;
        v[g][1].y = (p0 || r1 || c1) ? p.cval : v[g][1].y;
        v[g][2].x = (p1 || r0 || c0) ? p.cval : v[g][2].x;
        v[g][2].y = (p1 || r0 || c1) ? p.cval : v[g][2].y;
        v[g][3].x = (p1 || r1 || c0) ? p.cval : v[g][3].x;
        v[g][3].y = (p1 || r1 || c1) ? p.cval : v[g][3].y;
      }
    }
    float res[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float result;
      if constexpr (F32) {
        const float gx = static_cast<float>(ax[g].f), gy = static_cast<float>(ay[g].f), gz = static_cast<float>(az[g].f);
        const float a0 = fmaf(gx, v[g][0].y - v[g][0].x, v[g][0].x), a1 = fmaf(gx, v[g][1].y - v[g][1].x, v[g][1].x);
        const float b0 = fmaf(gx, v[g][2].y - v[g][2].x, v[g][2].x), b1 = fmaf(gx, v[g][3].y - v[g][3].x, v[g][3].x);
        const float c0 = fmaf(gy, a1 - a0, a0), c1 = fmaf(gy, b1 - b0, b0);
        result = fmaf(gz, c1 - c0, c0);
      } else {
        const double wz0 = 1.0 - az[g].f, wy0 = 1.0 - ay[g].f, wx0 = 1.0 - ax[g].f;
        const double wz1 = 1.0 - wz0, wy1 = 1.0 - wy0, wx1 = 1.0 - wx0;
        // scipy's corner order and product order: ((v * wz) * wy) * wx, summed in sequence
        double t = 0.0;
        auto corner = [&](float val, double wz, double wy, double wx) {
          t = lsr::dadd(t, lsr::dmul(lsr::dmul(lsr::dmul(static_cast<double>(val), wz), wy), wx));
        };
        corner(v[g][0].x, wz0, wy0, wx0);
        corner(v[g][0].y, wz0, wy0, wx1);
        corner(v[g][1].x, wz0, wy1, wx0);
        corner(v[g][1].y, wz0, wy1, wx1);
        corner(v[g][2].x, wz1, wy0, wx0);
        corner(v[g][2].y, wz1, wy0, wx1);
        corner(v[g][3].x, wz1, wy1, wx0);
        corner(v[g][3].y, wz1, wy1, wx1);
        result = static_cast<float>(t);
      }
      res[g] = inside[g] ? result : p.cval;
      // (an opaque use: the value exists whether or not the store below runs, so hipcc cannot sink
      // the whole voxel into the store's condition and serialise the group again)
      asm volatile("" : "+v"(res[g]));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int zo = z0 + dz + u;
      if (zo >= p.Zo) break;   // uniform
      float* const oplane = p.out + static_cast<int64_t>(zo) * p.oplane;   // scalar
#pragma unroll
      for (int j = 0; j < P; ++j)
        if (ok[j]) oplane[pix[j]] = res[u * P + j];
    }
  }
}

#ifdef LSR_BOX_PROBES
#define LSR_BOX_PROBE_VALUE(p) ((p).probe)
#else
#define LSR_BOX_PROBE_VALUE(p) 0
#endif

// One block per workgroup.  (Round 4: letting the walk along zo start as soon as the planes of its first output
// planes had landed -- the image is plane-major and LDS-DMA loads return in order, so a counted s_waitcnt plus a
// barrier per step is enough -- ran 3-8 % SLOWER, 3.49 against 3.30 ms exact and 2.41 against 2.33 ms f32 on the
// 1.5 deg tilt: the per-step waits also wait for the previous step's stores, and the extra barriers cost more than
// the earlier start buys.)  Up to four such workgroups share a CU (boxes under 39 KB): the others compute while
// one's box is in flight.  (A persistent, double-buffered form of the same walk
// -- one workgroup per CU, the DMA of block n + 1 issued before the arithmetic of block n -- was
// measured in round 2 at 512 and at 1024 threads and ran 8-30 % SLOWER: 3.97 / 3.09 ms and 3.83 / 2.86 ms
// against 3.30 / 2.30 ms (exact / f32, 1.5 deg tilt, config-3 size).  The arithmetic wants more
// resident waves than one workgroup brings)
template <bool F32, int TZ, int DEP, bool GRID, int NT>
__global__ __launch_bounds__(NT, 4) void affine_box_kernel(BoxArgs p) {
  extern __shared__ f32x4 smem4[];
  const float* const smem = reinterpret_cast<const float*>(smem4);
  const unsigned lds_base =
      static_cast<unsigned>(reinterpret_cast<size_t>((__attribute__((address_space(3))) char*)smem4));
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int probe = LSR_BOX_PROBE_VALUE(p);
  // workgroups b, b + 8, ... share an XCD (round-robin dispatch); every XCD gets a contiguous run of
  // the patch-major block order
  const int bid = blockIdx.x;
  const Blk b = locate<TZ, GRID>(p, p.linear ? bid : (bid & 7) * p.per_xcd + (bid >> 3));
  if (!b.valid) return;
  if (!b.blind && probe != 1) stage<NT>(p, b, lds_base, tid, wave);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  compute<F32, TZ, DEP, NT, GRID>(p, b, smem, tid, probe);
}

struct BoxShape {
  int tz, ty, tx;
  int bz, by, bx;
  int64_t lds_bytes;
};

// box of a tz x ty x tx block under M: span of floor() over the block + the upper neighbour
// [+ 3 + 3 in x: 16-byte alignment of the first column, rows rounded up to whole chunks]
bool box_of(const double M[12], int tz, int ty, int tx, BoxShape* s) {
  auto ab = [](double v) { return v < 0 ? -v : v; };
  const double e[3] = {ab(M[0]) * (tz - 1) + ab(M[1]) * (ty - 1) + ab(M[2]) * (tx - 1),
                       ab(M[4]) * (tz - 1) + ab(M[5]) * (ty - 1) + ab(M[6]) * (tx - 1),
                       ab(M[8]) * (tz - 1) + ab(M[9]) * (ty - 1) + ab(M[10]) * (tx - 1)};
  for (double v : e)
    if (!(v < 2048.0)) return false;
  // floor(cmax) - floor(cmin) <= floor(e) + 1 (e is summed here in another order than on the
  // device: 1e-6 absorbs that), + 1 for the upper neighbour, + 1 for the count
  s->tz = tz; s->ty = ty; s->tx = tx;
  s->bz = static_cast<int>(e[0] + 1e-6) + 3;
  s->by = static_cast<int>(e[1] + 1e-6) + 3;
  s->bx = (static_cast<int>(e[2] + 1e-6) + 3 + 3 + 3) & ~3;
  const int64_t floats = int64_t(s->bz) * s->by * s->bx;
  s->lds_bytes = ((floats + 255) & ~int64_t(255)) * 4;
  return true;
}

// Block shape and box of the box path for this matrix and moving volume; false = not applicable.
bool pick_shape(int64_t Zi, int64_t Yi, int64_t Xi, int64_t pitch, int64_t plane, const double M[12], BoxShape* best) {
  if (!lsr::volume_in_range(Zi, Yi, Xi) || !lsr::strides_in_range(pitch, plane)) return false;
  // rows start on 16-byte boundaries (LDS-DMA moves 16-byte chunks) and hold whole chunks up to the last column
  if (pitch % 4 != 0 || plane % 4 != 0 || pitch < ((Xi + 3) & ~int64_t(3)) || Xi < 8 || Yi < 2 || Zi < 2) return false;
  if (plane >= (int64_t(1) << 32)) return false;
  // 4096 voxels, tx >= 32 (stores stay 128-byte runs), tz in {8, 16} (the compiled walks)
  static const int shapes[][3] = {{8, 8, 64}, {16, 8, 32}, {8, 16, 32}, {16, 4, 64}, {8, 4, 128}, {16, 2, 128}};
  // workgroups per CU: four by registers (256 threads on up to 128 VGPRs), by LDS as many boxes as fit into
  // 156 KB -- more resident boxes first, then the smaller box
  auto resident = [](int64_t lds) { return static_cast<int>(std::min<int64_t>(4, (156 * 1024) / lds)); };
  bool found = false;
  for (const auto& sh : shapes) {
    BoxShape s;
    if (!box_of(M, sh[0], sh[1], sh[2], &s)) continue;
    if (s.lds_bytes > 150 * 1024 || lsr::ceil_div(s.lds_bytes / 16, kThreads) > kMaxLoads) continue;
    if (int64_t(s.bz) * plane * 4 >= (int64_t(1) << 32)) continue;   // 32-bit byte offsets inside a box
    // (first listed wins a tie: pricing the 160-byte rows of tx = 32 shapes higher than their byte
    // count picked the longer rows at equal bytes and ran 12 % slower)
    const int r = resident(s.lds_bytes), best_r = found ? resident(best->lds_bytes) : 0;
    if (!found || r > best_r || (r == best_r && s.lds_bytes < best->lds_bytes)) {
      *best = s;
      found = true;
    }
  }
  return found;
}

// false: the device refuses the LDS budget (message in lsr_last_error()); the caller runs the gather kernel
template <bool F32, int TZ, int DEP, int NT, bool GRID = false>
bool launch_one(const BoxArgs& p, unsigned blocks, size_t lds_bytes, hipStream_t s) {
  static std::atomic<uint64_t> lds_allowed{0};
  auto kernel = affine_box_kernel<F32, TZ, DEP, GRID, NT>;
  if (lsr::allow_dynamic_lds(reinterpret_cast<const void*>(kernel), 150 * 1024, lds_allowed, "affine_box_kernel") != LSR_OK)
    return false;
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(NT), lds_bytes, s, p);
  return true;
}

template <bool F32, int TZ>
bool launch_shape(const BoxArgs& p, unsigned blocks, size_t lds_bytes, bool grid, hipStream_t s) {
  // the compiled walks: everything depends on zo / y_in does not (tilt about y) / x_in does not
  // (tilt about x); other patterns run the general walk (their zo * 0 terms are exact zeros).
  const bool dz = p.m[0] != 0.0, dy = p.m[4] != 0.0, dx = p.m[8] != 0.0;
  if (grid) {
    if (dz && !dy && dx) return launch_one<F32, TZ, 5, kThreads, true>(p, blocks, lds_bytes, s);
    if (dz && dy && !dx) return launch_one<F32, TZ, 3, kThreads, true>(p, blocks, lds_bytes, s);
    return launch_one<F32, TZ, 7, kThreads, true>(p, blocks, lds_bytes, s);
  }
  if (dz && !dy && dx) return launch_one<F32, TZ, 5, kThreads>(p, blocks, lds_bytes, s);
  if (dz && dy && !dx) return launch_one<F32, TZ, 3, kThreads>(p, blocks, lds_bytes, s);
  return launch_one<F32, TZ, 7, kThreads>(p, blocks, lds_bytes, s);
}

}  // namespace

namespace lsr {

bool affine_box_geometry(int64_t Zi, int64_t Yi, int64_t Xi, int64_t pitch, int64_t plane, const double M[12],
                         int* box_z, int* box_y, int* box_x, int64_t* lds_bytes) {
  BoxShape s;
  if (!pick_shape(Zi, Yi, Xi, pitch, plane, M, &s)) return false;
  *box_z = s.bz; *box_y = s.by; *box_x = s.bx; *lds_bytes = s.lds_bytes;
  return true;
}

bool affine_box_shape(int64_t Zi, int64_t Yi, int64_t Xi, const double M[12], int out6[6]) {
  BoxShape s;
  if (!volume_in_range(Zi, Yi, Xi)) return false;
  if (!pick_shape(Zi, Yi, Xi, Xi, Yi * Xi, M, &s)) return false;
  out6[0] = s.tz; out6[1] = s.ty; out6[2] = s.tx; out6[3] = s.bz; out6[4] = s.by; out6[5] = s.bx;
  return true;
}

bool launch_affine_box(const float* in, int64_t Zi, int64_t Yi, int64_t Xi, int64_t pitch, int64_t plane, float* out,
                       int64_t Zo, int64_t Yo, int64_t Xo, int64_t opitch, int64_t oplane, const double M[12], float cval,
                       bool f32, bool grid, hipStream_t s) {
  BoxShape sh;
  if ((reinterpret_cast<uintptr_t>(in) & 15) != 0 || Yo * opitch >= (int64_t(1) << 31)) return false;
  if (!pick_shape(Zi, Yi, Xi, pitch, plane, M, &sh)) return false;
  BoxArgs p;
  p.in = in; p.out = out;
  p.Zi = static_cast<int>(Zi); p.Yi = static_cast<int>(Yi); p.Xi = static_cast<int>(Xi);
  p.pitch = static_cast<unsigned>(pitch); p.plane = static_cast<unsigned>(plane);
  p.opitch = static_cast<unsigned>(opitch); p.oplane = oplane;
  p.Zo = static_cast<int>(Zo); p.Yo = static_cast<int>(Yo); p.Xo = static_cast<int>(Xo);
  for (int i = 0; i < 12; ++i) p.m[i] = M[i];
  p.cval = cval;
  p.ty = sh.ty; p.tx = sh.tx;
  p.tx_shift = sh.tx == 32 ? 5 : (sh.tx == 64 ? 6 : 7);
  p.box_z = sh.bz; p.box_y = sh.by; p.box_x = sh.bx;
  p.inv_cx = 1.0f / static_cast<float>(sh.bx / 4);
  p.inv_by = 1.0f / static_cast<float>(sh.by);
  p.tz_n = static_cast<int>(ceil_div(Zo, sh.tz));
  p.ty_n = static_cast<int>(ceil_div(Yo, sh.ty));
  p.tx_n = static_cast<int>(ceil_div(Xo, sh.tx));
  p.pz_n = static_cast<int>(ceil_div(p.tz_n, kPatch));
  p.py_n = static_cast<int>(ceil_div(p.ty_n, kPatch));
  p.px_n = static_cast<int>(ceil_div(p.tx_n, kPatch));
  const int64_t padded = int64_t(p.pz_n) * p.py_n * p.px_n * (kPatch * kPatch * kPatch);
  const int64_t per_xcd = ceil_div(padded, 8);
  if (per_xcd * 8 >= (int64_t(1) << 31)) return false;
  p.per_xcd = static_cast<int>(per_xcd);
  // f32 mode (HBM / LDS-DMA bound): the plain patch-major order, whose consecutive blocks go to the eight
  // XCDs in turn, is as fast or faster (2.52 against 2.71 ms on config 3 o tilt 3 deg); the exact mode keeps
  // the per-XCD runs (their L2 hits: 1.10x the source fetched).  LSR_BOX_LINEAR=1 / 0 forces either.
  p.linear = f32 ? 1 : 0;
  if (const char* e = std::getenv("LSR_BOX_LINEAR")) p.linear = e[0] != '0';
  // (blocks past `padded` decode to a patch index >= the patch count: px >= px_n -> bx >= tx_n -> exit)
  p.probe = 0;
#ifdef LSR_BOX_PROBES
  {
    const char* pe = getenv("LSR_BOX_PROBE");
    p.probe = pe ? atoi(pe) : 0;
    if (getenv("LSR_BOX_VERBOSE"))
      fprintf(stderr, "affine_box: block %dx%dx%d box %dx%dx%d (%lld bytes)\n", sh.tz, sh.ty, sh.tx, sh.bz, sh.by,
              sh.bx, (long long)sh.lds_bytes);
  }
#endif
  const unsigned blocks = static_cast<unsigned>(per_xcd * 8);
  const size_t lds = static_cast<size_t>(sh.lds_bytes);
  if (Yo * opitch >= (int64_t(1) << 31)) return false;   // 32-bit output offsets inside a plane
  switch (sh.tz * 2 + (f32 ? 1 : 0)) {
    case 8 * 2 + 1: return launch_shape<true, 8>(p, blocks, lds, grid, s);
    case 8 * 2 + 0: return launch_shape<false, 8>(p, blocks, lds, grid, s);
    case 16 * 2 + 1: return launch_shape<true, 16>(p, blocks, lds, grid, s);
    default: return launch_shape<false, 16>(p, blocks, lds, grid, s);
  }
}

}  // namespace lsr
