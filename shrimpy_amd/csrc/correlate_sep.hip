// Separable (rank-1) 3-D PSF correlation with fused Richardson-Lucy epilogues -- the HBM-bound
// headline kernel.  See correlate.hip for the streaming idea; this file is the tuned form with
// compile-time tap counts.
//
// Work decomposition (256 threads = 4 waves, tile 32 (y) x 64 (x), marching along z):
//   per input plane
//     stage   : the plane's (32+PY-1) x (64+PX-1) window, HBM -> registers -> LDS `A`.  The
//               registers are loaded one plane AHEAD (issued right after the barrier, consumed
//               at the top of the next iteration), so HBM latency hides behind a whole plane of
//               compute.  Rows are 256-B coalesced; the PX-1 halo columns go in one extra load.
//     x pass  : A -> B, 4 outputs per thread from a (4+PX-1)-float register window read with
//               ds_read_b64 (A pitch = 2 mod 4 floats: conflict-free at 16-B lane stride).
//     y pass  : B -> registers, 8 outputs per thread down one column (conflict-free ds_read_b32,
//               8+PY-1 reads for 8 outputs).
//     z       : PZ pending output planes per point live in registers; one FMA per pending plane
//               both shifts the window and adds this plane's contribution.
//     epilogue: the completed plane: ratio = y * rcp(c + eps)   or   x * c * rcp(H^T 1).
//               `aux` is loaded at the top of the iteration, BEFORE the prefetch, so waiting for
//               it leaves the prefetch in flight (vmcnt counts in order).
//   two workgroup barriers per plane; LDS 20 KB per workgroup.
//
// Algorithmic HBM bytes: 12 per voxel per launch.  Real traffic adds the in-plane halo
// ((38*70)/(32*64) = 1.30x on the `in` stream only, mostly L2/MALL hits thanks to the XCD-aware
// tile order) and PZ-1 planes per z-chunk.

#include "common.hpp"
#include "correlate_common.hpp"

namespace {

using lsr::CorrArgs;

constexpr int kThreads = 256;
constexpr int kTY = 32;
constexpr int kTX = 64;
constexpr int kRun = 8;  // consecutive y per thread in the y/z passes

template <int PY, int PX>
struct Tile {
  static constexpr int AR = kTY + PY - 1;                      // staged rows
  static constexpr int AC = kTX + PX - 1;                      // staged cols (even: PX odd)
  static constexpr int PA = (AC % 4 == 2) ? AC : AC + 2;       // pitch = 2 (mod 4) floats
  static constexpr int PB = kTX;
  static constexpr int TAIL = PX - 1;                          // halo columns beyond 64 (>= 2)
  static constexpr int TAIL_LOADS = (AR * TAIL + kThreads - 1) / kThreads;
  static constexpr int MAIN_LOADS = (AR + 3) / 4;              // rows per wave
  static constexpr int XITEMS = AR * (kTX / 4);                // (row, 4-wide x group) items
  static constexpr int XITERS = (XITEMS + kThreads - 1) / kThreads;
  static constexpr int WIN = 4 + PX - 1;                       // x-pass register window (even)
};

__device__ __forceinline__ float fast_rcp(float d) {
  // v_rcp_f32 (1 ulp) + one Newton step
  float r = __builtin_amdgcn_rcpf(d);
  return fmaf(fmaf(-d, r, 1.0f), r, r);
}

template <int PZ, int PY, int PX>
__global__ __launch_bounds__(kThreads) void correlate_sep_kernel(CorrArgs p) {
  using T = Tile<PY, PX>;
  __shared__ __attribute__((aligned(16))) float bufA[T::AR * T::PA];
  __shared__ __attribute__((aligned(16))) float bufB[T::AR * T::PB];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  // wave index as a SCALAR: everything derived from it (row predicates, row base pointers)
  // then lives in SGPRs / SALU instead of per-lane VGPRs
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- XCD-aware tile order: workgroups b, b+8, ... share an XCD (round-robin dispatch), so
  // give each XCD a contiguous run of tiles: x-neighbours then share halo columns in one L2.
  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x;
    const int per = nblk / 8, rem = nblk % 8;
    const int xcd = bid % 8, idx = bid / 8;
    bid = xcd * per + (xcd < rem ? xcd : rem) + idx;  // XCD k owns per + (k < rem) tiles
  }
  const int tiles_x = static_cast<int>(p.tiles_x), tiles_y = static_cast<int>(p.tiles_y);
  const int tx = bid % tiles_x;
  const int ty = (bid / tiles_x) % tiles_y;
  const int zc = bid / (tiles_x * tiles_y);

  const int Z = static_cast<int>(p.Z), Y = static_cast<int>(p.Y), X = static_cast<int>(p.X);
  const int x0 = tx * kTX;
  const int y0 = ty * kTY;
  const int zb = zc * static_cast<int>(p.z_chunk);
  const int ze = min(zb + static_cast<int>(p.z_chunk), Z);
  constexpr int cz = PZ / 2, cy = PY / 2, cx = PX / 2;
  const int64_t plane = static_cast<int64_t>(Y) * X;

  // ---- taps: wave-uniform, kept in scalar registers
  // (the caller's pz/py/px taps are centred inside the compiled PZ/PY/PX, zero elsewhere)
  float wz[PZ], wy[PY], wx[PX];
  const int oz = (PZ - p.pz) / 2, oy = (PY - p.py) / 2, ox = (PX - p.px) / 2;
#pragma unroll
  for (int i = 0; i < PZ; ++i) wz[i] = (i >= oz && i < oz + p.pz) ? p.wz[i - oz] : 0.0f;
#pragma unroll
  for (int i = 0; i < PY; ++i) wy[i] = (i >= oy && i < oy + p.py) ? p.wy[i - oy] : 0.0f;
#pragma unroll
  for (int i = 0; i < PX; ++i) wx[i] = (i >= ox && i < ox + p.px) ? p.wx[i - ox] : 0.0f;

  // ---- staging geometry (constant over z).  Main part: wave w stages rows w, w+4, ...; lane =
  // column.  Tail: the PX-1 columns beyond 64, one element per thread.
  const int row0 = y0 - cy;  // global y of staged row 0
  const int gx_main = x0 - cx + lane;
  const bool xok_main = gx_main >= 0 && gx_main < X;
  const int voff_main = xok_main ? gx_main : 0;  // per-lane part of the address (floats)
  // tail element e = tid + 256*t -> (row e / TAIL, column 64 + e % TAIL)
  int voff_tail[T::TAIL_LOADS];
  int lds_tail[T::TAIL_LOADS];  // < 0: not this thread's element
#pragma unroll
  for (int t = 0; t < T::TAIL_LOADS; ++t) {
    const int e = tid + t * kThreads;
    const int r = e / T::TAIL, c = kTX + e % T::TAIL;
    const int gy = row0 + r, gx = x0 - cx + c;
    const bool mine = r < T::AR;
    const bool ok = mine && gy >= 0 && gy < Y && gx >= 0 && gx < X;
    lds_tail[t] = mine ? r * T::PA + c : -1;
    voff_tail[t] = ok ? gy * X + gx : -1;
  }

  float stage[T::MAIN_LOADS];
  float stage_tail[T::TAIL_LOADS];

  auto prefetch = [&](int zi) {
    const float* src = p.in + zi * plane;
#pragma unroll
    for (int i = 0; i < T::MAIN_LOADS; ++i) {
      const int r = wave + 4 * i;     // scalar
      const int gy = row0 + r;        // scalar
      float v = 0.0f;
      if (r < T::AR && gy >= 0 && gy < Y) {            // wave-uniform branch
        const float* rowp = src + static_cast<int64_t>(gy) * X;  // scalar base
        if (xok_main) v = rowp[voff_main];
      }
      stage[i] = v;
    }
#pragma unroll
    for (int t = 0; t < T::TAIL_LOADS; ++t)
      stage_tail[t] = voff_tail[t] >= 0 ? src[voff_tail[t]] : 0.0f;
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < T::MAIN_LOADS; ++i) {
      const int r = wave + 4 * i;
      if (r < T::AR) bufA[r * T::PA + lane] = stage[i];
    }
#pragma unroll
    for (int t = 0; t < T::TAIL_LOADS; ++t)
      if (lds_tail[t] >= 0) bufA[lds_tail[t]] = stage_tail[t];
  };

  // ---- output geometry: thread owns column `lane`, rows wave*8 .. wave*8+7 (row part scalar)
  const int gx_out = x0 + lane;
  const bool xok_out = gx_out < X;
  const int voff_out = xok_out ? gx_out : 0;
  const int gy_out0 = y0 + wave * kRun;                     // scalar
  const int nrows_out = min(max(Y - gy_out0, 0), kRun);     // scalar: valid rows of this wave
  const int epi = p.epilogue;

  // reciprocal of the in-plane part of H^T 1 for the UPDATE epilogue
  float rnyx[kRun];
#pragma unroll
  for (int m = 0; m < kRun; ++m) rnyx[m] = 0.0f;
  if (epi == LSR_EPI_UPDATE && xok_out) {
    const float nxv = p.nx[gx_out];
#pragma unroll
    for (int m = 0; m < kRun; ++m)
      if (m < nrows_out) rnyx[m] = fast_rcp(p.ny[gy_out0 + m] * nxv);
  }

  // pending output planes: acc[j][m] <-> z_out = zi - cz + j once plane zi is absorbed
  float acc[PZ][kRun];
#pragma unroll
  for (int j = 0; j < PZ; ++j)
#pragma unroll
    for (int m = 0; m < kRun; ++m) acc[j][m] = 0.0f;

  // LDS offsets of this thread's first x-pass item and of its y-pass column
  const int xitem_a = (tid >> 4) * T::PA + 4 * (tid & 15);
  const int xitem_b = (tid >> 4) * T::PB + 4 * (tid & 15);
  const int ycol = (wave * kRun) * T::PB + lane;

  const int zi_begin = max(zb - cz, 0);
  const int zi_end = ze + cz;  // exclusive; planes >= Z contribute zeros
  if (zi_begin < Z) prefetch(zi_begin);

  for (int zi = zi_begin; zi < zi_end; ++zi) {
    const bool have_plane = zi < Z;
    const int z_out = zi - cz;
    const bool emit = z_out >= zb;

    float pl[kRun];
#pragma unroll
    for (int m = 0; m < kRun; ++m) pl[m] = 0.0f;
    float aux[kRun];
#pragma unroll
    for (int m = 0; m < kRun; ++m) aux[m] = 0.0f;

    if (have_plane) {
      commit();          // plane zi: registers -> A (all waves left the x pass of zi-1: barrier 2)
      __syncthreads();   // barrier 1: A complete; every wave is done with B of plane zi-1
    }
    // aux first, then the prefetch: waiting for aux later leaves the prefetch in flight
    if (emit && epi != LSR_EPI_NONE && xok_out) {
      const float* a = p.aux + z_out * plane + static_cast<int64_t>(gy_out0) * X;  // scalar
#pragma unroll
      for (int m = 0; m < kRun; ++m)
        if (m < nrows_out) aux[m] = (a + m * X)[voff_out];
    }
    if (zi + 1 < Z && zi + 1 < zi_end) prefetch(zi + 1);

    if (have_plane) {
      // ---- x pass: item = (row, group of 4 x); 16 items per row, 4 rows per wave-instruction
#pragma unroll 1
      for (int it = 0; it < T::XITERS; ++it) {
        if (tid + it * kThreads < T::XITEMS) {
          const float2* src =
              reinterpret_cast<const float2*>(bufA + xitem_a + it * (kThreads / 16) * T::PA);
          float win[T::WIN];
#pragma unroll
          for (int i = 0; i < T::WIN / 2; ++i) {
            const float2 v = src[i];
            win[2 * i] = v.x;
            win[2 * i + 1] = v.y;
          }
          float4 o;
          o.x = wx[0] * win[0];
          o.y = wx[0] * win[1];
          o.z = wx[0] * win[2];
          o.w = wx[0] * win[3];
#pragma unroll
          for (int c = 1; c < PX; ++c) {
            o.x = fmaf(wx[c], win[c], o.x);
            o.y = fmaf(wx[c], win[c + 1], o.y);
            o.z = fmaf(wx[c], win[c + 2], o.z);
            o.w = fmaf(wx[c], win[c + 3], o.w);
          }
          *reinterpret_cast<float4*>(bufB + xitem_b + it * (kThreads / 16) * T::PB) = o;
        }
      }
      __syncthreads();  // barrier 2: B complete; A free for the next commit

      // ---- y pass: 8 outputs down one column from 8+PY-1 reads
      const float* col = bufB + ycol;
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

    // ---- z: shift the pending planes and absorb this plane in the same FMA
#pragma unroll
    for (int j = 0; j < PZ - 1; ++j)
#pragma unroll
      for (int m = 0; m < kRun; ++m) acc[j][m] = fmaf(wz[PZ - 1 - j], pl[m], acc[j + 1][m]);
#pragma unroll
    for (int m = 0; m < kRun; ++m) acc[PZ - 1][m] = wz[0] * pl[m];

    // ---- epilogue for the completed plane z_out
    if (emit && xok_out) {
      float* o = p.out + z_out * plane + static_cast<int64_t>(gy_out0) * X;  // scalar
      if (epi == LSR_EPI_RATIO) {
#pragma unroll
        for (int m = 0; m < kRun; ++m)
          if (m < nrows_out) (o + m * X)[voff_out] = aux[m] * fast_rcp(acc[0][m] + p.eps);
      } else if (epi == LSR_EPI_UPDATE) {
        const float rz = fast_rcp(p.nz[z_out]);
#pragma unroll
        for (int m = 0; m < kRun; ++m)
          if (m < nrows_out) (o + m * X)[voff_out] = aux[m] * acc[0][m] * (rz * rnyx[m]);
      } else {
#pragma unroll
        for (int m = 0; m < kRun; ++m)
          if (m < nrows_out) (o + m * X)[voff_out] = acc[0][m];
      }
    }
  }
}

// Tap counts with a compiled specialisation; other (odd) sizes are zero-padded up to the next
// one by the dispatcher, which keeps the centre tap in place.
constexpr int kSizes[] = {3, 5, 7, 9, 11, 13, 15};

int round_up_taps(int n) {
  for (int s : kSizes)
    if (n <= s) return s;
  return -1;
}

template <int PZ, int PYX>
void launch_one(const CorrArgs& p, dim3 grid, hipStream_t s) {
  hipLaunchKernelGGL((correlate_sep_kernel<PZ, PYX, PYX>), grid, dim3(kThreads), 0, s, p);
}

template <int PZ>
bool launch_pz(int pyx, const CorrArgs& p, dim3 grid, hipStream_t s) {
  switch (pyx) {
    case 3: launch_one<PZ, 3>(p, grid, s); return true;
    case 5: launch_one<PZ, 5>(p, grid, s); return true;
    case 7: launch_one<PZ, 7>(p, grid, s); return true;
    case 9: launch_one<PZ, 9>(p, grid, s); return true;
    case 11: launch_one<PZ, 11>(p, grid, s); return true;
    case 13: launch_one<PZ, 13>(p, grid, s); return true;
    case 15: launch_one<PZ, 15>(p, grid, s); return true;
    default: return false;
  }
}

}  // namespace

namespace lsr {

bool sep_fast_supported(int pz, int py, int px, int* PZ, int* PYX) {
  const int a = round_up_taps(pz);
  const int b = round_up_taps(py > px ? py : px);
  if (a < 0 || b < 0) return false;
  *PZ = a;
  *PYX = b;
  return true;
}

int launch_sep_fast(const CorrArgs& p, int PZ, int PYX, hipStream_t s) {
  const int64_t blocks = p.tiles_x * p.tiles_y * ceil_div(p.Z, p.z_chunk);
  if (blocks >= (int64_t(1) << 31))
    return fail(LSR_E_SHAPE, "grid of %lld workgroups is too large", (long long)blocks);
  if (p.Y * p.X >= (int64_t(1) << 30))
    return fail(LSR_E_UNSUPPORTED, "plane of %lld voxels exceeds the 32-bit in-plane offsets",
                (long long)(p.Y * p.X));
  const dim3 grid(static_cast<unsigned>(blocks));
  bool ok = false;
  switch (PZ) {
    case 3: ok = launch_pz<3>(PYX, p, grid, s); break;
    case 5: ok = launch_pz<5>(PYX, p, grid, s); break;
    case 7: ok = launch_pz<7>(PYX, p, grid, s); break;
    case 9: ok = launch_pz<9>(PYX, p, grid, s); break;
    case 11: ok = launch_pz<11>(PYX, p, grid, s); break;
    case 13: ok = launch_pz<13>(PYX, p, grid, s); break;
    case 15: ok = launch_pz<15>(PYX, p, grid, s); break;
    default: break;
  }
  if (!ok) return fail(LSR_E_UNSUPPORTED, "no specialisation for taps (%d, %d)", PZ, PYX);
  return launch_status("lsr_correlate_sep_f32");
}

}  // namespace lsr
