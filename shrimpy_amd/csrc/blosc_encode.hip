// blosc-zstd frames written ON the MI355X: the result of the hot path leaves HBM already in the format the
// acquisition engine writes (blosc, zstd, byte shuffle: shrimpy/mantis/mantis_engine.py:474-481; asserted in
// shrimpy/tests/test_mantis_integration.py:177-190), so the download and the writer move compressed bytes and the
// host only pwrite()s them.  Round 4 measured the host alternative: zstd level 1 over a 1.75 GB float32 result is
// ~6 core-seconds, 0.38 s per config-4 unit on a rank's 16 cores against 30 ms of kernels.
//
// One c-blosc 1.x frame per Zarr chunk (a z-range of whole planes: contiguous bytes of the volume), blocks of
// `blocksize` bytes, one zstd frame per block (blocks are not split, c-blosc's own rule for zstd).  Inside a block the
// byte shuffle yields `typesize` planes; every plane becomes one zstd block: RLE (one value), Compressed (Huffman
// literals in four streams, no sequences) when that saves 1/64 or more, else Raw (csrc/zstd_huf.hpp).
//
// Three launches per volume (any number of frames):
//   encode_blocks   one workgroup (4 waves) per blosc block: histogram of every plane in one coalesced pass, the code
//                   tables built side by side (one wave per plane), then plane by plane: the plane's bytes staged in LDS,
//                   wave j writes stream j -- each lane a contiguous run of symbols, bit offsets by a wave suffix sum,
//                   whole dwords stored at byte addresses, the < 8 bits at a lane boundary handed over by a shuffle;
//   scan_frames     block sizes -> bstarts and the frame sizes / offsets;
//   gather_frames   block streams -> their place in the frame (a byte-shifted copy), headers.
// The host twin (lsr_blosc_encode_device_cpu) runs the same steps in loops: the same bytes (tests/test_device_codec.py).

#include <algorithm>
#include <cstring>
#include <vector>

#include "common.hpp"
#include "host_parallel.hpp"
#include "zstd_huf.hpp"

namespace {

using namespace lsr::zs;

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kMaxType = 4;                 // planes per block (typesize): 1, 2 or 4 -- one wave builds each code table
constexpr int kMaxDevPlane = 64 * 1024;     // symbols of one plane the encode kernel stages in LDS
constexpr int kBlockSlack = 64;             // a block's stream is at most blocksize + 4 + 9 + 4 * 3 + a few bytes
constexpr int kLanePad = 12;                // LDS bytes between lane runs: room for a merged tail (< 8), odd dword stride

struct Geometry {
  int64_t src_bytes, frame_bytes, blocksize;
  int typesize;
  int64_t n_frames, blocks_per_frame, n_blocks, block_stride;
};

inline int geometry(Geometry& g, int64_t src_bytes, int typesize, int64_t frame_bytes, int64_t blocksize) {
  LSR_REQUIRE(typesize == 1 || typesize == 2 || typesize == 4, LSR_E_UNSUPPORTED,
              "typesize %d: the device encoder shuffles 1-, 2- and 4-byte elements", typesize);
  LSR_REQUIRE(src_bytes > 0 && src_bytes < lsr::kMaxVoxels, LSR_E_ARG, "source of %lld bytes", (long long)src_bytes);
  LSR_REQUIRE(frame_bytes > 0 && frame_bytes <= 0x7FFFFFFF - 16 && frame_bytes % typesize == 0, LSR_E_ARG,
              "a blosc 1.x frame holds less than 2 GiB of whole elements, got %lld bytes", (long long)frame_bytes);
  LSR_REQUIRE(src_bytes % typesize == 0, LSR_E_ARG, "source of %lld bytes is not whole %d-byte elements",
              (long long)src_bytes, typesize);
  if (blocksize <= 0) blocksize = int64_t(typesize) * kMaxDevPlane;
  blocksize = std::min(blocksize, frame_bytes);
  LSR_REQUIRE(blocksize % typesize == 0 && blocksize / typesize <= kMaxDevPlane && blocksize >= typesize, LSR_E_UNSUPPORTED,
              "blocksize %lld: whole elements, at most %d per plane", (long long)blocksize, kMaxDevPlane);
  g.src_bytes = src_bytes; g.frame_bytes = frame_bytes; g.blocksize = blocksize; g.typesize = typesize;
  g.n_frames = lsr::ceil_div(src_bytes, frame_bytes);
  g.blocks_per_frame = lsr::ceil_div(frame_bytes, blocksize);
  g.n_blocks = g.n_frames * g.blocks_per_frame;
  g.block_stride = (blocksize + kBlockSlack + 15) / 16 * 16;
  LSR_REQUIRE(g.n_blocks < (int64_t(1) << 31), LSR_E_UNSUPPORTED, "%lld blocks", (long long)g.n_blocks);
  return LSR_OK;
}

inline int64_t frame_cap(const Geometry& g) {   // bytes one frame can take, a multiple of 16
  return (16 + g.blocks_per_frame * (4 + g.blocksize + kBlockSlack) + 15) / 16 * 16;
}

// scratch: [block streams: n_blocks * block_stride][sizes: int32 n_blocks][bstarts: int32 n_blocks][frame sizes: int64 n_frames]
inline int64_t sizes_offset(const Geometry& g) { return g.n_blocks * g.block_stride; }
inline int64_t frame_sizes_offset(const Geometry& g) { return sizes_offset(g) + (g.n_blocks * 8 + 15) / 16 * 16; }
inline int64_t scratch_bytes_of(const Geometry& g) { return frame_sizes_offset(g) + g.n_frames * 8 + 64; }

LSR_HD int64_t imin64(int64_t a, int64_t b) { return a < b ? a : b; }
LSR_HD int64_t imax64(int64_t a, int64_t b) { return a > b ? a : b; }

LSR_HD int pow2_ceil(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

// The run of stream symbols one lane writes: `c` symbols (a power of two, >= 8); a short tail (< 8 symbols) of the last
// lane is merged into its neighbour, so every lane in use holds at least 8 symbols (>= 8 bits of code).
struct Runs {
  int c, shift, lanes, last_len;
};
LSR_HD Runs stream_runs(int m) {
  Runs r;
  r.c = pow2_ceil((m + 63) / 64);
  if (r.c < 8) r.c = 8;
  r.shift = 0;
  while ((1 << r.shift) < r.c) ++r.shift;
  r.lanes = (m + r.c - 1) / r.c;
  r.last_len = m - (r.lanes - 1) * r.c;
  if (r.last_len < 8 && r.lanes > 1) {
    --r.lanes;
    r.last_len += r.c;
  }
  return r;
}

inline void put32(uint8_t* q, uint32_t v) { q[0] = uint8_t(v); q[1] = uint8_t(v >> 8); q[2] = uint8_t(v >> 16); q[3] = uint8_t(v >> 24); }

// ---------------------------------------------------------------------------------------------------------------------
// host twin
// ---------------------------------------------------------------------------------------------------------------------

struct PlaneCode {
  int mode;                 // PlaneMode
  uint8_t rle_value;
  uint32_t ct[256];         // value | length << 16
  uint8_t desc[kHufHeaderMax];
  int desc_size;
};

// steps 1-8 of zstd_huf.hpp for one plane, in loops
void build_plane_code_host(const uint32_t* count, int plane_len, PlaneCode& pc) {
  int distinct = 0, only = 0;
  for (int s = 0; s < 256; ++s)
    if (count[s]) { ++distinct; only = s; }
  pc.desc_size = 0;
  if (distinct == 1) { pc.mode = kPlaneRle; pc.rle_value = static_cast<uint8_t>(only); return; }
  pc.mode = kPlaneRaw;
  if (plane_len < kMinHufPlane || distinct < 2) return;
  uint64_t sum_sq = 0;
  for (int s = 0; s < 256; ++s) sum_sq += uint64_t(count[s]) * count[s];
  if (huf_hopeless(sum_sq, static_cast<uint64_t>(plane_len))) return;
  uint16_t order[256];
  int ns = 0;
  for (int i = 0; i < 256; ++i) {
    const int r = huf_rank_of(count, i);
    if (r >= 0) { order[r] = static_cast<uint16_t>(i); ++ns; }
  }
  uint32_t node_cnt[512];
  uint16_t parent[512];
  for (int k = 0; k < ns; ++k) node_cnt[k] = count[order[k]];
  huf_merge(node_cnt, parent, ns);
  uint32_t per_depth[kDepthSlots] = {0};
  for (int k = 0; k < ns; ++k) ++per_depth[huf_depth_of(parent, ns, k)];
  huf_limit(per_depth);
  uint8_t nbits[256] = {0};
  for (int k = 0; k < ns; ++k) nbits[order[k]] = static_cast<uint8_t>(huf_length_of_rank(per_depth, k));
  uint16_t first[kHufMaxBits + 1];
  const int max_bits = huf_first_values(per_depth, first);
  int64_t payload_bits = 0;
  for (int i = 0; i < 256; ++i) {
    pc.ct[i] = huf_code_of(nbits, first, i);
    payload_bits += int64_t(count[i]) * nbits[i];
  }
  HufScratch scratch;
  pc.desc_size = huf_write_description(nbits, max_bits, pc.desc, scratch);
  if (pc.desc_size > 0 && huf_pays(plane_len, payload_bits, pc.desc_size)) pc.mode = kPlaneHuf;
}

// one stream: symbols last to first, LSB-first bit container, closing '1'
int encode_stream_host(const uint8_t* sym, int m, const uint32_t* ct, uint8_t* dst) {
  uint64_t acc = 0;
  int bits = 0, n = 0;
  for (int i = m - 1; i >= 0; --i) {
    const uint32_t e = ct[sym[i]];
    acc |= uint64_t(e & 0xFFFF) << bits;
    bits += static_cast<int>(e >> 16);
    while (bits >= 8) { dst[n++] = static_cast<uint8_t>(acc); acc >>= 8; bits -= 8; }
  }
  acc |= uint64_t(1) << bits;
  dst[n++] = static_cast<uint8_t>(acc);
  return n;
}

// One blosc block: `bsize` decoded bytes of which the first `valid` come from src (the rest are the zero padding of
// an edge chunk).  Writes [int32 cbytes][stream] to dst, returns the bytes written.
int64_t encode_block_host(const uint8_t* src, int64_t valid, int64_t bsize, int T, uint8_t* dst) {
  const int plane_len = static_cast<int>(bsize / T);
  std::vector<uint8_t> planes(static_cast<size_t>(bsize));
  for (int64_t i = 0; i < bsize; ++i) planes[(i % T) * plane_len + i / T] = i < valid ? src[i] : 0;
  std::vector<PlaneCode> pcs(static_cast<size_t>(T));
  bool any = false;
  const int per = 16 / T;                       // elements of one 16-byte vector of the block
  for (int p = 0; p < T; ++p) {
    if (plane_len >= kSampleMinPlane) {         // hopeless on the sample: raw, and never counted (huf_hopeless_sample)
      uint32_t sc[256] = {0};
      uint64_t n = 0, coll = 0;
      for (int i = 0; i < plane_len; ++i)
        if ((i / per) % kSampleEvery == 0) { ++sc[planes[size_t(p) * plane_len + i]]; ++n; }
      for (int v = 0; v < 256; ++v) coll += uint64_t(sc[v]) * (sc[v] ? sc[v] - 1 : 0);
      if (huf_hopeless_sample(coll, n)) { pcs[p].mode = kPlaneRaw; pcs[p].desc_size = 0; continue; }
    }
    uint32_t count[256] = {0};
    for (int i = 0; i < plane_len; ++i) ++count[planes[size_t(p) * plane_len + i]];
    build_plane_code_host(count, plane_len, pcs[p]);
    any = any || pcs[p].mode != kPlaneRaw;
  }
  if (!any) {                                   // nothing compresses: the shuffled block verbatim (cbytes == bsize)
    put32(dst, static_cast<uint32_t>(bsize));
    std::memcpy(dst + 4, planes.data(), static_cast<size_t>(bsize));
    return 4 + bsize;
  }
  int64_t pos = 4;
  frame_header(dst + pos, static_cast<uint32_t>(bsize));
  pos += kFrameHeader;
  for (int p = 0; p < T; ++p) {
    const uint8_t* sym = planes.data() + size_t(p) * plane_len;
    const PlaneCode& pc = pcs[p];
    const bool last = p == T - 1;
    if (pc.mode == kPlaneRle) {
      block_header(dst + pos, 1, plane_len, last);
      dst[pos + 3] = pc.rle_value;
      pos += 4;
    } else if (pc.mode == kPlaneRaw) {
      block_header(dst + pos, 0, plane_len, last);
      std::memcpy(dst + pos + 3, sym, static_cast<size_t>(plane_len));
      pos += 3 + plane_len;
    } else {
      std::vector<uint8_t> streams[4];
      int sizes[4], at = 0;
      for (int j = 0; j < 4; ++j) {
        const int m = huf_stream_len(plane_len, j);
        streams[j].resize(static_cast<size_t>(m) * 2 + 16);
        sizes[j] = encode_stream_host(sym + at, m, pc.ct, streams[j].data());
        at += m;
      }
      const int csize = pc.desc_size + 6 + sizes[0] + sizes[1] + sizes[2] + sizes[3];
      const int lh = lit_header_size(plane_len, csize);
      block_header(dst + pos, 2, lh + csize + 1, last);
      pos += 3;
      pos += lit_header(dst + pos, plane_len, csize);
      std::memcpy(dst + pos, pc.desc, static_cast<size_t>(pc.desc_size));
      pos += pc.desc_size;
      for (int j = 0; j < 3; ++j) { dst[pos++] = uint8_t(sizes[j]); dst[pos++] = uint8_t(sizes[j] >> 8); }
      for (int j = 0; j < 4; ++j) {
        std::memcpy(dst + pos, streams[j].data(), static_cast<size_t>(sizes[j]));
        pos += sizes[j];
      }
      dst[pos++] = 0;                           // Sequences_Section: no sequences
    }
  }
  put32(dst, static_cast<uint32_t>(pos - 4));
  return pos;
}

LSR_HD void frame_head(uint8_t* p, int T, int64_t nbytes, int64_t blocksize, int64_t cbytes) {
  p[0] = 2;                                                            // BLOSC_VERSION_FORMAT
  p[1] = 1;                                                            // zstd format version
  p[2] = static_cast<uint8_t>(0x10 | (4 << 5) | (T > 1 ? 0x1 : 0));   // blocks not split, zstd, byte shuffle
  p[3] = static_cast<uint8_t>(T);
  const uint32_t v[3] = {static_cast<uint32_t>(nbytes), static_cast<uint32_t>(blocksize), static_cast<uint32_t>(cbytes)};
  for (int k = 0; k < 3; ++k)
    for (int i = 0; i < 4; ++i) p[4 + 4 * k + i] = static_cast<uint8_t>(v[k] >> (8 * i));
}

// ---------------------------------------------------------------------------------------------------------------------
// device
// ---------------------------------------------------------------------------------------------------------------------

struct EncArgs {
  const uint8_t* src;
  int64_t src_bytes, frame_bytes, blocksize;
  int64_t blocks_per_frame, block_stride;
  uint8_t* streams;      // n_blocks * block_stride
  int* sizes;            // bytes of every block's stream ([cbytes] included)
};

__device__ __forceinline__ uint32_t load_u32_any(const uint8_t* p) {   // any alignment (amdhsa: unaligned access mode)
  uint32_t v;
  __builtin_memcpy(&v, p, 4);
  return v;
}
__device__ __forceinline__ void store_u32_any(uint8_t* p, uint32_t v) { __builtin_memcpy(p, &v, 4); }

// 16 source bytes at byte offset `off` of the block (zero beyond `valid`)
__device__ __forceinline__ void load_block16(const uint8_t* blk, int64_t off, int64_t valid, uint32_t w[4]) {
  w[0] = w[1] = w[2] = w[3] = 0;
  if (off + 16 <= valid) {
    for (int k = 0; k < 4; ++k) w[k] = load_u32_any(blk + off + 4 * k);
  } else {
    for (int k = 0; k < 16; ++k)
      if (off + k < valid) w[k >> 2] |= uint32_t(blk[off + k]) << (8 * (k & 3));
  }
}

// The streaming loops of the encode kernel walk the block in 16-byte vectors, kThreads of them side by side.  With one
// vector in flight per thread a pass costs a memory round trip per step (the kernel runs two workgroups per CU: nothing
// else hides it); kAhead vectors are loaded before the first is used (4 -> 8: 5.75 -> 5.63 ms per config-4 result; 16 needs
// 320 VGPRs, one workgroup per CU: 9.9 ms).
constexpr int kAhead = 8;
__device__ __forceinline__ void load_ahead(const uint8_t* blk, int64_t off0, int64_t bsize, int64_t valid, int tid,
                                           uint32_t w[kAhead][4]) {
  if (off0 + int64_t(kAhead) * kThreads * 16 <= valid) {
    // the whole batch lies inside the source (a workgroup-uniform test): unconditional loads, all issued before the
    // first use -- under a per-lane condition hipcc waits for every load before it issues the next
#pragma unroll
    for (int u = 0; u < kAhead; ++u) {
      const uint8_t* q = blk + off0 + (int64_t(u) * kThreads + tid) * 16;
#pragma unroll
      for (int k = 0; k < 4; ++k) w[u][k] = load_u32_any(q + 4 * k);
    }
    return;
  }
#pragma unroll
  for (int u = 0; u < kAhead; ++u) {
    const int64_t off = off0 + (int64_t(u) * kThreads + tid) * 16;
    w[u][0] = w[u][1] = w[u][2] = w[u][3] = 0;
    if (off < bsize) load_block16(blk, off, valid, w[u]);
  }
}

// the 16 / T symbols of plane p in the vector, packed low byte first (T = 4: one dword; 2: two; 1: four)
template <int T>
__device__ __forceinline__ void plane_symbols(const uint32_t w[4], int p, uint32_t s[4]) {
  if (T == 1) {
    for (int k = 0; k < 4; ++k) s[k] = w[k];
  } else if (T == 2) {
    for (int h = 0; h < 2; ++h) {
      const uint32_t a = w[2 * h] >> (8 * p), b = w[2 * h + 1] >> (8 * p);
      s[h] = (a & 0xFF) | ((a >> 16) & 0xFF) << 8 | (b & 0xFF) << 16 | ((b >> 16) & 0xFF) << 24;
    }
    s[2] = s[3] = 0;
  } else {
    s[0] = ((w[0] >> (8 * p)) & 0xFF) | ((w[1] >> (8 * p)) & 0xFF) << 8 | ((w[2] >> (8 * p)) & 0xFF) << 16 |
           ((w[3] >> (8 * p)) & 0xFF) << 24;
    s[1] = s[2] = s[3] = 0;
  }
}

struct __attribute__((aligned(16))) EncShared {
  uint32_t hist[kMaxType][256];
  uint32_t ct[kMaxType][256];
  uint8_t nbits[kMaxType][256];
  uint8_t desc[kMaxType][kHufHeaderMax + 4];
  uint32_t per_depth[kMaxType][kDepthSlots];
  uint16_t first[kMaxType][kHufMaxBits + 5];
  uint32_t payload_bits[kMaxType];
  unsigned long long sum_sq[kMaxType];
  int mode[kMaxType], desc_size[kMaxType], ns[kMaxType], max_bits[kMaxType], rle[kMaxType];
  int stream_bytes[4];
  int skip[kMaxType];      // the plane is hopeless on the sample: raw, not counted
  int pos;
};
// the tree scratch of the waves overlays the symbol buffer (the code tables are finished before a plane is staged)
struct __attribute__((aligned(16))) TreeScratch {
  uint32_t node_cnt[512];
  uint16_t parent[512];
  uint16_t order[256];
  HufScratch desc;
};

constexpr int kHistCopies = 8, kHistCopy = kMaxType * 256 + 4;    // (words)
inline size_t encode_lds_bytes(int64_t blocksize, int T) {
  const int plane_len = static_cast<int>(blocksize / T);
  const Runs r = stream_runs(huf_stream_len(plane_len, 0));
  const size_t symbuf = size_t(4) * 64 * (r.c + kLanePad);
  return sizeof(EncShared) + std::max(std::max(symbuf, size_t(kWaves) * sizeof(TreeScratch)), size_t(kHistCopies) * kHistCopy * 4);
}

extern __shared__ __attribute__((aligned(16))) uint8_t enc_dyn_lds[];

// Measurement build (make EXTRA=-DLSR_ENC_PROBE): cycle stamps of the phases of the first 1 024 workgroups
#ifdef LSR_ENC_PROBE
__device__ long long lsr_enc_probe_cycles[1024 * 16];
#define LSR_ENC_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 1024) lsr_enc_probe_cycles[blockIdx.x * 16 + (k)] = clock64(); } while (0)
#else
#define LSR_ENC_STAMP(k) do { } while (0)
#endif

// (Step 2, the two-queue merge, stays zstd_huf.hpp's: one lane, queue heads in registers, ~620 cycles per merge against
// LDS.  Two wave-uniform versions with the queues spread over the lanes of registers were measured and lost: with scalar
// branches 920 cycles per merge -- 24 taken branches --, branch-free with eight readlanes and scalar selects 760: a chain of
// ~100 dependent scalar instructions is no faster than five dependent LDS round trips.  profiles/r05_encode_phases.txt)
template <int T>
__global__ __launch_bounds__(kThreads) void encode_blocks_kernel(EncArgs a) {
  static_assert(T <= kMaxType && kMaxType <= kWaves, "one wave per plane");
  EncShared& S = *reinterpret_cast<EncShared*>(enc_dyn_lds);
  uint8_t* const symbuf = enc_dyn_lds + sizeof(EncShared);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t b = blockIdx.x;
  const int64_t f = b / a.blocks_per_frame, bi = b % a.blocks_per_frame;
  const int64_t in_frame = bi * a.blocksize;
  const int64_t bsize = imin64(a.blocksize, a.frame_bytes - in_frame);
  const int64_t src_off = f * a.frame_bytes + in_frame;
  const int64_t valid = imax64(0, imin64(bsize, a.src_bytes - src_off));
  const uint8_t* const blk = a.src + src_off;
  uint8_t* const out = a.streams + b * a.block_stride;
  const int plane_len = static_cast<int>(bsize / T);
  constexpr int kPer = 16 / T;                  // symbols of one plane in a 16-byte vector

  LSR_ENC_STAMP(0);
  // ---- histograms of all planes: one coalesced pass --------------------------------------------------------------
  // Eight copies of the counters, a lane adds to copy (lane & 7): in the high planes of float32 data a third of a wave's
  // symbols are one value, and atomics of one wave on one address run one after another.  The copies lie in the symbol
  // buffer (not in use yet), kHistCopy words apart so that one symbol's eight counters sit in eight different banks.
  uint32_t* const hcopy = reinterpret_cast<uint32_t*>(symbuf) + (lane & (kHistCopies - 1)) * kHistCopy;
  for (int i = tid; i < kMaxType * 256; i += kThreads) (&S.hist[0][0])[i] = 0;
  for (int i = tid; i < kHistCopies * kHistCopy; i += kThreads) reinterpret_cast<uint32_t*>(symbuf)[i] = 0;
  if (tid < kMaxType) S.skip[tid] = 0;             // (read by every thread below: set in front of the barrier)
  __syncthreads();
  // the sample first (huf_hopeless_sample): a plane that is noise on one vector in sixteen is stored raw and not counted
  // -- for a float32 result that is the two low mantissa planes, i.e. half of the pass's LDS atomics
  if (plane_len >= kSampleMinPlane) {
    for (int64_t off = int64_t(tid) * 16 * kSampleEvery; off < bsize; off += int64_t(kThreads) * 16 * kSampleEvery) {
      uint32_t w[4] = {0, 0, 0, 0};
      load_block16(blk, off, valid, w);
      const int nsym = static_cast<int>(imin64(16, bsize - off)) / T;
      for (int p = 0; p < T; ++p) {
        uint32_t s[4];
        plane_symbols<T>(w, p, s);
        for (int k = 0; k < nsym; ++k) atomicAdd(&S.hist[p][(s[k >> 2] >> (8 * (k & 3))) & 0xFF], 1u);
      }
    }
    __syncthreads();
    if (wave < T) {
      unsigned long long n = 0, coll = 0;
      for (int q = 0; q < 4; ++q) {
        const unsigned long long c = S.hist[wave][lane + 64 * q];
        n += c;
        coll += c * (c ? c - 1 : 0);
      }
      for (int d = 32; d > 0; d >>= 1) { n += __shfl_xor(n, d); coll += __shfl_xor(coll, d); }
      if (lane == 0) S.skip[wave] = huf_hopeless_sample(coll, n) ? 1 : 0;
    }
    __syncthreads();
    for (int i = tid; i < kMaxType * 256; i += kThreads) (&S.hist[0][0])[i] = 0;
    __syncthreads();
  }
  int early = 0;                                   // leading planes that are hopeless on the sample
  while (early < T && S.skip[early]) ++early;
  if (early == T) early = 0;                       // (no plane left that could compress: the block is stored verbatim)
  for (int64_t off0 = 0; off0 < bsize; off0 += int64_t(kAhead) * kThreads * 16) {
    uint32_t wa[kAhead][4];
    load_ahead(blk, off0, bsize, valid, tid, wa);
#pragma unroll
    for (int u = 0; u < kAhead; ++u) {
      const int64_t off = off0 + (int64_t(u) * kThreads + tid) * 16;
      const bool in = off < bsize;
      const uint32_t* w = wa[u];
      const int nsym = in ? static_cast<int>(imin64(16, bsize - off)) / T : 0;
      for (int p = 0; p < T; ++p) {
        if (S.skip[p]) {
          // a leading plane that is stored raw leaves in THIS sweep (its place is known: nothing but raw planes of fixed
          // size lies in front of it), on the assumption that some other plane of the block compresses -- the plane
          // loop below repeats it in the other layout if none does
          if (p < early && in) {
            uint32_t s[4];
            plane_symbols<T>(w, p, s);
            uint8_t* const dst = out + 4 + kFrameHeader + p * (3 + plane_len) + 3;
            const int64_t s_at = off / T;
            if (nsym == kPer) {
              for (int k = 0; k < (kPer + 3) / 4; ++k) store_u32_any(dst + s_at + 4 * k, s[k]);
            } else {
              for (int k = 0; k < nsym; ++k) dst[s_at + k] = static_cast<uint8_t>(s[k >> 2] >> (8 * (k & 3)));
            }
          }
          continue;
        }
        uint32_t s[4];
        plane_symbols<T>(w, p, s);
        // one atomic for the whole wave when all its symbols of this plane agree (the exponent plane of float32 data)
        const uint32_t s0 = s[0] & 0xFF;
        bool same = nsym == kPer;
        for (int k = 0; k < (kPer + 3) / 4; ++k) same = same && s[k] == s0 * 0x01010101u;
        const uint32_t lead = __builtin_amdgcn_readfirstlane(s0);
        if (__all(same && s0 == lead)) {
          if (lane == 0) atomicAdd(&S.hist[p][lead], 64u * kPer);
        } else {
          for (int k = 0; k < nsym; ++k) atomicAdd(&hcopy[p * 256 + ((s[k >> 2] >> (8 * (k & 3))) & 0xFF)], 1u);
        }
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < T * 256; i += kThreads) {
    uint32_t c = 0;
    for (int r = 0; r < kHistCopies; ++r) c += reinterpret_cast<const uint32_t*>(symbuf)[r * kHistCopy + i];
    (&S.hist[0][0])[i] += c;
  }
  __syncthreads();

  LSR_ENC_STAMP(1);
  // ---- code tables: wave w builds plane w (steps 1-8 of zstd_huf.hpp; the per-element steps on the 64 lanes) ------
  TreeScratch& W = *reinterpret_cast<TreeScratch*>(symbuf + size_t(wave) * sizeof(TreeScratch));
  const int pw = wave;
  const bool own = pw < T && !S.skip[pw < T ? pw : 0];
  if (pw < T && !own && lane == 0) {             // hopeless on the sample: raw
    S.mode[pw] = kPlaneRaw;
    S.desc_size[pw] = 0;
    S.ns[pw] = 0;
    S.payload_bits[pw] = 0;
    S.max_bits[pw] = 0;
  }
  const uint32_t* const count = S.hist[own ? pw : 0];
  // Steps 1 and 7 of zstd_huf.hpp are O(256) loops per symbol there (the host twin, and what defines the result); here
  // the wave does them with 64 lanes at once -- a bitonic sort of the 256 (count, symbol) keys for the rank order, ballots
  // for "symbols below me with my length" -- and only for planes that will be coded: the loops were two thirds of the
  // kernel's VALU instructions.
  uint32_t cnt4[4];
  int ns;
  bool coded;
  {
    int present = 0, only = 0;
    unsigned long long sq = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = lane + 64 * q;
      cnt4[q] = own ? count[i] : 0;
      const unsigned long long m = __ballot(cnt4[q] != 0);
      if (m) only = q * 64 + __builtin_ctzll(m);
      present += __popcll(m);
      sq += static_cast<unsigned long long>(cnt4[q]) * cnt4[q];
      if (own) { S.nbits[pw][i] = 0; S.ct[pw][i] = 0; }
    }
    for (int d = 32; d > 0; d >>= 1) sq += __shfl_xor(sq, d);
    ns = present;
    coded = own && ns >= 2 && plane_len >= kMinHufPlane && !huf_hopeless(sq, static_cast<uint64_t>(plane_len));
    if (own && lane == 0) {
      S.ns[pw] = present;
      S.mode[pw] = ns == 1 ? kPlaneRle : kPlaneRaw;
      S.rle[pw] = only;
      S.desc_size[pw] = 0;
      S.payload_bits[pw] = 0;
      S.max_bits[pw] = 0;
      S.sum_sq[pw] = sq;
    }
    if (own && lane < kDepthSlots) S.per_depth[pw][lane] = 0;
    if (coded) {
      // ascending (count, symbol); absent symbols sort last.  Element e = q * 64 + lane.
      uint32_t key[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) key[q] = cnt4[q] ? (cnt4[q] << 8 | static_cast<uint32_t>(lane + 64 * q)) : 0xFFFFFFFFu;
#pragma unroll
      for (int k = 2; k <= 256; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
          if (j >= 64) {
            const int jq = j >> 6;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              if (q & jq) continue;
              const bool up = ((q * 64) & k) == 0;              // (k >= 128 here: the direction does not depend on the lane)
              const uint32_t a = key[q], b = key[q | jq];
              const bool swap = (a > b) == up;
              key[q] = swap ? b : a;
              key[q | jq] = swap ? a : b;
            }
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const uint32_t other = static_cast<uint32_t>(__shfl_xor(static_cast<int>(key[q]), j));
              const bool up = ((q * 64 + lane) & k) == 0, lower = (lane & j) == 0;
              const uint32_t lo = key[q] < other ? key[q] : other, hi = key[q] < other ? other : key[q];
              key[q] = lower == up ? lo : hi;
            }
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int e = q * 64 + lane;
        if (e < ns) {
          W.order[e] = static_cast<uint16_t>(key[q] & 0xFF);
          W.node_cnt[e] = key[q] >> 8;
        }
      }
    }
  }
  __syncthreads();
  LSR_ENC_STAMP(2);
  if (coded && lane == 0) huf_merge(W.node_cnt, W.parent, ns);
  __syncthreads();
  LSR_ENC_STAMP(3);
  if (coded)
    for (int k = lane; k < ns; k += 64) atomicAdd(&S.per_depth[pw][huf_depth_of(W.parent, ns, k)], 1u);
  __syncthreads();
  if (coded && lane == 0) {
    huf_limit(S.per_depth[pw]);
    S.max_bits[pw] = huf_first_values(S.per_depth[pw], S.first[pw]);
  }
  __syncthreads();
  if (coded)
    for (int k = lane; k < ns; k += 64) S.nbits[pw][W.order[k]] = static_cast<uint8_t>(huf_length_of_rank(S.per_depth[pw], k));
  __syncthreads();
  if (coded) {
    int len4[4], below[4] = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 4; ++q) len4[q] = S.nbits[pw][lane + 64 * q];
    const int top = S.max_bits[pw];
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int l = 1; l <= top; ++l) {
      int base = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const unsigned long long m = __ballot(len4[q] == l);
        if (len4[q] == l) below[q] = base + __popcll(m & lt);
        base += __popcll(m);
      }
    }
    uint32_t bits = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = lane + 64 * q;
      const uint32_t e = len4[q] ? static_cast<uint32_t>(S.first[pw][len4[q]] + below[q]) | static_cast<uint32_t>(len4[q]) << 16 : 0u;
      S.ct[pw][i] = e;
      bits += cnt4[q] * (e >> 16);
    }
    atomicAdd(&S.payload_bits[pw], bits);
  }
  __syncthreads();
  LSR_ENC_STAMP(4);
  // step 8: the weights and their histogram with all lanes, the FSE stream (more than 128 weights) by one lane, the
  // chosen form's bytes with all lanes again
  int nw = 0;
  if (coded) {
    int w4[4];
    unsigned long long present[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int l = S.nbits[pw][lane + 64 * q];
      present[q] = __ballot(l != 0);
      w4[q] = l ? S.max_bits[pw] + 1 - l : 0;
    }
    const int last = present[3] ? 255 - __builtin_clzll(present[3]) : present[2] ? 191 - __builtin_clzll(present[2])
                   : present[1] ? 127 - __builtin_clzll(present[1]) : present[0] ? 63 - __builtin_clzll(present[0]) : 0;
    nw = last;                                        // (the last present symbol's weight is implied)
    int hist_mine = 0;                                // lane s < 13 ends up with the count of weight s
    for (int v = 0; v < 13; ++v) {
      int n = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = lane + 64 * q;
        n += __popcll(__ballot(i < nw && w4[q] == v));
      }
      if (lane == v) hist_mine = n;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (lane + 64 * q < nw) W.desc.w[lane + 64 * q] = static_cast<uint8_t>(w4[q]);
    if (lane < 13) W.desc.hist[lane] = hist_mine;
  }
  __syncthreads();
  if (coded && lane == 0) S.stream_bytes[pw] = nw >= 1 ? huf_description_body(nw, W.desc) : -1;   // (a slot free until the plane loop)
  __syncthreads();
  if (coded) {
    const int fse_size = S.stream_bytes[pw];
    const int ds = nw >= 1 ? huf_description_size(nw, fse_size) : -1;
    for (int i = lane; i < ds; i += 64) S.desc[pw][i] = huf_description_byte(i, nw, fse_size, W.desc);
    if (lane == 0) {
      S.desc_size[pw] = ds;
      if (ds > 0 && huf_pays(plane_len, S.payload_bits[pw], ds)) S.mode[pw] = kPlaneHuf;
    }
  }
  __syncthreads();

  LSR_ENC_STAMP(5);
  // ---- the block's stream, plane by plane ---------------------------------------------------------------------------
  bool any = false;
  for (int p = 0; p < T; ++p) any = any || S.mode[p] != kPlaneRaw;
  if (tid == 0) {
    S.pos = any ? 4 + kFrameHeader : 4;
    if (any) frame_header(out + 4, static_cast<uint32_t>(bsize));
  }
  const int q4 = (plane_len + 3) / 4;
  const Runs ra = stream_runs(huf_stream_len(plane_len, 0)), rb = stream_runs(huf_stream_len(plane_len, 3) > 0 ? huf_stream_len(plane_len, 3) : 1);
  const int c_pad = ra.c + kLanePad;
  for (int p = 0; p < T; ++p) {
    __syncthreads();
    const int pos = S.pos;
    const int mode = S.mode[p];
    const bool last = p == T - 1;
    int next_pos = pos;
    if (mode == kPlaneRaw) {
      const int hdr = any ? 3 : 0;
      if (any && tid == 0) block_header(out + pos, 0, plane_len, last);
      uint8_t* const dst = out + pos + hdr;
      const bool written = any && p < early;       // (the histogram sweep has put it where `pos` now points)
      for (int64_t off0 = 0; off0 < bsize && !written; off0 += int64_t(kAhead) * kThreads * 16) {
        uint32_t wa[kAhead][4];
        load_ahead(blk, off0, bsize, valid, tid, wa);
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
          const int64_t off = off0 + (int64_t(u) * kThreads + tid) * 16;
          if (off >= bsize) continue;
          uint32_t s[4];
          plane_symbols<T>(wa[u], p, s);
          const int nsym = static_cast<int>(imin64(16, bsize - off)) / T;
          const int64_t s_at = off / T;
          if (nsym == kPer) {
            for (int k = 0; k < (kPer + 3) / 4; ++k) store_u32_any(dst + s_at + 4 * k, s[k]);
          } else {
            for (int k = 0; k < nsym; ++k) dst[s_at + k] = static_cast<uint8_t>(s[k >> 2] >> (8 * (k & 3)));
          }
        }
      }
      next_pos = pos + hdr + plane_len;
    } else if (mode == kPlaneRle) {
      if (tid == 0) {
        block_header(out + pos, 1, plane_len, last);
        out[pos + 3] = static_cast<uint8_t>(S.rle[p]);
      }
      next_pos = pos + 4;
    } else {
      // stage the plane: symbol s of the plane -> stream j, lane l, place `within` of that lane's run
      for (int64_t off0 = 0; off0 < bsize; off0 += int64_t(kAhead) * kThreads * 16) {
        uint32_t wa[kAhead][4];
        load_ahead(blk, off0, bsize, valid, tid, wa);
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
          const int64_t off = off0 + (int64_t(u) * kThreads + tid) * 16;
          if (off >= bsize) continue;
          uint32_t s[4];
          plane_symbols<T>(wa[u], p, s);
          const int nsym = static_cast<int>(imin64(16, bsize - off)) / T;
          const int s_at = static_cast<int>(off / T);
          // (the kPer symbols of a vector are consecutive: they share stream and lane unless a boundary falls among them)
          const int j0 = (s_at >= q4) + (s_at >= 2 * q4) + (s_at >= 3 * q4);
          const Runs& r0 = j0 < 3 ? ra : rb;
          const int idx0 = s_at - j0 * q4;
          const int l0 = idx0 >> r0.shift;
          const int last = s_at + nsym - 1;
          const bool together = nsym == kPer && kPer <= 4 && l0 < r0.lanes - 1 && ((idx0 + nsym - 1) >> r0.shift) == l0 &&
                                last < (j0 + 1) * q4 && (idx0 & 3) == 0;
          if (together) {             // one aligned dword store
            *reinterpret_cast<uint32_t*>(symbuf + (j0 * 64 + l0) * c_pad + (idx0 & (r0.c - 1))) = s[0];
            continue;
          }
          for (int k = 0; k < nsym; ++k) {
            const int si = s_at + k;
            const int j = (si >= q4) + (si >= 2 * q4) + (si >= 3 * q4);
            const int idx = si - j * q4;
            const Runs& r = j < 3 ? ra : rb;
            int l = idx >> r.shift, within = idx & (r.c - 1);
            if (l >= r.lanes) { l = r.lanes - 1; within = idx - l * r.c; }
            symbuf[(j * 64 + l) * c_pad + within] = static_cast<uint8_t>(s[k >> 2] >> (8 * (k & 3)));
          }
        }
      }
      __syncthreads();
      // pass A: bits of every lane's run, the lane's place in its stream (streams are written last symbol first, so
      // lane 63's run comes first in the stream and lane 0's last)
      const int j = wave;
      const Runs& r = j < 3 ? ra : rb;
      const int n_l = lane < r.lanes - 1 ? r.c : (lane == r.lanes - 1 ? r.last_len : 0);
      const uint8_t* const run = symbuf + (j * 64 + lane) * c_pad;
      const uint32_t* const ct = S.ct[p];
      uint32_t my_bits = 0;
      {
        int i = 0;
        for (; i + 4 <= n_l; i += 4) {         // the run starts on a dword of LDS: four symbols, four independent look-ups
          const uint32_t four = *reinterpret_cast<const uint32_t*>(run + i);
          const uint32_t e0 = ct[four & 0xFF], e1 = ct[(four >> 8) & 0xFF], e2 = ct[(four >> 16) & 0xFF], e3 = ct[four >> 24];
          my_bits += (e0 >> 16) + (e1 >> 16) + (e2 >> 16) + (e3 >> 16);
        }
        for (; i < n_l; ++i) my_bits += ct[run[i]] >> 16;
      }
      uint32_t suffix = my_bits;                       // inclusive sum over lanes >= this one
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_down(suffix, d);
        if (lane + d < 64) suffix += t;
      }
      const uint32_t o = suffix - my_bits;             // bits of the stream in front of this lane's run
      const uint32_t stream_bits = __shfl(suffix, 0);
      if (lane == 0) S.stream_bytes[j] = static_cast<int>((stream_bits + 1 + 7) / 8);
      __syncthreads();
      const int ds = S.desc_size[p];
      const int sb0 = S.stream_bytes[0], sb1 = S.stream_bytes[1], sb2 = S.stream_bytes[2], sb3 = S.stream_bytes[3];
      const int csize = ds + 6 + sb0 + sb1 + sb2 + sb3;
      const int lh = lit_header_size(plane_len, csize);
      const int body = pos + 3 + lh;                   // tree description starts here
      if (tid == 0) {
        block_header(out + pos, 2, lh + csize + 1, last);
        lit_header(out + pos + 3, plane_len, csize);
        uint8_t* jt = out + body + ds;
        jt[0] = uint8_t(sb0); jt[1] = uint8_t(sb0 >> 8); jt[2] = uint8_t(sb1); jt[3] = uint8_t(sb1 >> 8);
        jt[4] = uint8_t(sb2); jt[5] = uint8_t(sb2 >> 8);
        out[body + csize] = 0;                         // Sequences_Section: no sequences
      }
      for (int i = tid; i < ds; i += kThreads) out[body + i] = S.desc[p][i];
      // pass B: every lane writes the bytes whose first bit lies in its run.  The `skip` bits in front of its first
      // byte boundary belong to a byte of the lane before it in the stream (lane + 1), which gets them by a shuffle
      // and closes its last byte with them; lane 0 closes the stream with the end mark instead.
      uint8_t* const sbase = out + body + ds + 6 + (j > 0 ? sb0 : 0) + (j > 1 ? sb1 : 0) + (j > 2 ? sb2 : 0);
      const uint32_t skip = (8 - (o & 7)) & 7;
      uint8_t* wp = sbase + ((o + skip) >> 3);
      uint64_t acc = 0;
      int nb = 0, i = n_l - 1;
      for (int t = 0; t < 7; ++t) {
        if (i >= 0 && nb < 7) {
          const uint32_t e = ct[run[i]];
          acc |= uint64_t(e & 0xFFFF) << nb;
          nb += static_cast<int>(e >> 16);
          --i;
        }
      }
      const uint32_t lead = static_cast<uint32_t>(acc) & ((1u << skip) - 1u);
      uint32_t recv = __shfl_up(lead, 1), recv_bits = __shfl_up(skip, 1);
      if (lane == 0) { recv = 1; recv_bits = 1; }
      if (n_l > 0) {
        acc >>= skip;
        nb -= static_cast<int>(skip);
        auto put = [&](uint32_t e) {
          acc |= uint64_t(e & 0xFFFF) << nb;
          nb += static_cast<int>(e >> 16);
        };
        auto flush = [&]() {
          if (nb >= 32) {
            store_u32_any(wp, static_cast<uint32_t>(acc));
            wp += 4;
            acc >>= 32;
            nb -= 32;
          }
        };
        for (; i >= 0 && ((i + 1) & 3); --i) { put(ct[run[i]]); flush(); }
        for (; i >= 3; i -= 4) {                 // symbols i-3 .. i: one dword of LDS, four look-ups in flight
          const uint32_t four = *reinterpret_cast<const uint32_t*>(run + i - 3);
          const uint32_t e3 = ct[four >> 24], e2 = ct[(four >> 16) & 0xFF], e1 = ct[(four >> 8) & 0xFF], e0 = ct[four & 0xFF];
          put(e3); put(e2); flush();
          put(e1); put(e0); flush();
        }
        acc |= uint64_t(recv) << nb;
        nb += static_cast<int>(recv_bits);
        for (int k = 0; k < (nb + 7) / 8; ++k) wp[k] = static_cast<uint8_t>(acc >> (8 * k));
      }
      next_pos = pos + 3 + lh + csize + 1;
    }
    __syncthreads();
    if (tid == 0) S.pos = next_pos;
    LSR_ENC_STAMP(6 + p);
  }
  __syncthreads();
  LSR_ENC_STAMP(10);
  if (tid == 0) {
    const int total = S.pos;
    const uint32_t cbytes = static_cast<uint32_t>(total - 4);
    out[0] = uint8_t(cbytes); out[1] = uint8_t(cbytes >> 8); out[2] = uint8_t(cbytes >> 16); out[3] = uint8_t(cbytes >> 24);
    a.sizes[b] = total;
  }
}

// bstarts of every block and the size of every frame: one workgroup per frame
__global__ __launch_bounds__(kThreads) void scan_frames_kernel(const int* sizes, int* bstarts, int64_t* frame_sizes,
                                                               int64_t blocks_per_frame) {
  __shared__ int part[kThreads];
  __shared__ int64_t running;
  const int tid = threadIdx.x;
  const int64_t f = blockIdx.x, b0 = f * blocks_per_frame;
  if (tid == 0) running = 16 + 4 * blocks_per_frame;
  __syncthreads();
  for (int64_t base = 0; base < blocks_per_frame; base += kThreads) {
    const int64_t i = base + tid;
    const int v = i < blocks_per_frame ? sizes[b0 + i] : 0;
    part[tid] = v;
    __syncthreads();
    for (int d = 1; d < kThreads; d <<= 1) {       // inclusive scan
      const int t = tid >= d ? part[tid - d] : 0;
      __syncthreads();
      part[tid] += t;
      __syncthreads();
    }
    if (i < blocks_per_frame) bstarts[b0 + i] = static_cast<int>(running + part[tid] - v);
    __syncthreads();
    if (tid == 0) running += part[kThreads - 1];
    __syncthreads();
  }
  if (tid == 0) frame_sizes[f] = running;
}

// frames[2 f] = offset of frame f in the output (16-byte aligned), frames[2 f + 1] = its size
__global__ void place_frames_kernel(const int64_t* frame_sizes, int64_t* frames, int64_t n_frames) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int64_t at = 0;
  for (int64_t f = 0; f < n_frames; ++f) {
    frames[2 * f] = at;
    frames[2 * f + 1] = frame_sizes[f];
    at += (frame_sizes[f] + 15) / 16 * 16;
  }
}

struct GatherArgs {
  const uint8_t* streams;
  const int* sizes;
  const int* bstarts;
  const int64_t* frames;
  uint8_t* out;
  int64_t out_cap, blocks_per_frame, block_stride, frame_bytes, blocksize;
  int typesize;
};

__global__ __launch_bounds__(kThreads) void gather_frames_kernel(GatherArgs a) {
  const int tid = threadIdx.x;
  const int64_t b = blockIdx.x, f = b / a.blocks_per_frame, bi = b % a.blocks_per_frame;
  const int64_t frame_at = a.frames[2 * f], frame_size = a.frames[2 * f + 1];
  if (frame_at + frame_size > a.out_cap) return;   // (the host checks the last frame's end: nothing is written past the buffer)
  uint8_t* const frame = a.out + frame_at;
  const int n = a.sizes[b];
  const int64_t at = a.bstarts[b];
  if (bi == 0 && tid == 0) frame_head(frame, a.typesize, a.frame_bytes, a.blocksize, frame_size);
  if (tid == 0) {
    const uint32_t v = static_cast<uint32_t>(at);
    uint8_t* q = frame + 16 + 4 * bi;
    q[0] = uint8_t(v); q[1] = uint8_t(v >> 8); q[2] = uint8_t(v >> 16); q[3] = uint8_t(v >> 24);
  }
  const uint8_t* const src = a.streams + b * a.block_stride;
  uint8_t* const dst = frame + at;
  // aligned dword stores, source read at the matching (unaligned) byte offset; head and tail bytewise
  const int head = static_cast<int>((4 - (reinterpret_cast<uintptr_t>(dst) & 3)) & 3);
  const int h = head < n ? head : n;
  if (tid < h) dst[tid] = src[tid];
  const int words = (n - h) / 4;
  for (int w = tid; w < words; w += kThreads)
    *reinterpret_cast<uint32_t*>(dst + h + 4 * w) = load_u32_any(src + h + 4 * w);
  const int tail0 = h + 4 * words;
  if (tid < n - tail0) dst[tail0 + tid] = src[tail0 + tid];
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------------

extern "C" int lsr_blosc_encode_device_plan(int64_t src_bytes, int typesize, int64_t frame_bytes, int64_t blocksize,
                                            int64_t* n_frames, int64_t* scratch_bytes, int64_t* out_cap) {
  Geometry g;
  const int rc = geometry(g, src_bytes, typesize, frame_bytes, blocksize);
  if (rc != LSR_OK) return rc;
  if (n_frames) *n_frames = g.n_frames;
  if (scratch_bytes) *scratch_bytes = scratch_bytes_of(g);
  if (out_cap) *out_cap = g.n_frames * frame_cap(g);
  return LSR_OK;
}

extern "C" int lsr_blosc_encode_device(const void* src, int64_t src_bytes, int typesize, int64_t frame_bytes,
                                       int64_t blocksize, void* scratch, int64_t scratch_bytes, uint8_t* out,
                                       int64_t out_cap, int64_t* frames, lsr_stream_t stream) {
  LSR_REQUIRE_PTR(src);
  LSR_REQUIRE_PTR(scratch);
  LSR_REQUIRE_PTR(out);
  LSR_REQUIRE_PTR(frames);
  Geometry g;
  const int rc = geometry(g, src_bytes, typesize, frame_bytes, blocksize);
  if (rc != LSR_OK) return rc;
  LSR_REQUIRE(scratch_bytes >= scratch_bytes_of(g), LSR_E_ARG, "scratch of %lld bytes, lsr_blosc_encode_device_plan asks for %lld",
              (long long)scratch_bytes, (long long)scratch_bytes_of(g));
  LSR_REQUIRE(out_cap >= g.n_frames * frame_cap(g), LSR_E_ARG, "output of %lld bytes, lsr_blosc_encode_device_plan asks for %lld",
              (long long)out_cap, (long long)(g.n_frames * frame_cap(g)));
  LSR_REQUIRE(reinterpret_cast<uintptr_t>(scratch) % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0, LSR_E_ARG,
              "scratch and output must be 16-byte aligned");
  hipStream_t s = lsr::as_stream(stream);
  uint8_t* const streams = static_cast<uint8_t*>(scratch);
  int* const sizes = reinterpret_cast<int*>(streams + sizes_offset(g));
  int* const bstarts = sizes + g.n_blocks;
  int64_t* const frame_sizes = reinterpret_cast<int64_t*>(streams + frame_sizes_offset(g));
  EncArgs a{static_cast<const uint8_t*>(src), g.src_bytes, g.frame_bytes, g.blocksize, g.blocks_per_frame, g.block_stride,
            streams, sizes};
  const size_t lds = encode_lds_bytes(g.blocksize, g.typesize);
  LSR_REQUIRE(lsr::lds_fits(lds), LSR_E_UNSUPPORTED, "the encode kernel needs %zu bytes of LDS", lds);
  const void* kernel = g.typesize == 4 ? reinterpret_cast<const void*>(&encode_blocks_kernel<4>)
                       : g.typesize == 2 ? reinterpret_cast<const void*>(&encode_blocks_kernel<2>)
                                         : reinterpret_cast<const void*>(&encode_blocks_kernel<1>);
  static std::atomic<uint64_t> done4{0}, done2{0}, done1{0};
  if (lds > 64 * 1024) {
    const int e = lsr::allow_dynamic_lds(kernel, static_cast<int>(lds), g.typesize == 4 ? done4 : g.typesize == 2 ? done2 : done1,
                                         "lsr_blosc_encode_device");
    if (e != LSR_OK) return e;
  }
  const dim3 grid(static_cast<unsigned>(g.n_blocks)), block(kThreads);
  if (g.typesize == 4) hipLaunchKernelGGL(encode_blocks_kernel<4>, grid, block, lds, s, a);
  else if (g.typesize == 2) hipLaunchKernelGGL(encode_blocks_kernel<2>, grid, block, lds, s, a);
  else hipLaunchKernelGGL(encode_blocks_kernel<1>, grid, block, lds, s, a);
  hipLaunchKernelGGL(scan_frames_kernel, dim3(static_cast<unsigned>(g.n_frames)), block, 0, s, sizes, bstarts, frame_sizes,
                     g.blocks_per_frame);
  hipLaunchKernelGGL(place_frames_kernel, dim3(1), dim3(64), 0, s, frame_sizes, frames, g.n_frames);
  GatherArgs ga{streams, sizes, bstarts, frames, out, out_cap, g.blocks_per_frame, g.block_stride, g.frame_bytes, g.blocksize,
                g.typesize};
  hipLaunchKernelGGL(gather_frames_kernel, grid, block, 0, s, ga);
  return lsr::launch_status("lsr_blosc_encode_device");
}

#ifdef LSR_ENC_PROBE
extern "C" int lsr_debug_enc_probe(long long* out, int n) {
  return static_cast<int>(hipMemcpyFromSymbol(out, HIP_SYMBOL(lsr_enc_probe_cycles), sizeof(long long) * static_cast<size_t>(n)));
}
#endif

// The host twin: the same frames, byte for byte, from and to host memory (`frames`: 2 n_frames int64 as above).
extern "C" int lsr_blosc_encode_device_cpu(const void* src, int64_t src_bytes, int typesize, int64_t frame_bytes,
                                           int64_t blocksize, void* scratch, int64_t scratch_bytes, uint8_t* out,
                                           int64_t out_cap, int64_t* frames, lsr_stream_t) {
  LSR_REQUIRE_PTR(src);
  LSR_REQUIRE_PTR(out);
  LSR_REQUIRE_PTR(frames);
  (void)scratch;
  (void)scratch_bytes;
  Geometry g;
  const int rc = geometry(g, src_bytes, typesize, frame_bytes, blocksize);
  if (rc != LSR_OK) return rc;
  LSR_REQUIRE(out_cap >= g.n_frames * frame_cap(g), LSR_E_ARG, "output of %lld bytes, lsr_blosc_encode_device_plan asks for %lld",
              (long long)out_cap, (long long)(g.n_frames * frame_cap(g)));
  const uint8_t* const in = static_cast<const uint8_t*>(src);
  std::vector<std::vector<uint8_t>> blocks(static_cast<size_t>(g.n_blocks));
  std::atomic<bool> failed{false};
  lsr::parallel_ranges(g.n_blocks, [&](int64_t b0, int64_t b1) {
    for (int64_t b = b0; b < b1; ++b) {
      const int64_t f = b / g.blocks_per_frame, bi = b % g.blocks_per_frame, in_frame = bi * g.blocksize;
      const int64_t bsize = std::min(g.blocksize, g.frame_bytes - in_frame);
      const int64_t src_off = f * g.frame_bytes + in_frame;
      const int64_t valid = std::max<int64_t>(0, std::min(bsize, g.src_bytes - src_off));
      blocks[size_t(b)].resize(static_cast<size_t>(bsize + kBlockSlack));
      const int64_t n = encode_block_host(in + (valid > 0 ? src_off : 0), valid, bsize, g.typesize, blocks[size_t(b)].data());
      blocks[size_t(b)].resize(static_cast<size_t>(n));
    }
  }, failed);
  LSR_REQUIRE(!failed.load(), LSR_E_ARG, "out of memory in the host encoder");
  int64_t at = 0;
  for (int64_t f = 0; f < g.n_frames; ++f) {
    uint8_t* const frame = out + at;
    int64_t pos = 16 + 4 * g.blocks_per_frame;
    for (int64_t bi = 0; bi < g.blocks_per_frame; ++bi) {
      const std::vector<uint8_t>& blk = blocks[size_t(f * g.blocks_per_frame + bi)];
      put32(frame + 16 + 4 * bi, static_cast<uint32_t>(pos));
      std::memcpy(frame + pos, blk.data(), blk.size());
      pos += static_cast<int64_t>(blk.size());
    }
    frame_head(frame, g.typesize, g.frame_bytes, g.blocksize, pos);
    frames[2 * f] = at;
    frames[2 * f + 1] = pos;
    at += (pos + 15) / 16 * 16;
  }
  return LSR_OK;
}
