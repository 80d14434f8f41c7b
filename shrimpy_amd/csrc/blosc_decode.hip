// blosc-zstd chunks decoded ON the MI355X: the reader threads only pread() the compressed chunks of a shard into a
// pinned buffer; the upload moves compressed bytes (a third to a half of the camera counts); this file turns them into
// the uint16 stack the deskew kernel reads.  Round 4 measured the host alternative at 0.087 s per config-4 unit on a
// rank's 16 cores (entropy decoding + unshuffle of 2.1 GB), the stage that bounds the store-to-store rate.
//
// Format (c-blosc 1.x, restated in csrc/blosc_frame.hip): a frame = 16-byte header, bstarts, blocks; with the zstd
// compressor every block (or, in split frames, every one of its `typesize` streams) is one complete zstd frame, or the
// shuffled bytes verbatim when they did not shrink.  The acquisition writes 32 KB blocks (mantis_engine.py:474-481 ->
// c-blosc clevel 1): a config-4 stack is 64 chunks x 1 024 blocks = 65 536 independent zstd frames.
//
// Two launches per volume:
//   decode_blocks   ONE LANE PER BLOCK runs the scalar decoder of csrc/zstd_lane.hpp (Huffman symbols in LDS, sequence
//                   tables in a per-lane workspace interleaved by lane): 65 536 lanes = 1 024 waves, one per SIMD;
//   unshuffle       shuffled blocks -> elements, 16 bytes per lane, coalesced both ways.
// Errors (a damaged chunk) never fault: the first failing block's index and code land in a status word.
// The host twin (lsr_blosc_decode_device_cpu) runs the same decoder block by block: tests without a GPU compare it with
// the system libzstd on frames written by libzstd at many levels, by c-blosc (tests/golden/blosc_frames.npz) and by
// this library's own encoders.

#include <algorithm>
#include <cstring>
#include <mutex>
#include <vector>

#include "common.hpp"
#include "host_parallel.hpp"
#ifdef LSR_DEC_PROBE
__device__ long long lsr_dec_probe_cycles[1024 * 8];
#endif
#include "zstd_lane.hpp"

namespace {

using namespace lsr::zd;

constexpr int kThreads = 256;
constexpr int kSymStride = 304 + 4 * kFastRuns + 4;   // LDS bytes per lane: 256 symbols, 12 class entries, the run list; an odd number of dwords between lanes
static_assert((kSymStride / 4) % 2 == 1 && kSymStride % 4 == 0, "lanes must sit an odd number of dwords apart");

struct Frames {
  int64_t n_frames, frame_nbytes, blocksize, blocks_per_frame, n_blocks;
  int typesize;
};

inline int frames_geometry(Frames& g, int64_t n_frames, int64_t frame_nbytes, int64_t blocksize, int typesize) {
  LSR_REQUIRE(typesize == 1 || typesize == 2 || typesize == 4, LSR_E_UNSUPPORTED,
              "typesize %d: the device decoder unshuffles 1-, 2- and 4-byte elements", typesize);
  LSR_REQUIRE(n_frames > 0 && n_frames < (int64_t(1) << 24), LSR_E_ARG, "%lld frames", (long long)n_frames);
  LSR_REQUIRE(frame_nbytes > 0 && frame_nbytes <= 0x7FFFFFFF - 16 && frame_nbytes % typesize == 0, LSR_E_ARG,
              "frames of %lld bytes", (long long)frame_nbytes);
  LSR_REQUIRE(blocksize > 0 && blocksize <= frame_nbytes && blocksize % typesize == 0, LSR_E_ARG, "blocksize %lld of %lld",
              (long long)blocksize, (long long)frame_nbytes);
  // one zstd frame per stream: a lane's destination is at most one block
  LSR_REQUIRE(blocksize <= (int64_t(1) << 24), LSR_E_UNSUPPORTED, "blocks of %lld bytes", (long long)blocksize);
  g.n_frames = n_frames; g.frame_nbytes = frame_nbytes; g.blocksize = blocksize; g.typesize = typesize;
  g.blocks_per_frame = lsr::ceil_div(frame_nbytes, blocksize);
  g.n_blocks = n_frames * g.blocks_per_frame;
  LSR_REQUIRE(g.n_blocks < (int64_t(1) << 30), LSR_E_UNSUPPORTED, "%lld blocks", (long long)g.n_blocks);
  return LSR_OK;
}

// The 64 lanes of a wave work through 64 blocks at about the same pace: were the blocks' scratch images `blocksize` apart
// (a power of two: 32 KB), every store of the wave would fall into addresses that are equal modulo 32 KB -- one memory
// channel.  kChannelSkew bytes between the images spread them over the channels (measured: see DESIGN.md section 5).
constexpr int64_t kChannelSkew = 256;
inline int64_t block_stride(const Frames& g) { return g.blocksize + kChannelSkew; }
// scratch: [shuffled blocks: n_blocks * block_stride (16-byte rounded)][block flags: n_blocks bytes][lane workspace]
inline int64_t flags_offset(const Frames& g) { return (g.n_blocks * block_stride(g) + 15) / 16 * 16; }
inline int64_t work_offset(const Frames& g) { return flags_offset(g) + (g.n_blocks + 15) / 16 * 16; }
inline int64_t lane_slots(const Frames& g) { return lsr::ceil_div(g.n_blocks, 64) * 64; }
inline int64_t decode_scratch_bytes(const Frames& g) { return work_offset(g) + lane_slots(g) * kWorkEntries * 4 + 64; }

// ---- what one lane does: one blosc block --------------------------------------------------------------------------------------
struct BlockJob {
  const uint8_t* frame;     // the blosc frame
  int64_t frame_len;
  int64_t bi;               // block index inside the frame
  uint8_t* dst;             // the block's shuffled bytes
};

// status word: 0, or (block index + 1) << 8 | (-code) of the first failing block
LSR_HD uint64_t status_of(int64_t block, int code) { return static_cast<uint64_t>(block + 1) << 8 | static_cast<uint64_t>(-code & 0xFF); }

// Returns kOk or an error; *flag = 1 when the bytes written are NOT shuffled (a "memcpyed" frame).
template <class Store>
LSR_HD int decode_block(Lane<Store>& L, const BlockJob& job, int64_t frame_nbytes, int64_t blocksize, int typesize, uint8_t* flag) {
  const uint8_t* f = job.frame;
  *flag = 0;
  const int64_t off = job.bi * blocksize;
  const int bsize = static_cast<int>(off + blocksize <= frame_nbytes ? blocksize : frame_nbytes - off);
  if (job.frame_len == 0) {                                   // an absent chunk: the fill value (zero)
    for (int i = 0; i < bsize; ++i) job.dst[i] = 0;
    *flag = 1;
    return kOk;
  }
  if (job.frame_len < 16) return kErrCorrupt;
  const int flags = f[2];
  const int T = f[3] ? f[3] : 1;
  const int64_t nbytes = static_cast<int64_t>(load_le(f + 4, 4)), bs = static_cast<int64_t>(load_le(f + 8, 4));
  if (nbytes != frame_nbytes) return kErrSize;
  if (flags & 0x2) {                                          // stored: the plain bytes follow the header
    if (job.frame_len < 16 + nbytes) return kErrCorrupt;
    copy_forward(job.dst, f + 16 + off, bsize);
    *flag = 1;
    return kOk;
  }
  if (bs != blocksize || T != typesize) return kErrSize;
  if ((flags >> 5) != 4) return kErrUnsupported;             // zstd streams only
  if ((flags & 0x4) && !((flags & 0x1) && T > 1)) return kErrUnsupported;   // bit shuffle
  if (!(flags & 0x1) || T == 1) *flag = 1;                    // not byte-shuffled
  const int64_t nblocks = (nbytes + blocksize - 1) / blocksize;
  if (16 + 4 * nblocks > job.frame_len) return kErrCorrupt;
  const bool leftover = bsize != blocksize;
  const bool split = !(flags & 0x10) && T <= 16 && blocksize / T >= 128 && !leftover;
  const int nsplits = split ? T : 1;
  if (split && blocksize % T) return kErrCorrupt;
  const int neblock = bsize / nsplits;
  int64_t pos = static_cast<int32_t>(load_le(f + 16 + 4 * job.bi, 4));
  for (int k = 0; k < nsplits; ++k) {
    if (pos < 16 + 4 * nblocks || pos + 4 > job.frame_len) return kErrCorrupt;
    const int64_t cb = static_cast<int32_t>(load_le(f + pos, 4));
    pos += 4;
    if (cb <= 0 || pos + cb > job.frame_len) return kErrCorrupt;
    uint8_t* d = job.dst + k * neblock;
    if (cb == neblock) {
      copy_forward(d, f + pos, neblock);
    } else {
      const int got = decode_frame(L, f + pos, static_cast<int>(cb), d, neblock);
      if (got < 0) return got;
      if (got != neblock) return kErrSize;
    }
    pos += cb;
  }
  return kOk;
}

// ---- host store ---------------------------------------------------------------------------------------------------------------------
struct HostStore {
  uint32_t ws[kWorkEntries];
  uint8_t sym[256];
  uint32_t cls[12];
  uint32_t runs[kFastRuns];
  uint32_t run_get(int i) const { return runs[i]; }
  void run_set(int i, uint32_t v) { runs[i] = v; }
  uint32_t cls_get(int i) const { return cls[i]; }
  void cls_set(int i, uint32_t v) { cls[i] = v; }
  uint32_t ws_get(int i) const { return ws[i]; }
  void ws_set(int i, uint32_t v) { ws[i] = v; }
  uint8_t sym_get(int i) const { return sym[i & 255]; }
  void sym_set(int i, uint8_t v) { sym[i & 255] = v; }
};

const Predefined& host_predefined() {
  static const Predefined p = [] { Predefined q; build_predefined(q); return q; }();
  return p;
}

// dst[i * T + k] = src[k * n + i]
void unshuffle_host(const uint8_t* src, uint8_t* dst, int64_t nbytes, int T) {
  const int64_t n = nbytes / T;
  for (int k = 0; k < T; ++k)
    for (int64_t i = 0; i < n; ++i) dst[i * T + k] = src[k * n + i];
}

// ---- device ---------------------------------------------------------------------------------------------------------------------------
// The symbol table is LDS, and the pointer says so: through a generic pointer every lookup is a FLAT load, which counts
// on the vector-memory counter as well -- each literal then waits for the wave's outstanding global loads and stores.
using lds_u8 = __attribute__((address_space(3))) uint8_t;
using lds_u32 = __attribute__((address_space(3))) uint32_t;
struct DevStore {
  uint32_t* ws;     // this lane's entry 0; entries are 64 dwords apart (interleaved by lane)
  lds_u8* sym;      // LDS: 256 symbols in rank order, the 11 class entries of the Huffman code (at 256), the run list (at 304)
  __device__ uint32_t cls_get(int i) const { return reinterpret_cast<const lds_u32*>(sym + 256)[i]; }
  __device__ void cls_set(int i, uint32_t v) { reinterpret_cast<lds_u32*>(sym + 256)[i] = v; }
  __device__ uint32_t run_get(int i) const { return reinterpret_cast<const lds_u32*>(sym + 304)[i]; }
  __device__ void run_set(int i, uint32_t v) { reinterpret_cast<lds_u32*>(sym + 304)[i] = v; }
  __device__ uint32_t ws_get(int i) const { return ws[static_cast<int64_t>(i) * 64]; }
  __device__ void ws_set(int i, uint32_t v) { ws[static_cast<int64_t>(i) * 64] = v; }
  __device__ uint8_t sym_get(int i) const { return sym[i & 255]; }
  __device__ void sym_set(int i, uint8_t v) { sym[i & 255] = v; }
};

struct DecArgs {
  const uint8_t* comp;
  const int64_t* frames;    // (offset, size) per frame; size 0 = absent chunk
  int64_t comp_bytes;
  int64_t n_blocks, blocks_per_frame, frame_nbytes, blocksize, block_stride;
  int typesize;
  uint8_t* shuffled;        // n_blocks * block_stride
  uint8_t* flags;           // per block
  uint32_t* work;           // lane workspace
  const Predefined* pre;
  unsigned long long* status;
};

__global__ __launch_bounds__(kThreads) void decode_blocks_kernel(DecArgs a) {
  __shared__ __attribute__((aligned(16))) uint8_t sym_lds[kThreads * kSymStride];
  const int tid = threadIdx.x, lane = tid & 63;
  const int64_t b = int64_t(blockIdx.x) * kThreads + tid;
  if (b >= a.n_blocks) return;
  const int64_t wave = b >> 6;
  Lane<DevStore> L;
  L.store.ws = a.work + wave * (int64_t(kWorkEntries) * 64) + lane;
  L.store.sym = (lds_u8*)(sym_lds) + tid * kSymStride;
  L.pre = a.pre;
  const int64_t f = b / a.blocks_per_frame;
  const int64_t at = a.frames[2 * f], size = a.frames[2 * f + 1];
  int rc;
  uint8_t flag = 0;
  if (at < 0 || size < 0 || at + size > a.comp_bytes) {
    rc = kErrCorrupt;
  } else {
    BlockJob job{a.comp + at, size, b % a.blocks_per_frame, a.shuffled + b * a.block_stride};
    rc = decode_block(L, job, a.frame_nbytes, a.blocksize, a.typesize, &flag);
  }
  LSR_DEC_STAMP(6);
  a.flags[b] = flag;
  if (rc != kOk) atomicCAS(a.status, 0ull, static_cast<unsigned long long>(status_of(b, rc)));
}

struct UnshuffleArgs {
  const uint8_t* shuffled;
  const uint8_t* flags;
  uint8_t* out;
  int64_t out_bytes, n_blocks, blocks_per_frame, frame_nbytes, blocksize, block_stride;
};

// one workgroup per block; lane: 16 output bytes = 16 / T elements
template <int T>
__global__ __launch_bounds__(kThreads) void unshuffle_kernel(UnshuffleArgs a) {
  const int64_t b = blockIdx.x;
  const int64_t f = b / a.blocks_per_frame, bi = b % a.blocks_per_frame;
  const int64_t off = bi * a.blocksize;
  const int64_t bsize = off + a.blocksize <= a.frame_nbytes ? a.blocksize : a.frame_nbytes - off;
  const int64_t base = f * a.frame_nbytes + off;
  if (base >= a.out_bytes) return;                            // the zero padding of an edge chunk
  const int64_t room = a.out_bytes - base < bsize ? a.out_bytes - base : bsize;
  const uint8_t* src = a.shuffled + b * a.block_stride;
  uint8_t* dst = a.out + base;
  const bool plain = T == 1 || a.flags[b];
  const int64_t n = bsize / T;                                // elements per plane
  for (int64_t o = int64_t(threadIdx.x) * 16; o < room; o += int64_t(kThreads) * 16) {
    uint32_t w[4] = {0, 0, 0, 0};
    const int64_t take = room - o < 16 ? room - o : 16;
    if (plain) {
      if (take == 16) { __builtin_memcpy(w, src + o, 16); }
      else { for (int k = 0; k < take; ++k) w[k >> 2] |= uint32_t(src[o + k]) << (8 * (k & 3)); }
    } else {
      const int64_t e = o / T;                                // first element of this lane
      const int ne = static_cast<int>((take + T - 1) / T);
      if (T == 2) {
        uint64_t lo = 0, hi = 0;
        if (ne == 8 && e + 8 <= n) { __builtin_memcpy(&lo, src + e, 8); __builtin_memcpy(&hi, src + n + e, 8); }
        else { for (int k = 0; k < ne; ++k) { lo |= uint64_t(src[e + k]) << (8 * k); hi |= uint64_t(src[n + e + k]) << (8 * k); } }
        for (int k = 0; k < 8; ++k) w[k >> 1] |= (uint32_t((lo >> (8 * k)) & 0xFF) | uint32_t((hi >> (8 * k)) & 0xFF) << 8) << (16 * (k & 1));
      } else {
        uint32_t p[4] = {0, 0, 0, 0};
        if (ne == 4 && e + 4 <= n) { for (int q = 0; q < 4; ++q) __builtin_memcpy(&p[q], src + q * n + e, 4); }
        else { for (int q = 0; q < 4; ++q) for (int k = 0; k < ne; ++k) p[q] |= uint32_t(src[q * n + e + k]) << (8 * k); }
        for (int k = 0; k < 4; ++k)
          w[k] = ((p[0] >> (8 * k)) & 0xFF) | ((p[1] >> (8 * k)) & 0xFF) << 8 | ((p[2] >> (8 * k)) & 0xFF) << 16 | ((p[3] >> (8 * k)) & 0xFF) << 24;
      }
    }
    if (take == 16) { __builtin_memcpy(dst + o, w, 16); }
    else { for (int k = 0; k < take; ++k) dst[o + k] = static_cast<uint8_t>(w[k >> 2] >> (8 * (k & 3))); }
  }
}

// the predefined sequence tables in device memory, one copy per device
const Predefined* device_predefined(int* status) {
  static Predefined* per_device[64] = {nullptr};
  static std::atomic<uint64_t> done{0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
    *status = lsr::fail(LSR_E_UNSUPPORTED, "no HIP device (or a device index above 63) for the decoder's tables");
    return nullptr;
  }
  const uint64_t bit = uint64_t(1) << dev;
  if (!(done.load(std::memory_order_acquire) & bit)) {
    static std::mutex* guard = new std::mutex;
    std::lock_guard<std::mutex> lock(*guard);
    if (!(done.load(std::memory_order_acquire) & bit)) {
      Predefined* p = nullptr;
      hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), sizeof(Predefined));
      if (e == hipSuccess) e = hipMemcpy(p, &host_predefined(), sizeof(Predefined), hipMemcpyHostToDevice);
      if (e != hipSuccess) {
        *status = lsr::fail(static_cast<int>(e), "predefined zstd tables: %s", hipGetErrorString(e));
        return nullptr;
      }
      per_device[dev] = p;
      done.fetch_or(bit, std::memory_order_release);
    }
  }
  *status = LSR_OK;
  return per_device[dev];
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------------

extern "C" int lsr_blosc_decode_device_plan(int64_t n_frames, int64_t frame_nbytes, int64_t blocksize, int typesize,
                                            int64_t* scratch_bytes) {
  Frames g;
  const int rc = frames_geometry(g, n_frames, frame_nbytes, blocksize, typesize);
  if (rc != LSR_OK) return rc;
  if (scratch_bytes) *scratch_bytes = decode_scratch_bytes(g);
  return LSR_OK;
}

extern "C" int lsr_blosc_decode_device(const uint8_t* comp, int64_t comp_bytes, const int64_t* frames, int64_t n_frames,
                                       int64_t frame_nbytes, int64_t blocksize, int typesize, uint8_t* out,
                                       int64_t out_bytes, void* scratch, int64_t scratch_bytes,
                                       unsigned long long* status, lsr_stream_t stream) {
  LSR_REQUIRE_PTR(comp);
  LSR_REQUIRE_PTR(frames);
  LSR_REQUIRE_PTR(out);
  LSR_REQUIRE_PTR(scratch);
  LSR_REQUIRE_PTR(status);
  Frames g;
  const int rc = frames_geometry(g, n_frames, frame_nbytes, blocksize, typesize);
  if (rc != LSR_OK) return rc;
  LSR_REQUIRE(scratch_bytes >= decode_scratch_bytes(g), LSR_E_ARG, "scratch of %lld bytes, lsr_blosc_decode_device_plan asks for %lld",
              (long long)scratch_bytes, (long long)decode_scratch_bytes(g));
  LSR_REQUIRE(out_bytes > 0 && out_bytes <= g.n_frames * g.frame_nbytes && out_bytes > (g.n_frames - 1) * g.frame_nbytes, LSR_E_SHAPE,
              "%lld frames of %lld bytes do not cover a volume of %lld bytes", (long long)n_frames, (long long)frame_nbytes,
              (long long)out_bytes);
  LSR_REQUIRE(comp_bytes >= 0 && comp_bytes < lsr::kMaxVoxels, LSR_E_ARG, "compressed buffer of %lld bytes", (long long)comp_bytes);
  LSR_REQUIRE(reinterpret_cast<uintptr_t>(scratch) % 16 == 0, LSR_E_ARG, "scratch must be 16-byte aligned");
  int st = LSR_OK;
  const Predefined* pre = device_predefined(&st);
  if (pre == nullptr) return st;
  hipStream_t s = lsr::as_stream(stream);
  uint8_t* const base = static_cast<uint8_t*>(scratch);
  DecArgs a{comp, frames, comp_bytes, g.n_blocks, g.blocks_per_frame, g.frame_nbytes, g.blocksize, block_stride(g), g.typesize, base,
            base + flags_offset(g), reinterpret_cast<uint32_t*>(base + work_offset(g)), pre, status};
  hipError_t e = hipMemsetAsync(status, 0, sizeof(unsigned long long), s);
  if (e != hipSuccess) return lsr::fail(static_cast<int>(e), "lsr_blosc_decode_device: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(decode_blocks_kernel, dim3(static_cast<unsigned>(lsr::ceil_div(g.n_blocks, kThreads))), dim3(kThreads), 0, s, a);
  UnshuffleArgs u{base, base + flags_offset(g), out, out_bytes, g.n_blocks, g.blocks_per_frame, g.frame_nbytes, g.blocksize,
                  block_stride(g)};
  const dim3 grid(static_cast<unsigned>(g.n_blocks)), block(kThreads);
  if (g.typesize == 4) hipLaunchKernelGGL(unshuffle_kernel<4>, grid, block, 0, s, u);
  else if (g.typesize == 2) hipLaunchKernelGGL(unshuffle_kernel<2>, grid, block, 0, s, u);
  else hipLaunchKernelGGL(unshuffle_kernel<1>, grid, block, 0, s, u);
  return lsr::launch_status("lsr_blosc_decode_device");
}

// Host twin: the same decoder, block by block (host pointers; `scratch` may be NULL).
extern "C" int lsr_blosc_decode_device_cpu(const uint8_t* comp, int64_t comp_bytes, const int64_t* frames, int64_t n_frames,
                                           int64_t frame_nbytes, int64_t blocksize, int typesize, uint8_t* out,
                                           int64_t out_bytes, void* scratch, int64_t scratch_bytes,
                                           unsigned long long* status, lsr_stream_t) {
  LSR_REQUIRE_PTR(comp);
  LSR_REQUIRE_PTR(frames);
  LSR_REQUIRE_PTR(out);
  LSR_REQUIRE_PTR(status);
  (void)scratch;
  (void)scratch_bytes;
  Frames g;
  const int rc = frames_geometry(g, n_frames, frame_nbytes, blocksize, typesize);
  if (rc != LSR_OK) return rc;
  LSR_REQUIRE(out_bytes > 0 && out_bytes <= g.n_frames * g.frame_nbytes && out_bytes > (g.n_frames - 1) * g.frame_nbytes, LSR_E_SHAPE,
              "%lld frames of %lld bytes do not cover a volume of %lld bytes", (long long)n_frames, (long long)frame_nbytes,
              (long long)out_bytes);
  std::atomic<unsigned long long> first{0};
  std::atomic<bool> failed{false};
  lsr::parallel_ranges(g.n_blocks, [&](int64_t b0, int64_t b1) {
    Lane<HostStore> L;
    L.pre = &host_predefined();
    std::vector<uint8_t> tmp(static_cast<size_t>(g.blocksize) + 16);
    for (int64_t b = b0; b < b1; ++b) {
      const int64_t f = b / g.blocks_per_frame, bi = b % g.blocks_per_frame;
      const int64_t at = frames[2 * f], size = frames[2 * f + 1];
      const int64_t off = bi * g.blocksize;
      const int64_t bsize = off + g.blocksize <= g.frame_nbytes ? g.blocksize : g.frame_nbytes - off;
      int code;
      uint8_t flag = 0;
      if (at < 0 || size < 0 || at + size > comp_bytes) {
        code = kErrCorrupt;
      } else {
        BlockJob job{comp + at, size, bi, tmp.data()};
        code = decode_block(L, job, g.frame_nbytes, g.blocksize, g.typesize, &flag);
      }
      if (code != kOk) {
        unsigned long long zero = 0;
        first.compare_exchange_strong(zero, static_cast<unsigned long long>(status_of(b, code)));
        continue;
      }
      const int64_t base = f * g.frame_nbytes + off;
      if (base >= out_bytes) continue;
      const int64_t room = std::min(bsize, out_bytes - base);
      if (flag || g.typesize == 1) {
        std::memcpy(out + base, tmp.data(), static_cast<size_t>(room));
      } else if (room == bsize) {
        unshuffle_host(tmp.data(), out + base, bsize, g.typesize);
      } else {
        std::vector<uint8_t> whole(static_cast<size_t>(bsize));
        unshuffle_host(tmp.data(), whole.data(), bsize, g.typesize);
        std::memcpy(out + base, whole.data(), static_cast<size_t>(room));
      }
    }
  }, failed);
  LSR_REQUIRE(!failed.load(), LSR_E_ARG, "out of memory in the host decoder");
  *status = first.load();
  return LSR_OK;
}

#ifdef LSR_DEC_PROBE
extern "C" int lsr_debug_dec_probe(long long* out, int n) {
  return static_cast<int>(hipMemcpyFromSymbol(out, HIP_SYMBOL(lsr_dec_probe_cycles), sizeof(long long) * static_cast<size_t>(n)));
}
#endif

// One zstd frame through the lane decoder on the host (tests and fuzzing of csrc/zstd_lane.hpp): *out_n = decoded bytes.
// Returns LSR_OK, or LSR_E_ARG with the decoder's code in the message.
extern "C" int lsr_zstd_lane_decode_cpu(const uint8_t* src, int64_t n, uint8_t* dst, int64_t cap, int64_t* out_n) {
  LSR_REQUIRE_PTR(src);
  LSR_REQUIRE_PTR(out_n);
  LSR_REQUIRE(n >= 0 && n <= 0x7FFFFFF0 && cap >= 0 && cap <= 0x7FFFFFF0, LSR_E_ARG, "sizes %lld / %lld", (long long)n, (long long)cap);
  if (cap > 0) LSR_REQUIRE_PTR(dst);
  Lane<HostStore> L;
  L.pre = &host_predefined();
  uint8_t none = 0;
  const int got = decode_frame(L, src, static_cast<int>(n), cap > 0 ? dst : &none, static_cast<int>(cap));
  LSR_REQUIRE(got >= 0, LSR_E_ARG, "zstd frame does not decode (code %d: -1 corrupt, -2 destination too small, -3 unsupported, -4 size)", got);
  *out_n = got;
  return LSR_OK;
}
