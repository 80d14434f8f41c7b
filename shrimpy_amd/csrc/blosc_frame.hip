// Host side of the blosc ingest (no kernel in this file): walk one c-blosc 1.x frame and run the entropy
// decoder of every stream straight into the caller's buffer -- typically a slab of a pinned staging
// slot.  The acquisition writes such frames (blosc, zstd, byte shuffle; shrimpy/mantis/mantis_engine.py:
// 474-481) with c-blosc's own parameters: 32 KB blocks, one zstd stream each (c-blosc does not split zstd
// blocks), i.e. ~1000 streams per 32-plane chunk of a camera stack.  Walking them in Python costs ~10 us
// per stream under the GIL -- 2.5 s per config-4 volume however many threads decode -- so the walk lives
// here, one call per chunk with the GIL released, and a reader thread pool scales (0.11 s per volume on 16
// threads, the rate of c-blosc itself).  The byte shuffle is undone here too, block
// by block while the block is in cache (undoing it on the device behind the upload was built and measured:
// no gain, zstd is what the time goes to -- DESIGN.md section 5).
//
// Frame layout (c-blosc 1.x, blosc.h / blosc.c; restated, not copied): 16-byte header
//   [0] format version  [1] codec format  [2] flags  [3] typesize  [4:8] nbytes  [8:12] blocksize  [12:16] cbytes
// flags: 0x1 byte shuffle, 0x2 stored ("memcpyed": the nbytes follow the header), 0x4 bit shuffle,
// 0x10 blocks are not split, bits 5-7 compressor (0 blosclz, 1 lz4, 2 snappy, 3 zlib, 4 zstd); then
// int32 bstarts[nblocks]; a block is 1 or `typesize` streams (split when !0x10, typesize <= 16,
// blocksize / typesize >= 128 and the block is not the short last one), each int32 cbytes + payload,
// payload stored verbatim when cbytes equals the stream's decoded size.
// The zstd / lz4 / zlib decoders come from the system libraries at run time (dlopen): no headers needed.

#include <dlfcn.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "common.hpp"

namespace {

using zstd_decompress_t = size_t (*)(void*, size_t, const void*, size_t);
using zstd_is_error_t = unsigned (*)(size_t);
using zstd_compress_t = size_t (*)(void*, size_t, const void*, size_t, int);
using zstd_bound_t = size_t (*)(size_t);
using zstd_create_cctx_t = void* (*)();
using zstd_free_cctx_t = size_t (*)(void*);
using zstd_compress_cctx_t = size_t (*)(void*, void*, size_t, const void*, size_t, int);
using lz4_decompress_t = int (*)(const char*, char*, int, int);
using zlib_uncompress_t = int (*)(unsigned char*, unsigned long*, const unsigned char*, unsigned long);

struct Decoders {
  zstd_decompress_t zstd = nullptr;
  zstd_is_error_t zstd_is_error = nullptr;
  zstd_compress_t zstd_compress = nullptr;     // the writer's side (lsr_blosc_encode_host)
  zstd_bound_t zstd_bound = nullptr;
  zstd_create_cctx_t zstd_create_cctx = nullptr;   // one context per frame: ZSTD_compress alone builds and frees one per
  zstd_free_cctx_t zstd_free_cctx = nullptr;       // call -- an mmap / munmap pair per 256 KB block, serialised process-wide
  zstd_compress_cctx_t zstd_compress_cctx = nullptr;
  lz4_decompress_t lz4 = nullptr;
  zlib_uncompress_t zlib = nullptr;
};

void* open_first(const char* env, const char* a, const char* b) {
  const char* names[3] = {env ? std::getenv(env) : nullptr, a, b};
  for (const char* n : names) {
    if (n == nullptr || *n == 0) continue;
    if (void* h = dlopen(n, RTLD_NOW | RTLD_LOCAL)) return h;
  }
  return nullptr;
}

const Decoders& decoders() {
  static const Decoders d = [] {
    Decoders r;
    if (void* h = open_first("LSR_LIBZSTD", "libzstd.so.1", "libzstd.so")) {
      r.zstd = reinterpret_cast<zstd_decompress_t>(dlsym(h, "ZSTD_decompress"));
      r.zstd_is_error = reinterpret_cast<zstd_is_error_t>(dlsym(h, "ZSTD_isError"));
      if (r.zstd == nullptr || r.zstd_is_error == nullptr) r.zstd = nullptr;
      r.zstd_compress = reinterpret_cast<zstd_compress_t>(dlsym(h, "ZSTD_compress"));
      r.zstd_bound = reinterpret_cast<zstd_bound_t>(dlsym(h, "ZSTD_compressBound"));
      r.zstd_create_cctx = reinterpret_cast<zstd_create_cctx_t>(dlsym(h, "ZSTD_createCCtx"));
      r.zstd_free_cctx = reinterpret_cast<zstd_free_cctx_t>(dlsym(h, "ZSTD_freeCCtx"));
      r.zstd_compress_cctx = reinterpret_cast<zstd_compress_cctx_t>(dlsym(h, "ZSTD_compressCCtx"));
      if (r.zstd_is_error == nullptr || r.zstd_bound == nullptr) r.zstd_compress = nullptr;
      if (r.zstd_create_cctx == nullptr || r.zstd_free_cctx == nullptr) r.zstd_compress_cctx = nullptr;
    }
    if (void* h = open_first("LSR_LIBLZ4", "liblz4.so.1", "liblz4.so"))
      r.lz4 = reinterpret_cast<lz4_decompress_t>(dlsym(h, "LZ4_decompress_safe"));
    if (void* h = open_first("LSR_LIBZ", "libz.so.1", "libz.so"))
      r.zlib = reinterpret_cast<zlib_uncompress_t>(dlsym(h, "uncompress"));
    return r;
  }();
  return d;
}

inline int32_t le32(const uint8_t* p) {
  return static_cast<int32_t>(uint32_t(p[0]) | uint32_t(p[1]) << 8 | uint32_t(p[2]) << 16 | uint32_t(p[3]) << 24);
}

// dst[i * T + k] = src[k * n + i]; trailing nbytes % T bytes verbatim
void unshuffle_block(const uint8_t* src, uint8_t* dst, int64_t nbytes, int T) {
  const int64_t n = nbytes / T;
  if (T == 2) {
    const uint8_t *lo = src, *hi = src + n;
    for (int64_t i = 0; i < n; ++i) {
      dst[2 * i] = lo[i];
      dst[2 * i + 1] = hi[i];
    }
  } else if (T == 4) {
    const uint8_t *a = src, *b = src + n, *c = src + 2 * n, *d = src + 3 * n;
    for (int64_t i = 0; i < n; ++i) {
      dst[4 * i] = a[i];
      dst[4 * i + 1] = b[i];
      dst[4 * i + 2] = c[i];
      dst[4 * i + 3] = d[i];
    }
  } else {
    for (int k = 0; k < T; ++k)
      for (int64_t i = 0; i < n; ++i) dst[i * T + k] = src[k * n + i];
  }
  std::memcpy(dst + n * T, src + n * T, static_cast<size_t>(nbytes - n * T));
}

// dst[k * n + i] = src[i * T + k]; trailing nbytes % T bytes verbatim (the inverse of unshuffle_block)
void shuffle_block(const uint8_t* src, uint8_t* dst, int64_t nbytes, int T) {
  const int64_t n = nbytes / T;
  if (T == 4) {
    uint8_t *a = dst, *b = dst + n, *c = dst + 2 * n, *d = dst + 3 * n;
    for (int64_t i = 0; i < n; ++i) {
      a[i] = src[4 * i];
      b[i] = src[4 * i + 1];
      c[i] = src[4 * i + 2];
      d[i] = src[4 * i + 3];
    }
  } else if (T == 2) {
    uint8_t *lo = dst, *hi = dst + n;
    for (int64_t i = 0; i < n; ++i) {
      lo[i] = src[2 * i];
      hi[i] = src[2 * i + 1];
    }
  } else {
    for (int k = 0; k < T; ++k)
      for (int64_t i = 0; i < n; ++i) dst[k * n + i] = src[i * T + k];
  }
  std::memcpy(dst + n * T, src + n * T, static_cast<size_t>(nbytes - n * T));
}

inline void put32(uint8_t* p, uint32_t v) {
  p[0] = static_cast<uint8_t>(v); p[1] = static_cast<uint8_t>(v >> 8); p[2] = static_cast<uint8_t>(v >> 16);
  p[3] = static_cast<uint8_t>(v >> 24);
}

}  // namespace

// 1 when lsr_blosc_encode_host can write zstd streams here (libzstd's compressor is loadable)
extern "C" int lsr_blosc_host_encoder(void) { return decoders().zstd_compress != nullptr; }

// The writer's side: one c-blosc 1.x frame of `src` (nbytes < 2 GiB) with zstd streams, one per block (blocks are not
// split -- what c-blosc itself does for zstd), byte shuffle for typesize > 1 when `shuffle` is 1, none when 0.  The
// frames the Python encoder of shrimpy_amd/io/codecs.py writes, byte for byte (same block size rule: `blocksize` or
// 256 KB, cut to whole elements; a block zstd does not shrink is stored verbatim; a frame that does not shrink at all
// takes the "memcpyed" form) -- but with the GIL released, so the writer's thread pool scales: the CLI's default output
// (blosc-zstd, what the acquisition engine writes: shrimpy/mantis/mantis_engine.py:474-481) took 0.60 s per config-4
// result through the Python encoder on 16 threads.  `dst`: at least nbytes + 16 + 4 * blocks + 4 * blocks bytes
// (lsr_blosc_encode_bound); *out_bytes receives the frame's size.
extern "C" int64_t lsr_blosc_encode_bound(int64_t nbytes, int typesize, int64_t blocksize) {
  if (nbytes < 0 || nbytes > 0x7FFFFFFF - 16) return -1;
  const int T = typesize >= 1 && typesize <= 255 ? typesize : 1;
  int64_t bs = blocksize > 0 ? blocksize : 256 * 1024;
  bs = nbytes >= T ? std::max<int64_t>(T, std::min(bs, nbytes) / T * T) : std::max<int64_t>(nbytes, 1);
  const int64_t nblocks = nbytes ? (nbytes + bs - 1) / bs : 0;
  return 16 + nbytes + 8 * nblocks + 64;
}

extern "C" int lsr_blosc_encode_host(const uint8_t* src, int64_t nbytes, int typesize, int clevel, int shuffle,
                                     int64_t blocksize, uint8_t* dst, int64_t cap, int64_t* out_bytes) {
  LSR_REQUIRE_PTR(dst);
  LSR_REQUIRE_PTR(out_bytes);
  LSR_REQUIRE(nbytes >= 0 && nbytes <= 0x7FFFFFFF - 16, LSR_E_ARG, "a blosc 1.x frame holds less than 2 GiB, got %lld bytes",
              (long long)nbytes);
  if (nbytes > 0) LSR_REQUIRE_PTR(src);
  LSR_REQUIRE(shuffle == 0 || shuffle == 1, LSR_E_UNSUPPORTED, "shuffle %d: the native encoder writes byte shuffle (1) or none (0)",
              shuffle);
  LSR_REQUIRE(clevel >= 0 && clevel <= 22, LSR_E_ARG, "zstd level %d outside [0, 22]", clevel);
  const Decoders& z = decoders();
  LSR_REQUIRE(z.zstd_compress != nullptr, LSR_E_UNSUPPORTED, "libzstd's compressor is not loadable here (dlopen)");
  const int T = typesize >= 1 && typesize <= 255 ? typesize : 1;
  int64_t bs = blocksize > 0 ? blocksize : 256 * 1024;
  bs = nbytes >= T ? std::max<int64_t>(T, std::min(bs, nbytes) / T * T) : std::max<int64_t>(nbytes, 1);
  const int64_t nblocks = nbytes ? (nbytes + bs - 1) / bs : 0;
  LSR_REQUIRE(cap >= lsr_blosc_encode_bound(nbytes, typesize, blocksize), LSR_E_ARG,
              "destination of %lld bytes is smaller than lsr_blosc_encode_bound", (long long)cap);
  int flags = 0x10 | (4 << 5);                       // blocks not split, zstd
  const bool shuffled = shuffle == 1 && T > 1;
  if (shuffled) flags |= 0x1;
  const size_t bound = z.zstd_bound(static_cast<size_t>(bs));
  uint8_t* scratch = static_cast<uint8_t*>(std::malloc(static_cast<size_t>(bs) + bound));
  LSR_REQUIRE(scratch != nullptr, LSR_E_ARG, "out of memory for a %lld-byte block", (long long)bs);
  uint8_t* const shuf = scratch;
  uint8_t* const comp = scratch + bs;
  void* const cctx = z.zstd_compress_cctx != nullptr ? z.zstd_create_cctx() : nullptr;
  int64_t pos = 16 + 4 * nblocks;
  bool stored_whole = false;
  for (int64_t b = 0; b < nblocks; ++b) {
    const int64_t off = b * bs, n = std::min(bs, nbytes - off);
    const uint8_t* block = src + off;
    if (shuffled) {
      shuffle_block(block, shuf, n, T);
      block = shuf;
    }
    const size_t c = cctx != nullptr ? z.zstd_compress_cctx(cctx, comp, bound, block, static_cast<size_t>(n), clevel)
                                     : z.zstd_compress(comp, bound, block, static_cast<size_t>(n), clevel);
    const bool verbatim = z.zstd_is_error(c) || static_cast<int64_t>(c) >= n;
    const int64_t cb = verbatim ? n : static_cast<int64_t>(c);
    if (pos + 4 + cb > cap || pos + 4 + cb >= nbytes + 16) {   // not shrinking: the memcpyed form below
      stored_whole = true;
      break;
    }
    put32(dst + 16 + 4 * b, static_cast<uint32_t>(pos));
    put32(dst + pos, static_cast<uint32_t>(cb));
    std::memcpy(dst + pos + 4, verbatim ? block : comp, static_cast<size_t>(cb));
    pos += 4 + cb;
  }
  std::free(scratch);
  if (cctx != nullptr) z.zstd_free_cctx(cctx);
  if (stored_whole || pos >= nbytes + 16) {
    flags |= 0x2;
    if (nbytes) std::memcpy(dst + 16, src, static_cast<size_t>(nbytes));
    pos = 16 + nbytes;
  }
  dst[0] = 2;                                        // BLOSC_VERSION_FORMAT
  dst[1] = 1;                                        // zstd format version
  dst[2] = static_cast<uint8_t>(flags);
  dst[3] = static_cast<uint8_t>(T);
  put32(dst + 4, static_cast<uint32_t>(nbytes));
  put32(dst + 8, static_cast<uint32_t>(bs));
  put32(dst + 12, static_cast<uint32_t>(pos));
  *out_bytes = pos;
  return LSR_OK;
}

// 1 when the decoder of blosc compressor code `compressor` (1 lz4, 3 zlib, 4 zstd) is loadable here
extern "C" int lsr_blosc_host_codec(int compressor) {
  const Decoders& d = decoders();
  return compressor == 4 ? d.zstd != nullptr : compressor == 1 ? d.lz4 != nullptr : compressor == 3 ? d.zlib != nullptr : 0;
}

extern "C" int lsr_blosc_decode_host(const uint8_t* frame, int64_t frame_bytes, uint8_t* out, int64_t out_bytes,
                                     int* typesize_out) {
  LSR_REQUIRE_PTR(frame);
  LSR_REQUIRE(frame_bytes >= 16, LSR_E_ARG, "blosc frame shorter than its 16-byte header");
  const int flags = frame[2];
  const int T = frame[3] ? frame[3] : 1;
  const int64_t nbytes = static_cast<uint32_t>(le32(frame + 4));
  const int64_t blocksize = static_cast<uint32_t>(le32(frame + 8));
  if (typesize_out) *typesize_out = T;
  LSR_REQUIRE(nbytes == out_bytes, LSR_E_SHAPE, "blosc frame holds %lld bytes, destination has %lld",
              (long long)nbytes, (long long)out_bytes);
  if (nbytes == 0) return LSR_OK;
  LSR_REQUIRE_PTR(out);
  if (flags & 0x2) {  // stored
    LSR_REQUIRE(frame_bytes >= 16 + nbytes, LSR_E_ARG, "corrupt blosc frame: stored payload runs past the end");
    std::memcpy(out, frame + 16, static_cast<size_t>(nbytes));
    return LSR_OK;
  }
  const bool byte_shuffled = (flags & 0x1) && T > 1;
  LSR_REQUIRE(byte_shuffled || !(flags & 0x4), LSR_E_UNSUPPORTED, "bit-shuffled blosc frame: decoded by the Python codec");
  LSR_REQUIRE(blocksize > 0 && blocksize <= nbytes, LSR_E_ARG, "corrupt blosc frame: blocksize %lld of %lld bytes",
              (long long)blocksize, (long long)nbytes);
  // a split block is T streams of blocksize / T bytes each: a remainder would leave the tail of the scratch undecoded
  LSR_REQUIRE((flags & 0x10) || T > 16 || blocksize / T < 128 || blocksize % T == 0, LSR_E_ARG,
              "corrupt blosc frame: split blocks of %lld bytes are not a multiple of typesize %d", (long long)blocksize, T);
  const int compressor = flags >> 5;
  const Decoders& dec = decoders();
  LSR_REQUIRE(lsr_blosc_host_codec(compressor), LSR_E_UNSUPPORTED,
              "blosc compressor %d: its decoder library is not loadable here (zstd 4, lz4 1, zlib 3 via dlopen)", compressor);
  const int64_t nblocks = (nbytes + blocksize - 1) / blocksize;
  LSR_REQUIRE(16 + 4 * nblocks <= frame_bytes, LSR_E_ARG, "corrupt blosc frame: block table runs past the end");
  uint8_t* scratch = nullptr;
  if (byte_shuffled) {
    scratch = static_cast<uint8_t*>(std::calloc(static_cast<size_t>(blocksize), 1));
    LSR_REQUIRE(scratch != nullptr, LSR_E_ARG, "out of memory for a %lld-byte block", (long long)blocksize);
  }
  int status = LSR_OK;
  for (int64_t b = 0; b < nblocks && status == LSR_OK; ++b) {
    const int64_t bsize = b * blocksize + blocksize <= nbytes ? blocksize : nbytes - b * blocksize;
    const bool leftover = bsize != blocksize;
    const bool split = !(flags & 0x10) && T <= 16 && blocksize / T >= 128 && !leftover;
    const int nsplits = split ? T : 1;
    const int64_t neblock = bsize / nsplits;
    uint8_t* dest = out + b * blocksize;
    uint8_t* target = scratch ? scratch : dest;
    int64_t pos = le32(frame + 16 + 4 * b);
    for (int k = 0; k < nsplits; ++k) {
      if (pos < 0 || pos + 4 > frame_bytes) { status = lsr::fail(LSR_E_ARG, "corrupt blosc frame: stream runs past the end"); break; }
      const int64_t cb = le32(frame + pos);
      pos += 4;
      if (cb < 0 || pos + cb > frame_bytes) { status = lsr::fail(LSR_E_ARG, "corrupt blosc frame: stream runs past the end"); break; }
      uint8_t* dst = target + k * neblock;
      if (cb == neblock) {
        std::memcpy(dst, frame + pos, static_cast<size_t>(cb));
      } else if (compressor == 4) {
        const size_t got = dec.zstd(dst, static_cast<size_t>(neblock), frame + pos, static_cast<size_t>(cb));
        if (dec.zstd_is_error(got) || static_cast<int64_t>(got) != neblock)
          status = lsr::fail(LSR_E_ARG, "corrupt blosc frame: zstd stream does not decode to the block size");
      } else if (compressor == 1) {
        const int got = dec.lz4(reinterpret_cast<const char*>(frame + pos), reinterpret_cast<char*>(dst), static_cast<int>(cb),
                                static_cast<int>(neblock));
        if (got != neblock) status = lsr::fail(LSR_E_ARG, "corrupt blosc frame: lz4 stream does not decode to the block size");
      } else {
        unsigned long got = static_cast<unsigned long>(neblock);
        const int rc = dec.zlib(dst, &got, frame + pos, static_cast<unsigned long>(cb));
        if (rc != 0 || static_cast<int64_t>(got) != neblock)
          status = lsr::fail(LSR_E_ARG, "corrupt blosc frame: zlib stream does not decode to the block size");
      }
      if (status != LSR_OK) break;
      pos += cb;
    }
    if (status != LSR_OK) break;
    if (byte_shuffled) unshuffle_block(scratch, dest, bsize, T);
  }
  std::free(scratch);
  return status;
}

// ---- CRC-32C (Castagnoli), host side -------------------------------------------------------------------------------
// The Zarr v3 `crc32c` codec appends this checksum to the shard index and, where a store lists it in the chunk
// codec chain, to every chunk: a 32-plane camera chunk is ~270 MB, which a byte loop under Python's GIL checks at
// ~10 MB/s.  Here: the SSE4.2 crc32 instruction eight bytes at a time (~10 GB/s), slice-by-8 tables on a CPU without
// it; one call per buffer with the GIL released (ctypes), `seed` = the running value for a buffer in pieces (0 first).
namespace {

struct Crc32cTables {
  uint32_t t[8][256];
  Crc32cTables() {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
      t[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int k = 1; k < 8; ++k) t[k][i] = (t[k - 1][i] >> 8) ^ t[0][t[k - 1][i] & 0xFF];
  }
};

uint32_t crc32c_tables(uint32_t crc, const uint8_t* p, int64_t n) {
  static const Crc32cTables T;
  while (n > 0 && (reinterpret_cast<uintptr_t>(p) & 7)) { crc = T.t[0][(crc ^ *p++) & 0xFF] ^ (crc >> 8); --n; }
  for (; n >= 8; n -= 8, p += 8) {
    uint64_t w;
    std::memcpy(&w, p, 8);
    w ^= crc;
    crc = T.t[7][w & 0xFF] ^ T.t[6][(w >> 8) & 0xFF] ^ T.t[5][(w >> 16) & 0xFF] ^ T.t[4][(w >> 24) & 0xFF] ^
          T.t[3][(w >> 32) & 0xFF] ^ T.t[2][(w >> 40) & 0xFF] ^ T.t[1][(w >> 48) & 0xFF] ^ T.t[0][w >> 56];
  }
  for (; n > 0; --n) crc = T.t[0][(crc ^ *p++) & 0xFF] ^ (crc >> 8);
  return crc;
}

__attribute__((target("sse4.2"))) uint32_t crc32c_sse42(uint32_t crc, const uint8_t* p, int64_t n) {
  while (n > 0 && (reinterpret_cast<uintptr_t>(p) & 7)) { crc = __builtin_ia32_crc32qi(crc, *p++); --n; }
  // one stream, eight bytes per instruction (its 3-cycle latency bounds this at ~8 GB/s per thread: far above the
  // entropy decoder that runs next to it)
  uint64_t c = crc;
  for (; n >= 8; n -= 8, p += 8) {
    uint64_t w;
    std::memcpy(&w, p, 8);
    c = __builtin_ia32_crc32di(c, w);
  }
  crc = static_cast<uint32_t>(c);
  for (; n > 0; --n) crc = __builtin_ia32_crc32qi(crc, *p++);
  return crc;
}

}  // namespace

extern "C" int lsr_crc32c_host(const uint8_t* data, int64_t n, uint32_t seed, uint32_t* out) {
  LSR_REQUIRE_PTR(out);
  LSR_REQUIRE(n >= 0 && n < lsr::kMaxVoxels, LSR_E_ARG, "crc32c over %lld bytes", (long long)n);
  if (n > 0) LSR_REQUIRE_PTR(data);
  static const bool sse42 = __builtin_cpu_supports("sse4.2") != 0;
  const uint32_t crc = ~seed;
  *out = ~(sse42 ? crc32c_sse42(crc, data, n) : crc32c_tables(crc, data, n));
  return LSR_OK;
}
// (for the parity test of the two implementations: the table walk on any CPU)
extern "C" int lsr_crc32c_host_portable(const uint8_t* data, int64_t n, uint32_t seed, uint32_t* out) {
  LSR_REQUIRE_PTR(out);
  LSR_REQUIRE(n >= 0 && n < lsr::kMaxVoxels, LSR_E_ARG, "crc32c over %lld bytes", (long long)n);
  if (n > 0) LSR_REQUIRE_PTR(data);
  *out = ~crc32c_tables(~seed, data, n);
  return LSR_OK;
}
