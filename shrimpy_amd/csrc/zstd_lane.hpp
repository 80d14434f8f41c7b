// A zstd frame decoder (RFC 8878) written as scalar code that one LANE runs: on the MI355X every lane of the decode
// kernel (csrc/blosc_decode.hip) decodes one blosc block -- one zstd frame of <= 128 KB per block is how c-blosc, and
// therefore the acquisition engine, stores zstd data (shrimpy/mantis/mantis_engine.py:474-481) -- so a config-4 volume
// is 65 536 independent frames, one per lane of 1 024 waves.  The same functions compile for the host, where the
// GPU-less container checks them against frames the system libzstd wrote (tests/test_device_codec.py) and where they
// run under the address sanitiser on damaged input (tools/fuzz_zstd_lane.py): every read is checked against the end
// of the source, every write against the end of the destination; a damaged frame is an error code, never a fault.
//
// Covered: single frames (magic, header with or without window descriptor / content size / checksum flag; dictionary
// IDs are refused), Raw / RLE / Compressed blocks; literals Raw, RLE, Huffman (one or four streams, direct or FSE-coded
// weights) and "treeless"; sequences with predefined, RLE, FSE-compressed and repeat tables, repeat offsets.
// The frame checksum is skipped, not verified.  This states the published format; it is not libzstd's code.
//
// Layout choices for the lane: the Huffman code is decoded canonically (12 thresholds in registers + the 256 symbols
// in rank order: LDS on the device) instead of through a 2^11-entry table; the three sequence tables live in a
// per-lane workspace (global memory, entries interleaved by lane); literals are decoded INTO the tail of the block's
// output range, which the sequence execution provably never overtakes, so there is no literal buffer.
#pragma once

#include <stdint.h>

#include "zstd_huf.hpp"   // LSR_HD, highbit

// Functions the kernel must see inlined: only then does the compiler know that the stream and workspace pointers come
// from kernel arguments (global memory) and emit global_load / global_store for them.  Out of line they are generic
// pointers, every access is a FLAT one, and a flat access counts on the LDS counter and the vector-memory counter both.
#if defined(__HIP_DEVICE_COMPILE__)
#define LSR_HD_INL LSR_HD __attribute__((always_inline))
#else
#define LSR_HD_INL LSR_HD
#endif

// Measurement build (-DLSR_DEC_PROBE, device code only): cycle stamps of the phases of a block, taken by lane 0 of the
// first waves (csrc/blosc_decode.hip defines the buffer)
#if defined(LSR_DEC_PROBE) && defined(__HIP_DEVICE_COMPILE__)
extern __device__ long long lsr_dec_probe_cycles[1024 * 8];
#define LSR_DEC_STAMP(k)                                                                                          \
  do {                                                                                                            \
    const long long w_ = (static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;                    \
    if ((threadIdx.x & 63) == 0 && w_ < 1024) lsr_dec_probe_cycles[w_ * 8 + (k)] = clock64();                     \
  } while (0)
#else
#define LSR_DEC_STAMP(k) do { } while (0)
#endif

namespace lsr {
namespace zd {

enum : int {
  kOk = 0,
  kErrCorrupt = -1,      // the source does not follow the format / ends early
  kErrDstSmall = -2,     // the frame decodes to more than the destination holds
  kErrUnsupported = -3,  // a dictionary, or not a zstd frame
  kErrSize = -4,         // decoded size differs from the expected one
};

constexpr int kLLLogMax = 9, kOFLogMax = 8, kMLLogMax = 9;
// 32-bit entries of the lane workspace: the three sequence tables (they persist from block to block: Repeat_Mode), the
// Huffman weights while a tree description is read (eight 4-bit weights per entry) and their FSE table
constexpr int kLLBase = 0, kOFBase = 512, kMLBase = 768, kWeightBase = 1280, kWeightTabBase = 1312, kSeqBase = 1376;
// ... and the match offsets of a block's sequences when there are few enough of them to place the literals directly
// (decode_block_body); their literal and match lengths sit in the store's run list (run_get / run_set: LDS on the
// device), 16 bits each, which is why the direct path takes blocks of at most 64 KB
// (the first kFastRuns of them; later ones in the workspace behind the offsets)
constexpr int kMaxDirectSeq = 192, kFastRuns = 72, kMaxDirectRoom = 65536, kWorkEntries = kSeqBase + 2 * kMaxDirectSeq;
constexpr int kBlockMax = 128 * 1024;

LSR_HD uint64_t load_le(const uint8_t* p, int n) {   // n <= 8 bytes, little endian
  uint64_t v = 0;
  for (int i = 0; i < n; ++i) v |= static_cast<uint64_t>(p[i]) << (8 * i);
  return v;
}
LSR_HD uint64_t load64(const uint8_t* p) {
  uint64_t v;
  __builtin_memcpy(&v, p, 8);
  return v;
}

// ---- bit readers --------------------------------------------------------------------------------------------------------
// Backward reader of a stream that ends with a '1' mark (Huffman and FSE payloads): `pos` unread bits below the mark.
struct BackBits {
  const uint8_t* p;
  int len;
  int pos;        // bits still unread; negative once more were asked for than the stream holds
  uint64_t win;   // bits [lo, lo + 64) of the stream
  int lo;
};
LSR_HD void back_refill(BackBits& b) {
  if (b.len >= 8) {
    int sb = b.pos > 0 ? ((b.pos - 1) >> 3) - 7 : 0;
    if (sb < 0) sb = 0;
    b.win = load64(b.p + sb);
    b.lo = 8 * sb;
  } else {
    b.win = load_le(b.p, b.len);
    b.lo = 0;
  }
}
LSR_HD bool back_init(BackBits& b, const uint8_t* p, int len) {
  if (len <= 0) return false;
  const uint8_t last = p[len - 1];
  if (last == 0) return false;
  b.p = p;
  b.len = len;
  b.pos = 8 * (len - 1) + zs::highbit(last);
  back_refill(b);
  return true;
}
// the next n bits (n <= 32) without consuming them; bits "before the start" read as zero (the format's rule)
LSR_HD uint32_t back_peek(BackBits& b, int n) {
  if (n == 0) return 0;
  if (b.pos >= n) {
    if (b.pos - n < b.lo) back_refill(b);
    return static_cast<uint32_t>(b.win >> (b.pos - n - b.lo)) & (n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u));
  }
  if (b.pos <= 0) return 0;
  if (b.lo > 0) back_refill(b);
  const uint32_t have = static_cast<uint32_t>(b.win) & ((1u << b.pos) - 1u);   // pos < n <= 32
  return have << (n - b.pos);
}
LSR_HD uint32_t back_read(BackBits& b, int n) {
  const uint32_t v = back_peek(b, n);
  b.pos -= n;
  return v;
}

// Forward little-endian reader (the normalised-count header)
struct FwdBits {
  const uint8_t* p;
  int len, pos;   // bytes, bit position
};
LSR_HD uint32_t fwd_peek(const FwdBits& f, int n) {   // n <= 25; zero beyond the end
  const int byte = f.pos >> 3;
  uint64_t v = 0;
  for (int i = 0; i < 5; ++i)
    if (byte + i < f.len) v |= static_cast<uint64_t>(f.p[byte + i]) << (8 * i);
  return static_cast<uint32_t>(v >> (f.pos & 7)) & ((1u << n) - 1u);
}

// ---- FSE ------------------------------------------------------------------------------------------------------------------
// normalised counts of up to 53 symbols; -1 = "less than one"
struct NCount {
  int16_t norm[64];
  int max_symbol, log;
};
// Returns the bytes used (> 0) or an error.
LSR_HD int read_ncount(const uint8_t* p, int len, int max_symbol_allowed, int max_log, NCount& nc) {
  FwdBits f{p, len, 0};
  const int log = 5 + static_cast<int>(fwd_peek(f, 4));
  f.pos += 4;
  if (log > max_log) return kErrCorrupt;
  int remaining = (1 << log) + 1, threshold = 1 << log, nb = log + 1, sym = 0;
  bool prev0 = false;
  for (int i = 0; i < 64; ++i) nc.norm[i] = 0;
  while (remaining > 1 && sym <= max_symbol_allowed) {
    if (prev0) {
      for (;;) {
        const int r = static_cast<int>(fwd_peek(f, 2));
        f.pos += 2;
        sym += r;
        if (r < 3) break;
        if (sym > max_symbol_allowed) return kErrCorrupt;
      }
      if (sym > max_symbol_allowed) return kErrCorrupt;
    }
    const int max = (2 * threshold - 1) - remaining;
    int count;
    const int low = static_cast<int>(fwd_peek(f, nb - 1));
    if (low < max) {
      count = low;
      f.pos += nb - 1;
    } else {
      count = static_cast<int>(fwd_peek(f, nb));
      if (count >= threshold) count -= max;
      f.pos += nb;
    }
    --count;
    remaining -= count < 0 ? -count : count;
    nc.norm[sym++] = static_cast<int16_t>(count);
    prev0 = count == 0;
    if (remaining < 1) return kErrCorrupt;
    while (remaining < threshold) { --nb; threshold >>= 1; }
    if ((f.pos >> 3) > len) return kErrCorrupt;
  }
  if (remaining != 1 || sym > max_symbol_allowed + 1) return kErrCorrupt;
  nc.max_symbol = sym - 1;
  nc.log = log;
  const int used = (f.pos + 7) >> 3;
  return used <= len ? used : kErrCorrupt;
}

// decoding-table entry: symbol | bits << 8 | base << 16
LSR_HD uint32_t fse_entry(int symbol, int bits, int base) {
  return static_cast<uint32_t>(symbol) | static_cast<uint32_t>(bits) << 8 | static_cast<uint32_t>(base) << 16;
}

// `T` addresses 32-bit table entries: T.get(i) / T.set(i, v)
template <class T>
LSR_HD void fse_build(T tab, const NCount& nc) {
  const int size = 1 << nc.log, mask = size - 1, step = (size >> 1) + (size >> 3) + 3;
  uint16_t next[64];
  int high = size - 1;
  for (int s = 0; s <= nc.max_symbol; ++s) {
    if (nc.norm[s] == -1) {
      tab.set(high--, static_cast<uint32_t>(s));
      next[s] = 1;
    } else {
      next[s] = static_cast<uint16_t>(nc.norm[s]);
    }
  }
  int pos = 0;
  for (int s = 0; s <= nc.max_symbol; ++s) {
    for (int i = 0; i < nc.norm[s]; ++i) {
      tab.set(pos, static_cast<uint32_t>(s));
      do { pos = (pos + step) & mask; } while (pos > high);
    }
  }
  for (int u = 0; u < size; ++u) {
    const int s = static_cast<int>(tab.get(u) & 0xFF);
    const int x = next[s]++;
    const int bits = nc.log - zs::highbit(static_cast<uint32_t>(x));
    tab.set(u, fse_entry(s, bits, (x << bits) - size));
  }
}

// ---- sequences: code -> (baseline, extra bits) --------------------------------------------------------------------------------
LSR_HD void ll_code(int code, uint32_t& base, int& bits) {
  if (code < 16) { base = static_cast<uint32_t>(code); bits = 0; return; }
  const uint32_t b[20] = {16, 18, 20, 22, 24, 28, 32, 40, 48, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536};
  const uint8_t n[20] = {1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
  base = b[code - 16];
  bits = n[code - 16];
}
LSR_HD void ml_code(int code, uint32_t& base, int& bits) {
  if (code < 32) { base = static_cast<uint32_t>(code + 3); bits = 0; return; }
  const uint32_t b[21] = {35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099, 8195, 16387, 32771, 65539};
  const uint8_t n[21] = {1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
  base = b[code - 32];
  bits = n[code - 32];
}

LSR_HD void predefined_ncount(int which /* 0 LL, 1 OF, 2 ML */, NCount& nc) {
  const int8_t ll[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
  const int8_t of[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
  const int8_t ml[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                         1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
  for (int i = 0; i < 64; ++i) nc.norm[i] = 0;
  if (which == 0) { for (int i = 0; i < 36; ++i) nc.norm[i] = ll[i]; nc.max_symbol = 35; nc.log = 6; }
  else if (which == 1) { for (int i = 0; i < 29; ++i) nc.norm[i] = of[i]; nc.max_symbol = 28; nc.log = 5; }
  else { for (int i = 0; i < 53; ++i) nc.norm[i] = ml[i]; nc.max_symbol = 52; nc.log = 6; }
}

// The three predefined tables, built once by the host (plain arrays) and handed to every lane.
struct Predefined {
  uint32_t ll[64], of[32], ml[64];
};
struct PlainTab {
  uint32_t* t;
  LSR_HD uint32_t get(int i) const { return t[i]; }
  LSR_HD void set(int i, uint32_t v) const { t[i] = v; }
};
inline void build_predefined(Predefined& p) {
  NCount nc;
  predefined_ncount(0, nc); fse_build(PlainTab{p.ll}, nc);
  predefined_ncount(1, nc); fse_build(PlainTab{p.of}, nc);
  predefined_ncount(2, nc); fse_build(PlainTab{p.ml}, nc);
}

// ---- the lane's state ----------------------------------------------------------------------------------------------------------
// Store: where the lane keeps its tables.
//   ws_get(i) / ws_set(i, v)    32-bit entries, i < kWorkEntries (LL | OF | ML sequence tables; scratch while the Huffman
//                               weights are decoded)
//   sym_get(i) / sym_set(i, v)  the 256 Huffman symbols in rank order
//   cls_get(j) / cls_set(j, v)  the 11 weight classes of the Huffman code: first code position (15-bit aligned) | first rank << 16
//   run_get(k) / run_set(k, v)  the block's sequences, k < kFastRuns: literal length | (match length - 3) << 16
template <class Store>
struct WsTab {
  Store* s;
  int base;
  LSR_HD uint32_t get(int i) const { return s->ws_get(base + i); }
  LSR_HD void set(int i, uint32_t v) const { s->ws_set(base + i, v); }
};

struct HufCode {         // canonical code of the literals: classes by weight 1 .. 11 (weight w <-> length max_bits + 1 - w)
  uint16_t end[12];      // end[w]: first code-space position (max_bits wide) past the symbols of weight <= w
  uint16_t rank_end[12]; // symbols of weight <= w
  uint32_t pair[5];      // end[1 .. 10] aligned to 15 bits, two per dword: what the fast path compares against (huf_class)
  int max_bits;          // 0 = no table yet
};

struct SeqTable {        // one of LL / OF / ML as currently in force
  int mode;              // 0 predefined, 1 RLE, 2 in the workspace
  int log;
  int rle_symbol;
};

template <class Store>
struct Lane {
  Store store;
  const Predefined* pre;
  HufCode huf;
  SeqTable tab[3];
  uint32_t rep[3];
  bool have_tab[3];
};

template <class Store>
LSR_HD void lane_reset(Lane<Store>& L) {
  L.huf.max_bits = 0;
  L.rep[0] = 1; L.rep[1] = 4; L.rep[2] = 8;
  for (int i = 0; i < 3; ++i) { L.have_tab[i] = false; L.tab[i].mode = 0; L.tab[i].log = 0; L.tab[i].rle_symbol = 0; }
}

// ---- Huffman tree description ----------------------------------------------------------------------------------------------------
// Weights -> HufCode + symbols in rank order.  Returns bytes of the description, or an error.
template <class Store>
LSR_HD int weight_get(Store& st, int i) {
  return static_cast<int>((st.ws_get(kWeightBase + (i >> 3)) >> (4 * (i & 7))) & 15);
}
template <class Store>
LSR_HD void weight_set(Store& st, int i, uint32_t w) {
  const int e = kWeightBase + (i >> 3), sh = 4 * (i & 7);
  const uint32_t old = (i & 7) ? st.ws_get(e) : 0;          // weights are written in order: entry starts empty
  st.ws_set(e, (old & ~(15u << sh)) | (w & 15u) << sh);
}

template <class Store>
LSR_HD_INL int read_huf_description(Lane<Store>& L, const uint8_t* p, int len) {
  if (len < 1) return kErrCorrupt;
  const int hb = p[0];
  int used, nw;
  if (hb >= 128) {
    nw = hb - 127;
    used = 1 + (nw + 1) / 2;
    if (used > len) return kErrCorrupt;
    for (int i = 0; i < nw; ++i) {
      const int byte = p[1 + i / 2];
      weight_set(L.store, i, static_cast<uint32_t>((i & 1) ? (byte & 15) : (byte >> 4)));
    }
  } else {
    used = 1 + hb;
    if (hb < 1 || used > len) return kErrCorrupt;
    NCount nc;
    const int h = read_ncount(p + 1, hb, 255, 6, nc);
    if (h < 0) return h;
    if (nc.max_symbol > 12) return kErrCorrupt;
    WsTab<Store> t{&L.store, kWeightTabBase};
    fse_build(t, nc);
    BackBits b;
    if (!back_init(b, p + 1 + h, hb - h)) return kErrCorrupt;
    uint32_t s1 = back_read(b, nc.log), s2 = back_read(b, nc.log);
    if (b.pos < 0) return kErrCorrupt;
    nw = 0;
    for (;;) {
      // two interleaved states; the stream is used up exactly when the last two weights sit in the states
      const uint32_t e1 = t.get(static_cast<int>(s1));
      if (nw >= 255) return kErrCorrupt;
      if ((e1 & 0xFF) > 11) return kErrCorrupt;
      weight_set(L.store, nw++, e1 & 0xFF);
      s1 = (e1 >> 16) + back_read(b, static_cast<int>((e1 >> 8) & 0xFF));
      if (b.pos < 0) {
        if (nw >= 255) return kErrCorrupt;
        weight_set(L.store, nw++, t.get(static_cast<int>(s2)) & 0xFF);
        break;
      }
      const uint32_t e2 = t.get(static_cast<int>(s2));
      if (nw >= 255) return kErrCorrupt;
      if ((e2 & 0xFF) > 11) return kErrCorrupt;
      weight_set(L.store, nw++, e2 & 0xFF);
      s2 = (e2 >> 16) + back_read(b, static_cast<int>((e2 >> 8) & 0xFF));
      if (b.pos < 0) {
        if (nw >= 255) return kErrCorrupt;
        weight_set(L.store, nw++, t.get(static_cast<int>(s1)) & 0xFF);
        break;
      }
    }
  }
  // the implied last weight completes the sum of 2^(w-1) to a power of two
  uint32_t total = 0;
  int per_weight[13];
  for (int w = 0; w < 13; ++w) per_weight[w] = 0;
  for (int i = 0; i < nw; ++i) {
    const int w = weight_get(L.store, i);
    if (w > 11) return kErrCorrupt;
    ++per_weight[w];
    if (w) total += 1u << (w - 1);
  }
  if (total == 0) return kErrCorrupt;
  const int max_bits = zs::highbit(total) + 1;
  if (max_bits > 11) return kErrCorrupt;
  const uint32_t rest = (1u << max_bits) - total;
  if (rest == 0 || (rest & (rest - 1))) return kErrCorrupt;
  const int last_w = zs::highbit(rest) + 1;
  weight_set(L.store, nw, static_cast<uint32_t>(last_w));
  ++per_weight[last_w];
  ++nw;
  if (per_weight[1] < 2 || (per_weight[1] & 1)) return kErrCorrupt;
  // rank order: weight ascending (longest codes first), symbol ascending inside a weight
  int rank_start[13];
  uint32_t code_pos = 0;
  int rank = 0;
  L.huf.end[0] = 0;
  L.huf.rank_end[0] = 0;
  for (int w = 1; w <= 11; ++w) {
    rank_start[w] = rank;
    rank += per_weight[w];
    code_pos += static_cast<uint32_t>(per_weight[w]) << (w - 1);
    L.huf.end[w] = static_cast<uint16_t>(code_pos);
    L.huf.rank_end[w] = static_cast<uint16_t>(rank);
  }
  for (int i = 0; i < nw; ++i) {
    const int w = weight_get(L.store, i);
    if (w) L.store.sym_set(rank_start[w]++, static_cast<uint8_t>(i));
  }
  L.huf.max_bits = max_bits;
  // the fast path's view of the same code: class bounds aligned to 15 bits (end[w] <= 2^max_bits becomes <= 0x8000), and
  // per class the pair (first code position, first rank) where the store keeps it for an indexed read
  const int up = 15 - max_bits;
  for (int q = 0; q < 5; ++q)
    L.huf.pair[q] = (static_cast<uint32_t>(L.huf.end[2 * q + 1]) << up) | ((static_cast<uint32_t>(L.huf.end[2 * q + 2]) << up) << 16);
  for (int j = 0; j <= 10; ++j)
    L.store.cls_set(j, (static_cast<uint32_t>(L.huf.end[j]) << up) | (static_cast<uint32_t>(L.huf.rank_end[j]) << 16));
  return used;
}

// one literal from the top of the stream
template <class Store>
LSR_HD uint8_t huf_decode_one(Lane<Store>& L, BackBits& b) {
  const HufCode& h = L.huf;
  const uint32_t v = back_peek(b, h.max_bits);
  int w = 1;
  uint32_t start = 0, rstart = 0;
#pragma unroll
  for (int k = 1; k <= 10; ++k) {
    const bool ge = v >= h.end[k];
    start = ge ? h.end[k] : start;
    rstart = ge ? h.rank_end[k] : rstart;
    w += ge;
  }
  b.pos -= h.max_bits + 1 - w;
  return L.store.sym_get(static_cast<int>(rstart + ((v - start) >> (w - 1))));
}

template <class Store>
LSR_HD int huf_decode_stream(Lane<Store>& L, const uint8_t* p, int len, uint8_t* out, int n) {
  BackBits b;
  if (!back_init(b, p, len)) return kErrCorrupt;
  for (int i = 0; i < n; ++i) out[i] = huf_decode_one(L, b);
  return b.pos == 0 ? kOk : kErrCorrupt;
}

// ---- the fast path of the four-stream decoder ------------------------------------------------------------------------------
// A lane that waits for every 8-byte window and stores every literal as a byte spends its time in memory latency
// (measured: 70 000 dependent global operations per 32 KB block, ~1 000 cycles each).  Here each stream keeps 128 bits
// of its tail in two registers plus the next 64 already on their way (loaded one batch before they are needed), a batch
// decodes four symbols per stream from one 64-bit extract without touching memory, and the four literals leave as one
// dword.  The slow path above finishes the last symbols of every stream (and takes streams shorter than 24 bytes).
struct FastStream {
  const uint8_t* p;
  int pos;            // unread bits
  int base;           // stream bit index of lo's bit 0
  uint64_t hi, lo, nxt;
};
LSR_HD bool fast_init(FastStream& f, const BackBits& b) {
  const int top = (b.pos + 7) >> 3;           // bytes that still hold unread bits
  if (top < 24 || b.pos < 64) return false;
  f.p = b.p;
  f.pos = b.pos;
  f.base = 8 * (top - 16);
  f.hi = load64(b.p + top - 8);
  f.lo = load64(b.p + top - 16);
  f.nxt = load64(b.p + top - 24);
  return true;
}
// Before a batch: make sure the unread tail starts in `hi`.  False when the next window would lie before the stream.
LSR_HD bool fast_step(FastStream& f) {
  if (f.pos - f.base > 64) return true;
  if (f.base < 128) return false;
  f.hi = f.lo;
  f.lo = f.nxt;
  f.base -= 64;
  f.nxt = load64(f.p + (f.base >> 3) - 8);
  return true;
}
LSR_HD uint64_t fast_extract(const FastStream& f) {      // stream bits [pos - 64, pos)
  const int sh = f.pos - f.base - 64;                     // 1 .. 64
  return sh >= 64 ? f.hi : ((f.hi << (64 - sh)) | (f.lo >> sh));
}

// the canonical search on a value already aligned to max_bits; returns the symbol's rank and its length
LSR_HD void huf_rank(const HufCode& h, uint32_t v, uint32_t& rank, int& length) {
  int w = 1;
  uint32_t start = 0, rstart = 0;
#pragma unroll
  for (int k = 1; k <= 10; ++k) {
    const bool ge = v >= h.end[k];
    start = ge ? h.end[k] : start;
    rstart = ge ? h.rank_end[k] : rstart;
    w += ge;
  }
  rank = rstart + ((v - start) >> (w - 1));
  length = h.max_bits + 1 - w;
}

// The same search for the fast path, on the top 15 bits of the stream.  The ten comparisons `v >= end[k]` run two to a
// dword: with bit 15 of each half set in the minuend, the halves of `v2 - pair` cannot borrow from each other and keep
// that bit exactly where v >= end (both operands are at most 0x8000); the flags are collected in one word and counted.
// 16 instructions instead of 40, and the literal's LENGTH -- the only thing the next literal of the stream waits for --
// is known before any table is read: class bounds and symbol come from the store behind it.
LSR_HD int huf_class(const HufCode& h, uint32_t v15) {       // symbols' weight class - 1 = number of bounds at or below v
  const uint32_t v2 = v15 * 0x10001u + 0x80008000u;
  uint32_t m = 0;
#pragma unroll
  for (int q = 0; q < 5; ++q) m = (m >> 1) | ((v2 - h.pair[q]) & 0x80008000u);
#if defined(__HIP_DEVICE_COMPILE__)
  return __popc(m);
#else
  return __builtin_popcount(m);
#endif
}

// Four literals of one stream in three steps, so that the caller can run each step for all four streams before the
// next: (a) classes and lengths -- pure arithmetic on the window, the only serial chain -- with the four class entries
// requested from the store; (b) ranks, with the four symbols requested; (c) the symbols packed, first in the low byte.
// Written as one function the compiler waited for every table read right behind its request (one exposed LDS round trip
// per literal); in steps, sixteen reads are in flight per wait.
struct FourLits {
  uint32_t v15[4], c[4];
  int shift[4];          // 15 - length
  uint8_t sym[4];
};
template <class Store>
LSR_HD void four_classes(Lane<Store>& L, FastStream& f, FourLits& q) {
  uint64_t cur = fast_extract(f);
  const int mb = L.huf.max_bits;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    q.v15[k] = static_cast<uint32_t>(cur >> 49);
    const int j = huf_class(L.huf, q.v15[k]);
    const int length = mb - j;
    q.c[k] = L.store.cls_get(j);
    q.shift[k] = 15 - length;
    cur <<= length;
    f.pos -= length;
  }
}
template <class Store>
LSR_HD void four_symbols(Lane<Store>& L, FourLits& q) {
#pragma unroll
  for (int k = 0; k < 4; ++k)
    q.sym[k] = L.store.sym_get(static_cast<int>((q.c[k] >> 16) + ((q.v15[k] - (q.c[k] & 0xFFFFu)) >> q.shift[k])));
}
LSR_HD uint32_t four_packed(const FourLits& q) {
  return q.sym[0] | (static_cast<uint32_t>(q.sym[1]) << 8) | (static_cast<uint32_t>(q.sym[2]) << 16) | (static_cast<uint32_t>(q.sym[3]) << 24);
}

LSR_HD void store32(uint8_t* p, uint32_t v) { __builtin_memcpy(p, &v, 4); }

// Where decoded literals go.  PlainSink: literal i at out[i] (the literal buffer).  The direct sink further down writes
// every literal at its final place in the block.
struct Bytes16 { uint64_t a, b; };
LSR_HD Bytes16 load16(const uint8_t* p) { Bytes16 v; __builtin_memcpy(&v, p, 16); return v; }
LSR_HD void store16(uint8_t* p, const Bytes16& v) { __builtin_memcpy(p, &v, 16); }

struct PlainSink {
  uint8_t* out;
  LSR_HD void put16(int, int i, const Bytes16& v) { store16(out + i, v); }
  LSR_HD void put1(int, int i, uint8_t v) { out[i] = v; }
};

// four streams side by side: four independent dependency chains per lane.  Literal i of stream s is literal s * q + i
// of the section; sink.put4(s, index, four literals) / sink.put1(s, index, literal) receive them in increasing order
// per stream.
template <class Store, class Sink>
LSR_HD int huf_decode_4(Lane<Store>& L, const uint8_t* p, int len, int n, Sink& sink) {
  if (len < 10) return kErrCorrupt;
  const int s0 = static_cast<int>(load_le(p, 2)), s1 = static_cast<int>(load_le(p + 2, 2)), s2 = static_cast<int>(load_le(p + 4, 2));
  const int s3 = len - 6 - s0 - s1 - s2;
  if (s3 < 1 || s0 < 1 || s1 < 1 || s2 < 1) return kErrCorrupt;
  const int q = (n + 3) / 4;
  const int n3 = n - 3 * q;
  if (n3 < 0) return kErrCorrupt;
  BackBits b0, b1, b2, b3;
  const uint8_t* s = p + 6;
  if (!back_init(b0, s, s0) || !back_init(b1, s + s0, s1) || !back_init(b2, s + s0 + s1, s2) ||
      !back_init(b3, s + s0 + s1 + s2, s3))
    return kErrCorrupt;
  int i = 0;
  {
    FastStream f0, f1, f2, f3;
    if (fast_init(f0, b0) && fast_init(f1, b1) && fast_init(f2, b2) && fast_init(f3, b3)) {
      // A group = four batches of four literals per stream: 16 literals leave as ONE 16-byte store per stream.  Stores
      // and loads share the wave's memory counter in issue order, so every wait for a window load also waits for the
      // stores in front of it -- ~5 us each for 64 lanes on 64 cache lines; one store per 16 literals instead of per 4
      // is what that costs less (measured: DESIGN.md section 5).  A group needs 16 more literals in every stream,
      // 176 unread bits, and room for three window steps.
      while (i + 16 <= n3 && f0.pos >= 200 && f1.pos >= 200 && f2.pos >= 200 && f3.pos >= 200 &&
             f0.base >= 320 && f1.base >= 320 && f2.base >= 320 && f3.base >= 320) {
        uint32_t g0[4], g1[4], g2[4], g3[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          fast_step(f0); fast_step(f1); fast_step(f2); fast_step(f3);
          FourLits q0, q1, q2, q3;
#if defined(LSR_DEC_FUSED_LOOKUP)
          four_classes(L, f0, q0); four_symbols(L, q0); g0[k] = four_packed(q0);
          four_classes(L, f1, q1); four_symbols(L, q1); g1[k] = four_packed(q1);
          four_classes(L, f2, q2); four_symbols(L, q2); g2[k] = four_packed(q2);
          four_classes(L, f3, q3); four_symbols(L, q3); g3[k] = four_packed(q3);
#else
          four_classes(L, f0, q0); four_classes(L, f1, q1); four_classes(L, f2, q2); four_classes(L, f3, q3);
          four_symbols(L, q0); four_symbols(L, q1); four_symbols(L, q2); four_symbols(L, q3);
          g0[k] = four_packed(q0);
          g1[k] = four_packed(q1);
          g2[k] = four_packed(q2);
          g3[k] = four_packed(q3);
#endif
        }
        sink.put16(0, i, Bytes16{g0[0] | static_cast<uint64_t>(g0[1]) << 32, g0[2] | static_cast<uint64_t>(g0[3]) << 32});
        sink.put16(1, q + i, Bytes16{g1[0] | static_cast<uint64_t>(g1[1]) << 32, g1[2] | static_cast<uint64_t>(g1[3]) << 32});
        sink.put16(2, 2 * q + i, Bytes16{g2[0] | static_cast<uint64_t>(g2[1]) << 32, g2[2] | static_cast<uint64_t>(g2[3]) << 32});
        sink.put16(3, 3 * q + i, Bytes16{g3[0] | static_cast<uint64_t>(g3[1]) << 32, g3[2] | static_cast<uint64_t>(g3[3]) << 32});
        i += 16;
      }
      b0.pos = f0.pos; b1.pos = f1.pos; b2.pos = f2.pos; b3.pos = f3.pos;
      back_refill(b0); back_refill(b1); back_refill(b2); back_refill(b3);
    }
  }
  for (; i < n3; ++i) {
    sink.put1(0, i, huf_decode_one(L, b0));
    sink.put1(1, q + i, huf_decode_one(L, b1));
    sink.put1(2, 2 * q + i, huf_decode_one(L, b2));
    sink.put1(3, 3 * q + i, huf_decode_one(L, b3));
  }
  for (i = n3 > i ? n3 : i; i < q; ++i) {
    sink.put1(0, i, huf_decode_one(L, b0));
    sink.put1(1, q + i, huf_decode_one(L, b1));
    sink.put1(2, 2 * q + i, huf_decode_one(L, b2));
  }
  return (b0.pos == 0 && b1.pos == 0 && b2.pos == 0 && b3.pos == 0) ? kOk : kErrCorrupt;
}

// ---- literals section ----------------------------------------------------------------------------------------------------------
struct Literals {
  const uint8_t* ptr;   // where the literals are (source for Raw, destination tail for Huffman); unused for RLE
  int size;
  bool rle;
  uint8_t rle_value;
};

struct LitHeader {
  int type;               // 0 Raw, 1 RLE, 2 Huffman with a tree description, 3 Huffman with the previous tree
  bool four;              // four streams
  int size;               // literals of the section
  const uint8_t* body;    // Raw: the literals; RLE: the value; Huffman: the streams (behind the description)
  int body_len;
};

// The header of the literals section (a Huffman tree description is read, nothing is decoded).  Returns the bytes the
// whole section takes.
template <class Store>
LSR_HD int parse_literals(Lane<Store>& L, const uint8_t* p, int len, LitHeader& h) {
  if (len < 1) return kErrCorrupt;
  const int type = p[0] & 3, fmt = (p[0] >> 2) & 3;
  h.type = type;
  h.four = false;
  if (type < 2) {                          // Raw / RLE
    int hs, size;
    if ((fmt & 1) == 0) { hs = 1; size = p[0] >> 3; }
    else if (fmt == 1) { if (len < 2) return kErrCorrupt; hs = 2; size = static_cast<int>(load_le(p, 2) >> 4); }
    else { if (len < 3) return kErrCorrupt; hs = 3; size = static_cast<int>(load_le(p, 3) >> 4); }
    if (size > kBlockMax) return kErrCorrupt;
    h.size = size;
    h.body = p + hs;
    h.body_len = type == 0 ? size : 1;
    if (hs + h.body_len > len) return kErrCorrupt;
    return hs + h.body_len;
  }
  int hs, regen, csize;
  bool four = true;
  if (fmt < 2) {
    if (len < 3) return kErrCorrupt;
    const uint32_t v = static_cast<uint32_t>(load_le(p, 3));
    hs = 3; regen = (v >> 4) & 0x3FF; csize = (v >> 14) & 0x3FF; four = fmt == 1;
  } else if (fmt == 2) {
    if (len < 4) return kErrCorrupt;
    const uint32_t v = static_cast<uint32_t>(load_le(p, 4));
    hs = 4; regen = (v >> 4) & 0x3FFF; csize = v >> 18;
  } else {
    if (len < 5) return kErrCorrupt;
    const uint64_t v = load_le(p, 5);
    hs = 5; regen = static_cast<int>((v >> 4) & 0x3FFFF); csize = static_cast<int>(v >> 22);
  }
  if (regen > kBlockMax || regen < 1 || hs + csize > len) return kErrCorrupt;
  h.body = p + hs;
  h.body_len = csize;
  if (type == 2) {
    const int d = read_huf_description(L, h.body, h.body_len);
    if (d < 0) return d;
    h.body += d;
    h.body_len -= d;
  } else if (L.huf.max_bits == 0) {
    return kErrCorrupt;                    // "treeless" without a previous tree
  }
  h.four = four;
  h.size = regen;
  return hs + csize;
}

// The literals as a buffer: Raw ones stay in the source; Huffman ones are decoded to op + room' - size where room' =
// min(room, kBlockMax) -- the block's output cannot overtake them (see the header of this file).
template <class Store>
LSR_HD int literals_in_place(Lane<Store>& L, const LitHeader& h, uint8_t* op, int room, Literals& lit) {
  lit.size = h.size;
  lit.rle = h.type == 1;
  lit.rle_value = h.type == 1 ? h.body[0] : 0;
  lit.ptr = h.body;
  if (h.type < 2) return kOk;
  const int span = room < kBlockMax ? room : kBlockMax;
  if (h.size > span) return kErrDstSmall;
  uint8_t* dst = op + span - h.size;
  PlainSink sink{dst};
  const int rc = h.four ? huf_decode_4(L, h.body, h.body_len, h.size, sink) : huf_decode_stream(L, h.body, h.body_len, dst, h.size);
  if (rc < 0) return rc;
  lit.ptr = dst;
  return kOk;
}

// ---- sequences --------------------------------------------------------------------------------------------------------------------
template <class Store>
LSR_HD uint32_t seq_entry(Lane<Store>& L, int which, uint32_t state) {
  const SeqTable& t = L.tab[which];
  if (t.mode == 0) return which == 0 ? L.pre->ll[state] : which == 1 ? L.pre->of[state] : L.pre->ml[state];
  if (t.mode == 1) return fse_entry(t.rle_symbol, 0, 0);
  return L.store.ws_get((which == 0 ? kLLBase : which == 1 ? kOFBase : kMLBase) + static_cast<int>(state));
}

// table `which` in `mode` (0 predefined, 1 RLE, 2 FSE, 3 repeat); returns the bytes used
template <class Store>
LSR_HD int read_seq_table(Lane<Store>& L, int which, int mode, const uint8_t* p, int len) {
  const int max_sym = which == 0 ? 35 : which == 1 ? 31 : 52;
  const int max_log = which == 0 ? kLLLogMax : which == 1 ? kOFLogMax : kMLLogMax;
  SeqTable& t = L.tab[which];
  if (mode == 0) {
    t.mode = 0;
    t.log = which == 1 ? 5 : 6;
    L.have_tab[which] = true;
    return 0;
  }
  if (mode == 1) {
    if (len < 1 || p[0] > max_sym) return kErrCorrupt;
    t.mode = 1;
    t.log = 0;
    t.rle_symbol = p[0];
    L.have_tab[which] = true;
    return 1;
  }
  if (mode == 3) return L.have_tab[which] ? 0 : kErrCorrupt;
  NCount nc;
  const int used = read_ncount(p, len, max_sym, max_log, nc);
  if (used < 0) return used;
  WsTab<Store> tab{&L.store, which == 0 ? kLLBase : which == 1 ? kOFBase : kMLBase};
  fse_build(tab, nc);
  t.mode = 2;
  t.log = nc.log;
  L.have_tab[which] = true;
  return used;
}

// the first n < 16 bytes of the 128-bit value (a, b), low byte first: at most four stores of 8 / 4 / 2 / 1 bytes
LSR_HD void store_head(uint8_t* d, uint64_t a, uint64_t b, int n) {
  if (n & 8) { __builtin_memcpy(d, &a, 8); d += 8; a = b; }
  if (n & 4) { const uint32_t w = static_cast<uint32_t>(a); __builtin_memcpy(d, &w, 4); d += 4; a >>= 32; }
  if (n & 2) { const uint16_t w = static_cast<uint16_t>(a); __builtin_memcpy(d, &w, 2); d += 2; a >>= 16; }
  if (n & 1) *d = static_cast<uint8_t>(a);
}

// copy n bytes within the destination from `offset` back (may overlap: the pattern repeats).  Nothing past o + n is
// written (the literals behind a match are already in place) and nothing past o - 1 is read unless this copy wrote it.
// No byte loops and no local arrays: the lanes of a wave run this side by side, every load -> store pair is a memory
// round trip for all of them, and a local array is scratch memory on the device.  A rest shorter than a chunk is the
// LAST chunk of the match once more (overlapping what was just written, with the same values) or, for a match shorter
// than one chunk, the head of a chunk-sized load (which ends before o).
LSR_HD void copy_match(uint8_t* o, uint32_t offset, int n) {
  const uint8_t* m = o - offset;
  int i = 0;
  if (offset >= 16) {
    if (offset >= 32) {          // two independent 16-byte loads in flight
      for (; i + 32 <= n; i += 32) {
        const Bytes16 x = load16(m + i), y = load16(m + i + 16);
        store16(o + i, x);
        store16(o + i + 16, y);
      }
    }
    for (; i + 16 <= n; i += 16) store16(o + i, load16(m + i));
    if (i < n) {
      if (n >= 16) {
        store16(o + n - 16, load16(m + n - 16));
      } else {
        const Bytes16 x = load16(m);
        store_head(o, x.a, x.b, n);
      }
    }
    return;
  }
  if (offset >= 8) {
    for (; i + 8 <= n; i += 8) {
      uint64_t v;
      __builtin_memcpy(&v, m + i, 8);
      __builtin_memcpy(o + i, &v, 8);
    }
    if (i < n) {
      uint64_t v;
      if (n >= 8) {
        __builtin_memcpy(&v, m + n - 8, 8);
        __builtin_memcpy(o + n - 8, &v, 8);
      } else {
        __builtin_memcpy(&v, m, 8);
        store_head(o, v, 0, n);
      }
    }
    return;
  }
  // period p < 8: sixteen bytes of the pattern in two registers, stored in strides of the largest multiple of p
  const int p = static_cast<int>(offset);
  uint64_t v = 0;
  for (int k = 0; k < 7; ++k)
    if (k < p) v |= static_cast<uint64_t>(m[k]) << (8 * k);          // (independent byte loads, none past o - 1)
  const int r = p == 3 ? 2 : p == 5 ? 3 : p == 6 ? 2 : p == 7 ? 1 : 0;    // 8 mod p
  uint64_t a = v, b = r ? (v >> (8 * r)) | ((v << (8 * (p - r))) & ((1ull << (8 * p)) - 1ull)) : v;   // b: the pattern from byte 8 on
  for (int len = p; len < 8; len *= 2) {
    a |= a << (8 * len);
    b |= b << (8 * len);
  }
  const int step = p == 3 || p == 5 ? 15 : p == 6 ? 12 : p == 7 ? 14 : 16;
  const Bytes16 pat{a, b};
  for (; i + 16 <= n; i += step) store16(o + i, pat);
  if (i < n) store_head(o + i, a, b, n - i);
}

LSR_HD void copy_forward(uint8_t* o, const uint8_t* s, int n) {   // s >= o or disjoint: exact length, 16 bytes at a time
  int i = 0;
  for (; i + 32 <= n; i += 32) {          // (both loads before the stores: s >= o, so no store lands on unread source)
    const Bytes16 x = load16(s + i), y = load16(s + i + 16);
    store16(o + i, x);
    store16(o + i + 16, y);
  }
  for (; i + 8 <= n; i += 8) {
    uint64_t v;
    __builtin_memcpy(&v, s + i, 8);
    __builtin_memcpy(o + i, &v, 8);
  }
  for (; i < n; ++i) o[i] = s[i];
}

// The sequences section: header, tables, then the sequences one by one.
struct SeqReader {
  BackBits b;
  uint32_t s_ll, s_of, s_ml;
  int nseq, i;
};

template <class Store>
LSR_HD int seq_open(Lane<Store>& L, const uint8_t* p, int len, SeqReader& r) {
  if (len < 1) return kErrCorrupt;
  int nseq = p[0], hs = 1;
  if (nseq >= 128) {
    if (nseq == 255) {
      if (len < 3) return kErrCorrupt;
      nseq = static_cast<int>(load_le(p + 1, 2)) + 0x7F00;
      hs = 3;
    } else {
      if (len < 2) return kErrCorrupt;
      nseq = ((nseq - 128) << 8) + p[1];
      hs = 2;
    }
  }
  r.nseq = nseq;
  r.i = 0;
  if (nseq == 0) return hs == len ? kOk : kErrCorrupt;
  if (hs + 1 > len) return kErrCorrupt;
  const int modes = p[hs];
  if (modes & 3) return kErrCorrupt;
  int at = hs + 1;
  for (int which = 0; which < 3; ++which) {
    const int mode = (modes >> (6 - 2 * which)) & 3;
    const int used = read_seq_table(L, which, mode, p + at, len - at);
    if (used < 0) return used;
    at += used;
  }
  if (!back_init(r.b, p + at, len - at)) return kErrCorrupt;
  r.s_ll = back_read(r.b, L.tab[0].log);
  r.s_of = back_read(r.b, L.tab[1].log);
  r.s_ml = back_read(r.b, L.tab[2].log);
  return r.b.pos < 0 ? kErrCorrupt : kOk;
}

// the next sequence (repeat offsets resolved); after the last one the stream must be used up exactly
template <class Store>
LSR_HD int seq_next(Lane<Store>& L, SeqReader& r, int& lit_len, int& match_len, uint32_t& offset) {
  BackBits& b = r.b;
  const uint32_t e_ll = seq_entry(L, 0, r.s_ll), e_of = seq_entry(L, 1, r.s_of), e_ml = seq_entry(L, 2, r.s_ml);
  const int of_code = static_cast<int>(e_of & 0xFF);
  if (of_code > 31) return kErrCorrupt;
  const uint32_t offset_value = (1u << of_code) + back_read(b, of_code);
  uint32_t ml_base, ll_base;
  int ml_bits, ll_bits;
  ml_code(static_cast<int>(e_ml & 0xFF), ml_base, ml_bits);
  ll_code(static_cast<int>(e_ll & 0xFF), ll_base, ll_bits);
  match_len = static_cast<int>(ml_base + back_read(b, ml_bits));
  lit_len = static_cast<int>(ll_base + back_read(b, ll_bits));
  if (++r.i < r.nseq) {
    r.s_ll = (e_ll >> 16) + back_read(b, static_cast<int>((e_ll >> 8) & 0xFF));
    r.s_ml = (e_ml >> 16) + back_read(b, static_cast<int>((e_ml >> 8) & 0xFF));
    r.s_of = (e_of >> 16) + back_read(b, static_cast<int>((e_of >> 8) & 0xFF));
    if (b.pos < 0) return kErrCorrupt;
  } else if (b.pos != 0) {
    return kErrCorrupt;
  }
  if (offset_value > 3) {
    offset = offset_value - 3;
    L.rep[2] = L.rep[1]; L.rep[1] = L.rep[0]; L.rep[0] = offset;
  } else {
    const uint32_t idx = offset_value + (lit_len == 0 ? 1 : 0);
    if (idx == 1) {
      offset = L.rep[0];
    } else {
      offset = idx == 4 ? L.rep[0] - 1 : L.rep[idx - 1];
      if (offset == 0) return kErrCorrupt;
      if (idx != 2) L.rep[2] = L.rep[1];
      L.rep[1] = L.rep[0];
      L.rep[0] = offset;
    }
  }
  return kOk;
}

// The general execution: literals from a buffer, sequence by sequence.  `dst0`: start of the frame's output (matches may
// reach back to it), `op`: where this block's output starts, `room`: bytes available from op.  Returns the bytes produced.
template <class Store>
LSR_HD int run_sequences(Lane<Store>& L, SeqReader& r, const Literals& lit, uint8_t* dst0, uint8_t* op, int room) {
  uint8_t* o = op;
  uint8_t* const oend = op + room;
  int lit_at = 0;
  for (int i = 0; i < r.nseq; ++i) {
    int lit_len, match_len;
    uint32_t offset;
    const int rc = seq_next(L, r, lit_len, match_len, offset);
    if (rc < 0) return rc;
    if (lit_len > lit.size - lit_at) return kErrCorrupt;
    if (lit_len + match_len > oend - o) return kErrDstSmall;
    if (lit.rle) { for (int k = 0; k < lit_len; ++k) o[k] = lit.rle_value; }
    else copy_forward(o, lit.ptr + lit_at, lit_len);
    o += lit_len;
    lit_at += lit_len;
    if (offset > static_cast<uint32_t>(o - dst0)) return kErrCorrupt;
    copy_match(o, offset, match_len);
    o += match_len;
  }
  const int rest = lit.size - lit_at;
  if (rest > oend - o) return kErrDstSmall;
  if (lit.rle) { for (int k = 0; k < rest; ++k) o[k] = lit.rle_value; }
  else if (o != lit.ptr + lit_at) copy_forward(o, lit.ptr + lit_at, rest);
  o += rest;
  return static_cast<int>(o - op);
}

// ---- the direct execution ----------------------------------------------------------------------------------------------------
// For a block with four Huffman streams and at most kMaxDirectSeq sequences (camera stacks: 20-30 per 32 KB block):
// the sequences are decoded FIRST into the workspace, every literal is then decoded straight to its final place (no
// literal buffer, no copy of 16 KB through memory per block), and the matches are executed in one loop whose trip count
// is the lane's own total -- lanes of a wave whose long matches sit at different sequence numbers do not wait for each
// other sequence by sequence.
template <class Store>
LSR_HD uint32_t run_of(const Store& st, int k) {
  return k < kFastRuns ? st.run_get(k) : st.ws_get(kSeqBase + kMaxDirectSeq + k);
}
template <class Store>
LSR_HD void run_put(Store& st, int k, uint32_t v) {
  if (k < kFastRuns) st.run_set(k, v);
  else st.ws_set(kSeqBase + kMaxDirectSeq + k, v);
}
template <class Store>
struct DirectSink {
  Store* st;
  uint8_t* op;
  int nseq;
  int k[4], run_end[4], shift[4];    // per stream: literals below run_end[s] belong to run k[s] and go to op[i + shift[s]]
  LSR_HD int lit_of(int i) const { return static_cast<int>(run_of(*st, i) & 0xFFFFu); }
  LSR_HD int match_of(int i) const { return static_cast<int>(run_of(*st, i) >> 16) + 3; }
  LSR_HD void seek(int s, int li) {               // the run literal li belongs to
    int kk = 0, sh = 0, end = nseq > 0 ? lit_of(0) : 0x7FFFFFFF;
    while (kk < nseq && li >= end) {
      sh += match_of(kk);
      ++kk;
      end = kk < nseq ? end + lit_of(kk) : 0x7FFFFFFF;
    }
    k[s] = kk; shift[s] = sh; run_end[s] = end;
  }
  LSR_HD void advance(int s) {
    shift[s] += match_of(k[s]);
    ++k[s];
    run_end[s] = k[s] < nseq ? run_end[s] + lit_of(k[s]) : 0x7FFFFFFF;
  }
  LSR_HD void put1(int s, int i, uint8_t v) {
    while (i >= run_end[s]) advance(s);
    op[i + shift[s]] = v;
  }
  // Sixteen literals.  Inside a run they are one store.  Across a run's end (one lane in eighty, hence every other
  // group of a wave) they leave piece by piece, a piece = the literals up to the next end, as at most four stores of
  // 8 / 4 / 2 / 1 bytes: no loop over bytes, and nothing is read from memory on the way (the run list is in LDS) --
  // a global load here would wait for every store the wave has in flight.
  LSR_HD void put16(int s, int i, const Bytes16& v) {
    if (i + 16 <= run_end[s]) {
      store16(op + i + shift[s], v);
      return;
    }
#if defined(LSR_DEC_BYTE_BOUNDARY)
    for (int q = 0; q < 8; ++q) put1(s, i + q, static_cast<uint8_t>(v.a >> (8 * q)));
    for (int q = 0; q < 8; ++q) put1(s, i + 8 + q, static_cast<uint8_t>(v.b >> (8 * q)));
    return;
#endif
    uint64_t a = v.a, b = v.b;
    int pos = 0;
    while (pos < 16) {
      while (i + pos >= run_end[s]) advance(s);
      const int left = run_end[s] - (i + pos);
      const int n = left < 16 - pos ? left : 16 - pos;
      uint8_t* d = op + i + pos + shift[s];
      if (n == 16) {                       // (the previous run ended exactly at i)
        store16(d, Bytes16{a, b});
        return;
      }
      uint64_t x = a;
      if (n & 8) { __builtin_memcpy(d, &x, 8); d += 8; x = b; }
      if (n & 4) { const uint32_t w = static_cast<uint32_t>(x); __builtin_memcpy(d, &w, 4); d += 4; x >>= 32; }
      if (n & 2) { const uint16_t w = static_cast<uint16_t>(x); __builtin_memcpy(d, &w, 2); d += 2; x >>= 16; }
      if (n & 1) *d = static_cast<uint8_t>(x);
      // drop the n bytes from the 128-bit value
      if (n >= 8) { a = b; b = 0; }
      const int sh = 8 * (n & 7);
      if (sh) { a = (a >> sh) | (b << (64 - sh)); b >>= sh; }
      pos += n;
    }
  }
};

// matches of the sequences in the workspace, executed over literals that are already in place
template <class Store>
LSR_HD int execute_matches(Lane<Store>& L, int nseq, uint8_t* dst0, uint8_t* op) {
  uint8_t* o = op;
  for (int k = 0; k < nseq; ++k) {
    const uint32_t run = run_of(L.store, k);
    o += static_cast<int>(run & 0xFFFFu);
    const int n = static_cast<int>(run >> 16) + 3;
    const uint32_t off = L.store.ws_get(kSeqBase + k);
    if (off == 0 || off > static_cast<uint32_t>(o - dst0)) return kErrCorrupt;
    copy_match(o, off, n);
    o += n;
  }
  return static_cast<int>(o - op);
}

// One compressed block.  Returns the bytes produced.
template <class Store>
LSR_HD_INL int decode_block_body(Lane<Store>& L, const uint8_t* p, int size, uint8_t* dst0, uint8_t* op, int room) {
  LSR_DEC_STAMP(0);
  LitHeader h;
  const int lu = parse_literals(L, p, size, h);
  if (lu < 0) return lu;
  LSR_DEC_STAMP(1);
  SeqReader r;
  const int so = seq_open(L, p + lu, size - lu, r);
  if (so < 0) return so;
  LSR_DEC_STAMP(2);
  if (h.type >= 2 && h.four && r.nseq >= 1 && r.nseq <= kMaxDirectSeq && room <= kMaxDirectRoom) {
    // sequences first: lengths and offsets into the workspace, with the totals checked before anything is written
    int lits = 0, total = 0;
    for (int i = 0; i < r.nseq; ++i) {
      int lit_len, match_len;
      uint32_t offset;
      const int rc = seq_next(L, r, lit_len, match_len, offset);
      if (rc < 0) return rc;
      lits += lit_len;
      total += lit_len + match_len;
      if (lits > h.size || total > room) return lits > h.size ? kErrCorrupt : kErrDstSmall;
      // (both lengths are below 65536 now: a match is at least three bytes and the block at most kMaxDirectRoom)
      if (match_len < 3) return kErrCorrupt;
      run_put(L.store, i, static_cast<uint32_t>(lit_len) | (static_cast<uint32_t>(match_len - 3) << 16));
      L.store.ws_set(kSeqBase + i, offset);
    }
    total += h.size - lits;
    if (total > room) return kErrDstSmall;
    LSR_DEC_STAMP(3);
    DirectSink<Store> sink;
    sink.st = &L.store;
    sink.op = op;
    sink.nseq = r.nseq;
    const int q = (h.size + 3) / 4;
    for (int s = 0; s < 4; ++s) sink.seek(s, s * q < h.size ? s * q : h.size);
    const int rc = huf_decode_4(L, h.body, h.body_len, h.size, sink);
    if (rc < 0) return rc;
    LSR_DEC_STAMP(4);
    const int done = execute_matches(L, r.nseq, dst0, op);
    if (done < 0) return done;
    LSR_DEC_STAMP(5);
    return done + (h.size - lits);
  }
  Literals lit;
  const int rc = literals_in_place(L, h, op, room, lit);
  if (rc < 0) return rc;
  return run_sequences(L, r, lit, dst0, op, room);
}

// ---- frame ------------------------------------------------------------------------------------------------------------------------
// One zstd frame src[0, len) -> dst[0, cap).  Returns the decoded size or an error code.
template <class Store>
LSR_HD_INL int decode_frame(Lane<Store>& L, const uint8_t* src, int len, uint8_t* dst, int cap) {
  if (len < 6) return kErrCorrupt;
  if (src[0] != 0x28 || src[1] != 0xB5 || src[2] != 0x2F || src[3] != 0xFD) return kErrUnsupported;
  const int fhd = src[4];
  const int fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, checksum = (fhd >> 2) & 1, dict_flag = fhd & 3;
  if (fhd & 0x08) return kErrCorrupt;                 // reserved bit
  int at = 5;
  if (!single) ++at;                                  // window descriptor: the destination bounds every match anyway
  const int dict_bytes = dict_flag == 3 ? 4 : dict_flag;
  if (at + dict_bytes > len) return kErrCorrupt;
  if (dict_bytes && load_le(src + at, dict_bytes) != 0) return kErrUnsupported;
  at += dict_bytes;
  const int fcs_bytes = fcs_flag == 0 ? single : 1 << fcs_flag;
  if (at + fcs_bytes > len) return kErrCorrupt;
  int64_t content = -1;
  if (fcs_bytes) {
    content = static_cast<int64_t>(load_le(src + at, fcs_bytes));
    if (fcs_bytes == 2) content += 256;
    if (content < 0 || content > cap) return kErrDstSmall;
  }
  at += fcs_bytes;
  lane_reset(L);
  uint8_t* o = dst;
  for (;;) {
    if (at + 3 > len) return kErrCorrupt;
    const uint32_t bh = static_cast<uint32_t>(load_le(src + at, 3));
    at += 3;
    const int last = bh & 1, type = (bh >> 1) & 3, size = static_cast<int>(bh >> 3);
    const int room = static_cast<int>(dst + cap - o);
    if (type == 0) {
      if (at + size > len) return kErrCorrupt;
      if (size > room) return kErrDstSmall;
      copy_forward(o, src + at, size);
      o += size;
      at += size;
    } else if (type == 1) {
      if (at + 1 > len) return kErrCorrupt;
      if (size > room) return kErrDstSmall;
      const uint8_t v = src[at];
      uint64_t pat = v * 0x0101010101010101ull;
      int i = 0;
      for (; i + 8 <= size; i += 8) __builtin_memcpy(o + i, &pat, 8);
      for (; i < size; ++i) o[i] = v;
      o += size;
      at += 1;
    } else if (type == 2) {
      if (size > kBlockMax || at + size > len) return kErrCorrupt;
      const int produced = decode_block_body(L, src + at, size, dst, o, room);
      if (produced < 0) return produced;
      o += produced;
      at += size;
    } else {
      return kErrCorrupt;
    }
    if (last) break;
  }
  if (checksum) at += 4;
  if (at > len) return kErrCorrupt;
  const int64_t total = o - dst;
  if (content >= 0 && total != content) return kErrCorrupt;
  return static_cast<int>(total);
}

}  // namespace zd
}  // namespace lsr
