// The entropy stage of the device-side blosc-zstd writer: Huffman code construction and the zstd "Huffman tree
// description" (RFC 8878 section 4.2.1) for one byte plane of a shuffled block, written so that the SAME functions run
// inside the encode kernel (csrc/blosc_device.hip: the steps marked "per element" are called by one thread per
// element, the serial ones by one lane) and in its host twin (a plain loop) -- the two produce the same bytes, and the
// host twin's bytes are checked against the system libzstd in the GPU-less container (tests/test_device_codec.py).
//
// What is written is a subset of the zstd format, not a restatement of libzstd's compressor: every plane of a block
// becomes one zstd block -- Raw, RLE, or Compressed with Huffman-coded literals in four streams and an empty
// sequences section.  On byte-shuffled float32 / uint16 volumes that is what zstd level 1 (the acquisition's
// setting, shrimpy/mantis/mantis_engine.py:474-481) finds too: the order-0 entropy of the planes is within 1 % of its
// output (profiles/r05_codec_ratio.jsonl).  Any conforming decoder reads the frames.
//
// Code construction (lengths need only be a complete prefix code, not libzstd's): leaves sorted by (count, symbol),
// two-queue Huffman merge, depths limited to 11 bits by moving leaf pairs up (the rule of JPEG's Annex K.3, which
// keeps the Kraft sum at exactly one), lengths handed out by rank, canonical values in the order the zstd decoder
// rebuilds them (longest codes first, symbol order within a length).
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define LSR_HD __host__ __device__ inline
#else
#define LSR_HD inline
#endif

namespace lsr {
namespace zs {

constexpr int kHufMaxBits = 11;        // Max_Number_of_Bits of a zstd literals Huffman code
constexpr int kDepthSlots = 48;        // a plane of < 2^17 symbols cannot make a leaf deeper than 25 (Fibonacci bound)
constexpr int kMaxPlane = 128 * 1024;  // Block_Maximum_Size: one plane is one zstd block
constexpr int kWeightLog = 6;          // accuracy log of the FSE table of the Huffman weights (the format's maximum)
constexpr int kHufHeaderMax = 132;     // header byte + at most 127 bytes of FSE payload / 64 of direct nibbles

// ---- step 1 (per symbol i): position of symbol i among the present symbols sorted by (count, symbol) -------------
// Returns -1 for an absent symbol.  O(256) per call: 256 threads do it side by side on the device.
LSR_HD int huf_rank_of(const uint32_t* count, int i) {
  const uint32_t ci = count[i];
  if (ci == 0) return -1;
  int r = 0;
  for (int j = 0; j < 256; ++j) {
    const uint32_t cj = count[j];
    r += (cj != 0) & ((cj < ci) | ((cj == ci) & (j < i)));
  }
  return r;
}

// ---- step 2 (serial): two-queue merge over the sorted leaves -------------------------------------------------------
// node k < ns is leaf order[k]; nodes ns .. 2 ns - 2 are internal, created in non-decreasing weight; parent[] of every
// node but the root (2 ns - 2).  node_cnt: 2 ns - 1 entries, the first ns filled by the caller with the leaf counts.
LSR_HD void huf_merge(uint32_t* node_cnt, uint16_t* parent, int ns) {
  // the heads of the two queues stay in registers: one pair of (independent) loads per merge instead of four
  // dependent ones -- on the device this loop is ONE lane talking to LDS, ~100 cycles per round trip.  (Keeping the entry
  // BEHIND each head in registers as well, so that a taken head is replaced without a load, was measured: 162 K cycles per
  // plane against 151 K -- the loop is paced by the LDS pipe it shares with the other workgroup's atomics, not by its loads.)
  int leaf = 0, inner = ns, next = ns;
  uint32_t leaf_cnt = node_cnt[0], inner_cnt = 0xFFFFFFFFu;     // an empty inner queue never wins
  for (int k = 0; k < ns - 1; ++k) {
    int pick[2];
    uint32_t sum = 0;
    for (int t = 0; t < 2; ++t) {
      const bool take_leaf = leaf < ns && (inner >= next || leaf_cnt <= inner_cnt);
      if (take_leaf) {
        pick[t] = leaf++;
        sum += leaf_cnt;
        leaf_cnt = leaf < ns ? node_cnt[leaf] : 0xFFFFFFFFu;
      } else {
        pick[t] = inner++;
        sum += inner_cnt;
        inner_cnt = inner < next ? node_cnt[inner] : 0xFFFFFFFFu;
      }
    }
    node_cnt[next] = sum;
    if (inner == next) inner_cnt = sum;          // the node just made is the head of the inner queue
    parent[pick[0]] = static_cast<uint16_t>(next);
    parent[pick[1]] = static_cast<uint16_t>(next);
    ++next;
  }
}

// ---- step 3 (per leaf k): its depth in the tree ---------------------------------------------------------------------
LSR_HD int huf_depth_of(const uint16_t* parent, int ns, int k) {
  const int root = 2 * ns - 2;
  int d = 0;
  for (int n = k; n != root; n = parent[n]) ++d;
  return d < kDepthSlots - 1 ? d : kDepthSlots - 1;
}

// ---- step 4 (serial): limit the depth histogram to kHufMaxBits ------------------------------------------------------
// per_depth[d] = leaves at depth d (d < kDepthSlots).  Two deepest siblings: one takes their parent's place, the other
// becomes the sibling of a leaf taken from the nearest shallower level that has one -- the Kraft sum stays exactly 1.
LSR_HD void huf_limit(uint32_t* per_depth) {
  for (int i = kDepthSlots - 1; i > kHufMaxBits; --i) {
    while (per_depth[i] > 0) {
      int j = i - 2;
      while (j > 0 && per_depth[j] == 0) --j;   // (a complete code always has a shallower leaf)
      per_depth[i] -= 2;
      per_depth[i - 1] += 1;
      per_depth[j + 1] += 2;
      per_depth[j] -= 1;
    }
  }
}

// ---- step 5 (per leaf k): the k-th rarest leaf gets the k-th longest length ------------------------------------------
LSR_HD int huf_length_of_rank(const uint32_t* per_depth, int k) {
  int acc = 0;
  for (int l = kHufMaxBits; l >= 1; --l) {
    acc += per_depth[l];
    if (k < acc) return l;
  }
  return 1;   // not reached for a complete code
}

// ---- step 6 (serial): first code value of every length, longest first (what the zstd decoder's table order implies) --
LSR_HD int huf_first_values(const uint32_t* per_depth, uint16_t* first /* [kHufMaxBits + 1] */) {
  int max_bits = 0;
  for (int l = kHufMaxBits; l >= 1; --l)
    if (per_depth[l]) { max_bits = l; break; }
  unsigned v = 0;
  for (int l = kHufMaxBits; l >= 1; --l) {
    first[l] = static_cast<uint16_t>(v);
    v = (v + (l <= max_bits ? per_depth[l] : 0)) >> 1;
  }
  first[0] = 0;
  return max_bits;
}

// ---- step 7 (per symbol i): value of symbol i = first[len] + symbols below i with the same length ---------------------
// nbits[]: code length of every symbol (0 = absent).  Returns value | length << 16 (0 for an absent symbol).
LSR_HD uint32_t huf_code_of(const uint8_t* nbits, const uint16_t* first, int i) {
  const int l = nbits[i];
  if (l == 0) return 0;
  int below = 0;
  for (int j = 0; j < i; ++j) below += nbits[j] == l;
  return static_cast<uint32_t>(first[l] + below) | static_cast<uint32_t>(l) << 16;
}

// ---- FSE over the Huffman weights (serial) ---------------------------------------------------------------------------
// A small forward bit writer (FSE streams and the normalised-count header are little-endian bit fields).
struct BitOut {
  uint8_t* p;
  int cap, n;          // bytes available / written
  uint64_t acc;
  int bits;
  bool overflow;
};
LSR_HD void bit_init(BitOut& b, uint8_t* p, int cap) { b.p = p; b.cap = cap; b.n = 0; b.acc = 0; b.bits = 0; b.overflow = false; }
LSR_HD void bit_put(BitOut& b, uint32_t v, int nb) {
  b.acc |= static_cast<uint64_t>(v & ((nb >= 32) ? 0xFFFFFFFFu : ((1u << nb) - 1u))) << b.bits;
  b.bits += nb;
  while (b.bits >= 8) {
    if (b.n < b.cap) b.p[b.n] = static_cast<uint8_t>(b.acc); else b.overflow = true;
    ++b.n;
    b.acc >>= 8;
    b.bits -= 8;
  }
}
// the closing '1' and zero padding of a backward-read stream; plain flush for a forward-read header
LSR_HD int bit_close(BitOut& b, bool end_mark) {
  if (end_mark) bit_put(b, 1, 1);
  if (b.bits > 0) bit_put(b, 0, 8 - b.bits);
  return b.overflow ? -1 : b.n;
}

LSR_HD int highbit(uint32_t v) {   // floor(log2 v), v > 0
  int r = 0;
  while (v >>= 1) ++r;
  return r;
}

// Normalised counts of the weight alphabet (0..12, table of 64): each present weight >= 1 and <= 32 -- with no count
// above half the table every state transition reads at least one bit, so the decoder's "until the stream is used up"
// ends on exactly the symbols written.  Needs >= 2 distinct weights (else returns false: caller writes nibbles / Raw).
LSR_HD bool fse_normalise_weights(const int* hist /* [13] */, int total, int* norm /* [13] */, int* max_symbol) {
  int present = 0, sum = 0, top = -1;
  for (int s = 0; s < 13; ++s) {
    norm[s] = 0;
    if (hist[s] == 0) continue;
    ++present;
    top = s;
    int v = (hist[s] * 64 + total / 2) / total;
    v = v < 1 ? 1 : (v > 32 ? 32 : v);
    norm[s] = v;
    sum += v;
  }
  if (present < 2) return false;
  while (sum > 64) {          // take from the largest (lowest symbol first among equals)
    int best = -1;
    for (int s = 0; s <= top; ++s)
      if (norm[s] > 1 && (best < 0 || norm[s] > norm[best])) best = s;
    --norm[best];
    --sum;
  }
  while (sum < 64) {          // give to the symbol most short-changed: largest hist / norm, among those below 32
    int best = -1;
    for (int s = 0; s <= top; ++s) {
      if (norm[s] == 0 || norm[s] >= 32) continue;
      if (best < 0 || static_cast<int64_t>(hist[s]) * norm[best] > static_cast<int64_t>(hist[best]) * norm[s]) best = s;
    }
    ++norm[best];
    ++sum;
  }
  *max_symbol = top;
  return true;
}

// The normalised-count header (RFC 8878 section 4.1.1) of `norm[0..max_symbol]`, table log `log` (counts >= 0 here:
// no "less than one" probabilities are written).
LSR_HD void fse_write_ncount(BitOut& out, const int* norm, int max_symbol, int log) {
  const int table = 1 << log;
  int remaining = table + 1, threshold = table, nb = log + 1;
  bit_put(out, static_cast<uint32_t>(log - 5), 4);
  int s = 0;
  bool prev0 = false;
  const int alphabet = max_symbol + 1;
  while (s < alphabet && remaining > 1) {
    if (prev0) {
      int start = s;
      while (s < alphabet && norm[s] == 0) ++s;
      if (s == alphabet) break;
      while (s >= start + 24) { start += 24; bit_put(out, 0xFFFF, 16); }
      while (s >= start + 3) { start += 3; bit_put(out, 3, 2); }
      bit_put(out, static_cast<uint32_t>(s - start), 2);
    }
    int count = norm[s++];
    const int max = (2 * threshold - 1) - remaining;
    remaining -= count;
    ++count;
    if (count >= threshold) count += max;
    bit_put(out, static_cast<uint32_t>(count), nb - (count < max ? 1 : 0));
    prev0 = count == 1;
    while (remaining < threshold) { --nb; threshold >>= 1; }
  }
}

struct FseEnc {            // encoding table of one distribution over <= 13 symbols, 64 states
  uint16_t next_state[64];
  int delta_bits[13];
  int delta_state[13];
  uint8_t spread[64];
  int fill[14];            // (work space of fse_build_enc)
};

// Work space of huf_write_description: LDS on the device (one per wave), the stack on the host -- as function-local
// arrays they would be per-lane scratch memory on the device, ~500 cycles per access for a single lane
struct HufScratch {
  uint8_t w[256];
  uint8_t tmp[kHufHeaderMax];
  FseEnc enc;
  int hist[13], norm[13];
};

LSR_HD void fse_build_enc(FseEnc& t, const int* norm, int max_symbol, int log) {
  const int table = 1 << log, mask = table - 1, step = (table >> 1) + (table >> 3) + 3;
  uint8_t* const spread = t.spread;
  int* const fill = t.fill;
  int pos = 0, cumul = 0;
  for (int s = 0; s <= max_symbol; ++s) {
    const int n = norm[s];
    fill[s] = cumul;
    cumul += n;
    for (int i = 0; i < n; ++i) {
      spread[pos] = static_cast<uint8_t>(s);
      pos = (pos + step) & mask;
    }
  }
  for (int u = 0; u < table; ++u) t.next_state[fill[spread[u]]++] = static_cast<uint16_t>(table + u);
  int total = 0;
  for (int s = 0; s <= max_symbol; ++s) {
    if (norm[s] == 0) {
      t.delta_bits[s] = ((log + 1) << 16) - table;
      t.delta_state[s] = 0;
    } else if (norm[s] == 1) {
      t.delta_bits[s] = (log << 16) - table;
      t.delta_state[s] = total - 1;
      ++total;
    } else {
      const int max_out = log - highbit(static_cast<uint32_t>(norm[s] - 1));
      t.delta_bits[s] = (max_out << 16) - (norm[s] << max_out);
      t.delta_state[s] = total - norm[s];
      total += norm[s];
    }
  }
}

// FSE-compress `n` weights (n >= 2) with two interleaved states, as the zstd decoder expects them (state 1 carries the
// even positions).  Returns the byte count or -1 when `cap` is too small.
LSR_HD int fse_encode_weights(const FseEnc& t, const uint8_t* w, int n, int log, uint8_t* dst, int cap) {
  BitOut out;
  bit_init(out, dst, cap);
  uint32_t st[2];
  auto init = [&](int which, int sym) {
    const int nb = (t.delta_bits[sym] + (1 << 15)) >> 16;
    const int v = (nb << 16) - t.delta_bits[sym];
    st[which] = t.next_state[(v >> nb) + t.delta_state[sym]];
  };
  int i = n - 1;
  init(i & 1, w[i]); --i;       // position parity picks the state: even -> state "1" (index 0), odd -> state "2"
  init(i & 1, w[i]); --i;
  // Two symbols per trip, one of each state.  On the device this loop is one lane against LDS, and what it waits for is
  // the chain symbol -> its two table entries -> the next state: the two states' chains are independent, so both symbols'
  // entries are read first and the two state lookups follow side by side -- one round trip per pair where the plain loop
  // pays three per symbol.  The bits leave in the plain loop's order.
  for (; i >= 1; i -= 2) {
    const int sa = w[i], sb = w[i - 1];
    const int dba = t.delta_bits[sa], dsa = t.delta_state[sa], dbb = t.delta_bits[sb], dsb = t.delta_state[sb];
    const int wa = i & 1, wb = wa ^ 1;
    const uint32_t a = st[wa], b = st[wb];
    const int nba = static_cast<int>((a + dba) >> 16), nbb = static_cast<int>((b + dbb) >> 16);
    const uint32_t na = t.next_state[(a >> nba) + dsa], nb2 = t.next_state[(b >> nbb) + dsb];
    bit_put(out, a, nba);
    bit_put(out, b, nbb);
    st[wa] = na;
    st[wb] = nb2;
  }
  if (i == 0) {
    const int sym = w[0];
    const int nb = static_cast<int>((st[0] + t.delta_bits[sym]) >> 16);
    bit_put(out, st[0], nb);
    st[0] = t.next_state[(st[0] >> nb) + t.delta_state[sym]];
  }
  bit_put(out, st[1], log);     // state 2 first, state 1 last: the decoder reads state 1 first (from the end)
  bit_put(out, st[0], log);
  return bit_close(out, true);
}

// ---- step 8: the Huffman tree description ---------------------------------------------------------------------------------
// nbits[256] / max_bits as built above.  Three parts, so that the device can run the first and the last with all lanes:
//   (a) huf_weights: weights of symbols 0 .. last - 1 (the last present symbol's is implied) and their histogram;
//   (b) huf_description_body (serial): more than 128 weights -> the FSE form into scratch.tmp, its size (or -1);
//   (c) huf_description_size / huf_description_byte: which form is written, and byte i of it.
// huf_write_description = the three in a row: the description in hdr (kHufHeaderMax bytes), its size, or -1 when the
// weights can be written neither as nibbles (more than 128 of them) nor as an FSE stream below 128 bytes.
LSR_HD int huf_weight_count(const uint8_t* nbits) {
  int last = 255;
  while (last > 0 && nbits[last] == 0) --last;
  return last;
}
LSR_HD void huf_weights(const uint8_t* nbits, int max_bits, int nw, HufScratch& scratch) {
  for (int s = 0; s < 13; ++s) scratch.hist[s] = 0;
  for (int s = 0; s < nw; ++s) {
    scratch.w[s] = nbits[s] ? static_cast<uint8_t>(max_bits + 1 - nbits[s]) : 0;
    ++scratch.hist[scratch.w[s]];
  }
}
LSR_HD int huf_description_body(int nw, HufScratch& scratch) {
  // up to 128 weights fit the nibble form (at most 64 bytes; an FSE stream would save a few dozen of them per plane of
  // tens of kilobytes -- not worth ~250 dependent steps of one lane); more than 128 need the FSE form
  if (nw <= 128) return -1;
  int top = 0;
  if (!fse_normalise_weights(scratch.hist, nw, scratch.norm, &top)) return -1;
  BitOut head;
  bit_init(head, scratch.tmp, 127);
  fse_write_ncount(head, scratch.norm, top, kWeightLog);
  const int hb = bit_close(head, false);
  if (hb <= 0) return -1;
  fse_build_enc(scratch.enc, scratch.norm, top, kWeightLog);
  const int body = fse_encode_weights(scratch.enc, scratch.w, nw, kWeightLog, scratch.tmp + hb, 127 - hb);
  return body > 0 ? hb + body : -1;
}
LSR_HD int huf_description_size(int nw, int fse_size) {      // -1: neither form
  const int direct_size = nw <= 128 ? (nw + 1) / 2 : -1;
  if (fse_size > 0 && fse_size < 128 && (direct_size < 0 || fse_size < direct_size)) return 1 + fse_size;
  return direct_size < 0 ? -1 : 1 + direct_size;
}
LSR_HD uint8_t huf_description_byte(int i, int nw, int fse_size, const HufScratch& scratch) {
  const int direct_size = nw <= 128 ? (nw + 1) / 2 : -1;
  const bool fse = fse_size > 0 && fse_size < 128 && (direct_size < 0 || fse_size < direct_size);
  if (i == 0) return static_cast<uint8_t>(fse ? fse_size : 127 + nw);
  if (fse) return scratch.tmp[i - 1];
  const int k = i - 1;
  const int hi = scratch.w[2 * k], lo = 2 * k + 1 < nw ? scratch.w[2 * k + 1] : 0;
  return static_cast<uint8_t>(hi << 4 | lo);
}
LSR_HD int huf_write_description(const uint8_t* nbits, int max_bits, uint8_t* hdr, HufScratch& scratch) {
  const int nw = huf_weight_count(nbits);
  if (nw < 1) return -1;
  huf_weights(nbits, max_bits, nw, scratch);
  const int fse_size = huf_description_body(nw, scratch);
  const int size = huf_description_size(nw, fse_size);
  for (int i = 0; i < size; ++i) hdr[i] = huf_description_byte(i, nw, fse_size, scratch);
  return size;
}

// ---- layout of one plane's zstd block ------------------------------------------------------------------------------------
enum PlaneMode : int { kPlaneRaw = 0, kPlaneRle = 1, kPlaneHuf = 2 };

// Symbols of the four streams of `n` literals: the first three hold (n + 3) / 4, the last one the rest.
LSR_HD int huf_stream_len(int n, int j) {
  const int q = (n + 3) / 4;
  return j < 3 ? q : n - 3 * q;
}

// Literals_Section_Header of a Huffman-compressed, four-stream section (sizes: regenerated, compressed).
LSR_HD int lit_header(uint8_t* p, int regen, int csize) {
  if (regen < 1024 && csize < 1024) {
    const uint32_t v = 2u | 1u << 2 | static_cast<uint32_t>(regen) << 4 | static_cast<uint32_t>(csize) << 14;
    p[0] = static_cast<uint8_t>(v); p[1] = static_cast<uint8_t>(v >> 8); p[2] = static_cast<uint8_t>(v >> 16);
    return 3;
  }
  if (regen < 16384 && csize < 16384) {
    const uint32_t v = 2u | 2u << 2 | static_cast<uint32_t>(regen) << 4 | static_cast<uint32_t>(csize) << 18;
    p[0] = static_cast<uint8_t>(v); p[1] = static_cast<uint8_t>(v >> 8); p[2] = static_cast<uint8_t>(v >> 16);
    p[3] = static_cast<uint8_t>(v >> 24);
    return 4;
  }
  const uint64_t v = 2u | 3u << 2 | static_cast<uint64_t>(regen) << 4 | static_cast<uint64_t>(csize) << 22;
  for (int i = 0; i < 5; ++i) p[i] = static_cast<uint8_t>(v >> (8 * i));
  return 5;
}
LSR_HD int lit_header_size(int regen, int csize) {
  return (regen < 1024 && csize < 1024) ? 3 : (regen < 16384 && csize < 16384) ? 4 : 5;
}

LSR_HD void block_header(uint8_t* p, int type /* 0 raw, 1 rle, 2 compressed */, int size, bool last) {
  const uint32_t v = static_cast<uint32_t>(last) | static_cast<uint32_t>(type) << 1 | static_cast<uint32_t>(size) << 3;
  p[0] = static_cast<uint8_t>(v); p[1] = static_cast<uint8_t>(v >> 8); p[2] = static_cast<uint8_t>(v >> 16);
}

// zstd frame header of a single-segment frame with a four-byte content size: 9 bytes
constexpr int kFrameHeader = 9;
LSR_HD void frame_header(uint8_t* p, uint32_t content) {
  p[0] = 0x28; p[1] = 0xB5; p[2] = 0x2F; p[3] = 0xFD;
  p[4] = 0xA0;   // Frame_Content_Size_flag 2, Single_Segment_flag 1, no checksum, no dictionary
  p[5] = static_cast<uint8_t>(content); p[6] = static_cast<uint8_t>(content >> 8);
  p[7] = static_cast<uint8_t>(content >> 16); p[8] = static_cast<uint8_t>(content >> 24);
}

// A plane is worth a Huffman block when the (upper bound of the) coded size saves at least 1/64 of it.
LSR_HD bool huf_pays(int plane_len, int64_t payload_bits, int desc_size) {
  const int64_t upper = desc_size + 6 + (payload_bits + 7) / 8 + 4 + 5 + 1;   // + jump table, stream round-ups, literals header, sequences byte
  return upper < plane_len - plane_len / 64;
}

// A plane whose collision entropy -log2(sum p^2) -- a lower bound of its Shannon entropy -- is at least 7.875 bits cannot
// save 1/64 with any prefix code: no tree is built for it (the noise planes of float32 / uint16 data).
// sum_sq = sum of count^2 over the 256 symbols, n = their sum:  sum p^2 <= 2^-7.875  <=>  235 * sum_sq <= n^2 (rounded safe).
LSR_HD bool huf_hopeless(uint64_t sum_sq, uint64_t n) { return 235ull * sum_sq <= n * n; }
// The same test on a SAMPLE of the plane, before the plane is counted at all: one 16-byte vector of the block in every
// kSampleEvery (an unbiased collision count: sum c (c - 1) against n (n - 1)).  Planes of at least kSampleMinPlane symbols
// only: 4096 sampled symbols put a uniform plane's estimate within 1 % of 1 / 256, nine per cent below the threshold.
constexpr int kSampleEvery = 16, kSampleMinPlane = 32768;
LSR_HD bool huf_hopeless_sample(uint64_t sum_c_cm1, uint64_t n) { return n >= 1024 && 235ull * sum_c_cm1 <= n * (n - 1); }

constexpr int kMinHufPlane = 2048;  // shorter planes (the tail block of a chunk) are written Raw / RLE: every lane of
                                    // the encode kernel then owns a run of at least 8 symbols of its stream

}  // namespace zs
}  // namespace lsr
