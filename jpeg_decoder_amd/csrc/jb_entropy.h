// jb_entropy.h -- internal: the entropy-decoding primitives shared by the fast baseline front end
// (jb_frontend.cpp) and the general multi-scan / progressive one (jb_frontend_ext.cpp): zig-zag
// order, canonical Huffman tables with lookahead, the MSB-first bit reader, one sequential block.
#ifndef JB_ENTROPY_H
#define JB_ENTROPY_H

#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/jpegblk.h"

#ifndef JB_WIDE_BITS
#define JB_WIDE_BITS 11
#endif

namespace jbe {

// zig-zag position -> natural index (ITU-T T.81 Figure A.6; reference types.hpp:23-31)
static const uint8_t kZigZag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
                             12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                             35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
                             58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct HuffTable {
  bool set = false;
  uint8_t counts[17] = {0};
  uint8_t symbols[256] = {0};
  // canonical decode (reference huffman.hpp:17-29 generates the same codes)
  int32_t maxcode[18];
  int32_t valptr[17];
  int32_t mincode[17];
  // 9-bit lookahead: (length << 8) | symbol, 0 = longer than 9 bits
  uint16_t fast[512];
  // the same over an 11-bit window: what the device-side decoder (jb_huff.hip) resolves codes with
  uint16_t code11[2048];
  // AC tables: kWideBits-bit lookahead that resolves up to TWO run/size symbols together with their
  // magnitude bits in one step (T.81 F.2.2.1 EXTEND folded in).  The entropy stage is one serial
  // dependency chain per scan -- bit buffer -> index -> table load -> shift -> bit buffer -- so what
  // shortens it is fewer lookups per coefficient: at the usual qualities two short symbols fit the
  // window often.  A symbol is (run, value, inc): k += run; out[zigzag[k]] = value; k += inc -- a
  // ZRL is (16, 0, 0) and an absent second symbol (0, 0, 0), both of which store a zero at a
  // position that is still zero, so the second symbol needs no branch of its own.
  //   bits  0..3  bits consumed by both symbols (0 = take the general path)
  //   bits  4..7  bits consumed by the first symbol alone (used when it completes the block)
  //   bits  8..12 run1      13..17 run2      18 inc1   19 inc2   20 first is EOB   21 second is EOB
  //   bits 32..47 value1    48..63 value2
  static constexpr int kWideBits = JB_WIDE_BITS;
  static_assert(kWideBits <= 15, "bit counts are stored in 4 bits");
  uint64_t pair[1 << kWideBits];
  // DC tables: code + difference bits in one lookup: (difference << 8) | bits consumed, 0 = general path
  static constexpr int kDcBits = 10;
  int32_t dcw[1 << kDcBits];

  static constexpr uint64_t kInc1 = 1ull << 18, kInc2 = 1ull << 19, kEob1 = 1ull << 20, kEob2 = 1ull << 21;

  // code length and symbol of the code that starts a `width`-bit window `w`; 0 = longer than the window
  int window_symbol(uint32_t w, int width, int *sym) const {
    int32_t code = 0;
    for (int len = 1; len <= 16 && len <= width; len++) {
      code = (int32_t)(w >> (width - len));
      if (counts[len] && code <= maxcode[len] && code >= mincode[len]) {
        *sym = symbols[valptr[len] + code - mincode[len]];
        return len;
      }
    }
    return 0;
  }
  // one AC symbol with its magnitude bits out of a `width`-bit window; returns bits used (0 = does
  // not fit / not resolvable here) and the (run, value, inc, eob) it stands for
  int window_ac(uint32_t w, int width, int *run, int *value, int *inc, bool *eob) const {
    int rs = 0;
    const int len = window_symbol(w, width, &rs);
    if (!len) return 0;
    const int r = rs >> 4, mag = rs & 15;
    if (len + mag > width || mag > 10) return 0;
    *eob = false;
    if (rs == 0) {
      *eob = true, *run = 0, *value = 0, *inc = 0;
      return len;
    }
    if (rs == 0xf0) {
      *run = 16, *value = 0, *inc = 0;
      return len;
    }
    // run-only symbols 0x10..0xE0: left to the general path, which (like reference jpeg.cpp:377-385) skips the run and stores nothing
    if (mag == 0) return 0;
    const int m = (int)((w >> (width - len - mag)) & ((1u << mag) - 1));
    *value = m < (1 << (mag - 1)) ? m - (1 << mag) + 1 : m;
    *run = r, *inc = 1;
    return len + mag;
  }

  // build() through a small per-thread cache keyed by the DHT content: the files of a batch mostly
  // carry the same tables (Annex K, or one encoder's), and filling the 2048-entry pair table costs
  // as much as decoding a few hundred blocks
  bool build_cached(bool is_ac) {
    struct Slot {
      bool used = false, is_ac = false;
      uint8_t counts[17], symbols[256];
      std::unique_ptr<HuffTable> built;  // freed when the thread ends
    };
    static thread_local Slot cache[8];
    static thread_local int next = 0;
    for (Slot &c : cache)
      if (c.used && c.is_ac == is_ac && memcmp(c.counts, counts, 17) == 0 && memcmp(c.symbols, symbols, 256) == 0) {
        *this = *c.built;
        return true;
      }
    if (!build(is_ac)) return false;
    Slot &c = cache[next];
    next = (next + 1) % 8;
    if (!c.built) c.built.reset(new HuffTable());
    *c.built = *this;
    c.used = true;
    c.is_ac = is_ac;
    memcpy(c.counts, counts, 17);
    memcpy(c.symbols, symbols, 256);
    return true;
  }

  bool build(bool is_ac) {
    int code = 0, k = 0;
    for (int len = 1; len <= 16; len++) {
      valptr[len] = k;
      mincode[len] = code;
      k += counts[len];
      code += counts[len];
      maxcode[len] = counts[len] ? code - 1 : -1;
      if (code > (1 << len)) return false;  // over-subscribed
      code <<= 1;
    }
    maxcode[17] = 0x7fffffff;
    memset(fast, 0, sizeof fast);
    code = 0;
    k = 0;
    for (int len = 1; len <= 9; len++) {
      for (int i = 0; i < counts[len]; i++, k++, code++) {
        const int first = code << (9 - len);
        for (int j = 0; j < (1 << (9 - len)); j++) fast[first + j] = (uint16_t)((len << 8) | symbols[k]);
      }
      code <<= 1;
    }
    memset(code11, 0, sizeof code11);
    code = 0;
    k = 0;
    for (int len = 1; len <= 11; len++) {
      for (int i = 0; i < counts[len]; i++, k++, code++) {
        const int first = code << (11 - len);
        for (int j = 0; j < (1 << (11 - len)); j++) code11[first + j] = (uint16_t)((len << 8) | symbols[k]);
      }
      code <<= 1;
    }
    memset(pair, 0, sizeof pair);
    memset(dcw, 0, sizeof dcw);
    if (is_ac) {
      for (uint32_t w = 0; w < (1u << kWideBits); w++) {
        int run1, v1, inc1, run2 = 0, v2 = 0, inc2 = 0;
        bool eob1, eob2 = false;
        const int n1 = window_ac(w, kWideBits, &run1, &v1, &inc1, &eob1);
        if (!n1) continue;
        int n = n1;
        if (!eob1 && n1 < kWideBits) {
          const int rest = kWideBits - n1;
          const int n2 = window_ac(w & ((1u << rest) - 1), rest, &run2, &v2, &inc2, &eob2);
          if (n2) n += n2;
          else run2 = v2 = inc2 = 0, eob2 = false;
        }
        pair[w] = (uint64_t)n | ((uint64_t)n1 << 4) | ((uint64_t)run1 << 8) | ((uint64_t)run2 << 13) |
                  (inc1 ? kInc1 : 0) | (inc2 ? kInc2 : 0) | (eob1 ? kEob1 : 0) | (eob2 ? kEob2 : 0) |
                  ((uint64_t)(uint16_t)(int16_t)v1 << 32) | ((uint64_t)(uint16_t)(int16_t)v2 << 48);
      }
    } else {
      for (uint32_t w = 0; w < (1u << kDcBits); w++) {
        int s = 0;
        const int len = window_symbol(w, kDcBits, &s);
        if (!len || s > 11 || len + s > kDcBits) continue;
        int diff = 0;
        if (s) {
          const int m = (int)((w >> (kDcBits - len - s)) & ((1u << s) - 1));
          diff = m < (1 << (s - 1)) ? m - (1 << s) + 1 : m;
        }
        dcw[w] = (int32_t)((uint32_t)diff << 8) | (len + s);
      }
    }
    set = true;
    return true;
  }
};

struct Err {
  int code = JB_OK;
  std::string msg;
};

inline int set_err(Err &e, int code, const char *msg) {
  e.code = code;
  e.msg = msg;
  return code;
}

// MSB-first bit reader over the stuffed scan bytes: FF00 -> FF, a marker stops the stream
// (zero bits are supplied past it).  Reference equivalent: readImageData + BitStream
// (file.hpp:59-104, 130-164).
struct BitReader {
  const uint8_t *p, *end;
  uint64_t acc = 0;
  int nbits = 0;
  int marker = 0;  // pending marker byte (0 = none)
  int pad = 0;     // zero bytes supplied past a marker / the end of the data

  BitReader(const uint8_t *b, const uint8_t *e) : p(b), end(e) {}

  inline void refill() {
    // fast path: 8 bytes at once when none of them is 0xFF (no stuffing, no marker)
    if (!marker && p + 8 <= end) {
      uint64_t w;
      memcpy(&w, p, 8);
      const uint64_t nw = ~w;  // a 0xFF byte in w is a zero byte in ~w
      if ((((nw - 0x0101010101010101ull) & ~nw) & 0x8080808080808080ull) == 0) {
        w = __builtin_bswap64(w);
        const int take = (64 - nbits) >> 3;  // whole bytes that fit
        if (take > 0) {
          const uint64_t m = take == 8 ? ~0ull : ~(~0ull >> (8 * take));
          acc |= (w & m) >> nbits;
          p += take;
          nbits += 8 * take;
        }
        return;
      }
    }
    while (nbits <= 56) {
      uint32_t byte = 0;
      if (!marker && p < end) {
        byte = *p++;
        if (byte == 0xff) {
          while (p < end && *p == 0xff) p++;  // fill bytes (reference file.hpp:88-91)
          const uint8_t m = p < end ? *p++ : 0xd9;
          if (m == 0) byte = 0xff;
          else {
            marker = m;
            byte = 0;
            pad++;
          }
        }
      } else {
        pad++;
      }
      acc |= (uint64_t)byte << (56 - nbits);
      nbits += 8;
    }
  }
  inline uint32_t peek(int n) { return (uint32_t)(acc >> (64 - n)); }
  inline void drop(int n) {
    acc <<= n;
    nbits -= n;
  }
  inline uint32_t get(int n) {
    if (n == 0) return 0;
    const uint32_t v = peek(n);
    drop(n);
    return v;
  }
  // restart boundary: discard the partial byte, step over the RSTn marker
  // (reference: BitStream::align, file.hpp:161-164, after readImageData dropped the marker)
  // true when bits past the real data (padding zeros) have been consumed: truncated/corrupt
  bool overran() const { return pad * 8 > nbits; }
  bool restart() {
    if (overran()) return false;
    pad = 0;
    if (!marker) {
      // marker not reached by the lookahead yet: the remaining buffered bits are padding;
      // scan forward to it
      acc = 0;
      nbits = 0;
      while (p + 1 < end && !(p[0] == 0xff && p[1] >= 0xd0 && p[1] <= 0xd7)) p++;
      if (p + 1 >= end) return false;
      p += 2;
      return true;
    }
    if (marker < 0xd0 || marker > 0xd7) return false;
    marker = 0;
    acc = 0;
    nbits = 0;
    return true;
  }
};

inline int decode_symbol(BitReader &br, const HuffTable &t) {
  if (br.nbits < 32) br.refill();
  const uint32_t look = br.peek(9);
  const uint16_t f = t.fast[look];
  if (f) {
    br.drop(f >> 8);
    return f & 0xff;
  }
  int32_t code = (int32_t)br.peek(10);
  int len = 10;
  while (len <= 16 && code > t.maxcode[len]) {
    len++;
    code = (int32_t)br.peek(len);
  }
  if (len > 16) return -1;
  br.drop(len);
  return t.symbols[t.valptr[len] + code - t.mincode[len]];
}

inline int extend(uint32_t v, int n) {
  // T.81 F.2.2.1 EXTEND; reference jpeg.cpp:340-343, 394-397
  return (int)v < (1 << (n - 1)) ? (int)v - (1 << n) + 1 : (int)v;
}

// one block, reference decodeMCUComponent (jpeg.cpp:322-403)
inline bool decode_block(BitReader &br, const HuffTable &dc, const HuffTable &ac, int &pred, int16_t *out) {
  memset(out, 0, 128);
  if (br.nbits < 32) br.refill();
  int diff;
  if (const int32_t fd = dc.dcw[br.peek(HuffTable::kDcBits)]) {  // code and difference bits in one lookup
    br.drop(fd & 0xff);
    diff = fd >> 8;
  } else {
    const int s = decode_symbol(br, dc);
    if (s < 0 || s > 11) return false;
    diff = s ? extend(br.get(s), s) : 0;  // decode_symbol left >= 16 bits buffered
  }
  pred += diff;
  if (pred < -32768 || pred > 32767) return false;
  out[0] = (int16_t)pred;
  int k = 1;
  while (k < 64) {
    if (br.nbits < 32) br.refill();
    const uint64_t e = ac.pair[br.peek(HuffTable::kWideBits)];
    const int n = (int)(e & 15);
    if (n) {  // up to two symbols, magnitude bits included, resolved by one lookup
      if (e & HuffTable::kEob1) {
        br.drop(n);
        break;
      }
      k += (int)(e >> 8) & 31;
      if (k > 63) return false;  // reference jpeg.cpp:372-376
      out[kZigZag[k]] = (int16_t)(e >> 32);
      k += (int)(e >> 18) & 1;
      if (k > 63) {  // the block is complete: what follows belongs to the next block
        br.drop((int)(e >> 4) & 15);
        break;
      }
      br.drop(n);
      if (e & HuffTable::kEob2) break;
      k += (int)(e >> 13) & 31;
      if (k > 63) return false;
      out[kZigZag[k]] = (int16_t)(e >> 48);  // (a zero onto a zero when there is no second symbol)
      k += (int)(e >> 19) & 1;
      continue;
    }
    const int rs = decode_symbol(br, ac);
    if (rs < 0) return false;
    if (rs == 0) break;  // EOB
    int r = rs >> 4;
    const int nb = rs & 15;
    if (rs == 0xf0) r = 16;
    if (k + r >= 64 || nb > 10) return false;  // reference jpeg.cpp:372-385
    k += r;
    if (nb) {
      out[kZigZag[k]] = (int16_t)extend(br.get(nb), nb);  // <= 16 + 10 bits since the last refill
      k++;
    }
  }
  return true;
}

// ---- the fast path of the baseline front end: de-stuffed scan + branch-free refill ----------
//
// The entropy-coded segment is first copied into a clean buffer (FF00 -> FF, fill bytes dropped,
// RSTn markers removed and their positions kept: memchr speed, ~1 % of the decode), so that the
// bit reader can refill with one unaligned 8-byte load and no branch -- and restart intervals are
// plain byte ranges, which is what lets several of them be decoded side by side.

struct CleanScan {
  // Worst case one block reads: 27 bits of DC + 63 x 26 bits of AC = 209 bytes; the readers check
  // their position once per block, so this much zero padding behind the data keeps every load in bounds.
  static constexpr size_t kPad = 512;
  std::vector<uint8_t> bytes;  // clean data, then kPad zero bytes
  std::vector<size_t> start;   // start[i] = first byte of restart interval i; the last entry = end of the data
  int n_intervals() const { return (int)start.size() - 1; }
};

// A stretch of a scan without a branch per 0xFF: 32 bytes per step, the stuffed zero bytes squeezed out of each
// 8-byte word with PEXT.  A synthetic stream has one FF00 per ~30 bytes (a photograph one per ~250), and a loop
// that branches on every one of them pays a misprediction each time: 23 ns per pair, 0.45 ms for a 1080p file of
// tools/jpegwriter, which was most of what jb_huff_prepare_ cost.  Works through at most `limit` bytes; stops in
// front of the 32 bytes that hold a 0xFF followed by anything other than 0x00 (RSTn, EOI, fill bytes) and in front
// of the last 32 bytes of the buffer; p / o are left where the byte-exact loop of unstuff() can take over.
// Returns the number of stuffed bytes it removed (what unstuff() chooses its loop by).
#if defined(__x86_64__)
__attribute__((target("avx2,bmi2,popcnt"))) inline size_t unstuff_words_x86(const uint8_t *&p_io, const uint8_t *end,
                                                                             uint8_t *&o_io, size_t limit) {
  typedef long long v4di __attribute__((vector_size(32)));
  typedef char v32qi __attribute__((vector_size(32)));
  const uint8_t *p = p_io;
  uint8_t *o = o_io;
  const uint8_t *const stop = (size_t)(end - p) > limit ? p + limit : end;
  size_t removed = 0;
  uint64_t carry = 0;  // 1 when the byte in front of these 32 was a 0xFF (already written out)
  const v32qi all_ff = (v32qi)(v4di){-1, -1, -1, -1}, all_00 = (v32qi)(v4di){0, 0, 0, 0};
  while (p + 32 <= stop) {
    v32qi v;
    memcpy(&v, p, 32);
    const uint32_t ff = (uint32_t)__builtin_ia32_pmovmskb256((v32qi)(v == all_ff));
    const uint32_t zz = (uint32_t)__builtin_ia32_pmovmskb256((v32qi)(v == all_00));
    const uint64_t second64 = ((uint64_t)ff << 1) | carry;  // bytes that follow a 0xFF
    const uint32_t second = (uint32_t)second64;
    if (second & ~zz) break;                                // ... and are not the stuffed zero: rare, the caller's
    const uint32_t keep = ~second;
    for (int j = 0; j < 4; j++) {
      const uint64_t kb = (keep >> (8 * j)) & 0xff;
      const uint64_t m = __builtin_ia32_pdep_di(kb, 0x0101010101010101ull) * 0xffull;  // bits -> whole bytes
      uint64_t w;
      memcpy(&w, p + 8 * j, 8);
      const uint64_t packed = __builtin_ia32_pext_di(w, m);
      memcpy(o, &packed, 8);  // up to 8 bytes ahead of the data kept: CleanScan::kPad covers it
      o += __builtin_popcountll(kb);
    }
    removed += (size_t)__builtin_popcount(second);
    p += 32;
    carry = second64 >> 32;
  }
  if (carry) p -= 1, o -= 1;  // the 0xFF in front belongs to what follows: hand it back as well
  p_io = p, o_io = o;
  return removed;
}
#endif

// [p, end): the entropy-coded bytes from the first byte after SOS to the end of the file buffer;
// stops at EOI / any marker that is not RSTn (reference equivalent: readImageData, file.hpp:59-104)
inline void unstuff(const uint8_t *p, const uint8_t *end, CleanScan &cs) {
  cs.bytes.resize((size_t)(end - p) + CleanScan::kPad);
  uint8_t *const base = cs.bytes.data();
  uint8_t *o = base;
  cs.start.clear();
  cs.start.push_back(0);
  // Two loops, chosen per 4 KiB by how many 0xFF the last 4 KiB held: memchr + memcpy per run of plain bytes
  // (7 GB/s on a photograph), or the word loop above (3x faster than that on a dense synthetic stream, slower on
  // a sparse one).
  constexpr size_t kWindow = 4096, kDenseOneIn = 96;
  bool dense = false;
#if defined(__x86_64__)
  static const bool have_words =
      __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("popcnt");
  dense = have_words;
#else
  constexpr bool have_words = false;
#endif
  const uint8_t *win = p;
  size_t events = 0;
  while (p < end) {
#if defined(__x86_64__)
    if (dense) {
      const uint8_t *const p0 = p;
      const size_t removed = unstuff_words_x86(p, end, o, kWindow);
      if ((size_t)(p - p0) >= kWindow - 64) dense = removed * kDenseOneIn >= (size_t)(p - p0);
      win = p, events = 0;
      if (p >= end) break;
    }
#endif
    // byte-exact from here to the next 0xFF, and what that 0xFF means
    const uint8_t *q = (const uint8_t *)memchr(p, 0xff, (size_t)(end - p));
    if (!q) q = end;
    memcpy(o, p, (size_t)(q - p));
    o += q - p;
    p = q;
    if (p + 1 >= end) break;  // the end of the buffer, or a lone FF as its last byte
    const uint8_t m = p[1];
    if (m == 0x00) {
      *o++ = 0xff;
      p += 2;
    } else if (m == 0xff) {
      p += 1;  // fill byte (reference file.hpp:88-91)
    } else if (m >= 0xd0 && m <= 0xd7) {
      cs.start.push_back((size_t)(o - base));
      p += 2;
    } else {
      break;  // EOI or any other marker ends the scan
    }
    events++;
    if (have_words && (size_t)(p - win) >= kWindow) {
      dense = events * kDenseOneIn >= (size_t)(p - win);
      win = p, events = 0;
    }
  }
  cs.start.push_back((size_t)(o - base));
  memset(o, 0, CleanScan::kPad);
}

// MSB-first reader over clean bytes.  refill() needs 8 readable bytes at p and leaves >= 56 valid
// bits; it is unconditional and branch-free, so the caller places it by a static bit budget
// instead of testing the fill level per symbol.
struct CleanReader {
  const uint8_t *p;
  uint64_t acc = 0;
  int nbits = 0;
  explicit CleanReader(const uint8_t *b) : p(b) {}
  inline void refill() {
    uint64_t w;
    memcpy(&w, p, 8);
    acc |= __builtin_bswap64(w) >> nbits;  // nbits <= 63 always
    p += (63 - nbits) >> 3;
    nbits |= 56;
  }
  inline uint32_t peek(int n) const { return (uint32_t)(acc >> (64 - n)); }
  inline void drop(int n) {
    acc <<= n;
    nbits -= n;
  }
  inline uint32_t get(int n) {
    if (n == 0) return 0;
    const uint32_t v = peek(n);
    drop(n);
    return v;
  }
  // bits consumed since the reader was placed at `from`
  inline int64_t consumed_bits(const uint8_t *from) const { return (int64_t)(p - from) * 8 - nbits; }
};

// canonical decode without a refill of its own: the caller guarantees >= 16 valid bits
inline int decode_symbol_clean(CleanReader &br, const HuffTable &t) {
  const uint16_t f = t.fast[br.peek(9)];
  if (f) {
    br.drop(f >> 8);
    return f & 0xff;
  }
  int32_t code = (int32_t)br.peek(10);
  int len = 10;
  while (len <= 16 && code > t.maxcode[len]) {
    len++;
    code = (int32_t)br.peek(len);
  }
  if (len > 16) return -1;
  br.drop(len);
  return t.symbols[t.valptr[len] + code - t.mincode[len]];
}

// One block, reference decodeMCUComponent (jpeg.cpp:322-403); same results as decode_block above.
// Bit budget: refill() leaves >= 56 bits; the DC lookup takes <= 10, each AC lookup <= kWideBits,
// so four AC lookups follow one refill (the general path refills for itself).
inline bool decode_block_clean(CleanReader &br, const HuffTable &dc, const HuffTable &ac, int &pred, int16_t *out) {
  static_assert(HuffTable::kDcBits + 4 * HuffTable::kWideBits <= 56, "bit budget of the unrolled loop");
  memset(out, 0, 128);
  br.refill();
  int diff;
  if (const int32_t fd = dc.dcw[br.peek(HuffTable::kDcBits)]) {
    br.drop(fd & 0xff);
    diff = fd >> 8;
  } else {
    const int s = decode_symbol_clean(br, dc);
    if (s < 0 || s > 11) return false;
    diff = s ? extend(br.get(s), s) : 0;  // 16 + 11 bits of the 56
    br.refill();
  }
  pred += diff;
  if (pred < -32768 || pred > 32767) return false;
  out[0] = (int16_t)pred;
  int k = 1;
  for (;;) {
#pragma GCC unroll 4
    for (int u = 0; u < 4; u++) {
      const uint64_t e = ac.pair[br.peek(HuffTable::kWideBits)];
      const int n = (int)(e & 15);
      if (__builtin_expect(n != 0, 1)) {  // up to two symbols, magnitude bits included, in one lookup
        if (e & HuffTable::kEob1) {
          br.drop(n);
          return true;
        }
        k += (int)(e >> 8) & 31;
        if (k > 63) return false;  // reference jpeg.cpp:372-376
        out[kZigZag[k]] = (int16_t)(e >> 32);
        k += (int)(e >> 18) & 1;
        if (k > 63) {  // the block is complete: what follows belongs to the next block
          br.drop((int)(e >> 4) & 15);
          return true;
        }
        br.drop(n);
        if (e & HuffTable::kEob2) return true;
        k += (int)(e >> 13) & 31;
        if (k > 63) return false;
        out[kZigZag[k]] = (int16_t)(e >> 48);  // (a zero onto a zero when there is no second symbol)
        k += (int)(e >> 19) & 1;
        if (k > 63) return true;
      } else {
        br.refill();
        const int rs = decode_symbol_clean(br, ac);
        if (rs < 0) return false;
        if (rs == 0) return true;  // EOB
        int r = rs >> 4;
        const int nb = rs & 15;
        if (rs == 0xf0) r = 16;
        if (k + r >= 64 || nb > 10) return false;  // reference jpeg.cpp:372-385
        k += r;
        if (nb) {
          out[kZigZag[k]] = (int16_t)extend(br.get(nb), nb);
          k++;
          if (k > 63) return true;
        }
        br.refill();
      }
    }
    br.refill();
  }
}

}  // namespace jbe

#endif  // JB_ENTROPY_H
