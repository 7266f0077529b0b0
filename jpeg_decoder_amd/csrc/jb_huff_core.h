// jb_huff_core.h -- the device entropy decoder's per-symbol step and its coordinates, in one place:
// jb_huff.hip compiles it for gfx950; tools/huff_emu compiles the same text for the host, where the
// pass structure of the kernels is replayed lane by lane (test infrastructure: a step-exact model
// to study synchronisation on, never linked into the product).
//
// What a step does is what reference decodeMCUComponent does per symbol (jpeg.cpp:322-403): resolve
// one Huffman code, take its magnitude bits, EXTEND (jpeg.cpp:340-343, 394-397), advance the
// position in the block -- as a per-lane state machine (k == 0: the DC symbol is next).
//
// Local coordinates of a lane.  Its chunk starts at byte `start` of the image's clean scan; the lane
// keeps the big-endian dwords from byte A = (start & ~3) - 4 on, TRANSPOSED in LDS: dword j of lane l
// at stream[j * 256 + l] -- the bank of an access is the lane's, whatever j is, so the lanes of a
// wave never conflict however far apart they are in their chunks.  A position is u = (bits from
// byte A) - 1, so that the 32 bits at u are alignbit(dword[u >> 5], dword[(u >> 5) + 1], ~u) with no
// special case for dword-aligned positions (the shift count is taken mod 32).  u >= 31 always.
#pragma once
#include <stdint.h>

#include "jb_huff.h"

#if defined(__HIPCC__)
#define JBH_FN __host__ __device__ inline __attribute__((always_inline))
#else
#define JBH_FN inline
#endif

JBH_FN uint32_t jbh_alignbit(uint32_t hi, uint32_t lo, uint32_t sh) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_alignbit(hi, lo, sh);
#else
  return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (sh & 31u));
#endif
}
JBH_FN uint32_t jbh_ubfe(uint32_t v, uint32_t off, uint32_t width) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_ubfe(v, off, width);
#else
  off &= 31u, width &= 31u;
  return width ? (v >> off) & ((1u << width) - 1u) : 0u;
#endif
}
JBH_FN int32_t jbh_sbfe(uint32_t v, uint32_t off, uint32_t width) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_sbfe((int32_t)v, off, width);
#else
  off &= 31u, width &= 31u;
  if (!width) return 0;
  const uint32_t f = (v >> off) & ((1u << width) - 1u);
  return (int32_t)(f << (32u - width)) >> (32u - width);
#endif
}

JBH_FN uint32_t jbh_mul24(uint32_t a, uint32_t b) { return (a & 0xffffffu) * (b & 0xffffffu); }  // (v_mul_u32_u24: a, b < 2^24)

// (tools/huff_emu counts the steps, per lane and per pass)
#ifndef JBH_TRACE_STEP
#define JBH_TRACE_STEP() ((void)0)
#define JBH_TRACE_PASS_END(pass) ((void)0)
#endif

constexpr uint32_t kJbhLanes = (uint32_t)kJbHuffLanes;
// rows of the transposed stream a lane may touch: the dword before the chunk's first aligned one,
// the chunk (misaligned by up to 3 bytes: one more), and the dword a last symbol's window reaches into
constexpr uint32_t jbh_rows(uint32_t chunk_bytes) { return chunk_bytes / 4u + 3u; }

// local coordinates of a chunk that starts at byte `start` of the image's clean scan
JBH_FN uint32_t jbh_base_byte(uint32_t start) { return (start & ~3u) - 4u; }                   // A (may wrap below 0 for start < 4: scan_off >= 16)
JBH_FN uint32_t jbh_u_of_bit(uint32_t start, uint32_t bit) { return bit - 8u * jbh_base_byte(start) - 1u; }
JBH_FN uint32_t jbh_bit_of_u(uint32_t start, uint32_t u) { return u + 8u * jbh_base_byte(start) + 1u; }

struct JbhCtx {            // what a lane's steps read besides its own state
  const uint32_t *scol;    // this lane's column of the transposed stream: dword j at scol[j * kJbhLanes]
  const uint16_t *tab;     // first-level tables (n_tabs x kJbT1Entries), then the second-level ones
  const uint8_t *zz2;      // 2 * natural index of zig-zag position i (writing pass)
  uint32_t lut_ac, lut_dc, lut_comp;  // JbHuffImage
  uint32_t nb4;            // 4 * blocks per MCU
  uint32_t blk_bytes;      // JbHuffImage (writing pass)
  uint32_t t2_first;       // index of the first second-level entry in tab: n_tabs * kJbT1Entries
};
struct JbhLane {
  uint32_t u;              // position of the next symbol
  uint32_t k;              // 0: a DC symbol is next; 1..63: an AC symbol for zig-zag position k
  uint32_t blk4;           // 4 * block-in-MCU
  uint32_t nblk;           // blocks completed
};
// (the three DC values of a lane -- synchronisation: sums of the DC differences so far; writing pass: the DC
// predictors -- are separate variables on purpose: as neighbouring fields of a struct the compiler reads them as
// an array indexed by the component and puts the struct into scratch memory)

// One symbol, without a branch (but the rare second-level lookup): with 64 lanes at 64 different places of their
// blocks every path is taken by some lane in every step, so a path not taken saves nothing.
// kStore == false (synchronisation): only the state moves -- no magnitudes, no DC values; bits that are no
// possible continuation of the state (the lane is out of step, or the data is corrupt -- which the writing pass will
// report) move the lane on by one bit, to expect a block's start, at the NEXT place in the MCU: a wrong guess of
// the block's place (luma tables on a chroma block) is what produces most impossible symbols and does not correct
// itself.  kStore == true (writing pass): `coef` = the image's coefficient blocks, `block0` = the block the lane's
// chunk starts in (an image the device decoder takes has fewer than 2^24 blocks, 2^32 bytes of them); a DC symbol
// stores its DIFFERENCE (jb_huff_dc_kernel turns the differences into predictors) and adds it to the chunk's sum
// of its component; after impossible bits the state is not to be used.
// Returns true for impossible bits.
template <bool kStore>
JBH_FN bool jbh_step(const JbhCtx &cx, JbhLane &st, uint32_t &dc0, uint32_t &dc1, uint32_t &dc2, uint8_t *coef, uint32_t block0) {
  JBH_TRACE_STEP();
  const uint32_t u = st.u;
  const uint32_t j = jbh_ubfe(u, 5, 27);
  const uint32_t w = jbh_alignbit(cx.scol[j * kJbhLanes], cx.scol[j * kJbhLanes + kJbhLanes], ~u);
  const bool isdc = st.k == 0;
  const uint32_t tix = jbh_ubfe(isdc ? cx.lut_dc : cx.lut_ac, st.blk4, 4);
  uint32_t e = cx.tab[(tix << kJbT1Bits) + (w >> (32 - kJbT1Bits))];
  if (e - 1u < 31u)  // a code longer than the first level resolves: the entry names its second-level table
    e = cx.tab[cx.t2_first + ((e - 1u) << kJbT2Bits) + ((w >> 16) & (kJbT2Entries - 1u))];
  const uint32_t total = e & 31u, size = (e >> 5) & 15u, adv = e >> 9;
  const uint32_t k1 = st.k + adv;
  // no code here, or a run that leaves the block (reference jpeg.cpp:372 "Invalid AC length": k + run >= 64)
  const bool bad = e == 0 || (k1 - 65u) < 17u;
  if (kStore) {
    // EXTEND without a branch: x = the magnitude bits read as a signed field; a field that starts with 1 stands for
    // itself (x + 2^size), one that starts with 0 for x - (2^size - 1)
    const int32_t x = jbh_sbfe(w, 32u - total, size);
    const int32_t val = x - (int32_t)(((1u << size) - 1u) ^ (uint32_t)(x >> 31));
    const uint32_t c = jbh_ubfe(cx.lut_comp, st.blk4, 4);
    const uint32_t dcv = (isdc && !bad) ? (uint32_t)val : 0u;
    dc0 += c == 0 ? dcv : 0u;
    dc1 += c == 1 ? dcv : 0u;
    dc2 += c == 2 ? dcv : 0u;
    if (!bad && (isdc || size != 0))  // (k1 - 1 = k + run, 0 for DC)
      *(int16_t *)(coef + (jbh_mul24(block0 + st.nblk, cx.blk_bytes) + cx.zz2[k1 - 1u])) = (int16_t)val;
  }
  const uint32_t k2 = k1 - ((size == 0 && !isdc) ? 1u : 0u);  // a run without a coefficient (ZRL, 0x10..0xE0): its adv counted one
  const bool done = k2 >= 64u;                                  // (an EOB lands far beyond)
  const bool next = done || bad;
  st.k = next ? 0u : k2;
  st.nblk += (done && !bad) ? 1u : 0u;
  const uint32_t b4 = st.blk4 + (next ? 4u : 0u);
  st.blk4 = b4 == cx.nb4 ? 0u : b4;
  st.u = u + (bad ? 1u : total);
  return bad;
}

JBH_FN uint32_t jbh_pack_state(const JbhLane &st) { return st.u | (st.k << 11) | (st.blk4 << 15) | (st.nblk << 20); }  // (blk4 <= 20: bits 15..19)

// ---- host: table construction ---------------------------------------------------------------------
#include <string.h>

// entry of a symbol of `len` code bits: see jb_huff.h
inline uint16_t jbh_entry_(bool is_ac, int len, int sym) {
  if (!is_ac) {
    if (sym > 11) return 0;  // reference jpeg.cpp:330 "Invalid DC length"
    return (uint16_t)((len + sym) | (sym << 5) | (1 << 9));
  }
  const int run = sym >> 4, size = sym & 15;
  if (size > 10) return 0;  // reference jpeg.cpp:381 "Invalid AC length > 10"
  if (sym == 0) return (uint16_t)(len | (kJbAdvEob << 9));
  if (sym == 0xf0) return (uint16_t)(len | (17 << 9));  // 16 zeros, no coefficient
  return (uint16_t)((len + size) | (size << 5) | ((run + 1) << 9));
}

inline bool jb_huff_fill_table_impl_(const uint8_t counts[17], const uint8_t *symbols, bool is_ac, JbHuffTables *set, uint32_t tix, uint32_t n_tabs,
                                     uint32_t *n_t2) {
  uint16_t *t1 = set->t1[tix];
  memset(t1, 0, sizeof set->t1[0]);
  // canonical codes, shortest first (reference huffman.hpp:17-29 generates the same codes)
  uint32_t code = 0;
  int k = 0;
  int32_t sub_of_prefix = -1;
  uint32_t sub_prefix = 0;
  for (int len = 1; len <= 16; len++) {
    for (int i = 0; i < counts[len]; i++, k++, code++) {
      if (code >= (1u << len)) return false;  // over-subscribed
      const uint16_t e = jbh_entry_(is_ac, len, symbols[k]);
      if (len <= kJbT1Bits) {
        const uint32_t first = code << (kJbT1Bits - len);
        for (uint32_t q = 0; q < (1u << (kJbT1Bits - len)); q++) t1[first + q] = e;
      } else {
        const uint32_t prefix = code >> (len - kJbT1Bits);
        if (sub_of_prefix < 0 || sub_prefix != prefix) {
          if (*n_t2 >= kJbT2Tables) return false;  // the pool is full: this file stays with the host decoder
          sub_of_prefix = (int32_t)(*n_t2)++;
          sub_prefix = prefix;
          memset(set->t2[sub_of_prefix], 0, sizeof set->t2[0]);
          t1[prefix] = (uint16_t)(sub_of_prefix + 1);  // 1..24: no symbol's entry is that small (adv >= 1: 512 and up)
        }
        const uint32_t rest = len - kJbT1Bits;  // bits of the code behind the prefix
        const uint32_t first = (code & ((1u << rest) - 1u)) << (kJbT2Bits - rest);
        for (uint32_t q = 0; q < (1u << (kJbT2Bits - rest)); q++) set->t2[sub_of_prefix][first + q] = e;
      }
    }
    code <<= 1;
  }
  return true;
}
