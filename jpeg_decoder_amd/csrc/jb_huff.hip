// jb_huff.hip -- Huffman decoding of baseline scans ON THE DEVICE (gfx950).  Beyond the reference
// (its decodeHuffman, jpeg.cpp:405-446, is serial host code and north_star keeps the entropy stage
// on the host): SURVEY.md section 8(f) rank 4.  Files with short restart intervals take the
// interval decoder below (one lane per interval), files without DRI and files with long intervals
// the self-synchronising decoder further down (one lane per 256-byte chunk); progressive /
// grayscale / multi-scan files stay on the host (jb_frontend_ext.cpp).
//
// Why it works: the DC predictors reset at every restart marker (T.81 F.2.1.3.1; reference
// jpeg.cpp:419-425), so the intervals of a scan are independent bit streams -- after the host has
// removed the byte stuffing and recorded where each interval starts (jbe::unstuff: memchr speed),
// interval i is "decode ri MCUs from byte start[i]".  A lane does what the host decoder does for
// one interval -- the same canonical codes (the host front end's 11-bit code table and canonical
// arrays), the same EXTEND, the same checks -- one symbol per step; tests pin the coefficients
// against the reference's coefficient dumps
// (images/img4.jpg has DRI = 100) and against the host decoder on writer- and libjpeg-made files.
//
// Work decomposition: a workgroup = 256 lanes = 256 consecutive intervals of one image.
//   LDS: the image's table set (18 KiB: an 11-bit code table per AC / DC slot + canonical arrays for
//   longer codes) + a 64-byte ring of upcoming stream bytes per lane (16 KiB): four workgroups per CU.
//   Every lane walks its own interval as a state machine, one symbol per step (DC and AC symbols
//   are the same step); lanes are not held together at block boundaries, so a wave takes as many
//   steps as its longest interval has symbols.  The coefficient area is zeroed first (one kernel per
//   submission, microseconds); a lane stores its non-zero coefficients, de-zigzagged, straight
//   into its block's 128-byte line -- the layout is the fused pixel kernel's input.
//   The bit stream is read through a 3-dword register window per lane (two dwords in use, one
//   ahead) fed from the lane's LDS ring, which is topped up from HBM every 4 steps (struct Stream);
//   a symbol consumes at most 27 bits, so the window advances by at most one dword per step.
// Safety: every stream read is clamped into the image's padded scan; k only grows inside a block
// (at most 63 iterations); output indices come from host-validated counts.  Corrupt data sets the
// image's status word and the host re-decodes that image with the serial reader for the precise
// answer.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>

#include "jb_huff.h"
#include "jb_kernels.h"

namespace {


constexpr uint32_t kRing = 64;   // bytes of the stream a lane keeps staged in LDS

// A lane's view of its interval's bit stream.  Bits are taken from a two-dword register window
// (d0:d1 at bit offset `off`, d2 one dword ahead); the dwords come from a 128-byte ring in LDS that
// is topped up from HBM in aligned 16-byte chunks every 4 steps -- a wave-uniform point, and the
// chunk asked for at one such point is written into the ring at the next, so its latency is hidden
// behind four steps of decoding.  (Fetching per dword from HBM instead put a memory round
// trip into every iteration of the symbol loop: with 64 lanes in flight some lane crosses a dword
// in every iteration, and the wave waits for it.)  A block that outruns the ring reads straight
// from HBM, which is only slower.
struct Stream {
  const uint8_t *base;  // the image's clean scan (16-byte aligned)
  uint32_t limit;       // highest byte offset a 16-byte read may start at (inside the zero padding)
  uint8_t *ring;        // this lane's ring in LDS
  uint32_t wr;          // the ring holds stream bytes [wr - kRing, wr); multiple of 16
  uint32_t pos;         // byte offset of the next dword to fetch into the window
  uint32_t d0, d1, d2;  // window: bits come from d0:d1 at bit offset `off`; d2 is the next dword
  uint32_t off;         // 0..31
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
  u32x4_t pend0, pend1; // chunks on their way from HBM (asked for at the previous block boundary)
  uint32_t npend;       // 0..2

  __device__ __forceinline__ u32x4_t chunk(uint32_t at) const { return *(const u32x4_t *)(base + (at < limit ? at : limit)); }
  // The window is fed from LDS ONLY.  (A fetch with an HBM path in it makes d2 "possibly the result
  // of a global load", and every use of it then waits for vmcnt(0) -- which on gfx950 also means
  // every coefficient store issued so far: a memory round trip per iteration of the symbol loop.
  // Measured: 8.5-16 us per block round with that path, against the ~3 us the LDS latencies allow.)
  __device__ __forceinline__ uint32_t fetch(uint32_t at) const {
    return __builtin_bswap32(*(const uint32_t *)(ring + (at & (kRing - 1))));
  }
  // A block that outruns the ring (more than ~36 bytes of entropy-coded data in one block) refills
  // it on the spot: rare, and the only place inside a block that waits for HBM.  Replacing the ring's
  // oldest chunk is safe here: the window (from pos - 12 on) is at the ring's newest bytes.
  __device__ __forceinline__ void refill_now() {
    if (npend >= 1) {
      *(u32x4_t *)(ring + (wr & (kRing - 1))) = pend0;
      wr += 16;
    }
    if (npend >= 2) {
      *(u32x4_t *)(ring + (wr & (kRing - 1))) = pend1;
      wr += 16;
    }
    npend = 0;
    while (pos + 4 > wr) {
      *(u32x4_t *)(ring + (wr & (kRing - 1))) = chunk(wr);
      wr += 16;
    }
  }
  __device__ __forceinline__ void open(uint32_t start) {
    wr = start & ~15u;
#pragma unroll
    for (int i = 0; i < 4; i++) *(u32x4_t *)(ring + ((wr + 16u * i) & (kRing - 1))) = chunk(wr + 16u * i);
    wr += 64;
    npend = 0;
    pos = start & ~3u;
    off = (start & 3u) * 8u;
    d0 = fetch(pos);
    d1 = fetch(pos + 4);
    d2 = fetch(pos + 8);
    pos += 12;
  }
  // block boundary: what was asked for last time goes into the ring, then ask for what fits now
  __device__ __forceinline__ void top_up() {
    if (npend >= 1) {
      *(u32x4_t *)(ring + (wr & (kRing - 1))) = pend0;
      wr += 16;
    }
    if (npend >= 2) {
      *(u32x4_t *)(ring + (wr & (kRing - 1))) = pend1;
      wr += 16;
    }
    // a chunk may replace ring bytes [wr - kRing, wr - kRing + 16) once the window has moved past them
    // (the window's first dword is at pos - 12)
    npend = 0;
    if (wr + 16 + 12 <= pos + kRing) {
      pend0 = chunk(wr);
      npend = 1;
      if (wr + 32 + 12 <= pos + kRing) {
        pend1 = chunk(wr + 16);
        npend = 2;
      }
    }
  }
  // the next 32 bits of the stream
  __device__ __forceinline__ uint32_t window() const { return (uint32_t)(((((uint64_t)d0) << 32) | d1) << off >> 32); }
  // any n <= 32: the window moves by one dword at most
  __device__ __forceinline__ void consume(uint32_t n) {
    off += n;
    if (off >= 32) {
      off -= 32;
      d0 = d1;
      d1 = d2;
      if (__builtin_expect(pos + 4 > wr, 0)) refill_now();
      d2 = fetch(pos);
      pos += 4;
    }
  }
  // bit position in the scan (from its first byte)
  __device__ __forceinline__ uint64_t bitpos() const { return (uint64_t)(pos - 12) * 8 + off; }
};

__device__ __forceinline__ int extend(uint32_t v, int n) {  // T.81 F.2.2.1 EXTEND; reference jpeg.cpp:340-343
  return (int)v < (1 << (n - 1)) ? (int)v - (1 << n) + 1 : (int)v;
}

struct LdsTables {
  JbHuffTables t;
};

// A code of 12..16 bits (the 11-bit window table said "longer"): all five candidate lengths compared
// at once -- the canonical rule "the first length whose code does not exceed that length's largest
// code" (reference huffman.hpp:17-29 builds the same codes) -- no loop, because with 64 lanes in
// flight some lane is here in many iterations and the whole wave walks the path.
// -> (length << 8) | symbol, or 0 when no code matches.
__device__ __forceinline__ uint32_t long_code(uint32_t bits, const JbHuffTables &t, int slot) {
  int len = 17;
#pragma unroll
  for (int l = 16; l >= 12; l--)
    if ((int32_t)(bits >> (32 - l)) <= t.maxcode[slot][l]) len = l;
  if (len > 16) return 0;
  const int32_t code = (int32_t)(bits >> (32 - len));
  return ((uint32_t)len << 8) | t.symbols[slot][(t.valptr[slot][len] + code - t.mincode[slot][len]) & 255];
}

// Timing experiments (tools/build_huff_variant.sh, never the product; results are wrong with them):
//   JBH_NO_STORE    no coefficient stores        JBH_NO_TOPUP  the ring is only refilled when it runs dry
#ifdef JBH_NO_STORE
#define JBH_STORE(lhs, v) ((void)(v))
#else
#define JBH_STORE(lhs, v) ((lhs) = (v))
#endif

__constant__ uint8_t kZigZagDev[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
                                       12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                                       35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
                                       58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

}  // namespace

__global__ __launch_bounds__(kJbHuffLanes) void jb_huff_kernel(const JbHuffLaunch p) {
  __shared__ __attribute__((aligned(16))) LdsTables lds;
  __shared__ __attribute__((aligned(16))) uint8_t rings[kJbHuffLanes * kRing];
  __shared__ uint8_t zz[64];
  const int tid = threadIdx.x;
  const JbHuffWg wg = p.wgs[blockIdx.x];
  const JbHuffImage img = p.images[wg.image];
  // the table slots of the three components, packed (a dynamically indexed copy of `img` would be
  // promoted to LDS by the compiler: 12 KiB per workgroup)
  const uint32_t slots = (uint32_t)img.dc_slot[0] | ((uint32_t)img.dc_slot[1] << 1) | ((uint32_t)img.dc_slot[2] << 2) |
                         ((uint32_t)img.ac_slot[0] << 4) | ((uint32_t)img.ac_slot[1] << 5) | ((uint32_t)img.ac_slot[2] << 6);

  // the image's table set -> LDS (16-byte copies, the struct is a multiple of 16 bytes)
  {
    const uint4 *src = (const uint4 *)(p.tables + img.table_set);
    uint4 *dst = (uint4 *)&lds.t;
    for (int i = tid; i < (int)(sizeof(JbHuffTables) / 16); i += kJbHuffLanes) dst[i] = src[i];
    if (tid < 64) zz[tid] = kZigZagDev[tid];
  }
  __syncthreads();

  const uint32_t iv = wg.first_interval + (uint32_t)tid;
  const bool active = iv < img.n_int;
  const uint32_t nb = img.ny + 2;
  uint32_t count = 0, start = 0, end = 0;
  if (active) {
    const uint32_t m0 = iv * img.ri;
    count = img.n_mcus - m0 < img.ri ? img.n_mcus - m0 : img.ri;
    start = p.starts[img.int_off + iv];
    end = p.starts[img.int_off + iv + 1];
  }
  Stream s;
  s.base = p.scan + img.scan_off;
  s.limit = (img.scan_len + 48u) & ~15u;  // a 16-byte read from here still lies inside the 64 zero bytes behind the data
  s.ring = rings + tid * kRing;
  s.open(start);

  // The coefficient area was zeroed by the host's memset before this launch: a lane only stores the
  // non-zero coefficients, de-zigzagged, straight into its block's 128-byte line; the L2 gathers a
  // line's stores before the line leaves for HBM.
  int16_t *out = (int16_t *)((uint8_t *)p.coef + img.coef_off) + (int64_t)iv * img.ri * nb * 64;
  int pred0 = 0, pred1 = 0, pred2 = 0;
  uint32_t err = 0;

  // ---- one symbol per step, every lane at its own place in its own interval -------------------
  // What reference decodeMCUComponent does per block (jpeg.cpp:322-403; same results as the host
  // decoder's decode_block_clean) as a per-lane state machine: k == 0 means "the DC symbol is next",
  // 1..63 "an AC symbol for zig-zag position k is next".  Lanes are NOT held together at block
  // boundaries: a wave then takes as many steps as its longest INTERVAL has symbols (sums of 720
  // blocks barely differ between lanes), not the sum over block rounds of the longest BLOCK of the
  // round (a block of 35 symbols among 64 lanes in most rounds, against 11 on average) -- a third
  // of the steps on a photographic file.  DC and AC symbols are the same step: code from the 11-bit
  // window table of the lane's current component and kind, magnitude bits out of the same 32-bit
  // window, one consume, one predicated store.  The ring is topped up every 4 steps, a wave-uniform
  // point: a lane consumes at most 14 bytes in 4 steps, a top-up asks for up to 32 as soon as fewer
  // than 37 lie ahead, and what it asks for arrives at the next one -- the ring never runs dry.
  bool live = active && count > 0;
  uint32_t blk = 0, mcus_left = count;
  int k = 0;
  // (four steps per top-up as an unrolled inner loop: a step counter tested inside the loop body made the
  // compiler copy the loop-carried registers back and forth in every iteration -- 16 % of a pass)
  for (;;) {
    if (__builtin_amdgcn_ballot_w64(live) == 0) break;  // wave-uniform: every lane of this wave is through
#ifndef JBH_NO_TOPUP
    if (live) s.top_up();
#endif
#pragma unroll
    for (int u = 0; u < 4; u++)
    if (live) {
      const int c = blk < img.ny ? 0 : (int)(blk - img.ny) + 1;
      const bool isdc = k == 0;
      const uint32_t slot = (slots >> (isdc ? c : 4 + c)) & 1u;
      const uint32_t bits = s.window();
      // acl[2][2048] and dcl[2][2048] lie back to back: one 16-bit read at a per-lane offset
      uint32_t t = ((const uint16_t *)lds.t.acl)[(isdc ? 4096u : 0u) + slot * 2048u + (bits >> 21)];
      if (t == 0) t = long_code(bits, lds.t, (int)((isdc ? 0u : 2u) + slot));
      const uint32_t len = t >> 8, rs = t & 0xffu;
      const uint32_t sz = isdc ? rs : (rs & 15u);
      const bool eob = !isdc && rs == 0;
      const int kk = k + (isdc ? 0 : rs == 0xf0u ? 16 : (int)(rs >> 4));  // (a ZRL, or the run of a run/size symbol)
      bool bad = t == 0 || (isdc ? sz > 11 : (!eob && (kk > 63 || sz > 10)));  // reference jpeg.cpp:372-385
      int val = 0;
      if (sz && !bad) val = extend((bits << len) >> (32 - sz), (int)sz);  // len + sz <= 27 bits of the 32
      if (isdc) {
        int pr = (c == 0 ? pred0 : c == 1 ? pred1 : pred2) + val;
        if (pr < -32768 || pr > 32767) bad = true, pr = 0;
        if (c == 0) pred0 = pr;
        else if (c == 1) pred1 = pr;
        else pred2 = pr;
        val = pr;
      }
      if (!bad && (isdc ? val != 0 : sz != 0)) JBH_STORE(out[zz[kk & 63]], (int16_t)val);
      if (!bad) s.consume(len + sz);
      k = isdc ? 1 : eob ? 64 : kk + (sz ? 1 : 0);
      if (bad) {
        err |= 1;
        live = false;  // what follows in this interval is garbage; the host re-decodes the image
      }
      if (k > 63) {  // the block is complete: on to the next one of this lane's interval
        k = 0;
        out += 64;
        if (++blk == nb) {
          blk = 0;
          if (--mcus_left == 0) live = false;
        }
      }
    }
  }
  if (active) {
    // more bits consumed than the interval holds: truncated or corrupt data
    if (s.bitpos() > (uint64_t)end * 8) err |= 2;
    if (err) atomicOr(p.status + wg.image, err);
  }
}

// ================================================================================================
// Scans WITHOUT restart intervals, and scans whose intervals are long: the self-synchronising decoder.
//
// A Huffman stream can only be decoded from its start -- but a decoder started at a wrong place,
// in a wrong state, falls into step with the true symbol sequence after a few symbols or blocks
// and stays in step from then on.  So every restart interval (a scan without DRI is one interval)
// is cut into chunks of kJbChunkBytes bytes from its own first byte, one lane per chunk -- a 1080p
// file with one restart interval per MCU row is 3,000 lanes this way instead of the 135 of the
// interval decoder above:
//   sync pass 0      every lane decodes its chunk from the chunk's first bit, assuming "a DC symbol
//                    of the MCU's first block is next" (true for the first chunk of an interval), and
//                    records its EXIT state: the bit at which the first symbol of the next chunk
//                    starts, the position k inside the block, the block's place in the MCU, how many
//                    blocks it completed -- and the sum of the DC differences it decoded, per component;
//   sync pass r > 0  every lane decodes its chunk again, now from the exit state its left neighbour
//                    recorded in pass r - 1 (an interval's first chunk: from the true start).  Correct
//                    states spread from the interval starts, at least one chunk per pass, in practice
//                    across a whole scan in a few passes because most lanes had fallen into step inside
//                    their own chunk already.  A lane whose start state is the one it decoded from in the
//                    pass before keeps its results and does nothing: passes after convergence cost a launch;
//   scan             exclusive prefix sums over the chunks of each interval: the block a chunk starts
//                    in (an interval's first block is known: interval x ri x blocks per MCU) and the
//                    three DC predictors at its start (0 at an interval's start: T.81 F.2.1.3.1;
//                    reference jpeg.cpp:419-425);
//   write pass       every lane decodes its chunk once more from its neighbour's final exit state
//                    and this time stores the coefficients, the DC ones as running predictors
//                    (reference jpeg.cpp:335-345) -- and VERIFIES that it ends in the exit state
//                    recorded for it, an interval's last chunk that it ends with the interval's last
//                    block: if every lane does, the chain from every true start is consistent, i.e.
//                    this is the one true decode; if not (not yet synchronised after kJbSyncRounds
//                    passes, or corrupt data) the image's status word is set and the host decodes
//                    that image.
// A step is the same as in the interval decoder above; the tables, the stream ring and the layout
// of the output are shared.  (Idea: Klein & Wiseman 2003; Weissenberger & Schmidt 2018 for JPEG on GPUs.)

namespace {

struct ChunkLane {
  uint32_t k, blk, nblk;  // position in the block (0 = DC next), block within the MCU, blocks completed
};
// (the three DC values of a lane -- sync passes: sums of the DC differences so far; write pass: the DC
// predictors -- are separate variables on purpose: as neighbouring fields of the struct the compiler
// reads them as an array indexed by the component, and puts the struct into scratch memory -- whose
// set-up costs the first such kernel of a burst 130 us -- or, promoted, into 6 KiB more LDS)

// one symbol; kStore == false: only the state moves (and the DC differences are summed).  Returns
// false when the data cannot be what the state says (the caller decides what that means).
template <bool kStore>
__device__ __forceinline__ bool chunk_step(Stream &s, const JbHuffTables &t, const uint8_t *zz, uint32_t slots, uint32_t ny, uint32_t nb,
                                           ChunkLane &st, uint32_t &dc0, uint32_t &dc1, uint32_t &dc2, int16_t *block_out) {
  const int c = st.blk < ny ? 0 : (int)(st.blk - ny) + 1;
  const bool isdc = st.k == 0;
  const uint32_t slot = (slots >> (isdc ? c : 4 + c)) & 1u;
  const uint32_t bits = s.window();
  uint32_t e = ((const uint16_t *)t.acl)[(isdc ? 4096u : 0u) + slot * 2048u + (bits >> 21)];
  if (e == 0) e = long_code(bits, t, (int)((isdc ? 0u : 2u) + slot));
  const uint32_t len = e >> 8, rs = e & 0xffu;
  const uint32_t sz = isdc ? rs : (rs & 15u);
  const bool eob = !isdc && rs == 0;
  const uint32_t kk = st.k + (isdc ? 0u : rs == 0xf0u ? 16u : (rs >> 4));
  if (e == 0 || (isdc ? sz > 11 : (!eob && (kk > 63 || sz > 10)))) return false;
  int val = 0;
  if ((kStore || isdc) && sz) val = extend((bits << len) >> (32 - sz), (int)sz);
  if (isdc) {
    // (unsigned: the sums of a lane that is out of step are garbage and may wrap)
    const uint32_t pr = (c == 0 ? dc0 : c == 1 ? dc1 : dc2) + (uint32_t)val;
    if (kStore && ((int32_t)pr < -32768 || (int32_t)pr > 32767)) return false;  // a predictor the reference's short cannot hold
    dc0 = c == 0 ? pr : dc0;
    dc1 = c == 1 ? pr : dc1;
    dc2 = c == 2 ? pr : dc2;
    val = (int32_t)pr;
  }
  if (kStore && (isdc ? val != 0 : sz != 0)) JBH_STORE(block_out[zz[kk & 63]], (int16_t)val);
  s.consume(len + sz);
  st.k = isdc ? 1u : eob ? 64u : kk + (sz ? 1u : 0u);
  if (st.k > 63) {
    st.k = 0;
    st.nblk++;
    if (++st.blk == nb) st.blk = 0;
  }
  return true;
}

__device__ __forceinline__ void open_at_bit(Stream &s, uint32_t bit) {
  s.open(bit >> 3);
  s.off += bit & 7u;  // (start & 3) * 8 + (bit & 7) <= 31
}

// where a lane's chunk lies
struct ChunkExtent {
  uint32_t start_bit, end_bit;  // the chunk's bits in the image's clean scan (end: the interval's end at the latest)
  uint32_t seg;                 // its restart interval
  bool first, last;             // of its interval
};
__device__ __forceinline__ ChunkExtent chunk_extent(const JbHuffLaunch &p, const JbHuffImage &img, uint32_t ci) {
  const JbChunkDesc cd = p.chunks[img.state_off + ci];
  ChunkExtent x;
  x.seg = cd.seg & 0x7fffffffu;
  x.first = (cd.seg >> 31) != 0;
  uint32_t seg_end = p.starts[img.int_off + x.seg + 1];
  if (seg_end > img.scan_len) seg_end = img.scan_len;
  uint32_t start = cd.start < seg_end ? cd.start : seg_end;
  uint32_t end = start + img.chunk_bytes < seg_end ? start + img.chunk_bytes : seg_end;
  x.last = end == seg_end;
  x.start_bit = start * 8u;
  x.end_bit = end * 8u;
  return x;
}

}  // namespace

// one synchronisation pass (round 0: from the chunk starts; later rounds: from the left neighbour's exit state)
__global__ __launch_bounds__(kJbHuffLanes) void jb_huff_sync_kernel(const JbHuffLaunch p, const int round, const JbChunkState *src, JbChunkState *dst) {
  __shared__ __attribute__((aligned(16))) LdsTables lds;
  __shared__ __attribute__((aligned(16))) uint8_t rings[kJbHuffLanes * kRing];
  __shared__ uint8_t zz[64];
  const int tid = threadIdx.x;
  const JbHuffWg wg = p.sync_wgs[blockIdx.x];
  const JbHuffImage img = p.images[wg.image];
  const uint32_t slots = (uint32_t)img.dc_slot[0] | ((uint32_t)img.dc_slot[1] << 1) | ((uint32_t)img.dc_slot[2] << 2) |
                         ((uint32_t)img.ac_slot[0] << 4) | ((uint32_t)img.ac_slot[1] << 5) | ((uint32_t)img.ac_slot[2] << 6);
  const uint32_t ci = wg.first_interval + (uint32_t)tid;  // this lane's chunk
  const bool active = ci < img.n_chunks;
  const uint32_t nb = img.ny + 2;
  uint32_t bit = 0, end_bit = 0, nominal_start = 0;
  ChunkLane st{0, 0, 0};
  uint32_t dc0 = 0, dc1 = 0, dc2 = 0;
  bool skip = false;  // the start state is the one this chunk was decoded from last time: same results
  if (active) {
    const ChunkExtent x = chunk_extent(p, img, ci);
    end_bit = x.end_bit;
    nominal_start = x.start_bit;
    JbChunkState in{x.start_bit, 0};
    if (!x.first && round > 0) {
      const JbChunkState prev = src[img.state_off + ci - 1];
      in.bitpos = prev.bitpos;
      in.meta = prev.meta & 0xffffu;  // k and the block's place in the MCU
    }
    if (round > 0) {
      const JbChunkState last = p.state_in[img.state_off + ci];
      skip = last.bitpos == in.bitpos && last.meta == in.meta;
    }
    if (skip) dst[img.state_off + ci] = src[img.state_off + ci];  // (this chunk's exit state of the pass before)
    else p.state_in[img.state_off + ci] = in;
    bit = in.bitpos;
    st.k = in.meta & 0xffu;
    st.blk = (in.meta >> 8) & 0xffu;
    if (st.k > 63) st.k = 0;
    if (st.blk >= nb) st.blk = 0;
  }
  // a workgroup whose chunks are all in step has nothing to decode: the passes after convergence
  // cost a launch and these few loads, not 18 KiB of tables per workgroup
  if (!__syncthreads_or(active && !skip)) return;
  {
    const uint4 *src4 = (const uint4 *)(p.tables + img.table_set);
    uint4 *dst4 = (uint4 *)&lds.t;
    for (int i = tid; i < (int)(sizeof(JbHuffTables) / 16); i += kJbHuffLanes) dst4[i] = src4[i];
    if (tid < 64) zz[tid] = kZigZagDev[tid];
  }
  __syncthreads();
  if (bit > end_bit) bit = end_bit;
  Stream s;
  s.base = p.scan + img.scan_off;
  s.limit = (img.scan_len + 48u) & ~15u;
  s.ring = rings + tid * kRing;
  open_at_bit(s, (active && !skip) ? bit : 0u);
  bool live = active && !skip && bit < end_bit;
  // checkpoints: the first symbol boundary at or behind every kJbCheckpointBits bits of the chunk.  Pass 0
  // records the state and the counts there; a later pass that arrives at a checkpoint in the state
  // recorded for it has met the path of the chunk's previous decode and stops: what follows is known.
  JbCheckpoint *const cps = p.cps + (size_t)(img.state_off + (active ? ci : 0u)) * 8u;
  const uint32_t n_cp = img.chunk_bytes * 8u / kJbCheckpointBits - 1u;  // checkpoints inside a chunk of this image
  uint32_t cp_i = 0, next_b = nominal_start + kJbCheckpointBits;
  while (next_b <= bit && cp_i < n_cp) {  // a start behind the chunk's first checkpoints (a lane far out of step):
    if (live) ((uint4 *)(cps + cp_i))[0] = make_uint4(0xffffffffu, 0u, 0u, 0u);  // what they hold is no longer on this chunk's path
    cp_i++;
    next_b += kJbCheckpointBits;
  }
  bool met = false;
  uint32_t met_at = 0;
  for (;;) {
    if (__builtin_amdgcn_ballot_w64(live) == 0) break;
    if (live) s.top_up();
#pragma unroll
    for (int u = 0; u < 4; u++)
    if (live) {
      if (!chunk_step<false>(s, lds.t, zz, slots, img.ny, nb, st, dc0, dc1, dc2, nullptr)) {
        // not a possible continuation of this state: the lane is out of step (or the data is corrupt,
        // which the write pass will report) -- move on by one bit and expect a block to start; and
        // since a wrong guess of the block's place in the MCU (luma tables on a chroma block) is what
        // produces most impossible symbols and does not correct itself, try the next place: one
        // synchronisation pass fewer on 4:4:4 and on 4:2:0 files, +8 % on 4:2:0 batches
        s.consume(1);
        st.k = 0;
        if (++st.blk >= nb) st.blk = 0;  // (jumping between the luma and the chroma places instead: no better)
      }
      const uint32_t bp = (uint32_t)s.bitpos();
      live = bp < end_bit;
      if (bp >= next_b && cp_i < n_cp) {  // rare: at most n_cp times per chunk
        const uint32_t meta = st.k | (st.blk << 8);
        uint4 *rec = (uint4 *)(cps + cp_i);
        if (round > 0 && live) {
          const uint4 old = rec[0];
          if (old.x == bp && old.y == meta) {
            met = true;
            met_at = cp_i;
            live = false;
          }
        }
        if (!met) {
          rec[0] = make_uint4(bp, meta, st.nblk, dc0);
          rec[1] = make_uint4(dc1, dc2, 0u, 0u);
        }
        cp_i++;
        next_b += kJbCheckpointBits;
      }
    }
  }
  if (active && !skip) {
    if (met) {
      // the rest of the chunk is what the previous decode of this chunk found: its exit state, and its
      // counts shifted by the difference of the counts at the meeting place; the later checkpoints
      // (recorded relative to the old counts) move by the same amounts
      const JbChunkState old_exit = src[img.state_off + ci];
      const uint4 old_sum = *(const uint4 *)(p.dcsum + 4 * (size_t)(img.state_off + ci));
      const uint4 a = ((const uint4 *)(cps + met_at))[0], b = ((const uint4 *)(cps + met_at))[1];
      const uint32_t d_n = st.nblk - a.z, d0 = dc0 - a.w, d1 = dc1 - b.x, d2 = dc2 - b.y;
      for (uint32_t i = met_at; i < n_cp; i++) {
        uint4 *rec = (uint4 *)(cps + i);
        const uint4 r0 = rec[0], r1 = rec[1];
        rec[0] = make_uint4(r0.x, r0.y, r0.z + d_n, r0.w + d0);
        rec[1] = make_uint4(r1.x + d1, r1.y + d2, 0u, 0u);
      }
      dst[img.state_off + ci] = JbChunkState{old_exit.bitpos, (old_exit.meta & 0xffffu) | ((((old_exit.meta >> 16) + d_n) & 0xffffu) << 16)};
      *(uint4 *)(p.dcsum + 4 * (size_t)(img.state_off + ci)) = make_uint4(old_sum.x + d0, old_sum.y + d1, old_sum.z + d2, 0u);
    } else {
      dst[img.state_off + ci] = JbChunkState{(uint32_t)s.bitpos(), st.k | (st.blk << 8) | ((st.nblk & 0xffffu) << 16)};
      *(uint4 *)(p.dcsum + 4 * (size_t)(img.state_off + ci)) = make_uint4(dc0, dc1, dc2, 0u);
    }
  }
}

// exclusive prefix sums over the chunks of each interval -- blocks completed, DC differences per
// component -- restarting at every interval's first chunk: one workgroup per image
__global__ __launch_bounds__(kJbHuffLanes) void jb_huff_scan_kernel(const JbHuffLaunch p, const JbChunkState *fin) {
  __shared__ uint4 part[kJbHuffLanes];
  __shared__ uint32_t restarted[kJbHuffLanes];
  const JbHuffImage img = p.images[p.sync_images[blockIdx.x]];
  const uint32_t n = img.n_chunks, per = (n + kJbHuffLanes - 1) / kJbHuffLanes;
  const uint32_t lo = threadIdx.x * per < n ? threadIdx.x * per : n, hi = lo + per < n ? lo + per : n;
  const uint32_t blocks_per_interval = img.ri * (img.ny + 2);
  const uint4 *sums = (const uint4 *)p.dcsum;
  uint4 acc = make_uint4(0, 0, 0, 0);
  uint32_t any = 0;
  for (uint32_t i = lo; i < hi; i++) {
    const uint32_t sg = p.chunks[img.state_off + i].seg;
    if (sg >> 31) {
      acc = make_uint4((sg & 0x7fffffffu) * blocks_per_interval, 0, 0, 0);
      any = 1;
    }
    const uint4 d = sums[img.state_off + i];
    acc.x += fin[img.state_off + i].meta >> 16;
    acc.y += d.x;
    acc.z += d.y;
    acc.w += d.z;
  }
  part[threadIdx.x] = acc;
  restarted[threadIdx.x] = any;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint4 run = make_uint4(0, 0, 0, 0);
    for (int i = 0; i < kJbHuffLanes; i++) {
      const uint4 v = part[i];
      part[i] = run;
      if (restarted[i]) run = v;  // (v is absolute from the last interval start in lane i's range on)
      else run = make_uint4(run.x + v.x, run.y + v.y, run.z + v.z, run.w + v.w);
    }
  }
  __syncthreads();
  acc = part[threadIdx.x];
  uint4 *base = (uint4 *)p.base;
  for (uint32_t i = lo; i < hi; i++) {
    const uint32_t sg = p.chunks[img.state_off + i].seg;
    if (sg >> 31) acc = make_uint4((sg & 0x7fffffffu) * blocks_per_interval, 0, 0, 0);
    base[img.state_off + i] = acc;
    const uint4 d = sums[img.state_off + i];
    acc.x += fin[img.state_off + i].meta >> 16;
    acc.y += d.x;
    acc.z += d.y;
    acc.w += d.z;
  }
}

// the writing pass: decode from the left neighbour's final exit state, store, verify
__global__ __launch_bounds__(kJbHuffLanes) void jb_huff_write_kernel(const JbHuffLaunch p, const JbChunkState *fin) {
  __shared__ __attribute__((aligned(16))) LdsTables lds;
  __shared__ __attribute__((aligned(16))) uint8_t rings[kJbHuffLanes * kRing];
  __shared__ uint8_t zz[64];
  const int tid = threadIdx.x;
  const JbHuffWg wg = p.sync_wgs[blockIdx.x];
  const JbHuffImage img = p.images[wg.image];
  const uint32_t slots = (uint32_t)img.dc_slot[0] | ((uint32_t)img.dc_slot[1] << 1) | ((uint32_t)img.dc_slot[2] << 2) |
                         ((uint32_t)img.ac_slot[0] << 4) | ((uint32_t)img.ac_slot[1] << 5) | ((uint32_t)img.ac_slot[2] << 6);
  {
    const uint4 *src4 = (const uint4 *)(p.tables + img.table_set);
    uint4 *dst4 = (uint4 *)&lds.t;
    for (int i = tid; i < (int)(sizeof(JbHuffTables) / 16); i += kJbHuffLanes) dst4[i] = src4[i];
    if (tid < 64) zz[tid] = kZigZagDev[tid];
  }
  __syncthreads();
  const uint32_t ci = wg.first_interval + (uint32_t)tid;
  const bool active = ci < img.n_chunks;
  const uint32_t nb = img.ny + 2;
  uint32_t bit = 0, end_bit = 0, block = 0, block_end = 0, base0 = 0, base1 = 0, base2 = 0;
  bool last = false;
  ChunkLane st{0, 0, 0};
  uint32_t dc0 = 0, dc1 = 0, dc2 = 0;
  JbChunkState want{0, 0};
  if (active) {
    const ChunkExtent x = chunk_extent(p, img, ci);
    end_bit = x.end_bit;
    last = x.last;
    bit = x.start_bit;
    want = fin[img.state_off + ci];
    const uint4 b = ((const uint4 *)p.base)[img.state_off + ci];
    block = b.x;
    dc0 = base0 = b.y;
    dc1 = base1 = b.z;
    dc2 = base2 = b.w;
    // the blocks of this chunk's interval end here (the padding bits behind them are not symbols)
    const uint32_t m1 = (x.seg + 1) * img.ri < img.n_mcus ? (x.seg + 1) * img.ri : img.n_mcus;
    block_end = m1 * nb;
    if (!x.first) {
      const JbChunkState prev = fin[img.state_off + ci - 1];
      bit = prev.bitpos;
      st.k = prev.meta & 0xffu;
      st.blk = (prev.meta >> 8) & 0xffu;
      if (st.k > 63) st.k = 0;
      if (st.blk >= nb) st.blk = 0;
    }
  }
  const bool overran = active && bit > end_bit;  // the chunk before consumed bits beyond this one's (= the interval's) end
  if (bit > end_bit) bit = end_bit;
  if (block_end > img.n_blocks) block_end = img.n_blocks;  // (the output is sized for n_blocks)
  Stream s;
  s.base = p.scan + img.scan_off;
  s.limit = (img.scan_len + 48u) & ~15u;
  s.ring = rings + tid * kRing;
  open_at_bit(s, active ? bit : 0u);
  int16_t *const coef = (int16_t *)((uint8_t *)p.coef + img.coef_off);
  uint32_t err = 0;
  bool live = active && bit < end_bit && block < block_end;
  // (four steps per top-up, written as an unrolled inner loop: with the step counter tested in the
  // loop body the compiler copied a dozen loop-carried registers back and forth in every iteration)
  for (;;) {
    if (__builtin_amdgcn_ballot_w64(live) == 0) break;
    if (live) s.top_up();
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (live) {
        if (!chunk_step<true>(s, lds.t, zz, slots, img.ny, nb, st, dc0, dc1, dc2, coef + (int64_t)(block + st.nblk) * 64)) {
          err |= 1;  // reference jpeg.cpp:372-385: the stream is corrupt (or the chunks are not in step: bit 2 below)
          live = false;
        } else {
          live = (uint32_t)s.bitpos() < end_bit && block + st.nblk < block_end;
        }
      }
    }
  }
  if (active) {
    if (last) {
      // the interval's data ends before its blocks do, or its last symbol reaches beyond its last byte
      // (the host decoder's "entropy-coded data ends early": jb_frontend.cpp decode_interval)
      if (block + st.nblk != block_end || st.k != 0 || (uint32_t)s.bitpos() > end_bit || overran) err |= 2;
    } else {
      // this chunk must end where the synchronisation passes said it would, after as many blocks, and with
      // the DC differences adding up to what they recorded (the predictors of the chunks behind it rest on those)
      const uint4 sums = *(const uint4 *)(p.dcsum + 4 * (size_t)(img.state_off + ci));
      if ((uint32_t)s.bitpos() != want.bitpos || (st.k | (st.blk << 8) | ((st.nblk & 0xffffu) << 16)) != want.meta ||
          dc0 - base0 != sums.x || dc1 - base1 != sums.y || dc2 - base2 != sums.z)
        err |= 4;
    }
    if (err) atomicOr(p.status + wg.image, err);
  }
}

// A small packed submission (one image) is fetched from the pinned host blob by a kernel instead of a
// copy-engine transfer: the decoding kernels behind it on the stream then start without the ~100 us
// a compute queue waits for the copy engine's completion (one 1080p decode(bytes): 1.2 ms in all).
__global__ __launch_bounds__(256) void jb_huff_fetch_kernel(uint4 *dst, const uint4 *src, uint32_t n16) {
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n16; i += gridDim.x * 256u) dst[i] = src[i];
}
// the coefficient area and the status words are zeroed by a kernel of this library as well (the decoders store
// non-zero coefficients only)
__global__ __launch_bounds__(256) void jb_huff_zero_kernel(uint4 *dst, uint64_t n16) {
  const uint4 z = make_uint4(0, 0, 0, 0);
  for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256u) dst[i] = z;
}
hipError_t jbk_huff_zero(void *d_dst, size_t bytes, hipStream_t stream) {
  (void)hipGetLastError();
  const uint64_t n16 = (bytes + 15) / 16;
  uint64_t blocks = (n16 + 255u) / 256u;
  if (blocks > 8192u) blocks = 8192u;
  hipLaunchKernelGGL(jb_huff_zero_kernel, dim3((unsigned)(blocks ? blocks : 1u)), dim3(256), 0, stream, (uint4 *)d_dst, n16);
  return hipGetLastError();
}
hipError_t jbk_huff_fetch(void *d_dst, const void *h_pinned_src, size_t bytes, hipStream_t stream) {
  (void)hipGetLastError();
  const uint32_t n16 = (uint32_t)((bytes + 15) / 16);
  unsigned blocks = (n16 + 255u) / 256u;
  if (blocks > 1024u) blocks = 1024u;
  hipLaunchKernelGGL(jb_huff_fetch_kernel, dim3(blocks ? blocks : 1u), dim3(256), 0, stream, (uint4 *)d_dst, (const uint4 *)h_pinned_src, n16);
  return hipGetLastError();
}

hipError_t jbk_huff_launch(const JbHuffLaunch &p, hipStream_t stream) {
  (void)hipGetLastError();
  if (p.n_wgs > 0) hipLaunchKernelGGL(jb_huff_kernel, dim3((unsigned)p.n_wgs), dim3(kJbHuffLanes), 0, stream, p);
  if (p.n_sync_wgs > 0 && p.n_sync_images > 0) {
    const dim3 grid((unsigned)p.n_sync_wgs), block(kJbHuffLanes);
    // JPEGBLK_HUFF_EXTRA_LDS=N (experiment): N bytes of unused dynamic LDS per workgroup of the decoding
    // kernels -- fewer workgroups per CU; how much the chunk decoder depends on occupancy
    static const unsigned extra_lds = getenv("JPEGBLK_HUFF_EXTRA_LDS") ? (unsigned)atoi(getenv("JPEGBLK_HUFF_EXTRA_LDS")) : 0u;
    const JbChunkState *fin = nullptr;
    const int rounds = p.sync_rounds > 0 ? p.sync_rounds : kJbSyncRounds;
    for (int r = 0; r < rounds; r++) {
      const JbChunkState *src = (r & 1) ? p.state_a : p.state_b;
      JbChunkState *dst = (r & 1) ? p.state_b : p.state_a;
      hipLaunchKernelGGL(jb_huff_sync_kernel, grid, block, extra_lds, stream, p, r, src, dst);
      fin = dst;
    }
    hipLaunchKernelGGL(jb_huff_scan_kernel, dim3((unsigned)p.n_sync_images), block, 0, stream, p, fin);
    hipLaunchKernelGGL(jb_huff_write_kernel, grid, block, extra_lds, stream, p, fin);
  }
  return hipGetLastError();
}
