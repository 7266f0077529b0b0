// jb_huff.hip -- Huffman decoding of baseline scans ON THE DEVICE (gfx950).  Beyond the reference
// (its decodeHuffman, jpeg.cpp:405-446, is serial host code and north_star keeps the entropy stage
// on the host): SURVEY.md section 8(f) rank 4.  Progressive / multi-scan files stay on the host
// (jb_frontend_ext.cpp).
//
// A Huffman stream can only be decoded from its start -- but a decoder started at a wrong place, in
// a wrong state, falls into step with the true symbol sequence after a few symbols or blocks and
// stays in step (Klein & Wiseman 2003; Weissenberger & Schmidt 2018 for JPEG on GPUs).  The DC
// predictors reset at every restart marker (T.81 F.2.1.3.1; reference jpeg.cpp:419-425), so the
// restart intervals of a scan are independent streams (a scan without DRI is one interval).  The
// host removes the byte stuffing and records where each interval starts (jbe::unstuff: memchr
// speed); every interval is cut into chunks of 128 (small scans: 64) bytes from its own first byte,
// ONE LANE PER CHUNK, 256 consecutive chunks of one image per workgroup:
//
//   synchronisation  jb_huff_sync_kernel.  Pass 0: every lane decodes its chunk from the chunk's first
//                    bit, assuming "the DC symbol of the MCU's first block is next" (true for the first
//                    chunk of an interval), and leaves its EXIT state in LDS: the bit at which the first
//                    symbol of the next chunk starts, the position k in the block, the block's place in
//                    the MCU -- with the blocks it completed and the DC differences it summed, per
//                    component.  Pass r > 0, same launch: a lane whose left neighbour's exit state is not
//                    the state it decoded from decodes again, from there -- and stops as soon as it MEETS
//                    the path of its previous decode (checkpoints: the first symbol boundary behind every
//                    256 bits, state in LDS, counts in device memory): from equal (bit, k, block) on
//                    everything is the same, so the old exit state stands and the counts shift by what
//                    differed up to the meeting place.  Passes repeat until no lane of the workgroup
//                    changed: the 256 chunks then are ONE consistent decode from the state the
//                    workgroup's first lane assumed.  A second launch hands every workgroup the final
//                    exit state of the chunk before its first and repairs the few chunks that change.
//   writing pass     jb_huff_write_kernel: segmented prefix sums over the chunks of each interval (the
//                    workgroups' totals, then a scan inside the workgroup) give every chunk the block it
//                    starts in (an interval's first block is known: interval x ri x blocks per MCU) and
//                    the three DC predictors there (0 at an interval's start); every lane decodes its
//                    chunk once more from its neighbour's final exit state and stores the coefficients,
//                    the DC ones as running predictors (reference jpeg.cpp:335-345) -- and VERIFIES: the
//                    state it starts from is the one the synchronisation decoded it from, it ends in the
//                    exit state recorded for it with the recorded counts, an interval's last chunk ends
//                    with the interval's last block.  If every lane does, the chain from every true start
//                    is consistent, i.e. this is the one true decode; if not (corrupt data, or not in step
//                    yet) the image's status word is set and the host decodes that image.
//
// The step (jb_huff_core.h): one symbol, DC and AC alike -- the 32 bits at the lane's position out of
// two LDS dwords of the transposed chunk (conflict free whatever the lanes' positions), one 16-bit
// table entry that holds bits consumed, magnitude size and the advance of k, a branch-free EXTEND,
// branch-free state update.  No memory access but LDS in the synchronisation passes' loop (device
// memory only where a checkpoint is crossed); the writing pass adds one 2-byte store of the
// de-zigzagged coefficient straight into the layout the fused pixel kernel reads (the area is
// zeroed once per submission; the L2 gathers a line's stores).
// Safety: every position stays inside the lane's rows of LDS (u < u_end <= 32 + 24 + 8 * chunk, a
// symbol takes at most 27 bits); k only grows inside a block; output indices come from host-validated
// counts and are checked against the interval's last block.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "jb_huff_core.h"
#include "jb_kernels.h"

namespace {

constexpr uint32_t L = kJbhLanes;
constexpr int kMaxPasses = (int)L + 4;  // correct states travel at least one chunk per pass

// 2 * natural index of zig-zag position i (ITU-T T.81 Figure A.6; reference types.hpp:23-31)
__constant__ uint8_t kZz2Dev[64] = {0,   2,   16,  32,  18,  4,   6,   20,  34,  48, 64,  50,  36,  22,  8,   10,
                                    24,  38,  52,  66,  80,  96,  82,  68,  54,  40, 26,  12,  14,  28,  42,  56,
                                    70,  84,  98,  112, 114, 100, 86,  72,  58,  44, 30,  46,  60,  74,  88,  102,
                                    116, 118, 104, 90,  76,  62,  78,  92,  106, 120, 122, 108, 94,  110, 124, 126};

// LDS of a workgroup: the transposed stream, the checkpoint states, the exit states the lanes hand to their
// right neighbours, 256 bytes of odds and ends, the table set
template <uint32_t kMaxChunk>
struct Lay {
  static constexpr uint32_t kRows = jbh_rows(kMaxChunk);
  static constexpr uint32_t kNcp = kMaxChunk * 8u / kJbCheckpointBits;  // boundaries a chunk can cross
  static_assert(kNcp <= kJbCheckpoints, "records per chunk");
  static constexpr uint32_t off_stream = 0;
  static constexpr uint32_t off_cp = off_stream + kRows * L * 4u;
  static constexpr uint32_t off_xbit = off_cp + kNcp * L * 4u;
  static constexpr uint32_t off_xmeta = off_xbit + L * 4u;
  static constexpr uint32_t off_misc = off_xmeta + L * 4u;
  static constexpr uint32_t off_tab = off_misc + 256u;
  static uint32_t bytes(uint32_t n_tabs) { return off_tab + n_tabs * kJbT1Entries * 2u + kJbT2Tables * kJbT2Entries * 2u; }
};

struct ChunkGeo {
  uint32_t start, end;  // the chunk's bytes in the image's clean scan (end: the interval's end at the latest)
  uint32_t seg;         // its restart interval
  bool first, last;     // of its interval
};
__device__ __forceinline__ ChunkGeo chunk_geo(const JbHuffLaunch &p, const JbHuffImage &img, uint32_t gidx) {
  const JbChunkDesc cd = p.chunks[gidx];
  ChunkGeo x;
  x.seg = cd.seg & 0x7fffffffu;
  x.first = (cd.seg >> 31) != 0;
  uint32_t seg_end = p.starts[img.int_off + x.seg + 1];
  if (seg_end > img.scan_len) seg_end = img.scan_len;
  x.start = cd.start < seg_end ? cd.start : seg_end;
  x.end = x.start + img.chunk_bytes < seg_end ? x.start + img.chunk_bytes : seg_end;
  x.last = x.end == seg_end;
  return x;
}

// the lane's rows of the transposed stream: big-endian dwords from byte A = (start & ~3) - 4 of the image's scan
// (behind the last image's scan lie 64 zero bytes and the submission's device scratch: nothing is read out of bounds)
template <uint32_t kRows>
__device__ __forceinline__ void load_stream(uint32_t *stream, const uint8_t *img_scan, uint32_t start, uint32_t tid) {
  const uint8_t *src = img_scan + (int32_t)jbh_base_byte(start);
#pragma unroll
  for (uint32_t q = 0; q < (kRows + 3u) / 4u; q++) {
    const uint4 v = *(const uint4 *)(src + 16u * q);
    const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (uint32_t i = 0; i < 4; i++)
      if (4u * q + i < kRows) stream[(4u * q + i) * L + tid] = __builtin_bswap32(d[i]);
  }
}

__device__ __forceinline__ void load_tables(uint16_t *tab, const JbHuffLaunch &p, const JbHuffImage &img, uint32_t tid) {
  const JbHuffTables *set = p.tables + img.table_set;
  const uint4 *s1 = (const uint4 *)set->t1;
  uint4 *d = (uint4 *)tab;
  const uint32_t n1 = img.n_tabs * kJbT1Entries * 2u / 16u;
  for (uint32_t i = tid; i < n1; i += L) d[i] = s1[i];
  const uint4 *s2 = (const uint4 *)set->t2;
  constexpr uint32_t n2 = kJbT2Tables * kJbT2Entries * 2u / 16u;
  for (uint32_t i = tid; i < n2; i += L) d[n1 + i] = s2[i];
}

// Segmented sums over chunks: (f, n, d) = "an interval started here or to the left (then n counts from the image's
// first block, d from that interval's start)", blocks, DC sums.  a (+) b with a to the left of b.
struct Seg {
  uint32_t f, n, d0, d1, d2;
};
__device__ __forceinline__ Seg seg_combine(const Seg &a, const Seg &b) {
  Seg r;
  r.f = a.f | b.f;
  r.n = b.f ? b.n : a.n + b.n;
  r.d0 = b.f ? b.d0 : a.d0 + b.d0;
  r.d1 = b.f ? b.d1 : a.d1 + b.d1;
  r.d2 = b.f ? b.d2 : a.d2 + b.d2;
  return r;
}
__device__ __forceinline__ Seg seg_shfl_up(const Seg &x, int off) {
  Seg r;
  r.f = __shfl_up(x.f, off);
  r.n = __shfl_up(x.n, off);
  r.d0 = __shfl_up(x.d0, off);
  r.d1 = __shfl_up(x.d1, off);
  r.d2 = __shfl_up(x.d2, off);
  return r;
}
// inclusive scan over the 256 lanes of the workgroup (scratch: 4 x 5 words of LDS); *total = all 256
__device__ __forceinline__ Seg seg_scan_wg(Seg x, uint32_t *scratch, uint32_t tid, Seg *total) {
  const uint32_t lane = tid & 63u, wave = tid >> 6;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const Seg t = seg_shfl_up(x, off);
    if (lane >= (uint32_t)off) x = seg_combine(t, x);
  }
  __syncthreads();  // (the scratch may still be read from an earlier use)
  if (lane == 63u) {
    uint32_t *s = scratch + wave * 5u;
    s[0] = x.f, s[1] = x.n, s[2] = x.d0, s[3] = x.d1, s[4] = x.d2;
  }
  __syncthreads();
  Seg run{0, 0, 0, 0, 0};
  Seg all{0, 0, 0, 0, 0};
#pragma unroll
  for (uint32_t w = 0; w < L / 64u; w++) {
    const uint32_t *s = scratch + w * 5u;
    const Seg t{s[0], s[1], s[2], s[3], s[4]};
    if (w < wave) run = seg_combine(run, t);
    all = seg_combine(all, t);
  }
  *total = all;
  return seg_combine(run, x);
}

// what the workgroups of this image to the left of workgroup w_me add up to, folded in order: the last one that
// holds an interval start counts from the image's first block (its DC sums from that start), the ones behind it add
// up.  `red`: 8 words of LDS.
__device__ __forceinline__ Seg fold_left_workgroups(const JbHuffLaunch &p, uint32_t w_first, uint32_t w_me, uint32_t *red, uint32_t tid) {
  if (tid < 8) red[tid] = 0;
  __syncthreads();
  uint32_t last_f = 0;
  for (uint32_t w = w_first + tid; w < w_me; w += L)
    if (p.wgsum[w].has_first) last_f = w - w_first + 1u;
  if (last_f) atomicMax(&red[0], last_f);
  __syncthreads();
  const uint32_t from = red[0];  // (0 only when there is no workgroup to the left: an image's first chunk starts an interval)
  uint32_t a_n = 0, a0 = 0, a1 = 0, a2 = 0;
  for (uint32_t w = w_first + tid; w < w_me; w += L) {
    if (w - w_first + 1u >= from) {
      const JbWgSum s = p.wgsum[w];
      a_n += s.blocks, a0 += s.dc[0], a1 += s.dc[1], a2 += s.dc[2];
    }
  }
  if (a_n) atomicAdd(&red[1], a_n);
  if (a0) atomicAdd(&red[2], a0);
  if (a1) atomicAdd(&red[3], a1);
  if (a2) atomicAdd(&red[4], a2);
  __syncthreads();
  return Seg{from ? 1u : 0u, red[1], red[2], red[3], red[4]};
}
// the exclusive scan of the workgroup's lanes behind `carry`: what lies to the left of this lane
__device__ __forceinline__ Seg scan_left(const Seg &mine, const Seg &carry, uint32_t *scratch, uint32_t tid, Seg *total) {
  const Seg incl = seg_scan_wg(mine, scratch, tid, total);
  Seg left = seg_shfl_up(incl, 1);
  if ((tid & 63u) == 0) {
    // (the wave's first lane: the waves to the left, as seg_scan_wg left them in the scratch)
    left = Seg{0, 0, 0, 0, 0};
    for (uint32_t w = 0; w < (tid >> 6); w++) {
      const uint32_t *s = scratch + w * 5u;
      left = seg_combine(left, Seg{s[0], s[1], s[2], s[3], s[4]});
    }
  }
  return seg_combine(carry, left);
}

}  // namespace

// ---- synchronisation ----------------------------------------------------------------------------
constexpr uint32_t kMet = 0x80000000u;  // a lane's position once it has met its previous path: kMet | checkpoint

template <uint32_t kMaxChunk>
__global__ __launch_bounds__(kJbHuffLanes) void jb_huff_sync_kernel(const JbHuffLaunch p, const int launch) {
  using Ly = Lay<kMaxChunk>;
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  uint32_t *const stream = (uint32_t *)(lds + Ly::off_stream);
  uint32_t *const cpst = (uint32_t *)(lds + Ly::off_cp);
  uint32_t *const xbit = (uint32_t *)(lds + Ly::off_xbit);
  uint32_t *const xmeta = (uint32_t *)(lds + Ly::off_xmeta);
  uint32_t *const misc = (uint32_t *)(lds + Ly::off_misc);
  uint16_t *const tab = (uint16_t *)(lds + Ly::off_tab);
  const uint32_t tid = threadIdx.x;
  const JbHuffWg wg = p.sync_wgs[blockIdx.x];
  const JbHuffImage img = p.images[wg.image];
  // lanes 0..kJbWarmChunks-1: the chunks in front of the workgroup's own (none in front of an image's first chunk)
  const uint32_t ci = wg.first_chunk + tid - kJbWarmChunks;  // (wraps below 0 for the lanes in front of chunk 0)
  const bool owned = tid >= kJbWarmChunks && ci < img.n_chunks;
  // (the warm-up lanes only matter for the first launch: afterwards the own chunks' states come from device memory)
  const bool active = ci < img.n_chunks && (owned || launch == 0);
  const uint32_t gidx = img.state_off + (ci < img.n_chunks ? ci : img.n_chunks - 1u);  // (idle lanes: valid addresses, nothing stored)
  const ChunkGeo g = chunk_geo(p, img, gidx);
  const uint32_t u0 = jbh_u_of_bit(g.start, g.start * 8u), u_end = u0 + (g.end - g.start) * 8u;
  const size_t n_all = p.n_chunks_total;

  JbChunkState entry{g.start * 8u, 0u}, exitst{0u, 0u};
  bool changed = active;
  if (launch > 0) {
    // nothing but a new state at the workgroup's first chunk can change anything in this launch
    bool c0 = false;
    JbChunkState ne{0u, 0u};
    if (tid == kJbWarmChunks && owned && !g.first) {
      ne = p.exit[gidx - 1u];
      const JbChunkState en = p.entry[gidx];
      c0 = ne.bitpos != en.bitpos || ((ne.meta ^ en.meta) & 0xffffu) != 0;
    }
    if (!__syncthreads_or(c0)) return;
    entry = p.entry[gidx];
    exitst = p.exit[gidx];
    changed = c0;
    if (c0) entry = JbChunkState{ne.bitpos, ne.meta & 0xffffu};
  }
  load_tables(tab, p, img, tid);
  if (active) load_stream<Ly::kRows>(stream, p.scan + img.scan_off, g.start, tid);
#pragma unroll
  for (uint32_t i = 0; i < Ly::kNcp; i++) cpst[i * L + tid] = launch > 0 ? p.cps[i * n_all + gidx] : 0xffffffffu;
  __syncthreads();

  JbhCtx cx;
  cx.scol = stream + tid;
  cx.tab = tab;
  cx.zz2 = nullptr;
  cx.lut_ac = img.lut_ac, cx.lut_dc = img.lut_dc, cx.lut_comp = img.lut_comp;
  cx.nb4 = img.nb * 4u;
  cx.blk_bytes = img.blk_bytes;
  cx.t2_first = img.n_tabs * kJbT1Entries;
  uint32_t *const cpl = cpst + tid;  // this lane's checkpoints: cpl[i * L]

  for (int pass = 0; pass < kMaxPasses; pass++) {
    // (wave-uniform entry: a wave none of whose lanes changed goes straight to the barrier)
    if (__builtin_amdgcn_ballot_w64(changed) != 0) {
      JbhLane st;
      {
        const uint32_t bit = entry.bitpos < g.start * 8u ? g.start * 8u : entry.bitpos;
        const uint32_t u = jbh_u_of_bit(g.start, bit);
        st.u = (changed && u < u_end) ? u : u_end;
        st.k = entry.meta & 0xffu;
        st.blk4 = ((entry.meta >> 8) & 0xffu) * 4u;
        if (st.k > 63u) st.k = 0;
        if (st.blk4 >= cx.nb4) st.blk4 = 0;
        st.nblk = 0;
      }
      uint32_t none = 0;
      while (st.u < u_end) {
        const uint32_t up = st.u;
        (void)jbh_step<false>(cx, st, none, none, none, nullptr, 0u);
        if (((st.u ^ up) >> 8) != 0) {
          // a checkpoint: the first symbol boundary behind a multiple of 256 bits.  In the state recorded there (and
          // still inside the chunk): the path of this chunk's previous decode -- what follows is known
          uint32_t *rec = cpl + ((st.u >> 8) - 1u) * L;
          const uint32_t s = jbh_pack_state(st), old = *rec;
          const bool met = (st.u < u_end) & (((old ^ s) & 0xfffffu) == 0);  // (&: no branch)
          *rec = met ? old : s;
          st.u = met ? (kMet | ((st.u >> 8) - 1u)) : st.u;
        }
      }
      JBH_TRACE_PASS_END(pass);
      if (changed) {
        if (st.u >= kMet) {
          // the rest of the chunk is what the previous decode found: its exit state, its block count shifted by the
          // difference of the counts at the meeting place; the later checkpoints move by the same amount
          const uint32_t met_i = st.u & 0xffu;
          const uint32_t dn = st.nblk - (cpl[met_i * L] >> 20);
          for (uint32_t i = met_i; i < Ly::kNcp; i++) cpl[i * L] += dn << 20;
          exitst.meta = (exitst.meta & 0xffffu) | ((((exitst.meta >> 16) + dn) & 0xffffu) << 16);
        } else {
          exitst.bitpos = jbh_bit_of_u(g.start, st.u);
          exitst.meta = st.k | (st.blk4 << 6) | ((st.nblk & 0xffffu) << 16);
        }
      }
    }
    xbit[tid] = exitst.bitpos;
    xmeta[tid] = exitst.meta & 0xffffu;
    __syncthreads();
    changed = false;
    if (active && !g.first && tid > (launch == 0 ? 0u : kJbWarmChunks)) {
      const uint32_t nbit = xbit[tid - 1u], nmeta = xmeta[tid - 1u];
      if (nbit != entry.bitpos || nmeta != (entry.meta & 0xffffu)) {
        entry = JbChunkState{nbit, nmeta};
        changed = true;
      }
    }
    if (!__syncthreads_or(changed)) break;
  }
  if (owned) {
    p.entry[gidx] = entry;
    p.exit[gidx] = exitst;
#pragma unroll
    for (uint32_t i = 0; i < Ly::kNcp; i++) p.cps[i * n_all + gidx] = cpl[i * L];
  }
  // what the workgroup's own chunks add up to, for the writing pass
  Seg mine{0, 0, 0, 0, 0};
  if (owned) {
    mine.f = g.first ? 1u : 0u;
    mine.n = (exitst.meta >> 16) + (g.first ? g.seg * img.ri * img.nb : 0u);
  }
  Seg total;
  (void)seg_scan_wg(mine, misc + 16, tid, &total);
  if (tid == 0) {
    JbWgSum *o = p.wgsum + img.wg0 + wg.first_chunk / kJbOwnChunks;
    o->has_first = total.f;
    o->blocks = total.n;
  }
}

// ---- the writing pass: decode from the left neighbour's final exit state, store, verify ----------------
template <uint32_t kMaxChunk>
__global__ __launch_bounds__(kJbHuffLanes) void jb_huff_write_kernel(const JbHuffLaunch p) {
  using Ly = Lay<kMaxChunk>;
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  uint32_t *const stream = (uint32_t *)(lds + Ly::off_stream);
  uint32_t *const misc = (uint32_t *)(lds + Ly::off_misc);
  uint16_t *const tab = (uint16_t *)(lds + Ly::off_tab);
  const uint32_t tid = threadIdx.x;
  const JbHuffWg wg = p.wgs[blockIdx.x];
  const JbHuffImage img = p.images[wg.image];
  const uint32_t ci = wg.first_chunk + tid;
  const bool active = tid < kJbOwnChunks && ci < img.n_chunks;
  const uint32_t gidx = img.state_off + (active ? ci : img.n_chunks - 1u);
  const ChunkGeo g = chunk_geo(p, img, gidx);
  const uint32_t u0 = jbh_u_of_bit(g.start, g.start * 8u), u_end = u0 + (g.end - g.start) * 8u;
  const uint32_t bpi = img.ri * img.nb;  // blocks per restart interval

  load_tables(tab, p, img, tid);
  if (tid < 16) ((uint32_t *)misc)[tid] = ((const uint32_t *)kZz2Dev)[tid];
  if (active) load_stream<Ly::kRows>(stream, p.scan + img.scan_off, g.start, tid);

  // where the chunk starts: the state, and the block
  uint32_t err = 0;
  JbChunkState entry{g.start * 8u, 0u}, want{0u, 0u};
  uint32_t block = g.seg * bpi;
  if (img.needs_sync) {
    Seg mine{0, 0, 0, 0, 0};
    if (active) {
      want = p.exit[gidx];
      mine.f = g.first ? 1u : 0u;
      mine.n = (want.meta >> 16) + (g.first ? g.seg * bpi : 0u);
      if (!g.first) {
        // the state this chunk starts from: its left neighbour's final exit state -- which must be the state the
        // synchronisation decoded this chunk from, or the chunks are not in step
        const JbChunkState prev = p.exit[gidx - 1u];
        const JbChunkState en = p.entry[gidx];
        entry = JbChunkState{prev.bitpos, prev.meta & 0xffffu};
        if (en.bitpos != prev.bitpos || ((en.meta ^ prev.meta) & 0xffffu) != 0) err |= 4u;
      }
    }
    const Seg carry = fold_left_workgroups(p, img.wg0, img.wg0 + wg.first_chunk / kJbOwnChunks, misc + 40, tid);
    Seg total;
    const Seg left = scan_left(mine, carry, misc + 16, tid, &total);
    if (!g.first) block = left.n;
  }
  __syncthreads();

  JbhCtx cx;
  cx.scol = stream + tid;
  cx.tab = tab;
  cx.zz2 = (const uint8_t *)misc;
  cx.lut_ac = img.lut_ac, cx.lut_dc = img.lut_dc, cx.lut_comp = img.lut_comp;
  cx.nb4 = img.nb * 4u;
  cx.blk_bytes = img.blk_bytes;
  cx.t2_first = img.n_tabs * kJbT1Entries;

  // the blocks of this chunk's interval end here (the padding bits behind them are not symbols)
  uint32_t block_end = (g.seg + 1u) * img.ri < img.n_mcus ? (g.seg + 1u) * bpi : img.n_mcus * img.nb;
  if (block_end > img.n_blocks) block_end = img.n_blocks;  // (the output is sized for n_blocks)
  JbhLane st;
  const bool overran = active && entry.bitpos > g.end * 8u;  // the chunk before consumed bits beyond this one's (= the interval's) end
  {
    const uint32_t bit = entry.bitpos < g.start * 8u ? g.start * 8u : entry.bitpos;
    const uint32_t u = jbh_u_of_bit(g.start, bit);
    st.u = u < u_end ? u : u_end;
    st.k = entry.meta & 0xffu;
    st.blk4 = ((entry.meta >> 8) & 0xffu) * 4u;
    if (st.k > 63u) st.k = 0;
    if (st.blk4 >= cx.nb4) st.blk4 = 0;
    st.nblk = 0;
  }
  const uint32_t k_in = st.k;
  uint32_t dc0 = 0, dc1 = 0, dc2 = 0;
  uint8_t *const coef = (uint8_t *)p.coef + img.coef_off;
  bool live = active && st.u < u_end && block < block_end;
  while (live) {
    if (jbh_step<true>(cx, st, dc0, dc1, dc2, coef, block)) {
      err |= 1u;  // reference jpeg.cpp:372-385: the stream is corrupt (or the chunks are not in step: bit 2)
      live = false;
    } else {
      live = st.u < u_end && block + st.nblk < block_end;
    }
  }
  JBH_TRACE_PASS_END(0);
  if (active) {
    if (g.last) {
      // the interval's data ends before its blocks do, or its last symbol reaches beyond its last byte
      // (the host decoder's "entropy-coded data ends early": jb_frontend.cpp decode_interval)
      if (block + st.nblk != block_end || st.k != 0 || st.u > u_end || overran) err |= 2u;
    } else if (img.needs_sync) {
      // this chunk must end where the synchronisation passes said it would, after as many blocks
      if (jbh_bit_of_u(g.start, st.u) != want.bitpos || (st.k | (st.blk4 << 6) | ((st.nblk & 0xffffu) << 16)) != want.meta) err |= 4u;
    }
    if (err) atomicOr(p.status + wg.image, err);
    // the blocks whose DC symbol lies in this chunk hold DC DIFFERENCES: jb_huff_dc_kernel makes predictors of them
    JbChunkDc o;
    o.dc[0] = dc0, o.dc[1] = dc1, o.dc[2] = dc2;
    o.first_block = block + (k_in != 0 ? 1u : 0u);
    o.count = err ? 0u : st.nblk + (st.k != 0 ? 1u : 0u) - (k_in != 0 ? 1u : 0u);
    o.pad[0] = o.pad[1] = o.pad[2] = 0;
    p.chunk_dc[gidx] = o;
  }
  if (img.needs_sync) {
    // what the workgroup's DC differences add up to (behind its last interval start), for the chunks to the right
    Seg mine{0, 0, 0, 0, 0};
    if (active) mine = Seg{g.first ? 1u : 0u, 0u, dc0, dc1, dc2};
    Seg total;
    (void)seg_scan_wg(mine, misc + 16, tid, &total);
    if (tid == 0) {
      JbWgSum *o = p.wgsum + img.wg0 + wg.first_chunk / kJbOwnChunks;
      o->dc[0] = total.d0, o->dc[1] = total.d1, o->dc[2] = total.d2;
    }
  }
}

// ---- DC differences -> DC predictors (reference jpeg.cpp:335-345: component[0] = coeff + previousDC) --------
// One lane per chunk again: the predictors at the chunk's start (0 at an interval's start: T.81 F.2.1.3.1,
// reference jpeg.cpp:419-425) are a segmented prefix sum of the chunks' sums; the lane then walks the blocks whose
// DC symbol lies in its chunk.  A predictor the coefficient format cannot hold is corrupt data (bit 0).
__global__ __launch_bounds__(kJbHuffLanes) void jb_huff_dc_kernel(const JbHuffLaunch p) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  uint32_t *const misc = (uint32_t *)lds;  // 256 bytes
  const uint32_t tid = threadIdx.x;
  const JbHuffWg wg = p.wgs[blockIdx.x];
  const JbHuffImage img = p.images[wg.image];
  const uint32_t ci = wg.first_chunk + tid;
  const bool active = tid < kJbOwnChunks && ci < img.n_chunks;
  const uint32_t gidx = img.state_off + (active ? ci : img.n_chunks - 1u);
  const bool first = (p.chunks[gidx].seg >> 31) != 0;
  JbChunkDc cd = p.chunk_dc[gidx];
  if (!active) cd.count = 0;
  uint32_t p0 = 0, p1 = 0, p2 = 0;
  if (img.needs_sync) {
    Seg mine{0, 0, 0, 0, 0};
    if (active) mine = Seg{first ? 1u : 0u, 0u, cd.dc[0], cd.dc[1], cd.dc[2]};
    const Seg carry = fold_left_workgroups(p, img.wg0, img.wg0 + wg.first_chunk / kJbOwnChunks, misc + 40, tid);
    Seg total;
    const Seg left = scan_left(mine, carry, misc + 16, tid, &total);
    if (!first) p0 = left.d0, p1 = left.d1, p2 = left.d2;
  }
  uint8_t *const coef = (uint8_t *)p.coef + img.coef_off;
  uint32_t pos4 = (cd.first_block % img.nb) * 4u;  // 4 * place of the block in its MCU
  uint32_t bad = 0;
  // (a chunk of 128 bytes holds about 20 blocks of a quality-75 stream: all their loads are in flight at once; a
  // batch of 8, three round trips to memory per lane, made this kernel 23 % of the device's time in a batch)
  constexpr uint32_t kBatch = 24;
  for (uint32_t i = 0; i < cd.count; i += kBatch) {
    int32_t diff[kBatch];
#pragma unroll
    for (uint32_t q = 0; q < kBatch; q++) diff[q] = i + q < cd.count ? *(const int16_t *)(coef + (size_t)jbh_mul24(cd.first_block + i + q, img.blk_bytes)) : 0;
#pragma unroll
    for (uint32_t q = 0; q < kBatch; q++) {
      if (i + q < cd.count) {
        const uint32_t c = jbh_ubfe(img.lut_comp, pos4, 4);
        const uint32_t pr = (c == 0 ? p0 : c == 1 ? p1 : p2) + (uint32_t)diff[q];
        bad |= (pr + 32768u) > 65535u ? 1u : 0u;
        p0 = c == 0 ? pr : p0;
        p1 = c == 1 ? pr : p1;
        p2 = c == 2 ? pr : p2;
        *(int16_t *)(coef + (size_t)jbh_mul24(cd.first_block + i + q, img.blk_bytes)) = (int16_t)pr;
        pos4 = pos4 + 4u == img.nb * 4u ? 0u : pos4 + 4u;
      }
    }
  }
  if (bad) atomicOr(p.status + wg.image, 1u);
}

// A small packed submission (one image) is fetched from the pinned host blob by a kernel instead of a
// copy-engine transfer: the decoding kernels behind it on the stream then start without the ~100 us
// a compute queue waits for the copy engine's completion.
__global__ __launch_bounds__(256) void jb_huff_fetch_kernel(uint4 *dst, const uint4 *src, uint32_t n16) {
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n16; i += gridDim.x * 256u) dst[i] = src[i];
}
// the coefficient area and the status words are zeroed by a kernel of this library as well (the decoder stores
// non-zero coefficients only)
// (`small`: a second, small region -- the submission's status words, at most 256 of them -- zeroed by the first
// workgroup of the same launch: a launch of its own cost 5 us per submission)
__global__ __launch_bounds__(256) void jb_huff_zero_kernel(uint4 *dst, uint64_t n16, uint4 *small, uint32_t small16) {
  const uint4 z = make_uint4(0, 0, 0, 0);
  if (blockIdx.x == 0 && threadIdx.x < small16) small[threadIdx.x] = z;
  for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256u) dst[i] = z;
}
hipError_t jbk_huff_zero(void *d_dst, size_t bytes, hipStream_t stream, void *d_small, size_t small_bytes) {
  (void)hipGetLastError();
  const uint64_t n16 = (bytes + 15) / 16;
  uint64_t blocks = (n16 + 255u) / 256u;
  if (blocks > 8192u) blocks = 8192u;
  const uint32_t small16 = d_small ? (uint32_t)((small_bytes + 15) / 16) : 0u;
  if (small16 > 256u) return hipErrorInvalidValue;
  hipLaunchKernelGGL(jb_huff_zero_kernel, dim3((unsigned)(blocks ? blocks : 1u)), dim3(256), 0, stream, (uint4 *)d_dst, n16, (uint4 *)d_small, small16);
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
  const dim3 block(kJbHuffLanes);
  const uint32_t n_tabs = p.max_tabs < 2u ? 2u : p.max_tabs > kJbMaxTabs ? kJbMaxTabs : p.max_tabs;
  const bool small = p.max_chunk_bytes <= 64u;
  const unsigned lds = small ? Lay<64>::bytes(n_tabs) : Lay<128>::bytes(n_tabs);
  if (p.n_sync_wgs > 0) {
    const int launches = p.sync_launches > 0 ? p.sync_launches : kJbSyncLaunches;
    for (int l = 0; l < launches; l++) {
      if (small) hipLaunchKernelGGL(jb_huff_sync_kernel<64>, dim3((unsigned)p.n_sync_wgs), block, lds, stream, p, l);
      else hipLaunchKernelGGL(jb_huff_sync_kernel<128>, dim3((unsigned)p.n_sync_wgs), block, lds, stream, p, l);
    }
  }
  if (p.n_wgs > 0) {
    if (small) hipLaunchKernelGGL(jb_huff_write_kernel<64>, dim3((unsigned)p.n_wgs), block, lds, stream, p);
    else hipLaunchKernelGGL(jb_huff_write_kernel<128>, dim3((unsigned)p.n_wgs), block, lds, stream, p);
    hipLaunchKernelGGL(jb_huff_dc_kernel, dim3((unsigned)p.n_wgs), block, 256, stream, p);
  }
  return hipGetLastError();
}
