// jb_kernels.hip -- fused dequantize -> 8x8 inverse DCT -> YCbCr->RGB for gfx950 (MI355X).
//
// Replaces the three whole-image passes of the reference (dequantize(), inverseDCT(),
// YCbCrToRGB(): jpeg.cpp:572-590, 735-753, 544-561) with ONE kernel that reads every
// coefficient once and writes every pixel once.
//
// Work decomposition (wave64, no MFMA -- this is 8-point butterflies, not a GEMM):
//   * a workgroup owns one TILE = a run of consecutive MCUs (NB = hs*vs + 2 coded blocks per
//     MCU), sized so that every wave holds blocks of ONE kind (all luma or all chroma):
//       4:4:4  192 lanes = 64 MCUs (wave = Y | Cb | Cr)       24 KiB of coefficients, 24 KiB LDS
//       4:2:0  192 lanes = 32 MCUs (Y | Y | Cb+Cr)            24 KiB, 24 KiB
//       4:2:2, 4:4:0  256 lanes = 64 MCUs (Y | Y | Cb | Cr)   32 KiB, 32 KiB
//     always contiguous int16 coefficients in.
//     Two tilings, chosen by the host (jb_api.cpp) and compiled as separate instantiations:
//       row-bound (LINEAR=false): tiles never cross an MCU row; the last tile of a row is
//         partly empty unless mcus_x is a multiple of the tile length (4096/8192 px are);
//       linear (LINEAR=true): tiles cut the image's MCU stream every tile length regardless of
//         rows, so only the last tile of an image can be short; a colour segment may then
//         straddle one row end and is stored in two parts (1920 px: +4.2 %).
//     blockIdx -> tile is the identity (one compact advancing write window; XCD bands were slower).
//   * stage 1+2, one lane = one coded 8x8 block.  Lanes take the tile's blocks sorted by
//     component, so the quantisation table is wave-uniform (scalar loads -> SGPRs) wherever
//     possible (4:4:4: wave w = component w).  The lane's 128 coefficient bytes reach 32 VGPRs
//     either straight from HBM (8 global_load_dwordx4 per lane; measured 6.4-6.5 TB/s although
//     every lane of an instruction touches a different line, tools/probe_load.hip) or, in 4:4:4
//     where it measured 4 % faster, through LDS-DMA (global_load_lds_dwordx4) into the wave's own
//     8 KiB of LDS with an XOR swizzle that makes the 128-B-strided ds_read_b128 conflict free.
//     Then dequantise and 16 1-D AAN passes run entirely in that lane's registers with static
//     indexing (no cross-lane traffic, no redundant arithmetic).
//   * stage 3, twice (rows 0-3 / rows 4-7 of every block): the lanes write the integer-valued
//     f32 samples of that half into planar strips in LDS (Y, Cb, Cr: 24 KiB; the other half waits
//     in registers), XOR-swizzled per 16-B chunk so ds_write_b128 is conflict free; then one lane =
//     4 horizontally adjacent pixels: ds_read_b128 of Y, the (replicated) chroma samples, the
//     colour transform, a 12-byte pack under round-toward-zero mode, and ONE
//     buffer_store_dwordx3 whose descriptor range check drops lanes past the image edge.
//     Consecutive lanes are consecutive in the image row: a wave-instruction writes 768
//     contiguous bytes (six whole 128-B lines).  All per-iteration addressing is scalar.
//     Two phases keep a workgroup at 24 KiB of LDS = 6 workgroups (18 waves) per CU (32 KiB =
//     5 workgroups of 4 waves for the 256-lane tiles); the 3
//     workgroup barriers per tile order LDS traffic only (s_waitcnt lgkmcnt(0); s_barrier).
//
// Arithmetic is bit-exact with the reference: int32 dequantise (v_mul_i32_i24), the AAN graph
// of jpeg.cpp:598-662 evaluated in the same order with separate IEEE mul/add (this TU is built
// with -ffp-contract=off), truncation toward zero after each 1-D pass (v_trunc_f32: equal to the
// reference's float->int->float round trip for |x| < 2^31), colour per jpeg.cpp:521-535.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "jb_kernels.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;
typedef __attribute__((address_space(4))) int32_t q_const_t;  // scalar-loadable (constant) memory

__device__ __forceinline__ float kf(uint32_t bits) { return __builtin_bit_cast(float, bits); }

// f32 constants of reference include/types.hpp:5-19 (m*, s*) and jpeg.cpp:521-523, by bit pattern
#define JB_M1 kf(0x3FB504F3u)
#define JB_M2 kf(0x3F8A8BD4u)
#define JB_M3 kf(0x3FB504F3u)
#define JB_M4 kf(0x40273D74u)
#define JB_M5 kf(0x3F43EF15u)
#define JB_S0 kf(0x3EB504F3u)
#define JB_S1 kf(0x3EFB14BEu)
#define JB_S2 kf(0x3EEC835Eu)
#define JB_S3 kf(0x3ED4DB31u)
#define JB_S4 kf(0x3EB504F3u)
#define JB_S5 kf(0x3E8E39DAu)
#define JB_S6 kf(0x3E43EF15u)
#define JB_S7 kf(0x3DC7C5C2u)
#define JB_CR_R kf(0x3FB374BCu)
#define JB_CB_G kf(0x3EB020C5u)
#define JB_CR_G kf(0x3F36C8B4u)
#define JB_CB_B kf(0x3FE2D0E5u)

// Cache-policy bits (gfx940+: 1 = sc0, 2 = nt, 16 = sc1).  Every byte is touched once, so the
// LDS-DMA loads and the pixel stores are non-temporal: measured +4.8 % on the 4:4:4 stream
// (220 -> 210 us), neutral to +3 % for the stores of the other layouts.  The per-lane block
// loads of the direct path must NOT be nt: their 8 instructions re-use each 128-B line through
// L1, and nt made them 1.9x slower (154 -> 293 us on 4:2:0).
// (JB_LAB: tools/build_variant.sh builds measurement variants of this file -- other cache policies, stages
// skipped at run time -- next to the product; the product is always built without it)
#if !defined(JB_LAB) || !defined(JB_LOAD_AUX)
#undef JB_LOAD_AUX
#define JB_LOAD_AUX 2
#endif
#if !defined(JB_LAB) || !defined(JB_STORE_AUX)
#undef JB_STORE_AUX
#define JB_STORE_AUX 2
#endif
#define JB_SCHED_FENCE() ((void)0)
// Coded blocks per tile = lanes per workgroup: the smallest whole number of MCUs that fills whole
// waves with ONE component each.  4:4:4 (3 blocks per MCU) and 4:2:0 (6): 192 lanes = 64 / 32 MCUs
// (in 4:2:0 Cb and Cr share the third wave).  4:2:2 and 4:4:0 (4 blocks per MCU): 256 lanes = 64
// MCUs = two luma waves, a Cb wave, a Cr wave; with 192 lanes their waves mixed components, which
// cost a per-lane table select, per-lane strip addressing and a wave per SIMD.
constexpr int tile_blocks(int hs, int vs) { return hs * vs == 2 ? 256 : 192; }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() would also wait for the
// global stores of the previous colour phase (vmcnt(0)), putting HBM write latency on the
// critical path between the two phases.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// One 1-D pass of the AAN network (reference jpeg.cpp:598-662 / 666-730), in place, each
// output truncated toward zero exactly where the reference stores a float into an int.
__device__ __forceinline__ void aan_1d_io(float x0, float x1, float x2, float x3, float x4, float x5,
                                          float x6, float x7, float &y0, float &y1, float &y2,
                                          float &y3, float &y4, float &y5, float &y6, float &y7) {
  const float g0 = x0 * JB_S0;
  const float g1 = x4 * JB_S4;
  const float g2 = x2 * JB_S2;
  const float g3 = x6 * JB_S6;
  const float g4 = x5 * JB_S5;
  const float g5 = x1 * JB_S1;
  const float g6 = x7 * JB_S7;
  const float g7 = x3 * JB_S3;

  const float f4 = g4 - g7;
  const float f5 = g5 + g6;
  const float f6 = g5 - g6;
  const float f7 = g4 + g7;

  const float e2 = g2 - g3;
  const float e3 = g2 + g3;
  const float e5 = f5 - f7;
  const float e7 = f5 + f7;
  const float e8 = f4 + f6;

  const float d2 = e2 * JB_M1;
  const float d4 = f4 * JB_M2;
  const float d5 = e5 * JB_M3;
  const float d6 = f6 * JB_M4;
  const float d8 = e8 * JB_M5;

  const float c0 = g0 + g1;
  const float c1 = g0 - g1;
  const float c2 = d2 - e3;
  const float c4 = d4 + d8;
  const float c5 = d5 + e7;
  const float c6 = d6 - d8;
  const float c8 = c5 - c6;

  const float b0 = c0 + e3;
  const float b1 = c1 + c2;
  const float b2 = c1 - c2;
  const float b3 = c0 - e3;
  const float b4 = c4 - c8;
  const float b6 = c6 - e7;

  y0 = __builtin_truncf(b0 + e7);
  y1 = __builtin_truncf(b1 + b6);
  y2 = __builtin_truncf(b2 + c8);
  y3 = __builtin_truncf(b3 + b4);
  y4 = __builtin_truncf(b3 - b4);
  y5 = __builtin_truncf(b2 - c8);
  y6 = __builtin_truncf(b1 - b6);
  y7 = __builtin_truncf(b0 - e7);
}

// in place
__device__ __forceinline__ void aan_1d(float &x0, float &x1, float &x2, float &x3, float &x4,
                                       float &x5, float &x6, float &x7) {
  aan_1d_io(x0, x1, x2, x3, x4, x5, x6, x7, x0, x1, x2, x3, x4, x5, x6, x7);
}

__device__ __forceinline__ uint32_t pack_u8(float x, uint32_t byte, uint32_t old) {
  // reference jpeg.cpp:521-535: truncate toward zero, then clamp to 0..255.  v_cvt_pk_u8_f32
  // saturates to 0..255 by itself but rounds to nearest (probed on gfx950, tools/probe_cvt.hip),
  // so the truncation is explicit and the clamp is the instruction's own.
  return __builtin_amdgcn_cvt_pk_u8_f32(__builtin_truncf(x), byte, old);
}

// 12 colour values (4 pixels x RGB, floats before truncation) -> 12 packed bytes.  The wave's
// f32 rounding mode is switched to round-toward-zero around the twelve v_cvt_pk_u8_f32, which then
// truncate AND saturate in one half-rate instruction each (bit-identical to truncate + clamp of
// reference jpeg.cpp:521-535 on all inputs: probed on gfx950, tools/probe_cvt.hip); the mode is
// back to round-to-nearest-even before any other float instruction of this wave can issue.
__device__ __forceinline__ void pack12_rtz(const float (&r)[4], const float (&g)[4], const float (&b)[4],
                                           uint32_t &w0, uint32_t &w1, uint32_t &w2) {
  asm volatile(
      "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
      "v_cvt_pk_u8_f32 %0, %3, 0, 0\n\t"
      "v_cvt_pk_u8_f32 %1, %8, 0, 0\n\t"
      "v_cvt_pk_u8_f32 %2, %13, 0, 0\n\t"
      "v_cvt_pk_u8_f32 %0, %4, 1, %0\n\t"
      "v_cvt_pk_u8_f32 %1, %9, 1, %1\n\t"
      "v_cvt_pk_u8_f32 %2, %14, 1, %2\n\t"
      "v_cvt_pk_u8_f32 %0, %5, 2, %0\n\t"
      "v_cvt_pk_u8_f32 %1, %10, 2, %1\n\t"
      "v_cvt_pk_u8_f32 %2, %11, 2, %2\n\t"
      "v_cvt_pk_u8_f32 %0, %6, 3, %0\n\t"
      "v_cvt_pk_u8_f32 %1, %7, 3, %1\n\t"
      "v_cvt_pk_u8_f32 %2, %12, 3, %2\n\t"
      "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
      : "=&v"(w0), "=&v"(w1), "=&v"(w2)
      // byte order: w0 = r0 g0 b0 r1 | w1 = g1 b1 r2 g2 | w2 = b2 r3 g3 b3
      : "v"(r[0]), "v"(g[0]), "v"(b[0]), "v"(r[1]),   // %3..%6   -> w0 bytes 0..3
        "v"(g[2]), "v"(g[1]), "v"(b[1]), "v"(r[2]),   // %7 (w1 byte 3), %8..%10 -> w1 bytes 0..2
        "v"(g[3]), "v"(b[3]), "v"(b[2]), "v"(r[3]));  // %11 (w2 byte 2), %12 (w2 byte 3), %13, %14 -> w2 bytes 0,1
}

typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3_t __attribute__((ext_vector_type(3)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef u32x4_t u32x4_u __attribute__((aligned(1)));  // 16 bytes at any byte address


// Measurement variants (tools/ablate.sh through tools/build_variant.sh -DJB_LAB -DJB_EXP_NO_...): a stage is
// skipped at run time through a condition the compiler cannot fold, so the code and its registers stay.  The
// product is built without JB_LAB: every stage always runs.
#if defined(JB_LAB) && defined(JB_EXP_NO_LOAD)
#define JB_DO_LOAD(p) ((p).reserved == 777)
#else
#define JB_DO_LOAD(p) true
#endif
#if defined(JB_LAB) && defined(JB_EXP_NO_IDCT)
#define JB_DO_IDCT(p) ((p).reserved == 777)
#else
#define JB_DO_IDCT(p) true
#endif
#if defined(JB_LAB) && defined(JB_EXP_NO_COLOUR)
#define JB_DO_COLOUR(p) ((p).reserved == 777)
#else
#define JB_DO_COLOUR(p) true
#endif
#if defined(JB_LAB) && defined(JB_EXP_NO_STORE)
#define JB_DO_STORE(p) ((p).reserved == 777)
#else
#define JB_DO_STORE(p) true
#endif

}  // namespace

// Sorted-lane mapping: lanes take the tile's blocks sorted by component (all Y, then Cb, then
// Cr) so that a wave holds at most two components.  4:4:4: wave w = component w (its
// quantisation table is wave-uniform and lives in SGPRs); 4:2:0: waves 0,1 luma, wave 2 half Cb,
// half Cr.  s = sorted index 0..191.
template <int HS, int VS>
struct LaneMap {
  static constexpr int TB = tile_blocks(HS, VS);
  static constexpr int NY = HS * VS, NB = NY + 2, MCUS = TB / NB, NYT = NY * MCUS;
  __device__ static __forceinline__ int comp(int s) { return s < NYT ? 0 : (s < NYT + MCUS ? 1 : 2); }
  __device__ static __forceinline__ int mcu(int s) {
    const int c = comp(s);
    return c == 0 ? s / NY : s - NYT - (c - 1) * MCUS;
  }
  __device__ static __forceinline__ int slot(int s) { return comp(s) == 0 ? s % NY : 0; }  // luma block in MCU
  __device__ static __forceinline__ int block(int s) {                                      // decode order
    const int c = comp(s);
    return mcu(s) * NB + (c == 0 ? slot(s) : NY + c - 1);
  }
};

// One workgroup (192 or 256 lanes) per tile.  See the file header for the three stages.
// MIXQ = false: every wave dequantises with ONE table (4:4:4 always; 4:2:0 when Cb and Cr name the
// same table, the usual case).  MIXQ = true: a wave may hold blocks of two components with
// different tables and selects per lane; kept out of the MIXQ = false instantiation because its
// register pressure would cost the common case a wave per SIMD.
// STAGED = true (linear tiling only): the colour stage packs the pixels back into LDS and the same wave then writes
// the strip row as whole 64-byte lines plus byte runs at its two ends, instead of one 12-byte store per lane at
// whatever alignment the row has (see "staged" below).  A MEASURED VARIANT, not a product path: it is only
// instantiated in -DJB_LAB builds (tools/build_variant.sh; JPEGBLK_STAGED_STORE=1 selects it there).  On the sizes
// it was meant for it is 1-3 % slower than the stores it replaces, and 4-10 % slower on aligned rows
// (profiles/r03/probe_staged.json): the aligned lines are worth about 5 %, the second trip
// through LDS and the serial pack-then-copy of a row per wave cost more.
template <int HS, int VS, bool MIXQ, bool LINEAR, bool STAGED = false>
// (5 waves/SIMD are asked for where that costs no spill: 4:4:4 and 4:4:0; forcing it on 4:2:0 or
// 4:2:2 spills and measured 9 % slower)
__global__ __launch_bounds__(tile_blocks(HS, VS), (HS == 1) ? 5 : 1) void jb_tile_kernel(const JbLaunch p) {
  static_assert(!STAGED || LINEAR, "the staged store stage is an instantiation of the linear tiling");
  using LM = LaneMap<HS, VS>;
  constexpr int kTileBlocks = LM::TB;
  constexpr int kStripBytes = kTileBlocks * 128;  // half of the tile's f32 samples: 24 or 32 KiB
  constexpr int NB = LM::NB, MCUS = LM::MCUS, NYT = LM::NYT;
  constexpr int YW = MCUS * 8 * HS;             // luma strip width in pixels
  constexpr int CW = MCUS * 8;                  // chroma strip width in samples
  constexpr int YROWS = 4 * VS;                 // luma rows per phase (half of the tile's rows)
  constexpr int CB_OFF = YROWS * YW * 4;        // byte offsets of the strips in LDS
  constexpr int CR_OFF = CB_OFF + 4 * CW * 4;
  static_assert(CR_OFF + 4 * CW * 4 == kStripBytes, "strips must fill the strip area exactly");
  static_assert(NYT % 64 == 0, "whole luma waves: no wave mixes luma and chroma blocks");
  constexpr bool kPermChroma = (VS == 2) && (NYT % 64 == 0);  // 4:2:0: chroma blocks fill a whole wave
  // The 4-pixel group that straddles the right image edge (width % 4 != 0): two stores from the packed
  // words in the linear tiling -- the one small and odd-sized images take, where every row has such a
  // group -- and the byte-store loop elsewhere: in the row-bound instantiations (4096 / 8192-pixel
  // rows, where the case needs a width like 4093) and in 4:4:0 the extra code cost a wave per SIMD
  // or spilled (hipcc's allocation of this kernel is at the edge: 94-96 of 96 VGPRs).
  constexpr bool kTwoStoreTail = LINEAR && !(HS == 1 && VS == 2);
  constexpr bool kDirectLoad = !((NYT % 64 == 0) && (MCUS % 64 == 0) && (CB_OFF == 8192));  // all but 4:4:4
  __shared__ __attribute__((aligned(1024))) char lds[kStripBytes];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- which tile (all wave-uniform) ----
  const int tiles_per_image = p.tiles_per_image;
  // blockIdx -> tile is the identity: workgroups are dealt round-robin over the 8 XCDs, so at any
  // moment the resident workgroups of ALL XCDs write one compact, advancing window of the output.
  // The "XCD-aware" alternative (every XCD owns one contiguous band of tiles, so that its L2 holds
  // whole rows) was measured and is slower: neutral on 4:4:4, -3.5 % on 4:2:0 (write-heavy), -1 %
  // elsewhere -- nothing is re-read, so there is no L2 locality to win, and eight separate write
  // windows cost HBM page locality (tools/probe_store2.hip: the same 403 MB written by one
  // advancing window reach 6.4 TB/s, by many separate streams 5.2 TB/s).
  const int tile = blockIdx.x;
  const int img = tile / tiles_per_image;
  const int rem = tile - img * tiles_per_image;
  // Two tilings.  Linear (p.linear, the default): a tile is 192/NB consecutive MCUs of the image's
  // MCU stream, whatever MCU rows they fall in -- every tile but the image's last is full for any
  // image width.  Row-bound: a tile is a run of MCUs of ONE MCU row (the last run of a row may be
  // short); used where it leaves no tile ragged, and for very narrow images.
  int my, mx0, nvalid;
  if (LINEAR) {
    const int m0 = rem * MCUS;
    my = m0 / p.mcus_x;
    mx0 = m0 - my * p.mcus_x;
    nvalid = min(MCUS, p.mcus_x * p.mcus_y - m0);
  } else {
    my = rem / p.tiles_per_row;
    mx0 = (rem - my * p.tiles_per_row) * MCUS;
    nvalid = min(MCUS, p.mcus_x - mx0);
  }
  const int last_block = nvalid * NB - 1;
  const uint8_t *tile_coef = (const uint8_t *)p.coef + (int64_t)img * p.coef_image_stride +
                             ((int64_t)my * p.mcus_x + mx0) * (NB * 128);
  const q_const_t *qsrc = (const q_const_t *)((const uint8_t *)p.qtabs + (int64_t)img * p.qtab_image_stride);

  // ---- stage 2: this lane's block -> registers ----
  const int comp_a = LM::comp(wave * 64);       // first lane's component (wave-uniform)
  const int comp_b = LM::comp(wave * 64 + 63);  // last lane's component
  const q_const_t *qa = qsrc + comp_a * 64;
  const q_const_t *qb = qsrc + comp_b * 64;
  float v[64];
  {
    // ---- stage 1: this lane's block (128 contiguous bytes) -> 32 VGPRs.  Default: straight from
    // HBM with 8 dwordx4 loads.  Every lane of a wave-instruction touches a different 128-B line,
    // but the 8 instructions of the wave consume those 64 lines completely: measured 6.4-6.5 TB/s
    // on MI355X (tools/probe_load.hip), the same as perfectly coalesced loads.
    uint32_t raw[32];  // row k = dwords 4k..4k+3, two int16 (columns 2j, 2j+1) per dword
    if (kDirectLoad) {
      const int n = min(LM::block(tid), last_block);  // ragged tile: re-read its last block
      const u32x4_t *src = (const u32x4_t *)(tile_coef + (uint32_t)n * 128u);
#pragma unroll
      for (int j = 0; j < 8; j++) {
        u32x4_t t = {0, 0, 0, 0};
        if (JB_DO_LOAD(p)) t = src[j];  // default cache policy: the line is re-used by the next 7 loads
        raw[j * 4 + 0] = t.x;
        raw[j * 4 + 1] = t.y;
        raw[j * 4 + 2] = t.z;
        raw[j * 4 + 3] = t.w;
      }
    } else {
      // 4:4:4 only (measured 4 % faster there than the direct loads): LDS-DMA
      // (global_load_lds_dwordx4: no VGPRs, 1 KiB per wave-instruction, every 8 lanes fetch one
      // whole 128-B line).  Wave w = component w fetches exactly the 64 blocks its own lanes
      // consume into its own 8 KiB of LDS -- the same 8 KiB its strip occupies later -- so neither
      // the load nor the hand-over to the strips needs a workgroup barrier.  Chunk j of the block
      // of lane l lands at chunk position j ^ ((l>>1)&7) (applied to the per-lane GLOBAL address,
      // LDS-DMA writes LDS linearly), which makes the 128-B-strided ds_read_b128 conflict free.
      char *const wave_lds = lds + wave * 8192;
      if (JB_DO_LOAD(p))
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const int l2 = i * 8 + (lane >> 3);  // the lane whose block this chunk belongs to
          const int f2 = (l2 >> 1) & 7;
          const int nsrc = min(LM::block(wave * 64 + l2), last_block);
          const uint32_t off = (uint32_t)nsrc * 128u + (uint32_t)(((lane & 7) ^ f2) << 4);
          __builtin_amdgcn_global_load_lds((gbl_void_t *)(tile_coef + off), (lds_void_t *)(wave_lds + i * 1024), 16, 0, JB_LOAD_AUX);
        }
      const int f = (lane >> 1) & 7;
      const char *base = wave_lds + lane * 128;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const uint4 t = *(const uint4 *)(base + ((j ^ f) << 4));
        raw[j * 4 + 0] = t.x;
        raw[j * 4 + 1] = t.y;
        raw[j * 4 + 2] = t.z;
        raw[j * 4 + 3] = t.w;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }

    // dequantise (jpeg.cpp:563-569): int32 product, int->float on first use (jpeg.cpp:598)
    constexpr bool kUniformWaves = (NYT % 64 == 0) && (MCUS % 64 == 0);  // 4:4:4: one component per wave
    // p.chroma_q_equal: Cb and Cr name the same table (the usual case), so a wave that mixes
    // Cb and Cr blocks is still uniform as far as dequantisation goes
    if (!MIXQ || kUniformWaves || comp_a == comp_b || (p.chroma_q_equal && comp_a != 0)) {
#pragma unroll
      for (int k = 0; k < 8; k++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const uint32_t w = raw[k * 4 + (i >> 1)];
          const int c = (i & 1) ? ((int)w >> 16) : (int)(short)(w & 0xffffu);
          v[k * 8 + i] = (float)__mul24(c, qa[k * 8 + i]);
        }
      }
    } else {
      const bool second = LM::comp(tid) == comp_b;
#pragma unroll
      for (int k = 0; k < 8; k++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const uint32_t w = raw[k * 4 + (i >> 1)];
          const int c = (i & 1) ? ((int)w >> 16) : (int)(short)(w & 0xffffu);
          // both entries through scalar loads, then a per-lane select (v_cndmask)
          const int q0 = __builtin_amdgcn_readfirstlane(qa[k * 8 + i]);
          const int q1 = __builtin_amdgcn_readfirstlane(qb[k * 8 + i]);
          v[k * 8 + i] = (float)__mul24(c, second ? q1 : q0);
        }
      }
    }
  }
  if (JB_DO_IDCT(p)) {
  // 4:2:0: a chroma block is needed as rows {0,1,4,5} in the first colour phase and {2,3,6,7} in
  // the second (chroma row r covers luma rows 2r, 2r+1), a luma block as rows 0-3 / 4-7.  The
  // chroma wave therefore stores column-pass output row k in register row sigma(k), sigma =
  // (0 1 4 5 2 3 6 7), so that for every lane "register rows 0-3" is what phase 0 consumes and
  // only 32 registers have to wait for phase 1.  (Row passes do not care which row they hold.)
  if (kPermChroma && comp_a != 0) {
#pragma unroll
    for (int i = 0; i < 8; i++)  // column pass, jpeg.cpp:596-663, outputs permuted by sigma
      aan_1d_io(v[0 * 8 + i], v[1 * 8 + i], v[2 * 8 + i], v[3 * 8 + i], v[4 * 8 + i], v[5 * 8 + i],
                v[6 * 8 + i], v[7 * 8 + i],  //
                v[0 * 8 + i], v[1 * 8 + i], v[4 * 8 + i], v[5 * 8 + i], v[2 * 8 + i], v[3 * 8 + i],
                v[6 * 8 + i], v[7 * 8 + i]), JB_SCHED_FENCE();
  } else {
#pragma unroll
    for (int i = 0; i < 8; i++)  // column pass, jpeg.cpp:596-663
      aan_1d(v[0 * 8 + i], v[1 * 8 + i], v[2 * 8 + i], v[3 * 8 + i], v[4 * 8 + i], v[5 * 8 + i],
             v[6 * 8 + i], v[7 * 8 + i]), JB_SCHED_FENCE();
  }
#pragma unroll
  for (int k = 0; k < 8; k++)  // row pass, jpeg.cpp:664-731
    aan_1d(v[k * 8 + 0], v[k * 8 + 1], v[k * 8 + 2], v[k * 8 + 3], v[k * 8 + 4], v[k * 8 + 5],
           v[k * 8 + 6], v[k * 8 + 7]), JB_SCHED_FENCE();
  }

  // ---- stage 3: two phases (upper / lower half of the tile's pixel rows) ----
  // Where this lane's block lands in the strips.  VS == 1: every block contributes rows
  // 4*phase..4*phase+3.  VS == 2: luma blocks of block-row bv contribute all 8 rows in phase
  // bv; chroma blocks contribute rows 4*phase..4*phase+3 (chroma row r covers luma rows 2r, 2r+1).
  // A strip row is a sequence of 16-B chunks (4 samples); chunk c is stored at position
  // c ^ ((c>>3)&1) so that the 8 lanes of a ds_write_b128 group hit 8 different bank quads.
  // (everything per-lane here is derived from an opaque copy of the thread id, so that it is
  // materialised after the IDCT instead of occupying registers across it)
  int tid_late = tid;
  asm volatile("" : "+v"(tid_late));
  const int comp_l = LM::comp(tid_late), mcu_l = LM::mcu(tid_late), slot_l = LM::slot(tid_late);
  const int bv = comp_l == 0 ? slot_l / HS : 0;
  const int bh = comp_l == 0 ? slot_l - bv * HS : 0;
  const int pitch = comp_l == 0 ? (YW * 4) : CW * 4;
  const int blk_col = comp_l == 0 ? mcu_l * HS + bh : mcu_l;  // 8-sample column of the block in its strip
  const int sw = (blk_col >> 2) & 1;
  const int luma_off = bv * 4 * (YW * 4) + blk_col * 32;
  char *const dst = lds + (comp_l == 0 ? luma_off : (comp_l == 1 ? CB_OFF : CR_OFF) + blk_col * 32);
  char *const dst_lo = dst + sw * 16;        // samples 0..3 of a row
  char *const dst_hi = dst + (sw ^ 1) * 16;  // samples 4..7

  constexpr int TASKS_PER_ROW = YW / 4;
  constexpr int TASKS = YROWS * TASKS_PER_ROW;
  static_assert(TASKS % 64 == 0, "whole wave-iterations");
  static_assert(TASKS_PER_ROW % 64 == 0, "a wave-iteration stays within one strip row");
  uint8_t *const img_rgb = p.rgb + (int64_t)img * p.rgb_image_stride;
  // loop-invariant lane offsets of the colour stage (row-uniform layouts): the lane's 16-B luma
  // chunk within a 64-chunk segment and its chroma chunk, both with the strip swizzle applied
  // (the swizzle bit is bit 3 of the chunk index, which the segment number does not touch)
  // (computed from an opaque copy of the lane id so they are materialised here, after the IDCT,
  // instead of being kept in registers across it)
  const int lane_late = tid_late & 63;
  const int lane_y_off = (lane_late ^ ((lane_late >> 3) & 1)) * 16;
  const int lane_c_off = HS == 1 ? lane_y_off : (((lane_late >> 1) ^ ((lane_late >> 4) & 1)) * 16 + (lane_late & 1) * 8);

#pragma unroll
  for (int phase = 0; phase < 2; phase++) {
    if (phase == 1) lds_barrier();  // phase-0 colour reads are done: the strips may be rewritten
    // Every lane contributes rows 4*phase..4*phase+3 of its block (so half of every block is
    // consumed per phase and only 32 values wait in registers).  Luma block-row bv lands in
    // strip rows 4*bv..4*bv+3 (VS == 2: the strip holds image rows 4p..4p+3 and 8+4p..8+4p+3);
    // a chroma block contributes the chroma rows those luma rows need: row/VS of each, i.e.
    // rows 2p, 2p+1, 4+2p, 5+2p for VS == 2 (reference jpeg.cpp:518-520).
    static_assert(VS == 1 || kPermChroma, "luma and chroma lanes take the same register rows per phase");
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
      const int k = phase * 4 + kk;
      *(float4 *)(dst_lo + kk * pitch) = make_float4(v[k * 8 + 0], v[k * 8 + 1], v[k * 8 + 2], v[k * 8 + 3]);
      *(float4 *)(dst_hi + kk * pitch) = make_float4(v[k * 8 + 4], v[k * 8 + 5], v[k * 8 + 6], v[k * 8 + 7]);
    }
    lds_barrier();

    if constexpr (STAGED) {
      // ---- staged: pack into LDS, then write line-aligned ----
      // Pass A.  One wave takes a whole strip row, segment after segment: the 12 packed bytes of a lane go back
      // into the row's own luma strip at segment * 768 + lane * 12 -- at or below every byte the wave has still to
      // read, and no other wave touches this row -- so the row ends up as npixels * 3 contiguous output bytes.
      constexpr int IPR = TASKS_PER_ROW / 64;
      for (int row = wave; row < YROWS; row += kTileBlocks / 64) {
        char *const rowp = lds + row * (YW * 4);
#pragma unroll
        for (int seg = 0; seg < IPR; seg++) {
          const float4 Y = *(const float4 *)(rowp + lane_y_off + seg * 1024);
          float cb[4], cr[4];
          const int coff = (row / VS) * (CW * 4) + seg * (1024 / HS);
          if (HS == 1) {
            const float4 a = *(const float4 *)(lds + CB_OFF + lane_c_off + coff);
            const float4 b = *(const float4 *)(lds + CR_OFF + lane_c_off + coff);
            cb[0] = a.x, cb[1] = a.y, cb[2] = a.z, cb[3] = a.w;
            cr[0] = b.x, cr[1] = b.y, cr[2] = b.z, cr[3] = b.w;
          } else {
            const float2 a = *(const float2 *)(lds + CB_OFF + lane_c_off + coff);
            const float2 b = *(const float2 *)(lds + CR_OFF + lane_c_off + coff);
            cb[0] = cb[1] = a.x, cb[2] = cb[3] = a.y;
            cr[0] = cr[1] = b.x, cr[2] = cr[3] = b.y;
          }
          const float yy[4] = {Y.x, Y.y, Y.z, Y.w};
          float r[4], g[4], b[4];
#pragma unroll
          for (int i = 0; i < 4; i++) {
            r[i] = (yy[i] + JB_CR_R * cr[i]) + 128.0f;
            g[i] = ((yy[i] - JB_CB_G * cb[i]) - JB_CR_G * cr[i]) + 128.0f;
            b[i] = (yy[i] + JB_CB_B * cb[i]) + 128.0f;
          }
          uint32_t w0, w1, w2;
          pack12_rtz(r, g, b, w0, w1, w2);
          asm volatile("" ::: "memory");  // (the reads of this segment stay in front of its writes, the writes in front of the next reads)
          uint32_t *const o = (uint32_t *)(rowp + seg * 768 + lane_late * 12);
          o[0] = w0, o[1] = w1, o[2] = w2;
          asm volatile("" ::: "memory");
        }
        // Pass B, by the same wave on the row it has just packed (so nothing but its own LDS traffic has to be waited
        // for).  A strip row is the pixels of the tile's MCUs, which lie in one image row per MCU row the tile
        // touches: each such piece is written as [bytes up to the first 64-byte line][whole 16-byte chunks from
        // there][the last bytes], the chunks fetched from LDS at whatever byte offset that takes.  Whole lines
        // wherever the piece covers them, whatever the row stride.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int y_in = phase * 4 + (row >> 2) * 8 + (row & 3);
        int m = 0, my_k = my, mx_k = mx0;
        while (m < nvalid) {
          const int cnt = min(nvalid - m, p.mcus_x - mx_k);
          const int y = my_k * 8 * VS + y_in, x0 = mx_k * 8 * HS;
          const int len = min(cnt * 8 * HS, p.width - x0) * 3;   // bytes of the piece inside the image
          if (y < p.height && len > 0) {
            uint8_t *const dstp = img_rgb + (int64_t)y * p.rgb_row_stride + (int64_t)x0 * 3;
            const int src = row * (YW * 4) + m * (8 * HS * 3);     // LDS byte offset of the piece (a multiple of 8)
            const int head = min(len, (int)((64 - ((uintptr_t)dstp & 63)) & 63));
            const int nbody = (len - head) >> 4;
            const int tail = len - head - nbody * 16;
            if (lane_late < head) dstp[lane_late] = (uint8_t)lds[src + lane_late];
            if (lane_late < tail) dstp[len - tail + lane_late] = (uint8_t)lds[src + len - tail + lane_late];
            // (gfx950 reads 16 bytes of LDS at any byte address in one ds_read_b128: unaligned DS access is on under ROCm)
            const char *const from = lds + src + head;
            u32x4_t *const lines = (u32x4_t *)(dstp + head);
            for (int c = lane_late; c < nbody; c += 64) {
              const u32x4_t v4 = *(const u32x4_u *)(from + c * 16);
              __builtin_nontemporal_store(v4, lines + c);
            }
          }
          m += cnt;
          mx_k = 0;
          my_k++;
        }
      }
    } else
    // colour transform + store: one lane = 4 adjacent pixels of one row, one wave-iteration =
    // 256 adjacent pixels (768 contiguous output bytes)
    // image row of strip row sr: block-row sr/4, row 4*phase + sr%4 within the block
    for (int it = wave; it < TASKS / 64; it += kTileBlocks / 64) {
      {
        // Everything but the data is wave-uniform here: the row, the 256-pixel segment of the row,
        // the output address (a buffer descriptor per segment; lanes address it with the loop-
        // invariant offset lane*12 and the hardware range check drops lanes past the image edge).
        constexpr int IPR = TASKS_PER_ROW / 64;  // wave-iterations per strip row
        const int row = it / IPR, seg = it - row * IPR;
        // where the segment's 256 pixels (SEG_MCUS MCUs) go: MCU row my_s from MCU column mx_s on;
        // in the linear tiling the segment may run past the end of the MCU row and continue at
        // the start of the next one (at most once: the host only selects the linear tiling when
        // an MCU row holds at least one whole segment)
        constexpr int SEG_MCUS = 256 / (8 * HS);
        const int seg_valid = min(SEG_MCUS, nvalid - seg * SEG_MCUS);  // MCUs of the segment that exist
        if (seg_valid <= 0) continue;
        int my_s = my, mx_s = mx0 + seg * SEG_MCUS;
        if (LINEAR)
          while (mx_s >= p.mcus_x) {
            mx_s -= p.mcus_x;
            my_s++;
          }
        const int n_row = min(seg_valid, p.mcus_x - mx_s);  // MCUs before the wrap
        const int y_in = phase * 4 + (row >> 2) * 8 + (row & 3);  // pixel row within the MCU row
        if (!LINEAR && (my * 8 * VS + y_in >= p.height || mx_s * 8 * HS >= p.width)) continue;
        const float4 Y = *(const float4 *)(lds + lane_y_off + row * (YW * 4) + seg * 1024);
        float cb[4], cr[4];
        // chroma sample of luma pixel (row, col): (row/VS, col/HS) -- reference jpeg.cpp:518-520
        const int coff = (row / VS) * (CW * 4) + seg * (1024 / HS);
        if (HS == 1) {
          const float4 a = *(const float4 *)(lds + CB_OFF + lane_c_off + coff);
          const float4 b = *(const float4 *)(lds + CR_OFF + lane_c_off + coff);
          cb[0] = a.x, cb[1] = a.y, cb[2] = a.z, cb[3] = a.w;
          cr[0] = b.x, cr[1] = b.y, cr[2] = b.z, cr[3] = b.w;
        } else {
          const float2 a = *(const float2 *)(lds + CB_OFF + lane_c_off + coff);
          const float2 b = *(const float2 *)(lds + CR_OFF + lane_c_off + coff);
          cb[0] = cb[1] = a.x, cb[2] = cb[3] = a.y;
          cr[0] = cr[1] = b.x, cr[2] = cr[3] = b.y;
        }
        const float yy[4] = {Y.x, Y.y, Y.z, Y.w};
        float r[4], g[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          if (JB_DO_COLOUR(p)) {
            r[i] = (yy[i] + JB_CR_R * cr[i]) + 128.0f;
            g[i] = ((yy[i] - JB_CB_G * cb[i]) - JB_CR_G * cr[i]) + 128.0f;
            b[i] = (yy[i] + JB_CB_B * cb[i]) + 128.0f;
          } else {
            r[i] = yy[i], g[i] = cb[i], b[i] = cr[i];
          }
        }
        if (!JB_DO_STORE(p)) continue;
        uint32_t w0 = 0, w1 = 0, w2 = 0;
        // part 0: the MCUs before the wrap; part 1 (linear tiling only): the rest, one MCU row down
#pragma unroll
        for (int part = 0; part < (LINEAR ? 2 : 1); part++) {
          const int px0 = part == 0 ? 0 : n_row * 8 * HS;  // first segment pixel of the part
          const int y = (my_s + part) * 8 * VS + y_in;
          const int x0 = part == 0 ? mx_s * 8 * HS : 0;     // image column of that pixel
          const int npx = min((part == 0 ? n_row : seg_valid - n_row) * 8 * HS, p.width - x0);
          if (npx > 0 && y < p.height) {
            uint8_t *const segp = img_rgb + (int64_t)y * p.rgb_row_stride + (int64_t)x0 * 3;  // address of pixel px0
            const int rel = lane_late * 4 - px0;  // this lane's first pixel relative to the part
            if (p.fast_store && (part == 0 || rel >= 0)) {
              // whole 4-pixel groups through one 12-byte store per lane; the descriptor's range
              // check drops the lanes past the part's end
              pack12_rtz(r, g, b, w0, w1, w2);  // (again for the rare second part: cheaper than keeping it live)
              const int voff = LINEAR ? rel * 3 : lane_late * 12;
              const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(segp, 0, (npx >> 2) * 12, 0x00020000);
              __builtin_amdgcn_raw_buffer_store_b96(u32x3_t{w0, w1, w2}, rsrc, voff, 0, JB_STORE_AUX);
              if (kTwoStoreTail && (npx & 3)) {
                // the one group that straddles the right edge: its lane holds the 12 packed bytes and
                // stores the 3, 6 or 9 inside the image with two stores (nine byte stores, each a
                // wave-wide instruction, cost the odd-width bundled-image size 5 points of roofline)
                // (the scalar offset carries the +2 / +4 / +8, and the lane test goes through an opaque
                // copy: nothing of this rare path is hoisted into registers that live across the IDCT)
                const __amdgpu_buffer_rsrc_t tail = __builtin_amdgcn_make_buffer_rsrc(segp, 0, 0x7ffffff0, 0x00020000);
                int tv = voff;
                asm volatile("" : "+v"(tv));
                if (tv == (npx >> 2) * 12) {
                  const int t = npx & 3;
                  if (t == 3) {
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{w0, w1}, tail, tv, 0, JB_STORE_AUX);
                    __builtin_amdgcn_raw_buffer_store_b8((uint8_t)w2, tail, tv, 8, JB_STORE_AUX);
                  } else if (t == 2) {
                    __builtin_amdgcn_raw_buffer_store_b32(w0, tail, tv, 0, JB_STORE_AUX);
                    __builtin_amdgcn_raw_buffer_store_b16((uint16_t)w1, tail, tv, 4, JB_STORE_AUX);
                  } else {
                    __builtin_amdgcn_raw_buffer_store_b16((uint16_t)w0, tail, tv, 0, JB_STORE_AUX);
                    __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(w0 >> 16), tail, tv, 2, JB_STORE_AUX);
                  }
                }
              }
            }
            if (!p.fast_store || (!kTwoStoreTail && (npx & 3))) {
              // the byte-store knob (JPEGBLK_BYTE_STORE=1): every pixel through byte stores; and the
              // straddling group of the instantiations without the two-store tail
              const int first = p.fast_store ? (npx & ~3) : 0;
              // opaque copy of a value that is live anyway: keeps this rare path's address
              // arithmetic from being hoisted out of the loop into registers
              int tail_src = LINEAR ? rel : lane_late;
              asm volatile("" : "+v"(tail_src));
#pragma unroll
              for (int i = 0; i < 4; i++) {
                const int px = (LINEAR ? tail_src : tail_src * 4) + i;
                if (px >= first && px < npx) {
                  uint8_t *o = segp + px * 3;
                  o[0] = (uint8_t)pack_u8(r[i], 0, 0);
                  o[1] = (uint8_t)pack_u8(g[i], 0, 0);
                  o[2] = (uint8_t)pack_u8(b[i], 0, 0);
                }
              }
            }
          }
          if (!LINEAR || n_row >= seg_valid) break;  // no second part
        }
      }
    }
  }
}

// ---- small grids (4:4:4): one WAVE per 16 MCUs -------------------------------------------------------------
// A single 1080p 4:4:4 image is 507 tiles of the kernel above on 256 CUs: every CU runs its two workgroups through
// load -> IDCT -> colour in step, and the launch is bounded by the length of that chain, not by traffic (DESIGN.md
// section 5.2).  This variant cuts the same work into four times as many independent pieces: a 64-lane workgroup
// owns 16 MCUs of one MCU row -- lanes 0-15 their Y blocks, 16-31 Cb, 32-47 Cr, 48-63 idle -- with the
// quantisation tables in LDS (per-lane component), 6 KiB of strips, and no barrier other workgroups wait on, so
// the phases of the eight or so waves on a CU interleave.  Arithmetic and its order are those of the kernel above.
// Selected by the host for launches of fewer than 3 workgroups per CU (jb_api.cpp; JPEGBLK_SMALL_GRID).
constexpr int kSmallMcus = 16;
__global__ __launch_bounds__(64) void jb_small_kernel_444(const JbLaunch p) {
  constexpr int kStrip = 2048 + 64;  // 4 rows x 128 samples x 4 B, skewed so the three strips start in different banks
  constexpr int kQPitch = 64 + 4;    // dwords per table in LDS (the three tables start in different banks)
  __shared__ __attribute__((aligned(16))) char lds[3 * kStrip + 3 * kQPitch * 4];
  int32_t *const qlds = (int32_t *)(lds + 3 * kStrip);
  const int lane = threadIdx.x;
  const int tile = blockIdx.x;
  const int img = tile / p.tiles_per_image;
  const int rem = tile - img * p.tiles_per_image;
  const int my = rem / p.tiles_per_row;
  const int mx0 = (rem - my * p.tiles_per_row) * kSmallMcus;
  const int nvalid = min(kSmallMcus, p.mcus_x - mx0);
  const int comp = min(lane >> 4, 2);  // (lanes 48-63 repeat the Cr lanes' work and write nothing)
  const bool active = lane < 48;
  const int m = lane & 15;
  const uint8_t *tile_coef = (const uint8_t *)p.coef + (int64_t)img * p.coef_image_stride + ((int64_t)my * p.mcus_x + mx0) * 384;
  const int32_t *qsrc = (const int32_t *)((const uint8_t *)p.qtabs + (int64_t)img * p.qtab_image_stride);
#pragma unroll
  for (int i = 0; i < 3; i++) qlds[i * kQPitch + lane] = qsrc[i * 64 + lane];
  float v[64];
  {
    uint32_t raw[32];
    const u32x4_t *src = (const u32x4_t *)(tile_coef + (uint32_t)(min(m, nvalid - 1) * 3 + comp) * 128u);
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const u32x4_t t = src[j];
      raw[j * 4 + 0] = t.x, raw[j * 4 + 1] = t.y, raw[j * 4 + 2] = t.z, raw[j * 4 + 3] = t.w;
    }
    __syncthreads();  // the tables are in LDS
    const int32_t *q = qlds + comp * kQPitch;
#pragma unroll
    for (int k = 0; k < 8; k++) {
#pragma unroll
      for (int i = 0; i < 8; i += 4) {
        const int4 q4 = *(const int4 *)(q + k * 8 + i);
        const int qq[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const uint32_t w = raw[k * 4 + ((i + e) >> 1)];
          const int c = ((i + e) & 1) ? ((int)w >> 16) : (int)(short)(w & 0xffffu);
          v[k * 8 + i + e] = (float)__mul24(c, qq[e]);  // jpeg.cpp:563-569
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; i++)  // column pass, jpeg.cpp:596-663
    aan_1d(v[0 * 8 + i], v[1 * 8 + i], v[2 * 8 + i], v[3 * 8 + i], v[4 * 8 + i], v[5 * 8 + i], v[6 * 8 + i], v[7 * 8 + i]);
#pragma unroll
  for (int k = 0; k < 8; k++)  // row pass, jpeg.cpp:664-731
    aan_1d(v[k * 8 + 0], v[k * 8 + 1], v[k * 8 + 2], v[k * 8 + 3], v[k * 8 + 4], v[k * 8 + 5], v[k * 8 + 6], v[k * 8 + 7]);

  // colour stage, rows 0-3 then rows 4-7 of the MCU row: strips of 4 rows x 128 samples per component, 16-B chunk c
  // of a row stored at c ^ ((c >> 3) & 1) (as above); then one lane = 4 adjacent pixels, a wave-iteration = 2 rows
  const int sw = (m >> 2) & 1;
  char *const dst = lds + comp * kStrip + m * 32;
  const int x4 = lane & 31;
  const int rd = (x4 ^ ((x4 >> 3) & 1)) * 16;
  uint8_t *const img_rgb = p.rgb + (int64_t)img * p.rgb_image_stride;
  const int x = mx0 * 8 + x4 * 4;                 // image column of this lane's first pixel
  const int npx = min(4, min(nvalid * 8, p.width - mx0 * 8) - x4 * 4);  // its pixels inside the image (<= 0: none)
#pragma unroll
  for (int phase = 0; phase < 2; phase++) {
    if (phase == 1) __syncthreads();
    if (active) {
#pragma unroll
      for (int kk = 0; kk < 4; kk++) {
        const int k = phase * 4 + kk;
        *(float4 *)(dst + kk * 512 + sw * 16) = make_float4(v[k * 8 + 0], v[k * 8 + 1], v[k * 8 + 2], v[k * 8 + 3]);
        *(float4 *)(dst + kk * 512 + (sw ^ 1) * 16) = make_float4(v[k * 8 + 4], v[k * 8 + 5], v[k * 8 + 6], v[k * 8 + 7]);
      }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; it++) {
      const int r = it * 2 + (lane >> 5);
      const int y = my * 8 + phase * 4 + r;
      const float4 Y = *(const float4 *)(lds + r * 512 + rd);
      const float4 B = *(const float4 *)(lds + kStrip + r * 512 + rd);
      const float4 R = *(const float4 *)(lds + 2 * kStrip + r * 512 + rd);
      const float yy[4] = {Y.x, Y.y, Y.z, Y.w}, cb[4] = {B.x, B.y, B.z, B.w}, cr[4] = {R.x, R.y, R.z, R.w};
      float rr[4], gg[4], bb[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {  // jpeg.cpp:521-535
        rr[i] = (yy[i] + JB_CR_R * cr[i]) + 128.0f;
        gg[i] = ((yy[i] - JB_CB_G * cb[i]) - JB_CR_G * cr[i]) + 128.0f;
        bb[i] = (yy[i] + JB_CB_B * cb[i]) + 128.0f;
      }
      if (p.fast_store) {
        uint32_t w0, w1, w2;
        pack12_rtz(rr, gg, bb, w0, w1, w2);
        if (y < p.height && npx == 4) {
          // a wave-uniform descriptor at the first of the iteration's two rows; the lane adds its row and column
          uint8_t *const rows = img_rgb + (int64_t)(my * 8 + phase * 4 + it * 2) * p.rgb_row_stride + (int64_t)mx0 * 24;
          const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(rows, 0, 0x7ffffff0, 0x00020000);
          __builtin_amdgcn_raw_buffer_store_b96(u32x3_t{w0, w1, w2}, rsrc, (lane >> 5) * (int)p.rgb_row_stride + x4 * 12, 0, JB_STORE_AUX);
        }
      }
      uint8_t *const o = img_rgb + (int64_t)y * p.rgb_row_stride + (int64_t)x * 3;
      if (y < p.height && npx > 0 && (!p.fast_store || npx < 4)) {
#pragma unroll
        for (int i = 0; i < 4; i++)
          if (i < npx) {
            o[i * 3 + 0] = (uint8_t)pack_u8(rr[i], 0, 0);
            o[i * 3 + 1] = (uint8_t)pack_u8(gg[i], 0, 0);
            o[i * 3 + 2] = (uint8_t)pack_u8(bb[i], 0, 0);
          }
      }
    }
  }
}

// The same for 4:2:0: a 64-lane workgroup owns 8 MCUs of one MCU row -- lanes 0-31 their luma blocks (MCU l >> 2,
// block l & 3 in decode order: top left, top right, bottom left, bottom right), 32-39 Cb, 40-47 Cr, 48-63 idle.
// Strips per phase: 8 luma rows (rows 4p..4p+3 of both block rows) x 128 samples, 4 chroma rows (2p, 2p+1, 4+2p,
// 5+2p: reference jpeg.cpp:518-520) x 64 samples per chroma component; a chroma lane picks those rows of its block
// with a select per value (the 192-lane kernel permutes them in the column pass, which a wave of mixed lanes cannot).
constexpr int kSmallMcus420 = 8;
__global__ __launch_bounds__(64) void jb_small_kernel_420(const JbLaunch p) {
  constexpr int kYStrip = 8 * 128 * 4;       // 4 KiB
  constexpr int kCStrip = 4 * 64 * 4 + 64;   // 1 KiB, skewed
  constexpr int kQPitch = 64 + 4;
  __shared__ __attribute__((aligned(16))) char lds[kYStrip + 2 * kCStrip + 3 * kQPitch * 4];
  int32_t *const qlds = (int32_t *)(lds + kYStrip + 2 * kCStrip);
  const int lane = threadIdx.x;
  const int tile = blockIdx.x;
  const int img = tile / p.tiles_per_image;
  const int rem = tile - img * p.tiles_per_image;
  const int my = rem / p.tiles_per_row;
  const int mx0 = (rem - my * p.tiles_per_row) * kSmallMcus420;
  const int nvalid = min(kSmallMcus420, p.mcus_x - mx0);
  const bool active = lane < 48;
  const int comp = lane < 32 ? 0 : (lane < 40 ? 1 : 2);  // (lanes 48-63 repeat Cr lanes' work and write nothing)
  const int m = lane < 32 ? lane >> 2 : (lane & 7);
  const int slot = lane < 32 ? (lane & 3) : 4 + comp - 1;  // block of the MCU in decode order
  const uint8_t *tile_coef = (const uint8_t *)p.coef + (int64_t)img * p.coef_image_stride + ((int64_t)my * p.mcus_x + mx0) * 768;
  const int32_t *qsrc = (const int32_t *)((const uint8_t *)p.qtabs + (int64_t)img * p.qtab_image_stride);
#pragma unroll
  for (int i = 0; i < 3; i++) qlds[i * kQPitch + lane] = qsrc[i * 64 + lane];
  float v[64];
  {
    uint32_t raw[32];
    const u32x4_t *src = (const u32x4_t *)(tile_coef + (uint32_t)(min(m, nvalid - 1) * 6 + slot) * 128u);
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const u32x4_t t = src[j];
      raw[j * 4 + 0] = t.x, raw[j * 4 + 1] = t.y, raw[j * 4 + 2] = t.z, raw[j * 4 + 3] = t.w;
    }
    __syncthreads();  // the tables are in LDS
    const int32_t *q = qlds + comp * kQPitch;
#pragma unroll
    for (int k = 0; k < 8; k++) {
#pragma unroll
      for (int i = 0; i < 8; i += 4) {
        const int4 q4 = *(const int4 *)(q + k * 8 + i);
        const int qq[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const uint32_t w = raw[k * 4 + ((i + e) >> 1)];
          const int c = ((i + e) & 1) ? ((int)w >> 16) : (int)(short)(w & 0xffffu);
          v[k * 8 + i + e] = (float)__mul24(c, qq[e]);  // jpeg.cpp:563-569
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; i++)  // column pass, jpeg.cpp:596-663
    aan_1d(v[0 * 8 + i], v[1 * 8 + i], v[2 * 8 + i], v[3 * 8 + i], v[4 * 8 + i], v[5 * 8 + i], v[6 * 8 + i], v[7 * 8 + i]);
#pragma unroll
  for (int k = 0; k < 8; k++)  // row pass, jpeg.cpp:664-731
    aan_1d(v[k * 8 + 0], v[k * 8 + 1], v[k * 8 + 2], v[k * 8 + 3], v[k * 8 + 4], v[k * 8 + 5], v[k * 8 + 6], v[k * 8 + 7]);

  // where this lane's block lands: luma block (bv, bh) of MCU m in strip rows 4*bv.., 8-sample column 2*m + bh;
  // a chroma block in its component's strip, column m.  16-B chunk c of a row is stored at c ^ ((c >> 3) & 1).
  const bool luma = comp == 0;
  const int bv = luma ? (slot >> 1) : 0, bh = luma ? (slot & 1) : 0;
  const int col = luma ? m * 2 + bh : m;
  const int sw = (col >> 2) & 1;
  const int pitch = luma ? 512 : 256;
  char *const dst = lds + (luma ? bv * 4 * 512 : kYStrip + (comp - 1) * kCStrip) + col * 32;
  const int x4 = lane & 31;
  const int rd_y = (x4 ^ ((x4 >> 3) & 1)) * 16;
  const int rd_c = ((x4 >> 1) ^ ((x4 >> 4) & 1)) * 16 + (x4 & 1) * 8;  // chroma chunk x4 / 2, its half x4 & 1
  uint8_t *const img_rgb = p.rgb + (int64_t)img * p.rgb_image_stride;
  const int x = mx0 * 16 + x4 * 4;
  const int npx = min(4, min(nvalid * 16, p.width - mx0 * 16) - x4 * 4);
#pragma unroll
  for (int phase = 0; phase < 2; phase++) {
    if (phase == 1) __syncthreads();
    if (active) {
#pragma unroll
      for (int kk = 0; kk < 4; kk++) {
        const int kl = phase * 4 + kk;                         // luma: rows 4p .. 4p+3 of the block
        const int kc = phase * 2 + (kk & 1) + (kk >> 1) * 4;   // chroma: rows 2p, 2p+1, 4+2p, 5+2p
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; e++) o[e] = luma ? v[kl * 8 + e] : v[kc * 8 + e];
        *(float4 *)(dst + kk * pitch + sw * 16) = make_float4(o[0], o[1], o[2], o[3]);
        *(float4 *)(dst + kk * pitch + (sw ^ 1) * 16) = make_float4(o[4], o[5], o[6], o[7]);
      }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; it++) {
      const int r = it * 2 + (lane >> 5);                      // luma strip row
      const int y = my * 16 + phase * 4 + (r >> 2) * 8 + (r & 3);
      const float4 Y = *(const float4 *)(lds + r * 512 + rd_y);
      const float2 B = *(const float2 *)(lds + kYStrip + (r >> 1) * 256 + rd_c);
      const float2 R = *(const float2 *)(lds + kYStrip + kCStrip + (r >> 1) * 256 + rd_c);
      const float yy[4] = {Y.x, Y.y, Y.z, Y.w}, cb[4] = {B.x, B.x, B.y, B.y}, cr[4] = {R.x, R.x, R.y, R.y};
      float rr[4], gg[4], bb[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {  // jpeg.cpp:521-535
        rr[i] = (yy[i] + JB_CR_R * cr[i]) + 128.0f;
        gg[i] = ((yy[i] - JB_CB_G * cb[i]) - JB_CR_G * cr[i]) + 128.0f;
        bb[i] = (yy[i] + JB_CB_B * cb[i]) + 128.0f;
      }
      if (p.fast_store) {
        uint32_t w0, w1, w2;
        pack12_rtz(rr, gg, bb, w0, w1, w2);
        if (y < p.height && npx == 4) {
          // a wave-uniform descriptor at row 0 of the MCU row; the lane adds its row and column
          uint8_t *const rows = img_rgb + (int64_t)(my * 16) * p.rgb_row_stride + (int64_t)mx0 * 48;
          const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(rows, 0, 0x7ffffff0, 0x00020000);
          __builtin_amdgcn_raw_buffer_store_b96(u32x3_t{w0, w1, w2}, rsrc, (y - my * 16) * (int)p.rgb_row_stride + x4 * 12, 0, JB_STORE_AUX);
        }
      }
      uint8_t *const o = img_rgb + (int64_t)y * p.rgb_row_stride + (int64_t)x * 3;
      if (y < p.height && npx > 0 && (!p.fast_store || npx < 4)) {
#pragma unroll
        for (int i = 0; i < 4; i++)
          if (i < npx) {
            o[i * 3 + 0] = (uint8_t)pack_u8(rr[i], 0, 0);
            o[i * 3 + 1] = (uint8_t)pack_u8(gg[i], 0, 0);
            o[i * 3 + 2] = (uint8_t)pack_u8(bb[i], 0, 0);
          }
      }
    }
  }
}

// 4:2:2 (HS = 2, VS = 1) and 4:4:0 (HS = 1, VS = 2): four blocks per MCU, so 16 MCUs fill a wave exactly -- lanes
// 0-31 the luma blocks (MCU l >> 1, block l & 1: left / right, or top / bottom), 32-47 Cb, 48-63 Cr.  4:2:2: strips of
// 4 rows x 256 luma / 128 chroma samples, a pixel row per wave-iteration; 4:4:0: 8 luma rows (rows 4p..4p+3 of both
// block rows) x 128 samples and the 4 chroma rows they need (selected per value, as in the 4:2:0 variant), two pixel
// rows per wave-iteration.
template <int HS, int VS>
__global__ __launch_bounds__(64) void jb_small_kernel_16(const JbLaunch p) {
  static_assert((HS == 2 && VS == 1) || (HS == 1 && VS == 2), "the four-blocks-per-MCU layouts");
  constexpr int kMcus = 16;
  constexpr int YW = kMcus * 8 * HS;            // luma strip width in samples: 256 / 128
  constexpr int YROWS = 4 * VS;                 // luma strip rows per phase: 4 / 8
  constexpr int kYStrip = YROWS * YW * 4;       // 4 KiB either way
  constexpr int kCPitch = kMcus * 8 * 4;        // 512 B: 128 chroma samples per row
  constexpr int kCStrip = 4 * kCPitch + 64;     // 2 KiB, skewed
  constexpr int kQPitch = 64 + 4;
  __shared__ __attribute__((aligned(16))) char lds[kYStrip + 2 * kCStrip + 3 * kQPitch * 4];
  int32_t *const qlds = (int32_t *)(lds + kYStrip + 2 * kCStrip);
  const int lane = threadIdx.x;
  const int tile = blockIdx.x;
  const int img = tile / p.tiles_per_image;
  const int rem = tile - img * p.tiles_per_image;
  const int my = rem / p.tiles_per_row;
  const int mx0 = (rem - my * p.tiles_per_row) * kMcus;
  const int nvalid = min(kMcus, p.mcus_x - mx0);
  const int comp = lane < 32 ? 0 : (lane < 48 ? 1 : 2);
  const int m = lane < 32 ? lane >> 1 : (lane & 15);
  const int slot = lane < 32 ? (lane & 1) : 2 + comp - 1;  // block of the MCU in decode order
  const uint8_t *tile_coef = (const uint8_t *)p.coef + (int64_t)img * p.coef_image_stride + ((int64_t)my * p.mcus_x + mx0) * 512;
  const int32_t *qsrc = (const int32_t *)((const uint8_t *)p.qtabs + (int64_t)img * p.qtab_image_stride);
#pragma unroll
  for (int i = 0; i < 3; i++) qlds[i * kQPitch + lane] = qsrc[i * 64 + lane];
  float v[64];
  {
    uint32_t raw[32];
    const u32x4_t *src = (const u32x4_t *)(tile_coef + (uint32_t)(min(m, nvalid - 1) * 4 + slot) * 128u);
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const u32x4_t t = src[j];
      raw[j * 4 + 0] = t.x, raw[j * 4 + 1] = t.y, raw[j * 4 + 2] = t.z, raw[j * 4 + 3] = t.w;
    }
    __syncthreads();  // the tables are in LDS
    const int32_t *q = qlds + comp * kQPitch;
#pragma unroll
    for (int k = 0; k < 8; k++) {
#pragma unroll
      for (int i = 0; i < 8; i += 4) {
        const int4 q4 = *(const int4 *)(q + k * 8 + i);
        const int qq[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const uint32_t w = raw[k * 4 + ((i + e) >> 1)];
          const int c = ((i + e) & 1) ? ((int)w >> 16) : (int)(short)(w & 0xffffu);
          v[k * 8 + i + e] = (float)__mul24(c, qq[e]);  // jpeg.cpp:563-569
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; i++)  // column pass, jpeg.cpp:596-663
    aan_1d(v[0 * 8 + i], v[1 * 8 + i], v[2 * 8 + i], v[3 * 8 + i], v[4 * 8 + i], v[5 * 8 + i], v[6 * 8 + i], v[7 * 8 + i]);
#pragma unroll
  for (int k = 0; k < 8; k++)  // row pass, jpeg.cpp:664-731
    aan_1d(v[k * 8 + 0], v[k * 8 + 1], v[k * 8 + 2], v[k * 8 + 3], v[k * 8 + 4], v[k * 8 + 5], v[k * 8 + 6], v[k * 8 + 7]);

  const bool luma = comp == 0;
  const int bsel = lane & 1;                                     // luma: right (4:2:2) or bottom (4:4:0) block of the MCU
  const int col = luma ? (HS == 2 ? m * 2 + bsel : m) : m;       // 8-sample column of the block in its strip
  const int row0 = luma && VS == 2 ? bsel * 4 : 0;               // first strip row of the block
  const int sw = (col >> 2) & 1;
  const int pitch = luma ? YW * 4 : kCPitch;
  char *const dst = lds + (luma ? row0 * (YW * 4) : kYStrip + (comp - 1) * kCStrip) + col * 32;
  constexpr int TPR = YW / 4;                                    // 4-pixel tasks per row: 64 / 32
  const int x4 = lane & (TPR - 1);
  const int rd_y = (x4 ^ ((x4 >> 3) & 1)) * 16;
  const int rd_c = HS == 2 ? ((x4 >> 1) ^ ((x4 >> 4) & 1)) * 16 + (x4 & 1) * 8 : rd_y;
  uint8_t *const img_rgb = p.rgb + (int64_t)img * p.rgb_image_stride;
  const int x = mx0 * 8 * HS + x4 * 4;
  const int npx = min(4, min(nvalid * 8 * HS, p.width - mx0 * 8 * HS) - x4 * 4);
#pragma unroll
  for (int phase = 0; phase < 2; phase++) {
    if (phase == 1) __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
      const int kl = phase * 4 + kk;                                            // rows 4p .. 4p+3 of the block
      const int kc = VS == 2 ? phase * 2 + (kk & 1) + (kk >> 1) * 4 : kl;       // 4:4:0 chroma: rows 2p, 2p+1, 4+2p, 5+2p
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; e++) o[e] = (VS == 2 && !luma) ? v[kc * 8 + e] : v[kl * 8 + e];
      *(float4 *)(dst + kk * pitch + sw * 16) = make_float4(o[0], o[1], o[2], o[3]);
      *(float4 *)(dst + kk * pitch + (sw ^ 1) * 16) = make_float4(o[4], o[5], o[6], o[7]);
    }
    __syncthreads();
    constexpr int ROWS_PER_IT = 64 / TPR;  // 1 / 2
#pragma unroll
    for (int it = 0; it < YROWS / ROWS_PER_IT; it++) {
      const int r = it * ROWS_PER_IT + (ROWS_PER_IT == 2 ? lane >> 5 : 0);   // luma strip row
      const int y_in = phase * 4 + (r >> 2) * 8 + (r & 3);
      const int y = my * 8 * VS + y_in;
      const float4 Y = *(const float4 *)(lds + r * (YW * 4) + rd_y);
      float cb[4], cr[4];
      const int crow = (r / VS) * kCPitch;
      if (HS == 2) {
        const float2 B = *(const float2 *)(lds + kYStrip + crow + rd_c);
        const float2 R = *(const float2 *)(lds + kYStrip + kCStrip + crow + rd_c);
        cb[0] = cb[1] = B.x, cb[2] = cb[3] = B.y;
        cr[0] = cr[1] = R.x, cr[2] = cr[3] = R.y;
      } else {
        const float4 B = *(const float4 *)(lds + kYStrip + crow + rd_c);
        const float4 R = *(const float4 *)(lds + kYStrip + kCStrip + crow + rd_c);
        cb[0] = B.x, cb[1] = B.y, cb[2] = B.z, cb[3] = B.w;
        cr[0] = R.x, cr[1] = R.y, cr[2] = R.z, cr[3] = R.w;
      }
      const float yy[4] = {Y.x, Y.y, Y.z, Y.w};
      float rr[4], gg[4], bb[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {  // jpeg.cpp:521-535
        rr[i] = (yy[i] + JB_CR_R * cr[i]) + 128.0f;
        gg[i] = ((yy[i] - JB_CB_G * cb[i]) - JB_CR_G * cr[i]) + 128.0f;
        bb[i] = (yy[i] + JB_CB_B * cb[i]) + 128.0f;
      }
      if (p.fast_store) {
        uint32_t w0, w1, w2;
        pack12_rtz(rr, gg, bb, w0, w1, w2);
        if (y < p.height && npx == 4) {
          uint8_t *const rows = img_rgb + (int64_t)(my * 8 * VS) * p.rgb_row_stride + (int64_t)mx0 * (24 * HS);
          const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(rows, 0, 0x7ffffff0, 0x00020000);
          __builtin_amdgcn_raw_buffer_store_b96(u32x3_t{w0, w1, w2}, rsrc, y_in * (int)p.rgb_row_stride + x4 * 12, 0, JB_STORE_AUX);
        }
      }
      uint8_t *const o = img_rgb + (int64_t)y * p.rgb_row_stride + (int64_t)x * 3;
      if (y < p.height && npx > 0 && (!p.fast_store || npx < 4)) {
#pragma unroll
        for (int i = 0; i < 4; i++)
          if (i < npx) {
            o[i * 3 + 0] = (uint8_t)pack_u8(rr[i], 0, 0);
            o[i * 3 + 1] = (uint8_t)pack_u8(gg[i], 0, 0);
            o[i * 3 + 2] = (uint8_t)pack_u8(bb[i], 0, 0);
          }
      }
    }
  }
}

template <int HS, int VS>
static hipError_t launch_t(const JbLaunch &p, hipStream_t stream) {
  using LM = LaneMap<HS, VS>;
  // does any wave hold two components whose tables may differ?
  constexpr bool kLumaChromaMixed = (LM::NYT % 64 != 0);                      // no layout any more (see tile_blocks)
  constexpr bool kCbCrMixed = (LM::MCUS % 64 != 0);                           // 4:2:0: Cb and Cr share the third wave
  const bool mixq = kLumaChromaMixed || (kCbCrMixed && !p.chroma_q_equal);
  // the linear tiling is a separate instantiation: where the row-bound tiling leaves no tile
  // ragged (mcus_x a multiple of the tile length, e.g. 4096- and 8192-pixel rows) the simpler
  // row-bound code is 2 % faster
  constexpr bool kCanLinear = ((LM::MCUS * 8 * HS / 4) % 64 == 0);
  const dim3 grid(p.n_tiles), block(LM::TB);
  // hipGetLastError below must report THIS launch: an error left in the thread's error slot by an
  // unrelated earlier call (a failed attribute query, say) is not this launch's
  (void)hipGetLastError();
  constexpr unsigned extra_lds = 0;
#if defined(JB_LAB)
  if (kCanLinear && p.linear && p.staged) {
    if (mixq) hipLaunchKernelGGL((jb_tile_kernel<HS, VS, true, kCanLinear, kCanLinear>), grid, block, extra_lds, stream, p);
    else hipLaunchKernelGGL((jb_tile_kernel<HS, VS, false, kCanLinear, kCanLinear>), grid, block, extra_lds, stream, p);
  } else
#endif
  if (kCanLinear && p.linear) {
    if (mixq) hipLaunchKernelGGL((jb_tile_kernel<HS, VS, true, kCanLinear>), grid, block, extra_lds, stream, p);
    else hipLaunchKernelGGL((jb_tile_kernel<HS, VS, false, kCanLinear>), grid, block, extra_lds, stream, p);
  } else {
    if (mixq) hipLaunchKernelGGL((jb_tile_kernel<HS, VS, true, false>), grid, block, extra_lds, stream, p);
    else hipLaunchKernelGGL((jb_tile_kernel<HS, VS, false, false>), grid, block, extra_lds, stream, p);
  }
  return hipGetLastError();
}

int jbk_mcus_per_tile(int hs, int vs) { return tile_blocks(hs, vs) / (hs * vs + 2); }

// The linear tiling needs MCU rows at least one 256-pixel segment long, so that a segment wraps to
// the next MCU row at most once.
int jbk_linear_ok(int hs, int vs, int mcus_x) {
  const int strip_tasks_per_row = jbk_mcus_per_tile(hs, vs) * 8 * hs / 4;
  if (strip_tasks_per_row % 64 != 0) return 0;  // (no layout: see tile_blocks)
  return mcus_x >= 256 / (8 * hs);
}

int jbk_small_mcus(int hs, int vs) { return hs == 1 && vs == 1 ? kSmallMcus : hs == 2 && vs == 2 ? kSmallMcus420 : 16; }

hipError_t jbk_launch(const JbLaunch &p, int hs, int vs, hipStream_t stream) {
  if (p.n_tiles <= 0) return hipSuccess;
  if (p.small_grid) {  // (the host sets it with the tile counts of this tiling)
    if (jbk_small_mcus(hs, vs) == 0) return hipErrorInvalidValue;
    (void)hipGetLastError();
    if (hs == 1 && vs == 1) hipLaunchKernelGGL(jb_small_kernel_444, dim3(p.n_tiles), dim3(64), 0, stream, p);
    else if (hs == 2 && vs == 2) hipLaunchKernelGGL(jb_small_kernel_420, dim3(p.n_tiles), dim3(64), 0, stream, p);
    else if (hs == 2) hipLaunchKernelGGL((jb_small_kernel_16<2, 1>), dim3(p.n_tiles), dim3(64), 0, stream, p);
    else hipLaunchKernelGGL((jb_small_kernel_16<1, 2>), dim3(p.n_tiles), dim3(64), 0, stream, p);
    return hipGetLastError();
  }
  if (hs == 1 && vs == 1) return launch_t<1, 1>(p, stream);
  if (hs == 2 && vs == 1) return launch_t<2, 1>(p, stream);
  if (hs == 1 && vs == 2) return launch_t<1, 2>(p, stream);
  if (hs == 2 && vs == 2) return launch_t<2, 2>(p, stream);
  return hipErrorInvalidValue;
}

const char *jbk_kernel_name(int hs, int vs) {
  // <HS, VS, MIXQ, LINEAR>: LINEAR depends on the image width; MIXQ only exists for 4:2:0 whose
  // Cb and Cr name different tables
  if (hs == 1 && vs == 1) return "jb_tile_kernel<1, 1, false, *>";
  if (hs == 2 && vs == 1) return "jb_tile_kernel<2, 1, false, *>";
  if (hs == 1 && vs == 2) return "jb_tile_kernel<1, 2, false, *>";
  return "jb_tile_kernel<2, 2, *, *>";
}
