// jb_huff.h -- internal interface of the device-side entropy decoder (jb_huff.hip): Huffman
// decoding of a baseline scan on the GPU, one lane per chunk of at most 128 bytes of every restart
// interval (self-synchronising, verified).  Not part of the public ABI.  Plain C structs shared by
// the host code that fills them (jb_frontend.cpp, jb_api.cpp) and the kernels that read them; the
// per-symbol step both sides agree on is jb_huff_core.h.
#pragma once
#include <stdint.h>

// ---- lookup tables ------------------------------------------------------------------------------
// A symbol is resolved by ONE 16-bit entry that already holds everything the step needs:
//   bits 0..4   total = code length + magnitude bits (1..27); 0 = no symbol here (see below)
//   bits 5..8   size  = magnitude bits (DC 0..11, AC 0..10)
//   bits 9..15  adv   = what the position k in the block advances by:
//                 DC symbol 1 | AC run/size run + 1 | AC run without a coefficient (ZRL = 16 zeros,
//                 and the run-only symbols 0x10..0xE0 the reference accepts, jpeg.cpp:366-385)
//                 run + 1, and the step takes 1 off again | EOB 82
//               so "k + adv in 65..81" is exactly the reference's "Invalid AC length" (jpeg.cpp:372).
// First level: the first kJbT1Bits bits of the window.  A code longer than that (all of them sit at
// the top of the code space) has an entry of 1..kJbT2Tables: the number + 1 of its second-level table
// of 2^(16 - kJbT1Bits) entries, indexed by the following bits (no symbol's entry is below 512).  Symbols the
// reference rejects (DC size > 11, AC size > 10) are not in the tables at all: entry 0.
// Up to three tables per kind (a frame has three components): the AC tables in use first, then the
// DC tables in use (JbHuffImage::n_tabs first-level tables in all); in LDS the second-level pool
// follows them directly.
constexpr int kJbT1Bits = 10;
constexpr int kJbT2Bits = 16 - kJbT1Bits;
constexpr uint32_t kJbT1Entries = 1u << kJbT1Bits;
constexpr uint32_t kJbT2Entries = 1u << kJbT2Bits;
constexpr uint32_t kJbT2Tables = 24;  // second-level tables a set can hold (Annex K needs 11)
constexpr uint32_t kJbMaxTabs = 6;    // first-level tables: up to three AC and three DC
struct JbHuffTables {
  uint16_t t1[kJbMaxTabs][kJbT1Entries];
  uint16_t t2[kJbT2Tables][kJbT2Entries];
};
constexpr uint32_t kJbAdvEob = 82;

struct JbHuffImage {
  uint32_t scan_off;   // byte offset of the image's clean scan in the scan buffer (multiple of 16, at least 16)
  uint32_t scan_len;   // clean bytes; at least 64 zero bytes follow
  uint32_t int_off;    // index of the image's first entry in `starts` (n_int + 1 entries, bytes from scan_off)
  uint32_t n_int;      // restart intervals
  uint32_t ri;         // MCUs per interval
  uint32_t n_mcus;     // MCUs in the image
  uint32_t nb;         // blocks per MCU (1 for a single-component frame, else luma blocks + 2)
  uint32_t table_set;  // index into the table sets
  int64_t coef_off;    // byte offset of the image's coefficient blocks in the output
  // block-in-MCU -> table / component, 4 bits per block (block b at bits 4b..4b+3):
  uint32_t lut_ac;     // first-level table index of the block's AC table
  uint32_t lut_dc;     // first-level table index of the block's DC table
  uint32_t lut_comp;   // component (0..2) of the block
  uint32_t n_chunks;   // chunks of the image: per interval ceil(bytes / chunk_bytes), at least 1
  uint32_t state_off;  // index of the image's first entry in the per-chunk arrays
  uint32_t n_blocks;   // coded blocks in the image: n_mcus * nb
  uint32_t chunk_bytes; // bytes of clean scan per lane: 64 or 128 (jb_huff_prepare_ chooses)
  uint32_t wg0;        // index of the image's first workgroup in the launch's list
  uint32_t needs_sync; // 0: every chunk is the first of its interval (start states known: no synchronisation passes)
  uint32_t blk_bytes;  // bytes from one decoded block to the next in the output: 128 (a single-component frame
                       // delivered as three: 384)
  uint32_t n_tabs;     // first-level tables of its table set (AC slots first, then DC slots: lut_dc counts from the number of AC slots)
};

struct JbHuffWg {  // one workgroup OWNS up to kJbOwnChunks consecutive chunks of one image
  uint32_t image;
  uint32_t first_chunk;
};

constexpr uint32_t kJbChunkBytes = 128;  // bytes of clean scan per lane, at most (JbHuffImage::chunk_bytes)
constexpr int kJbHuffLanes = 256;        // lanes per workgroup
// A workgroup of the synchronisation decodes the kJbWarmChunks chunks in front of the ones it owns as well: its first
// own chunk then starts from a state that has had eight chunks to fall into step -- so the exit state its left
// neighbour arrives at for the same chunk is, as a rule, the one it already started from, and the second launch finds
// nothing to do (it still runs: it is what makes the result right when the rule fails).  All three kernels cut an image
// the same way, so that the workgroups' sums line up: the writing pass and the DC pass leave those lanes idle (3 %).
constexpr uint32_t kJbWarmChunks = 8;
constexpr uint32_t kJbOwnChunks = kJbHuffLanes - kJbWarmChunks;
constexpr int kJbSyncLaunches = 2;       // synchronisation launches before the writing pass (which verifies).  The lanes
                                         // of a workgroup fall into step with each other inside ONE launch (passes over
                                         // LDS, until nothing changes); a further launch carries the exit state of a
                                         // workgroup's last chunk into the next workgroup's first
constexpr int kJbSyncLaunchesMax = 8;    // what a retry after "not in step" may ask for

// Chunks never straddle a restart boundary: interval i is cut into chunks from its own first byte
// (an empty interval still has one chunk, so that its missing blocks are noticed).
inline uint32_t jb_chunks_of_(uint32_t interval_bytes, uint32_t chunk_bytes) { return interval_bytes ? (interval_bytes + chunk_bytes - 1) / chunk_bytes : 1u; }
struct JbChunkDesc {
  uint32_t start;  // first byte of the chunk in the image's clean scan
  uint32_t seg;    // the restart interval it lies in; bit 31: it is the interval's first chunk (its start state is known)
};

// State between two symbols: where the next symbol starts and what it is
struct JbChunkState {
  uint32_t bitpos;  // bit position in the image's clean scan
  uint32_t meta;    // k (0 = a DC symbol is next, else zig-zag position) | block-in-MCU << 8 | blocks completed in the chunk << 16 (exit states)
};

// A re-decode of a chunk can stop as soon as it meets the path of the chunk's previous decode: from
// equal (bit position, k, block-in-MCU) on, everything is the same.  The previous path is remembered
// at the first symbol boundary behind every kJbCheckpointBits bits (counted in the lane's local
// coordinates, see jb_huff_core.h), with the counts up to there.
constexpr uint32_t kJbCheckpointBits = 256;
constexpr uint32_t kJbCheckpoints = 4;  // records per chunk (a 128-byte chunk crosses at most four boundaries), stored [checkpoint][chunk]:
                                        // local position u | k << 11 | block-in-MCU << 17 | blocks completed before this place << 20 (jbh_pack_state)

// What a workgroup's chunks add up to, for the writing pass's bases (segmented sums over the chunks
// of each interval, restarting at every interval's first chunk)
struct JbWgSum {
  uint32_t has_first;   // 1: a chunk of this workgroup is the first of its interval
  uint32_t blocks;      // has_first: index of the block behind the workgroup's last chunk, counted from the image's
                        // first; else blocks completed by the workgroup's chunks (written by the synchronisation)
  uint32_t dc[3];       // DC differences summed: behind the last interval start (has_first), else over all chunks
                        // (written by the writing pass)
  uint32_t pad[3];
};
// What the writing pass leaves for jb_huff_dc_kernel, per chunk
struct JbChunkDc {
  uint32_t dc[3];        // DC differences of the blocks whose DC symbol lies in the chunk, summed per component
  uint32_t first_block;  // the first of those blocks ...
  uint32_t count;        // ... and how many they are
  uint32_t pad[3];
};

struct JbHuffLaunch {
  const uint8_t *scan;         // device: clean entropy-coded bytes of all images
  const uint32_t *starts;      // device: interval start offsets
  const JbHuffTables *tables;  // device
  const JbHuffImage *images;   // device
  const JbHuffWg *wgs;         // device: every workgroup of the submission
  const JbHuffWg *sync_wgs;    // device: the workgroups of the images with needs_sync (a sub-list, same order)
  int16_t *coef;               // device: output, decode-order int16 blocks (include/jpegblk.h)
  uint32_t *status;            // device: one word per image: bit 0 corrupt data, bit 1 overrun, bit 2 the chunks are not in step
  int32_t n_wgs;
  int32_t n_sync_wgs;
  const JbChunkDesc *chunks;   // device: one descriptor per chunk
  // device scratch, one entry per chunk each
  JbChunkState *entry;         // the state the chunk was last decoded from
  JbChunkState *exit;          // and the state that decode ended in
  uint32_t *cps;               // kJbCheckpoints records per chunk, [checkpoint][chunk] (kept between the synchronisation launches)
  JbChunkDc *chunk_dc;         // what the writing pass leaves for jb_huff_dc_kernel
  JbWgSum *wgsum;              // one per workgroup (indexed like `wgs`)
  uint32_t n_chunks_total;     // chunks of the submission (the [checkpoint][chunk] arrays' row length)
  int32_t sync_launches;       // synchronisation launches (>= 1)
  uint32_t max_chunk_bytes;    // the largest JbHuffImage::chunk_bytes of the submission (sizes the workgroups' LDS)
  uint32_t max_tabs;           // the largest JbHuffImage::n_tabs
};

// (the launch function is declared in jb_kernels.h: this header stays free of HIP types, the host
// front end is also built for the CPU alone by tools/fuzz)

// ---- host side (jb_frontend.cpp prepares, jb_api.cpp packs + launches) ----------------------
#include <string>
#include <vector>

#include "../../include/jpegblk.h"

// One image readied for the device decoder: headers parsed, scan de-stuffed, tables built.
struct JbHuffJob {
  jb_image_desc desc;
  jb_geometry geo;
  uint16_t qtabs[256];
  JbHuffImage img;               // n_int, ri, n_mcus, nb, luts; offsets are filled by the packer
  JbHuffTables tables;
  uint32_t n_tabs = 0;           // first-level tables in use
  std::vector<uint8_t> scan;     // clean bytes + >= 64 zero bytes
  size_t scan_len = 0;           // clean bytes
  std::vector<uint32_t> starts;  // n_int + 1 entries
};
// JB_OK: the image is eligible and `job` is filled.  JB_ERR_UNSUPPORTED: a valid stream the device
// decoder does not take (markers that do not match the frame, Huffman tables whose long codes do
// not fit the second-level pool, a frame for the general front end) -- use the host decoder.
// Other negatives: the header errors of jb_entropy_decode.
// chunk_bytes: 0 = the default (kJbChunkBytes), else 64 or 128 (JbKnobs::chunk_bytes of the calling object).
int jb_huff_prepare_(const uint8_t *jpeg, size_t jpeg_bytes, JbHuffJob *job, std::string *err, uint32_t chunk_bytes = 0);

// Where the pieces of a packed submission lie in its blob:
//   [JbHuffImage x n][JbHuffWg x n_wg][JbHuffWg x n_sync_wg][JbHuffTables x n_sets][starts][chunk descriptors][scans, 16-byte aligned each]
// Table sets are shared by the images that use identical tables (the usual case: one set).
struct JbHuffLayout {
  size_t off_img = 0, off_wg = 0, off_sync_wg = 0, off_tab = 0, off_starts = 0, off_chunks = 0, off_scan = 0, total = 0;
  // device-only scratch behind the uploaded bytes
  size_t off_entry = 0, off_exit = 0, off_cps = 0, off_chunk_dc = 0, off_wgsum = 0, device_total = 0;
  int n = 0, n_wg = 0, n_sync_wg = 0;
  uint32_t n_chunks = 0, max_chunk_bytes = 0, max_tabs = 0;
  int64_t coef_stride = 0;
};
// Is the device decoder worth taking for this image?  At least `min_chunks` chunks.
inline bool jb_huff_worth_it_(const JbHuffJob &job, uint32_t min_chunks) { return job.img.n_chunks >= min_chunks; }
// Upper bound of the blob size for these jobs; and the packing itself (pure host code, no HIP: the
// batch decoder's threads pack into their own pinned buffers WITHOUT holding the shared context's
// lock -- twelve megabytes of memcpy per group under that lock serialised the whole decoder).
size_t jb_huff_pack_size_(const JbHuffJob *const *jobs, int n);
int jb_huff_pack_(const JbHuffJob *const *jobs, int n, int64_t coef_stride, uint8_t *dst, JbHuffLayout *lay);
// Fill one first-level table (and the second-level tables its long codes need) from a canonical
// table (counts of codes per length 1..16, symbols in code order).  false: the pool is full.
bool jb_huff_fill_table_(const uint8_t counts[17], const uint8_t *symbols, bool is_ac, JbHuffTables *set, uint32_t tix, uint32_t n_tabs, uint32_t *n_t2);
