// jb_huff.h -- internal interface of the device-side entropy decoder (jb_huff.hip): Huffman
// decoding of a baseline scan on the GPU -- one lane per restart interval where the intervals are
// short, one lane per 256-byte chunk of the scan (self-synchronising, verified) otherwise.  Not
// part of the public ABI.  Plain C structs shared by the host code that fills them (jb_frontend.cpp, jb_api.cpp) and
// the kernel that reads them.
#pragma once
#include <stdint.h>

// One table set: what the scans of one or more images decode with.  Slots 0/1 of each kind; an
// image maps its three components onto them (JbHuffImage::dc_slot / ac_slot).  `acl` / `dcl`
// resolve a CODE of up to 11 bits from an 11-bit window -- (length << 8) | symbol, 0 = longer -- and
// the lane takes the magnitude bits that follow out of the same 32-bit window; the canonical
// arrays (the host decoder's, reference huffman.hpp:17-29 builds the same codes) serve codes of
// 12..16 bits.  Small on purpose: 18 KiB of LDS per workgroup, so four workgroups fit a CU.
struct JbHuffTables {
  uint16_t acl[2][2048];
  uint16_t dcl[2][2048];
  int32_t maxcode[4][18];   // [0,1] = DC slots, [2,3] = AC slots
  int32_t valptr[4][20];
  int32_t mincode[4][20];
  uint8_t symbols[4][256];
};

struct JbHuffImage {
  uint32_t scan_off;   // byte offset of the image's clean scan in the scan buffer (multiple of 16)
  uint32_t scan_len;   // clean bytes; at least 64 zero bytes follow
  uint32_t int_off;    // index of the image's first entry in `starts` (n_int + 1 entries, bytes from scan_off)
  uint32_t n_int;      // restart intervals
  uint32_t ri;         // MCUs per interval
  uint32_t n_mcus;     // MCUs in the image
  uint32_t ny;         // luma blocks per MCU (blocks per MCU = ny + 2)
  uint32_t table_set;  // index into the table sets
  int64_t coef_off;    // byte offset of the image's coefficient blocks in the output
  uint8_t dc_slot[4];  // table slot (0 / 1) of Y, Cb, Cr
  uint8_t ac_slot[4];
  // scans without restart intervals, and scans whose intervals are long: the self-synchronising
  // decoder works on chunks of the intervals (a scan without DRI is one interval)
  uint32_t n_chunks;   // chunks of the image (0 = the interval decoder is used): per interval ceil(bytes / kJbChunkBytes), at least 1
  uint32_t state_off;  // index of the image's first entry in the chunk descriptor / state / sum / base arrays
  uint32_t n_blocks;   // coded blocks in the image: n_mcus * (ny + 2)
  uint32_t chunk_bytes; // bytes of clean scan per chunk lane: 64, 128 or kJbChunkBytes (jb_huff_prepare_ chooses)
};

struct JbHuffWg {  // one workgroup = up to kJbHuffLanes consecutive restart intervals of one image
  uint32_t image;
  uint32_t first_interval;
};

constexpr uint32_t kJbLongInterval = 1024; // mean bytes per restart interval from which the chunk decoder is used for a file with DRI
constexpr uint32_t kJbChunkBytes = 256;  // bytes of clean scan per lane of the self-synchronising decoder, at most (JbHuffImage::chunk_bytes)
constexpr int kJbSyncRounds = 16;        // synchronisation passes before the writing pass (which verifies): the default (24 with chunks below kJbChunkBytes);
                                         // a lane whose start state did not change since the last pass skips its decode

// Chunks never straddle a restart boundary: interval i is cut into chunks from its own first byte
// (an empty interval still has one chunk, so that its missing blocks are noticed).
inline uint32_t jb_chunks_of_(uint32_t interval_bytes, uint32_t chunk_bytes) { return interval_bytes ? (interval_bytes + chunk_bytes - 1) / chunk_bytes : 1u; }
struct JbChunkDesc {
  uint32_t start;  // first byte of the chunk in the image's clean scan
  uint32_t seg;    // the restart interval it lies in; bit 31: it is the interval's first chunk (its start state is known)
};

// Exit state of a chunk's decode: where the first symbol of the next chunk starts and in which state
struct JbChunkState {
  uint32_t bitpos;  // bit position in the clean scan
  uint32_t meta;    // k (0 = a DC symbol is next, else zig-zag position) | block-in-MCU << 8 | blocks completed in the chunk << 16
};

// A re-decode of a chunk (sync passes r > 0) can stop as soon as it meets the path of the chunk's
// previous decode: from equal (bit position, k, block-in-MCU) on, everything is the same.  The
// previous path is remembered at kJbCheckpoints places inside the chunk -- the first symbol boundary
// at or behind every kJbCheckpointBits bits -- with the counts up to there.
constexpr uint32_t kJbCheckpointBits = 256;
constexpr uint32_t kJbCheckpoints = kJbChunkBytes * 8 / kJbCheckpointBits - 1;  // 7 inside a chunk of kJbChunkBytes (fewer in a smaller one)
struct JbCheckpoint {   // 32 bytes
  uint32_t bitpos, meta;     // as in JbChunkState, without the block count
  uint32_t nblk;             // blocks completed in the chunk before this place
  uint32_t dc[3];            // DC differences summed before this place, per component
  uint32_t pad[2];
};

struct JbHuffLaunch {
  const uint8_t *scan;         // device: clean entropy-coded bytes of all images
  const uint32_t *starts;      // device: interval start offsets
  const JbHuffTables *tables;  // device
  const JbHuffImage *images;   // device
  const JbHuffWg *wgs;         // device
  int16_t *coef;               // device: output, decode-order int16 blocks (include/jpegblk.h)
  uint32_t *status;            // device: one word per image: bit 0 corrupt data, bit 1 overrun, bit 2 the chunks did not synchronise
  int32_t n_wgs;
  // the self-synchronising decoder (images with n_chunks > 0): its own workgroup list and scratch
  const JbHuffWg *sync_wgs;    // device: one workgroup = up to kJbHuffLanes consecutive chunks of one image
  int32_t n_sync_wgs;
  int32_t n_sync_images;       // images with n_chunks > 0 ...
  const uint32_t *sync_images; // ... and their indices (device)
  JbChunkState *state_a, *state_b;  // device scratch, one entry per chunk each: exit states of the passes, ping-pong
  JbChunkState *state_in;           // device scratch: the start state each chunk was last decoded from
  const JbChunkDesc *chunks;   // device: one descriptor per chunk
  JbCheckpoint *cps;           // device scratch, 8 records per chunk (kJbCheckpoints used)
  uint32_t *dcsum;             // device scratch, 4 words per chunk: sum of the DC differences decoded in the chunk, per component (Y, Cb, Cr, -)
  uint32_t *base;              // device scratch, 4 words per chunk: index of the block the chunk starts in; DC predictors (Y, Cb, Cr) at its start
  int32_t sync_rounds;         // synchronisation passes (>= 1); n_chunks of them always suffice
};

constexpr int kJbHuffLanes = 256;  // restart intervals per workgroup (LDS: 18 KiB of tables + a 64-byte stream ring per lane)
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
  JbHuffImage img;               // n_int, ri, n_mcus, ny, slots; offsets are filled by the packer
  JbHuffTables tables;
  std::vector<uint8_t> scan;     // clean bytes + >= 64 zero bytes
  size_t scan_len = 0;           // clean bytes
  std::vector<uint32_t> starts;  // n_int + 1 entries
};
// JB_OK: the image is eligible and `job` is filled (img.n_chunks says which decoder: files without
// DRI and files whose intervals average kJbLongInterval bytes or more take the chunk decoder;
// JPEGBLK_HUFF_MODE=interval / chunk forces one for files with DRI).  JB_ERR_UNSUPPORTED: a valid stream the device
// decoder does not take (markers that do not match the frame, more than two
// DC or AC tables in use, a frame for the general front end) -- use the host decoder.  Other
// negatives: the header errors of jb_entropy_decode.
int jb_huff_prepare_(const uint8_t *jpeg, size_t jpeg_bytes, JbHuffJob *job, std::string *err);

// Where the pieces of a packed submission lie in its blob:
//   [JbHuffImage x n][JbHuffWg x n_wg][JbHuffTables x n_sets][starts][scans, 16-byte aligned each]
// Table sets are shared by the images that use identical tables (the usual case: one set).
struct JbHuffLayout {
  size_t off_img = 0, off_wg = 0, off_tab = 0, off_starts = 0, off_scan = 0, total = 0;
  int n = 0, n_wg = 0;
  int64_t coef_stride = 0;
  // the self-synchronising decoder's part: its workgroup list and image list (uploaded), and the
  // device-only scratch behind the uploaded bytes (chunk states x 2, chunk bases)
  size_t off_sync_wg = 0, off_sync_img = 0, off_chunks = 0, off_state_a = 0, off_state_b = 0, off_state_in = 0, off_cps = 0, off_dcsum = 0, off_base = 0, device_total = 0;
  int n_sync_wg = 0, n_sync_images = 0;
  uint32_t min_chunk_bytes = kJbChunkBytes;  // the smallest chunk size among the submission's images (smaller chunks: more passes)
};
// Is the device decoder worth taking for this image?  Interval decoder: at least `min_intervals`
// intervals; self-synchronising decoder (no DRI, or long intervals): at least 16 chunks.
inline bool jb_huff_worth_it_(const JbHuffJob &job, uint32_t min_intervals) {
  return job.img.n_chunks > 0 ? job.img.n_chunks >= 16u : job.img.n_int >= min_intervals;
}
// Upper bound of the blob size for these jobs; and the packing itself (pure host code, no HIP: the
// batch decoder's threads pack into their own pinned buffers WITHOUT holding the shared context's
// lock -- twelve megabytes of memcpy per group under that lock serialised the whole decoder).
size_t jb_huff_pack_size_(const JbHuffJob *const *jobs, int n);
int jb_huff_pack_(const JbHuffJob *const *jobs, int n, int64_t coef_stride, uint8_t *dst, JbHuffLayout *lay);
