/* jpegblk.h -- C ABI of the MI355X-native JPEG block pipeline
 *             (dequantize -> 8x8 inverse DCT -> YCbCr->RGB).
 *
 * This library replaces ONE seam of the reference decoder (aswanthabam/JPEG_Decoder): the
 * three consecutive calls
 *
 *     this->mcus = decodeHuffman();
 *     dequantize();  inverseDCT();  YCbCrToRGB();        reference jpeg.cpp:785-788
 *
 * inside Image::process_image_data (jpeg.cpp:755-789).  Everything before the seam (marker
 * parsing, Huffman decoding) runs on the host; everything behind this ABI runs as
 * hand-written HIP kernels on gfx950.  There is no CPU fallback in this library: if no HIP
 * device is usable every compute entry point returns JB_ERR_HIP.
 *
 * DATA CONTRACT (what the seam carries)
 *   coefficients  int16, natural (de-zigzagged) order, 64 per 8x8 block (128 B), blocks in
 *                 decode order = the order decodeHuffman() visits them (jpeg.cpp:415-443):
 *                 MCUs in raster order; per MCU hs*vs luma blocks (v-major, h-minor), then
 *                 one Cb block, then one Cr block.  The reference stores them as
 *                 MCU::y/cb/cr int[64] (include/types.hpp:32-67).
 *   quant tables  up to 4 tables x 64 entries, natural order, uint16.  (The reference keeps
 *                 one byte per entry -- jpeg.cpp:216,223, types.hpp:86-92 -- so parity with it
 *                 is defined for entries <= 255.)
 *   geometry      jb_image_desc; derived sizes as read_sof computes them (jpeg.cpp:77-80,
 *                 118-127) are returned by jb_geometry().
 *   pixels        uint8 R,G,B interleaved, row stride given in bytes, cropped to
 *                 width x height: rgb[y*stride + 3*x + c] =
 *                 mcus[(y/8)*mcuWidthReal + x/8].{r,g,b}[(y%8)*8 + x%8], the linearisation
 *                 both reference sinks use (include/display.hpp:19-34, jpeg.cpp:488-499).
 *
 * ARITHMETIC CONTRACT: bit-exact with the reference CPU path (int32 dequantize, the float AAN
 * butterfly network of jpeg.cpp:594-732 with truncation toward zero after each 1-D pass, the
 * float colour transform of jpeg.cpp:521-535 with truncation then clamp) for every int16
 * coefficient and every table entry <= 255.
 *
 * OWNERSHIP: all buffers are caller-owned; the library never frees or retains caller pointers
 * past the call (async: past jb_wait).  ERRORS: every function returns a jb_status, never
 * calls exit() (the reference logs and exit(1)s, e.g. jpeg.cpp:71-72,85-86).
 * THREADING: a jb_ctx is bound to one device and its own HIP streams; calls on one ctx must be
 * serialised by the caller, different contexts may be used concurrently from different threads.
 *
 * ENVIRONMENT.  None of these is needed for normal use: they select a path for tests and A/B
 * measurements, or adapt the host side to its machine.  Each is read ONCE PER OBJECT -- when a
 * context (jb_ctx_create) or a batch decoder (jb_batch_decoder_create*, jb_decode_batch) is created
 * -- and kept there (csrc/jb_knobs.h is the one place that reads them):
 *   JPEGBLK_GPU_HUFFMAN    where the entropy stage runs: unset = batch decoders on the device for files
 *                          of 16 chunks (2 KB of scan) or more, single images from 128 KB of scan on;
 *                          0 = always the host threads (north_star's split); 1 = the device for every
 *                          file of 16 chunks or more; 2 = the device for every file it takes
 *   JPEGBLK_CHUNK_BYTES    64 | 128: scan bytes per lane of the device entropy decoder (default 128)
 *   JPEGBLK_BYTE_STORE     1 = every pixel through byte stores (the second store implementation)
 *   JPEGBLK_ROW_TILING     1 = the row-bound tiling for every image
 *   JPEGBLK_PASS1          1 = a batch run always reads every file's headers first (default: only while the decoder's
 *                          buffers do not exist yet; otherwise a file is parsed when its group is formed)
 *   JPEGBLK_GROUP_RAMP     1 = a host thread's first two device groups are a quarter and a half of the full size
 *   JPEGBLK_GROUP_MB       MB of coefficients per group of small images on the host path (16; 0 = one image per submission)
 *   JPEGBLK_DEV_GROUP_MB   MB of coefficients per group whose entropy stage runs on the device (96)
 *   JPEGBLK_NUMA           0 = leave the host threads' CPU affinity alone, 1 = always bind them to the GPU's node
 *   JPEGBLK_OVERSUBSCRIBE  1 = allow more host threads than CPUs the process may use
 *   JPEGBLK_STAGED_STORE   1 = (measurement builds of the kernels only, tools/build_variant.sh; the product ignores it) the
 *                          staged, line-aligned store stage for every image that takes the linear tiling
 *   JPEGBLK_SMALL_GRID     1 = always the one-wave kernels (every layout has one), 0 = never (default: launches of
 *                          up to 8 workgroups per CU of the 192-lane kernel, e.g. one to four 1080p images, one 4096x4096 4:2:0)
 *   JPEGBLK_TIMING         1 | 2 | 3 = where one decode(bytes) / one device-entropy submission / one batch run spends its time (stderr)
 *   JPEGBLK_HW_QUEUES      read when the library is LOADED: hardware queues to ask the HIP runtime for
 *                          (GPU_MAX_HW_QUEUES; default 16, 0 = the runtime's default).  Process-wide, and only
 *                          effective before HIP initialises: an application that initialises HIP first sets
 *                          GPU_MAX_HW_QUEUES=16 itself (batch decoders: +13 % host path, +13-40 % device path).
 */
#ifndef JPEGBLK_H
#define JPEGBLK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JB_ABI_VERSION 1

typedef enum jb_status {
  JB_OK = 0,
  JB_ERR_NULL = -1,        /* a required pointer is NULL                                   */
  JB_ERR_GEOMETRY = -2,    /* width/height/stride out of range                             */
  JB_ERR_SAMPLING = -3,    /* luma factors not in {1,2}x{1,2} (reference jpeg.cpp:110-136) */
  JB_ERR_QTAB = -4,        /* qtab_id outside 0..3 (reference jpeg.cpp:205-209)            */
  JB_ERR_CAPACITY = -5,    /* image larger than the context / buffer was created for       */
  JB_ERR_HIP = -6,         /* HIP runtime error; text via jb_last_error()                  */
  JB_ERR_STATE = -7,       /* bad ticket / nothing in flight / context busy                */
  JB_ERR_FORMAT = -8,      /* front end: not a JPEG / corrupt segment                      */
  JB_ERR_UNSUPPORTED = -9  /* front end: a frame type it does not decode (12-bit, arithmetic, 4 components ...) */
} jb_status;

/* What Image state the seam reads: image_width/height (jpeg.cpp:792-793), the luma sampling
 * factors (jpeg.cpp:32-33) and color_components[i].quantizationTableID (jpeg.cpp:23). */
typedef struct jb_image_desc {
  int32_t width;      /* pixels, 1..65535 */
  int32_t height;     /* pixels, 1..65535 */
  int32_t hs;         /* luma horizontal sampling factor, 1 or 2; chroma is always 1x1 */
  int32_t vs;         /* luma vertical sampling factor, 1 or 2 */
  int32_t qtab_id[3]; /* quantisation table of Y, Cb, Cr: 0..3 */
  int32_t reserved;   /* must be 0 */
} jb_image_desc;

/* Sizes derived from a descriptor (read_sof, jpeg.cpp:77-80 and 118-127). */
typedef struct jb_geometry {
  int32_t mcu_w, mcu_h;           /* 8x8 block columns/rows covering the image: (W+7)/8, (H+7)/8 */
  int32_t mcu_w_real, mcu_h_real; /* rounded up to a multiple of hs / vs                         */
  int32_t mcus_x, mcus_y;         /* coded MCUs per row / column                                 */
  int32_t blocks_per_mcu;         /* hs*vs + 2                                                   */
  int32_t reserved;
  int64_t n_coded_blocks;         /* mcus_x*mcus_y*blocks_per_mcu                                */
  int64_t coef_bytes;             /* n_coded_blocks * 128                                        */
  int64_t rgb_bytes;              /* width*height*3 (tight rows)                                 */
} jb_geometry;

typedef struct jb_ctx jb_ctx;

/* ---- library / context --------------------------------------------------------------- */
int jb_abi_version(void);
/* Number of HIP devices visible, or a negative jb_status. */
int jb_device_count(void);
/* Validate a descriptor and derive its sizes.  Pure host code, no device needed. */
int jb_geometry_of(const jb_image_desc *desc, jb_geometry *out);
/* Create a context on `device_id` with its own stream.  `max_coef_bytes`/`max_rgb_bytes`
 * size the per-slot device and pinned staging buffers used by the host-buffer entry points
 * (0,0: device-pointer entry points only).  `n_slots` (1..64) = depth of the staging ring
 * (each slot holds one image's coefficients and pixels in device memory). */
int jb_ctx_create(int device_id, size_t max_coef_bytes, size_t max_rgb_bytes, int n_slots,
                  jb_ctx **out);
void jb_ctx_destroy(jb_ctx *ctx);
/* Grow the staging ring of a context to images of up to (max_coef_bytes, max_rgb_bytes); a
 * context created with (0,0) gets its ring (of the depth given at creation) here.  Waits for
 * everything in flight first; never shrinks.  jb_decode_file / jb_decode_memory call this with the
 * parsed frame's sizes, so a context need not know its largest image in advance (the reference's
 * Image allocates `new MCU[...]` per file from the SOF sizes, jpeg.cpp:407). */
int jb_ctx_reserve(jb_ctx *ctx, size_t max_coef_bytes, size_t max_rgb_bytes);
/* The HIP device a context lives on. */
int jb_ctx_device(const jb_ctx *ctx);
/* Text of the last error on this context (or of the last context-less error on this thread
 * when ctx is NULL).  Never NULL. */
const char *jb_last_error(const jb_ctx *ctx);
/* The context's primary HIP stream (hipStream_t): what device-resident launches with a NULL
 * stream argument run on, so callers can order their own work against them.  (The staging ring
 * of jb_submit uploads and computes on this stream and downloads on a second one, so that the
 * device->host copy of image i overlaps the host->device copy of image i+1; order against the
 * downloads with jb_wait.) */
void *jb_ctx_stream(jb_ctx *ctx);
/* Block until everything submitted to the context (either stream) has finished. */
int jb_ctx_synchronize(jb_ctx *ctx);

/* ---- the seam: host buffers (drop-in for jpeg.cpp:786-788) ----------------------------- */
/* Synchronous: copies coefficients to the device, runs the fused kernel, copies pixels back.
 * `qtabs` = 4*64 uint16 natural order (tables not referenced by desc->qtab_id may be 0). */
int jb_blocks_to_rgb(jb_ctx *ctx, const jb_image_desc *desc, const int16_t *coef,
                     const uint16_t *qtabs, uint8_t *rgb, int64_t rgb_stride);
/* Asynchronous flavour over the staging ring, so the host Huffman stage of image i+1 overlaps
 * the device work of image i.  `coef`/`rgb` should come from jb_pinned_alloc for true overlap
 * and must stay valid until jb_wait(ticket) returns.  Blocks only when all slots are busy.
 * Submissions may complete out of order (small ones run on several stream pairs in turn): wait
 * for the ticket, or jb_ctx_synchronize for everything. */
int jb_submit(jb_ctx *ctx, const jb_image_desc *desc, const int16_t *coef,
              const uint16_t *qtabs, uint8_t *rgb, int64_t rgb_stride, int *ticket);
/* Several images of ONE geometry in one submission (one upload, one launch, one download): for
 * small images, where the per-submission cost (tens of microseconds of driver calls) would
 * otherwise bound the rate.  coef = n_images consecutive images (coef_bytes each), qtabs =
 * n_images x 4*64 uint16, rgb = n_images consecutive images with tightly packed rows.
 * n_images <= 256 and n_images x (coef_bytes, rgb_bytes) within the context's capacity. */
int jb_submit_batch(jb_ctx *ctx, const jb_image_desc *desc, int n_images, const int16_t *coef,
                    const uint16_t *qtabs, uint8_t *rgb, int *ticket);
int jb_wait(jb_ctx *ctx, int ticket);
/* Non-blocking jb_wait: JB_OK once the submission has completed, JB_PENDING (> 0, not an error)
 * while it is still in flight. */
#define JB_PENDING 1
int jb_poll(jb_ctx *ctx, int ticket);
/* Pinned host memory for the coefficient / pixel buffers handed to jb_submit (the reference's
 * `new MCU[...]`, jpeg.cpp:407, is the buffer this replaces).  Pinned memory is pinned against a
 * device and placed on the host NUMA node closest to it: jb_pinned_alloc_on names the device --
 * use it with the device of the context the buffer will be submitted to, above all from threads
 * that never called hipSetDevice (their current device is 0 whatever GPU the process drives).
 * jb_pinned_alloc = jb_pinned_alloc_on(the calling thread's current device). */
void *jb_pinned_alloc_on(int device_id, size_t bytes);
void *jb_pinned_alloc(size_t bytes);
void jb_pinned_free(void *p);
/* Host NUMA node closest to a device (>= 0), or a negative jb_status when unknown.  One process
 * per GPU (the reference decodes one image per process, jpeg.cpp:916-929): a rank keeps its
 * entropy threads and its staging on this node. */
int jb_device_numa_node(int device_id);

/* ---- the seam: device-resident buffers (what bench.py and multi-image batches use) ------ */
/* A batch = n_images images of identical geometry, processed by ONE kernel launch on `stream`
 * (NULL = the context's stream).  Every pointer is a device pointer.  d_qtabs holds, per image
 * (or once, when qtab_image_stride == 0), three tables int32[3][64] already resolved per
 * component (Y, Cb, Cr) in natural order; build them with jb_resolve_qtabs(). */
typedef struct jb_device_batch {
  jb_image_desc desc;
  int32_t n_images;
  int32_t reserved;
  const int16_t *d_coef;
  int64_t coef_image_stride; /* bytes between images, multiple of 16 */
  const int32_t *d_qtabs;
  int64_t qtab_image_stride; /* bytes between images' [3][64] tables; 0 = shared */
  uint8_t *d_rgb;
  int64_t rgb_image_stride; /* bytes between images */
  int64_t rgb_row_stride;   /* bytes between pixel rows, >= 3*width; any value works (lanes store 12 bytes
                             * at byte-aligned addresses), multiples of 64 are fastest: on small images
                             * a tightly packed odd stride costs about ten points of roofline (partial
                             * lines at the ends of every 768-byte wave store) */
} jb_device_batch;

int jb_blocks_to_rgb_device(jb_ctx *ctx, const jb_device_batch *batch, void *stream);
/* Host helper: expand (qtabs uint16[4][64], qtab_id[3]) into the int32[3][64] the kernel reads. */
int jb_resolve_qtabs(const jb_image_desc *desc, const uint16_t *qtabs, int32_t *out192);
/* Name of the kernel jb_blocks_to_rgb_device launches for this descriptor (for profilers): the 192 / 256-lane
 * kernel of the layout.  Launches of up to 8 of its workgroups per CU (one to four 1080p images, one 4096x4096
 * 4:2:0) run as jb_small_kernel_444 / _420 / _16<2,1> / _16<1,2> instead (JPEGBLK_SMALL_GRID). */
const char *jb_kernel_name(const jb_image_desc *desc);

/* ---- host front end ("next" rows of the scope table; reference jpeg.cpp:67-446, 826-907,
 *      include/file.hpp, include/huffman.hpp) --------------------------------------------- */
/* Parse a JFIF byte stream and Huffman-decode it into packed int16 blocks in the order
 * described above.  Two-call protocol: with coef == NULL only the headers are parsed and
 * *desc / qtabs are filled (size the buffer with jb_geometry_of); with coef != NULL (capacity
 * coef_cap_bytes) the entropy-coded data is decoded too.
 * What the reference accepts -- baseline, three components, one interleaved scan -- takes the
 * fast path and is integer-exact against the reference's decodeHuffman().  Beyond the reference
 * (it rejects them, jpeg.cpp:69-87, 255-264): progressive frames (SOF2), frames coded in several
 * scans, and grayscale frames, which are delivered as 4:4:4 with all-zero Cb/Cr blocks so that
 * the pixel path yields R = G = B.  Still rejected: luma factors outside {1,2}, chroma not 1x1
 * (JB_ERR_SAMPLING); 12-bit, lossless, hierarchical, arithmetic-coded, 2- or 4-component frames
 * (JB_ERR_UNSUPPORTED). */
int jb_entropy_decode(const uint8_t *jpeg, size_t jpeg_bytes, jb_image_desc *desc,
                      uint16_t *qtabs /* 4*64 */, int16_t *coef, size_t coef_cap_bytes);
/* The same with the restart intervals of ONE image (DRI; e.g. the reference's images/img4.jpg)
 * decoded by n_threads host threads: intervals are independent because the DC predictors reset
 * at every restart (reference jpeg.cpp:419-425).  Images without restart markers, or with
 * markers that do not match the frame, take the serial path.  Output is identical. */
int jb_entropy_decode_mt(const uint8_t *jpeg, size_t jpeg_bytes, jb_image_desc *desc,
                         uint16_t *qtabs /* 4*64 */, int16_t *coef, size_t coef_cap_bytes,
                         int n_threads);
/* The entropy stage ON THE DEVICE (beyond the reference, whose decodeHuffman() -- jpeg.cpp:405-446 --
 * is serial host code): the host only parses the headers and removes the byte stuffing.  Every restart
 * interval of the scan (the DC predictors reset at each restart, jpeg.cpp:419-425, so intervals are
 * independent; a scan without DRI is one interval) is cut into chunks of 128 bytes, one GPU lane per chunk:
 * the lanes of a workgroup fall into step with the true symbol sequence in a few passes over LDS
 * (self-synchronising decoding), a prefix sum gives every chunk its block index, a writing pass decodes
 * every chunk from its neighbour's final state, stores the coefficients and verifies the chain of chunk
 * states, and a last pass turns the DC differences into DC values (csrc/jb_huff.hip).  The coefficients land
 * in the layout described above.  d_coef is a DEVICE pointer (16-byte aligned, capacity coef_cap_bytes); the
 * result is integer-exact with jb_entropy_decode.  Synchronous.  Takes baseline frames of three components
 * (any of the four sampling layouts) or one component, with up to three Huffman tables of each kind.
 * JB_ERR_UNSUPPORTED: a valid stream this decoder does not take (restart markers that do not match the
 * frame, progressive or multi-scan files, Huffman tables with more long codes than its lookup tables
 * hold) -- use jb_entropy_decode.  JB_ERR_FORMAT: corrupt data, or chunks that did not fall into step
 * (dense adversarial data; one retry with more launches is made first) -- jb_entropy_decode is the authority.
 * The BATCH decoders (jb_decode_batch, jb_batch_decoder_*) take this path by default for every image it
 * accepts (16 chunks = 2 KB of scan or more) and fall back to the host decoder per image for whatever it
 * does not take or flags; JPEGBLK_GPU_HUFFMAN=0 keeps the entropy stage on the host threads (north_star's
 * split), =2 drops the size threshold.  The single-image jb_decode_file / jb_decode_memory take it for files
 * with 128 KB of entropy-coded data or more, where one image's latency is lower on the device (1920x1080
 * 4:4:4: 0.51 ms against 3.9 ms on one host core; 679x451: 0.75-0.95 against 0.7), with
 * JPEGBLK_GPU_HUFFMAN=1 or 2 for every file the device decoder takes, with =0 never. */
int jb_entropy_decode_device(jb_ctx *ctx, const uint8_t *jpeg, size_t jpeg_bytes, jb_image_desc *desc,
                             uint16_t *qtabs /* 4*64, may be NULL */, int16_t *d_coef, size_t coef_cap_bytes);
/* How many images this context has decoded with the entropy stage on the device (through
 * jb_decode_file / jb_decode_memory / a batch decoder): lets callers and tests see which path ran. */
long long jb_ctx_device_entropy_images(const jb_ctx *ctx);
/* decode(path) -> RGB: the reference's whole `Image(path); readJPEG();` surface
 * (jpeg.cpp:797-807, 826-907) minus the X11 sink.  *rgb is malloc'd (tight rows, width*3);
 * release it with jb_free(). */
int jb_decode_file(jb_ctx *ctx, const char *path, uint8_t **rgb, int32_t *width, int32_t *height);
int jb_decode_memory(jb_ctx *ctx, const uint8_t *jpeg, size_t jpeg_bytes, uint8_t **rgb,
                     int32_t *width, int32_t *height);
void jb_free(void *p);
/* Frame descriptor (sampling factors, table ids) of the last image jb_decode_file / jb_decode_memory
 * decoded on this context -- with jb_geometry_of it yields the reference's public fields
 * mcuWidthReal / mcuHeightReal (jpeg.cpp:794-795, computed at :118-125). */
int jb_ctx_last_desc(const jb_ctx *ctx, jb_image_desc *out);

/* Batch of files: the multi-image form of decode(path) (BASELINE.json configs 4-5).  `n_threads`
 * host threads share one context on `device_id` (pinned staging ring) and walk the files
 * i = t, t + n_threads, ...  With JPEGBLK_GPU_HUFFMAN=0 -- north_star's split, "host Huffman on all
 * cores overlapped with device IDCT" -- a thread parses and Huffman-decodes image i into pinned
 * memory, submits it, and collects image i-1 while the device works, so the entropy stage of one
 * image overlaps the copies and the kernel of another, within a thread and across threads.  By
 * default the threads only parse, remove the byte stuffing and pack, and the entropy stage runs on
 * the device too (jb_entropy_decode_device above).  Per file: rgb[i] (malloc'd, tight rows; NULL on
 * failure, release with jb_free), widths[i], heights[i], statuses[i] (a jb_status).  `times`
 * (optional, 4 doubles) receives seconds: wall, summed entropy-decode, summed submit+wait,
 * summed file read.  Returns JB_OK when every file decoded, else the first failing status. */
int jb_decode_batch(int device_id, const char *const *paths, int n_paths, int n_threads,
                    uint8_t **rgb, int32_t *widths, int32_t *heights, int *statuses, double *times);
/* The same with the shared context and the per-thread pinned buffers kept across runs (creating
 * them -- page pinning above all -- costs milliseconds per thread).  n_threads is capped at the
 * CPUs the process may use (affinity mask and cgroup CPU quota): more entropy threads than that
 * only slow the batch down.  max_*_bytes pre-size every thread's
 * staging (0,0: sized lazily by the first run); a later run with larger images re-sizes. */
typedef struct jb_batch_decoder jb_batch_decoder;
int jb_batch_decoder_create(int device_id, int n_threads, size_t max_coef_bytes, size_t max_rgb_bytes,
                            jb_batch_decoder **out);
int jb_batch_decoder_run(jb_batch_decoder *dec, const char *const *paths, int n_paths, uint8_t **rgb,
                         int32_t *widths, int32_t *heights, int *statuses, double *times);
void jb_batch_decoder_destroy(jb_batch_decoder *dec);
/* Batches in a stream: submit returns at once and the batch runs on the decoder's own host threads; collect
 * waits for it and returns what jb_batch_decoder_run would have (status, and `times` if given).  Up to TWO
 * batches are in flight -- a third submit is refused with JB_ERR_STATE until the older one has been collected --
 * so that the start-up of batch k+1 (reading headers, the first groups' entropy stage and uploads) runs under
 * the tail of batch k (its last kernels and downloads): the seam the reference leaves synchronous per image
 * (jpeg.cpp:785-788) overlapped per batch as well.  Batches alternate between two sides, each with its own
 * host threads, staging and ring on the same device(s); the second side is built by the first submit (tens of
 * milliseconds, as creating a decoder).  Outputs: without an arena as for run (malloc'ed, jb_free); with a
 * pinned arena each side has one of the size given to jb_batch_decoder_set_arena, and with device regions side 0
 * writes into the first half of every region and side 1 into the second -- rgb[i] of batch k stay valid until
 * batch k+2 is submitted.  rgb / widths / heights / statuses must stay valid until the batch is collected (the
 * path strings are copied by submit).  jb_batch_decoder_run / _set_arena / _set_device_output[s] are refused
 * with JB_ERR_STATE while a batch is in flight; jb_batch_decoder_destroy waits for batches still running.
 * One thread at a time calls submit / collect on a decoder. */
int jb_batch_decoder_submit(jb_batch_decoder *dec, const char *const *paths, int n_paths, uint8_t **rgb,
                            int32_t *widths, int32_t *heights, int *statuses, int *ticket);
int jb_batch_decoder_collect(jb_batch_decoder *dec, int ticket, double *times);
/* One decoder over SEVERAL devices of the node (BASELINE.json configs 4-5 as ONE call; images are
 * independent -- reference jpeg.cpp:574-589 touches each block on its own, jpeg.cpp:916-929 decodes
 * one image per process -- so file i goes to device_ids[i % n_devices], no data crosses devices).
 * Per listed device: one shared-context staging ring and an equal share of the n_threads host
 * threads, each bound to the CPUs of that device's NUMA node with its pinned staging allocated
 * there.  A device may be listed more than once (two rings on one GPU).  The handle is used with
 * jb_batch_decoder_run / _set_arena / _destroy like a single-device one; an arena is one pinned
 * allocation shared by all devices. */
int jb_batch_decoder_create_multi(const int *device_ids, int n_devices, int n_threads, size_t max_coef_bytes,
                                  size_t max_rgb_bytes, jb_batch_decoder **out);
/* Images this decoder's current context(s) decoded with the entropy stage on the device (files with
 * restart intervals, see jb_entropy_decode_device); the count restarts when a run re-sizes the ring. */
long long jb_batch_decoder_device_entropy_images(const jb_batch_decoder *dec);
/* Optional pinned output arena owned by the decoder (bytes = 0 releases it).  With an arena,
 * jb_batch_decoder_run places every decoded image in it -- rgb[i] points INTO the arena: do not
 * jb_free it; it stays valid until the next run, set_arena or destroy -- and the device writes the
 * pixels straight to their final place: no per-image allocation, no host copy.  An image that
 * does not fit fails with JB_ERR_CAPACITY.  Without an arena the pixels go through per-thread
 * pinned staging and are copied into malloc'ed buffers (release with jb_free). */
int jb_batch_decoder_set_arena(jb_batch_decoder *dec, size_t bytes);
/* Device-resident output: the decoded images stay in HBM.  `d_base` is DEVICE memory of the
 * decoder's device that the caller owns (hipMalloc, a torch tensor's storage; 256-byte aligned),
 * `bytes` its size; jb_batch_decoder_run then places every image in it like in an arena -- rgb[i]
 * is a DEVICE pointer into the region (tight rows, 3*width bytes each; images decoded in one group
 * lie back to back, so an image's address has no particular alignment), valid until the next run --
 * and the fused kernel writes the pixels straight there:
 * nothing is downloaded, which removes what bounds the host-output forms (the device-to-host link).
 * For consumers that work on the pixels on the GPU.  The run returns when every image is complete
 * in device memory.  A region that is not device memory of the decoder's device (a host pointer,
 * another GPU's memory, a size that reaches beyond the allocation) is refused with JB_ERR_GEOMETRY.
 * (NULL, 0) returns to host output.  A multi-device decoder takes one region per
 * device: jb_batch_decoder_set_device_outputs. */
int jb_batch_decoder_set_device_output(jb_batch_decoder *dec, void *d_base, size_t bytes);
/* The same for a multi-device decoder: one region per listed device, in the order given to
 * jb_batch_decoder_create_multi (d_bases[k] is memory of device_ids[k]); file i lands in the region of
 * device_ids[i % n_devices].  n = 0: host output again. */
int jb_batch_decoder_set_device_outputs(jb_batch_decoder *dec, void *const *d_bases, const size_t *bytes, int n);
/* Output sink replacing the reference's X11 window / unused BMP writer (display.hpp,
 * jpeg.cpp:462-509): binary PPM (P6). */
int jb_write_ppm(const char *path, const uint8_t *rgb, int32_t width, int32_t height,
                 int64_t rgb_stride);
/* Replaces Image::saveToBMP / writeBMP (jpeg.cpp:462-509, 809-816): an uncompressed 24-bit
 * Windows BMP (BITMAPINFOHEADER, bottom-up rows padded to 4 bytes, bytes in B,G,R order).  The
 * reference writes the 12-byte OS/2 core header (16-bit sizes) and puts the planes out as
 * R,B,G (jpeg.cpp:497-499) with a padding of width%4 bytes (jpeg.cpp:472), which viewers show
 * with swapped colours or skewed rows; this writer emits what viewers expect. */
int jb_write_bmp(const char *path, const uint8_t *rgb, int32_t width, int32_t height,
                 int64_t rgb_stride);

#ifdef __cplusplus
}
#endif
#endif /* JPEGBLK_H */
