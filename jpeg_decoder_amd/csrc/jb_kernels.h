// jb_kernels.h -- internal interface between the C-ABI layer (jb_api.cpp) and the HIP
// kernels (jb_kernels.hip).  Not part of the public ABI (include/jpegblk.h).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

// Kernel arguments of one launch: a batch of images of identical geometry.
struct JbLaunch {
  const int16_t *coef;        // device; decode-order int16 blocks
  const int32_t *qtabs;       // device; per image int32[3][64] (Y, Cb, Cr), natural order
  uint8_t *rgb;               // device; interleaved RGB
  int64_t coef_image_stride;  // bytes
  int64_t qtab_image_stride;  // bytes (0 = tables shared by the batch)
  int64_t rgb_image_stride;   // bytes
  int64_t rgb_row_stride;     // bytes
  int32_t width, height;      // pixels
  int32_t mcus_x, mcus_y;     // coded MCUs per row / column
  int32_t tiles_per_row;      // row-bound tiling: ceil(mcus_x / jbk_mcus_per_tile(hs, vs))
  int32_t tiles_per_image;    // linear: ceil(mcus_x*mcus_y / per_tile); row-bound: tiles_per_row*mcus_y
  int32_t n_tiles;            // n_images * tiles_per_image = workgroups launched
  int32_t linear;             // 1 = tiles follow the MCU stream, 0 = tiles are runs of one MCU row
  int32_t fast_store;         // 1: 12-byte stores (any byte alignment); 0: byte stores (JPEGBLK_BYTE_STORE=1)
  int32_t chroma_q_equal;     // 1 when Cb and Cr use the same table (desc.qtab_id[1] == qtab_id[2])
  int32_t reserved;           // 0 (777 = skip switch of the timing-experiment builds)
  int32_t staged;             // 1 (linear tiling only): the line-aligned store stage for rows that are not 64-byte aligned
  int32_t small_grid;         // 1: 4:4:4 and 4:2:0 only, one 64-lane workgroup per jbk_small_mcus() MCUs of an MCU row (row-bound)
};

// MCUs covered by one workgroup (a tile is always 192 coded blocks): 64 / 48 / 32.
int jbk_mcus_per_tile(int hs, int vs);
// MCUs per workgroup of the small-grid kernels (4:4:4: 16, 4:2:0: 8; 0: the layout has none)
int jbk_small_mcus(int hs, int vs);
// Can the layout use the linear (MCU-stream) tiling for an image with mcus_x MCUs per row?
int jbk_linear_ok(int hs, int vs, int mcus_x);
// Launch the fused kernel for luma sampling (hs, vs): one 192-lane workgroup per tile.
hipError_t jbk_launch(const JbLaunch &p, int hs, int vs, hipStream_t stream);
const char *jbk_kernel_name(int hs, int vs);

// Device-side entropy decoder (jb_huff.hip); structures in jb_huff.h.
struct JbHuffLaunch;
hipError_t jbk_huff_launch(const JbHuffLaunch &p, hipStream_t stream);
// small pinned host blob -> device by a kernel (no copy-engine hand-over in front of the decoding kernels);
// bytes is rounded up to 16: both buffers are allocated with that slack
// zero `bytes` (rounded up to 16) of device memory by a kernel
// (d_small / small_bytes: a second region of at most 4 KB, rounded up to 16 bytes, zeroed by the same launch)
hipError_t jbk_huff_zero(void *d_dst, size_t bytes, hipStream_t stream, void *d_small = nullptr, size_t small_bytes = 0);
hipError_t jbk_huff_fetch(void *d_dst, const void *h_pinned_src, size_t bytes, hipStream_t stream);
