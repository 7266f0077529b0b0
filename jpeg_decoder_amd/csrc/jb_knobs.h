// jb_knobs.h -- every environment variable the library reads, and the ONE place each is read.
// A knob belongs to an object: it is read when a context (jb_ctx_create) or a batch decoder
// (jb_batch_decoder_create*, jb_decode_batch) is created and kept there, so that one object behaves
// the same from its first call to its last whatever the environment does meanwhile.  The table of
// include/jpegblk.h ("Environment") is this list.  None of them is needed for normal use: they
// select a path for tests and A/B measurements, or adapt the host side to its machine.
//
//   JPEGBLK_GPU_HUFFMAN   where the entropy stage runs.  unset: batch decoders on the device for every file with
//                         16 chunks or more, single images (decode(path)) from 128 KB of scan on; 0: always on the
//                         host threads (north_star's split); 1: the device for every file with 16 chunks or more;
//                         2: the device for every file it takes
//   JPEGBLK_CHUNK_BYTES   64 | 128: bytes of scan per lane of the device entropy decoder (default 128)
//   JPEGBLK_BYTE_STORE    1: every pixel through byte stores (the path odd widths took before the 12-byte
//                         stores at byte-aligned addresses; kept as the second implementation tests compare)
//   JPEGBLK_ROW_TILING    1: the row-bound tiling for every image (default: the linear tiling where rows are ragged)
//   JPEGBLK_STAGED_STORE  1: (measurement builds of jb_kernels.hip with -DJB_LAB only; the product ignores it) the staged,
//                         line-aligned store stage for every image that takes the linear tiling
//   JPEGBLK_SMALL_GRID    1 = always the one-wave kernels (every layout has one), 0 = never (default: for launches
//                         of up to 8 workgroups per CU of the 192-lane kernel, e.g. one to four 1080p images, one 4096x4096 4:2:0)
//   JPEGBLK_PASS1         1: a batch run always reads every file's headers first (default: only a decoder whose
//                         buffers do not exist yet does; otherwise files are parsed as their groups are formed)
//   JPEGBLK_GROUP_RAMP    1: a thread's first two device groups are a quarter and a half of the full size (default: full)
//   JPEGBLK_GROUP_MB      MB of coefficients per group of small images decoded on the host threads (default 16; 0: one image per submission)
//   JPEGBLK_DEV_GROUP_MB  MB of coefficients per group whose entropy stage runs on the device (default 96)
//   JPEGBLK_NUMA          0: leave the host threads' CPU affinity alone; 1: bind them to the GPU's NUMA node even
//                         under a CPU quota (default: bind only when the process owns a node's worth of CPUs)
//   JPEGBLK_OVERSUBSCRIBE 1: allow more host threads than CPUs the process may use
//   JPEGBLK_TIMING        1 | 2 | 3: where one decode(bytes) / one device-entropy submission / one batch run spends its time, on stderr
//   JPEGBLK_HW_QUEUES     read when the LIBRARY IS LOADED (before HIP initialises, jb_api.cpp): hardware queues to
//                         ask the runtime for (GPU_MAX_HW_QUEUES; default 16, 0 = leave the runtime's default)
#pragma once
#include <stdint.h>
#include <stdlib.h>

struct JbKnobs {
  int gpu_huffman = -1;      // -1: unset
  uint32_t chunk_bytes = 0;  // 0: the default
  bool byte_store = false;
  bool row_tiling = false;
  bool pass1 = false;
  bool group_ramp = false;
  int small_grid = -1;       // -1: automatic
  int staged_store = 0;
  long group_mb = -1;        // -1: the default
  long dev_group_mb = -1;
  int numa = -1;             // -1: automatic, 0: off, 1: forced
  bool oversubscribe = false;
  int timing = 0;
};

inline JbKnobs jb_knobs_read() {
  JbKnobs k;
  auto flag = [](const char *name) {
    const char *e = getenv(name);
    return e && e[0] == '1';
  };
  if (const char *e = getenv("JPEGBLK_GPU_HUFFMAN"))
    if (e[0] >= '0' && e[0] <= '2') k.gpu_huffman = e[0] - '0';
  if (const char *e = getenv("JPEGBLK_CHUNK_BYTES")) {
    const int v = atoi(e);
    if (v == 64 || v == 128) k.chunk_bytes = (uint32_t)v;
  }
  k.byte_store = flag("JPEGBLK_BYTE_STORE");
  k.row_tiling = flag("JPEGBLK_ROW_TILING");
  k.pass1 = flag("JPEGBLK_PASS1");
  k.group_ramp = flag("JPEGBLK_GROUP_RAMP");
  if (const char *e = getenv("JPEGBLK_STAGED_STORE")) k.staged_store = e[0] == '1' ? 1 : 0;
  if (const char *e = getenv("JPEGBLK_SMALL_GRID")) k.small_grid = e[0] == '0' ? 0 : e[0] == '1' ? 1 : -1;
  if (const char *e = getenv("JPEGBLK_GROUP_MB")) k.group_mb = atol(e) < 0 ? 0 : atol(e);
  if (const char *e = getenv("JPEGBLK_DEV_GROUP_MB")) k.dev_group_mb = atol(e) < 0 ? 0 : atol(e);
  if (const char *e = getenv("JPEGBLK_NUMA")) k.numa = e[0] == '0' ? 0 : e[0] == '1' ? 1 : -1;
  k.oversubscribe = flag("JPEGBLK_OVERSUBSCRIBE");
  if (const char *e = getenv("JPEGBLK_TIMING")) k.timing = e[0] >= '1' && e[0] <= '3' ? e[0] - '0' : 0;
  return k;
}
