// jb_geometry.cpp -- the host-only arithmetic of the ABI (include/jpegblk.h): frame geometry and
// quantisation-table resolution.  No HIP dependency, so the front end can be built and fuzzed on
// a CPU-only toolchain with sanitizers (tools/fuzz/).
#include <cstring>

#include "../../include/jpegblk.h"

extern "C" {

// read_sof's derivations, reference jpeg.cpp:77-80 (block counts) and 110-127 (sampling checks,
// padding of odd block counts when the luma factor is 2)
int jb_geometry_of(const jb_image_desc *d, jb_geometry *g) {
  if (!d || !g) return JB_ERR_NULL;
  if (d->width < 1 || d->height < 1 || d->width > 65535 || d->height > 65535) return JB_ERR_GEOMETRY;
  if ((d->hs != 1 && d->hs != 2) || (d->vs != 1 && d->vs != 2)) return JB_ERR_SAMPLING;
  for (int i = 0; i < 3; i++)
    if (d->qtab_id[i] < 0 || d->qtab_id[i] > 3) return JB_ERR_QTAB;
  memset(g, 0, sizeof *g);
  g->mcu_w = (d->width + 7) / 8;
  g->mcu_h = (d->height + 7) / 8;
  g->mcu_w_real = g->mcu_w + ((d->hs == 2 && (g->mcu_w & 1)) ? 1 : 0);
  g->mcu_h_real = g->mcu_h + ((d->vs == 2 && (g->mcu_h & 1)) ? 1 : 0);
  g->mcus_x = g->mcu_w_real / d->hs;
  g->mcus_y = g->mcu_h_real / d->vs;
  g->blocks_per_mcu = d->hs * d->vs + 2;
  g->n_coded_blocks = (int64_t)g->mcus_x * g->mcus_y * g->blocks_per_mcu;
  g->coef_bytes = g->n_coded_blocks * 128;
  g->rgb_bytes = (int64_t)d->width * d->height * 3;
  return JB_OK;
}

int jb_resolve_qtabs(const jb_image_desc *d, const uint16_t *qtabs, int32_t *out192) {
  if (!d || !qtabs || !out192) return JB_ERR_NULL;
  for (int c = 0; c < 3; c++) {
    if (d->qtab_id[c] < 0 || d->qtab_id[c] > 3) return JB_ERR_QTAB;
    // the table each component names (reference jpeg.cpp:584), natural order
    for (int i = 0; i < 64; i++) out192[c * 64 + i] = qtabs[d->qtab_id[c] * 64 + i];
  }
  return JB_OK;
}

}  // extern "C"
