/* oracle/jpegblk_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Scalar CPU restatement of the reference's block pipeline, written from the numerics spec
 * (SURVEY.md section 8a), not from the reference text.  It is the checker the GPU path is
 * compared with and the "port" CPU baseline of bench.py; it is never linked into, imported by
 * or executed from the product library (jpeg_decoder_amd/csrc).  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may use it.
 *
 * PARITY PIN: tests/test_oracle.py checks this file byte-for-byte against
 *   (1) the genuine reference compiled in place (oracle/_ref/libjpegref.so, built by
 *       oracle/Makefile from /root/reference/jpeg.cpp) on the six bundled baseline images and
 *       on seeded synthetic blocks in all four sampling layouts, whenever oracle/_ref exists;
 *   (2) the committed golden fixtures under tests/golden/ (generated from that reference
 *       build by oracle/gen_golden.py), which travel to the GPU box.
 *
 * Build: gcc -O2 -ffp-contract=off (no -ffast-math, no -march=native): every float
 * multiply/add below must stay a separate IEEE-754 binary32 operation.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../include/jpegblk.h"

/* f32 constants by bit pattern.  Reference include/types.hpp:5-19 computes them in double at
 * static-init time and rounds to float (m2 = m0 - m5 and m4 = m0 + m5 in float); jpeg.cpp:521-523
 * holds the colour literals.  Patterns verified against the compiled reference by
 * ref_constants() (oracle/ref_harness.cpp) in tests/test_oracle.py. */
static float f32_bits(uint32_t u) {
  float f;
  memcpy(&f, &u, 4);
  return f;
}
#define K_M1 f32_bits(0x3FB504F3u)
#define K_M2 f32_bits(0x3F8A8BD4u)
#define K_M3 f32_bits(0x3FB504F3u)
#define K_M4 f32_bits(0x40273D74u)
#define K_M5 f32_bits(0x3F43EF15u)
#define K_S0 f32_bits(0x3EB504F3u)
#define K_S1 f32_bits(0x3EFB14BEu)
#define K_S2 f32_bits(0x3EEC835Eu)
#define K_S3 f32_bits(0x3ED4DB31u)
#define K_S4 f32_bits(0x3EB504F3u)
#define K_S5 f32_bits(0x3E8E39DAu)
#define K_S6 f32_bits(0x3E43EF15u)
#define K_S7 f32_bits(0x3DC7C5C2u)
#define K_CR_R f32_bits(0x3FB374BCu) /* 1.402f */
#define K_CB_G f32_bits(0x3EB020C5u) /* 0.344f */
#define K_CR_G f32_bits(0x3F36C8B4u) /* 0.714f */
#define K_CB_B f32_bits(0x3FE2D0E5u) /* 1.772f */

/* ---- geometry (reference read_sof, jpeg.cpp:77-80, 110-127) ---------------------------- */
int jbo_geometry_of(const jb_image_desc *d, jb_geometry *g) {
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

/* ---- dequantize one block: c[i] *= q[i], int32 (reference jpeg.cpp:563-569) -------------- */
void jbo_dequant_block(int32_t *blk, const int32_t *q) {
  for (int i = 0; i < 64; i++) blk[i] = blk[i] * q[i];
}

/* ---- one 1-D pass of the AAN network over 8 values with stride `st`
 *      (reference jpeg.cpp:598-662 column form, :666-730 row form).  Inputs are int32,
 *      converted to float; each output is a float truncated toward zero on the store. ------ */
static void idct_1d(int32_t *p, int st) {
  const float g0 = (float)p[0 * st] * K_S0;
  const float g1 = (float)p[4 * st] * K_S4;
  const float g2 = (float)p[2 * st] * K_S2;
  const float g3 = (float)p[6 * st] * K_S6;
  const float g4 = (float)p[5 * st] * K_S5;
  const float g5 = (float)p[1 * st] * K_S1;
  const float g6 = (float)p[7 * st] * K_S7;
  const float g7 = (float)p[3 * st] * K_S3;

  const float f4 = g4 - g7;
  const float f5 = g5 + g6;
  const float f6 = g5 - g6;
  const float f7 = g4 + g7;

  const float e2 = g2 - g3;
  const float e3 = g2 + g3;
  const float e5 = f5 - f7;
  const float e7 = f5 + f7;
  const float e8 = f4 + f6;

  const float d2 = e2 * K_M1;
  const float d4 = f4 * K_M2;
  const float d5 = e5 * K_M3;
  const float d6 = f6 * K_M4;
  const float d8 = e8 * K_M5;

  const float c0 = g0 + g1;
  const float c1 = g0 - g1;
  const float c2 = d2 - e3;
  const float c3 = e3;
  const float c4 = d4 + d8;
  const float c5 = d5 + e7;
  const float c6 = d6 - d8;
  const float c7 = e7;
  const float c8 = c5 - c6;

  const float b0 = c0 + c3;
  const float b1 = c1 + c2;
  const float b2 = c1 - c2;
  const float b3 = c0 - c3;
  const float b4 = c4 - c8;
  const float b5 = c8;
  const float b6 = c6 - c7;
  const float b7 = c7;

  p[0 * st] = (int32_t)(b0 + b7);
  p[1 * st] = (int32_t)(b1 + b6);
  p[2 * st] = (int32_t)(b2 + b5);
  p[3 * st] = (int32_t)(b3 + b4);
  p[4 * st] = (int32_t)(b3 - b4);
  p[5 * st] = (int32_t)(b2 - b5);
  p[6 * st] = (int32_t)(b1 - b6);
  p[7 * st] = (int32_t)(b0 - b7);
}

/* ---- 2-D IDCT of one block: 8 column passes, then 8 row passes (reference jpeg.cpp:594-732) */
void jbo_idct_block(int32_t *blk) {
  for (int i = 0; i < 8; i++) idct_1d(blk + i, 8);
  for (int i = 0; i < 8; i++) idct_1d(blk + 8 * i, 1);
}

/* ---- one pixel of the colour transform (reference jpeg.cpp:521-535) ---------------------- */
static inline void colour_pixel(int32_t y, int32_t cb, int32_t cr, uint8_t *o) {
  int32_t r = (int32_t)(((float)y + K_CR_R * (float)cr) + 128.0f);
  int32_t g = (int32_t)((((float)y - K_CB_G * (float)cb) - K_CR_G * (float)cr) + 128.0f);
  int32_t b = (int32_t)(((float)y + K_CB_B * (float)cb) + 128.0f);
  if (r < 0) r = 0;
  if (r > 255) r = 255;
  if (g < 0) g = 0;
  if (g > 255) g = 255;
  if (b < 0) b = 0;
  if (b > 255) b = 255;
  o[0] = (uint8_t)r;
  o[1] = (uint8_t)g;
  o[2] = (uint8_t)b;
}

/* ---- one MCU: all of its coded blocks -> cropped pixels.
 *      Block visiting order and chroma replication index as reference jpeg.cpp:574-589 and
 *      :511-520 (cbcr index = (y/vs + 4v)*8 + x/hs + 4h). ---------------------------------- */
static void mcu_to_rgb(const jb_image_desc *d, const int16_t *coef, const int32_t *q3,
                       int mx, int my, uint8_t *rgb, int64_t stride) {
  const int hs = d->hs, vs = d->vs;
  int32_t lum[4][64], cb[64], cr[64];
  const int ny = hs * vs;
  for (int b = 0; b < ny + 2; b++) {
    int32_t *dst = b < ny ? lum[b] : (b == ny ? cb : cr);
    const int32_t *q = q3 + 64 * (b < ny ? 0 : (b == ny ? 1 : 2));
    for (int i = 0; i < 64; i++) dst[i] = coef[b * 64 + i];
    jbo_dequant_block(dst, q);
    jbo_idct_block(dst);
  }
  for (int v = 0; v < vs; v++)
    for (int h = 0; h < hs; h++) {
      const int32_t *yb = lum[v * hs + h];
      const int x0 = (mx * hs + h) * 8, y0 = (my * vs + v) * 8;
      for (int y = 0; y < 8; y++) {
        if (y0 + y >= d->height) break;
        for (int x = 0; x < 8; x++) {
          if (x0 + x >= d->width) break;
          const int ci = (y / vs + 4 * v) * 8 + x / hs + 4 * h;
          colour_pixel(yb[y * 8 + x], cb[ci], cr[ci], rgb + (int64_t)(y0 + y) * stride + 3 * (x0 + x));
        }
      }
    }
}

int jbo_resolve_qtabs(const jb_image_desc *d, const uint16_t *qtabs, int32_t *out192) {
  if (!d || !qtabs || !out192) return JB_ERR_NULL;
  for (int c = 0; c < 3; c++) {
    if (d->qtab_id[c] < 0 || d->qtab_id[c] > 3) return JB_ERR_QTAB;
    for (int i = 0; i < 64; i++) out192[c * 64 + i] = qtabs[d->qtab_id[c] * 64 + i];
  }
  return JB_OK;
}

/* MCU rows [my0, my1) of one image */
static void rows_to_rgb(const jb_image_desc *d, const jb_geometry *g, const int16_t *coef,
                        const int32_t *q3, uint8_t *rgb, int64_t stride, int my0, int my1) {
  for (int my = my0; my < my1; my++)
    for (int mx = 0; mx < g->mcus_x; mx++)
      mcu_to_rgb(d, coef + ((int64_t)my * g->mcus_x + mx) * g->blocks_per_mcu * 64, q3, mx, my, rgb, stride);
}

/* ---- the seam, single thread (the reference is single-threaded): dequantize(); inverseDCT();
 *      YCbCrToRGB(); of reference jpeg.cpp:786-788 on one image ----------------------------- */
int jbo_blocks_to_rgb(const jb_image_desc *d, const int16_t *coef, const uint16_t *qtabs,
                      uint8_t *rgb, int64_t stride) {
  jb_geometry g;
  int32_t q3[192];
  if (!coef || !qtabs || !rgb) return JB_ERR_NULL;
  int rc = jbo_geometry_of(d, &g);
  if (rc) return rc;
  if (stride < 3LL * d->width) return JB_ERR_GEOMETRY;
  rc = jbo_resolve_qtabs(d, qtabs, q3);
  if (rc) return rc;
  rows_to_rgb(d, &g, coef, q3, rgb, stride, 0, g.mcus_y);
  return JB_OK;
}

/* ---- same, MCU rows split over `nthreads` host threads (all-cores baseline) -------------- */
typedef struct {
  const jb_image_desc *d;
  const jb_geometry *g;
  const int16_t *coef;
  const int32_t *q3;
  uint8_t *rgb;
  int64_t stride;
  int my0, my1;
} jbo_job;

static void *jbo_worker(void *p) {
  jbo_job *j = (jbo_job *)p;
  rows_to_rgb(j->d, j->g, j->coef, j->q3, j->rgb, j->stride, j->my0, j->my1);
  return NULL;
}

int jbo_blocks_to_rgb_mt(const jb_image_desc *d, const int16_t *coef, const uint16_t *qtabs,
                         uint8_t *rgb, int64_t stride, int nthreads) {
  jb_geometry g;
  int32_t q3[192];
  if (!coef || !qtabs || !rgb) return JB_ERR_NULL;
  int rc = jbo_geometry_of(d, &g);
  if (rc) return rc;
  if (stride < 3LL * d->width) return JB_ERR_GEOMETRY;
  rc = jbo_resolve_qtabs(d, qtabs, q3);
  if (rc) return rc;
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 256) nthreads = 256;
  if (nthreads > g.mcus_y) nthreads = g.mcus_y;
  pthread_t th[256];
  jbo_job jobs[256];
  for (int t = 0; t < nthreads; t++) {
    jobs[t] = (jbo_job){d, &g, coef, q3, rgb, stride, (int)((int64_t)g.mcus_y * t / nthreads),
                        (int)((int64_t)g.mcus_y * (t + 1) / nthreads)};
    if (pthread_create(&th[t], NULL, jbo_worker, &jobs[t]) != 0) {
      jbo_worker(&jobs[t]);
      th[t] = 0;
    }
  }
  for (int t = 0; t < nthreads; t++)
    if (th[t]) pthread_join(th[t], NULL);
  return JB_OK;
}

/* Timed repeat of the seam for bench.py's cpu_baseline leg: returns seconds for `reps` passes. */
double jbo_time_blocks_to_rgb(const jb_image_desc *d, const int16_t *coef, const uint16_t *qtabs,
                              uint8_t *rgb, int64_t stride, int nthreads, int reps) {
  struct timespec a, b;
  clock_gettime(CLOCK_MONOTONIC, &a);
  for (int r = 0; r < reps; r++) {
    int rc = nthreads <= 1 ? jbo_blocks_to_rgb(d, coef, qtabs, rgb, stride)
                           : jbo_blocks_to_rgb_mt(d, coef, qtabs, rgb, stride, nthreads);
    if (rc) return -1.0;
  }
  clock_gettime(CLOCK_MONOTONIC, &b);
  return (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
}

void jbo_constants(uint32_t *out18) {
  const float c[18] = {f32_bits(0x3FEC835Eu), K_M1, K_M2, K_M3, K_M4, K_M5, K_S0, K_S1, K_S2,
                       K_S3, K_S4, K_S5, K_S6, K_S7, K_CR_R, K_CB_G, K_CR_G, K_CB_B};
  memcpy(out18, c, sizeof c);
}
