// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Drives the GENUINE reference (aswanthabam/JPEG_Decoder) stage by stage.  The reference
// translation unit is #included from where it lies (REF_JPEG_CPP, normally
// /root/reference/jpeg.cpp); nothing of it is copied into this repository.  Built by
// oracle/Makefile into oracle/_ref/ (git-ignored) with -fno-access-control, because every
// hot-path member of `class Image` is private (reference jpeg.cpp:19-35), and without
// _FORTIFY_SOURCE, because hex_to_int overflows a 2-byte buffer (reference
// include/utils.hpp:16-23) and aborts under fortify.
//
// Two entry points:
//   * ref_decode_file()   replays Image::readJPEG's marker dispatch (reference
//                         jpeg.cpp:826-907) and process_image_data (jpeg.cpp:755-789) so the
//                         MCU array can be snapshotted after decodeHuffman() (coefficients)
//                         and after YCbCrToRGB() (pixels), with each stage timed.
//   * ref_blocks_to_rgb() runs the reference's own dequantize()/inverseDCT()/YCbCrToRGB()
//                         (jpeg.cpp:572-590, 735-753, 544-561) on caller-supplied
//                         coefficient blocks -- the oracle for synthetic block tests and the
//                         "reference" CPU baseline of bench.py.
//
// Coefficient exchange format = the C-ABI's (include/jpegblk.h): int16, natural order,
// MCU-interleaved decode order (per MCU: hs*vs luma blocks v-major, then Cb, then Cr), which
// is exactly the order decodeHuffman() visits blocks (jpeg.cpp:415-443).

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>

#define main ref_main__
#include REF_JPEG_CPP
#undef main

namespace {

struct CoutMute {
  std::streambuf *saved;
  CoutMute() : saved(std::cout.rdbuf(nullptr)) {}
  ~CoutMute() {
    std::cout.rdbuf(saved);
    std::cout.clear();
  }
};

double now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// visit blocks in the reference's scan order (jpeg.cpp:415-443) and call f(int* block64)
template <class F>
void for_each_coded_block(Image &img, F f) {
  for (int y = 0; y < img.mcuHeight; y += img.verticalSamplingFactor)
    for (int x = 0; x < img.mcuWidth; x += img.horizontalSamplingFactor)
      for (int i = 0; i < img.numComponents; ++i)
        for (int v = 0; v < img.color_components[i].vertical_sampling_factor; ++v)
          for (int h = 0; h < img.color_components[i].horizontal_sampling_factor; ++h)
            f(img.mcus[(y + v) * img.mcuWidthReal + (x + h)][i]);
}

void extract_rgb(Image &img, uint8_t *rgb, long stride) {
  // linearisation used by both reference sinks (display.hpp:19-34, jpeg.cpp:488-499)
  for (int y = 0; y < img.image_height; ++y)
    for (int x = 0; x < img.image_width; ++x) {
      const MCU &m = img.mcus[(y / 8) * img.mcuWidthReal + x / 8];
      const int p = (y % 8) * 8 + x % 8;
      uint8_t *o = rgb + (long)y * stride + 3L * x;
      o[0] = (uint8_t)m.r[p];
      o[1] = (uint8_t)m.g[p];
      o[2] = (uint8_t)m.b[p];
    }
}

const char *two_byte_soi_file() {
  static char path[64] = {0};
  if (!path[0]) {
    snprintf(path, sizeof path, "/tmp/jpegref_soi_%d.bin", (int)getpid());
    FILE *f = fopen(path, "wb");
    if (!f) return nullptr;
    fputc(0xff, f);
    fputc(0xd8, f);
    fclose(f);
  }
  return path;
}

}  // namespace

extern "C" {

struct RefInfo {
  int width, height, hs, vs;
  int mcu_w, mcu_h, mcu_w_real, mcu_h_real;
  int restart_interval;
  int qtab_id[3];
  int n_coded_blocks;
  int coef_min, coef_max;
  double ms_huffman, ms_dequant, ms_idct, ms_colour, ms_parse;
};

// Decode `path` with the reference front end + hot path.
//  coef_out : if non-null, capacity coef_cap int16 values; receives n_coded_blocks*64.
//  qtabs_out: if non-null, 4*64 uint16, natural order (0 where a table is absent).
//  rgb_out  : if non-null, capacity rgb_cap bytes; receives width*height*3 (tight stride).
// Returns 0, or -1 if no SOS was reached, -2 capacity.  NOTE: the reference calls exit(1) on
// unsupported input (progressive, !=3 components ...): call this from a child process when
// the input may be rejected.
int ref_decode_file(const char *path, RefInfo *info, int16_t *coef_out, long coef_cap,
                    uint16_t *qtabs_out, uint8_t *rgb_out, long rgb_cap) {
  CoutMute mute;
  double t0 = now_ms();
  Image *img = new Image(path);
  int rc = -1;
  while (true) {
    if (img->file->eof()) break;
    unsigned char *b = img->file->read(1);
    if (b[0] != 0xff) break;
    Marker *m = new Marker(img->file);
    if (m->type == MarkerType::SOF) img->read_sof(m);
    else if (m->type == MarkerType::DHT) img->read_huffman_table(m);
    else if (m->type == MarkerType::DQT) img->read_quantization_table(m);
    else if (m->type == MarkerType::DRI) img->read_dri(m);
    else if (m->type == MarkerType::PAD) continue;
    else if (m->type == MarkerType::EOI || m->type == MarkerType::INVALID) break;
    else if (m->type == MarkerType::SOS) {
      img->read_sos(m);
      double t1 = now_ms();
      img->mcus = img->decodeHuffman();
      double t2 = now_ms();
      info->width = img->image_width;
      info->height = img->image_height;
      info->hs = img->horizontalSamplingFactor;
      info->vs = img->verticalSamplingFactor;
      info->mcu_w = img->mcuWidth;
      info->mcu_h = img->mcuHeight;
      info->mcu_w_real = img->mcuWidthReal;
      info->mcu_h_real = img->mcuHeightReal;
      info->restart_interval = img->restartInterval;
      for (int i = 0; i < 3; i++) info->qtab_id[i] = img->color_components[i].quantizationTableID;
      long n = 0;
      int lo = 0, hi = 0;
      bool overflow = false;
      for_each_coded_block(*img, [&](int *blk) {
        for (int k = 0; k < 64; k++) {
          if (blk[k] < lo) lo = blk[k];
          if (blk[k] > hi) hi = blk[k];
          if (coef_out) {
            if ((n + 1) * 64 > coef_cap) overflow = true;
            else coef_out[n * 64 + k] = (int16_t)blk[k];
          }
        }
        n++;
      });
      info->n_coded_blocks = (int)n;
      info->coef_min = lo;
      info->coef_max = hi;
      if (qtabs_out) {
        for (int t = 0; t < 4; t++)
          for (int k = 0; k < 64; k++) {
            auto it = img->quantization_tables.find(t);
            qtabs_out[t * 64 + k] = (it == img->quantization_tables.end()) ? 0 : (uint16_t)(*it->second)[k];
          }
      }
      double t3 = now_ms();
      img->dequantize();
      double t4 = now_ms();
      img->inverseDCT();
      double t5 = now_ms();
      img->YCbCrToRGB();
      double t6 = now_ms();
      info->ms_parse = t1 - t0;
      info->ms_huffman = t2 - t1;
      info->ms_dequant = t4 - t3;
      info->ms_idct = t5 - t4;
      info->ms_colour = t6 - t5;
      rc = 0;
      if (rgb_out) {
        if ((long)info->width * info->height * 3 > rgb_cap) overflow = true;
        else extract_rgb(*img, rgb_out, 3L * info->width);
      }
      if (overflow || lo < -32768 || hi > 32767) rc = -2;
      break;
    }
    // APP/META/DRM: ignored, as readJPEG does (jpeg.cpp:868-877)
  }
  delete img;
  return rc;
}

// The reference hot path on caller-supplied blocks.  qtabs: 4 x 64 uint16 natural order
// (values must be <= 255: the reference keeps only a byte per entry, jpeg.cpp:216,223;
// types.hpp:86-92).  stage_ms (optional) receives {dequant, idct, colour} wall times.
int ref_blocks_to_rgb(int width, int height, int hs, int vs, const int16_t *coef,
                      const uint16_t *qtabs, const int *qtab_id, uint8_t *rgb, long stride,
                      double *stage_ms) {
  const char *soi = two_byte_soi_file();
  if (!soi) return -1;
  CoutMute mute;
  Image *img = new Image(soi);
  img->image_width = width;
  img->image_height = height;
  // geometry exactly as read_sof derives it (jpeg.cpp:77-80, 118-127)
  img->mcuWidth = (width + 7) / 8;
  img->mcuHeight = (height + 7) / 8;
  img->mcuWidthReal = img->mcuWidth;
  img->mcuHeightReal = img->mcuHeight;
  if (hs == 2 && img->mcuWidth % 2 == 1) img->mcuWidthReal += 1;
  if (vs == 2 && img->mcuHeight % 2 == 1) img->mcuHeightReal += 1;
  img->horizontalSamplingFactor = hs;
  img->verticalSamplingFactor = vs;
  img->numComponents = 3;
  for (int i = 0; i < 3; i++) {
    img->color_components[i].horizontal_sampling_factor = (i == 0) ? hs : 1;
    img->color_components[i].vertical_sampling_factor = (i == 0) ? vs : 1;
    img->color_components[i].quantizationTableID = qtab_id[i];
    img->color_components[i].set = true;
  }
  for (int t = 0; t < 4; t++) {
    unsigned char raw[64];
    for (int k = 0; k < 64; k++) {
      if (qtabs[t * 64 + zigZagMap[k]] > 255) {
        delete img;
        return -3;
      }
      raw[k] = (unsigned char)qtabs[t * 64 + zigZagMap[k]];  // ctor de-zigzags (types.hpp:88-90)
    }
    img->quantization_tables[t] = new QuantizationTable(t, raw);
    img->quantization_tables[t]->set = true;
  }
  img->mcus = new MCU[(long)img->mcuWidthReal * img->mcuHeightReal];
  long n = 0;
  for_each_coded_block(*img, [&](int *blk) {
    for (int k = 0; k < 64; k++) blk[k] = coef[n * 64 + k];
    n++;
  });
  double t0 = now_ms();
  img->dequantize();
  double t1 = now_ms();
  img->inverseDCT();
  double t2 = now_ms();
  img->YCbCrToRGB();
  double t3 = now_ms();
  if (stage_ms) {
    stage_ms[0] = t1 - t0;
    stage_ms[1] = t2 - t1;
    stage_ms[2] = t3 - t2;
  }
  extract_rgb(*img, rgb, stride);
  delete[] img->mcus;
  img->mcus = nullptr;
  for (int t = 0; t < 4; t++) delete img->quantization_tables[t];
  delete img;
  return 0;
}

// f32 bit patterns of the reference's static-init constants (types.hpp:5-19) and the colour
// literals (jpeg.cpp:521-523): order m0..m5, s0..s7, 1.402f, 0.344f, 0.714f, 1.772f.
void ref_constants(uint32_t *out18) {
  const float c[18] = {m0, m1, m2, m3, m4, m5, s0, s1, s2, s3, s4, s5, s6, s7,
                       1.402f, 0.344f, 0.714f, 1.772f};
  for (int i = 0; i < 18; i++) memcpy(&out18[i], &c[i], 4);
}

}  // extern "C"
