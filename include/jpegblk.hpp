// jpegblk.hpp -- C++ host-side mirror of the reference's public surface over the C ABI
// (include/jpegblk.h).  Header-only; link with jpeg_decoder_amd/libjpegblk.so.
//
// The reference's whole public surface is `class Image` (jpeg.cpp:19-914): Image(string)
// (:797-807), readJPEG() (:826-907), display() (:818-824, X11), saveToBMP(string) (:809-816) and
// the fields image_width, image_height, mcuWidthReal, mcuHeightReal (:792-795).  This class keeps
// those names and meanings so code written against the reference reads the same, with two
// deliberate differences: errors throw jpegblk::Error (the reference logs and exit(1)s), and the
// decoded picture is a plain interleaved RGB buffer (`rgb()`), not the blocked MCU array.
#pragma once
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>

#include "jpegblk.h"

namespace jpegblk {

struct Error : std::runtime_error {
  int status;
  Error(int s, const std::string &what) : std::runtime_error(what), status(s) {}
};

// One HIP device + stream + staging ring (jb_ctx).  Share one Context between Images that are
// decoded from the same host thread; use one Context per thread / per GPU otherwise.  By default
// the staging ring is sized lazily from the first frame decoded (jb_ctx_reserve) and grows with
// larger ones -- as the reference sizes `new MCU[...]` per file (jpeg.cpp:407) -- so a thumbnail
// costs a thumbnail's worth of device memory and a 65535x65535 frame still decodes.
class Context {
 public:
  explicit Context(int device = 0, size_t max_coef_bytes = 0, size_t max_rgb_bytes = 0, int n_slots = 2) {
    int rc = jb_ctx_create(device, max_coef_bytes, max_rgb_bytes, n_slots, &ctx_);
    if (rc) throw Error(rc, jb_last_error(nullptr));
  }
  ~Context() { jb_ctx_destroy(ctx_); }
  Context(const Context &) = delete;
  Context &operator=(const Context &) = delete;
  jb_ctx *get() const { return ctx_; }

 private:
  jb_ctx *ctx_ = nullptr;
};

class Image {
 public:
  int image_width = 0;    // reference jpeg.cpp:792
  int image_height = 0;   // reference jpeg.cpp:793
  int mcuWidthReal = 0;   // reference jpeg.cpp:794
  int mcuHeightReal = 0;  // reference jpeg.cpp:795

  // reference: Image(string filename), jpeg.cpp:797-807 (opens the file, checks FF D8)
  Image(std::string filename, std::shared_ptr<Context> ctx = nullptr)
      : path_(std::move(filename)), ctx_(ctx ? std::move(ctx) : std::make_shared<Context>()) {}
  ~Image() { jb_free(rgb_); }
  Image(const Image &) = delete;
  Image &operator=(const Image &) = delete;

  // reference: readJPEG(), jpeg.cpp:826-907 -- marker loop, Huffman decode, then the three
  // passes dequantize(); inverseDCT(); YCbCrToRGB(); (jpeg.cpp:786-788), which here run as one
  // fused HIP kernel behind jb_blocks_to_rgb.
  void readJPEG() {
    jb_free(rgb_);
    rgb_ = nullptr;
    int32_t w = 0, h = 0;
    int rc = jb_decode_file(ctx_->get(), path_.c_str(), &rgb_, &w, &h);
    if (rc) throw Error(rc, jb_last_error(ctx_->get()));
    image_width = w;
    image_height = h;
    // (w+7)/8 rounded up to the luma sampling factor, as read_sof computes them (jpeg.cpp:118-125)
    jb_image_desc d;
    jb_geometry g;
    rc = jb_ctx_last_desc(ctx_->get(), &d);
    if (!rc) rc = jb_geometry_of(&d, &g);
    if (rc) throw Error(rc, jb_last_error(nullptr));
    mcuWidthReal = g.mcu_w_real;
    mcuHeightReal = g.mcu_h_real;
  }

  // The decoded picture: image_height rows of image_width R,G,B byte triples.  Equals
  // mcus[(y/8)*mcuWidthReal + x/8].{r,g,b}[(y%8)*8 + x%8] of the reference (display.hpp:19-34).
  const uint8_t *rgb() const { return rgb_; }

  // reference: saveToBMP(string), jpeg.cpp:809-816 (never called from its main; its writer puts
  // the planes out in R,B,G order behind an OS/2 core header -- jpeg.cpp:462-509).  Same name,
  // standard 24-bit BMP.
  void saveToBMP(const std::string &filename) const {
    int rc = jb_write_bmp(filename.c_str(), rgb_, image_width, image_height, 3LL * image_width);
    if (rc) throw Error(rc, jb_last_error(nullptr));
  }
  // binary PPM (P6): the sink used in place of the reference's X11 window (display.hpp)
  void savePPM(const std::string &filename) const {
    int rc = jb_write_ppm(filename.c_str(), rgb_, image_width, image_height, 3LL * image_width);
    if (rc) throw Error(rc, jb_last_error(nullptr));
  }

 private:
  std::string path_;
  std::shared_ptr<Context> ctx_;
  uint8_t *rgb_ = nullptr;
};

}  // namespace jpegblk
