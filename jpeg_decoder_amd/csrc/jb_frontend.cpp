// jb_frontend.cpp -- host front end: JFIF marker parser + baseline Huffman decoder.
//
// This is the part of the reference that STAYS on the host (reference jpeg.cpp:67-446 and
// 826-907, include/file.hpp, include/huffman.hpp).  It produces exactly what the reference's
// decodeHuffman() produces (jpeg.cpp:405-446): de-zigzagged integer coefficient blocks in
// MCU-interleaved scan order -- here packed as int16 straight into a caller buffer (pinned
// memory when it comes from jb_pinned_alloc) that is handed to the device seam.
//
// Written table-driven (9-bit lookahead + canonical fallback, 64-bit bit buffer) instead of
// the reference's bit-at-a-time linear code search (jpeg.cpp:300-320) and its per-byte heap
// allocation (file.hpp:26-52); the decoded integers are identical.
//
// This file is the fast path for what the reference accepts: baseline SOF0 (jpeg.cpp:69-73),
// exactly 3 components (jpeg.cpp:83-87), luma factors in {1,2}, chroma 1x1 (jpeg.cpp:110-136),
// table ids 0..3, one scan with Ss=0, Se=63, Ah=Al=0 (jpeg.cpp:255-264), APPn/COM ignored
// (file.hpp:201-207).  Files outside that (progressive, grayscale, several scans) are handed to
// the general front end, jb_frontend_ext.cpp.  Deliberate differences from the reference:
//  * errors are returned (jb_status), never exit(1);
//  * 16-bit quantisation tables keep all 16 bits (the reference keeps the low byte only,
//    jpeg.cpp:216 -- no bundled image has one);
//  * restart intervals are counted in MCUs as ITU-T T.81 says; the reference's test
//    (jpeg.cpp:414,419) agrees with that only when an interval is a whole number of MCU rows
//    (true for its bundled images/img4.jpg: interval 100 = one row).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../include/jpegblk.h"
#include "jb_entropy.h"
#include "jb_hostmem.h"
#include "jb_huff.h"
#include "jb_huff_core.h"
#include "jb_knobs.h"

struct jb_ctx;
int jb_fail_(jb_ctx *ctx, int code, const char *msg);
void jb_ctx_set_last_desc_(jb_ctx *ctx, const jb_image_desc *d);
const JbKnobs *jb_ctx_knobs_(const jb_ctx *ctx);  // jb_api.cpp
extern "C" int jb_decode_job_(jb_ctx *ctx, const JbHuffJob *job, uint8_t *rgb, int64_t rgb_stride);  // jb_api.cpp
// general front end (jb_frontend_ext.cpp): progressive, grayscale, multi-scan files
int jb_ext_decode_(const uint8_t *jpeg, size_t n, jb_image_desc *desc, uint16_t *qtabs, int16_t *coef,
                   size_t coef_cap_bytes, std::string *err);

namespace {

using namespace jbe;

struct Frame {
  jb_image_desc desc;
  bool have_sof = false;
  uint16_t qtabs[4][64];
  bool qset[4] = {false, false, false, false};
  HuffTable dc[4], ac[4];
  int ncomp = 3;                  // 1: a single-component frame (only jb_huff_prepare_ asks for those: allow_gray)
  int comp_id[3] = {0, 0, 0};
  int dc_id[3] = {0, 0, 0}, ac_id[3] = {0, 0, 0};
  int restart_interval = 0;
  const uint8_t *scan = nullptr;  // first entropy-coded byte
  size_t scan_len = 0;            // bytes up to the end of the buffer
};

int parse_headers(const uint8_t *d, size_t n, Frame &fr, Err &e, bool allow_gray = false) {
  if (!d || n < 4 || d[0] != 0xff || d[1] != 0xd8) return set_err(e, JB_ERR_FORMAT, "not a JPEG file (no SOI)");
  size_t pos = 2;
  memset(fr.qtabs, 0, sizeof fr.qtabs);
  while (true) {
    if (pos + 1 >= n) return set_err(e, JB_ERR_FORMAT, "truncated file (no SOS)");
    if (d[pos] != 0xff) return set_err(e, JB_ERR_FORMAT, "marker expected");
    while (pos < n && d[pos] == 0xff) pos++;
    if (pos >= n) return set_err(e, JB_ERR_FORMAT, "truncated file");
    const uint8_t m = d[pos++];
    if (m == 0xd8 || m == 0x01 || (m >= 0xd0 && m <= 0xd7)) continue;  // no payload
    if (m == 0xd9) return set_err(e, JB_ERR_FORMAT, "EOI before SOS");
    if (pos + 2 > n) return set_err(e, JB_ERR_FORMAT, "truncated segment");
    const size_t len = ((size_t)d[pos] << 8) | d[pos + 1];
    if (len < 2 || pos + len > n) return set_err(e, JB_ERR_FORMAT, "bad segment length");
    const uint8_t *s = d + pos + 2;
    const size_t sl = len - 2;
    pos += len;
    if (m == 0xdb) {  // DQT, reference jpeg.cpp:197-231
      size_t i = 0;
      while (i < sl) {
        const int pq = s[i] >> 4, tq = s[i] & 15;
        i++;
        if (tq > 3) return set_err(e, JB_ERR_QTAB, "quantisation table id > 3");
        const size_t need = pq ? 128 : 64;
        if (pq > 1 || i + need > sl) return set_err(e, JB_ERR_FORMAT, "bad DQT segment");
        for (int k = 0; k < 64; k++) {
          const uint16_t v = pq ? (uint16_t)((s[i + 2 * k] << 8) | s[i + 2 * k + 1]) : s[i + k];
          fr.qtabs[tq][kZigZag[k]] = v;  // stored de-zigzagged, reference types.hpp:88-90
        }
        fr.qset[tq] = true;
        i += need;
      }
    } else if (m == 0xc0) {  // SOF0, reference jpeg.cpp:67-146
      if (sl < 6) return set_err(e, JB_ERR_FORMAT, "bad SOF segment");
      if (s[0] != 8) return set_err(e, JB_ERR_UNSUPPORTED, "only 8-bit precision is supported");
      fr.desc.height = (s[1] << 8) | s[2];
      fr.desc.width = (s[3] << 8) | s[4];
      if (s[5] == 1 && allow_gray) {
        // one component (rejected by the reference, jpeg.cpp:83-87; the general front end delivers it as a 4:4:4 frame
        // whose Cb and Cr blocks are zero, jb_frontend_ext.cpp parse_sof): its sampling factors are irrelevant
        if (sl < 6 + 3) return set_err(e, JB_ERR_FORMAT, "bad SOF segment");
        const int h = s[7] >> 4, v = s[7] & 15, tq = s[8];
        if (tq > 3) return set_err(e, JB_ERR_QTAB, "quantisation table id > 3");
        if (h < 1 || h > 4 || v < 1 || v > 4) return set_err(e, JB_ERR_SAMPLING, "bad sampling factor");
        fr.ncomp = 1;
        fr.comp_id[0] = s[6];
        fr.desc.hs = fr.desc.vs = 1;
        fr.desc.qtab_id[0] = fr.desc.qtab_id[1] = fr.desc.qtab_id[2] = tq;
        fr.desc.reserved = 0;
        if (fr.desc.width < 1 || fr.desc.height < 1) return set_err(e, JB_ERR_GEOMETRY, "empty image");
        fr.have_sof = true;
        continue;
      }
      if (s[5] != 3) return set_err(e, JB_ERR_UNSUPPORTED, "only 3 components are supported");
      if (sl < 6 + 9) return set_err(e, JB_ERR_FORMAT, "bad SOF segment");
      for (int c = 0; c < 3; c++) {
        fr.comp_id[c] = s[6 + 3 * c];
        const int h = s[7 + 3 * c] >> 4, v = s[7 + 3 * c] & 15;
        const int tq = s[8 + 3 * c];
        if (tq > 3) return set_err(e, JB_ERR_QTAB, "quantisation table id > 3");
        fr.desc.qtab_id[c] = tq;
        if (c == 0) {
          if ((h != 1 && h != 2) || (v != 1 && v != 2)) return set_err(e, JB_ERR_SAMPLING, "luma sampling factors must be 1 or 2");
          fr.desc.hs = h;
          fr.desc.vs = v;
        } else if (h != 1 || v != 1) {
          return set_err(e, JB_ERR_SAMPLING, "chroma sampling factors must be 1x1");
        }
      }
      fr.desc.reserved = 0;
      if (fr.desc.width < 1 || fr.desc.height < 1) return set_err(e, JB_ERR_GEOMETRY, "empty image");
      fr.have_sof = true;
    } else if (m == 0xc2) {
      return set_err(e, JB_ERR_UNSUPPORTED, "progressive JPEG is not supported");  // jpeg.cpp:69-73
    } else if (m >= 0xc1 && m <= 0xcf && m != 0xc4 && m != 0xc8 && m != 0xcc) {
      return set_err(e, JB_ERR_UNSUPPORTED, "only baseline (SOF0) frames are supported");
    } else if (m == 0xc4) {  // DHT, reference jpeg.cpp:148-196
      size_t i = 0;
      while (i < sl) {
        if (i + 17 > sl) return set_err(e, JB_ERR_FORMAT, "bad DHT segment");
        const int tc = s[i] >> 4, th = s[i] & 15;
        if (th > 3 || tc > 1) return set_err(e, JB_ERR_FORMAT, "bad Huffman table id");
        HuffTable &t = tc ? fr.ac[th] : fr.dc[th];
        int total = 0;
        t.counts[0] = 0;
        for (int l = 1; l <= 16; l++) {
          t.counts[l] = s[i + l];
          total += s[i + l];
        }
        i += 17;
        if (total > 256 || i + total > sl) return set_err(e, JB_ERR_FORMAT, "bad DHT segment");
        memcpy(t.symbols, s + i, total);
        i += total;
        memset(t.symbols + total, 0, (size_t)(256 - total));  // (the cache compares whole arrays)
        if (!t.build_cached(tc != 0)) return set_err(e, JB_ERR_FORMAT, "over-subscribed Huffman table");
      }
    } else if (m == 0xdd) {  // DRI, reference jpeg.cpp:289-298
      if (sl != 2) return set_err(e, JB_ERR_FORMAT, "bad DRI segment");
      fr.restart_interval = (s[0] << 8) | s[1];
    } else if (m == 0xda) {  // SOS, reference jpeg.cpp:233-287
      if (!fr.have_sof) return set_err(e, JB_ERR_FORMAT, "SOS before SOF");
      if (fr.ncomp == 1) {
        if (sl < 1 || s[0] != 1) return set_err(e, JB_ERR_FORMAT, "bad number of scan components");
        if (sl != 1 + 2 + 3) return set_err(e, JB_ERR_FORMAT, "bad SOS length");
        if (s[1] != fr.comp_id[0]) return set_err(e, JB_ERR_FORMAT, "scan component not in the frame");
        fr.dc_id[0] = fr.dc_id[1] = fr.dc_id[2] = s[2] >> 4;
        fr.ac_id[0] = fr.ac_id[1] = fr.ac_id[2] = s[2] & 15;
        if (fr.dc_id[0] > 3 || fr.ac_id[0] > 3) return set_err(e, JB_ERR_FORMAT, "bad Huffman table id in SOS");
        if (s[3] != 0 || s[4] != 63 || s[5] != 0) return set_err(e, JB_ERR_UNSUPPORTED, "a sequential scan of all 64 coefficients only");
        if (!fr.qset[fr.desc.qtab_id[0]]) return set_err(e, JB_ERR_QTAB, "quantisation table not found");
        if (!fr.dc[fr.dc_id[0]].set) return set_err(e, JB_ERR_FORMAT, "Huffman DC table not found");
        if (!fr.ac[fr.ac_id[0]].set) return set_err(e, JB_ERR_FORMAT, "Huffman AC table not found");
        fr.scan = d + pos;
        fr.scan_len = n - pos;
        return JB_OK;
      }
      if (sl < 1 || s[0] != 3) return set_err(e, JB_ERR_UNSUPPORTED, "only 3-component scans are supported");
      if (sl != 1 + 6 + 3) return set_err(e, JB_ERR_FORMAT, "bad SOS length");
      for (int c = 0; c < 3; c++) {
        if (s[1 + 2 * c] != fr.comp_id[c]) return set_err(e, JB_ERR_UNSUPPORTED, "scan component order differs from frame");
        fr.dc_id[c] = s[2 + 2 * c] >> 4;
        fr.ac_id[c] = s[2 + 2 * c] & 15;
        if (fr.dc_id[c] > 3 || fr.ac_id[c] > 3) return set_err(e, JB_ERR_FORMAT, "bad Huffman table id in SOS");
      }
      if (s[7] != 0 || s[8] != 63) return set_err(e, JB_ERR_UNSUPPORTED, "spectral selection must be 0..63");
      if (s[9] != 0) return set_err(e, JB_ERR_UNSUPPORTED, "successive approximation must be 0");
      // process_image_data's table checks, reference jpeg.cpp:757-774
      for (int c = 0; c < 3; c++) {
        if (!fr.qset[fr.desc.qtab_id[c]]) return set_err(e, JB_ERR_QTAB, "quantisation table not found");
        if (!fr.dc[fr.dc_id[c]].set) return set_err(e, JB_ERR_FORMAT, "Huffman DC table not found");
        if (!fr.ac[fr.ac_id[c]].set) return set_err(e, JB_ERR_FORMAT, "Huffman AC table not found");
      }
      fr.scan = d + pos;
      fr.scan_len = n - pos;
      return JB_OK;
    }
    // APPn (E0-EF), COM (FE), anything else with a length: ignored (reference file.hpp:201-207)
  }
}

// reference decodeHuffman (jpeg.cpp:405-446): MCUs in raster order, per MCU hs*vs luma blocks
// (v-major), Cb, Cr; DC predictors reset at restart boundaries
int decode_scan(const Frame &fr, const jb_geometry &g, int16_t *coef, Err &e) {
  BitReader br(fr.scan, fr.scan + fr.scan_len);
  int pred[3] = {0, 0, 0};
  const int ny = fr.desc.hs * fr.desc.vs;
  const int64_t n_mcus = (int64_t)g.mcus_x * g.mcus_y;
  int until_restart = fr.restart_interval;
  int16_t *out = coef;
  for (int64_t m = 0; m < n_mcus; m++) {
    if (fr.restart_interval && until_restart == 0) {
      if (!br.restart()) return set_err(e, JB_ERR_FORMAT, "restart marker missing");
      pred[0] = pred[1] = pred[2] = 0;
      until_restart = fr.restart_interval;
    }
    for (int b = 0; b < ny + 2; b++) {
      const int c = b < ny ? 0 : b - ny + 1;
      if (!decode_block(br, fr.dc[fr.dc_id[c]], fr.ac[fr.ac_id[c]], pred[c], out))
        return set_err(e, JB_ERR_FORMAT, "corrupt entropy-coded data");
      out += 64;
    }
    until_restart--;
  }
  if (br.overran()) return set_err(e, JB_ERR_FORMAT, "entropy-coded data ends early");
  return JB_OK;
}

// One restart interval on its own: MCUs [m0, m1) from the clean bytes [b, b + len) that lay
// between two RSTn markers.  DC predictors start at 0 at every restart (T.81 F.2.1.3.1; reference
// jpeg.cpp:419-425), so intervals are independent.
int decode_interval(const Frame &fr, const jb_geometry &g, const uint8_t *b, size_t len, const uint8_t *hard_limit,
                    int64_t m0, int64_t m1, int16_t *coef) {
  CleanReader br(b);
  int pred[3] = {0, 0, 0};
  const int ny = fr.desc.hs * fr.desc.vs;
  int16_t *out = coef + m0 * g.blocks_per_mcu * 64;
  for (int64_t m = m0; m < m1; m++)
    for (int blk = 0; blk < ny + 2; blk++) {
      const int c = blk < ny ? 0 : blk - ny + 1;
      if (!decode_block_clean(br, fr.dc[fr.dc_id[c]], fr.ac[fr.ac_id[c]], pred[c], out)) return JB_ERR_FORMAT;
      if (br.p > hard_limit) return JB_ERR_FORMAT;  // ran off the end of the scan (the padding keeps the loads in bounds)
      out += 64;
    }
  // more bits consumed than the interval holds: truncated or corrupt data
  return br.consumed_bits(b) > (int64_t)len * 8 ? JB_ERR_FORMAT : JB_OK;
}

// The scan: de-stuffed once (unstuff), then its restart intervals decoded by n_threads host
// threads (1 = this thread).  When the RSTn markers found do not match the frame -- a marker
// missing, an extra one, markers in a file without DRI -- the bit-serial reader above takes over:
// it tolerates what the reference's reader tolerates and produces the precise error.
int decode_scan_mt(const Frame &fr, const jb_geometry &g, int16_t *coef, int n_threads, Err &e) {
  const int64_t n_mcus = (int64_t)g.mcus_x * g.mcus_y;
  const int ri = fr.restart_interval;
  const int64_t n_int = ri > 0 ? (n_mcus + ri - 1) / ri : 1;
  static thread_local CleanScan tls_scan;  // (kept per host thread: no allocation per image)
  CleanScan &cs = tls_scan;  // a plain reference: the worker threads below must see THIS thread's buffer
  unstuff(fr.scan, fr.scan + fr.scan_len, cs);
  if ((int64_t)cs.n_intervals() != n_int) return decode_scan(fr, g, coef, e);
  const uint8_t *base = cs.bytes.data();
  const uint8_t *hard_limit = base + cs.start.back() + 8;
  auto run = [&](int64_t i0, int64_t i1) {
    for (int64_t i = i0; i < i1; i++) {
      const int64_t m0 = ri > 0 ? i * ri : 0, m1 = (ri > 0 && m0 + ri < n_mcus) ? m0 + ri : n_mcus;
      const int rc = decode_interval(fr, g, base + cs.start[(size_t)i], cs.start[(size_t)i + 1] - cs.start[(size_t)i],
                                     hard_limit, m0, m1, coef);
      if (rc != JB_OK) return rc;
    }
    return (int)JB_OK;
  };
  if (n_threads > n_int) n_threads = (int)n_int;
  if (n_threads <= 1) {
    // a failure is re-examined by the bit-serial reader, which words the error ("restart marker
    // missing", "ends early", ...) and is the authority on what is tolerated
    return run(0, n_int) == JB_OK ? JB_OK : decode_scan(fr, g, coef, e);
  }
  std::vector<int> rcs((size_t)n_threads, JB_OK);
  std::vector<std::thread> th;
  for (int t = 0; t < n_threads; t++)
    th.emplace_back([&, t] { rcs[(size_t)t] = run(n_int * t / n_threads, n_int * (t + 1) / n_threads); });
  for (auto &x : th) x.join();
  for (int rc : rcs)
    if (rc != JB_OK) return decode_scan(fr, g, coef, e);
  return JB_OK;
}

int report(jb_ctx *ctx, const Err &e) { return jb_fail_(ctx, e.code, e.msg.c_str()); }

}  // namespace

bool jb_huff_fill_table_(const uint8_t counts[17], const uint8_t *symbols, bool is_ac, JbHuffTables *set, uint32_t tix, uint32_t n_tabs, uint32_t *n_t2) {
  return jb_huff_fill_table_impl_(counts, symbols, is_ac, set, tix, n_tabs, n_t2);
}

// Ready one image for the device-side entropy decoder (jb_huff.hip): see jb_huff.h.
int jb_huff_prepare_(const uint8_t *jpeg, size_t jpeg_bytes, JbHuffJob *job, std::string *err, uint32_t chunk_knob) {
  Frame *fr = new Frame();
  Err e;
  int rc = parse_headers(jpeg, jpeg_bytes, *fr, e, true);
  auto done = [&](int code, const char *msg) {
    if (err) *err = msg ? msg : e.msg;
    delete fr;
    return code;
  };
  if (rc != JB_OK) return done(rc, nullptr);
  job->desc = fr->desc;
  memcpy(job->qtabs, fr->qtabs, sizeof job->qtabs);
  rc = jb_geometry_of(&fr->desc, &job->geo);
  if (rc != JB_OK) return done(rc, "bad frame geometry");
  const int64_t n_mcus = (int64_t)job->geo.mcus_x * job->geo.mcus_y;
  const int ri = fr->restart_interval > 0 ? fr->restart_interval : 0;
  const int64_t n_int = ri > 0 ? (n_mcus + ri - 1) / ri : 1;
  // the tables in use: a frame's three components name at most three tables of each kind (reference
  // jpeg.cpp:148-196 reads up to four ids of each kind); AC tables first, then DC
  int ac_ids[3] = {-1, -1, -1}, dc_ids[3] = {-1, -1, -1}, ac_slot[3], dc_slot[3], n_ac = 0, n_dc = 0;
  for (int c = 0; c < fr->ncomp; c++) {
    for (int kind = 0; kind < 2; kind++) {
      int *ids = kind ? ac_ids : dc_ids, &n = kind ? n_ac : n_dc;
      const int id = kind ? fr->ac_id[c] : fr->dc_id[c];
      int slot = -1;
      for (int q = 0; q < n; q++)
        if (ids[q] == id) slot = q;
      if (slot < 0) ids[slot = n++] = id;
      (kind ? ac_slot : dc_slot)[c] = slot;
    }
  }
  job->n_tabs = (uint32_t)(n_ac + n_dc);
  job->img.n_tabs = job->n_tabs;
  memset(&job->tables, 0, sizeof job->tables);
  {
    uint32_t n_t2 = 0;
    for (int q = 0; q < n_ac + n_dc; q++) {
      const bool is_ac = q < n_ac;
      const HuffTable &t = is_ac ? fr->ac[ac_ids[q]] : fr->dc[dc_ids[q - n_ac]];
      if (!jb_huff_fill_table_(t.counts, t.symbols, is_ac, &job->tables, (uint32_t)q, job->n_tabs, &n_t2))
        return done(JB_ERR_UNSUPPORTED, "Huffman tables with more long codes than the device decoder's tables hold: host decoder");
    }
  }
  // a single-component frame: one block per MCU in the scan, delivered as the Y block of a 4:4:4 MCU (Cb, Cr stay zero)
  const uint32_t ny = (uint32_t)(fr->desc.hs * fr->desc.vs);
  job->img.nb = fr->ncomp == 1 ? 1u : ny + 2;
  job->img.lut_ac = job->img.lut_dc = job->img.lut_comp = 0;
  for (uint32_t b = 0; b < job->img.nb; b++) {
    const uint32_t c = b < ny ? 0u : b - ny + 1u;
    job->img.lut_ac |= (uint32_t)ac_slot[c] << (4 * b);
    job->img.lut_dc |= (uint32_t)(n_ac + dc_slot[c]) << (4 * b);
    job->img.lut_comp |= c << (4 * b);
  }
  static thread_local CleanScan tls_scan;
  CleanScan &cs = tls_scan;
  unstuff(fr->scan, fr->scan + fr->scan_len, cs);
  if ((int64_t)cs.n_intervals() != n_int) return done(JB_ERR_UNSUPPORTED, "restart markers do not match the frame: host decoder");
  if (cs.start.back() > 0x0ffff000u) return done(JB_ERR_UNSUPPORTED, "scan too large for 32-bit bit offsets: host decoder");
  job->scan_len = cs.start.back();
  job->scan.assign(cs.bytes.begin(), cs.bytes.begin() + (long)(job->scan_len + 64));  // (unstuff zero-pads far beyond 64)
  job->starts.resize(cs.start.size());
  for (size_t i = 0; i < cs.start.size(); i++) job->starts[i] = (uint32_t)cs.start[i];
  job->img.scan_off = job->img.int_off = job->img.table_set = 0;
  job->img.scan_len = (uint32_t)job->scan_len;
  job->img.n_int = (uint32_t)n_int;
  job->img.ri = (uint32_t)(ri > 0 ? ri : n_mcus);
  job->img.n_mcus = (uint32_t)n_mcus;
  job->img.coef_off = 0;
  job->img.blk_bytes = fr->ncomp == 1 ? 384u : 128u;
  job->img.wg0 = 0;
  if ((uint64_t)n_mcus * job->img.nb >= (1u << 24)) return done(JB_ERR_UNSUPPORTED, "more blocks than the device decoder's 24-bit block index holds: host decoder");
  job->img.n_blocks = (uint32_t)(n_mcus * job->img.nb);
  // Every restart interval (a scan without DRI is one interval) is cut into chunks from its own first byte, one
  // lane per chunk (jb_huff.h); the caller's knob may force the size (JPEGBLK_CHUNK_BYTES, jb_knobs.h: tests, A/B runs)
  const uint32_t chunk_bytes = (chunk_knob == 64 || chunk_knob == 128) ? chunk_knob : kJbChunkBytes;
  job->img.chunk_bytes = chunk_bytes;
  {
    uint64_t n_chunks = 0;
    bool needs_sync = false;
    for (size_t i = 0; i + 1 < job->starts.size(); i++) {
      if (job->starts[i + 1] < job->starts[i]) return done(JB_ERR_FORMAT, "restart intervals out of order");
      const uint32_t k = jb_chunks_of_(job->starts[i + 1] - job->starts[i], chunk_bytes);
      needs_sync |= k > 1;
      n_chunks += k;
    }
    if (n_chunks > 0x3fffffffu) return done(JB_ERR_UNSUPPORTED, "scan too large for the device decoder: host decoder");
    job->img.n_chunks = (uint32_t)n_chunks;
    job->img.needs_sync = needs_sync ? 1u : 0u;
  }
  job->img.state_off = 0;
  if (ri == 0 && job->scan_len == 0) return done(JB_ERR_FORMAT, "empty scan");
  return done(JB_OK, "");
}

extern "C" {

int jb_entropy_decode(const uint8_t *jpeg, size_t jpeg_bytes, jb_image_desc *desc, uint16_t *qtabs,
                      int16_t *coef, size_t coef_cap_bytes) {
  return jb_entropy_decode_mt(jpeg, jpeg_bytes, desc, qtabs, coef, coef_cap_bytes, 1);
}

int jb_entropy_decode_mt(const uint8_t *jpeg, size_t jpeg_bytes, jb_image_desc *desc, uint16_t *qtabs,
                         int16_t *coef, size_t coef_cap_bytes, int n_threads) {
  if (!jpeg || !desc) return jb_fail_(nullptr, JB_ERR_NULL, "jb_entropy_decode: NULL pointer");
  Frame *fr = new Frame();
  Err e;
  int rc = parse_headers(jpeg, jpeg_bytes, *fr, e);
  if (rc == JB_ERR_UNSUPPORTED) {
    // not the common case (baseline, three components, one interleaved scan): the general front
    // end takes progressive, grayscale and multi-scan files, and rejects the rest
    delete fr;
    std::string text;
    rc = jb_ext_decode_(jpeg, jpeg_bytes, desc, qtabs, coef, coef_cap_bytes, &text);
    return rc ? jb_fail_(nullptr, rc, text.c_str()) : JB_OK;
  }
  if (rc == JB_OK) {
    *desc = fr->desc;
    if (qtabs) memcpy(qtabs, fr->qtabs, sizeof fr->qtabs);
    if (coef) {
      jb_geometry g;
      rc = jb_geometry_of(&fr->desc, &g);
      if (rc) set_err(e, rc, "bad frame geometry");
      else if ((size_t)g.coef_bytes > coef_cap_bytes) rc = set_err(e, JB_ERR_CAPACITY, "coefficient buffer too small");
      else rc = decode_scan_mt(*fr, g, coef, n_threads, e);
    }
  }
  delete fr;
  return rc ? report(nullptr, e) : JB_OK;
}

int jb_decode_memory(jb_ctx *ctx, const uint8_t *jpeg, size_t jpeg_bytes, uint8_t **rgb, int32_t *width,
                     int32_t *height) {
  if (!ctx) return jb_fail_(nullptr, JB_ERR_NULL, "jb_decode_memory: ctx is NULL");
  if (!jpeg || !rgb || !width || !height) return jb_fail_(ctx, JB_ERR_NULL, "jb_decode_memory: NULL pointer");
  *rgb = nullptr;
  // The entropy stage of a baseline file can run on the device too (jb_huff.hip): the host then only
  // parses the headers and removes the byte stuffing.  One image is one latency-bound submission
  // (a dozen and a half launches: about 1 ms whatever the size, then ~0.4 ms per megabyte of scan)
  // against 5.5 ms per megabyte on one host core, so by default the device takes files of
  // kAutoDeviceScan bytes of scan or more -- measured with round 3's kernels: 679x451 (80 KB) 0.56-0.85 ms against
  // 0.67-0.73 on the host, 1024x768 4:2:0 (204 KB) 0.63 against 1.19, 1280x720 4:2:0 (238 KB) 0.57 against 1.36,
  // 1920x1080 4:4:4 (760 KB) 0.51 against 3.9, 8192x8192 4:2:0 (17 MB) 7.5 against 94
  // (tools/single_latency.py, profiles/r03/single_latency.txt; the threshold was 256 KB with round 2's kernels).  JPEGBLK_GPU_HUFFMAN=0: always the host decoder;
  // =1: the device for every file with 16 restart intervals / chunks or more; =2: also fewer intervals.
  // Whatever the device decoder does not take or flags as corrupt goes through the host decoder
  // below, which gives the precise answer.
  {
    constexpr size_t kAutoDeviceScan = (size_t)128 << 10;
    const JbKnobs &knobs = *jb_ctx_knobs_(ctx);  // (read when the context was created: jb_knobs.h)
    const bool forced = knobs.gpu_huffman == 1 || knobs.gpu_huffman == 2;
    const bool automatic = knobs.gpu_huffman < 0;
    const uint32_t min_int = knobs.gpu_huffman == 2 ? 1u : 16u;
    if (forced || (automatic && jpeg_bytes >= kAutoDeviceScan)) {
      // JPEGBLK_TIMING=1: where one decode(bytes) through the device path spends its time, on stderr
      const bool timing = knobs.timing == 1;
      auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
      const double t0 = timing ? now() : 0;
      std::unique_ptr<JbHuffJob> job(new JbHuffJob());
      if (jb_huff_prepare_(jpeg, jpeg_bytes, job.get(), nullptr, knobs.chunk_bytes) == JB_OK && jb_huff_worth_it_(*job, min_int) &&
          (forced || job->scan_len >= kAutoDeviceScan)) {
        const double t1 = timing ? now() : 0;
        uint8_t *out = jb_alloc_pixels_((size_t)job->geo.rgb_bytes);
        if (!out) return jb_fail_(ctx, JB_ERR_CAPACITY, "out of host memory");
        const int rc = jb_decode_job_(ctx, job.get(), out, 3LL * job->desc.width);
        if (timing)
          fprintf(stderr, "jb_decode_memory(device path): prepare %.3f ms, submit + wait %.3f ms (%u intervals, %u chunks, %zu bytes of scan), rc %d\n",
                  (t1 - t0) * 1e3, (now() - t1) * 1e3, job->img.n_int, job->img.n_chunks, job->scan_len, rc);
        if (rc == JB_OK) {
          *rgb = out;
          *width = job->desc.width;
          *height = job->desc.height;
          jb_ctx_set_last_desc_(ctx, &job->desc);
          return JB_OK;
        }
        free(out);
        if (rc != JB_ERR_FORMAT) return rc;  // a HIP / capacity error is not the stream's fault
      }
    }
  }
  jb_image_desc desc;
  uint16_t qtabs[256];
  int rc = jb_entropy_decode(jpeg, jpeg_bytes, &desc, qtabs, nullptr, 0);
  if (rc) return jb_fail_(ctx, rc, jb_last_error(nullptr));
  jb_geometry g;
  rc = jb_geometry_of(&desc, &g);
  if (rc) return jb_fail_(ctx, rc, "bad frame geometry");
  // the staging ring follows the frame (a context sized for another image, or created with (0,0))
  rc = jb_ctx_reserve(ctx, (size_t)g.coef_bytes, (size_t)g.rgb_bytes);
  if (rc) return rc;
  int16_t *coef = (int16_t *)jb_pinned_alloc_on(jb_ctx_device(ctx), (size_t)g.coef_bytes);
  if (!coef) return jb_fail_(ctx, JB_ERR_HIP, jb_last_error(nullptr));
  rc = jb_entropy_decode(jpeg, jpeg_bytes, &desc, qtabs, coef, (size_t)g.coef_bytes);
  uint8_t *out = nullptr;
  if (rc) jb_fail_(ctx, rc, jb_last_error(nullptr));
  else {
    out = jb_alloc_pixels_((size_t)g.rgb_bytes);
    if (!out) rc = jb_fail_(ctx, JB_ERR_CAPACITY, "out of host memory");
    else rc = jb_blocks_to_rgb(ctx, &desc, coef, qtabs, out, 3LL * desc.width);
  }
  jb_pinned_free(coef);
  if (rc) {
    free(out);
    return rc;
  }
  *rgb = out;
  *width = desc.width;
  *height = desc.height;
  jb_ctx_set_last_desc_(ctx, &desc);
  return JB_OK;
}

int jb_decode_file(jb_ctx *ctx, const char *path, uint8_t **rgb, int32_t *width, int32_t *height) {
  if (!ctx) return jb_fail_(nullptr, JB_ERR_NULL, "jb_decode_file: ctx is NULL");
  if (!path) return jb_fail_(ctx, JB_ERR_NULL, "jb_decode_file: path is NULL");
  FILE *f = fopen(path, "rb");
  if (!f) return jb_fail_(ctx, JB_ERR_FORMAT, (std::string("cannot open ") + path).c_str());
  std::vector<uint8_t> buf;
  uint8_t chunk[1 << 16];
  size_t got;
  while ((got = fread(chunk, 1, sizeof chunk, f)) > 0) buf.insert(buf.end(), chunk, chunk + got);
  fclose(f);
  return jb_decode_memory(ctx, buf.data(), buf.size(), rgb, width, height);
}

}  // extern "C"
