// tools/fuzz/fuzz_frontend.cpp -- mutation fuzzer for the host front end (marker parser + Huffman
// decoder, csrc/jb_frontend.cpp) built for the CPU with AddressSanitizer + UBSan.  A decoder library
// sits on untrusted input: whatever the bytes, jb_entropy_decode must return a status -- never
// read or write out of bounds, overflow, loop forever or abort (the reference exit(1)s or reads
// past its buffers on such input, jpeg.cpp:886-907, file.hpp:59-104).
//
//   make -C tools/fuzz && tools/fuzz/fuzz_frontend <seconds> <seed> file.jpg [file.jpg ...]
//
// Seeds: the given files plus streams from the build's own writer (all layouts, restart
// intervals, 16-bit tables).  The device side is stubbed; only the host code is under test.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/jpegblk.h"
#include "../../jpeg_decoder_amd/csrc/jb_huff.h"
#include "../../jpeg_decoder_amd/csrc/jb_knobs.h"

extern "C" long jw_encode_ex(const int16_t *coef, int width, int height, int hs, int vs, const uint16_t *qtabs,
                             const int *qtab_id, const uint8_t *dht, int restart_interval, int dqt16, int scan_mode,
                             uint8_t *out, long cap);

// ---- stubs for the parts of the library that need a device ----
struct jb_ctx;
static std::string g_err;
int jb_fail_(jb_ctx *, int code, const char *msg) {
  g_err = msg ? msg : "";
  return code;
}
void jb_ctx_set_last_desc_(jb_ctx *, const jb_image_desc *) {}
const JbKnobs *jb_ctx_knobs_(const jb_ctx *) {
  static const JbKnobs k;
  return &k;
}
extern "C" {
const char *jb_last_error(const jb_ctx *) { return g_err.c_str(); }
void *jb_pinned_alloc(size_t n) { return malloc(n); }
void *jb_pinned_alloc_on(int, size_t n) { return malloc(n); }
int jb_ctx_reserve(jb_ctx *, size_t, size_t) { return JB_OK; }
int jb_ctx_device(const jb_ctx *) { return 0; }
struct JbHuffJob;
int jb_decode_job_(jb_ctx *, const JbHuffJob *, uint8_t *, int64_t) { return JB_OK; }
void jb_pinned_free(void *p) { free(p); }
void jb_free(void *p) { free(p); }
int jb_blocks_to_rgb(jb_ctx *, const jb_image_desc *, const int16_t *, const uint16_t *, uint8_t *, int64_t) { return JB_OK; }
}

static uint64_t rng_state = 88172645463325252ull;
static inline uint64_t rnd() {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return rng_state;
}

static std::vector<uint8_t> writer_seed(int w, int h, int hs, int vs, int ri, int dqt16, int scan_mode) {
  // Annex K.3 tables are not needed here: any complete prefix code is a legal DHT.  Build simple
  // ones: DC 12 symbols (lengths 2..), AC 162 symbols in run/size order.
  static uint8_t dht[4 * 272];
  static bool init = false;
  if (!init) {
    init = true;
    const uint8_t dc_bits[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
    const uint8_t ac_bits[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
    for (int t = 0; t < 4; t++) {
      uint8_t *p = dht + t * 272;
      if (t % 2 == 0) {
        memcpy(p, dc_bits, 16);
        for (int i = 0; i < 12; i++) p[16 + i] = (uint8_t)i;
      } else {
        memcpy(p, ac_bits, 16);
        int k = 0;
        p[16 + k++] = 0x00;
        p[16 + k++] = 0xf0;
        for (int size = 1; size <= 10; size++)
          for (int run = 0; run < 16; run++) p[16 + k++] = (uint8_t)((run << 4) | size);
      }
    }
  }
  const int bw = (w + 7) / 8, bh = (h + 7) / 8;
  const int mx = (bw + (hs == 2 && (bw & 1))) / hs, my = (bh + (vs == 2 && (bh & 1))) / vs;
  const long n = (long)mx * my * (hs * vs + 2);
  std::vector<int16_t> coef((size_t)n * 64, 0);
  for (long b = 0; b < n; b++) {
    coef[b * 64] = (int16_t)((int)(rnd() % 129) - 64);
    for (int k = 1; k < 64; k++)
      if (rnd() % 100 < (uint64_t)(60 / (1 + k / 4))) coef[b * 64 + k] = (int16_t)((int)(rnd() % 41) - 20);
  }
  uint16_t q[256];
  for (int i = 0; i < 256; i++) q[i] = (uint16_t)(1 + rnd() % 255);
  const int ids[3] = {0, 1, (int)(rnd() % 2) + 1};
  std::vector<uint8_t> out(2048 + coef.size() * 4);
  long len = jw_encode_ex(coef.data(), w, h, hs, vs, q, ids, dht, ri, dqt16, scan_mode, out.data(), (long)out.size());
  if (len < 0) {
    fprintf(stderr, "writer failed: %ld\n", len);
    exit(2);
  }
  out.resize((size_t)len);
  return out;
}

static void mutate(std::vector<uint8_t> &d) {
  const int ops = 1 + (int)(rnd() % 4);
  for (int o = 0; o < ops && !d.empty(); o++) {
    const size_t n = d.size();
    // bias positions toward the headers (first 700 bytes) half of the time
    const size_t pos = (rnd() & 1) ? rnd() % n : rnd() % (n < 700 ? n : 700);
    switch (rnd() % 9) {
      case 0: d[pos] ^= (uint8_t)(1u << (rnd() % 8)); break;
      case 1: d[pos] = (uint8_t)rnd(); break;
      case 2: d[pos] = 0xff; break;
      case 3: d.resize(pos); break;                                   // truncate
      case 4: d.insert(d.begin() + (long)pos, (size_t)(1 + rnd() % 8), (uint8_t)rnd()); break;
      case 5: { size_t k = rnd() % 16; if (k > n - pos) k = n - pos; d.erase(d.begin() + (long)pos, d.begin() + (long)(pos + k)); } break;
      case 6: if (pos + 1 < n) { d[pos] = 0xff; d[pos + 1] = (uint8_t)(0xc0 + rnd() % 0x40); } break;  // a marker
      case 7: if (pos + 3 < n) { d[pos + 2] = (uint8_t)rnd(); d[pos + 3] = (uint8_t)rnd(); } break;     // a length
      case 8: if (pos + 4 < n) { uint32_t v = (rnd() & 1) ? 0xffffffffu : 0u; memcpy(&d[pos], &v, 4); } break;
    }
  }
}

// The host half of the device-side entropy decoder (jb_huff_prepare_: header parse, table slots,
// de-stuffing, interval table) on the same mutant: it must either refuse or produce a job whose
// interval table is monotonic and inside the clean scan -- what the kernel relies on.
static void prepare_once(const std::vector<uint8_t> &in) {
  static JbHuffJob job;
  if (jb_huff_prepare_(in.data(), in.size(), &job, nullptr) != JB_OK) return;
  const JbHuffImage &im = job.img;
  bool bad = job.starts.size() != (size_t)im.n_int + 1 || job.scan.size() < job.scan_len + 64 || job.starts.back() != job.scan_len ||
             (uint64_t)im.n_int * im.ri < im.n_mcus || !(im.nb == 1 || im.nb == 3 || im.nb == 4 || im.nb == 6) ||
             (uint64_t)im.n_mcus * im.nb != im.n_blocks || im.n_blocks >= (1u << 24) || im.n_tabs < 2 || im.n_tabs > kJbMaxTabs ||
             im.n_tabs != job.n_tabs || !(im.chunk_bytes == 64 || im.chunk_bytes == 128) || !(im.blk_bytes == 128 || im.blk_bytes == 384);
  uint64_t chunks = 0;
  bool multi = false;
  for (size_t i = 1; !bad && i < job.starts.size(); i++) {
    bad = job.starts[i] < job.starts[i - 1];
    const uint32_t k = jb_chunks_of_(job.starts[i] - job.starts[i - 1], im.chunk_bytes);
    chunks += k;
    multi |= k > 1;
  }
  bad = bad || chunks != im.n_chunks || (multi ? 1u : 0u) != im.needs_sync;
  // block-in-MCU -> table / component: indices the kernels use without a check
  for (uint32_t blk = 0; !bad && blk < im.nb; blk++)
    bad = ((im.lut_ac >> (4 * blk)) & 15u) >= im.n_tabs || ((im.lut_dc >> (4 * blk)) & 15u) >= im.n_tabs || ((im.lut_comp >> (4 * blk)) & 15u) > 2;
  // table entries: nothing, a second-level table that exists, or a symbol of at most 27 bits
  for (uint32_t t = 0; !bad && t < im.n_tabs; t++)
    for (uint32_t i = 0; !bad && i < kJbT1Entries; i++) {
      const uint16_t e = job.tables.t1[t][i];
      bad = e != 0 && !(e >= 1 && e <= kJbT2Tables) && !(e >= 512 && (e & 31) >= 1 && (e & 31) <= 27 && ((e >> 5) & 15) <= 11);
    }
  for (uint32_t t = 0; !bad && t < kJbT2Tables; t++)
    for (uint32_t i = 0; !bad && i < kJbT2Entries; i++) {
      const uint16_t e = job.tables.t2[t][i];
      bad = e != 0 && !(e >= 512 && (e & 31) >= 1 && (e & 31) <= 27 && ((e >> 5) & 15) <= 11);
    }
  if (bad) {
    fprintf(stderr, "jb_huff_prepare_ produced an inconsistent job\n");
    abort();
  }
}

int main(int argc, char **argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: %s <seconds> <seed> [file.jpg ...]\n", argv[0]);
    return 2;
  }
  const double budget = atof(argv[1]);
  rng_state ^= (uint64_t)strtoull(argv[2], nullptr, 0) * 0x9E3779B97F4A7C15ull;
  std::vector<std::vector<uint8_t>> seeds;
  for (int i = 3; i < argc; i++) {
    FILE *f = fopen(argv[i], "rb");
    if (!f) continue;
    std::vector<uint8_t> b;
    uint8_t tmp[65536];
    size_t got;
    while ((got = fread(tmp, 1, sizeof tmp, f)) > 0) b.insert(b.end(), tmp, tmp + got);
    fclose(f);
    if (b.size() <= (1u << 20)) seeds.push_back(std::move(b));  // keep iterations fast
  }
  // {w, h, hs, vs, restart interval, 16-bit DQT, scan mode (1 = one scan per component)}
  const int shapes[][7] = {{33, 17, 1, 1, 0, 0, 0}, {100, 60, 2, 1, 3, 0, 0}, {64, 80, 1, 2, 1, 1, 0}, {130, 70, 2, 2, 5, 0, 0},
                           {16, 16, 2, 2, 0, 1, 0}, {257, 9, 1, 1, 7, 0, 0}, {97, 61, 2, 2, 4, 0, 1}, {40, 40, 1, 1, 0, 0, 1}};
  for (auto &s : shapes) seeds.push_back(writer_seed(s[0], s[1], s[2], s[3], s[4], s[5], s[6]));
  // every unmutated seed must decode, or be rejected as unsupported (a progressive file is a fine seed)
  for (auto &s : seeds) {
    jb_image_desc d;
    uint16_t q[256];
    const int rc = jb_entropy_decode(s.data(), s.size(), &d, q, nullptr, 0);
    if (rc != JB_OK && rc != JB_ERR_UNSUPPORTED) {
      fprintf(stderr, "seed rejected: %s\n", g_err.c_str());
      return 1;
    }
  }
  const auto t0 = std::chrono::steady_clock::now();
  long iters = 0, ok = 0, counts[16] = {0};
  std::vector<int16_t> coef;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < budget) {
    std::vector<uint8_t> d = seeds[rnd() % seeds.size()];
    mutate(d);
    prepare_once(d);
    jb_image_desc desc;
    uint16_t q[256];
    int rc = jb_entropy_decode(d.data(), d.size(), &desc, q, nullptr, 0);
    if (rc == JB_OK) {
      jb_geometry g;
      rc = jb_geometry_of(&desc, &g);
      if (rc == JB_OK && g.coef_bytes <= (64ll << 20)) {
        // exact-size heap buffer (ASan sees one byte too many); sometimes too small on purpose
        const size_t cap = (rnd() % 8 == 0 && g.coef_bytes > 128) ? (size_t)g.coef_bytes - 128 : (size_t)g.coef_bytes;
        coef.assign(cap / 2, 0);
        coef.shrink_to_fit();
        rc = jb_entropy_decode_mt(d.data(), d.size(), &desc, q, coef.data(), cap, (rnd() % 4 == 0) ? 3 : 1);
      }
    }
    if (rc > 0 || rc < -9) {
      fprintf(stderr, "status %d outside the jb_status range\n", rc);
      return 1;
    }
    counts[-rc]++;
    ok += rc == JB_OK;
    iters++;
  }
  printf("%ld mutants, %ld decoded, statuses:", iters, ok);
  for (int i = 0; i < 10; i++) printf(" %d:%ld", -i, counts[i]);
  printf("\n");
  return 0;
}
