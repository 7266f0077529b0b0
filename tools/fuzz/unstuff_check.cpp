// Differential check of unstuff() (csrc/jb_entropy.h: the word loop with PEXT and the memchr loop, switched per
// 4 KiB) against the one-byte-at-a-time statement of the same rule, on random scans of every density of 0xFF,
// with RSTn markers, fill bytes, lone 0xFF at the end and an EOI somewhere.  Built with ASan + UBSan (Makefile).
//   unstuff_check <seconds> <seed>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../jpeg_decoder_amd/csrc/jb_entropy.h"

using namespace jbe;

static uint64_t rng_state = 0x2545F4914F6CDD1Dull;
static uint64_t rnd() {
  rng_state ^= rng_state << 13, rng_state ^= rng_state >> 7, rng_state ^= rng_state << 17;
  return rng_state;
}

static void plain(const uint8_t *p, const uint8_t *end, std::vector<uint8_t> &out, std::vector<size_t> &start) {
  out.clear(), start.clear();
  start.push_back(0);
  while (p < end) {
    if (*p != 0xff) {
      out.push_back(*p++);
      continue;
    }
    if (p + 1 >= end) break;
    const uint8_t m = p[1];
    if (m == 0x00) out.push_back(0xff), p += 2;
    else if (m == 0xff) p += 1;
    else if (m >= 0xd0 && m <= 0xd7) start.push_back(out.size()), p += 2;
    else break;
  }
  start.push_back(out.size());
}

int main(int argc, char **argv) {
  const double budget = argc > 1 ? atof(argv[1]) : 5.0;
  if (argc > 2) rng_state ^= strtoull(argv[2], nullptr, 0) * 0x9E3779B97F4A7C15ull;
  const auto t0 = std::chrono::steady_clock::now();
  CleanScan cs;
  std::vector<uint8_t> in, want;
  std::vector<size_t> want_start;
  long cases = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < budget) {
    const size_t n = rnd() % 7 == 0 ? rnd() % 70 : rnd() % 40000;
    const unsigned ff_one_in = 1u << (rnd() % 10);        // 1 .. 512
    const unsigned marker_one_in = 1u << (2 + rnd() % 14);
    const bool switch_density = rnd() & 1;
    in.clear();
    while (in.size() < n) {
      unsigned d = ff_one_in;
      if (switch_density && ((in.size() >> 13) & 1)) d = 512;  // dense and sparse stretches of 8 KiB in one scan
      if (rnd() % d == 0) {
        in.push_back(0xff);
        const uint64_t r = rnd();
        if (r % marker_one_in == 0) in.push_back(0xd0 + (r >> 20) % 8);
        else if (r % (marker_one_in * 8) == 1) in.push_back(0xff);            // fill byte, then whatever follows
        else if (r % (marker_one_in * 64) == 2) in.push_back(0xd9);           // EOI in the middle
        else if (r % (marker_one_in * 8) == 3) { /* a bare 0xFF followed by a random byte */ }
        else in.push_back(0x00);
      } else {
        in.push_back((uint8_t)(rnd() % 255));  // never 0xFF by itself
      }
    }
    // the caller's buffer is exactly this long: ASan sees any read past its end
    std::vector<uint8_t> exact(in.begin(), in.end());
    unstuff(exact.data(), exact.data() + exact.size(), cs);
    plain(exact.data(), exact.data() + exact.size(), want, want_start);
    const size_t got_n = cs.start.back();
    bool ok = got_n == want.size() && cs.start == want_start && cs.bytes.size() >= got_n + CleanScan::kPad &&
              (got_n == 0 || memcmp(cs.bytes.data(), want.data(), got_n) == 0);
    for (size_t i = 0; ok && i < CleanScan::kPad; i++) ok = cs.bytes[got_n + i] == 0;
    if (!ok) {
      printf("MISMATCH case %ld: n=%zu ff_one_in=%u got %zu bytes / %zu starts, want %zu / %zu\n", cases, in.size(),
             ff_one_in, got_n, cs.start.size(), want.size(), want_start.size());
      return 1;
    }
    cases++;
  }
  printf("unstuff_check: %ld scans equal\n", cases);
  return 0;
}
