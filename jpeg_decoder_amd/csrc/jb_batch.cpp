// jb_batch.cpp -- multi-threaded decode(path) over a batch of files: host entropy decoding on
// n_threads cores overlapped with the device block pipeline (include/jpegblk.h, jb_decode_batch).
// The reference decodes one file per process, single-threaded (jpeg.cpp:916-929); images are
// independent, so the batch parallelises by image with no shared state between threads.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <sys/mman.h>

#include "../../include/jpegblk.h"

struct jb_ctx;
int jb_fail_(jb_ctx *ctx, int code, const char *msg);

namespace {

double now_s() {
  using namespace std::chrono;
  return duration<double>(steady_clock::now().time_since_epoch()).count();
}

bool read_file(const char *path, std::vector<uint8_t> &buf) {
  FILE *f = fopen(path, "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  if (n < 0) {
    fclose(f);
    return false;
  }
  buf.resize((size_t)n);
  size_t got = n ? fread(buf.data(), 1, (size_t)n, f) : 0;
  fclose(f);
  return got == (size_t)n;
}

// A caller-owned pixel buffer (released with jb_free = free).  Large images are first-touched by
// the copy out of the pinned staging: with 4 KiB pages that is 49,000 page faults for one
// 8192x8192 image, taken by 16 threads at once, so ask for transparent huge pages.
uint8_t *alloc_pixels(size_t bytes) {
  constexpr size_t kHuge = (size_t)2 << 20;
  if (bytes < 2 * kHuge) return (uint8_t *)malloc(bytes);
  void *p = nullptr;
  if (posix_memalign(&p, kHuge, (bytes + kHuge - 1) & ~(kHuge - 1)) != 0) return nullptr;
  madvise(p, (bytes + kHuge - 1) & ~(kHuge - 1), MADV_HUGEPAGE);  // advisory: ignoring a failure is fine
  return (uint8_t *)p;
}

struct Parsed {
  std::vector<uint8_t> bytes;
  jb_image_desc desc;
  jb_geometry geo;
  uint16_t qtabs[256];
  int status = JB_OK;
};

struct Totals {
  std::mutex mu;
  double t_entropy = 0, t_device = 0, t_read = 0;
  int first_error = JB_OK;
  std::string first_error_text;
};

constexpr int kSlots = 2;

// what one worker thread owns across runs: a context (stream + device staging ring) and the
// pinned coefficient buffers the Huffman stage decodes into
struct Lane {
  jb_ctx *ctx = nullptr;
  int16_t *coef[kSlots] = {nullptr, nullptr};
  uint8_t *out[kSlots] = {nullptr, nullptr};  // pinned pixel staging (unused when an arena takes the pixels)
  size_t cap_coef = 0, cap_rgb = 0;
  bool has_out = false;
  int device = 0;

  // with_out: also the pinned pixel staging the device copies into when the caller's buffers are
  // pageable (a device-to-host copy into pageable memory blocks the submitting thread until the
  // kernel has run, which would serialise Huffman decoding and device work)
  int ensure(size_t need_coef, size_t need_rgb, bool with_out) {
    if (ctx && need_coef <= cap_coef && need_rgb <= cap_rgb && (has_out || !with_out)) return JB_OK;
    if (ctx && need_coef < cap_coef) need_coef = cap_coef;
    if (ctx && need_rgb < cap_rgb) need_rgb = cap_rgb;
    release();
    int rc = jb_ctx_create(device, need_coef, need_rgb, kSlots, &ctx);
    for (int s = 0; s < kSlots && rc == JB_OK; s++) {
      coef[s] = (int16_t *)jb_pinned_alloc(need_coef);
      if (!coef[s]) rc = JB_ERR_HIP;
      if (with_out && rc == JB_OK) {
        out[s] = (uint8_t *)jb_pinned_alloc(need_rgb);
        if (!out[s]) rc = JB_ERR_HIP;
      }
    }
    if (rc == JB_OK) has_out = with_out;
    if (rc == JB_OK) {
      cap_coef = need_coef;
      cap_rgb = need_rgb;
    } else {
      release();
    }
    return rc;
  }
  void release() {
    for (int s = 0; s < kSlots; s++) {
      jb_pinned_free(coef[s]);
      jb_pinned_free(out[s]);
      coef[s] = nullptr;
      out[s] = nullptr;
    }
    jb_ctx_destroy(ctx);
    ctx = nullptr;
    cap_coef = cap_rgb = 0;
    has_out = false;
  }
};

// Pinned output arena (optional, jb_batch_decoder_set_arena): images are placed by an atomic bump
// pointer, so the device writes every pixel straight to its final place.
struct Arena {
  uint8_t *base = nullptr;
  size_t bytes = 0;
  std::atomic<size_t> used{0};
  uint8_t *take(size_t n) {
    n = (n + 255) & ~(size_t)255;
    size_t at = used.fetch_add(n);
    if (at + n > bytes) return nullptr;
    return base + at;
  }
};

void worker(Lane *lane, Arena *arena, int t, int n_threads, int inner_threads, const char *const *paths, int n_paths,
            uint8_t **rgb, int32_t *widths, int32_t *heights, int *statuses, Totals *tot) {
  double t_entropy = 0, t_device = 0, t_read = 0, t_copy = 0;
  const bool use_arena = arena && arena->base;
  // pass 1: read + parse headers of this thread's files (sizes the context once)
  std::vector<int> mine;
  for (int i = t; i < n_paths; i += n_threads) mine.push_back(i);
  std::vector<Parsed> parsed(mine.size());
  size_t max_coef = 0, max_rgb = 0;
  for (size_t k = 0; k < mine.size(); k++) {
    Parsed &p = parsed[k];
    double a = now_s();
    if (!read_file(paths[mine[k]], p.bytes)) {
      p.status = JB_ERR_FORMAT;
      continue;
    }
    t_read += now_s() - a;
    p.status = jb_entropy_decode(p.bytes.data(), p.bytes.size(), &p.desc, p.qtabs, nullptr, 0);
    if (p.status == JB_OK) p.status = jb_geometry_of(&p.desc, &p.geo);
    if (p.status == JB_OK) {
      if ((size_t)p.geo.coef_bytes > max_coef) max_coef = (size_t)p.geo.coef_bytes;
      if ((size_t)p.geo.rgb_bytes > max_rgb) max_rgb = (size_t)p.geo.rgb_bytes;
    }
  }
  int rc = max_coef ? lane->ensure(max_coef, max_rgb, !use_arena) : JB_OK;
  jb_ctx *ctx = lane->ctx;
  int16_t **coef = lane->coef;
  max_coef = lane->cap_coef;
  // pass 2: decode image k into pinned slot k%2, submit, then wait for image k-1
  int pending_ticket = -1, pending_k = -1;
  auto finish = [&](int k, int st) {
    const int i = mine[k];
    statuses[i] = st;
    if (st == JB_OK && !use_arena) {  // pinned staging -> the caller's (pageable) buffer
      double a = now_s();
      memcpy(rgb[i], lane->out[k % kSlots], (size_t)parsed[k].geo.rgb_bytes);
      t_copy += now_s() - a;
    }
    if (st != JB_OK) {
      if (!use_arena) jb_free(rgb[i]);
      rgb[i] = nullptr;
      std::lock_guard<std::mutex> g(tot->mu);
      if (tot->first_error == JB_OK) {
        tot->first_error = st;
        tot->first_error_text = std::string(paths[i]) + ": " + jb_last_error(ctx);
      }
    }
  };
  for (size_t k = 0; k < mine.size(); k++) {
    Parsed &p = parsed[k];
    const int i = mine[k];
    rgb[i] = nullptr;
    widths[i] = heights[i] = 0;
    int st = rc != JB_OK ? rc : p.status;
    int ticket = -1;
    if (st == JB_OK) {
      double a = now_s();
      // fewer files than host threads: the spare threads split each image's restart intervals
      st = jb_entropy_decode_mt(p.bytes.data(), p.bytes.size(), &p.desc, p.qtabs, coef[k % kSlots], max_coef,
                                inner_threads);
      t_entropy += now_s() - a;
    }
    if (st == JB_OK) {
      rgb[i] = use_arena ? arena->take((size_t)p.geo.rgb_bytes) : alloc_pixels((size_t)p.geo.rgb_bytes);
      widths[i] = p.desc.width;
      heights[i] = p.desc.height;
      if (!rgb[i]) st = jb_fail_(ctx, JB_ERR_CAPACITY, use_arena ? "output arena exhausted" : "out of memory");
    }
    if (st == JB_OK) {
      double a = now_s();
      // every copy of the submission is pinned <-> device, so this returns at once and the
      // transfers and the kernel run while this thread decodes its next image
      uint8_t *dst = use_arena ? rgb[i] : lane->out[k % kSlots];
      st = jb_submit(ctx, &p.desc, coef[k % kSlots], p.qtabs, dst, 3LL * p.desc.width, &ticket);
      t_device += now_s() - a;
    }
    // the previous image: its pinned slot is needed again two images from now
    if (pending_ticket >= 0) {
      double a = now_s();
      int wst = jb_wait(ctx, pending_ticket);
      t_device += now_s() - a;
      finish(pending_k, wst);
      pending_ticket = -1;
    }
    if (st == JB_OK) {
      pending_ticket = ticket;
      pending_k = (int)k;
    } else {
      finish((int)k, st);
    }
    p.bytes.clear();
    p.bytes.shrink_to_fit();
  }
  if (pending_ticket >= 0) {
    double a = now_s();
    int wst = jb_wait(ctx, pending_ticket);
    t_device += now_s() - a;
    finish(pending_k, wst);
  }
  std::lock_guard<std::mutex> g(tot->mu);
  tot->t_entropy += t_entropy;
  tot->t_device += t_device + t_copy;
  tot->t_read += t_read;
}

}  // namespace

struct jb_batch_decoder {
  int device = 0;
  std::vector<Lane> lanes;
  Arena arena;
};

extern "C" int jb_batch_decoder_create(int device_id, int n_threads, size_t max_coef_bytes,
                                       size_t max_rgb_bytes, jb_batch_decoder **out) {
  if (!out) return jb_fail_(nullptr, JB_ERR_NULL, "jb_batch_decoder_create: out is NULL");
  *out = nullptr;
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 256) n_threads = 256;
  jb_batch_decoder *d = new jb_batch_decoder();
  d->device = device_id;
  d->lanes.resize((size_t)n_threads);
  for (auto &l : d->lanes) l.device = device_id;
  if (max_coef_bytes && max_rgb_bytes) {
    // create the per-thread contexts and pinned buffers in parallel (page pinning is slow)
    std::vector<std::thread> th;
    std::vector<int> rcs(d->lanes.size(), JB_OK);
    for (size_t i = 0; i < d->lanes.size(); i++)
      th.emplace_back([&, i] { rcs[i] = d->lanes[i].ensure(max_coef_bytes, max_rgb_bytes, true); });
    for (auto &x : th) x.join();
    for (int rc : rcs)
      if (rc != JB_OK) {
        for (auto &l : d->lanes) l.release();
        delete d;
        return jb_fail_(nullptr, rc, jb_last_error(nullptr));
      }
  }
  *out = d;
  return JB_OK;
}

extern "C" void jb_batch_decoder_destroy(jb_batch_decoder *d) {
  if (!d) return;
  for (auto &l : d->lanes) l.release();
  jb_pinned_free(d->arena.base);
  delete d;
}

extern "C" int jb_batch_decoder_set_arena(jb_batch_decoder *d, size_t bytes) {
  if (!d) return jb_fail_(nullptr, JB_ERR_NULL, "jb_batch_decoder_set_arena: decoder is NULL");
  jb_pinned_free(d->arena.base);
  d->arena.base = nullptr;
  d->arena.bytes = 0;
  d->arena.used = 0;
  if (bytes) {
    d->arena.base = (uint8_t *)jb_pinned_alloc(bytes);
    if (!d->arena.base) return jb_fail_(nullptr, JB_ERR_HIP, "jb_batch_decoder_set_arena: pinned allocation failed");
    d->arena.bytes = bytes;
    // the per-thread pixel staging is not needed while an arena takes the pixels
    for (auto &l : d->lanes) {
      for (int k = 0; k < kSlots; k++) {
        jb_pinned_free(l.out[k]);
        l.out[k] = nullptr;
      }
      l.has_out = false;
    }
  }
  return JB_OK;
}

extern "C" int jb_batch_decoder_run(jb_batch_decoder *d, const char *const *paths, int n_paths,
                                    uint8_t **rgb, int32_t *widths, int32_t *heights, int *statuses,
                                    double *times) {
  if (!d || !paths || !rgb || !widths || !heights || !statuses)
    return jb_fail_(nullptr, JB_ERR_NULL, "jb_batch_decoder_run: NULL pointer");
  if (n_paths < 0) return jb_fail_(nullptr, JB_ERR_GEOMETRY, "jb_batch_decoder_run: negative count");
  int n_threads = (int)d->lanes.size();
  if (n_threads > n_paths && n_paths > 0) n_threads = n_paths;
  const int inner_threads = n_threads > 0 ? (int)d->lanes.size() / n_threads : 1;
  Totals tot;
  d->arena.used = 0;  // the previous run's images are released
  const double t0 = now_s();
  std::vector<std::thread> th;
  for (int t = 0; t < n_threads; t++)
    th.emplace_back(worker, &d->lanes[(size_t)t], &d->arena, t, n_threads, inner_threads, paths, n_paths, rgb, widths, heights,
                    statuses, &tot);
  for (auto &x : th) x.join();
  if (times) {
    times[0] = now_s() - t0;
    times[1] = tot.t_entropy;
    times[2] = tot.t_device;
    times[3] = tot.t_read;
  }
  if (tot.first_error != JB_OK) return jb_fail_(nullptr, tot.first_error, tot.first_error_text.c_str());
  return JB_OK;
}

extern "C" int jb_decode_batch(int device_id, const char *const *paths, int n_paths, int n_threads,
                               uint8_t **rgb, int32_t *widths, int32_t *heights, int *statuses,
                               double *times) {
  if (!paths || !rgb || !widths || !heights || !statuses)
    return jb_fail_(nullptr, JB_ERR_NULL, "jb_decode_batch: NULL pointer");
  if (n_threads > n_paths && n_paths > 0) n_threads = n_paths;
  jb_batch_decoder *d = nullptr;
  int rc = jb_batch_decoder_create(device_id, n_threads, 0, 0, &d);  // lanes size themselves
  if (rc) return rc;
  rc = jb_batch_decoder_run(d, paths, n_paths, rgb, widths, heights, statuses, times);
  jb_batch_decoder_destroy(d);
  return rc;
}
