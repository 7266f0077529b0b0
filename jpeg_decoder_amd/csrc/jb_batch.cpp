// jb_batch.cpp -- multi-threaded decode(path) over a batch of files (include/jpegblk.h,
// jb_decode_batch / jb_batch_decoder_*).  The reference decodes one file per process,
// single-threaded (jpeg.cpp:916-929); images are independent, so the batch parallelises by image.
//
// Structure: N host threads run the entropy decoder (the end-to-end bottleneck), each into its
// own pinned coefficient buffers, and submit to ONE shared jb_ctx (under a mutex) whose staging
// ring uploads + computes on one stream and downloads on another, so the link runs full duplex
// (97 GB/s instead of 57).  One context for all host threads rather than one per thread: 16 x 2
// streams on the few hardware queues of a process block each other (measured on 8192x8192 4:2:0:
// 65 images/s, against 120 for one single-stream context per thread).  There is no device
// thread: the host thread that finished an image is running by definition, so a submission never
// waits for the scheduler; a full ring blocks the submitter (back-pressure), and every thread
// waits for its own images outside the mutex.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <exception>
#include <thread>
#include <vector>

#include <sched.h>

#include <memory>

#include "../../include/jpegblk.h"
#include "jb_hostmem.h"
#include "jb_huff.h"
#include "jb_knobs.h"

struct jb_ctx;
int jb_fail_(jb_ctx *ctx, int code, const char *msg);
// device-side entropy decoding: several prepared images of one geometry in one submission; the
// images' status words (0 = decoded cleanly) are copied to `status_out` (pinned) with the pixels
// (dst_device: `rgb` is device memory of the context's device, nothing is downloaded)
extern "C" int jb_submit_packed_(jb_ctx *ctx, const jb_image_desc *desc, const uint16_t *qtabs, const uint8_t *packed,
                                 const JbHuffLayout *lay, uint8_t *rgb, uint32_t *status_out, int *ticket, int dst_device);
extern "C" int jb_submit_batch_dev_(jb_ctx *ctx, const jb_image_desc *desc, int n_images, const int16_t *coef, const uint16_t *qtabs,
                                    uint8_t *d_rgb, int *ticket);
// jb_wait in two halves, so that many threads can wait on one shared context (jb_api.cpp):
// under the caller's lock, the event to block on (nullptr: the submission has completed) ...
void *jb_wait_begin_(jb_ctx *ctx, int ticket);
// ... and the blocking part, without the lock
int jb_wait_block_(jb_ctx *ctx, void *event);
// binds the calling thread to the CPUs of the NUMA node closest to a device (jb_api.cpp)
int jb_bind_thread_near_device_(int device, int numa_knob);
extern "C" void jb_ctx_set_download_age_(jb_ctx *ctx, uint64_t age);
// JB_OK when [p, p + bytes) is device memory of `device` (jb_api.cpp)
int jb_check_device_region_(int device, const void *p, size_t bytes);

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

// The first `limit` bytes of a file (all of it when it is no longer than that); *whole says which.
bool read_prefix(const char *path, size_t limit, std::vector<uint8_t> &buf, bool *whole) {
  FILE *f = fopen(path, "rb");
  if (!f) return false;
  buf.resize(limit + 1);
  const size_t got = fread(buf.data(), 1, limit + 1, f);
  const bool bad = ferror(f) != 0;
  fclose(f);
  if (bad) return false;
  *whole = got <= limit;
  buf.resize(got < limit ? got : (*whole ? got : limit));
  return true;
}

// CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota
// (cgroup v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us).  More entropy threads than that only
// time-slice against each other and against the HIP runtime's own threads: measured on a 16-CPU
// quota, 24-32 threads halved the rate of 16 (8192x8192: 187 -> 98-110 images/s).
int available_cpus() {
  int n = 0;
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);
  if (n < 1) n = (int)std::thread::hardware_concurrency();
  if (n < 1) n = 1;
  long quota = -1, period = 100000;
  if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[32] = {0};
    if (fscanf(f, "%31s %ld", q, &period) >= 1 && strcmp(q, "max") != 0) quota = atol(q);
    fclose(f);
  } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
    if (fscanf(g, "%ld", &quota) != 1) quota = -1;
    fclose(g);
    if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
      if (fscanf(h, "%ld", &period) != 1) period = 100000;
      fclose(h);
    }
  }
  if (quota > 0 && period > 0) {
    const int by_quota = (int)((quota + period - 1) / period);
    if (by_quota >= 1 && by_quota < n) n = by_quota;
  }
  return n;
}

struct Parsed {
  std::vector<uint8_t> bytes;
  bool loaded = false;  // `bytes` is the whole file (pass 1 reads only the head of a large file)
  jb_image_desc desc;
  jb_geometry geo;
  uint16_t qtabs[256];
  int status = JB_OK;
  std::string error;
  bool have = true;  // headers parsed (false: a run without pass 1 parses a file when its group is formed)
};

constexpr int kMaxSlots = 4;
// pinned buffers per host thread = its submissions in flight: decode group k while k-1 is on the device.  Two, whatever
// the output: with the host's entropy stage the second slot is what lets a thread decode while its last group uploads;
// with the device's, 16 threads x 1 already keep the device and the link busy, and more than two changes nothing
// (profiles/r03/ab_slots_per_thread.txt: 1 / 2 / 3 / 4 slots, 1,024 1080p files left in HBM 42.5-43.3 / 41.1-41.5 /
// 41.1-41.9 / 40.4-40.6 k images/s, to the arena 8,250 / 8,180-8,260 / 8,150-8,170 / 8,160-8,200; 32 8192x8192 files
// left in HBM 2,060-2,170 / 1,730-1,810 / 1,750-1,780 / 1,760-1,770).  (Round 2's kernels wanted four for
// device-resident output -- +35 % -- because a group's launches took milliseconds: profiles/r02b/ab_pipeline_depth.txt.)
int lane_slots(bool /*device_output*/) { return 2; }

// what one host thread owns across runs: the pinned buffers its Huffman stage decodes into and,
// when the caller's pixel buffers are pageable (no arena), pinned pixel staging -- a
// device-to-host copy into pageable memory would block the device thread until the kernel has run
struct Lane {
  int16_t *coef[kMaxSlots] = {};
  uint8_t *out[kMaxSlots] = {};
  uint32_t *status[kMaxSlots] = {};  // per image of a group decoded on the device: its status word
  uint8_t *blob[kMaxSlots] = {};     // pinned: a device-entropy group packed for upload (jb_huff_pack_)
  size_t blob_cap[kMaxSlots] = {};
  int device_of_blobs = 0;
  // `need` bytes now; `full` = what a group of the full size would need by this group's bytes per image (the first
  // groups of a thread are smaller: a blob sized for one of them would be re-pinned, milliseconds, when a full group
  // comes to the slot)
  uint8_t *ensure_blob(int device, int s, size_t need, size_t full) {
    if (need <= blob_cap[s]) return blob[s];
    jb_pinned_free(blob[s]);
    blob_cap[s] = 0;
    if (full < need) full = need;
    const size_t cap = full + full / 4 + 65536;
    blob[s] = (uint8_t *)jb_pinned_alloc_on(device, cap);
    if (blob[s]) blob_cap[s] = cap;
    return blob[s];
  }
  size_t cap_coef = 0, cap_rgb = 0;
  bool has_out = false;
  int n_alloc = 0;  // slots with buffers
  int device_of_bufs = 0;

  // `device`: the GPU this lane's decoder drives -- the buffers are pinned against it and come from
  // its NUMA node, whatever device the allocating thread has current (a fresh std::thread: 0)
  // The coefficient staging of a slot is only pinned when the HOST entropy decoder first needs it (host
  // mode, or an image the device decoder handed back): with the entropy stage on the device it is never
  // touched -- 16 threads x 4 slots x 100 MB for 8192x8192 files, most of a second of page pinning.
  int16_t *coef_at(int s) {
    if (!coef[s]) coef[s] = (int16_t *)jb_pinned_alloc_on(device_of_bufs, cap_coef);
    return coef[s];
  }
  bool satisfied(size_t need_coef, size_t need_rgb, bool with_out, int n_slots) const {
    return status[0] && need_coef <= cap_coef && need_rgb <= cap_rgb && (has_out || !with_out) && n_alloc >= n_slots;
  }
  int ensure(int device, size_t need_coef, size_t need_rgb, bool with_out, int n_slots) {
    if (satisfied(need_coef, need_rgb, with_out, n_slots)) return JB_OK;
    if (need_coef < cap_coef) need_coef = cap_coef;
    if (need_rgb < cap_rgb) need_rgb = cap_rgb;
    release();
    int rc = JB_OK;
    device_of_bufs = device;
    for (int s = 0; s < n_slots && rc == JB_OK; s++) {
      if (rc == JB_OK && !status[s]) {
        status[s] = (uint32_t *)jb_pinned_alloc_on(device, 4 * 256);
        if (!status[s]) rc = JB_ERR_HIP;
      }
      if (with_out && rc == JB_OK) {
        out[s] = (uint8_t *)jb_pinned_alloc_on(device, need_rgb);
        if (!out[s]) rc = JB_ERR_HIP;
      }
    }
    if (rc == JB_OK) {
      cap_coef = need_coef;
      cap_rgb = need_rgb;
      has_out = with_out;
      n_alloc = n_slots;
    } else {
      release();
    }
    return rc;
  }
  void drop_out() {
    for (int s = 0; s < kMaxSlots; s++) {
      jb_pinned_free(out[s]);
      out[s] = nullptr;
    }
    has_out = false;
  }
  void release() {
    for (int s = 0; s < kMaxSlots; s++) {
      jb_pinned_free(coef[s]);
      coef[s] = nullptr;
      jb_pinned_free(status[s]);
      status[s] = nullptr;
      jb_pinned_free(blob[s]);
      blob[s] = nullptr;
      blob_cap[s] = 0;
    }
    drop_out();
    cap_coef = cap_rgb = 0;
    n_alloc = 0;
  }
};

// Pinned output arena (optional, jb_batch_decoder_set_arena): images are placed by an atomic bump
// pointer, so the device writes every pixel straight to its final place.
struct Arena {
  uint8_t *base = nullptr;
  size_t bytes = 0;
  bool on_device = false;  // the caller's DEVICE memory (jb_batch_decoder_set_device_output): the pixels stay in HBM
  bool owned = true;       // pinned memory this decoder allocated
  std::atomic<size_t> used{0};
  uint8_t *take(size_t n) {
    n = (n + 255) & ~(size_t)255;
    size_t at = used.fetch_add(n);
    if (at + n > bytes) return nullptr;
    return base + at;
  }
};

// the one context all host threads submit to
struct Shared {
  std::mutex mu;
  jb_ctx *ctx = nullptr;
};

struct Totals {
  std::mutex mu;
  double t_entropy = 0, t_device = 0, t_read = 0;
  double first_submit = 1e30, last_submit = 0, first_back = 1e30, first_thread_done = 1e30;  // JPEGBLK_TIMING=3
  int first_error = JB_OK;
  std::string first_error_text;
};

struct Run {
  int device;
  size_t dev_cap_coef, dev_cap_rgb;  // ring-slot capacity: bounds a group whose entropy stage runs on the device
  int n_slots;                       // submissions a host thread keeps in flight (lane_slots)
  const char *const *paths;
  int n_paths, n_threads, inner_threads;
  const std::vector<std::vector<int>> *lists;  // which files each host thread owns (indices into paths)
  uint8_t **rgb;
  int32_t *widths, *heights;
  int *statuses;
  Arena *arena;
  Shared *dev;
  Totals *tot;
  const JbKnobs *knobs;  // the decoder's (jb_knobs.h)
  // no pass 1: every file is read and parsed when its group is formed; an image the decoder's buffers do not hold
  // is left for a second, classic round (deferred[t]: positions in thread t's list)
  bool lazy = false;
  std::vector<std::vector<int>> *deferred = nullptr;
  size_t slot_coef = 0, slot_rgb = 0;  // what a ring slot holds (one image may be larger than a group's bound, not than this)
};

// pass 1 (per host thread): parse the headers of its files, so that the buffers can be sized once for
// the whole batch.  Only the head of a file is read here (kHeadBytes: the tables and the frame header
// of an ordinary file come long before that); the rest is read in pass 2, group by group, while the
// device works on the groups before -- reading a batch of 1,024 one-megabyte files whole took 30 ms
// of a 200 ms batch during which the device had nothing to do, and held every file in memory at once.
// A head that does not parse cleanly (headers longer than the head, progressive and multi-scan files,
// errors) is settled on the whole file, so every status is the one the whole file gives.
// (Two head sizes: the headers of a plain baseline file end within a few hundred bytes, so 4 KB are read first --
// 3 us instead of 5 per file on one thread; files with large APPn segments -- EXIF, ICC profiles -- get the 64 KB,
// then the whole file.  What pass 1 costs in a batch is opening and closing the files, not reading them: 1,024 files
// on 16 threads take 3.6-4.0 ms with either head size when the batch repeats eight files, as the benches do, every
// thread opening the same inodes -- 59 us per file against 12 us for one thread alone.)
constexpr size_t kHeadBytes[2] = {(size_t)4 << 10, (size_t)64 << 10};

void parse_one(Parsed &p) {
  p.status = jb_entropy_decode(p.bytes.data(), p.bytes.size(), &p.desc, p.qtabs, nullptr, 0);
  if (p.status == JB_OK) p.status = jb_geometry_of(&p.desc, &p.geo);
  if (p.status != JB_OK) p.error = jb_last_error(nullptr);
}

void parse_pass(const Run &r, int t, std::vector<Parsed> &parsed, size_t *max_coef, size_t *max_rgb, double *t_read) {
  jb_bind_thread_near_device_(r.device, r.knobs->numa);  // the file bytes are first touched here: keep them on the GPU's node
  for (size_t k = 0; k < parsed.size(); k++) {
    const int i = (*r.lists)[(size_t)t][k];
    Parsed &p = parsed[k];
    bool ok = true;
    p.status = JB_ERR_FORMAT;
    p.loaded = false;
    for (int level = 0; level < 3 && ok && p.status != JB_OK && !p.loaded; level++) {  // heads, then the whole file decides
      const double a = now_s();
      ok = level < 2 ? read_prefix(r.paths[i], kHeadBytes[level], p.bytes, &p.loaded) : read_file(r.paths[i], p.bytes);
      *t_read += now_s() - a;
      if (level == 2) p.loaded = ok;
      if (ok) parse_one(p);
    }
    if (!ok) {
      p.status = JB_ERR_FORMAT;
      p.error = "cannot read file";
      p.bytes.clear();
      continue;
    }
    if (p.status != JB_OK) continue;
    if (!p.loaded) {
      p.bytes.clear();
      p.bytes.shrink_to_fit();
    }
    if ((size_t)p.geo.coef_bytes > *max_coef) *max_coef = (size_t)p.geo.coef_bytes;
    if ((size_t)p.geo.rgb_bytes > *max_rgb) *max_rgb = (size_t)p.geo.rgb_bytes;
  }
}

// Small images are submitted in GROUPS: consecutive images of one geometry share one upload, one
// launch and one download (jb_submit_batch).  A submission costs tens of microseconds of driver
// calls under the shared mutex; at one submission per 679x451 image that capped the decoder at
// 11,000 images/s whatever the thread count, five times below what the entropy stage delivers.
constexpr size_t kGroupBytes = (size_t)16 << 20;  // coefficient bytes per group (and per pinned buffer)
constexpr int kMaxGroup = 64;

// pass 2 (per host thread): decode a group of images into pinned slot g%2 and submit it; before a
// slot is reused, the group that used it two steps ago is finished (it has long been through the
// device by then: entropy decoding takes ~10x the transfers)
void decode_pass(const Run &r, Lane *lane, int t, std::vector<Parsed> &parsed, int setup_rc, const std::string &setup_text) {
  jb_bind_thread_near_device_(r.device, r.knobs->numa);
  const bool use_arena = r.arena && r.arena->base;
  const bool to_device = use_arena && r.arena->on_device;
  double t_entropy = 0, t_wait = 0, t_read = 0;
  double first_submit = 1e30, last_submit = 0, first_back = 1e30;  // (JPEGBLK_TIMING=3)
  // the rest of a file whose head was parsed in pass 1
  auto load = [&](Parsed &p, int i) {
    if (p.loaded || p.status != JB_OK) return;
    const double a = now_s();
    if (read_file(r.paths[i], p.bytes)) {
      p.loaded = true;
    } else {
      p.status = JB_ERR_FORMAT;
      p.error = "cannot read file";
      p.bytes.clear();
    }
    t_read += now_s() - a;
  };
  // (a run without pass 1) headers of file k of this thread's list, from the whole file; -> false: the image does
  // not fit the buffers this decoder has (it waits for the second round)
  auto ready = [&](int k) {
    Parsed &p = parsed[(size_t)k];
    if (!p.have) {
      const int i = (*r.lists)[(size_t)t][(size_t)k];
      const double a = now_s();
      const bool ok = read_file(r.paths[i], p.bytes);
      t_read += now_s() - a;
      p.have = true;
      p.loaded = ok;
      if (ok) {
        parse_one(p);
      } else {
        p.status = JB_ERR_FORMAT;
        p.error = "cannot read file";
        p.bytes.clear();
      }
    }
    if (!r.lazy || p.status != JB_OK) return true;
    const size_t c = (size_t)p.geo.coef_bytes, x = (size_t)p.geo.rgb_bytes;
    return c <= lane->cap_coef && x <= lane->cap_rgb && c <= r.slot_coef && x <= r.slot_rgb;
  };
  struct Group {
    int ticket = -1, first = -1, n = 0;  // images first .. first+n-1 of this thread's list
    bool on_device = false;              // the group's entropy stage ran on the device (jb_huff.hip)
  } grp[kMaxSlots];
  const int kSlots = r.n_slots;
  // Where the entropy stage runs.  The batch decoder's default is the DEVICE for every baseline image
  // the device decoders take (restart intervals: one lane per interval; none: the self-synchronising
  // decoder), 16 intervals / chunks or more: measured against 16 host threads it is 1.06x (8,192 small
  // images) to 2.6x (4096x4096 files) as fast end to end for a tenth of the host CPU time (DESIGN.md
  // section 9); this thread then only parses, removes the byte stuffing and packs.  What the device
  // decoders refuse or flag goes through the host decoder, image by image.  JPEGBLK_GPU_HUFFMAN=0
  // keeps the entropy stage on the host threads (north_star's split), =2 drops the 16-interval
  // threshold.  A group is all-device or all-host.
  // (A hybrid -- a quarter of the threads feeding the device decoder with 60 % of the files, the rest
  // decoding on the host -- was measured and is slower than either pure mode: 1,885 images/s against
  // 2,681 host / 2,498 device on PIL 1080p files; large group downloads and many small uploads and
  // downloads at once share the link badly.  Removed.)
  const bool dev_entropy = r.knobs->gpu_huffman != 0;
  const uint32_t min_intervals = r.knobs->gpu_huffman == 2 ? 1u : 16u;
  const int dev_max_group = kMaxGroup;  // images per device-entropy group
  std::vector<uint16_t> qtabs;
  auto index_of = [&](int k) { return (*r.lists)[(size_t)t][(size_t)k]; };
  auto report = [&](int i, int st, const std::string &text) {
    r.statuses[i] = st;
    if (st == JB_OK) return;
    if (!use_arena) jb_free(r.rgb[i]);
    r.rgb[i] = nullptr;
    std::lock_guard<std::mutex> g(r.tot->mu);
    if (r.tot->first_error == JB_OK) {
      r.tot->first_error = st;
      r.tot->first_error_text = std::string(r.paths[i]) + ": " + text;
    }
  };
  auto finish_slot = [&](int s) {
    Group &g = grp[s];
    if (g.n == 0) return;
    double a = now_s();
    void *ev;
    {
      std::lock_guard<std::mutex> lk(r.dev->mu);
      ev = jb_wait_begin_(r.dev->ctx, g.ticket);
    }
    const int st = ev ? jb_wait_block_(r.dev->ctx, ev) : JB_OK;
    const std::string text = st == JB_OK ? "" : jb_last_error(nullptr);
    const size_t rgb_bytes = (size_t)parsed[(size_t)g.first].geo.rgb_bytes;
    for (int j = 0; j < g.n; j++) {
      const int i = index_of(g.first + j);
      Parsed &p = parsed[(size_t)(g.first + j)];
      int st_j = st;
      std::string text_j = text;
      uint8_t *const staged = use_arena ? r.rgb[i] : lane->out[s] + (size_t)j * rgb_bytes;
      if (st == JB_OK && g.on_device && lane->status[s][j] != 0) {
        // the device decoder met data it calls corrupt: the host decoder is the authority -- this one
        // image again, entropy stage on the host (the slot's coefficient buffer is free: the group is done)
        int16_t *const cbuf = lane->coef_at(s);
        st_j = cbuf ? jb_entropy_decode(p.bytes.data(), p.bytes.size(), &p.desc, p.qtabs, cbuf, lane->cap_coef) : (int)JB_ERR_HIP;
        if (st_j == JB_OK) {
          int ticket = -1;
          void *ev2 = nullptr;
          {
            std::lock_guard<std::mutex> lk(r.dev->mu);
            st_j = to_device ? jb_submit_batch_dev_(r.dev->ctx, &p.desc, 1, cbuf, p.qtabs, staged, &ticket)
                             : jb_submit_batch(r.dev->ctx, &p.desc, 1, cbuf, p.qtabs, staged, &ticket);
            if (st_j == JB_OK) ev2 = jb_wait_begin_(r.dev->ctx, ticket);
            else text_j = jb_last_error(r.dev->ctx);
          }
          if (st_j == JB_OK && ev2) st_j = jb_wait_block_(r.dev->ctx, ev2);
        } else {
          text_j = jb_last_error(nullptr);
        }
      }
      if (g.on_device) {
        p.bytes.clear();
        p.bytes.shrink_to_fit();
      }
      if (st_j == JB_OK && !use_arena)  // pinned staging -> the caller's (pageable) buffer
        memcpy(r.rgb[i], staged, rgb_bytes);
      report(i, st_j, text_j);
    }
    t_wait += now_s() - a;
    if (first_back > 1e29) first_back = now_s();
    g.n = 0;
  };
  auto same_geometry = [](const Parsed &a, const Parsed &b) {
    return a.desc.width == b.desc.width && a.desc.height == b.desc.height && a.desc.hs == b.desc.hs &&
           a.desc.vs == b.desc.vs && a.desc.qtab_id[0] == b.desc.qtab_id[0] &&
           a.desc.qtab_id[1] == b.desc.qtab_id[1] && a.desc.qtab_id[2] == b.desc.qtab_id[2];
  };
  const int n_mine = (int)parsed.size();
  int k = 0, slot = 0;
  int dev_groups = 0;  // device-entropy groups this thread has submitted
  while (k < n_mine) {
    Parsed &head = parsed[(size_t)k];
    if (!ready(k)) {  // (a run without pass 1) larger than anything this decoder is sized for: the second round's
      (*r.deferred)[(size_t)t].push_back(k);
      head.bytes.clear();
      head.bytes.shrink_to_fit();
      k++;
      continue;
    }
    load(head, index_of(k));
    r.rgb[index_of(k)] = nullptr;
    r.widths[index_of(k)] = r.heights[index_of(k)] = 0;
    if (head.status != JB_OK || setup_rc != JB_OK) {  // rejected in pass 1, or nothing could be set up
      report(index_of(k), head.status != JB_OK ? head.status : setup_rc, head.status != JB_OK ? head.error : setup_text);
      head.bytes.clear();
      head.bytes.shrink_to_fit();
      k++;
      continue;
    }
    const int s = slot;
    finish_slot(s);
    const size_t coef_bytes = (size_t)head.geo.coef_bytes, rgb_bytes = (size_t)head.geo.rgb_bytes;
    // a group must fit the coefficient AND the pixel capacity (pinned lane buffers and ring slots are
    // sized from the batch's largest image: rgb/coef is 0.5 for 4:4:4 and 1.0 for 4:2:0, so many
    // small 4:2:0 images next to one large 4:4:4 image are bounded by the pixel side)
    size_t room_c = lane->cap_coef / coef_bytes, room_p = lane->cap_rgb / rgb_bytes;
    int room = (int)(room_c < room_p ? room_c : room_p);
    if (room > kMaxGroup) room = kMaxGroup;
    if (room < 1) room = 1;  // (cannot happen: the capacities cover the largest single image)
    // a group decoded on the device does not pass through this thread's pinned coefficient buffer:
    // it is bounded by the ring slot (and, without an arena, by the pinned pixel staging)
    size_t droom_c = r.dev_cap_coef / coef_bytes, droom_p = (use_arena ? r.dev_cap_rgb : lane->cap_rgb) / rgb_bytes;
    int room_dev = (int)(droom_c < droom_p ? droom_c : droom_p);
    if (room_dev > dev_max_group) room_dev = dev_max_group;
    const int room_dev_full = room_dev < 1 ? 1 : room_dev;
    // (JPEGBLK_GROUP_RAMP=1: the thread's first two device groups a quarter and a half of the full size, so that the
    // device gets its first work sooner.  It paid while every run began with a pass over all headers; since files
    // are parsed as their groups are formed the first submission leaves after 1 ms anyway, and fewer, larger groups
    // win: 1,024 1080p files left in HBM 33-35 k -> 41.6-42.1 k images/s, 128 files 21 k -> 26 k, 128 files to the
    // arena 6,700 -> 7,240, interleaved on one box.  Off by default.)
    if (dev_groups < 2 && r.knobs->group_ramp) room_dev = room_dev >> (2 - dev_groups);
    if (room_dev < 1) room_dev = 1;
    // entropy-decode consecutive images of the head's geometry into the slot, back to back -- or,
    // for files with restart intervals, only ready them for the device decoder
    int n = 0;
    bool on_device = false;
    std::vector<std::unique_ptr<JbHuffJob>> jobs;
    while (n < (on_device ? room_dev : room) && k + n < n_mine) {
      Parsed &p = parsed[(size_t)(k + n)];
      if (n > 0 && !ready(k + n)) break;  // (the next group's head: it is set aside there)
      if (n > 0 && (p.status != JB_OK || !same_geometry(head, p))) break;
      load(p, index_of(k + n));
      if (p.status != JB_OK) break;  // (n > 0: the head was loaded above; the next group reports it)
      double a = now_s();
      std::unique_ptr<JbHuffJob> job;
      bool eligible = false;
      if (dev_entropy) {
        job.reset(new JbHuffJob());
        eligible = jb_huff_prepare_(p.bytes.data(), p.bytes.size(), job.get(), nullptr, r.knobs->chunk_bytes) == JB_OK && jb_huff_worth_it_(*job, min_intervals);
        // the frame this pass reads must be the frame pass 1 sized the group for (a file that changed in between):
        // else the host path below settles the image, with its own capacity checks
        if (eligible && (job->desc.width != p.desc.width || job->desc.height != p.desc.height || job->desc.hs != p.desc.hs ||
                         job->desc.vs != p.desc.vs || memcmp(job->desc.qtab_id, p.desc.qtab_id, sizeof p.desc.qtab_id) != 0))
          eligible = false;
      }
      if (n == 0) on_device = eligible;
      else if (eligible != on_device) break;  // the next group starts with this image
      if (on_device) {
        memcpy(p.qtabs, job->qtabs, sizeof p.qtabs);
        jobs.push_back(std::move(job));
        t_entropy += now_s() - a;
        n++;  // (the file bytes stay until the group has come back clean)
        continue;
      }
      // fewer files than host threads: the spare threads split each image's restart intervals
      int16_t *const cbuf = lane->coef_at(s);
      int st = cbuf ? jb_entropy_decode_mt(p.bytes.data(), p.bytes.size(), &p.desc, p.qtabs,
                                           cbuf + (size_t)n * (coef_bytes / 2), coef_bytes, r.inner_threads)
                    : (int)JB_ERR_HIP;
      t_entropy += now_s() - a;
      p.bytes.clear();
      p.bytes.shrink_to_fit();
      if (st != JB_OK) {  // a corrupt scan: it leaves the group, the group ends before it
        p.status = st;
        p.error = jb_last_error(nullptr);
        if (n == 0) {
          r.rgb[index_of(k)] = nullptr;
          report(index_of(k), st, p.error);
          k++;
        }
        break;
      }
      n++;
    }
    if (n == 0) continue;
    // where the pixels go
    uint8_t *dst = use_arena ? r.arena->take((size_t)n * rgb_bytes) : lane->out[s];
    int st = JB_OK;
    std::string text;
    if (!dst) {
      st = JB_ERR_CAPACITY;
      text = "output arena exhausted";
    }
    qtabs.resize((size_t)n * 256);
    for (int j = 0; j < n; j++) {
      const int i = index_of(k + j);
      Parsed &p = parsed[(size_t)(k + j)];
      memcpy(&qtabs[(size_t)j * 256], p.qtabs, sizeof p.qtabs);
      r.widths[i] = p.desc.width;
      r.heights[i] = p.desc.height;
      r.rgb[i] = nullptr;
      if (st == JB_OK) {
        r.rgb[i] = use_arena ? dst + (size_t)j * rgb_bytes : jb_alloc_pixels_(rgb_bytes);
        if (!r.rgb[i]) {
          st = JB_ERR_CAPACITY;
          text = "out of memory";
        }
      }
    }
    JbHuffLayout lay;
    if (st == JB_OK && on_device) {  // pack the group into this thread's pinned blob -- outside the shared lock
      std::vector<const JbHuffJob *> ptrs;
      for (auto &j : jobs) ptrs.push_back(j.get());
      const size_t blob_bytes = jb_huff_pack_size_(ptrs.data(), n);
      uint8_t *blob = lane->ensure_blob(r.device, s, blob_bytes, blob_bytes / (size_t)n * (size_t)room_dev_full);
      if (!blob) {
        st = JB_ERR_HIP;
        text = "pinned host allocation failed";
      } else if ((st = jb_huff_pack_(ptrs.data(), n, (int64_t)coef_bytes, blob, &lay)) != JB_OK) {
        text = "submission too large for the device entropy decoder";
      }
      jobs.clear();  // (everything the jobs held is in the blob now)
    }
    if (st == JB_OK) {
      double a = now_s();
      // every copy of the submission is pinned <-> device, so this returns at once and the
      // transfers and the kernel run while this thread decodes its next group
      std::lock_guard<std::mutex> lk(r.dev->mu);
      if (on_device) {
        st = jb_submit_packed_(r.dev->ctx, &head.desc, qtabs.data(), lane->blob[s], &lay, dst, lane->status[s], &grp[s].ticket, to_device);
      } else if (to_device) {
        st = jb_submit_batch_dev_(r.dev->ctx, &head.desc, n, lane->coef[s], qtabs.data(), dst, &grp[s].ticket);
      } else {
        st = jb_submit_batch(r.dev->ctx, &head.desc, n, lane->coef[s], qtabs.data(), dst, &grp[s].ticket);
      }
      if (st != JB_OK) text = jb_last_error(r.dev->ctx);
      t_wait += now_s() - a;
      last_submit = now_s();
      if (first_submit > 1e29) first_submit = last_submit;
    }
    if (st != JB_OK) {
      for (int j = 0; j < n; j++) report(index_of(k + j), st, text);
    } else {
      grp[s].first = k;
      grp[s].n = n;
      grp[s].on_device = on_device;
      if (on_device) dev_groups++;
      slot = (slot + 1) % kSlots;
    }
    k += n;
  }
  for (int j = 0; j < kSlots; j++) finish_slot((slot + j) % kSlots);  // oldest first
  std::lock_guard<std::mutex> g(r.tot->mu);
  r.tot->t_entropy += t_entropy;
  r.tot->t_device += t_wait;
  r.tot->t_read += t_read;
  if (first_submit < r.tot->first_submit) r.tot->first_submit = first_submit;
  if (last_submit > r.tot->last_submit) r.tot->last_submit = last_submit;
  if (first_back < r.tot->first_back) r.tot->first_back = first_back;
  const double done = now_s();
  if (done < r.tot->first_thread_done) r.tot->first_thread_done = done;
}

}  // namespace

struct jb_batch_decoder {
  JbKnobs knobs = jb_knobs_read();  // the environment as it was when the decoder was created (jb_knobs.h)
  int device = 0;
  std::vector<Lane> lanes;
  Arena own_arena;
  Arena *arena = &own_arena;  // a part of a multi-device decoder points at its parent's arena
  jb_ctx *ctx = nullptr;      // the one context all host threads of this device submit to
  size_t ctx_coef = 0, ctx_rgb = 0;
  int ctx_ring = 0;
  // (never more submissions in flight than the context's ring of 64 slots holds: a thread that found the ring full
  // would wait for the oldest submission while holding the lock every other thread submits under)
  int slots() const {
    const int want = lane_slots(arena->base && arena->on_device), fit = lanes.empty() ? want : 64 / (int)lanes.size();
    return want < fit ? want : fit < 1 ? 1 : fit;
  }
  // the caller's device region as it was given (jb_batch_decoder_set_device_output[s]); submit / collect
  // split it between the two sides, run() uses all of it
  uint8_t *region_base = nullptr;
  size_t region_bytes = 0;
  // how this decoder was made: its twin (submit / collect) is made the same way
  std::vector<int> made_devices;
  bool made_multi = false;
  int made_threads = 0;
  size_t made_coef = 0, made_rgb = 0;
  // jb_batch_decoder_submit / _collect: up to two batches in flight, batch k on side k & 1 -- side 0 is this
  // decoder, side 1 its twin (same devices, threads and sizes, its own ring, staging and arena) -- so that the
  // start-up of one batch (headers, first groups) runs under the tail of the other (last kernels, last downloads)
  struct Flight {
    std::thread th;
    bool busy = false;
    int ticket = -1;
    int rc = JB_OK;
    std::string text;
    double times[4] = {0, 0, 0, 0};
    std::vector<std::string> path_text;  // the batch's paths, copied: the caller's array need not outlive submit
    std::vector<const char *> path_ptr;
  };
  Flight flights[2];
  int tickets = 0;
  jb_batch_decoder *twin = nullptr;
  bool split_for_sides = false;  // the outputs are arranged for submit / collect (twin arena, halves of the regions)
  bool in_flight() const { return flights[0].busy || flights[1].busy; }
  // multi-device decoder (jb_batch_decoder_create_multi): one single-device decoder per listed
  // device; this object then only deals the files out and owns the shared arena
  std::vector<jb_batch_decoder *> parts;

  int ensure_ctx(size_t need_coef, size_t need_rgb) {
    // ring depth = everything the host threads can have in flight, so that a submission never blocks a
    // thread that could be decoding; device memory is not the scarce resource here (32 slots of
    // 8192x8192 4:2:0 are 13 GB of 288)
    int ring = slots() * (int)lanes.size();
    if (ring > 64) ring = 64;
    if (ctx && need_coef <= ctx_coef && need_rgb <= ctx_rgb && ring <= ctx_ring) return JB_OK;
    if (need_coef < ctx_coef) need_coef = ctx_coef;
    if (need_rgb < ctx_rgb) need_rgb = ctx_rgb;
    jb_ctx_destroy(ctx);
    ctx = nullptr;
    ctx_coef = ctx_rgb = 0;
    ctx_ring = 0;
    int rc = jb_ctx_create(device, need_coef, need_rgb, ring, &ctx);
    if (rc == JB_OK) {
      ctx_coef = need_coef;
      ctx_rgb = need_rgb;
      ctx_ring = ring;
    }
    return rc;
  }
  // size everything for images of up to (coef, rgb) bytes; page pinning is slow, so the host
  // threads' buffers are created in parallel
  int ensure_all(size_t need_coef, size_t need_rgb, int n_lanes, size_t ring_coef = 0, size_t ring_rgb = 0) {
    // (the ring slots may be larger than the lanes' pinned buffers: groups decoded on the device)
    int rc = ensure_ctx(ring_coef > need_coef ? ring_coef : need_coef, ring_rgb > need_rgb ? ring_rgb : need_rgb);
    if (rc != JB_OK) return rc;
    const bool with_out = !arena->base;
    {
      // nothing to make (every run after the first): no threads either -- sixteen of them started and joined for
      // nothing took half a millisecond of every run
      bool all = true;
      const int n_slots_now = slots();
      for (int i = 0; i < n_lanes && all; i++) all = lanes[(size_t)i].satisfied(need_coef, need_rgb, with_out, n_slots_now);
      if (all) return JB_OK;
    }
    std::vector<std::thread> th;
    std::vector<int> rcs((size_t)n_lanes, JB_OK);
    const int dev = device;
    const int n_slots = slots();
    for (int i = 0; i < n_lanes; i++)
      th.emplace_back([&, i, n_slots] { rcs[(size_t)i] = lanes[(size_t)i].ensure(dev, need_coef, need_rgb, with_out, n_slots); });
    for (auto &x : th) x.join();
    for (int r : rcs)
      if (r != JB_OK) return jb_fail_(nullptr, r, "pinned host allocation failed");
    return JB_OK;
  }
};

namespace {

int clamp_threads(int n_threads, const JbKnobs &knobs) {
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 64) n_threads = 64;  // one ring slot each at least (64 host threads decode 30 Gpixel/s: far beyond the link)
  // no more entropy threads than CPUs this process may use (JPEGBLK_OVERSUBSCRIBE=1 lifts that)
  if (!knobs.oversubscribe && n_threads > available_cpus()) n_threads = available_cpus();
  return n_threads;
}

int create_single(int device_id, int n_threads, size_t max_coef_bytes, size_t max_rgb_bytes, jb_batch_decoder **out) {
  jb_batch_decoder *d = new jb_batch_decoder();
  d->device = device_id;
  d->lanes.resize((size_t)n_threads);
  // the context exists from the start (also when the buffers are sized lazily), so that a
  // missing or unusable device is reported here
  // With sizes given, everything a batch of such images needs is built HERE -- the staging of the host path
  // (groups of kGroupBytes) and, where the entropy stage runs on the device, ring slots for groups of
  // JPEGBLK_DEV_GROUP_MB -- so that the first run does not spend 45-120 ms growing them (a decoder created
  // with (0, 0) sizes itself from its first batch instead).
  int rc;
  if (max_coef_bytes && max_rgb_bytes) {
    size_t group = kGroupBytes;
    if (d->knobs.group_mb >= 0) group = (size_t)d->knobs.group_mb << 20;
    const size_t mc = max_coef_bytes < group ? group : max_coef_bytes, mr = max_rgb_bytes < group ? group : max_rgb_bytes;
    size_t ring = 0;
    if (d->knobs.gpu_huffman != 0) {
      ring = (size_t)(d->knobs.dev_group_mb >= 0 ? d->knobs.dev_group_mb : 96) << 20;
      if (ring > (size_t)kMaxGroup * mc) ring = (size_t)kMaxGroup * mc;
    }
    rc = d->ensure_all(mc, mr, n_threads, ring, ring);
  } else {
    rc = d->ensure_ctx(128, 192);
  }
  if (rc != JB_OK) {
    std::string text = jb_last_error(nullptr);
    jb_batch_decoder_destroy(d);
    return jb_fail_(nullptr, rc, text.c_str());
  }
  *out = d;
  return JB_OK;
}

// one device's share of a run; `top` = this decoder owns the arena (and recycles it)
int run_single(jb_batch_decoder *d, const char *const *paths, int n_paths, uint8_t **rgb, int32_t *widths,
               int32_t *heights, int *statuses, double *times, bool top) {
  Totals tot;
  Shared dev;
  if (top) d->arena->used = 0;  // the previous run's images are released
  const double t0 = now_s();
  // every pinned buffer and ring slot holds one large image or a group of small ones: the group
  // figure bounds the coefficient and the pixel side alike (rgb_bytes <= coef_bytes in every layout)
  size_t group_bytes = kGroupBytes;
  if (d->knobs.group_mb >= 0) group_bytes = (size_t)d->knobs.group_mb << 20;  // (JPEGBLK_GROUP_MB; 0 = one image per submission)
  // Groups whose entropy stage runs on the device hold about 100 MB of coefficients (JPEGBLK_DEV_GROUP_MB;
  // 8 1080p images, one 8192x8192 image, at most 64 images).  (With the first versions of the device decoder,
  // whose launches took milliseconds whatever their size, large groups paid -- 2,626 images/s in groups of
  // 20, 3,947 in groups of 64 -- and the default was a thread's whole share up to 512 MB.  With today's
  // kernels many small groups in flight are as fast or faster with host output (medians of six runs:
  // 0.171 s against 0.18 s for 1,024 1080p files; 8192x8192 +5 %) and much faster with device-resident
  // output, and the ring slots are a fifth of the size: profiles/r02b/ab_group_size_final.txt.)
  size_t dev_group_bytes = 0;
  if (d->knobs.gpu_huffman != 0) {
    const long mb = d->knobs.dev_group_mb >= 0 ? d->knobs.dev_group_mb : 96;
    dev_group_bytes = mb > 0 ? (size_t)mb << 20 : 0;
  }
  // One round over `files` (indices into paths).  Classic: pass 1 reads every file's headers so that the buffers
  // can be sized once for the round, then pass 2 decodes.  Lazy (a decoder whose buffers exist already: created
  // with sizes, or after an earlier run): no pass 1 -- a file is read once and parsed when its group is formed, so
  // the device has its first work after two files' worth of host time instead of after every header of the batch
  // (1,024 files: 5 ms -> 1 ms to the first submission) -- and an image larger than the buffers is handed back in
  // `left_over` for a classic round.
  auto round = [&](const std::vector<int> &files, bool lazy, std::vector<int> *left_over) {
    int nt = (int)d->lanes.size();
    if (nt > (int)files.size()) nt = (int)files.size();
    if (nt < 1) return;
    std::vector<std::vector<int>> lists((size_t)nt);
    for (size_t j = 0; j < files.size(); j++) lists[j % (size_t)nt].push_back(files[j]);  // file j -> thread j % nt
    std::vector<std::vector<int>> deferred((size_t)nt);
    Run r{d->device, 0, 0, d->slots(), paths, n_paths, nt, (int)d->lanes.size() / nt,
          &lists, rgb, widths, heights, statuses, d->arena, &dev, &tot, &d->knobs};
    r.lazy = lazy;
    r.deferred = &deferred;
    const double tr0 = now_s();
    std::vector<std::vector<Parsed>> parsed((size_t)nt);
    for (int t = 0; t < nt; t++) parsed[(size_t)t].resize(lists[(size_t)t].size());
    int setup_rc = JB_OK;
    std::string setup_text;
    size_t ring_bytes = dev_group_bytes;
    double t_parsed = tr0;
    if (!lazy) {
      // pass 1: headers, in parallel
      std::vector<size_t> mc((size_t)nt, 0), mr((size_t)nt, 0);
      std::vector<double> tr((size_t)nt, 0.0);
      {
        std::vector<std::thread> th;
        for (int t = 0; t < nt; t++)
          th.emplace_back([&, t] { parse_pass(r, t, parsed[(size_t)t], &mc[(size_t)t], &mr[(size_t)t], &tr[(size_t)t]); });
        for (auto &x : th) x.join();
      }
      size_t max_coef = 0, max_rgb = 0;
      for (int t = 0; t < nt; t++) {
        if (mc[(size_t)t] > max_coef) max_coef = mc[(size_t)t];
        if (mr[(size_t)t] > max_rgb) max_rgb = mr[(size_t)t];
        tot.t_read += tr[(size_t)t];
      }
      if (max_coef && max_coef < group_bytes) max_coef = group_bytes;
      if (max_rgb && max_rgb < group_bytes) max_rgb = group_bytes;
      if (ring_bytes) {
        size_t share = (files.size() + (size_t)nt - 1) / (size_t)nt;
        if (share > (size_t)kMaxGroup) share = (size_t)kMaxGroup;
        if (ring_bytes > share * max_coef) ring_bytes = share * max_coef;
      }
      t_parsed = now_s();
      setup_rc = max_coef ? d->ensure_all(max_coef, max_rgb, nt, ring_bytes, ring_bytes) : JB_OK;
      // a device group is bounded by this round's group size (the ring slots may be larger: an earlier run's)
      const size_t group_cap = ring_bytes > max_coef ? ring_bytes : max_coef;
      r.dev_cap_coef = (ring_bytes && group_cap < d->ctx_coef) ? group_cap : d->ctx_coef;
      r.dev_cap_rgb = (ring_bytes && group_cap < d->ctx_rgb) ? group_cap : d->ctx_rgb;
    } else {
      for (auto &list : parsed)
        for (Parsed &p : list) p.have = false;
      // the buffers as they are (this call only re-makes what a change of output mode or ring depth asks for)
      size_t lc = 0, lr = 0;
      for (const Lane &l : d->lanes) {
        if (l.cap_coef > lc) lc = l.cap_coef;
        if (l.cap_rgb > lr) lr = l.cap_rgb;
      }
      setup_rc = d->ensure_all(lc, lr, nt, d->ctx_coef, d->ctx_rgb);
      r.dev_cap_coef = (ring_bytes && ring_bytes < d->ctx_coef) ? ring_bytes : d->ctx_coef;
      r.dev_cap_rgb = (ring_bytes && ring_bytes < d->ctx_rgb) ? ring_bytes : d->ctx_rgb;
    }
    r.slot_coef = d->ctx_coef;
    r.slot_rgb = d->ctx_rgb;
    const double t_setup = now_s();
    if (setup_rc != JB_OK) setup_text = jb_last_error(nullptr);
    // pass 2: entropy decoding on the host threads, all submitting to the shared context
    dev.ctx = d->ctx;
    {
      // of two batches in flight on one device (submit / collect, or two decoders) the older one's downloads go first
      static std::atomic<uint64_t> run_seq{0};
      jb_ctx_set_download_age_(d->ctx, ++run_seq);
    }
    tot.first_submit = tot.first_back = tot.first_thread_done = 1e30;
    tot.last_submit = 0;
    {
      std::vector<std::thread> th;
      for (int t = 0; t < nt; t++)
        th.emplace_back([&, t] { decode_pass(r, &d->lanes[(size_t)t], t, parsed[(size_t)t], setup_rc, setup_text); });
      for (auto &x : th) x.join();
    }
    if (left_over)
      for (int t = 0; t < nt; t++)
        for (int k : deferred[(size_t)t]) left_over->push_back(lists[(size_t)t][(size_t)k]);
    if (d->knobs.timing == 3)
      fprintf(stderr, "run_single %zu files%s: headers %.2f ms, setup %.2f, first submit at %.2f, first group back at %.2f, last submit at %.2f, "
              "first thread done at %.2f, all done at %.2f\n", files.size(), lazy ? " (no pass 1)" : "", (t_parsed - tr0) * 1e3, (t_setup - t_parsed) * 1e3,
              (tot.first_submit - tr0) * 1e3, (tot.first_back - tr0) * 1e3, (tot.last_submit - tr0) * 1e3, (tot.first_thread_done - tr0) * 1e3,
              (now_s() - tr0) * 1e3);
  };
  std::vector<int> all((size_t)(n_paths > 0 ? n_paths : 0));
  for (int i = 0; i < n_paths; i++) all[(size_t)i] = i;
  // without pass 1 when every buffer exists: the ring, and the staging of every lane this run uses
  bool lazy = !d->knobs.pass1 && d->ctx && d->ctx_coef > 0 && d->ctx_rgb > 0;
  for (size_t t = 0; lazy && t < d->lanes.size() && t < all.size(); t++) lazy = d->lanes[t].cap_coef > 0 && d->lanes[t].cap_rgb > 0;
  if (lazy) {
    std::vector<int> left;
    round(all, true, &left);
    if (!left.empty()) {
      std::sort(left.begin(), left.end());
      round(left, false, nullptr);
    }
  } else {
    round(all, false, nullptr);
  }
  if (d->ctx) jb_ctx_synchronize(d->ctx);
  if (times) {
    times[0] = now_s() - t0;
    times[1] = tot.t_entropy;
    times[2] = tot.t_device;
    times[3] = tot.t_read;
  }
  if (tot.first_error != JB_OK) return jb_fail_(nullptr, tot.first_error, tot.first_error_text.c_str());
  return JB_OK;
}

}  // namespace

extern "C" int jb_batch_decoder_create(int device_id, int n_threads, size_t max_coef_bytes,
                                       size_t max_rgb_bytes, jb_batch_decoder **out) {
  if (!out) return jb_fail_(nullptr, JB_ERR_NULL, "jb_batch_decoder_create: out is NULL");
  *out = nullptr;
  int rc = create_single(device_id, clamp_threads(n_threads, jb_knobs_read()), max_coef_bytes, max_rgb_bytes, out);
  if (rc == JB_OK) {
    (*out)->made_devices.assign(1, device_id);
    (*out)->made_threads = n_threads;
    (*out)->made_coef = max_coef_bytes;
    (*out)->made_rgb = max_rgb_bytes;
  }
  return rc;
}

extern "C" int jb_batch_decoder_create_multi(const int *device_ids, int n_devices, int n_threads,
                                             size_t max_coef_bytes, size_t max_rgb_bytes, jb_batch_decoder **out) {
  if (!out || !device_ids) return jb_fail_(nullptr, JB_ERR_NULL, "jb_batch_decoder_create_multi: NULL pointer");
  *out = nullptr;
  if (n_devices < 1 || n_devices > 64) return jb_fail_(nullptr, JB_ERR_GEOMETRY, "jb_batch_decoder_create_multi: 1..64 devices");
  n_threads = clamp_threads(n_threads, jb_knobs_read());
  if (n_threads < n_devices) n_threads = n_devices;  // every device needs a host thread to feed it
  jb_batch_decoder *top = new jb_batch_decoder();
  top->device = device_ids[0];
  top->made_devices.assign(device_ids, device_ids + n_devices);
  top->made_multi = true;
  top->made_threads = n_threads;
  top->made_coef = max_coef_bytes;
  top->made_rgb = max_rgb_bytes;
  for (int k = 0; k < n_devices; k++) {
    // host threads are dealt out evenly; the first (n_threads % n_devices) devices get one more
    const int share = n_threads / n_devices + (k < n_threads % n_devices ? 1 : 0);
    jb_batch_decoder *part = nullptr;
    int rc = create_single(device_ids[k], share, max_coef_bytes, max_rgb_bytes, &part);
    if (rc != JB_OK) {
      std::string text = jb_last_error(nullptr);
      jb_batch_decoder_destroy(top);
      return jb_fail_(nullptr, rc, text.c_str());
    }
    part->arena = &top->own_arena;
    top->parts.push_back(part);
  }
  *out = top;
  return JB_OK;
}

extern "C" void jb_batch_decoder_destroy(jb_batch_decoder *d) {
  if (!d) return;
  for (auto &f : d->flights)
    if (f.th.joinable()) f.th.join();  // batches still in flight finish first (their results are dropped)
  jb_batch_decoder_destroy(d->twin);
  for (jb_batch_decoder *p : d->parts) jb_batch_decoder_destroy(p);
  for (auto &l : d->lanes) l.release();
  jb_ctx_destroy(d->ctx);
  if (d->own_arena.owned) jb_pinned_free(d->own_arena.base);
  delete d;
}

extern "C" int jb_batch_decoder_set_arena(jb_batch_decoder *d, size_t bytes) {
  if (!d) return jb_fail_(nullptr, JB_ERR_NULL, "jb_batch_decoder_set_arena: decoder is NULL");
  if (d->arena != &d->own_arena) return jb_fail_(nullptr, JB_ERR_STATE, "jb_batch_decoder_set_arena: set the arena on the multi-device decoder, not on one of its parts");
  if (d->in_flight()) return jb_fail_(nullptr, JB_ERR_STATE, "jb_batch_decoder_set_arena: batches are in flight (collect them first)");
  d->split_for_sides = false;  // the next submit arranges the two sides' outputs again
  if (d->twin) (void)jb_batch_decoder_set_arena(d->twin, 0);  // (its arena or region halves go as well; the next submit gives it new ones)
  d->region_base = nullptr;
  d->region_bytes = 0;
  for (jb_batch_decoder *part : d->parts) part->region_base = nullptr, part->region_bytes = 0;
  if (d->own_arena.owned) jb_pinned_free(d->own_arena.base);
  d->own_arena.base = nullptr;
  d->own_arena.bytes = 0;
  d->own_arena.used = 0;
  d->own_arena.on_device = false;
  d->own_arena.owned = true;
  for (jb_batch_decoder *part : d->parts)  // (device regions of a multi-device decoder are forgotten as well)
    if (part->arena == &part->own_arena) {
      part->own_arena.base = nullptr;
      part->own_arena.bytes = 0;
      part->own_arena.used = 0;
      part->own_arena.on_device = false;
      part->own_arena.owned = true;
      part->arena = &d->own_arena;
    }
  if (bytes) {
    // pinned against the (first) device of the decoder; portable, so every device copies into it
    d->own_arena.base = (uint8_t *)jb_pinned_alloc_on(d->device, bytes);
    if (!d->own_arena.base) return jb_fail_(nullptr, JB_ERR_HIP, "jb_batch_decoder_set_arena: pinned allocation failed");
    d->own_arena.bytes = bytes;
    // no pixel staging while an arena takes the pixels
    for (auto &l : d->lanes) l.drop_out();
    for (jb_batch_decoder *p : d->parts)
      for (auto &l : p->lanes) l.drop_out();
  }
  return JB_OK;
}

extern "C" int jb_batch_decoder_set_device_output(jb_batch_decoder *d, void *d_base, size_t bytes) {
  if (!d) return jb_fail_(nullptr, JB_ERR_NULL, "jb_batch_decoder_set_device_output: decoder is NULL");
  if (!d->parts.empty() || d->arena != &d->own_arena)
    return jb_fail_(nullptr, JB_ERR_UNSUPPORTED, "jb_batch_decoder_set_device_output: a multi-device decoder takes one region per device (jb_batch_decoder_set_device_outputs)");
  if ((d_base == nullptr) != (bytes == 0)) return jb_fail_(nullptr, JB_ERR_NULL, "jb_batch_decoder_set_device_output: pointer and size must both be given or both be zero");
  if ((uintptr_t)d_base & 255) return jb_fail_(nullptr, JB_ERR_GEOMETRY, "jb_batch_decoder_set_device_output: the region must be 256-byte aligned");
  if (d->in_flight()) return jb_fail_(nullptr, JB_ERR_STATE, "jb_batch_decoder_set_device_output: batches are in flight (collect them first)");
  int rc = d_base ? jb_check_device_region_(d->device, d_base, bytes) : (int)JB_OK;
  if (rc != JB_OK) return rc;
  rc = jb_batch_decoder_set_arena(d, 0);  // releases a pinned arena, forgets an earlier device region
  if (rc != JB_OK) return rc;
  if (d_base) {
    d->own_arena.base = (uint8_t *)d_base;
    d->own_arena.bytes = bytes;
    d->own_arena.on_device = true;
    d->own_arena.owned = false;
    d->region_base = (uint8_t *)d_base;
    d->region_bytes = bytes;
    for (auto &l : d->lanes) l.drop_out();  // no pixel staging: nothing is downloaded
  }
  return JB_OK;
}

// the multi-device form: one region per listed device, in the order of jb_batch_decoder_create_multi
extern "C" int jb_batch_decoder_set_device_outputs(jb_batch_decoder *d, void *const *d_bases, const size_t *bytes, int n) {
  if (!d) return jb_fail_(nullptr, JB_ERR_NULL, "jb_batch_decoder_set_device_outputs: decoder is NULL");
  if (d->parts.empty()) {
    if (n != 1 || !d_bases || !bytes) return jb_fail_(nullptr, JB_ERR_GEOMETRY, "jb_batch_decoder_set_device_outputs: a single-device decoder takes one region");
    return jb_batch_decoder_set_device_output(d, d_bases[0], bytes[0]);
  }
  if (d->in_flight()) return jb_fail_(nullptr, JB_ERR_STATE, "jb_batch_decoder_set_device_outputs: batches are in flight (collect them first)");
  const bool off = n == 0;
  if (!off && (n != (int)d->parts.size() || !d_bases || !bytes))
    return jb_fail_(nullptr, JB_ERR_GEOMETRY, "jb_batch_decoder_set_device_outputs: one region per listed device (or n = 0: host output again)");
  if (!off)
    for (int k = 0; k < n; k++)
      if (!d_bases[k] || !bytes[k] || ((uintptr_t)d_bases[k] & 255))
        return jb_fail_(nullptr, JB_ERR_GEOMETRY, "jb_batch_decoder_set_device_outputs: every region needs a 256-byte aligned pointer and a size");
  int rc = JB_OK;
  for (int k = 0; !off && k < n && rc == JB_OK; k++) rc = jb_check_device_region_(d->parts[(size_t)k]->device, d_bases[k], bytes[k]);
  if (rc != JB_OK) return rc;
  rc = jb_batch_decoder_set_arena(d, 0);  // a shared pinned arena and device regions exclude each other
  if (rc != JB_OK) return rc;
  for (size_t k = 0; k < d->parts.size(); k++) {
    jb_batch_decoder *part = d->parts[k];
    part->own_arena.base = off ? nullptr : (uint8_t *)d_bases[k];
    part->own_arena.bytes = off ? 0 : bytes[k];
    part->own_arena.used = 0;
    part->own_arena.on_device = !off;
    part->own_arena.owned = off;
    part->region_base = part->own_arena.base;
    part->region_bytes = part->own_arena.bytes;
    part->arena = off ? &d->own_arena : &part->own_arena;  // (host output: the parts share the top decoder's arena, if any)
    if (!off)
      for (auto &l : part->lanes) l.drop_out();
  }
  return JB_OK;
}

namespace {

// the single-device decoders a decoder consists of
std::vector<jb_batch_decoder *> singles_of(jb_batch_decoder *d) {
  if (d->parts.empty()) return std::vector<jb_batch_decoder *>(1, d);
  return d->parts;
}

// Where the pixels of the two sides go.  for_sides = false (jb_batch_decoder_run): this decoder has all of the
// caller's device region(s).  for_sides = true (submit / collect): side 0 writes into the first half of every
// region and the twin into the second; with a pinned arena (or none) the twin has one of the same size of its own.
int arrange_outputs(jb_batch_decoder *d, bool for_sides) {
  std::vector<jb_batch_decoder *> mine = singles_of(d), theirs;
  if (d->twin) theirs = singles_of(d->twin);
  bool any_region = false;
  for (jb_batch_decoder *m : mine) any_region = any_region || m->region_base;
  if (for_sides && d->twin) {
    bool twin_on_device = false;
    for (jb_batch_decoder *t : theirs) twin_on_device = twin_on_device || t->own_arena.on_device;
    const size_t want = (!any_region && d->own_arena.base && d->own_arena.owned) ? d->own_arena.bytes : 0;
    const size_t have = (d->twin->own_arena.base && d->twin->own_arena.owned) ? d->twin->own_arena.bytes : 0;
    if (want != have || twin_on_device) {
      int rc = jb_batch_decoder_set_arena(d->twin, want);
      if (rc != JB_OK) return rc;
    }
  }
  for (size_t k = 0; k < mine.size(); k++) {
    jb_batch_decoder *m = mine[k];
    if (!m->region_base) continue;
    const size_t half = (m->region_bytes / 2) & ~(size_t)255;
    m->own_arena.bytes = for_sides ? half : m->region_bytes;
    if (for_sides && d->twin) {
      jb_batch_decoder *t = theirs[k];
      t->own_arena.base = m->region_base + half;
      t->own_arena.bytes = half;
      t->own_arena.used = 0;
      t->own_arena.on_device = true;
      t->own_arena.owned = false;
      t->arena = &t->own_arena;
      for (auto &l : t->lanes) l.drop_out();
    }
  }
  d->split_for_sides = for_sides;
  return JB_OK;
}

int run_impl(jb_batch_decoder *d, const char *const *paths, int n_paths, uint8_t **rgb, int32_t *widths, int32_t *heights,
             int *statuses, double *times) {
  if (d->parts.empty()) return run_single(d, paths, n_paths, rgb, widths, heights, statuses, times, d->arena == &d->own_arena);
  // multi-device: file i -> part i % n_parts (images are independent: nothing crosses devices);
  // every part runs its share on its own host threads, concurrently with the others
  const int np = (int)d->parts.size();
  d->own_arena.used = 0;
  for (jb_batch_decoder *part : d->parts)
    if (part->arena == &part->own_arena) part->own_arena.used = 0;  // device regions: recycled by every run
  const double t0 = now_s();
  struct Share {
    std::vector<const char *> paths;
    std::vector<uint8_t *> rgb;
    std::vector<int32_t> w, h;
    std::vector<int> st;
    double times[4] = {0, 0, 0, 0};
    int rc = JB_OK;
    std::string text;
  };
  std::vector<Share> sh((size_t)np);
  for (int i = 0; i < n_paths; i++) sh[(size_t)(i % np)].paths.push_back(paths[i]);
  std::vector<std::thread> th;
  for (int k = 0; k < np; k++) {
    Share &s = sh[(size_t)k];
    const size_t n = s.paths.size();
    s.rgb.assign(n, nullptr);
    s.w.assign(n, 0);
    s.h.assign(n, 0);
    s.st.assign(n, JB_OK);
    th.emplace_back([&, k] {
      Share &m = sh[(size_t)k];
      m.rc = run_single(d->parts[(size_t)k], m.paths.data(), (int)m.paths.size(), m.rgb.data(), m.w.data(), m.h.data(),
                        m.st.data(), m.times, false);
      if (m.rc != JB_OK) m.text = jb_last_error(nullptr);  // thread-local text: fetch it on this thread
    });
  }
  for (auto &x : th) x.join();
  int rc = JB_OK;
  std::string text;
  for (int i = 0; i < n_paths; i++) {
    Share &s = sh[(size_t)(i % np)];
    const size_t j = (size_t)(i / np);
    rgb[i] = s.rgb[j];
    widths[i] = s.w[j];
    heights[i] = s.h[j];
    statuses[i] = s.st[j];
  }
  for (int k = 0; k < np; k++)
    if (rc == JB_OK && sh[(size_t)k].rc != JB_OK) {
      rc = sh[(size_t)k].rc;
      text = sh[(size_t)k].text;
    }
  if (times) {
    times[0] = now_s() - t0;
    times[1] = times[2] = times[3] = 0;
    for (int k = 0; k < np; k++)
      for (int j = 1; j < 4; j++) times[j] += sh[(size_t)k].times[j];
  }
  return rc == JB_OK ? JB_OK : jb_fail_(nullptr, rc, text.c_str());
}

}  // namespace

extern "C" int jb_batch_decoder_run(jb_batch_decoder *d, const char *const *paths, int n_paths,
                                    uint8_t **rgb, int32_t *widths, int32_t *heights, int *statuses,
                                    double *times) {
  if (!d || !paths || !rgb || !widths || !heights || !statuses)
    return jb_fail_(nullptr, JB_ERR_NULL, "jb_batch_decoder_run: NULL pointer");
  if (n_paths < 0) return jb_fail_(nullptr, JB_ERR_GEOMETRY, "jb_batch_decoder_run: negative count");
  if (d->in_flight()) return jb_fail_(nullptr, JB_ERR_STATE, "jb_batch_decoder_run: batches are in flight (collect them first)");
  if (d->split_for_sides) {
    int rc = arrange_outputs(d, false);
    if (rc != JB_OK) return rc;
  }
  return run_impl(d, paths, n_paths, rgb, widths, heights, statuses, times);
}

extern "C" int jb_batch_decoder_submit(jb_batch_decoder *d, const char *const *paths, int n_paths, uint8_t **rgb,
                                       int32_t *widths, int32_t *heights, int *statuses, int *ticket) {
  if (!d || !paths || !rgb || !widths || !heights || !statuses || !ticket)
    return jb_fail_(nullptr, JB_ERR_NULL, "jb_batch_decoder_submit: NULL pointer");
  if (n_paths < 0) return jb_fail_(nullptr, JB_ERR_GEOMETRY, "jb_batch_decoder_submit: negative count");
  if (d->arena != &d->own_arena && d->parts.empty())
    return jb_fail_(nullptr, JB_ERR_STATE, "jb_batch_decoder_submit: submit to the multi-device decoder, not to one of its parts");
  for (int i = 0; i < n_paths; i++)
    if (!paths[i]) return jb_fail_(nullptr, JB_ERR_NULL, "jb_batch_decoder_submit: NULL path");
  const int side = d->tickets & 1;
  jb_batch_decoder::Flight &f = d->flights[side];
  if (f.busy) return jb_fail_(nullptr, JB_ERR_STATE, "jb_batch_decoder_submit: two batches are in flight: collect the older one first");
  if (!d->twin) {
    jb_batch_decoder *t = nullptr;
    int rc = d->made_multi ? jb_batch_decoder_create_multi(d->made_devices.data(), (int)d->made_devices.size(), d->made_threads,
                                                           d->made_coef, d->made_rgb, &t)
                           : jb_batch_decoder_create(d->made_devices.empty() ? d->device : d->made_devices[0],
                                                     d->made_threads > 0 ? d->made_threads : (int)d->lanes.size(), d->made_coef,
                                                     d->made_rgb, &t);
    if (rc != JB_OK) return rc;
    d->twin = t;
    d->split_for_sides = false;
  }
  if (!d->split_for_sides) {  // (only ever the case with nothing in flight: every call that clears it refuses otherwise)
    int rc = arrange_outputs(d, true);
    if (rc != JB_OK) return rc;
  }
  f.path_text.assign(paths, paths + n_paths);
  f.path_ptr.resize((size_t)n_paths);
  for (int i = 0; i < n_paths; i++) f.path_ptr[(size_t)i] = f.path_text[(size_t)i].c_str();
  f.busy = true;
  f.ticket = d->tickets++;
  f.rc = JB_OK;
  f.text.clear();
  jb_batch_decoder *const target = side == 0 ? d : d->twin;
  jb_batch_decoder::Flight *const fp = &f;
  try {
    f.th = std::thread([=] {
      fp->rc = run_impl(target, fp->path_ptr.data(), n_paths, rgb, widths, heights, statuses, fp->times);
      if (fp->rc != JB_OK) fp->text = jb_last_error(nullptr);  // thread-local text: fetch it on this thread
    });
  } catch (const std::exception &e) {  // no thread to be had: nothing is in flight
    f.busy = false;
    d->tickets--;
    return jb_fail_(nullptr, JB_ERR_CAPACITY, "jb_batch_decoder_submit: cannot start a thread");
  }
  *ticket = f.ticket;
  return JB_OK;
}

extern "C" int jb_batch_decoder_collect(jb_batch_decoder *d, int ticket, double *times) {
  if (!d) return jb_fail_(nullptr, JB_ERR_NULL, "jb_batch_decoder_collect: decoder is NULL");
  jb_batch_decoder::Flight &f = d->flights[ticket & 1];
  if (ticket < 0 || !f.busy || f.ticket != ticket)
    return jb_fail_(nullptr, JB_ERR_STATE, "jb_batch_decoder_collect: no such batch in flight");
  f.th.join();
  f.busy = false;
  if (times)
    for (int j = 0; j < 4; j++) times[j] = f.times[j];
  return f.rc == JB_OK ? JB_OK : jb_fail_(nullptr, f.rc, f.text.c_str());
}

extern "C" long long jb_batch_decoder_device_entropy_images(const jb_batch_decoder *d) {
  if (!d) return 0;
  long long n = d->ctx ? jb_ctx_device_entropy_images(d->ctx) : 0;
  for (const jb_batch_decoder *p : d->parts) n += jb_batch_decoder_device_entropy_images(p);
  if (d->twin) n += jb_batch_decoder_device_entropy_images(d->twin);  // (the second side of submit / collect)
  return n;
}

extern "C" int jb_decode_batch(int device_id, const char *const *paths, int n_paths, int n_threads,
                               uint8_t **rgb, int32_t *widths, int32_t *heights, int *statuses,
                               double *times) {
  if (!paths || !rgb || !widths || !heights || !statuses)
    return jb_fail_(nullptr, JB_ERR_NULL, "jb_decode_batch: NULL pointer");
  if (n_threads > n_paths && n_paths > 0) n_threads = n_paths;
  jb_batch_decoder *d = nullptr;
  int rc = jb_batch_decoder_create(device_id, n_threads, 0, 0, &d);  // buffers size themselves
  if (rc) return rc;
  rc = jb_batch_decoder_run(d, paths, n_paths, rgb, widths, heights, statuses, times);
  jb_batch_decoder_destroy(d);
  return rc;
}
