// jb_api.cpp -- the C ABI of include/jpegblk.h over the HIP kernels of jb_kernels.hip.
//
// Host side of the seam dequantize(); inverseDCT(); YCbCrToRGB(); (reference
// jpeg.cpp:786-788).  A jb_ctx owns two HIP streams and a ring of staging slots (device
// coefficient / pixel buffers + a pinned quant-table block each) so that the copies and the
// kernel of image i overlap the host Huffman stage of image i+1.  One stream uploads and
// computes (H2D + kernel), the other downloads (D2H, ordered after the kernel by an event), so
// the pixels of image i travel device->host while the coefficients of image i+1 travel
// host->device (the link is full duplex: 57 GB/s one way, 97 GB/s both ways, tools/probe_pcie.hip).  There is deliberately NO CPU
// fallback here: without a usable HIP device every compute entry point fails with JB_ERR_HIP.
#include <atomic>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <hip/hip_runtime_api.h>

#include <cstdarg>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include <sched.h>

#include "../../include/jpegblk.h"
#include "jb_kernels.h"
#include "jb_huff.h"
#include "jb_knobs.h"

namespace {

// The staging ring drives up to 8 stream pairs and 16 pool streams per context; the HIP runtime
// maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and streams that share a queue
// run their kernels one after the other.  Measured with 16 submitting threads: 1,024 1080p files
// 2,838 -> 3,211 images/s (host entropy), 4,807 -> 5,428 (device entropy), 128 files 2,398 -> 3,344,
// with 16 queues instead of 4.  So the library asks for 16 when it is loaded -- only if the variable
// is not set already, and only effective if the process has not initialised HIP yet: an application
// that initialises HIP first (torch, say) sets GPU_MAX_HW_QUEUES=16 itself, or loads this library
// first.  The setting is process-wide (every HIP user of the process gets 16 queues);
// JPEGBLK_HW_QUEUES=0 leaves the runtime's default alone, =N asks for N (include/jpegblk.h says the same).
__attribute__((constructor)) void jb_ask_for_hw_queues() {
  const char *k = getenv("JPEGBLK_HW_QUEUES");
  if (k && k[0] == '0' && k[1] == 0) return;
  if (getenv("GPU_MAX_HW_QUEUES")) return;  // the application's own choice stands
  setenv("GPU_MAX_HW_QUEUES", (k && k[0]) ? k : "16", 0);
}

thread_local std::string g_tls_error = "";

struct Slot {
  void *d_coef = nullptr;
  void *d_rgb = nullptr;
  int32_t *h_q = nullptr;  // pinned int32[3][64]
  int32_t *d_q = nullptr;
  // device-side entropy decoding (jb_huff.hip): the submission's packed scans, interval tables and
  // Huffman table sets (pinned host copy + device copy), and one status word per image
  uint8_t *h_blob = nullptr;
  size_t h_blob_cap = 0;
  void *d_blob = nullptr;
  size_t blob_cap = 0;
  uint32_t *h_status = nullptr;  // pinned, kMaxBatch words
  uint32_t *d_status = nullptr;
  int n_status = 0;              // images of the submission in flight whose status words must be checked
  hipEvent_t computed = nullptr;  // kernel finished (upload stream) -> the download may start
  hipEvent_t done = nullptr;      // pixels are in the caller's buffer
  // a download that the context's download thread has not issued yet (see jb_ctx::dl_*): `done` is only recorded
  // once it has, so whoever waits for the slot waits for this to clear first
  std::atomic<int> dl_pending{0};
  bool busy = false;
  int ticket = -1;
};

// one download handed to the device's download thread
struct DlItem {
  jb_ctx *ctx;
  uint64_t age;  // the context's dl_age when the copy was handed over: the engine serves the smallest first
  Slot *slot;
  void *dst;
  const void *src;
  size_t bytes;                                    // 1-D copy ...
  size_t dst_pitch, src_pitch, row_bytes, rows;    // ... or, rows > 0, a 2-D one
  void *status_dst;
  const void *status_src;
  size_t status_bytes;
};

// Downloads of the submissions whose entropy stage runs on the device, in the order their KERNELS FINISH, back to
// back on one stream -- one engine per DEVICE, shared by every context on it (two batch decoders on one GPU, the
// two sides of jb_batch_decoder_submit): the link has one direction to give, and copies of several contexts issued
// side by side share it at 45 GB/s where one stream gets 57.  Issued by the submitting threads on their own
// streams, several copies shared the link the same way; on one dedicated stream in SUBMISSION order a copy whose
// kernels were still queued held up every copy behind it.  So the submitter records `computed` behind its kernels
// and hands the copy to this thread, which issues whichever is ready, two deep.
struct DlEngine {
  int device = 0;
  int refs = 0;  // contexts holding it (under g_dl_mu)
  std::thread thread;
  std::mutex mu;
  std::condition_variable cv, issued_cv;
  std::deque<DlItem> queue;
  bool stop = false;
  hipStream_t stream = nullptr;
};

}  // namespace

struct jb_ctx {
  int device = 0;
  JbKnobs knobs;                  // the environment as it was when the context was created (jb_knobs.h)
  hipStream_t stream = nullptr;   // primary: uploads + kernels of the ring; device-resident launches with a NULL stream
  hipStream_t stream2 = nullptr;  // downloads of the staging ring
  // Submissions whose entropy stage runs on the device: a decoder launch is latency-bound (a lane
  // walks its interval's blocks one after the other: milliseconds, whatever the group size), so
  // several of them must be in flight at once; each such submission runs whole on one of these.
  static constexpr int kPool = 16;
  hipStream_t pool[kPool] = {};
  // Single-image submissions of host coefficients: K independent (upload + kernel, download) stream
  // pairs used in turn -- pair 0 is (stream, stream2).  Within a pair the download of image i
  // overlaps the upload of image i + K (the link runs both ways); across pairs the chains of
  // different submitters do not queue behind each other.
  // Measured with 16 submitting threads (profiles/r02b/ab_stream_pairs.txt): 1080p images (19 MB
  // per submission) 1,951 images/s with one pair, 2,631 with eight; 8192x8192 images (402 MB) 164
  // with one pair, 91 with eight -- several large copies in one direction at a time share the link
  // badly -- so submissions of 64 MB or more all use pair 0.
  static constexpr int kMaxPairs = 8;
  static constexpr size_t kLargeSubmission = (size_t)64 << 20;
  int n_pairs = 8;
  hipStream_t pair_up[kMaxPairs] = {}, pair_down[kMaxPairs] = {};
  unsigned n_single_submits = 0;
  unsigned n_group_submits = 0;
  size_t max_coef = 0, max_rgb = 0, rgb_alloc = 0;
  int n_slots = 0;
  int n_slots_req = 1;  // ring depth asked for at creation (used when jb_ctx_reserve builds the ring later)
  Slot slots[64];  // n_slots of them are in use
  Slot huff_aux;   // blob + status of jb_entropy_decode_device (the only fields of it in use)
  int next_slot = 0;
  int next_ticket = 1;
  int n_cus = 256;                 // compute units of the device
  size_t blob_hint = 0;            // the largest device blob any slot has been given (huff_stage)
  long long n_device_entropy = 0;  // images whose entropy stage ran on the device (jb_huff.hip)
  jb_image_desc last_desc = {0, 0, 0, 0, {0, 0, 0}, 0};  // frame of the last jb_decode_file / jb_decode_memory
  std::string error;
  DlEngine *dl = nullptr;   // the device's download engine, once this context has handed it a copy
  int dl_outstanding = 0;   // copies handed over and not issued yet (under dl->mu)
  // Which of several contexts' ready copies the engine issues first: the smaller number.  The batch decoder gives
  // every run the next number, so that of two batches in flight the OLDER one gets the link and finishes, instead of
  // both sharing it and finishing together (two batches that share evenly fall into step, and the start-up of the
  // next pair then overlaps nothing).
  std::atomic<uint64_t> dl_age{0};
  std::string dl_error;     // (under dl->mu)
};

namespace {

int fail(jb_ctx *ctx, int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (ctx) ctx->error = buf;
  g_tls_error = buf;
  return code;
}

#define JB_HIP(ctx, call)                                                                      \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess) return fail(ctx, JB_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
  } while (0)

int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// Makes the context's device current for the duration of a call and restores the caller's
// (a jb_ctx may live on any GPU of the node; the calling thread may be using another one).
struct DeviceGuard {
  int prev = -1;
  bool changed = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) == hipSuccess && prev != dev) changed = (hipSetDevice(dev) == hipSuccess);
  }
  ~DeviceGuard() {
    if (changed) (void)hipSetDevice(prev);
  }
};

// The staging ring of a context (device coefficient / pixel buffers, a pinned table block and two
// events per slot, plus the download stream), sized from ctx->max_coef / ctx->rgb_alloc.  The
// caller holds a DeviceGuard on ctx->device, so the pinned table blocks are pinned against THAT
// device (and land on its NUMA node), not against whatever device the calling thread had current.
hipError_t build_ring(jb_ctx *ctx) {
  hipError_t e = hipSuccess;
  if (!ctx->stream2) e = hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking);
  for (int i = 0; e == hipSuccess && i < ctx->n_slots; i++) {
    Slot &s = ctx->slots[i];
    e = hipMalloc(&s.d_coef, round_up((int64_t)ctx->max_coef, 256));
    if (e == hipSuccess) e = hipMalloc(&s.d_rgb, ctx->rgb_alloc);
    if (e == hipSuccess && !s.d_q) e = hipMalloc((void **)&s.d_q, 768 * 256);  // tables of up to 256 images
    if (e == hipSuccess && !s.h_q) e = hipHostMalloc((void **)&s.h_q, 768 * 256, hipHostMallocDefault);
    // (hipEventBlockingSync -- waiting threads sleep instead of spinning -- was measured with 16 waiting
    // threads on a 16-CPU share: no difference, 6,003 vs 5,671 and 216 vs 217 images/s)
    if (e == hipSuccess && !s.done) e = hipEventCreateWithFlags(&s.done, hipEventDisableTiming);
    if (e == hipSuccess && !s.computed) e = hipEventCreateWithFlags(&s.computed, hipEventDisableTiming);
    // what a submission decoded on the device needs beside the coefficients: status words, and a blob for the scan
    // bytes and the decoder's per-chunk state -- about a seventh of the coefficient bytes for quality-75 files --
    // made here, with the ring, rather than by the first submission that lands on the slot
    if (e == hipSuccess && !s.h_status && ctx->knobs.gpu_huffman != 0) {
      e = hipHostMalloc((void **)&s.h_status, 4 * 256, hipHostMallocDefault);
      if (e == hipSuccess) e = hipMalloc((void **)&s.d_status, 4 * 256);
    }
    if (e == hipSuccess && !s.d_blob && ctx->knobs.gpu_huffman != 0 && ctx->n_slots > 1) {
      const size_t cap = ctx->max_coef / 6 + ((size_t)256 << 10);
      e = hipMalloc(&s.d_blob, cap);
      if (e == hipSuccess) s.blob_cap = cap;
    }
  }
  return e;
}

// The ring slot of the next submission: strict round robin; a full ring blocks the submitter on the
// oldest submission (back-pressure).  (Taking the first slot whose submission has completed instead
// -- submissions on different streams complete out of order -- was measured with 16 submitting
// threads: no gain on 1,024 1080p or 64 8192x8192 files, 2-4 % slower on 8,192 small images and on
// the host-entropy path, where the queries under the shared lock cost more than the rare wait:
// profiles/r02b/ab_ring_order.txt.)
// the submission in slot `s` has completed: its download has been issued (by the download thread, if it went
// that way) and has finished
int slot_finish(jb_ctx *ctx, Slot &s) {
  if (s.dl_pending.load(std::memory_order_acquire)) {
    std::unique_lock<std::mutex> lk(ctx->dl->mu);
    ctx->dl->issued_cv.wait(lk, [&] { return s.dl_pending.load(std::memory_order_acquire) == 0; });
    if (!ctx->dl_error.empty()) return fail(ctx, JB_ERR_HIP, "%s", ctx->dl_error.c_str());
  }
  JB_HIP(ctx, hipEventSynchronize(s.done));
  return JB_OK;
}

int take_slot(jb_ctx *ctx, Slot **out) {
  Slot &s = ctx->slots[ctx->next_slot];
  if (s.busy) {
    const int rc = slot_finish(ctx, s);
    if (rc) return rc;
    s.busy = false;
  }
  *out = &s;
  return JB_OK;
}

std::mutex g_dl_mu;
std::map<int, DlEngine *> g_dl;  // device -> its download engine, while any context holds it

void dl_thread_main(DlEngine *eng) {
  (void)hipSetDevice(eng->device);
  // two copies deep, by the engine's OWN events (a context -- and its slots' events -- may be destroyed as soon as
  // its copies have finished, while this thread still remembers them)
  hipEvent_t mark[2] = {nullptr, nullptr};
  bool marked[2] = {false, false};
  for (hipEvent_t &m : mark) (void)hipEventCreateWithFlags(&m, hipEventDisableTiming);
  unsigned n_issued = 0;
  for (;;) {
    DlItem it;
    bool have = false;
    {
      std::unique_lock<std::mutex> lk(eng->mu);
      eng->cv.wait(lk, [&] { return eng->stop || !eng->queue.empty(); });
      if (eng->stop && eng->queue.empty()) break;
      // the oldest batch's ready copy; within a batch, the one handed over first
      auto best = eng->queue.end();
      for (auto q = eng->queue.begin(); q != eng->queue.end(); ++q) {
        if (best != eng->queue.end() && q->age >= best->age) continue;
        const hipError_t e = hipEventQuery(q->slot->computed);
        if (e != hipErrorNotReady) best = q;  // finished (or failed: the copy below will say so)
      }
      if (best != eng->queue.end()) {
        it = *best;
        eng->queue.erase(best);
        have = true;
      }
      (void)hipGetLastError();  // (hipErrorNotReady is not an error of this thread's next call)
    }
    if (!have) {
      std::this_thread::sleep_for(std::chrono::microseconds(20));
      continue;
    }
    // two copies deep: the engine always has the next one, and nothing queues up behind a slow host
    const unsigned m = n_issued++ & 1;
    if (marked[m]) (void)hipEventSynchronize(mark[m]);  // the copy before the last has finished
    hipError_t e;
    if (it.rows) e = hipMemcpy2DAsync(it.dst, it.dst_pitch, it.src, it.src_pitch, it.row_bytes, it.rows, hipMemcpyDeviceToHost, eng->stream);
    else e = hipMemcpyAsync(it.dst, it.src, it.bytes, hipMemcpyDeviceToHost, eng->stream);
    if (e == hipSuccess && it.status_bytes) e = hipMemcpyAsync(it.status_dst, it.status_src, it.status_bytes, hipMemcpyDeviceToHost, eng->stream);
    if (e == hipSuccess) e = hipEventRecord(it.slot->done, eng->stream);
    marked[m] = mark[m] && hipEventRecord(mark[m], eng->stream) == hipSuccess;
    {
      std::lock_guard<std::mutex> lk(eng->mu);
      if (e != hipSuccess && it.ctx->dl_error.empty()) it.ctx->dl_error = std::string("download thread: ") + hipGetErrorString(e);
      it.slot->dl_pending.store(0, std::memory_order_release);
      it.ctx->dl_outstanding--;
    }
    eng->issued_cv.notify_all();
  }
  for (hipEvent_t m : mark)
    if (m) (void)hipEventDestroy(m);
}

// hand a download to the device's download thread (engine and thread are made on first use); the caller has
// recorded item.slot->computed and holds a DeviceGuard on ctx->device
int dl_enqueue(jb_ctx *ctx, const DlItem &item) {
  if (!ctx->dl) {
    std::lock_guard<std::mutex> g(g_dl_mu);
    DlEngine *&eng = g_dl[ctx->device];
    if (!eng) {
      std::unique_ptr<DlEngine> fresh(new DlEngine());
      fresh->device = ctx->device;
      JB_HIP(ctx, hipStreamCreateWithFlags(&fresh->stream, hipStreamNonBlocking));
      fresh->thread = std::thread(dl_thread_main, fresh.get());
      eng = fresh.release();
    }
    eng->refs++;
    ctx->dl = eng;
  }
  item.slot->dl_pending.store(1, std::memory_order_release);
  {
    std::lock_guard<std::mutex> lk(ctx->dl->mu);
    ctx->dl_outstanding++;
    ctx->dl->queue.push_back(item);
  }
  ctx->dl->cv.notify_one();
  return JB_OK;
}

// every download this context handed over has been issued
void dl_drain(jb_ctx *ctx) {
  if (!ctx->dl) return;
  std::unique_lock<std::mutex> lk(ctx->dl->mu);
  ctx->dl->issued_cv.wait(lk, [&] { return ctx->dl_outstanding == 0; });
}

// ... and has finished: the slots' `done` events (the engine's stream also carries other contexts' copies)
hipError_t dl_wait_copies(jb_ctx *ctx) {
  dl_drain(ctx);
  for (int i = 0; i < ctx->n_slots; i++)
    if (ctx->slots[i].done) {
      const hipError_t e = hipEventSynchronize(ctx->slots[i].done);
      if (e != hipSuccess) return e;
    }
  return hipSuccess;
}

// the context lets go of the device's engine; the last one out stops the thread
void dl_release(jb_ctx *ctx) {
  if (!ctx->dl) return;
  (void)dl_wait_copies(ctx);
  DlEngine *eng = ctx->dl;
  ctx->dl = nullptr;
  std::lock_guard<std::mutex> g(g_dl_mu);
  if (--eng->refs > 0) return;
  g_dl.erase(eng->device);
  {
    std::lock_guard<std::mutex> lk(eng->mu);
    eng->stop = true;
  }
  eng->cv.notify_all();
  eng->thread.join();
  (void)hipStreamSynchronize(eng->stream);
  (void)hipStreamDestroy(eng->stream);
  delete eng;
}

int check_desc(jb_ctx *ctx, const jb_image_desc *d, jb_geometry *g) {
  int rc = jb_geometry_of(d, g);
  if (rc == JB_ERR_NULL) return fail(ctx, rc, "null descriptor");
  if (rc == JB_ERR_GEOMETRY) return fail(ctx, rc, "image size %dx%d outside 1..65535", d->width, d->height);
  if (rc == JB_ERR_SAMPLING) return fail(ctx, rc, "luma sampling factors %dx%d not in {1,2}x{1,2}", d->hs, d->vs);
  if (rc == JB_ERR_QTAB) return fail(ctx, rc, "quantisation table id outside 0..3");
  return rc;
}

}  // namespace

extern "C" {

int jb_abi_version(void) { return JB_ABI_VERSION; }

int jb_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return fail(nullptr, JB_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
  return n;
}

const char *jb_kernel_name(const jb_image_desc *d) {
  if (!d) return "";
  return jbk_kernel_name(d->hs, d->vs);
}

int jb_ctx_create(int device_id, size_t max_coef_bytes, size_t max_rgb_bytes, int n_slots, jb_ctx **out) {
  if (!out) return fail(nullptr, JB_ERR_NULL, "jb_ctx_create: out is NULL");
  *out = nullptr;
  int ndev = 0;
  JB_HIP(nullptr, hipGetDeviceCount(&ndev));
  if (device_id < 0 || device_id >= ndev)
    return fail(nullptr, JB_ERR_HIP, "device %d not available (%d HIP devices visible)", device_id, ndev);
  if ((max_coef_bytes == 0) != (max_rgb_bytes == 0))
    return fail(nullptr, JB_ERR_CAPACITY, "max_coef_bytes and max_rgb_bytes must both be zero or both non-zero");
  if (n_slots < 1) n_slots = 1;
  if (n_slots > 64) n_slots = 64;
  jb_ctx *ctx = new (std::nothrow) jb_ctx();
  if (!ctx) return fail(nullptr, JB_ERR_CAPACITY, "out of host memory");
  ctx->device = device_id;
  ctx->max_coef = max_coef_bytes;
  ctx->max_rgb = max_rgb_bytes;
  ctx->rgb_alloc = max_rgb_bytes ? (size_t)round_up((int64_t)max_rgb_bytes, 256) : 0;  // device rows are tightly packed
  ctx->n_slots = max_coef_bytes ? n_slots : 0;
  ctx->n_slots_req = n_slots;
  ctx->knobs = jb_knobs_read();
  DeviceGuard guard(device_id);
  hipError_t e = hipSuccess;
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && cus > 0) ctx->n_cus = cus;
    (void)hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
  if (e == hipSuccess && ctx->n_slots > 0) e = build_ring(ctx);
  if (e != hipSuccess) {
    int rc = fail(nullptr, JB_ERR_HIP, "jb_ctx_create: %s", hipGetErrorString(e));
    jb_ctx_destroy(ctx);
    return rc;
  }
  *out = ctx;
  return JB_OK;
}

void jb_ctx_destroy(jb_ctx *ctx) {
  if (!ctx) return;
  DeviceGuard guard(ctx->device);
  dl_release(ctx);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
  for (hipStream_t &ps : ctx->pool)
    if (ps) {
      (void)hipStreamSynchronize(ps);
      (void)hipStreamDestroy(ps);
      ps = nullptr;
    }
  for (int k = 1; k < jb_ctx::kMaxPairs; k++)
    for (hipStream_t *ps : {&ctx->pair_up[k], &ctx->pair_down[k]})
      if (*ps) {
        (void)hipStreamSynchronize(*ps);
        (void)hipStreamDestroy(*ps);
        *ps = nullptr;
      }
  for (int i = 0; i < 64; i++) {
    Slot &s = ctx->slots[i];
    if (s.d_coef) (void)hipFree(s.d_coef);
    if (s.d_rgb) (void)hipFree(s.d_rgb);
    if (s.d_q) (void)hipFree(s.d_q);
    if (s.h_q) (void)hipHostFree(s.h_q);
    if (s.h_blob) (void)hipHostFree(s.h_blob);
    if (s.d_blob) (void)hipFree(s.d_blob);
    if (s.h_status) (void)hipHostFree(s.h_status);
    if (s.d_status) (void)hipFree(s.d_status);
    if (s.done) (void)hipEventDestroy(s.done);
    if (s.computed) (void)hipEventDestroy(s.computed);
  }
  {
    Slot &s = ctx->huff_aux;
    if (s.h_blob) (void)hipHostFree(s.h_blob);
    if (s.d_blob) (void)hipFree(s.d_blob);
    if (s.h_status) (void)hipHostFree(s.h_status);
    if (s.d_status) (void)hipFree(s.d_status);
  }
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
  delete ctx;
}

// (jb_batch.cpp) the rank of this context's downloads among those of the other contexts on its device
void jb_ctx_set_download_age_(jb_ctx *ctx, uint64_t age) {
  if (ctx) ctx->dl_age.store(age, std::memory_order_relaxed);
}

const char *jb_last_error(const jb_ctx *ctx) { return ctx ? ctx->error.c_str() : g_tls_error.c_str(); }

void *jb_ctx_stream(jb_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int jb_ctx_synchronize(jb_ctx *ctx) {
  if (!ctx) return fail(nullptr, JB_ERR_NULL, "jb_ctx_synchronize: ctx is NULL");
  DeviceGuard guard(ctx->device);
  JB_HIP(ctx, dl_wait_copies(ctx));
  JB_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->stream2) JB_HIP(ctx, hipStreamSynchronize(ctx->stream2));
  for (hipStream_t ps : ctx->pool)
    if (ps) JB_HIP(ctx, hipStreamSynchronize(ps));
  for (int k = 1; k < jb_ctx::kMaxPairs; k++) {
    if (ctx->pair_up[k]) JB_HIP(ctx, hipStreamSynchronize(ctx->pair_up[k]));
    if (ctx->pair_down[k]) JB_HIP(ctx, hipStreamSynchronize(ctx->pair_down[k]));
  }
  return JB_OK;
}

// Launches of up to this many 192 / 256-lane workgroups per CU take the small-grid kernels when
// JPEGBLK_SMALL_GRID is unset.  Measured on one box, cold, events around every launch, the kernels interleaved
// (profiles/r03/probe_small_grid.json).  4:4:4: one 1080p image (507 workgroups) 12.1 -> 10.2 us, two (1,014)
// 14.6 -> 13.6, four (2,028) 21.5 -> 20.5, one 1280x720 9.4 -> 8.4, one 640x360 9.4 -> 7.1, one 4096x4096 (4,096)
// 33.6 = 33.9.  4:2:0: one 640x360 11.3 -> 7.8 us, one 1080p (255 workgroups) 12.2 -> 9.4, four (1,020) 17.3 -> 15.6,
// eight (2,040) 28.0 -> 25.7, one 4096x4096 (2,048: BASELINE config 3) 27.5 -> 25.6, two (4,096) 44.6 -> 43.1.
// 4:2:2 / 4:4:0 (all 64 lanes busy): one 1080p 11.2 -> 10.6 / 11.4 -> 9.6 us, one 4096x4096 25.0 -> 23.4 / 25.7 -> 26.3.
constexpr int kSmallGridBelowPerCu = 8;

int jb_blocks_to_rgb_device(jb_ctx *ctx, const jb_device_batch *b, void *stream) {
  if (!ctx) return fail(nullptr, JB_ERR_NULL, "jb_blocks_to_rgb_device: ctx is NULL");
  if (!b || !b->d_coef || !b->d_qtabs || !b->d_rgb) return fail(ctx, JB_ERR_NULL, "jb_blocks_to_rgb_device: NULL pointer");
  jb_geometry g;
  int rc = check_desc(ctx, &b->desc, &g);
  if (rc) return rc;
  if (b->n_images < 1) return fail(ctx, JB_ERR_GEOMETRY, "n_images = %d", b->n_images);
  if (b->rgb_row_stride < 3LL * b->desc.width)
    return fail(ctx, JB_ERR_GEOMETRY, "rgb_row_stride %lld < 3*width", (long long)b->rgb_row_stride);
  if (((uintptr_t)b->d_coef & 15) || (b->coef_image_stride & 15))
    return fail(ctx, JB_ERR_GEOMETRY, "coefficient pointer and image stride must be multiples of 16 bytes");
  if (((uintptr_t)b->d_qtabs & 3) || (b->qtab_image_stride & 3))
    return fail(ctx, JB_ERR_GEOMETRY, "quant-table pointer and stride must be multiples of 4 bytes");
  if (b->n_images > 1 && (b->coef_image_stride < g.coef_bytes || b->rgb_image_stride < b->rgb_row_stride * (int64_t)b->desc.height))
    return fail(ctx, JB_ERR_GEOMETRY, "image strides smaller than one image");
  const int per_tile = jbk_mcus_per_tile(b->desc.hs, b->desc.vs);
  JbLaunch p;
  memset(&p, 0, sizeof p);
  p.coef = b->d_coef;
  p.qtabs = b->d_qtabs;
  p.rgb = b->d_rgb;
  p.coef_image_stride = b->coef_image_stride;
  p.qtab_image_stride = b->qtab_image_stride;
  p.rgb_image_stride = b->rgb_image_stride;
  p.rgb_row_stride = b->rgb_row_stride;
  p.width = b->desc.width;
  p.height = b->desc.height;
  p.mcus_x = g.mcus_x;
  p.mcus_y = g.mcus_y;
  p.tiles_per_row = (g.mcus_x + per_tile - 1) / per_tile;
  // JPEGBLK_ROW_TILING=1 (debug / A-B knob) forces the row-bound tiling
  const bool force_row = ctx->knobs.row_tiling;
  // linear tiling only where the row-bound one would leave ragged tiles
  p.linear = (force_row || g.mcus_x % per_tile == 0) ? 0 : jbk_linear_ok(b->desc.hs, b->desc.vs, g.mcus_x);
  const int64_t tiles_per_image = p.linear ? ((int64_t)g.mcus_x * g.mcus_y + per_tile - 1) / per_tile
                                           : (int64_t)g.mcus_y * p.tiles_per_row;
  if (tiles_per_image > 0x7fffffffLL) return fail(ctx, JB_ERR_CAPACITY, "image too large");
  p.tiles_per_image = (int32_t)tiles_per_image;
  const int64_t n_tiles = (int64_t)b->n_images * tiles_per_image;
  if (n_tiles > 0x7fffffffLL) return fail(ctx, JB_ERR_CAPACITY, "batch too large for one launch (%lld tiles)", (long long)n_tiles);
  p.n_tiles = (int32_t)n_tiles;
  // Small launches (a single 1080p image is 507 / 255 workgroups on 256 CUs): four times as many one-wave
  // workgroups (jb_kernels.hip jb_small_kernel_*), row-bound.  JPEGBLK_SMALL_GRID = 1 / 0 forces / forbids it; so does
  // JPEGBLK_ROW_TILING=1 (that knob asks for the 192-lane kernel's row-bound instantiation).
  if (jbk_small_mcus(b->desc.hs, b->desc.vs) > 0 && !force_row && b->rgb_row_stride < (1LL << 26) &&  // (the lane's row offset is 32-bit)
      (ctx->knobs.small_grid == 1 || (ctx->knobs.small_grid < 0 && n_tiles <= (int64_t)kSmallGridBelowPerCu * ctx->n_cus))) {
    const int per = jbk_small_mcus(b->desc.hs, b->desc.vs);
    p.tiles_per_row = (g.mcus_x + per - 1) / per;
    const int64_t small_tiles = (int64_t)b->n_images * g.mcus_y * p.tiles_per_row;
    if (small_tiles <= 0x7fffffffLL) {
      p.linear = 0;
      p.small_grid = 1;
      p.tiles_per_image = (int32_t)((int64_t)g.mcus_y * p.tiles_per_row);
      p.n_tiles = (int32_t)small_tiles;
    } else {
      p.tiles_per_row = (g.mcus_x + per_tile - 1) / per_tile;
    }
  }
  // 12-byte stores at any byte address: gfx950 under ROCm runs with unaligned global/buffer access
  // enabled, and odd widths with tightly packed rows (row stride 3*W) are the common case --
  // measured 1.67x faster than byte stores on 679x451 (tests/test_gpu_parity.py covers both).
  // JPEGBLK_BYTE_STORE=1 forces the byte-store path (test / A-B knob).
  p.fast_store = ctx->knobs.byte_store ? 0 : 1;
  // (measurement builds of jb_kernels.hip only -- tools/build_variant.sh -DJB_LAB: the staged store stage of the linear
  // tiling; the product's kernels ignore the field)
  p.staged = (p.linear && p.fast_store && !p.small_grid && ctx->knobs.staged_store == 1) ? 1 : 0;
  p.chroma_q_equal = (b->desc.qtab_id[1] == b->desc.qtab_id[2]) ? 1 : 0;
  DeviceGuard guard(ctx->device);
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  JB_HIP(ctx, jbk_launch(p, b->desc.hs, b->desc.vs, s));
  return JB_OK;
}

int jb_ctx_device(const jb_ctx *ctx) { return ctx ? ctx->device : -1; }

long long jb_ctx_device_entropy_images(const jb_ctx *ctx) { return ctx ? ctx->n_device_entropy : 0; }

int jb_ctx_last_desc(const jb_ctx *ctx, jb_image_desc *out) {
  if (!ctx || !out) return fail(nullptr, JB_ERR_NULL, "jb_ctx_last_desc: NULL pointer");
  if (ctx->last_desc.width == 0) return fail(nullptr, JB_ERR_STATE, "jb_ctx_last_desc: nothing decoded on this context yet");
  *out = ctx->last_desc;
  return JB_OK;
}

int jb_ctx_reserve(jb_ctx *ctx, size_t max_coef_bytes, size_t max_rgb_bytes) {
  if (!ctx) return fail(nullptr, JB_ERR_NULL, "jb_ctx_reserve: ctx is NULL");
  if (max_coef_bytes == 0 || max_rgb_bytes == 0) return fail(ctx, JB_ERR_CAPACITY, "jb_ctx_reserve: sizes must be non-zero");
  if (ctx->n_slots > 0 && max_coef_bytes <= ctx->max_coef && max_rgb_bytes <= ctx->max_rgb) return JB_OK;
  DeviceGuard guard(ctx->device);
  // nothing may be in flight while the slots' buffers are replaced
  JB_HIP(ctx, dl_wait_copies(ctx));
  JB_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->stream2) JB_HIP(ctx, hipStreamSynchronize(ctx->stream2));
  for (hipStream_t ps : ctx->pool)
    if (ps) JB_HIP(ctx, hipStreamSynchronize(ps));
  for (int k = 1; k < jb_ctx::kMaxPairs; k++) {
    if (ctx->pair_up[k]) JB_HIP(ctx, hipStreamSynchronize(ctx->pair_up[k]));
    if (ctx->pair_down[k]) JB_HIP(ctx, hipStreamSynchronize(ctx->pair_down[k]));
  }
  if (max_coef_bytes < ctx->max_coef) max_coef_bytes = ctx->max_coef;
  if (max_rgb_bytes < ctx->max_rgb) max_rgb_bytes = ctx->max_rgb;
  const int n = ctx->n_slots > 0 ? ctx->n_slots : ctx->n_slots_req;
  for (int i = 0; i < n; i++) {
    Slot &s = ctx->slots[i];
    s.busy = false;
    if (s.d_coef) (void)hipFree(s.d_coef);
    if (s.d_rgb) (void)hipFree(s.d_rgb);
    s.d_coef = s.d_rgb = nullptr;
  }
  ctx->max_coef = max_coef_bytes;
  ctx->max_rgb = max_rgb_bytes;
  ctx->rgb_alloc = (size_t)round_up((int64_t)max_rgb_bytes, 256);
  ctx->n_slots = n;
  hipError_t e = build_ring(ctx);
  if (e != hipSuccess) {
    ctx->n_slots = 0;  // a half-built ring is not used; jb_ctx_destroy releases what exists
    ctx->max_coef = ctx->max_rgb = ctx->rgb_alloc = 0;
    return fail(ctx, JB_ERR_HIP, "jb_ctx_reserve(%zu, %zu): %s", max_coef_bytes, max_rgb_bytes, hipGetErrorString(e));
  }
  return JB_OK;
}

// Pinned host memory is pinned AGAINST a device: hipHostMalloc registers the pages with the
// calling thread's current device and (ROCm's default policy) takes them from the host NUMA node
// closest to that device.  Under one rank per GPU with every GPU visible, a fresh std::thread's
// current device is 0 -- so the device is always named explicitly here.
void *jb_pinned_alloc_on(int device_id, size_t bytes) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) {
    fail(nullptr, JB_ERR_HIP, "jb_pinned_alloc_on: device %d not available (%d HIP devices visible)", device_id, ndev);
    return nullptr;
  }
  DeviceGuard guard(device_id);
  void *p = nullptr;
  // portable: every device of the process may copy to / from it (a multi-device decoder's shared arena)
  hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable);
  if (e != hipSuccess) {
    fail(nullptr, JB_ERR_HIP, "hipHostMalloc(%zu) on device %d: %s", bytes, device_id, hipGetErrorString(e));
    return nullptr;
  }
  return p;
}

void *jb_pinned_alloc(size_t bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    fail(nullptr, JB_ERR_HIP, "jb_pinned_alloc: no usable HIP device");
    return nullptr;
  }
  return jb_pinned_alloc_on(dev, bytes);
}

int jb_device_numa_node(int device_id) {
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess) return fail(nullptr, JB_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
  if (device_id < 0 || device_id >= ndev) return fail(nullptr, JB_ERR_HIP, "device %d not available (%d HIP devices visible)", device_id, ndev);
  int node = -1;
  if (hipDeviceGetAttribute(&node, hipDeviceAttributeHostNumaId, device_id) == hipSuccess && node >= 0) return node;
  (void)hipGetLastError();  // a runtime without that attribute: not an error of the caller's next HIP call
  // fallback: the PCI function's numa_node in sysfs
  char bdf[32] = {0};
  if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, device_id) == hipSuccess) {
    for (char *c = bdf; *c; c++)
      if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');
    char path[96];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bdf);
    if (FILE *f = fopen(path, "r")) {
      int v = -1;
      if (fscanf(f, "%d", &v) != 1) v = -1;
      fclose(f);
      if (v >= 0) return v;
    }
  }
  return fail(nullptr, JB_ERR_STATE, "NUMA node of device %d unknown", device_id);
}

void jb_pinned_free(void *p) {
  if (p) (void)hipHostFree(p);
}

namespace {

constexpr int kMaxBatch = 256;  // images per submission (the slot's table block holds that many)

// One submission of the staging ring: n_images images of one geometry, coefficients contiguous
// (image stride = coef_bytes), tables per image, pixels contiguous with tight rows -- or, for
// n_images == 1, any row stride.
// dst_device: `rgb` is DEVICE memory of ctx's device -- the kernel writes the pixels there and nothing
// is downloaded (jb_batch_decoder_set_device_output).
int submit_impl(jb_ctx *ctx, const jb_image_desc *desc, int n_images, const int16_t *coef, const uint16_t *qtabs,
                uint8_t *rgb, int64_t rgb_stride, int *ticket, bool dst_device = false) {
  if (ctx->n_slots == 0) return fail(ctx, JB_ERR_CAPACITY, "context was created without staging buffers");
  jb_geometry g;
  int rc = check_desc(ctx, desc, &g);
  if (rc) return rc;
  if (n_images < 1 || n_images > kMaxBatch) return fail(ctx, JB_ERR_GEOMETRY, "n_images = %d outside 1..%d", n_images, kMaxBatch);
  const int64_t dev_stride = 3LL * desc->width;  // tight rows on the device (12-byte stores need no alignment)
  if (rgb_stride < dev_stride) return fail(ctx, JB_ERR_GEOMETRY, "rgb_stride %lld < 3*width", (long long)rgb_stride);
  const size_t coef_total = (size_t)g.coef_bytes * (size_t)n_images, rgb_total = (size_t)g.rgb_bytes * (size_t)n_images;
  if (coef_total > ctx->max_coef || (!dst_device && (rgb_total > ctx->rgb_alloc || rgb_total > ctx->max_rgb)))
    return fail(ctx, JB_ERR_CAPACITY, "%d image(s) of %dx%d exceed the capacity the context was created with", n_images,
                desc->width, desc->height);
  if (dst_device && rgb_stride != dev_stride) return fail(ctx, JB_ERR_GEOMETRY, "device output has tight rows");
  DeviceGuard guard(ctx->device);
  Slot *slot = nullptr;
  rc = take_slot(ctx, &slot);
  if (rc) return rc;
  Slot &s = *slot;
  // One image: upload + kernel on the primary stream, download on the second (ordered by an
  // event), so the link runs both ways even with a single submitter.  A group of small images
  // runs whole on one stream and consecutive groups alternate between the two streams: with many
  // submitters that overlaps uploads and downloads just as well, without a cross-stream event per
  // group (measured with 16 host threads on 679x451 images: 14,500 images/s with the event,
  // 24,700 with every group on one stream, 38,400 alternating).
  hipStream_t up = ctx->stream, down = ctx->stream2 ? ctx->stream2 : ctx->stream;
  if (n_images > 1) {
    if (ctx->stream2 && (ctx->n_group_submits++ & 1u)) up = ctx->stream2;
    down = up;
  } else if (ctx->stream2 && ctx->n_pairs > 1 && coef_total + rgb_total < jb_ctx::kLargeSubmission) {
    // one image: the next of the K (upload + kernel, download) stream pairs
    const int k = (int)(ctx->n_single_submits++ % (unsigned)ctx->n_pairs);
    if (k > 0) {
      if (!ctx->pair_up[k]) JB_HIP(ctx, hipStreamCreateWithFlags(&ctx->pair_up[k], hipStreamNonBlocking));
      if (!ctx->pair_down[k]) JB_HIP(ctx, hipStreamCreateWithFlags(&ctx->pair_down[k], hipStreamNonBlocking));
      up = ctx->pair_up[k];
      down = ctx->pair_down[k];
    }
  }
  for (int i = 0; i < n_images; i++) {
    rc = jb_resolve_qtabs(desc, qtabs + (size_t)i * 256, s.h_q + (size_t)i * 192);
    if (rc) return fail(ctx, rc, "bad quantisation table id");
  }
  JB_HIP(ctx, hipMemcpyAsync(s.d_q, s.h_q, 768u * (size_t)n_images, hipMemcpyHostToDevice, up));
  JB_HIP(ctx, hipMemcpyAsync(s.d_coef, coef, coef_total, hipMemcpyHostToDevice, up));
  jb_device_batch b;
  memset(&b, 0, sizeof b);
  b.desc = *desc;
  b.n_images = n_images;
  b.d_coef = (const int16_t *)s.d_coef;
  b.coef_image_stride = g.coef_bytes;  // a multiple of 128
  b.d_qtabs = s.d_q;
  b.qtab_image_stride = n_images > 1 ? 768 : 0;
  b.d_rgb = dst_device ? rgb : (uint8_t *)s.d_rgb;
  b.rgb_row_stride = dev_stride;
  b.rgb_image_stride = g.rgb_bytes;
  rc = jb_blocks_to_rgb_device(ctx, &b, up);
  if (rc) return rc;
  if (dst_device) {
    down = up;  // the pixels stay on the device: done when the kernel is
  } else {
    // the download runs on its own stream, after the kernel: it overlaps the next image's upload
    if (down != up) {
      JB_HIP(ctx, hipEventRecord(s.computed, up));
      JB_HIP(ctx, hipStreamWaitEvent(down, s.computed, 0));
    }
    if (rgb_stride == dev_stride)
      JB_HIP(ctx, hipMemcpyAsync(rgb, s.d_rgb, rgb_total, hipMemcpyDeviceToHost, down));
    else
      JB_HIP(ctx, hipMemcpy2DAsync(rgb, (size_t)rgb_stride, s.d_rgb, (size_t)dev_stride, (size_t)desc->width * 3,
                                   (size_t)desc->height, hipMemcpyDeviceToHost, down));
  }
  JB_HIP(ctx, hipEventRecord(s.done, down));
  s.busy = true;
  s.ticket = ctx->next_ticket++;
  if (ctx->next_ticket < 0) ctx->next_ticket = 1;
  *ticket = s.ticket;
  ctx->next_slot = (ctx->next_slot + 1) % ctx->n_slots;
  return JB_OK;
}

// ---- device-side entropy decoding (jb_huff.hip) -------------------------------------------------
inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

// upload a packed submission (jb_huff_pack_) and launch the decoder: image i's coefficient blocks
// land at d_out + i * coef_stride bytes, its status word at s.d_status[i]
int huff_stage(jb_ctx *ctx, Slot &s, const uint8_t *h, const JbHuffLayout &lay, size_t zero_bytes, int16_t *d_out, hipStream_t up,
               int sync_launches = kJbSyncLaunches) {
  if (lay.device_total > s.blob_cap || !s.d_blob) {  // (the slot is idle: its previous submission has been waited for)
    // hipFree / hipMalloc stall the whole device: a slot that has to grow takes the largest size any slot of this
    // context has needed, so that a ring of 64 slots stops growing after the first full-size groups instead of
    // once per slot (which kept the first five or six runs of a fresh decoder slower than the rest)
    if (s.d_blob) (void)hipFree(s.d_blob);
    s.d_blob = nullptr, s.blob_cap = 0;
    size_t cap = lay.device_total + lay.device_total / 4 + 65536;
    if (cap < ctx->blob_hint) cap = ctx->blob_hint;
    JB_HIP(ctx, hipMalloc(&s.d_blob, cap));
    s.blob_cap = cap;
    ctx->blob_hint = cap;
  }
  if (!s.h_status) {
    JB_HIP(ctx, hipHostMalloc((void **)&s.h_status, 4 * 256, hipHostMallocDefault));
    JB_HIP(ctx, hipMalloc((void **)&s.d_status, 4 * 256));
  }
  // a small submission is fetched from the pinned blob by a kernel: nothing in front of the decoding kernels waits
  // for a copy engine
  if (lay.total <= ((size_t)4 << 20)) JB_HIP(ctx, jbk_huff_fetch(s.d_blob, h, lay.total, up));
  else JB_HIP(ctx, hipMemcpyAsync(s.d_blob, h, lay.total, hipMemcpyHostToDevice, up));
  // the decoder stores non-zero coefficients only
  JB_HIP(ctx, jbk_huff_zero(d_out, zero_bytes, up, s.d_status, 4 * (size_t)lay.n));  // (status words: 4 * 256 bytes are allocated)
  JbHuffLaunch p;
  memset(&p, 0, sizeof p);
  const uint8_t *d = (const uint8_t *)s.d_blob;
  uint8_t *dw = (uint8_t *)s.d_blob;
  p.scan = d + lay.off_scan;
  p.starts = (const uint32_t *)(d + lay.off_starts);
  p.tables = (const JbHuffTables *)(d + lay.off_tab);
  p.images = (const JbHuffImage *)(d + lay.off_img);
  p.wgs = (const JbHuffWg *)(d + lay.off_wg);
  p.sync_wgs = (const JbHuffWg *)(d + lay.off_sync_wg);
  p.coef = d_out;
  p.status = s.d_status;
  p.n_wgs = (int32_t)lay.n_wg;
  p.n_sync_wgs = (int32_t)lay.n_sync_wg;
  p.chunks = (const JbChunkDesc *)(d + lay.off_chunks);
  p.entry = (JbChunkState *)(dw + lay.off_entry);
  p.exit = (JbChunkState *)(dw + lay.off_exit);
  p.cps = (uint32_t *)(dw + lay.off_cps);
  p.chunk_dc = (JbChunkDc *)(dw + lay.off_chunk_dc);
  p.wgsum = (JbWgSum *)(dw + lay.off_wgsum);
  p.n_chunks_total = lay.n_chunks;
  // (one workgroup per image: its first chunk starts an interval, nothing to hand over between launches)
  p.sync_launches = lay.n_sync_wg > 0 && lay.n_wg == lay.n ? 1 : sync_launches;
  p.max_chunk_bytes = lay.max_chunk_bytes;
  p.max_tabs = lay.max_tabs;
  JB_HIP(ctx, jbk_huff_launch(p, up));
  return JB_OK;
}

// the slot's own pinned staging for callers that did not pack themselves (single images)
int pack_into_slot(jb_ctx *ctx, Slot &s, const JbHuffJob *const *jobs, int n, int64_t coef_stride, JbHuffLayout *lay) {
  const size_t need = jb_huff_pack_size_(jobs, n);
  if (need > s.h_blob_cap) {
    if (s.h_blob) (void)hipHostFree(s.h_blob);
    s.h_blob = nullptr, s.h_blob_cap = 0;
    const size_t cap = need + need / 4 + 65536;
    JB_HIP(ctx, hipHostMalloc((void **)&s.h_blob, cap, hipHostMallocDefault));
    s.h_blob_cap = cap;
  }
  const int rc = jb_huff_pack_(jobs, n, coef_stride, s.h_blob, lay);
  return rc ? fail(ctx, rc, "submission too large for the device entropy decoder") : JB_OK;
}

// One submission whose coefficients are produced ON the device: n images of one geometry, each a
// prepared JbHuffJob.  Same ring, same ordering and same download as submit_impl; what is uploaded
// is the compressed scan (a tenth of the coefficients), and the status words come back with the
// pixels.  jb_wait / jb_poll report JB_ERR_FORMAT when the decoder met corrupt data.
// Either `jobs` (packed here, into the slot's pinned staging) or a blob the caller packed itself
// (`packed` + `lay`, pinned, valid until the submission has completed) with the images' descriptor
// and tables (`desc`, `qtabs` = n x 4*64).
int submit_jobs_impl(jb_ctx *ctx, const JbHuffJob *const *jobs, const uint8_t *packed, const JbHuffLayout *lay_in,
                     const jb_image_desc *desc_in, const uint16_t *qtabs_in, int n_images, uint8_t *rgb, int64_t rgb_stride,
                     uint32_t *status_out, int *ticket, bool dst_device = false) {
  if (ctx->n_slots == 0) return fail(ctx, JB_ERR_CAPACITY, "context was created without staging buffers");
  if (n_images < 1 || n_images > kMaxBatch) return fail(ctx, JB_ERR_GEOMETRY, "n_images = %d outside 1..%d", n_images, kMaxBatch);
  const jb_image_desc *desc = jobs ? &jobs[0]->desc : desc_in;
  jb_geometry g;
  int rc = check_desc(ctx, desc, &g);
  if (rc) return rc;
  const int64_t dev_stride = 3LL * desc->width;
  if (rgb_stride < dev_stride) return fail(ctx, JB_ERR_GEOMETRY, "rgb_stride %lld < 3*width", (long long)rgb_stride);
  const size_t coef_total = (size_t)g.coef_bytes * (size_t)n_images, rgb_total = (size_t)g.rgb_bytes * (size_t)n_images;
  if (coef_total > ctx->max_coef || (!dst_device && (rgb_total > ctx->rgb_alloc || rgb_total > ctx->max_rgb)))
    return fail(ctx, JB_ERR_CAPACITY, "%d image(s) of %dx%d exceed the capacity the context was created with", n_images,
                desc->width, desc->height);
  if (dst_device && rgb_stride != dev_stride) return fail(ctx, JB_ERR_GEOMETRY, "device output has tight rows");
  DeviceGuard guard(ctx->device);
  Slot *slot = nullptr;
  rc = take_slot(ctx, &slot);
  if (rc) return rc;
  Slot &s = *slot;
  // the whole submission on one stream of the pool, consecutive submissions on different ones
  hipStream_t &ps = ctx->pool[ctx->n_group_submits++ % jb_ctx::kPool];
  if (!ps) JB_HIP(ctx, hipStreamCreateWithFlags(&ps, hipStreamNonBlocking));
  hipStream_t up = ps;
  for (int i = 0; i < n_images; i++) {
    rc = jb_resolve_qtabs(desc, jobs ? jobs[i]->qtabs : qtabs_in + (size_t)i * 256, s.h_q + (size_t)i * 192);
    if (rc) return fail(ctx, rc, "bad quantisation table id");
  }
  JB_HIP(ctx, hipMemcpyAsync(s.d_q, s.h_q, 768u * (size_t)n_images, hipMemcpyHostToDevice, up));
  const bool timing = ctx->knobs.timing == 2;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double tt0 = timing ? now() : 0;
  JbHuffLayout lay_own;
  if (jobs) {
    rc = pack_into_slot(ctx, s, jobs, n_images, g.coef_bytes, &lay_own);
    if (rc) return rc;
    packed = s.h_blob;
    lay_in = &lay_own;
  }
  if (lay_in->n != n_images || lay_in->coef_stride != g.coef_bytes) return fail(ctx, JB_ERR_STATE, "packed submission does not match its descriptor");
  const double tt1 = timing ? now() : 0;
  rc = huff_stage(ctx, s, packed, *lay_in, coef_total, (int16_t *)s.d_coef, up);
  if (rc) return rc;
  const double tt2 = timing ? now() : 0;
  if (timing) (void)hipStreamSynchronize(up);
  const double tt3 = timing ? now() : 0;
  jb_device_batch b;
  memset(&b, 0, sizeof b);
  b.desc = *desc;
  b.n_images = n_images;
  b.d_coef = (const int16_t *)s.d_coef;
  b.coef_image_stride = g.coef_bytes;
  b.d_qtabs = s.d_q;
  b.qtab_image_stride = n_images > 1 ? 768 : 0;
  b.d_rgb = dst_device ? rgb : (uint8_t *)s.d_rgb;
  b.rgb_row_stride = dev_stride;
  b.rgb_image_stride = g.rgb_bytes;
  rc = jb_blocks_to_rgb_device(ctx, &b, up);
  if (rc) return rc;
  // the status words travel with the pixels: into the caller's (pinned) words when it keeps its own
  // -- many threads share this ring, a slot's words may be recycled before their owner looks -- else
  // into the slot's, which jb_wait / jb_poll check
  const double tt4 = timing ? now() : 0;
  if (dst_device) {
    // the pixels stay on the device; only the status words come back
    JB_HIP(ctx, hipMemcpyAsync(status_out ? status_out : s.h_status, s.d_status, 4 * (size_t)n_images, hipMemcpyDeviceToHost, up));
    JB_HIP(ctx, hipEventRecord(s.done, up));
  } else {
    // the download thread issues the copies once the kernels have finished (jb_ctx::dl_*; against the copies on the
    // submission's own stream: 1,024 1080p files 6,467-7,008 -> 7,742-7,815 images/s, 128 files 4,257-4,318 ->
    // 5,767-6,567, 64 8192x8192 files 227-238 -> 262: profiles/r03/ab_download_thread.txt)
    // (the status words travel on the submission's own stream, in front of `computed`: on the engine's stream the
    // small copy and its latency would sit between every two pixel copies of the device)
    JB_HIP(ctx, hipMemcpyAsync(status_out ? status_out : s.h_status, s.d_status, 4 * (size_t)n_images, hipMemcpyDeviceToHost, up));
    JB_HIP(ctx, hipEventRecord(s.computed, up));
    DlItem it;
    it.ctx = ctx;
    it.age = ctx->dl_age.load(std::memory_order_relaxed);
    it.slot = &s;
    it.dst = rgb, it.src = s.d_rgb, it.bytes = rgb_total;
    it.rows = 0, it.dst_pitch = it.src_pitch = it.row_bytes = 0;
    if (rgb_stride != dev_stride)
      it.rows = (size_t)desc->height, it.dst_pitch = (size_t)rgb_stride, it.src_pitch = (size_t)dev_stride, it.row_bytes = (size_t)desc->width * 3;
    it.status_dst = nullptr;
    it.status_src = nullptr;
    it.status_bytes = 0;
    rc = dl_enqueue(ctx, it);
    if (rc) return rc;
  }
  if (timing)
    fprintf(stderr, "submit (device entropy): pack %.3f ms, upload + launches issued %.3f ms, entropy kernels done after %.3f ms more, pixel kernel + download call %.3f ms\n",
            (tt1 - tt0) * 1e3, (tt2 - tt1) * 1e3, (tt3 - tt2) * 1e3, (tt4 - tt3) * 1e3);
  s.busy = true;
  ctx->n_device_entropy += n_images;
  s.n_status = status_out ? 0 : n_images;
  s.ticket = ctx->next_ticket++;
  if (ctx->next_ticket < 0) ctx->next_ticket = 1;
  *ticket = s.ticket;
  ctx->next_slot = (ctx->next_slot + 1) % ctx->n_slots;
  return JB_OK;
}

// after a submission has completed: did the device entropy decoder flag any of its images?
int check_status(jb_ctx *ctx, Slot &s) {
  const int n = s.n_status;
  s.n_status = 0;
  for (int i = 0; i < n; i++)
    if (s.h_status[i])
      return fail(ctx, JB_ERR_FORMAT, "device entropy decoder: corrupt entropy-coded data in image %d of the submission (status %u)", i, s.h_status[i]);
  return JB_OK;
}

}  // namespace

int jb_entropy_decode_device(jb_ctx *ctx, const uint8_t *jpeg, size_t jpeg_bytes, jb_image_desc *desc, uint16_t *qtabs,
                             int16_t *d_coef, size_t coef_cap_bytes) {
  if (!ctx) return fail(nullptr, JB_ERR_NULL, "jb_entropy_decode_device: ctx is NULL");
  if (!jpeg || !desc || !d_coef) return fail(ctx, JB_ERR_NULL, "jb_entropy_decode_device: NULL pointer");
  JbHuffJob *job = new (std::nothrow) JbHuffJob();
  if (!job) return fail(ctx, JB_ERR_CAPACITY, "out of host memory");
  std::string err;
  int rc = jb_huff_prepare_(jpeg, jpeg_bytes, job, &err, ctx->knobs.chunk_bytes);
  if (rc == JB_OK && (size_t)job->geo.coef_bytes > coef_cap_bytes) rc = JB_ERR_CAPACITY, err = "coefficient buffer too small";
  if (rc == JB_OK && ((uintptr_t)d_coef & 15)) rc = JB_ERR_GEOMETRY, err = "coefficient pointer must be a multiple of 16 bytes";
  if (rc != JB_OK) {
    delete job;
    return fail(ctx, rc, "%s", err.c_str());
  }
  *desc = job->desc;
  if (qtabs) memcpy(qtabs, job->qtabs, sizeof job->qtabs);
  DeviceGuard guard(ctx->device);
  Slot &s = ctx->huff_aux;
  const JbHuffJob *jobs[1] = {job};
  JbHuffLayout lay;
  rc = pack_into_slot(ctx, s, jobs, 1, job->geo.coef_bytes, &lay);
  const size_t coef_bytes = (size_t)job->geo.coef_bytes;
  delete job;  // (everything it held is in the pinned blob now)
  if (rc) return rc;
  // The chunks fall into step within the launches of the first attempt on ordinary data; dense adversarial data
  // (hardly any EOB to meet at) can need a workgroup's state handed on more often -- status bit 2 alone says "not
  // yet": one retry with more launches, then the caller is told (JB_ERR_FORMAT) and the host decoder
  // (jb_entropy_decode) is the authority.  Corrupt data (bits 0, 1) is never retried.
  int launches = kJbSyncLaunches;
  for (;;) {
    rc = huff_stage(ctx, s, s.h_blob, lay, coef_bytes, d_coef, ctx->stream, launches);
    if (rc) return rc;
    JB_HIP(ctx, hipMemcpyAsync(s.h_status, s.d_status, 4, hipMemcpyDeviceToHost, ctx->stream));
    JB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (s.h_status[0] != 4u || launches >= kJbSyncLaunchesMax) break;
    launches = kJbSyncLaunchesMax;
  }
  s.n_status = 1;
  return check_status(ctx, s);
}

// decode(bytes) with the entropy stage on the device: one prepared image through the ring
// (used by jb_decode_memory, jb_frontend.cpp); the staging ring follows the frame
int jb_decode_job_(jb_ctx *ctx, const JbHuffJob *job, uint8_t *rgb, int64_t rgb_stride) {
  const bool timing = ctx->knobs.timing == 1;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = timing ? now() : 0;
  int rc = jb_ctx_reserve(ctx, (size_t)job->geo.coef_bytes, (size_t)job->geo.rgb_bytes);
  if (rc) return rc;
  const double t1 = timing ? now() : 0;
  int ticket = -1;
  const JbHuffJob *jobs[1] = {job};
  rc = submit_jobs_impl(ctx, jobs, nullptr, nullptr, nullptr, nullptr, 1, rgb, rgb_stride, nullptr, &ticket);
  if (rc) return rc;
  const double t2 = timing ? now() : 0;
  rc = jb_wait(ctx, ticket);
  if (timing) fprintf(stderr, "jb_decode_job_: reserve %.3f ms, pack + submit %.3f ms, wait %.3f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3, (now() - t2) * 1e3);
  return rc;
}

// several prepared images of ONE geometry in one submission (jb_batch.cpp); pixels contiguous, tight rows
int jb_submit_packed_(jb_ctx *ctx, const jb_image_desc *desc, const uint16_t *qtabs, const uint8_t *packed, const JbHuffLayout *lay,
                      uint8_t *rgb, uint32_t *status_out, int *ticket, int dst_device) {
  return submit_jobs_impl(ctx, nullptr, packed, lay, desc, qtabs, lay->n, rgb, 3LL * desc->width, status_out, ticket, dst_device != 0);
}

// jb_submit_batch with the pixels left in DEVICE memory of the context's device (jb_batch.cpp)
int jb_submit_batch_dev_(jb_ctx *ctx, const jb_image_desc *desc, int n_images, const int16_t *coef, const uint16_t *qtabs,
                         uint8_t *d_rgb, int *ticket) {
  return submit_impl(ctx, desc, n_images, coef, qtabs, d_rgb, 3LL * desc->width, ticket, true);
}

int jb_submit(jb_ctx *ctx, const jb_image_desc *desc, const int16_t *coef, const uint16_t *qtabs,
              uint8_t *rgb, int64_t rgb_stride, int *ticket) {
  if (!ctx) return fail(nullptr, JB_ERR_NULL, "jb_submit: ctx is NULL");
  if (!desc || !coef || !qtabs || !rgb || !ticket) return fail(ctx, JB_ERR_NULL, "jb_submit: NULL pointer");
  return submit_impl(ctx, desc, 1, coef, qtabs, rgb, rgb_stride, ticket);
}

int jb_submit_batch(jb_ctx *ctx, const jb_image_desc *desc, int n_images, const int16_t *coef,
                    const uint16_t *qtabs, uint8_t *rgb, int *ticket) {
  if (!ctx) return fail(nullptr, JB_ERR_NULL, "jb_submit_batch: ctx is NULL");
  if (!desc || !coef || !qtabs || !rgb || !ticket) return fail(ctx, JB_ERR_NULL, "jb_submit_batch: NULL pointer");
  return submit_impl(ctx, desc, n_images, coef, qtabs, rgb, 3LL * desc->width, ticket);
}

int jb_wait(jb_ctx *ctx, int ticket) {
  if (!ctx) return fail(nullptr, JB_ERR_NULL, "jb_wait: ctx is NULL");
  DeviceGuard guard(ctx->device);
  for (int i = 0; i < ctx->n_slots; i++) {
    Slot &s = ctx->slots[i];
    if (s.ticket == ticket) {
      if (s.busy) {
        const int rc = slot_finish(ctx, s);
        if (rc) return rc;
        s.busy = false;
      }
      return check_status(ctx, s);
    }
  }
  return fail(ctx, JB_ERR_STATE, "ticket %d is not in flight (already waited for and its slot reused?)", ticket);
}

int jb_poll(jb_ctx *ctx, int ticket) {
  if (!ctx) return fail(nullptr, JB_ERR_NULL, "jb_poll: ctx is NULL");
  DeviceGuard guard(ctx->device);
  for (int i = 0; i < ctx->n_slots; i++) {
    Slot &s = ctx->slots[i];
    if (s.ticket == ticket) {
      if (!s.busy) return JB_OK;
      if (s.dl_pending.load(std::memory_order_acquire)) return JB_PENDING;
      hipError_t e = hipEventQuery(s.done);
      if (e == hipErrorNotReady) return JB_PENDING;
      if (e != hipSuccess) return fail(ctx, JB_ERR_HIP, "hipEventQuery: %s", hipGetErrorString(e));
      s.busy = false;
      return check_status(ctx, s);
    }
  }
  return fail(ctx, JB_ERR_STATE, "ticket %d is not in flight (already waited for and its slot reused?)", ticket);
}

int jb_blocks_to_rgb(jb_ctx *ctx, const jb_image_desc *desc, const int16_t *coef, const uint16_t *qtabs,
                     uint8_t *rgb, int64_t rgb_stride) {
  int ticket = -1;
  int rc = jb_submit(ctx, desc, coef, qtabs, rgb, rgb_stride, &ticket);
  if (rc) return rc;
  return jb_wait(ctx, ticket);
}

void jb_free(void *p) { free(p); }

int jb_write_ppm(const char *path, const uint8_t *rgb, int32_t width, int32_t height, int64_t rgb_stride) {
  if (!path || !rgb) return fail(nullptr, JB_ERR_NULL, "jb_write_ppm: NULL pointer");
  if (width < 1 || height < 1 || rgb_stride < 3LL * width) return fail(nullptr, JB_ERR_GEOMETRY, "jb_write_ppm: bad geometry");
  FILE *f = fopen(path, "wb");
  if (!f) return fail(nullptr, JB_ERR_FORMAT, "cannot open %s for writing", path);
  fprintf(f, "P6\n%d %d\n255\n", width, height);
  for (int y = 0; y < height; y++)
    if (fwrite(rgb + (int64_t)y * rgb_stride, 1, (size_t)width * 3, f) != (size_t)width * 3) {
      fclose(f);
      return fail(nullptr, JB_ERR_FORMAT, "short write to %s", path);
    }
  fclose(f);
  return JB_OK;
}

int jb_write_bmp(const char *path, const uint8_t *rgb, int32_t width, int32_t height, int64_t rgb_stride) {
  if (!path || !rgb) return fail(nullptr, JB_ERR_NULL, "jb_write_bmp: NULL pointer");
  if (width < 1 || height < 1 || rgb_stride < 3LL * width) return fail(nullptr, JB_ERR_GEOMETRY, "jb_write_bmp: bad geometry");
  const int64_t row_bytes = (3LL * width + 3) & ~3LL;  // rows are padded to 4 bytes
  const int64_t file_bytes = 54 + row_bytes * height;
  if (file_bytes > 0xffffffffLL) return fail(nullptr, JB_ERR_CAPACITY, "jb_write_bmp: %dx%d exceeds the 4 GiB BMP limit", width, height);
  FILE *f = fopen(path, "wb");
  if (!f) return fail(nullptr, JB_ERR_FORMAT, "cannot open %s for writing", path);
  uint8_t hdr[54] = {'B', 'M'};
  auto le32 = [&](int at, uint32_t v) { for (int i = 0; i < 4; i++) hdr[at + i] = (uint8_t)(v >> (8 * i)); };
  le32(2, (uint32_t)file_bytes);
  le32(10, 54);                    // offset of the pixel array
  le32(14, 40);                    // BITMAPINFOHEADER
  le32(18, (uint32_t)width);
  le32(22, (uint32_t)height);      // positive: bottom-up
  hdr[26] = 1;                     // planes
  hdr[28] = 24;                    // bits per pixel; compression 0 (BI_RGB)
  le32(34, (uint32_t)(row_bytes * height));
  le32(38, 2835);                  // 72 dpi
  le32(42, 2835);
  bool ok = fwrite(hdr, 1, sizeof hdr, f) == sizeof hdr;
  std::vector<uint8_t> row((size_t)row_bytes, 0);
  for (int y = height - 1; ok && y >= 0; y--) {
    const uint8_t *src = rgb + (int64_t)y * rgb_stride;
    for (int x = 0; x < width; x++) {
      row[3 * x + 0] = src[3 * x + 2];
      row[3 * x + 1] = src[3 * x + 1];
      row[3 * x + 2] = src[3 * x + 0];
    }
    ok = fwrite(row.data(), 1, row.size(), f) == row.size();
  }
  if (fclose(f) != 0) ok = false;
  return ok ? JB_OK : fail(nullptr, JB_ERR_FORMAT, "short write to %s", path);
}

}  // extern "C"

// jb_wait in two halves for jb_batch.cpp, where many host threads share one context: the lookup
// runs under the caller's lock, the blocking wait outside it.  The slot stays marked busy; the
// ring's own synchronisation on reuse (jb_submit) then returns at once.
void *jb_wait_begin_(jb_ctx *ctx, int ticket) {
  for (int i = 0; i < ctx->n_slots; i++)
    if (ctx->slots[i].ticket == ticket) return ctx->slots[i].busy ? (void *)&ctx->slots[i] : nullptr;
  return nullptr;  // the slot has been reused: that submission completed long ago
}
int jb_wait_block_(jb_ctx *ctx, void *slot) {
  DeviceGuard guard(ctx->device);
  return slot_finish(ctx, *(Slot *)slot);
}

// Bind the calling host thread to the CPUs of the NUMA node closest to `device` (intersected with
// the CPUs the thread may already use; nothing changes when the node is unknown, the intersection
// is empty, or JPEGBLK_NUMA=0).  The entropy threads of one rank then read their files, decode and
// write their pinned staging on the socket their GPU hangs off.  Returns the CPUs in the new mask,
// 0 = left as it was.
// Is [p, p + bytes) device memory of `device`?  (jb_batch_decoder_set_device_output: a host pointer or
// another GPU's memory would fault in the pixel kernel instead of failing here.)
int jb_check_device_region_(int device, const void *p, size_t bytes) {
  hipPointerAttribute_t a;
  memset(&a, 0, sizeof a);
  DeviceGuard guard(device);
  hipError_t e = hipPointerGetAttributes(&a, p);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(nullptr, JB_ERR_GEOMETRY, "not a pointer the HIP runtime knows (%s): device output needs device memory", hipGetErrorString(e));
  }
  if (a.type != hipMemoryTypeDevice) return fail(nullptr, JB_ERR_GEOMETRY, "device output needs device memory (hipMalloc), not host or managed memory");
  if (a.device != device) return fail(nullptr, JB_ERR_GEOMETRY, "the region is memory of device %d, the decoder drives device %d", a.device, device);
  // the last byte must belong to the same allocation
  hipPointerAttribute_t b;
  memset(&b, 0, sizeof b);
  e = hipPointerGetAttributes(&b, (const uint8_t *)p + (bytes ? bytes - 1 : 0));
  if (e != hipSuccess || b.type != hipMemoryTypeDevice || b.device != device) {
    (void)hipGetLastError();
    return fail(nullptr, JB_ERR_GEOMETRY, "the region of %zu bytes reaches beyond its device allocation", bytes);
  }
  return JB_OK;
}

// which CPUs the threads of a decoder on `device` are bound to (n = 0: none): worked out once per (device, knob) --
// sysfs, the cgroup quota and the affinity mask of the first caller, 16 threads of every pass of every run asked for
// them again -- then only applied
static int numa_cpus_for_(int device, int numa_knob, cpu_set_t *out) {
  if (numa_knob == 0) return 0;
  const bool forced = numa_knob == 1;
  const int node = jb_device_numa_node(device);
  if (node < 0) return 0;
  char path[96];
  snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
  FILE *f = fopen(path, "r");
  if (!f) return 0;
  char list[4096] = {0};
  const bool got = fgets(list, sizeof list, f) != nullptr;
  fclose(f);
  if (!got) return 0;
  cpu_set_t cur, want;
  if (sched_getaffinity(0, sizeof cur, &cur) != 0) return 0;
  CPU_ZERO(&want);
  int n = 0;
  for (char *p = list; *p;) {  // "0-15,128-143"
    char *end;
    long a = strtol(p, &end, 10), b = a;
    if (end == p) break;
    if (*end == '-') b = strtol(end + 1, &end, 10);
    for (long c = a; c <= b && c < CPU_SETSIZE; c++)
      if (c >= 0 && CPU_ISSET((int)c, &cur)) {
        CPU_SET((int)c, &want);
        n++;
      }
    p = (*end == ',') ? end + 1 : end;
    if (*end != ',') break;
  }
  if (n == 0 || n == CPU_COUNT(&cur)) return 0;  // nothing to narrow
  // Only where the process owns at least a node's worth of CPU time (a rank of a dedicated node).
  // Under a cgroup CPU quota smaller than the node -- a share of a machine other tenants use too --
  // the scheduler does better unpinned: measured on a 16-CPU share of a 256-CPU box, 16 entropy
  // threads on 8192x8192 files: 139 images/s free, 114 bound to the GPU's node (JPEGBLK_NUMA=1 forces).
  if (!forced) {
    long quota = -1, period = 100000;
    if (FILE *q = fopen("/sys/fs/cgroup/cpu.max", "r")) {
      char qs[32] = {0};
      if (fscanf(q, "%31s %ld", qs, &period) >= 1 && strcmp(qs, "max") != 0) quota = atol(qs);
      fclose(q);
    }
    if (quota > 0 && period > 0 && quota / period < n) return 0;
  }
  *out = want;
  return n;
}

int jb_bind_thread_near_device_(int device, int numa_knob) {  // numa_knob: JbKnobs::numa of the calling decoder
  struct Entry {
    int n;
    cpu_set_t set;
  };
  static std::mutex mu;
  static std::map<std::pair<int, int>, Entry> cache;
  Entry e;
  {
    std::lock_guard<std::mutex> g(mu);
    auto it = cache.find({device, numa_knob});
    if (it == cache.end()) {
      Entry fresh;
      CPU_ZERO(&fresh.set);
      fresh.n = numa_cpus_for_(device, numa_knob, &fresh.set);
      it = cache.emplace(std::make_pair(device, numa_knob), fresh).first;
    }
    e = it->second;
  }
  if (e.n <= 0) return 0;
  if (sched_setaffinity(0, sizeof e.set, &e.set) != 0) return 0;
  return e.n;
}

void jb_ctx_set_last_desc_(jb_ctx *ctx, const jb_image_desc *d) { ctx->last_desc = *d; }
const JbKnobs *jb_ctx_knobs_(const jb_ctx *ctx) { return &ctx->knobs; }

// used by jb_frontend.cpp to report through the same channel
int jb_fail_(jb_ctx *ctx, int code, const char *msg) { return fail(ctx, code, "%s", msg); }

// ---- packing of device-entropy submissions: pure host code (see jb_huff.h) ----------------------
size_t jb_huff_pack_size_(const JbHuffJob *const *jobs, int n) {
  size_t n_wg = 0, n_starts = 0, scan_bytes = 0, n_chunks = 0;
  for (int i = 0; i < n; i++) {
    n_wg += (jobs[i]->img.n_chunks + kJbOwnChunks - 1) / kJbOwnChunks;
    n_chunks += jobs[i]->img.n_chunks;
    n_starts += jobs[i]->starts.size();
    scan_bytes += ((jobs[i]->scan.size() + 15) & ~(size_t)15);
  }
  // (every workgroup twice: the list of all of them and the list of the ones that synchronise)
  return (size_t)n * sizeof(JbHuffImage) + 2 * n_wg * sizeof(JbHuffWg) + (size_t)n * sizeof(JbHuffTables) + n_starts * 4 +
         n_chunks * sizeof(JbChunkDesc) + scan_bytes + 512;
}

int jb_huff_pack_(const JbHuffJob *const *jobs, int n, int64_t coef_stride, uint8_t *h, JbHuffLayout *lay) {
  auto a16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
  std::vector<int> set_of((size_t)n, 0);
  std::vector<int> sets;  // index of the first job that owns each distinct table set
  size_t n_wg = 0, n_sync_wg = 0, n_starts = 0, scan_bytes = 0, n_chunks = 0;
  lay->max_chunk_bytes = 0;
  lay->max_tabs = 0;
  for (int i = 0; i < n; i++) {
    int found = -1;
    for (size_t k = 0; k < sets.size() && found < 0; k++)
      if (jobs[sets[k]]->n_tabs == jobs[i]->n_tabs && memcmp(&jobs[sets[k]]->tables, &jobs[i]->tables, sizeof(JbHuffTables)) == 0) found = (int)k;
    if (found < 0) {
      found = (int)sets.size();
      sets.push_back(i);
    }
    set_of[(size_t)i] = found;
    const size_t wgs = (jobs[i]->img.n_chunks + kJbOwnChunks - 1) / kJbOwnChunks;
    n_wg += wgs;
    if (jobs[i]->img.needs_sync) n_sync_wg += wgs;
    n_chunks += jobs[i]->img.n_chunks;
    n_starts += jobs[i]->starts.size();
    scan_bytes += a16(jobs[i]->scan.size());
    if (jobs[i]->img.chunk_bytes > lay->max_chunk_bytes) lay->max_chunk_bytes = jobs[i]->img.chunk_bytes;
    if (jobs[i]->n_tabs > lay->max_tabs) lay->max_tabs = jobs[i]->n_tabs;
    // (a frame that changed between the passes of a batch decoder must not write beyond its slot)
    if ((int64_t)jobs[i]->geo.coef_bytes > coef_stride) return JB_ERR_CAPACITY;
  }
  lay->off_img = 0;
  lay->off_wg = a16((size_t)n * sizeof(JbHuffImage));
  lay->off_sync_wg = a16(lay->off_wg + n_wg * sizeof(JbHuffWg));
  lay->off_tab = a16(lay->off_sync_wg + n_sync_wg * sizeof(JbHuffWg));
  lay->off_starts = lay->off_tab + sets.size() * sizeof(JbHuffTables);
  lay->off_chunks = a16(lay->off_starts + n_starts * 4);
  lay->off_scan = a16(lay->off_chunks + n_chunks * sizeof(JbChunkDesc));
  lay->total = lay->off_scan + scan_bytes + 256;  // (a lane reads up to 16 dwords beyond its chunk's last byte)
  // device-only scratch behind the uploaded bytes
  lay->off_entry = a16(lay->total);
  lay->off_exit = a16(lay->off_entry + n_chunks * sizeof(JbChunkState));
  lay->off_cps = a16(lay->off_exit + n_chunks * sizeof(JbChunkState));
  lay->off_chunk_dc = (a16(lay->off_cps + n_chunks * kJbCheckpoints * 4) + 31) & ~(size_t)31;
  lay->off_wgsum = lay->off_chunk_dc + n_chunks * sizeof(JbChunkDc);
  lay->device_total = a16(lay->off_wgsum + n_wg * sizeof(JbWgSum));
  lay->n = n;
  lay->n_wg = (int)n_wg;
  lay->n_sync_wg = (int)n_sync_wg;
  lay->n_chunks = (uint32_t)n_chunks;
  lay->coef_stride = coef_stride;
  if (lay->device_total > 0xffffff00u || n_wg > 0x7fffffffu || n_chunks > 0x3fffffffu) return JB_ERR_CAPACITY;
  JbHuffImage *im = (JbHuffImage *)(h + lay->off_img);
  JbHuffWg *wg = (JbHuffWg *)(h + lay->off_wg);
  JbHuffWg *swg = (JbHuffWg *)(h + lay->off_sync_wg);
  uint32_t *st = (uint32_t *)(h + lay->off_starts);
  size_t w = 0, sw = 0, si = 0, sc = lay->off_scan, chunk0 = 0;
  for (size_t k = 0; k < sets.size(); k++) memcpy(h + lay->off_tab + k * sizeof(JbHuffTables), &jobs[sets[k]]->tables, sizeof(JbHuffTables));
  for (int i = 0; i < n; i++) {
    const JbHuffJob &j = *jobs[i];
    im[i] = j.img;
    im[i].scan_off = (uint32_t)(sc - lay->off_scan);
    im[i].int_off = (uint32_t)si;
    im[i].table_set = (uint32_t)set_of[(size_t)i];
    im[i].coef_off = (int64_t)i * coef_stride;
    im[i].state_off = (uint32_t)chunk0;
    im[i].wg0 = (uint32_t)w;
    // the chunks of every restart interval, from the interval's first byte (jb_chunks_of_)
    JbChunkDesc *cd = (JbChunkDesc *)(h + lay->off_chunks) + chunk0;
    uint32_t c = 0;
    for (uint32_t seg = 0; seg + 1 < (uint32_t)j.starts.size(); seg++) {
      const uint32_t k = jb_chunks_of_(j.starts[seg + 1] - j.starts[seg], j.img.chunk_bytes);
      if (c + k > j.img.n_chunks) return JB_ERR_STATE;
      for (uint32_t q = 0; q < k; q++) cd[c++] = JbChunkDesc{j.starts[seg] + q * j.img.chunk_bytes, seg | (q == 0 ? 0x80000000u : 0u)};
    }
    if (c != j.img.n_chunks) return JB_ERR_STATE;
    chunk0 += j.img.n_chunks;
    for (uint32_t f = 0; f < j.img.n_chunks; f += kJbOwnChunks) {
      wg[w++] = JbHuffWg{(uint32_t)i, f};
      if (j.img.needs_sync) swg[sw++] = JbHuffWg{(uint32_t)i, f};
    }
    memcpy(st + si, j.starts.data(), j.starts.size() * 4);
    si += j.starts.size();
    memcpy(h + sc, j.scan.data(), j.scan.size());
    sc += a16(j.scan.size());
  }
  memset(h + sc, 0, 256);
  return JB_OK;
}
