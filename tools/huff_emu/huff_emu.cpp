// huff_emu -- the device entropy decoder's kernels (jpeg_decoder_amd/csrc/jb_huff.hip, included below as
// they are) run on the host through the SIMT shim of tools/huff_emu/hip/, on JPEG files, and compared
// with the host decoder (jb_entropy_decode, itself pinned to the reference's coefficient dumps).
// Test infrastructure: checks the kernels' logic where there is no GPU, and counts what the
// synchronisation costs (steps per pass) for design work.  Not part of the product.
//   huff_emu [--launches N] [--quiet] [--strict] file.jpg [file.jpg ...]     (all files in ONE submission)
// exit code 0: whatever the kernels accept (status 0) the host decoder accepts too, with the same coefficients.
// An image the kernels flag goes back to the host decoder in the product, so a flag on a stream the host
// accepts is allowed (damaged streams with bytes behind an interval's last block, say) -- unless --strict:
// then every image the host decoder accepts must come through with status 0 (undamaged files).
#include <stdio.h>
#include <stdlib.h>

#include <atomic>
#include <string>
#include <vector>

#include <map>
#include <mutex>
// steps of this lane since its last pass ended; the passes of this lane in this kernel
static thread_local uint32_t tl_steps = 0;
static thread_local std::vector<uint32_t> tl_passes;
#define JBH_TRACE_STEP() (tl_steps++)
static thread_local std::vector<int> tl_pass_no;
#define JBH_TRACE_PASS_END(pass) (tl_passes.push_back(tl_steps), tl_pass_no.push_back((int)(pass)), tl_steps = 0)

#include "../../jpeg_decoder_amd/csrc/jb_huff.hip"
#include "../../jpeg_decoder_amd/csrc/jb_knobs.h"

alignas(16) uint8_t lds[160 * 1024];
// per launch (in order): lane-steps, and wave-steps = sum over the waves' passes of the longest lane (what a wave pays)
static std::mutex g_mu;
static int g_launch = -1;
static std::map<std::pair<int, int>, std::vector<uint32_t>> g_wave_max;  // (block, wave) -> per pass: max over lanes
static std::vector<long long> g_lane_steps, g_wave_steps, g_wave_passes;
struct PassStat {
  long long lanes = 0, steps = 0, hist[5] = {0, 0, 0, 0, 0};
};
static std::map<std::pair<int, int>, PassStat> g_by_pass;  // (launch, pass)
static void emu_merge(unsigned b, unsigned t) {
  std::lock_guard<std::mutex> lock(g_mu);
  auto &v = g_wave_max[{(int)b, (int)(t >> 6)}];
  if (v.size() < tl_passes.size()) v.resize(tl_passes.size(), 0);
  for (size_t i = 0; i < tl_passes.size(); i++) {
    g_lane_steps[(size_t)g_launch] += tl_passes[i];
    if (tl_passes[i] > v[i]) v[i] = tl_passes[i];
    auto &ps = g_by_pass[{g_launch, tl_pass_no[i]}];
    if (tl_passes[i]) ps.lanes++, ps.steps += tl_passes[i];
    const unsigned bucket = tl_passes[i] == 0 ? 0 : tl_passes[i] < 32 ? 1 : tl_passes[i] < 64 ? 2 : tl_passes[i] < 128 ? 3 : 4;
    ps.hist[bucket]++;
  }
}
namespace emu {
thread_local Idx tl_thread, tl_block, tl_grid;
thread_local Group *tl_group = nullptr;
void launch(dim3 grid, dim3 block, const std::function<void()> &body) {
  g_launch++;
  g_lane_steps.push_back(0), g_wave_steps.push_back(0), g_wave_passes.push_back(0);
  g_wave_max.clear();
  struct Sum {
    ~Sum() {
      for (auto &kv : g_wave_max)
        for (uint32_t m : kv.second) g_wave_steps[(size_t)g_launch] += m, g_wave_passes[(size_t)g_launch] += m ? 1 : 0;
    }
  } sum_at_exit;
  for (unsigned b = 0; b < grid.x; b++) {
    Group group((int)block.x);
    std::vector<std::thread> th;
    for (unsigned t = 0; t < block.x; t++)
      th.emplace_back([&, t] {
        tl_thread.x = t;
        tl_block.x = b;
        tl_grid.x = grid.x;
        tl_group = &group;
        tl_steps = 0;
        tl_passes.clear();
        tl_pass_no.clear();
        body();
        emu_merge(b, t);
        // a lane that returns early still has to let the others through their barriers: the kernels
        // only return uniformly, so nothing to do here
      });
    for (auto &x : th) x.join();
  }
}
}  // namespace emu

static std::vector<uint8_t> read_file(const char *path) {
  std::vector<uint8_t> v;
  FILE *f = fopen(path, "rb");
  if (!f) return v;
  uint8_t buf[65536];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + n);
  fclose(f);
  return v;
}

int main(int argc, char **argv) {
  int launches = kJbSyncLaunches;
  bool quiet = false, strict = false;
  std::vector<const char *> paths;
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "--launches") && i + 1 < argc) launches = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--quiet")) quiet = true;
    else if (!strcmp(argv[i], "--strict")) strict = true;
    else paths.push_back(argv[i]);
  }
  if (paths.empty()) {
    fprintf(stderr, "usage: huff_emu [--launches N] [--quiet] file.jpg ...\n");
    return 2;
  }
  std::vector<std::unique_ptr<JbHuffJob>> jobs;
  std::vector<std::vector<uint8_t>> files;
  int64_t stride = 0;
  for (const char *p : paths) {
    files.push_back(read_file(p));
    std::unique_ptr<JbHuffJob> job(new JbHuffJob());
    std::string err;
    const int rc = jb_huff_prepare_(files.back().data(), files.back().size(), job.get(), &err, jb_knobs_read().chunk_bytes);
    if (rc != JB_OK) {
      printf("%s: not taken by the device decoder (%d: %s)\n", p, rc, err.c_str());
      files.pop_back();
      continue;
    }
    if (job->geo.coef_bytes > stride) stride = job->geo.coef_bytes;
    jobs.push_back(std::move(job));
  }
  if (jobs.empty()) return 0;
  std::vector<const JbHuffJob *> ptrs;
  for (auto &j : jobs) ptrs.push_back(j.get());
  const int n = (int)ptrs.size();
  std::vector<uint8_t> blob(jb_huff_pack_size_(ptrs.data(), n));
  JbHuffLayout lay;
  int rc = jb_huff_pack_(ptrs.data(), n, stride, blob.data(), &lay);
  if (rc != JB_OK) {
    printf("pack failed: %d\n", rc);
    return 1;
  }
  blob.resize(lay.device_total + 64);
  memset(blob.data() + lay.total, 0xa5, lay.device_total - lay.total);  // device scratch starts as garbage
  std::vector<int16_t> coef((size_t)stride / 2 * (size_t)n, 0);
  std::vector<uint32_t> status((size_t)n, 0);
  JbHuffLaunch p;
  memset(&p, 0, sizeof p);
  uint8_t *d = blob.data();
  p.scan = d + lay.off_scan;
  p.starts = (const uint32_t *)(d + lay.off_starts);
  p.tables = (const JbHuffTables *)(d + lay.off_tab);
  p.images = (const JbHuffImage *)(d + lay.off_img);
  p.wgs = (const JbHuffWg *)(d + lay.off_wg);
  p.sync_wgs = (const JbHuffWg *)(d + lay.off_sync_wg);
  p.coef = coef.data();
  p.status = status.data();
  p.n_wgs = lay.n_wg;
  p.n_sync_wgs = lay.n_sync_wg;
  p.chunks = (const JbChunkDesc *)(d + lay.off_chunks);
  p.entry = (JbChunkState *)(d + lay.off_entry);
  p.exit = (JbChunkState *)(d + lay.off_exit);
  p.cps = (uint32_t *)(d + lay.off_cps);
  p.chunk_dc = (JbChunkDc *)(d + lay.off_chunk_dc);
  p.wgsum = (JbWgSum *)(d + lay.off_wgsum);
  p.n_chunks_total = lay.n_chunks;
  p.sync_launches = launches;
  p.max_chunk_bytes = lay.max_chunk_bytes;
  p.max_tabs = lay.max_tabs;
  (void)jbk_huff_launch(p, nullptr);
  int bad = 0;
  for (int i = 0; i < n; i++) {
    const JbHuffJob &j = *jobs[(size_t)i];
    jb_image_desc desc;
    uint16_t q[256];
    std::vector<int16_t> want((size_t)j.geo.coef_bytes / 2);
    const int hrc = jb_entropy_decode(files[(size_t)i].data(), files[(size_t)i].size(), &desc, q, want.data(), want.size() * 2);
    const int16_t *got = coef.data() + (size_t)i * (size_t)stride / 2;
    size_t diff = 0, first = 0;
    if (hrc == JB_OK)
      for (size_t k = 0; k < want.size(); k++)
        if (got[k] != want[k]) { if (!diff++) first = k; if (getenv("EMU_DIFFS") && diff < 40) printf("  block %zu idx %zu got %d want %d\n", k / 64, k % 64, got[k], want[k]); }
    const bool ok = hrc == JB_OK ? (status[(size_t)i] == 0 ? diff == 0 : !strict) : status[(size_t)i] != 0;
    if (!ok) bad++;
    if (!quiet || !ok)
      printf("%s: %ux%u %dx%d, %u chunks of %u bytes in %u workgroups, sync %u: status %u, host rc %d, %zu coefficients differ%s -> %s\n", paths[(size_t)i],
             (unsigned)j.desc.width, (unsigned)j.desc.height, j.desc.hs, j.desc.vs, j.img.n_chunks, j.img.chunk_bytes,
             (j.img.n_chunks + kJbHuffLanes - 1) / kJbHuffLanes, j.img.needs_sync, status[(size_t)i], hrc, diff,
             diff ? (" (first at " + std::to_string(first) + ")").c_str() : "", ok ? "ok" : "MISMATCH");
  }
  if (!quiet)
    for (size_t l = 0; l < g_lane_steps.size(); l++)
      printf("launch %zu: %lld lane-steps, %lld wave-steps in %lld wave-passes (64 x wave-steps = %.2f x lane-steps)\n", l, g_lane_steps[l], g_wave_steps[l],
             g_wave_passes[l], g_lane_steps[l] ? 64.0 * (double)g_wave_steps[l] / (double)g_lane_steps[l] : 0.0);
  if (!quiet && getenv("EMU_PASSES"))
    for (auto &kv : g_by_pass)
      printf("  launch %d pass %3d: %6lld lanes decode, %8lld steps (mean %5.1f); lanes by steps: idle %lld, <32 %lld, <64 %lld, <128 %lld, more %lld\n", kv.first.first,
             kv.first.second, kv.second.lanes, kv.second.steps, kv.second.lanes ? (double)kv.second.steps / (double)kv.second.lanes : 0.0, kv.second.hist[0],
             kv.second.hist[1], kv.second.hist[2], kv.second.hist[3], kv.second.hist[4]);
  return bad ? 1 : 0;
}
