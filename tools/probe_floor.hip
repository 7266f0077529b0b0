// tools/probe_floor.hip -- latency floor of ONE small launch, cold (not product code).
// BASELINE configs 2 and 3 are single images: a 1920x1080 4:4:4 image is 507 tiles, less than one
// wave of workgroups on 256 CUs, so the launch is all ramp and drain.  This probe measures what the
// machine gives such a launch before any arithmetic: an empty kernel of the same grid, and math-free
// kernels that move the same bytes (24 KiB read then 12 / 24 KiB written per 192-lane workgroup, or
// the same bytes cut into 64-lane workgroups), every launch on a different buffer set (> 512 MiB in
// rotation, so nothing is in the Infinity Cache).  HIP events around every launch, median of 400.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef uint32_t u3 __attribute__((ext_vector_type(3)));

__global__ __launch_bounds__(192) void empty_kernel(const uint8_t *in, uint8_t *out) {
  if (threadIdx.x == 100000) out[0] = in[0];
}

// T lanes; every lane reads its own 128 B (8 x dwordx4), then the workgroup writes WPL x 12 B per lane
// in wave-contiguous 768-B runs with nt stores: T = 192, WPL = 5.33 -> 4:4:4 (12 KiB), 10.67 -> 4:2:0
template <int T, int ROWS>
__global__ __launch_bounds__(T) void mix(const uint8_t *in, uint8_t *out) {
  const long t = blockIdx.x;
  const uint8_t *src = in + t * (T * 128) + threadIdx.x * 128;
  u4 v[8];
#pragma unroll
  for (int i = 0; i < 8; i++) v[i] = *(const u4 *)(src + i * 16);
  u4 acc = v[0] ^ v[1] ^ v[2] ^ v[3] ^ v[4] ^ v[5] ^ v[6] ^ v[7];
  uint8_t *base = out + t * (ROWS * 768);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int it = wave; it < ROWS; it += T / 64) {
    u3 w = u3{acc.x + it, acc.y, acc.z ^ acc.w};
    __builtin_nontemporal_store(w, (u3 *)(base + (long)it * 768 + lane * 12));
  }
}

int main(int argc, char **argv) {
  struct Case { const char *name; long tiles; int rows192; } cases[] = {
      {"1920x1080 4:4:4 x1 (507 tiles)", 507, 16},
      {"4096x4096 4:2:0 x1 (2048 tiles)", 2048, 32},
      {"4096x4096 4:2:0 x8 (16384 tiles)", 16384, 32},
      {"679x451 4:2:0 x1 (39 tiles)", 39, 32},
      {"679x451 4:2:0 x512 (19968 tiles)", 19968, 32},
  };
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (auto &c : cases) {
    const long rb = c.tiles * 24576, wb = c.tiles * c.rows192 * 768;
    int sets = (int)std::max(2L, (long)((600L << 20) / (rb + wb)) + 1);
    if (sets > 512) sets = 512;
    std::vector<uint8_t *> din(sets), dout(sets);
    for (int s = 0; s < sets; s++) {
      (void)hipMalloc(&din[s], rb);
      (void)hipMalloc(&dout[s], wb);
      (void)hipMemset(din[s], s + 1, rb);
    }
    auto time = [&](const char *name, auto launch) {
      std::vector<float> ms;
      for (int rep = 0; rep < 500; rep++) {
        const int s = rep % sets;
        (void)hipEventRecord(e0);
        launch(din[s], dout[s]);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float m;
        (void)hipEventElapsedTime(&m, e0, e1);
        if (rep >= 100) ms.push_back(m);
      }
      std::sort(ms.begin(), ms.end());
      const float med = ms[ms.size() / 2];
      // the same launches back to back without events in between: per-launch time of a stream
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0);
      for (int rep = 0; rep < 400; rep++) launch(din[rep % sets], dout[rep % sets]);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float tot;
      (void)hipEventElapsedTime(&tot, e0, e1);
      printf("  %-46s median %7.2f us  min %7.2f us  %5.2f TB/s | back-to-back %7.2f us/launch %5.2f TB/s\n", name, med * 1e3, ms[0] * 1e3,
             (rb + wb) / (med * 1e-3) / 1e12, tot / 400 * 1e3, (rb + wb) / (tot / 400 * 1e-3) / 1e12);
      fflush(stdout);
    };
    printf("%s: %.2f MB read + %.2f MB written per launch, %d rotating sets\n", c.name, rb / 1e6, wb / 1e6, sets);
    time("empty kernel, same grid (192 lanes)", [&](const uint8_t *i, uint8_t *o) { empty_kernel<<<c.tiles, 192>>>(i, o); });
    if (c.rows192 == 16) {
      time("math-free, 192-lane WGs (24 KiB in, 12 KiB out)", [&](const uint8_t *i, uint8_t *o) { mix<192, 16><<<c.tiles, 192>>>(i, o); });
      time("math-free, 64-lane WGs (8 KiB in, 3.75 KiB out)", [&](const uint8_t *i, uint8_t *o) { mix<64, 5><<<c.tiles * 3, 64>>>(i, o); });
    } else {
      time("math-free, 192-lane WGs (24 KiB in, 24 KiB out)", [&](const uint8_t *i, uint8_t *o) { mix<192, 32><<<c.tiles, 192>>>(i, o); });
      time("math-free, 64-lane WGs (8 KiB in, 7.5 KiB out)", [&](const uint8_t *i, uint8_t *o) { mix<64, 10><<<c.tiles * 3, 64>>>(i, o); });
    }
    time("hipMemcpyAsync D2D of the written bytes", [&](const uint8_t *i, uint8_t *o) { (void)hipMemcpyAsync(o, i, std::min(rb, wb), hipMemcpyDeviceToDevice, 0); });
    for (int s = 0; s < sets; s++) {
      (void)hipFree(din[s]);
      (void)hipFree(dout[s]);
    }
  }
  return 0;
}
