// tools/probe_d2h.hip -- device->host copy rate into pinned memory as a function of how the copies
// are issued (not product code): N streams at once, each copying `chunk` bytes per call into its own
// part of a large pinned region (the batch decoder's output arena is such a region: several
// gigabytes, every image at a new address), with and without small host->device copies beside them.
// Question: do concurrent downloads share the link as well as one download at a time?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t region = (size_t)6 << 30, dev_bytes = (size_t)1 << 30;
  uint8_t *h, *d, *h_up, *d_up;
  if (hipHostMalloc((void **)&h, region, hipHostMallocPortable) != hipSuccess) return 1;
  if (hipMalloc((void **)&d, dev_bytes) != hipSuccess) return 1;
  (void)hipHostMalloc((void **)&h_up, 64 << 20, hipHostMallocDefault);
  (void)hipMalloc((void **)&d_up, 64 << 20);
  memset(h, 1, region);
  (void)hipMemset(d, 3, dev_bytes);
  hipStream_t st[16], up;
  for (auto &s : st) (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  (void)hipStreamCreateWithFlags(&up, hipStreamNonBlocking);
  const size_t chunks[] = {(size_t)6 << 20, (size_t)50 << 20, (size_t)200 << 20, (size_t)800 << 20};
  printf("%-10s %-8s %-10s %s\n", "chunk MB", "streams", "uploads", "GB/s device->host (total)");
  for (size_t chunk : chunks)
    for (int n : {1, 2, 3, 4, 8, 16})
      for (int with_up = 0; with_up < 2; with_up++) {
        const size_t total = (size_t)4 << 30;  // bytes moved per measurement
        const size_t calls = total / chunk;
        double best = 0;
        for (int rep = 0; rep < 3; rep++) {
          (void)hipDeviceSynchronize();
          const double t0 = now();
          for (size_t c = 0; c < calls; c++) {
            const size_t at = (c * chunk) % (region - chunk + 1);
            (void)hipMemcpyAsync(h + at, d + (c * chunk) % (dev_bytes - chunk + 1), chunk, hipMemcpyDeviceToHost, st[c % (size_t)n]);
            if (with_up) (void)hipMemcpyAsync(d_up, h_up, (chunk / 8) < ((size_t)64 << 20) ? chunk / 8 : ((size_t)64 << 20), hipMemcpyHostToDevice, up);
          }
          (void)hipDeviceSynchronize();
          const double r = (double)(calls * chunk) / (now() - t0) / 1e9;
          if (r > best) best = r;
        }
        printf("%-10zu %-8d %-10s %.1f\n", chunk >> 20, n, with_up ? "1/8 beside" : "none", best);
        fflush(stdout);
      }
  return 0;
}
