// tools/probe_pcie.hip -- host<->device copy rates with pinned memory (not product code):
// H2D alone, D2H alone, and both at once on two streams (is the link full duplex for us?).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t n = 201326592;  // one 8192x8192 4:2:0 image: coefficients = pixels = 201 MB
  void *h_in, *h_out, *d_in, *d_out;
  (void)hipHostMalloc(&h_in, n, hipHostMallocDefault);
  (void)hipHostMalloc(&h_out, n, hipHostMallocDefault);
  (void)hipMalloc(&d_in, n);
  (void)hipMalloc(&d_out, n);
  memset(h_in, 1, n);
  memset(h_out, 2, n);
  hipStream_t s0, s1;
  (void)hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
  (void)hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  const int reps = 20;
  for (int mode = 0; mode < 4; mode++) {
    for (int warm = 0; warm < 2; warm++) {
      (void)hipDeviceSynchronize();
      double t0 = now();
      for (int i = 0; i < reps; i++) {
        if (mode == 0 || mode >= 2) (void)hipMemcpyAsync(d_in, h_in, n, hipMemcpyHostToDevice, s0);
        if (mode == 1) (void)hipMemcpyAsync(h_out, d_out, n, hipMemcpyDeviceToHost, s0);
        if (mode == 2) (void)hipMemcpyAsync(h_out, d_out, n, hipMemcpyDeviceToHost, s1);
        if (mode == 3) (void)hipMemcpyAsync(h_out, d_out, n, hipMemcpyDeviceToHost, s0);
      }
      (void)hipDeviceSynchronize();
      double dt = now() - t0;
      if (warm) {
        const char *names[] = {"H2D alone", "D2H alone", "H2D + D2H on two streams", "H2D + D2H on one stream"};
        double bytes = (double)n * reps * (mode >= 2 ? 2 : 1);
        printf("%-28s %.1f GB/s total\n", names[mode], bytes / dt / 1e9);
      }
    }
  }
  return 0;
}
