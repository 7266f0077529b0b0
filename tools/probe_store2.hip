// tools/probe_store2.hip -- why does hipMemsetAsync write at 6.2 TB/s when one-shot store kernels
// reach 5.3-5.7?  Probe (not product code): linear fills of 403 MB with different grid shapes.
//   A. one-shot: every workgroup writes one 4 KiB / 16 KiB / 64 KiB piece (grid = bytes / piece)
//   B. grid-stride: G workgroups, workgroup i writes pieces i, i+G, ...
//   C. contiguous: G workgroups, workgroup i writes one contiguous 1/G of the buffer
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

template <int PIECES>  // 4 KiB pieces per workgroup, one-shot
__global__ __launch_bounds__(256) void fill_oneshot(uint8_t *out) {
  uint8_t *base = out + (long)blockIdx.x * PIECES * 4096 + threadIdx.x * 16;
#pragma unroll
  for (int i = 0; i < PIECES; i++) *(u4 *)(base + i * 4096) = u4{1u, 2u, 3u, 4u};
}
__global__ __launch_bounds__(256) void fill_stride(uint8_t *out, long pieces) {
  for (long pc = blockIdx.x; pc < pieces; pc += gridDim.x) *(u4 *)(out + pc * 4096 + threadIdx.x * 16) = u4{1u, 2u, 3u, 4u};
}
__global__ __launch_bounds__(256) void fill_contig(uint8_t *out, long pieces_per_wg) {
  uint8_t *base = out + (long)blockIdx.x * pieces_per_wg * 4096 + threadIdx.x * 16;
  for (long i = 0; i < pieces_per_wg; i++) *(u4 *)(base + i * 4096) = u4{1u, 2u, 3u, 4u};
}
// contiguous per WAVE: each wave owns a contiguous run, writes 1 KiB per instruction
__global__ __launch_bounds__(256) void fill_contig_wave(uint8_t *out, long kib_per_wave) {
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  uint8_t *base = out + wave * kib_per_wave * 1024 + (threadIdx.x & 63) * 16;
  for (long i = 0; i < kib_per_wave; i++) *(u4 *)(base + i * 1024) = u4{1u, 2u, 3u, 4u};
}

int main() {
  const long bytes = 8L * 4096 * 4096 * 3;  // 403 MB
  const long pieces = bytes / 4096;
  uint8_t *d;
  (void)hipMalloc(&d, bytes);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  auto time = [&](const char *name, auto launch) {
    float best = 1e9, sum = 0;
    int n = 0;
    for (int rep = 0; rep < 400; rep++) {
      (void)hipEventRecord(e0);
      launch();
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      if (rep >= 300) { sum += ms; n++; if (ms < best) best = ms; }
    }
    printf("%-40s mean %.1f us  min %.1f us  %.2f TB/s\n", name, sum / n * 1e3, best * 1e3, bytes / (sum / n * 1e-3) / 1e12);
    fflush(stdout);
  };
  // settle the power state first
  for (int i = 0; i < 3000; i++) fill_oneshot<1><<<pieces, 256>>>(d);
  (void)hipDeviceSynchronize();
  time("hipMemsetAsync", [&] { (void)hipMemsetAsync(d, 7, bytes, 0); });
  time("one-shot 4 KiB/WG", [&] { fill_oneshot<1><<<pieces, 256>>>(d); });
  time("one-shot 16 KiB/WG", [&] { fill_oneshot<4><<<pieces / 4, 256>>>(d); });
  time("one-shot 64 KiB/WG", [&] { fill_oneshot<16><<<pieces / 16, 256>>>(d); });
  for (int g : {256, 512, 1024, 2048, 4096, 8192}) {
    char nm[64];
    snprintf(nm, sizeof nm, "grid-stride, %d WGs", g);
    time(nm, [&] { fill_stride<<<g, 256>>>(d, pieces); });
  }
  for (int g : {256, 512, 1024, 2048, 4096, 8192}) {
    char nm[64];
    snprintf(nm, sizeof nm, "contiguous per WG, %d WGs", g);
    time(nm, [&] { fill_contig<<<g, 256>>>(d, pieces / g); });
  }
  for (int g : {512, 2048, 8192}) {
    char nm[64];
    snprintf(nm, sizeof nm, "contiguous per wave, %d WGs", g);
    time(nm, [&] { fill_contig_wave<<<g, 256>>>(d, bytes / 1024 / (g * 4L)); });
  }
  time("hipMemsetAsync (again)", [&] { (void)hipMemsetAsync(d, 7, bytes, 0); });
  return 0;
}
