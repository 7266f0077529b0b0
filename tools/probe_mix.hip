// tools/probe_mix.hip -- ceiling probe for the fused kernel's traffic mix (not product code):
// every workgroup READS R contiguous bytes (the coefficient stream) and, only after all its loads
// have returned, WRITES R/2 bytes (4:4:4: 9 B/pixel = 6 read + 3 written), nothing else.
// Varies: bytes per workgroup, threads per workgroup, dynamic LDS (caps workgroups per CU), and
// whether the writes are linear or in the kernel's tile pattern (rows of SEG bytes at the image
// pitch).  Total traffic = the bench batch: 805 MB read + 403 MB written.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef uint32_t u3 __attribute__((ext_vector_type(3)));

// LPT = 16-byte loads per thread.  Each thread then writes LPT/2 * 16 bytes... as dwordx4 stores
// (linear) -- R = T*LPT*16, W = R/2.
template <int T, int LPT>
__global__ __launch_bounds__(T) void mix_linear(const uint8_t *in, uint8_t *out) {
  extern __shared__ char dyn[];
  const uint8_t *src = in + (long)blockIdx.x * (T * LPT * 16) + threadIdx.x * 16;
  u4 v[LPT];
#pragma unroll
  for (int i = 0; i < LPT; i++) v[i] = *(const u4 *)(src + i * T * 16);
  uint8_t *dst = out + (long)blockIdx.x * (T * LPT * 8) + threadIdx.x * 16;
#pragma unroll
  for (int i = 0; i < LPT / 2; i++) *(u4 *)(dst + i * T * 16) = v[2 * i] ^ v[2 * i + 1];
  if (threadIdx.x == 100000) dyn[0] = 1;
}

// the kernel's shape: 192 threads, 24 KiB read (lane = its own 128 B, 8 x dwordx4), 12 KiB written
// as 8 rows x 1536 B at pitch (dwordx3 per lane, 768 B per wave-instruction), two bursts
template <bool NT>
__global__ __launch_bounds__(192) void mix_tile(const uint8_t *in, uint8_t *out, long pitch, int tiles_per_row) {
  extern __shared__ char dyn[];
  const int t = blockIdx.x;
  const uint8_t *src = in + (long)t * 24576 + threadIdx.x * 128;
  u4 v[8];
#pragma unroll
  for (int i = 0; i < 8; i++) v[i] = *(const u4 *)(src + i * 16);
  const int ty = t / tiles_per_row, tx = t - ty * tiles_per_row;
  uint8_t *base = out + (long)ty * 8 * pitch + (long)tx * 1536;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  u4 acc = v[0] ^ v[1] ^ v[2] ^ v[3] ^ v[4] ^ v[5] ^ v[6] ^ v[7];
  int k = 0;
  for (int it = wave; it < 16; it += 3, k++) {
    const int row = it >> 1, seg = it & 1;
    u3 w = u3{acc.x + k, acc.y, acc.z ^ acc.w};
    uint8_t *p = base + (long)row * pitch + seg * 768 + lane * 12;
    if (NT) __builtin_nontemporal_store(w, (u3 *)p);
    else *(u3 *)p = w;
  }
  if (threadIdx.x == 100000) dyn[0] = 1;
}

// the 4:4:4 shape with an L2 prefetch of a tile `ahead` tiles further on (one dword per 128-B line,
// issued after this tile's own loads have returned; the result is never used)
__global__ __launch_bounds__(192) void mix_tile_prefetch(const uint8_t *in, uint8_t *out, long pitch, int tiles_per_row,
                                                          int ahead, int n_tiles, uint32_t *sink) {
  extern __shared__ char dyn[];
  const int t = blockIdx.x;
  const uint8_t *src = in + (long)t * 24576 + threadIdx.x * 128;
  u4 v[8];
#pragma unroll
  for (int i = 0; i < 8; i++) v[i] = *(const u4 *)(src + i * 16);
  u4 acc = v[0] ^ v[1] ^ v[2] ^ v[3] ^ v[4] ^ v[5] ^ v[6] ^ v[7];
  uint32_t pf = 0;
  if (t + ahead < n_tiles) pf = *(const volatile uint32_t *)(in + (long)(t + ahead) * 24576 + threadIdx.x * 128);
  const int ty = t / tiles_per_row, tx = t - ty * tiles_per_row;
  uint8_t *base = out + (long)ty * 8 * pitch + (long)tx * 1536;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int k = 0;
  for (int it = wave; it < 16; it += 3, k++) {
    const int row = it >> 1, seg = it & 1;
    u3 w = u3{acc.x + k, acc.y, acc.z ^ acc.w};
    __builtin_nontemporal_store(w, (u3 *)(base + (long)row * pitch + seg * 768 + lane * 12));
  }
  if (pf == 0x12345678u) sink[0] = pf;  // keeps the prefetch alive
  if (threadIdx.x == 100000) dyn[0] = 1;
}

// 4:2:0 shape: 192 threads, 24 KiB read, 24 KiB written as 16 rows x 1536 B (1:2 read:write mix)
__global__ __launch_bounds__(192) void mix_tile420(const uint8_t *in, uint8_t *out, long pitch, int tiles_per_row) {
  extern __shared__ char dyn[];
  const int t = blockIdx.x;
  const uint8_t *src = in + (long)t * 24576 + threadIdx.x * 128;
  u4 v[8];
#pragma unroll
  for (int i = 0; i < 8; i++) v[i] = *(const u4 *)(src + i * 16);
  const int ty = t / tiles_per_row, tx = t - ty * tiles_per_row;
  uint8_t *base = out + (long)ty * 16 * pitch + (long)tx * 1536;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  u4 acc = v[0] ^ v[1] ^ v[2] ^ v[3] ^ v[4] ^ v[5] ^ v[6] ^ v[7];
  int k = 0;
  for (int it = wave; it < 32; it += 3, k++) {
    const int row = it >> 1, seg = it & 1;
    u3 w = u3{acc.x + k, acc.y, acc.z ^ acc.w};
    __builtin_nontemporal_store(w, (u3 *)(base + (long)row * pitch + seg * 768 + lane * 12));
  }
  if (threadIdx.x == 100000) dyn[0] = 1;
}

int main() {
  const long rbytes = 8L * 4096 * 4096 * 6, wbytes = rbytes / 2;
  uint8_t *din, *dout;
  (void)hipMalloc(&din, rbytes);
  (void)hipMalloc(&dout, wbytes);
  (void)hipMemset(din, 1, rbytes);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  auto time = [&](const char *name, auto launch) {
    float best = 1e9, sum = 0;
    int n = 0;
    for (int rep = 0; rep < 300; rep++) {
      (void)hipEventRecord(e0);
      launch();
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      if (rep >= 200) { sum += ms; n++; if (ms < best) best = ms; }
    }
    printf("%-52s mean %.1f us  min %.1f us  %.2f TB/s\n", name, sum / n * 1e3, best * 1e3, (rbytes + wbytes) / (sum / n * 1e-3) / 1e12);
    fflush(stdout);
  };
  for (int i = 0; i < 1500; i++) mix_linear<256, 2><<<rbytes / (256 * 2 * 16), 256>>>(din, dout);
  (void)hipDeviceSynchronize();
  time("hipMemcpyAsync D2D 604 MB (r+w = 1208 MB)", [&] { (void)hipMemcpyAsync(dout, din, wbytes, hipMemcpyDeviceToDevice, 0); (void)hipMemcpyAsync(dout, din + wbytes, wbytes / 2, hipMemcpyDeviceToDevice, 0); });
  time("linear T=256  R= 8 KiB/WG", [&] { mix_linear<256, 2><<<rbytes / (256 * 2 * 16), 256>>>(din, dout); });
  time("linear T=256  R=16 KiB/WG", [&] { mix_linear<256, 4><<<rbytes / (256 * 4 * 16), 256>>>(din, dout); });
  time("linear T=256  R=32 KiB/WG", [&] { mix_linear<256, 8><<<rbytes / (256 * 8 * 16), 256>>>(din, dout); });
  time("linear T=64   R= 2 KiB/WG", [&] { mix_linear<64, 2><<<rbytes / (64 * 2 * 16), 64>>>(din, dout); });
  time("linear T=64   R= 8 KiB/WG", [&] { mix_linear<64, 8><<<rbytes / (64 * 8 * 16), 64>>>(din, dout); });
  time("linear T=192  R=24 KiB/WG", [&] { mix_linear<192, 8><<<rbytes / (192 * 8 * 16), 192>>>(din, dout); });
  time("linear T=192  R=24 KiB/WG, 24 KiB LDS (6 WG/CU)", [&] { mix_linear<192, 8><<<rbytes / (192 * 8 * 16), 192, 24576>>>(din, dout); });
  time("linear T=192  R=24 KiB/WG, 40 KiB LDS (4 WG/CU)", [&] { mix_linear<192, 8><<<rbytes / (192 * 8 * 16), 192, 40960>>>(din, dout); });
  time("linear T=192  R=24 KiB/WG, 64 KiB LDS (2 WG/CU)", [&] { mix_linear<192, 8><<<rbytes / (192 * 8 * 16), 192, 65536>>>(din, dout); });
  const int ntiles = rbytes / 24576;
  time("tile pattern, plain stores", [&] { mix_tile<false><<<ntiles, 192>>>(din, dout, 12288, 8); });
  time("tile pattern, nt stores", [&] { mix_tile<true><<<ntiles, 192>>>(din, dout, 12288, 8); });
  time("tile pattern, nt stores, 24 KiB LDS (6 WG/CU)", [&] { mix_tile<true><<<ntiles, 192, 24576>>>(din, dout, 12288, 8); });
  time("tile pattern, nt stores, 40 KiB LDS (4 WG/CU)", [&] { mix_tile<true><<<ntiles, 192, 40960>>>(din, dout, 12288, 8); });
  time("tile pattern, nt stores, 64 KiB LDS (2 WG/CU)", [&] { mix_tile<true><<<ntiles, 192, 65536>>>(din, dout, 12288, 8); });
  {
    uint32_t *sink;
    (void)hipMalloc(&sink, 64);
    time("tile pattern, nt stores, 24 KiB LDS (again)", [&] { mix_tile<true><<<ntiles, 192, 24576>>>(din, dout, 12288, 8); });
    for (int ahead : {256, 512, 1024, 1536, 3072}) {
      char nm[96];
      snprintf(nm, sizeof nm, "  + L2 prefetch %d tiles ahead, 24 KiB LDS", ahead);
      time(nm, [&] { mix_tile_prefetch<<<ntiles, 192, 24576>>>(din, dout, 12288, 8, ahead, ntiles, sink); });
    }
  }
  {  // 4:2:0 batch: 8 x 4096^2: 402.7 MB read + 402.7 MB written
    const long r420 = 8L * 4096 * 4096 * 3;
    const int nt420 = r420 / 24576;
    auto time420 = [&](const char *name, auto launch) {
      float best = 1e9, sum = 0;
      int n = 0;
      for (int rep = 0; rep < 300; rep++) {
        (void)hipEventRecord(e0);
        launch();
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 200) { sum += ms; n++; if (ms < best) best = ms; }
      }
      printf("%-52s mean %.1f us  min %.1f us  %.2f TB/s\n", name, sum / n * 1e3, best * 1e3, 2.0 * r420 / (sum / n * 1e-3) / 1e12);
    };
    time420("4:2:0 mix (1:2), tile pattern, nt stores", [&] { mix_tile420<<<nt420, 192>>>(din, dout, 12288, 8); });
    time420("4:2:0 mix, 24 KiB LDS (6 WG/CU)", [&] { mix_tile420<<<nt420, 192, 24576>>>(din, dout, 12288, 8); });
    time420("4:2:0 mix, 40 KiB LDS (4 WG/CU)", [&] { mix_tile420<<<nt420, 192, 40960>>>(din, dout, 12288, 8); });
  }
  return 0;
}
