// tools/probe_store.hip -- store-side ceiling probe (not product code): how fast can 403 MB be
// written with (a) dwordx4 linear, (b) dwordx3 linear, (c) the kernel's tile pattern (8 rows x
// 1536 B at a 12288-B pitch, dwordx3), (d) tile pattern with rows of 3072 B.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u3 __attribute__((ext_vector_type(3)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

__global__ void st_x4(uint8_t *out, long n16) {  // each block of 256 threads writes 64 KiB
  long base = (long)blockIdx.x * 4096;
  for (int i = 0; i < 16; i++) {
    long idx = base + i * 256 + threadIdx.x;
    if (idx < n16) *(u4 *)(out + idx * 16) = u4{1u, 2u, 3u, (uint32_t)idx};
  }
}
__global__ void st_x3(uint8_t *out, long n12) {
  long base = (long)blockIdx.x * 4096;
  for (int i = 0; i < 16; i++) {
    long idx = base + i * 256 + threadIdx.x;
    if (idx < n12) *(u3 *)(out + idx * 12) = u3{1u, 2u, (uint32_t)idx};
  }
}
// tile pattern: block = 192 threads = 3 waves; tile = ROWS rows x SEGB bytes at pitch; wave-iteration = 768 B
template <int ROWS, int SEGB>
__global__ void st_tile(uint8_t *out, long pitch, int tiles_per_row, long img_stride, int tiles_per_img) {
  int t = blockIdx.x;
  int img = t / tiles_per_img, r = t % tiles_per_img;
  int ty = r / tiles_per_row, tx = r % tiles_per_row;
  uint8_t *base = out + img * img_stride + (long)ty * ROWS * pitch + (long)tx * SEGB;
  int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int ITERS = ROWS * SEGB / 768;
  for (int it = wave; it < ITERS; it += 3) {
    int row = it / (SEGB / 768), seg = it % (SEGB / 768);
    *(u3 *)(base + (long)row * pitch + seg * 768 + lane * 12) = u3{1u, (uint32_t)it, (uint32_t)t};
  }
}

// same tile pattern through raw buffer stores with an explicit cache policy (aux: 1 = sc0, 2 = nt, 16 = sc1)
template <int AUX>
__global__ void st_tile_aux(uint8_t *out, long pitch, int tiles_per_row, long img_stride, int tiles_per_img) {
  constexpr int ROWS = 8, SEGB = 1536;
  int t = blockIdx.x;
  int img = t / tiles_per_img, r = t % tiles_per_img;
  int ty = r / tiles_per_row, tx = r % tiles_per_row;
  uint8_t *base = out + img * img_stride + (long)ty * ROWS * pitch + (long)tx * SEGB;
  int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  constexpr int ITERS = ROWS * SEGB / 768;
  for (int it = wave; it < ITERS; it += 3) {
    int row = it / (SEGB / 768), seg = it % (SEGB / 768);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base + (long)row * pitch + seg * 768, 0, 768, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b96(u3{1u, (uint32_t)it, (uint32_t)t}, rsrc, lane * 12, 0, AUX);
  }
}

int main() {
  const long W = 4096, H = 4096, NIMG = 8, pitch = W * 3, bytes = NIMG * H * pitch;
  uint8_t *d;
  hipMalloc(&d, bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto time = [&](const char *name, auto launch) {
    float best = 1e9, sum = 0;
    int n = 0;
    for (int rep = 0; rep < 300; rep++) {
      hipEventRecord(e0);
      launch();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep >= 200) { sum += ms; n++; if (ms < best) best = ms; }
    }
    printf("%-34s mean %.1f us  min %.1f us  %.2f TB/s\n", name, sum / n * 1e3, best * 1e3, bytes / (sum / n * 1e-3) / 1e12);
  };
  long n16 = bytes / 16, n12 = bytes / 12;
  time("dwordx4 linear", [&] { st_x4<<<(n16 + 4095) / 4096, 256>>>(d, n16); });
  time("dwordx3 linear", [&] { st_x3<<<(n12 + 4095) / 4096, 256>>>(d, n12); });
  time("tile 8 rows x 1536 B, dwordx3", [&] { st_tile<8, 1536><<<NIMG * 512 * 8, 192>>>(d, pitch, 8, H * pitch, 512 * 8); });
  time("tile 8 rows x 3072 B, dwordx3", [&] { st_tile<8, 3072><<<NIMG * 512 * 4, 192>>>(d, pitch, 4, H * pitch, 512 * 4); });
  time("tile 16 rows x 1536 B, dwordx3", [&] { st_tile<16, 1536><<<NIMG * 256 * 8, 192>>>(d, pitch, 8, H * pitch, 256 * 8); });
  time("tile 8 rows x 12288 B (full rows)", [&] { st_tile<8, 12288><<<NIMG * 512, 192>>>(d, pitch, 1, H * pitch, 512); });
  time("tile 8x1536 buffer aux=0", [&] { st_tile_aux<0><<<NIMG * 512 * 8, 192>>>(d, pitch, 8, H * pitch, 512 * 8); });
  time("tile 8x1536 buffer aux=1 (sc0)", [&] { st_tile_aux<1><<<NIMG * 512 * 8, 192>>>(d, pitch, 8, H * pitch, 512 * 8); });
  time("tile 8x1536 buffer aux=2 (nt)", [&] { st_tile_aux<2><<<NIMG * 512 * 8, 192>>>(d, pitch, 8, H * pitch, 512 * 8); });
  time("tile 8x1536 buffer aux=3 (sc0 nt)", [&] { st_tile_aux<3><<<NIMG * 512 * 8, 192>>>(d, pitch, 8, H * pitch, 512 * 8); });
  time("tile 8x1536 buffer aux=16 (sc1)", [&] { st_tile_aux<16><<<NIMG * 512 * 8, 192>>>(d, pitch, 8, H * pitch, 512 * 8); });
  time("tile 8x1536 buffer aux=17 (sc0 sc1)", [&] { st_tile_aux<17><<<NIMG * 512 * 8, 192>>>(d, pitch, 8, H * pitch, 512 * 8); });
  time("tile 8x1536 buffer aux=18 (sc1 nt)", [&] { st_tile_aux<18><<<NIMG * 512 * 8, 192>>>(d, pitch, 8, H * pitch, 512 * 8); });
  time("tile 8x1536 buffer aux=19 (all)", [&] { st_tile_aux<19><<<NIMG * 512 * 8, 192>>>(d, pitch, 8, H * pitch, 512 * 8); });
  time("hipMemsetAsync", [&] { hipMemsetAsync(d, 7, bytes, 0); });
  return 0;
}
