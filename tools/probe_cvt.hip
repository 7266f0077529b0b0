// tools/probe_cvt.hip -- hardware probe (not product code): does v_cvt_pk_u8_f32 alone equal
// the reference's "truncate toward zero, then clamp to 0..255" (jpeg.cpp:521-535)?  Also times
// packed-f32 VALU ops against scalar ones (is v_pk_add_f32/v_pk_mul_f32 a lever on gfx950?).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ void cvt_probe(const float *in, uint32_t *direct, uint32_t *ref, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float x = in[i];
  direct[i] = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_truncf(x), 0, 0);  // shipped form: trunc, then the converter's own saturation
  float t = __builtin_amdgcn_fmed3f(__builtin_truncf(x), 0.0f, 255.0f);
  ref[i] = __builtin_amdgcn_cvt_pk_u8_f32(t, 0, 0);
}

// v_cvt_pk_u8_f32 with the wave's f32 rounding mode switched to round-toward-zero around it
__global__ void cvt_probe_rtz(const float *in, uint32_t *direct, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float x = in[i];
  uint32_t r;
  asm volatile(
      "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
      "v_cvt_pk_u8_f32 %0, %1, 0, 0\n\t"
      "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
      : "=v"(r)
      : "v"(x));
  direct[i] = r;
}

typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void valu_rate(float *out, int iters) {
  float a0 = threadIdx.x * 0.001f + 1.0f, a1 = a0 + 0.5f, a2 = a0 + 0.25f, a3 = a0 + 0.125f;
  float b0 = 1.0001f, b1 = 0.9999f;
  v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a1, a2}, p3 = {a3, a0};
  v2f q0 = {b0, b1};
  for (int i = 0; i < iters; i++) {
    if (MODE == 0) {  // 8 scalar ops (mul/add alternating), 8 independent chains / 2
#pragma unroll
      for (int u = 0; u < 8; u++) {
        a0 = a0 * b0; a1 = a1 + b1; a2 = a2 * b0; a3 = a3 + b1;
        a0 = a0 + b1; a1 = a1 * b0; a2 = a2 + b1; a3 = a3 * b0;
      }
    } else {  // the same arithmetic as 4 packed ops per 8 scalar
#pragma unroll
      for (int u = 0; u < 8; u++) {
        p0 = p0 * q0; p1 = p1 + q0; p2 = p2 * q0; p3 = p3 + q0;
        p0 = p0 + q0; p1 = p1 * q0; p2 = p2 + q0; p3 = p3 * q0;
      }
    }
  }
  if (MODE == 0) out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
  else out[blockIdx.x * blockDim.x + threadIdx.x] = p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

int main() {
  std::vector<float> h;
  for (int i = -70000; i <= 70000; i++) h.push_back(i / 64.0f);            // -1093.75 .. 1093.75 step 1/64
  const float special[] = {-0.0f, 0.0f, 254.99998f, 255.0f, 255.00002f, 255.5f, 255.99998f, 256.0f, 1e9f, -1e9f,
                           3.4e38f, -3.4e38f, 0.99999994f, -0.99999994f, 1e-40f, -1e-40f, 2147483648.0f, 4294967296.0f};
  for (float s : special) h.push_back(s);
  int n = (int)h.size();
  float *d_in; uint32_t *d_a, *d_b;
  hipMalloc(&d_in, n * 4); hipMalloc(&d_a, n * 4); hipMalloc(&d_b, n * 4);
  hipMemcpy(d_in, h.data(), n * 4, hipMemcpyHostToDevice);
  cvt_probe<<<(n + 255) / 256, 256>>>(d_in, d_a, d_b, n);
  std::vector<uint32_t> a(n), b(n);
  hipMemcpy(a.data(), d_a, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), d_b, n * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; i++)
    if (a[i] != b[i]) { if (bad < 12) printf("  x=%.9g direct=%u trunc+clamp=%u\n", h[i], a[i], b[i]); bad++; }
  printf("cvt_pk_u8_f32(trunc(x)) probe: %d inputs, %d mismatches vs trunc+med3+cvt path\n", n, bad);

  cvt_probe_rtz<<<(n + 255) / 256, 256>>>(d_in, d_a, n);
  hipMemcpy(a.data(), d_a, n * 4, hipMemcpyDeviceToHost);
  bad = 0;
  for (int i = 0; i < n; i++)
    if (a[i] != b[i]) { if (bad < 12) printf("  x=%.9g rtz-mode=%u trunc+clamp=%u\n", h[i], a[i], b[i]); bad++; }
  printf("cvt_pk_u8_f32 under MODE.round=RTZ: %d inputs, %d mismatches vs trunc+med3+cvt path\n", n, bad);

  float *d_out; hipMalloc(&d_out, 1024 * 256 * 4 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000, grid = 256 * 8, block = 256;
  for (int mode = 0; mode < 2; mode++) {
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(e0);
      if (mode == 0) valu_rate<0><<<grid, block>>>(d_out, iters); else valu_rate<1><<<grid, block>>>(d_out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double flops = (double)grid * block * iters * 64.0;  // 64 scalar-equivalent ops per iteration
      if (rep == 2) printf("valu_rate mode=%s: %.3f ms, %.1f Gop/s (scalar-equivalent f32 mul/add)\n", mode ? "packed" : "scalar", ms, flops / ms / 1e6);
    }
  }
  return 0;
}
