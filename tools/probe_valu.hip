// tools/probe_valu.hip -- per-instruction VALU issue cost on gfx950 (probe, not product code).
// Each kernel runs ITERS iterations of a block of 32 independent instructions of one kind;
// grid = 2048 blocks x 256 threads (8 waves/SIMD worth of work per CU, so issue-bound).
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

#define KERNEL(name, body)                                                         \
  __global__ void name(float *out, int iters) {                                    \
    float a = threadIdx.x * 0.37f + 1.0f, b = a + 3.0f, c = b + 5.0f, d = c + 7.0f; \
    float k;                                                                       \
    asm volatile("s_mov_b32 %0, 0x3e8e39da" : "=s"(k));                            \
    for (int i = 0; i < iters; i++) {                                              \
      asm volatile(REP8(body) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(k));      \
    }                                                                              \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;                    \
  }

KERNEL(k_mul_lit, "v_mul_f32_e32 %0, 0x3e8e39da, %0\n v_mul_f32_e32 %1, 0x3e8e39da, %1\n v_mul_f32_e32 %2, 0x3e8e39da, %2\n v_mul_f32_e32 %3, 0x3e8e39da, %3\n")
KERNEL(k_mul_sgpr, "v_mul_f32_e32 %0, %4, %0\n v_mul_f32_e32 %1, %4, %1\n v_mul_f32_e32 %2, %4, %2\n v_mul_f32_e32 %3, %4, %3\n")
KERNEL(k_add, "v_add_f32_e32 %0, %1, %0\n v_add_f32_e32 %1, %2, %1\n v_add_f32_e32 %2, %3, %2\n v_add_f32_e32 %3, %0, %3\n")
KERNEL(k_trunc, "v_trunc_f32_e32 %0, %0\n v_trunc_f32_e32 %1, %1\n v_trunc_f32_e32 %2, %2\n v_trunc_f32_e32 %3, %3\n")
KERNEL(k_cvt_f32_i32, "v_cvt_f32_i32_e32 %0, %0\n v_cvt_f32_i32_e32 %1, %1\n v_cvt_f32_i32_e32 %2, %2\n v_cvt_f32_i32_e32 %3, %3\n")
KERNEL(k_mul24_sdwa, "v_mul_i32_i24_sdwa %0, %1, sext(%0) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n v_mul_i32_i24_sdwa %1, %2, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_mul_i32_i24_sdwa %2, %3, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n v_mul_i32_i24_sdwa %3, %0, sext(%3) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n")
KERNEL(k_mul24, "v_mul_i32_i24_e32 %0, %1, %0\n v_mul_i32_i24_e32 %1, %2, %1\n v_mul_i32_i24_e32 %2, %3, %2\n v_mul_i32_i24_e32 %3, %0, %3\n")
KERNEL(k_cvt_pk_u8, "v_cvt_pk_u8_f32 %0, %1, 0, %0\n v_cvt_pk_u8_f32 %1, %2, 1, %1\n v_cvt_pk_u8_f32 %2, %3, 2, %2\n v_cvt_pk_u8_f32 %3, %0, 3, %3\n")
KERNEL(k_med3, "v_med3_f32 %0, %0, 0, %1\n v_med3_f32 %1, %1, 0, %2\n v_med3_f32 %2, %2, 0, %3\n v_med3_f32 %3, %3, 0, %0\n")
KERNEL(k_mul_lo_u32, "v_mul_lo_u32 %0, %1, %0\n v_mul_lo_u32 %1, %2, %1\n v_mul_lo_u32 %2, %3, %2\n v_mul_lo_u32 %3, %0, %3\n")
KERNEL(k_bfe, "v_bfe_i32 %0, %0, 0, 16\n v_bfe_i32 %1, %1, 0, 16\n v_bfe_i32 %2, %2, 0, 16\n v_bfe_i32 %3, %3, 0, 16\n")
KERNEL(k_fma, "v_fma_f32 %0, %1, %2, %0\n v_fma_f32 %1, %2, %3, %1\n v_fma_f32 %2, %3, %0, %2\n v_fma_f32 %3, %0, %1, %3\n")

template <class K>
void run(const char *name, K kern, float *d_out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 4000, grid = 2048, block = 256;
  float best = 1e9f;
  for (int rep = 0; rep < 4; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, d_out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  const double wave_instrs = (double)grid * (block / 64) * iters * 32.0;
  // cycles per wave-instruction per SIMD at 1024 SIMDs, assuming 2.0 GHz
  printf("%-14s %8.3f ms  %7.1f G wave-instr/s  -> %.2f ns/instr/SIMD\n", name, best, wave_instrs / best / 1e6,
         best * 1e6 / (wave_instrs / 1024.0));
}

int main() {
  float *d_out;
  hipMalloc(&d_out, 2048 * 256 * 4);
  run("mul_literal", k_mul_lit, d_out);
  run("mul_sgpr", k_mul_sgpr, d_out);
  run("add", k_add, d_out);
  run("fma", k_fma, d_out);
  run("trunc", k_trunc, d_out);
  run("cvt_f32_i32", k_cvt_f32_i32, d_out);
  run("mul24_sdwa", k_mul24_sdwa, d_out);
  run("mul24", k_mul24, d_out);
  run("bfe_i32", k_bfe, d_out);
  run("cvt_pk_u8", k_cvt_pk_u8, d_out);
  run("med3", k_med3, d_out);
  run("mul_lo_u32", k_mul_lo_u32, d_out);
  return 0;
}
