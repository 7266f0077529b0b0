// tools/huff_emu: a SIMT shim that lets jb_huff.hip -- the kernels' own text -- run on the HOST.
// Test infrastructure (tests/test_huff_emu.py, tools/huff_emu/huff_emu.cpp): one OS thread per lane, one
// workgroup at a time, barriers and wave exchanges through std::barrier; LDS is one global buffer.  What it
// buys: the pass structure, the bookkeeping at the checkpoints, the scans and the verification of the
// device entropy decoder can be checked bit for bit (and under AddressSanitizer / ThreadSanitizer) on a
// machine without a GPU.  Never part of the product: libjpegblk.so is built by hipcc from the same file.
#pragma once
#include <stdint.h>
#include <string.h>

#include <atomic>
#include <barrier>
#include <functional>
#include <memory>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__
#define __constant__ static const

struct uint4 {
  uint32_t x, y, z, w;
};
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
struct dim3 {
  unsigned x, y, z;
  dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
typedef void *hipStream_t;
typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
constexpr hipError_t hipErrorInvalidValue = 1;
static inline hipError_t hipGetLastError() { return hipSuccess; }

namespace emu {
struct Idx {
  unsigned x = 0, y = 0, z = 0;
};
struct Wave {
  std::barrier<> bar{64};
  uint32_t slot[64];
  explicit Wave(int n) : bar(n) {}
};
struct Group {
  std::barrier<> bar;
  std::vector<std::unique_ptr<Wave>> waves;
  std::atomic<int> votes{0};
  explicit Group(int n) : bar(n) {
    for (int w = 0; w < (n + 63) / 64; w++) waves.emplace_back(new Wave(n - 64 * w < 64 ? n - 64 * w : 64));
  }
};
extern thread_local Idx tl_thread, tl_block, tl_grid;
extern thread_local Group *tl_group;
void launch(dim3 grid, dim3 block, const std::function<void()> &body);
}  // namespace emu
#define threadIdx (emu::tl_thread)
#define blockIdx (emu::tl_block)
#define gridDim (emu::tl_grid)

extern "C" uint8_t lds[];  // the workgroup's LDS (`extern __shared__ uint8_t lds[]` in the kernels)

static inline void __syncthreads() { emu::tl_group->bar.arrive_and_wait(); }
static inline int __syncthreads_or(int pred) {
  emu::Group *g = emu::tl_group;
  if (pred) g->votes.fetch_add(1);
  g->bar.arrive_and_wait();
  const int any = g->votes.load() != 0;
  g->bar.arrive_and_wait();
  if (emu::tl_thread.x == 0) g->votes.store(0);
  g->bar.arrive_and_wait();
  return any;
}
static inline uint32_t __shfl_up(uint32_t v, int off) {
  emu::Wave *w = emu::tl_group->waves[emu::tl_thread.x >> 6].get();
  const unsigned lane = emu::tl_thread.x & 63u;
  w->slot[lane] = v;
  w->bar.arrive_and_wait();
  const uint32_t r = lane >= (unsigned)off ? w->slot[lane - (unsigned)off] : v;
  w->bar.arrive_and_wait();
  return r;
}
static inline uint64_t __builtin_amdgcn_ballot_w64(bool pred) {
  emu::Wave *w = emu::tl_group->waves[emu::tl_thread.x >> 6].get();
  const unsigned lane = emu::tl_thread.x & 63u;
  w->slot[lane] = pred ? 1u : 0u;
  w->bar.arrive_and_wait();
  uint64_t m = 0;
  for (unsigned i = 0; i < 64; i++)
    if (w->slot[i]) m |= 1ull << i;
  w->bar.arrive_and_wait();
  return m;
}
static inline uint32_t atomicMax(uint32_t *p, uint32_t v) {
  uint32_t old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
  }
  return old;
}
static inline uint32_t atomicAdd(uint32_t *p, uint32_t v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline uint32_t atomicOr(uint32_t *p, uint32_t v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }

#define hipLaunchKernelGGL(kernel, grid, block, lds_bytes, stream, ...) emu::launch((grid), (block), [&] { kernel(__VA_ARGS__); })
