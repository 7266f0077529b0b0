// jb_hostmem.h -- caller-owned pixel buffers (internal; released with jb_free = free).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <sys/mman.h>

// Large images are first touched by the copy out of pinned staging (or by the runtime's own staged
// device-to-host copy): with 4 KiB pages that is 49,000 page faults for one 8192x8192 image, so
// buffers of 4 MiB and more are 2 MiB-aligned and ask for transparent huge pages.
static inline uint8_t *jb_alloc_pixels_(size_t bytes) {
  const size_t kHuge = (size_t)2 << 20;
  if (bytes < 2 * kHuge) return (uint8_t *)malloc(bytes);
  void *p = nullptr;
  if (posix_memalign(&p, kHuge, (bytes + kHuge - 1) & ~(kHuge - 1)) != 0) return nullptr;
  madvise(p, (bytes + kHuge - 1) & ~(kHuge - 1), MADV_HUGEPAGE);  // advisory: ignoring a failure is fine
  return (uint8_t *)p;
}
