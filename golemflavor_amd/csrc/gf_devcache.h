// gf_devcache.h -- large device allocations are CACHED by the library instead of being handed back to the driver.
//
// Why (round 4, tools/vram_realloc_probe*.py, profiles/r04/host_register.txt): when device memory is freed the driver wipes it, and it
// does so with the DMA engine the read-backs use.  For ~0.6 s after a hipFree of 9.4 GB every device-to-host copy of the process runs at
// 30 GB/s instead of 57 -- whatever buffer it reads, on whatever stream.  A scan frees its chain and row buffers when it ends, i.e.
// right before the NEXT scan's read-back: the second and later scans of a process crossed PCIe at half speed (and C5's chain, read back
// while it is sampled, did so whenever C4's scan had just ended).  So buffers of GF_DEVCACHE_MIN bytes or more go to a per-device free
// list when they are "freed" and are reused for the next request they fit (same size up to +25 %); the list holds at most
// GF_DEVICE_CACHE_GB (default 96) and is emptied by gf_device_trim().  hipFree waits for the device before it releases memory; the
// cached free does the same, so no caller can tell the difference -- except that reused memory holds its previous content where
// freshly mapped memory holds zeros (nothing in the library reads a large buffer before writing it).
//
// Include AFTER <hip/hip_runtime.h>: the two macros at the end route this translation unit's hipMalloc / hipFree calls here.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

extern "C" {
hipError_t gf_cached_malloc(void** ptr, size_t bytes);
hipError_t gf_cached_free(void* ptr);
// release what the cache holds on `device` (< 0: every device); returns the bytes handed back to the driver
size_t gf_devcache_trim(int device);
// diagnostics: bytes held idle / handed out through the cache on `device`
void gf_devcache_stats(int device, size_t* idle_bytes, size_t* live_bytes, unsigned long long* reuses);
}

#ifndef GF_DEVCACHE_IMPL
#define hipMalloc(p, n) gf_cached_malloc((void**)(p), (size_t)(n))
#define hipFree(p) gf_cached_free((void*)(p))
#endif
