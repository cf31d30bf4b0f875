// tfa_host_util.h — host-side helpers shared by the forward and backward launchers.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>

namespace tfa {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per DEVICE: one bit per device ordinal in the launcher's mask, so a
// process that drives several GPUs opts every one of them in (devices >= 64 are simply set on every launch).
static inline hipError_t set_dyn_lds_once(std::atomic<unsigned long long>& mask, const void* kern, int lds) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const bool cacheable = dev >= 0 && dev < 64;
  const unsigned long long bit = cacheable ? (1ull << dev) : 0ull;
  if (cacheable && (mask.load(std::memory_order_acquire) & bit)) return hipSuccess;
  e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) return e;
  if (cacheable) mask.fetch_or(bit, std::memory_order_release);
  return hipSuccess;
}

// Multiprocessor count of the CURRENT device (cached per device ordinal; 256 = MI355X when no device is visible, which
// is the dry-run planning case on a CPU box).
static inline int num_cus_current_device() {
  static std::atomic<int> cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  const bool cacheable = dev >= 0 && dev < 64;
  if (cacheable) { const int c = cache[dev].load(std::memory_order_relaxed); if (c > 0) return c; }
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 256;
  if (cacheable) cache[dev].store(n, std::memory_order_relaxed);
  return n;
}

}  // namespace tfa
