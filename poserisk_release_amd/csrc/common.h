// Shared host-side helpers for libposerisk_hip.so (gfx950 only; no CUDA/HIP dual paths).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>

#include "host_common.h"

namespace pr {

#define PR_HIP(call)                                                                      \
  do {                                                                                    \
    hipError_t e__ = (call);                                                              \
    if (e__ != hipSuccess) {                                                              \
      pr::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__,     \
                    __LINE__);                                                            \
      return PR_ERR_HIP;                                                                  \
    }                                                                                     \
  } while (0)

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("launch of %s failed: %s", what, hipGetErrorString(e));
    return PR_ERR_HIP;
  }
  return PR_OK;
}

// Raises a kernel's dynamic-LDS limit once per DEVICE (the attribute is per device; a handle may be created on any
// device of the process).  `done` is one word per kernel instantiation, one bit per device ordinal.
inline int ensure_dynamic_lds(const void* kern, size_t bytes, std::atomic<uint64_t>& done) {
  int dev = 0;
  PR_HIP(hipGetDevice(&dev));
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return PR_OK;
  PR_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  done.fetch_or(bit, std::memory_order_release);
  return PR_OK;
}

// RAII device selection for create/destroy paths.
struct DeviceGuard {
  int prev = -1;
  bool ok = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) == hipSuccess && hipSetDevice(dev) == hipSuccess) ok = true;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

// Bijective block-index remap for the 8 XCDs: the hardware hands consecutive workgroup ids to the XCDs round-robin;
// this gives XCD x the x-th contiguous run of logical blocks, so neighbouring blocks (which share input rows or
// operand tiles) share an L2.
__device__ __forceinline__ long xcd_contiguous_block(long bid, long nb) {
  const long xcd = bid & 7, q8 = nb >> 3, rr = nb & 7;
  return (xcd < rr ? xcd * (q8 + 1) : rr * (q8 + 1) + (xcd - rr) * q8) + (bid >> 3);
}

// buffer_store_dwordx4 with a SCALAR-REGISTER soffset.  MEASURED on MI355X (round 3, scripts/micro/t_store_hazard.hip and the
// expand_res_bf16 stress test): when the instruction right behind such a store is a VALU write of the store's first data
// register, the NEW value can reach memory (the stored dword came out as the next tile's half-finished arithmetic, on the
// waves that lose the issue arbitration, a few hundred times per 25 M elements).  The ISA's "VMEM store of more than 64
// bits followed by a write of its data VGPRs" hazard; hipcc pads it only when soffset is NOT a register (LLVM
// GCNHazardRecognizer::createsVALUHazard), so a register soffset -- which the kernels use to keep wave-uniform terms out
// of the range-checked vector offset -- needs its own wait states.  The asm below keeps the data registers live and
// unwritten for four more issue slots; it must stay directly behind the store.
using pr_u32x4 = __attribute__((ext_vector_type(4))) unsigned;
template <typename Rsrc>
__device__ __forceinline__ void buffer_store_b128_sreg(pr_u32x4 v, Rsrc rsrc, unsigned voffset, int soffset) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, voffset, soffset, 0);
  asm volatile("s_nop 3" ::"v"(v) : "memory");
#endif
}

// Compute units of the current device, asked once per device (the persistent kernels size their grids by it at every launch).
inline int current_device_cus(int* cus) {
  static std::atomic<int> cache[64] = {};
  int dev = 0;
  PR_HIP(hipGetDevice(&dev));
  int n = cache[dev & 63].load(std::memory_order_relaxed);
  if (!n) {
    PR_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    if (n <= 0) n = 256;
    cache[dev & 63].store(n, std::memory_order_relaxed);
  }
  // A/B timing only: POSERISK_GRID_CUS=<n> makes every persistent kernel size its grid for n CUs (round 5: do two batches in
  // flight run better side by side on half the chip each than one behind the other on all of it?  profiles/r05_experiments.txt 7).
  // No result depends on a persistent kernel's grid.
  static const int forced = [] { const char* e = getenv("POSERISK_GRID_CUS"); return e ? atoi(e) : 0; }();
  *cus = forced > 0 && forced <= n ? forced : n;
  return PR_OK;
}

}  // namespace pr
