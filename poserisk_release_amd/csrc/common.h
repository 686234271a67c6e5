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

// Capture guard (ABI 10).  pr_*_create, pr_hmr_set_streams and pr_*_destroy allocate, copy and synchronise; reached
// while a stream of the calling thread is being captured into a hipGraph they invalidate the capture, and the process
// aborts at its next synchronisation (round 5, tests: a released SMPL handle re-created inside a capture).  Those entry
// points have no stream argument, so the caller DECLARES the stream it enqueues on (pr_declare_stream, thread-local; the
// Python binding declares torch's current stream); they ask HIP whether it is capturing and return PR_ERR_INVALID,
// having touched nothing.  Entry points that allocate AND take a stream (the stand-alone test entries) check that one.
void declared_stream(bool* declared, hipStream_t* s);
inline int refuse_if_capturing(hipStream_t s, const char* what) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &st) != hipSuccess) {
    (void)hipGetLastError();   // an invalid or foreign handle says nothing about a capture: not this guard's business
    return PR_OK;
  }
  PR_REQUIRE(st == hipStreamCaptureStatusNone,
             "%s: the caller's stream is being captured into a hipGraph; this call allocates / copies / synchronises and "
             "would invalidate the capture -- create, resize and destroy handles outside the capture", what);
  return PR_OK;
}
inline int refuse_under_declared_capture(const char* what) {
  bool declared = false;
  hipStream_t s = nullptr;
  declared_stream(&declared, &s);
  return declared ? refuse_if_capturing(s, what) : PR_OK;
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

// One bf16 MFMA step of the kernels on v_mfma_f32_32x32x16_bf16 (weights or rows as A, pixels as B).  `sel` is unused: round 6
// issued the step as two 16x16x32 MFMAs on alternating accumulator quarters in an experiment build to estimate the other
// shape's clock -- and that estimate (+5 % end to end) was WRONG, because the wrong results changed the data the chip
// switches on; the real conversion is below and its measurements are in profiles/r06_experiments.txt 1.
using pr_f32x16 = __attribute__((ext_vector_type(16))) float;
using pr_bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
__device__ __forceinline__ pr_f32x16 mfma_bf16_step(pr_bf16x8 w, pr_bf16x8 x, pr_f32x16 c, [[maybe_unused]] int sel) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, x, c, 0, 0, 0);
#else
  return c;
#endif
}

// ---- a 32 x 32 output tile on v_mfma_f32_16x16x32_bf16 (round 6) ----------------------------------------------------------
// The bf16 kernels were written for v_mfma_f32_32x32x16_bf16: a wave's unit of output is a 32 x 32 tile whose 16 accumulator
// registers hold, on lane (i = lane & 31, h = lane >> 5), column i of rows (e & 3) + 8 (e >> 2) + 4 h.  The chip can hold a
// higher clock on the 16x16x32 shape at equal cycles per FLOP (MI355X_MICROARCH.md, DVFS give-back 7).  Measured on these
// kernels (profiles/r06_experiments.txt 1): the TILE kernel (conv_dma_bf16) is level to 6 % faster per layer on it and uses it;
// the whole-block kernels were converted the same way, bit for bit, ran 1 - 5 % faster stand-alone (cycles + 4 %, clock + 10 %)
// and 4 - 6 % SLOWER inside the encoder (an MFMA of this shape holds its SIMD's issue port for half its cycles, and in the
// pipeline the clock does not rise), and stay on 32x32x16.  A K loop on this shape issues FOUR 16x16x32 MFMAs per 32 k on the
// tile's four 16 x 16 quadrants -- the same operand bytes from LDS, the same accumulator registers -- and converts ONCE, in
// front of the epilogue, with 8 v_permlane32_swap_b32: afterwards register e of lane l is column i of row
// (e & 3) + 8 (e >> 2) + 4 h for
//     i = 16 (l >> 5) + (l & 15),   h = (l >> 4) & 1        (acc_col / acc_half below)
// i.e. exactly the 32x32x16 registers on relabelled lanes, so every epilogue keeps its arithmetic and only takes (i, h) from
// these two functions.  Measured (scripts/micro/t_mfma16_swap.hip): no element misplaced, and on random bf16 data the sums
// have the 32x32x16 form's BITS (both shapes add the 32 products of a step in the same order).
// Operands of one step: lane (j = l & 15, g = l >> 4) supplies A[row 16 rt + j][k = 8 g .. 8 g + 7] and
// B[k = 8 g .. 8 g + 7][col 16 ct + j] of the step's 32 k (frag_row / frag_kblock below).
using pr_f32x4 = __attribute__((ext_vector_type(4))) float;
struct Acc32 {
  pr_f32x4 t[2][2];   // [rt: rows 16 rt ..][ct: columns 16 ct ..]
};
__device__ __forceinline__ int frag_row(int lane) { return lane & 15; }      // + 16 rt (A) / + 16 ct (B)
__device__ __forceinline__ int frag_kblock(int lane) { return lane >> 4; }   // 8 k each
__device__ __forceinline__ int acc_col(int lane) { return 16 * (lane >> 5) + (lane & 15); }
__device__ __forceinline__ int acc_half(int lane) { return (lane >> 4) & 1; }
__device__ __forceinline__ void acc32_zero(Acc32& c) {
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) c.t[rt][ct] = pr_f32x4{0.f, 0.f, 0.f, 0.f};
}
__device__ __forceinline__ void mfma_bf16_32x32x32(Acc32& c, pr_bf16x8 a0, pr_bf16x8 a1, pr_bf16x8 b0, pr_bf16x8 b1) {
#if defined(__HIP_DEVICE_COMPILE__)
  c.t[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, c.t[0][0], 0, 0, 0);
  c.t[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b1, c.t[0][1], 0, 0, 0);
  c.t[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b0, c.t[1][0], 0, 0, 0);
  c.t[1][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, c.t[1][1], 0, 0, 0);
#endif
}
// The tile's accumulators in the 32x32x16 register layout (lane labels acc_col / acc_half).  ONE asm statement: every
// accumulator is an operand, so all the tile's MFMAs have issued in front of it, and the pad in front is the matrix pipe's
// write -> VALU read wait (18 wait states cover a 16-pass MFMA; hipcc pads nothing for asm operands).  The builtin
// __builtin_amdgcn_permlane32_swap is not used: hipcc 7.2 drops its second result and merges calls in exactly this pattern.
__device__ __forceinline__ pr_f32x16 acc32_regs(const Acc32& c) {
  pr_f32x16 o = {};
#if defined(__HIP_DEVICE_COMPILE__)
  float x[8], y[8];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      x[4 * rt + r] = c.t[rt][0][r];
      y[4 * rt + r] = c.t[rt][1][r];
    }
  asm volatile("s_nop 15\n\ts_nop 3\n\t"
               "v_permlane32_swap_b32 %0, %8\n\tv_permlane32_swap_b32 %1, %9\n\tv_permlane32_swap_b32 %2, %10\n\t"
               "v_permlane32_swap_b32 %3, %11\n\tv_permlane32_swap_b32 %4, %12\n\tv_permlane32_swap_b32 %5, %13\n\t"
               "v_permlane32_swap_b32 %6, %14\n\tv_permlane32_swap_b32 %7, %15"
               : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]),
                 "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), "+v"(y[7]));
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      o[8 * rt + r] = x[4 * rt + r];
      o[8 * rt + 4 + r] = y[4 * rt + r];
    }
#endif
  return o;
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
  *cus = n;
  return PR_OK;
}

}  // namespace pr
