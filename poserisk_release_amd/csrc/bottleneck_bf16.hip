// A whole ResNet-50 Bottleneck of layer1 (64 planes at 56x56) in ONE persistent bf16 kernel:
//   y = relu( bn3(conv3( relu(bn2(conv2_3x3( relu(bn1(conv1(x))) ))) )) + x )        x, y: [M = B*H*W][256] bf16 NHWC
// (SPIN models/hmr.py Bottleneck.forward, restated in oracle/hmr_ref.py; call site lib/core/base.py:220).
//
// Why one kernel: as three launches the block moves 1.44 GB at B=256 (x read twice, the two 64-channel maps written
// and read back) and every 64-row workgroup is a chain of ~17 serial waits; fused it moves x in and y out (822 MB) and
// the next pixels' loads are in flight while the current ones compute.
//
// Structure (one workgroup of 4 waves per CU, one wave per SIMD, up to 512 VGPRs each):
//   * A workgroup walks a contiguous run of 64-pixel blocks of the flattened [B*H*W] pixel index.  conv1 of block
//     t+2 is computed while block t is finished, so the 3x3 conv2 of block t finds t1 = relu(conv1(x)) of blocks
//     t-1, t, t+1 (its +-(W+1) pixel neighbourhood, W <= 63) in a three-block LDS ring: every pixel's conv1 is computed once
//     per workgroup (two extra blocks per run), no halo tiles.  Zero padding = per-lane tap masks (a masked lane reads a zero row).
//   * The MFMAs run "transposed": the weights are the A operand (rows = output channels), the pixels the B operand, so an
//     accumulator lane IS a pixel and its 16 registers are 16 consecutive channels (the packed weight rows are permuted
//     for that on the host).  t1 / t2 go to LDS and y to HBM straight from the registers as 16-byte pieces: no LDS
//     transpose, no epilogue barrier.
//   * W2 (72 KB) stays in LDS for the whole kernel; W1 and W3 (32 KB each) stay in REGISTERS as MFMA A fragments
//     (each wave holds the rows of its own output tiles: 64 + 64 VGPRs); biases in LDS.
//   * x streams through a six-slice LDS ring by LDS-DMA (a slice = 64 pixels x 64 channels = 8 KB), issued 1.5 blocks
//     ahead; the block's own rows are picked out of the ring into registers as the residual when conv1 consumes them
//     (x is read from HBM exactly once).
//   * Same k order and the same 16-wide MFMA groups as conv_dma_bf16 / conv3x3_conv1x1_bf16, t1 and t2 rounded to bf16
//     exactly where the separate launches store them.
#include <algorithm>
#include <cstdlib>

#include "conv_igemm.h"

namespace pr {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
typedef __attribute__((address_space(3))) void lds_void;

[[maybe_unused]] constexpr unsigned kOOB = 0x80000000u;

constexpr int kRing = 192;                         // t1 ring: three 64-pixel blocks
constexpr int kXSlots = 6;                         // x ring: six 8 KB slices
constexpr int kOffW2 = 0;                          // 9 taps x [64 rows][64 k] bf16
constexpr int kOffT1 = 9 * 8192;                   // [192 pixels][64 channels] bf16
constexpr int kOffT2 = kOffT1 + kRing * 128;       // [64 pixels][64 channels] bf16
constexpr int kOffX = kOffT2 + 8192;               // kXSlots x [64 pixels][64 channels] bf16
constexpr int kOffB1 = kOffX + kXSlots * 8192;     // 64 floats
constexpr int kOffB2 = kOffB1 + 256;               // 64 floats
constexpr int kOffB3 = kOffB2 + 256;               // 256 floats
constexpr int kOffZero = kOffB3 + 1024;            // 128 zero bytes
constexpr int kLdsBytes = kOffZero + 128;
static_assert(kLdsBytes <= 160 * 1024, "LDS budget");

struct BnArgs {
  const unsigned short* x;
  unsigned short* y;
  const unsigned short* w1;   // [64][256]  rows permuted (bottleneck_pack_rows_bf16)
  const unsigned short* w2;   // [64][576]  rows permuted, k = tap * 64 + c1
  const unsigned short* w3;   // [256][64]  rows permuted
  const float* b1;            // true channel order
  const float* b2;
  const float* b3;
  unsigned x_bytes;
  int H, W, HW, M, nblocks;
};

using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
__device__ inline unsigned pack_bf16x2(float lo, float hi) {   // one v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}

// DBG (timing builds only, -DPR_TIMING_HOOKS; results are wrong when set): 1 no y stores, 2 no wait for the x slices,
// 4 no MFMAs, 8 no x loads at all.
template <int DBG>
__global__ __launch_bounds__(256, 1) void bottleneck64_bf16(const BnArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  auto MFMA = [](const bf16x8& wa, const bf16x8& xb, const f32x16& c) -> f32x16 {
    if (DBG & 4) {                                   // keep the operands live so nothing upstream is removed
      asm volatile("" ::"v"(wa), "v"(xb));
      return c;
    }
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, xb, c, 0, 0, 0);
  };
  const int b0 = (int)((long)blockIdx.x * a.nblocks / gridDim.x);
  const int b1 = (int)((long)(blockIdx.x + 1) * a.nblocks / gridDim.x);
  if (b0 >= b1) return;                              // the whole workgroup leaves before any barrier
  const int nc = b1 - b0 + 2;                        // conv1 blocks b0-1 .. b1 (local 0 .. nc-1)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, h = lane >> 5;
  const int ct = wave >> 1, pt = wave & 1;           // output-channel tile (conv1/conv2) or tile group (conv3), pixel tile
  const int prow = 32 * pt + i;                      // the lane's pixel inside a 64-pixel block

  const auto xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  const auto ysrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)a.x_bytes, 0x00020000);
  // ---- one-time loads ---------------------------------------------------------------------------------------------
  {
    // W2 -> LDS, tap by tap: [64 rows][128 B], 16-byte chunks XOR-swizzled on the source side (conv_dma_bf16.hip)
    const auto w2src = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.w2), 0, 64 * 576 * 2, 0x00020000);
    const int q = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int grp = wave + 4 * j, r = 8 * grp + (lane >> 3);
      const unsigned voff = (unsigned)((r * 576 + q * 8) * 2);
      for (int tap = 0; tap < 9; ++tap)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w2src, (lds_void*)(smem + kOffW2 + tap * 8192 + grp * 1024), 16, voff,
                                                 tap * 128, 0, 0);
    }
    if (tid < 64) {
      *reinterpret_cast<float*>(smem + kOffB1 + tid * 4) = a.b1[tid];
      *reinterpret_cast<float*>(smem + kOffB2 + tid * 4) = a.b2[tid];
    }
    *reinterpret_cast<float*>(smem + kOffB3 + tid * 4) = a.b3[tid];
    if (tid < 32) *reinterpret_cast<unsigned*>(smem + kOffZero + tid * 4) = 0u;
  }
  // W1 / W3 rows of this wave's output tiles as MFMA A fragments (lane: row i of the tile, k = 16 ks + 8 h .. + 7)
  bf16x8 w1f[16], w3f[4][4];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks)
    w1f[ks] = *reinterpret_cast<const bf16x8*>(a.w1 + (32 * ct + i) * 256 + 16 * ks + 8 * h);
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
      w3f[n][kk] = *reinterpret_cast<const bf16x8*>(a.w3 + (32 * (4 * ct + n) + i) * 64 + 16 * kk + 8 * h);

  // ---- x ring --------------------------------------------------------------------------------------------------------
  // local slice L = 4 * (local conv1 block) + s, s = 64-channel slice of x; LDS slot L % 6.  A slice is eight 1 KB DMA
  // groups of 8 pixels; wave w issues groups w and w + 4.
  const int dq = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
  auto issue_slice = [&](int L) {
    const int blk = b0 - 1 + (L >> 2), s = L & 3;
    char* slot = smem + kOffX + (L % kXSlots) * 8192;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int grp = wave + 4 * j;
      const int m = blk * 64 + 8 * grp + (lane >> 3);
      const unsigned voff = (m >= 0 && m < a.M) ? (unsigned)(m * 512 + dq * 16) : kOOB;
      if (!(DBG & 8)) __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, (lds_void*)(slot + grp * 1024), 16, voff, s * 128, 0, 0);
    }
  };
  int gi = 0;                                        // next slice to issue
  const int nslices = 4 * nc;
  for (; gi < kXSlots && gi < nslices; ++gi) issue_slice(gi);

  // fragment read offsets inside a [64 rows][128 B] swizzled block: the lane's row, logical chunk 2 kk + h
  int pfoff[4], wfoff[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    pfoff[kk] = prow * 128 + (((2 * kk + h) ^ ((prow >> 1) & 7)) << 4);
    wfoff[kk] = (32 * ct + i) * 128 + (((2 * kk + h) ^ ((i >> 1) & 7)) << 4);
  }

  unsigned r0[32], r1[32];                           // residual rows (packed bf16 pairs) of the next two output blocks
#pragma unroll
  for (int e = 0; e < 32; ++e) r0[e] = r1[e] = 0u;

  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  for (int j = 0; j <= nc; ++j) {
    if (j >= 3) {
      // ================= phase A: conv2 of output block t (needs t1 of blocks t-1, t, t+1) ==========================
      const int t = b0 + j - 3;
      const int m = t * 64 + prow;
      const int img = m / a.HW, rem = m - img * a.HW;
      const int ho = rem / a.W, wo = rem - ho * a.W;
      const int rbase = m % kRing;
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int dh = tap / 3 - 1, dw = tap % 3 - 1;
        int R = rbase + dh * a.W + dw;
        R += R < 0 ? kRing : 0;
        R -= R >= kRing ? kRing : 0;
        const bool ok = (unsigned)(ho + dh) < (unsigned)a.H && (unsigned)(wo + dw) < (unsigned)a.W;
        const int rowoff = kOffT1 + R * 128, sw = (R >> 1) & 7;
        bf16x8 af[4], bf[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int off = ok ? rowoff + (((2 * kk + h) ^ sw) << 4) : kOffZero;
          bf[kk] = *reinterpret_cast<const bf16x8*>(smem + off);
          af[kk] = *reinterpret_cast<const bf16x8*>(smem + kOffW2 + tap * 8192 + wfoff[kk]);
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) acc = MFMA(af[kk], bf[kk], acc);
      }
      {  // t2 = bf16(relu(acc + b2)) -> LDS [pixel][channel]; the lane's 16 registers are channels 32 ct + 16 h + r
        const float* bp = reinterpret_cast<const float*>(smem + kOffB2) + 32 * ct + 16 * h;
        unsigned pk[8];
#pragma unroll
        for (int e = 0; e < 8; ++e)
          pk[e] = pack_bf16x2(fmaxf(acc[2 * e] + bp[2 * e], 0.f), fmaxf(acc[2 * e + 1] + bp[2 * e + 1], 0.f));
        const int c0 = 4 * ct + 2 * h, sw = (prow >> 1) & 7;
        *reinterpret_cast<u32x4*>(smem + kOffT2 + prow * 128 + ((c0 ^ sw) << 4)) = u32x4{pk[0], pk[1], pk[2], pk[3]};
        *reinterpret_cast<u32x4*>(smem + kOffT2 + prow * 128 + (((c0 + 1) ^ sw) << 4)) = u32x4{pk[4], pk[5], pk[6], pk[7]};
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");

      // ================= phase B: conv3 + residual + ReLU of block t, stored from the registers =====================
      bf16x8 tf[4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) tf[kk] = *reinterpret_cast<const bf16x8*>(smem + kOffT2 + pfoff[kk]);
      const unsigned yoff = m < a.M ? (unsigned)(m * 512 + 32 * h) : kOOB;   // + the lane half's 16 channels
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        f32x16 c3;
#pragma unroll
        for (int e = 0; e < 16; ++e) c3[e] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) c3 = MFMA(w3f[n][kk], tf[kk], c3);
        const int cb = 32 * (4 * ct + n) + 16 * h;
        const float* bp = reinterpret_cast<const float*>(smem + kOffB3) + cb;
        unsigned pk[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const unsigned rr = r0[8 * n + e];
          float v0 = c3[2 * e] + bp[2 * e], v1 = c3[2 * e + 1] + bp[2 * e + 1];
          v0 += __uint_as_float(rr << 16);
          v1 += __uint_as_float(rr & 0xffff0000u);
          pk[e] = pack_bf16x2(fmaxf(v0, 0.f), fmaxf(v1, 0.f));
        }
        // rows >= M lie beyond the descriptor's range and are dropped by the hardware: the two stores are ALWAYS issued,
        // which is what the counted vmcnt below relies on
        if (DBG & 1) asm volatile("" ::"v"(pk[0]), "v"(pk[1]), "v"(pk[2]), "v"(pk[3]), "v"(pk[4]), "v"(pk[5]), "v"(pk[6]), "v"(pk[7]));
        if (!(DBG & 1)) {
          if (DBG & 16) {      // timing only: the same bytes as whole 64-byte segments (4 lanes per pixel), wrong places
            const unsigned o = (unsigned)((t * 64 + 32 * pt + (lane >> 2)) * 512 + 64 * (4 * ct + n) + 16 * (lane & 3));
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{pk[0], pk[1], pk[2], pk[3]}, ysrc, m < a.M ? o : kOOB, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{pk[4], pk[5], pk[6], pk[7]}, ysrc, m < a.M ? o + 16 * 512 : kOOB, 0, 0);
          } else {
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{pk[0], pk[1], pk[2], pk[3]}, ysrc, yoff, 64 * (4 * ct + n), 0);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{pk[4], pk[5], pk[6], pk[7]}, ysrc, yoff + 16, 64 * (4 * ct + n), 0);
          }
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 32; ++e) r0[e] = r1[e];

    if (j < nc) {
      // ================= phase C: conv1 of local block j (global block b0 - 1 + j), residual rows picked up =========
      // This wave's part of slices 4j .. 4j+3 must have landed.  Vector-memory operations retire in issue order, so it is
      // enough to leave the YOUNGER ones in flight: two DMA instructions per slice issued beyond 4j+3 (at most two slices)
      // and, behind them, the eight y stores of phase B -- the stores are never waited for inside the loop.
      if (!(DBG & 2)) {
        const int ahead = 2 * (gi - 4 * j - 4) + (j >= 3 && !(DBG & 1) ? 8 : 0);
        switch (ahead) {
          case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
          case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
          case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
          case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
          case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
          default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
      }
      __builtin_amdgcn_s_barrier();                      // ... and everyone's
      asm volatile("" ::: "memory");
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
      int slot[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) slot[s] = kOffX + ((4 * j + s) % kXSlots) * 8192;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bf16x8 xf[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) xf[kk] = *reinterpret_cast<const bf16x8*>(smem + slot[s] + pfoff[kk]);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) acc = MFMA(w1f[4 * s + kk], xf[kk], acc);
      }
      // the block's own x rows in the conv3 epilogue's layout: tile nt = 4 ct + n -> channels 32 nt + 16 h .. + 15,
      // i.e. slice nt >> 1, logical chunks 4 (nt & 1) + 2 h and + 1
      {
        const int sw = (prow >> 1) & 7;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          const int cl = 4 * (n & 1) + 2 * h;
          const char* base = smem + kOffX + ((4 * j + 2 * ct + (n >> 1)) % kXSlots) * 8192 + prow * 128;
          const u32x4 lo = *reinterpret_cast<const u32x4*>(base + ((cl ^ sw) << 4));
          const u32x4 hi = *reinterpret_cast<const u32x4*>(base + (((cl + 1) ^ sw) << 4));
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            r1[8 * n + e] = lo[e];
            r1[8 * n + 4 + e] = hi[e];
          }
        }
      }
      {  // t1 = bf16(relu(acc + b1)) -> ring row of the lane's global pixel
        const int m2 = (b0 - 1 + j) * 64 + prow;
        int R = m2 % kRing;
        R += R < 0 ? kRing : 0;
        const float* bp = reinterpret_cast<const float*>(smem + kOffB1) + 32 * ct + 16 * h;
        unsigned pk[8];
#pragma unroll
        for (int e = 0; e < 8; ++e)
          pk[e] = pack_bf16x2(fmaxf(acc[2 * e] + bp[2 * e], 0.f), fmaxf(acc[2 * e + 1] + bp[2 * e + 1], 0.f));
        const int c0 = 4 * ct + 2 * h, sw = (R >> 1) & 7;
        *reinterpret_cast<u32x4*>(smem + kOffT1 + R * 128 + ((c0 ^ sw) << 4)) = u32x4{pk[0], pk[1], pk[2], pk[3]};
        *reinterpret_cast<u32x4*>(smem + kOffT1 + R * 128 + (((c0 + 1) ^ sw) << 4)) = u32x4{pk[4], pk[5], pk[6], pk[7]};
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                      // t1 visible; the four slices are free
      asm volatile("" ::: "memory");
      for (; gi < 4 * (j + 1) + kXSlots && gi < nslices; ++gi) issue_slice(gi);
    }
  }
#endif
}

int g_num_cus[64] = {};

}  // namespace

// Packed weight rows for the transposed MFMAs: row 32 T + i of the packed matrix is output channel 32 T + sigma(i),
// sigma(i) = 16 ((i >> 2) & 1) + 4 (i >> 3) + (i & 3), so that accumulator register r of lane half h (MFMA row
// (r & 3) + 8 (r >> 2) + 4 h) is channel 32 T + 16 h + r.  `src` is [rows][K] (rows % 32 == 0).
void bottleneck_pack_rows_bf16(const unsigned short* src, int rows, int K, unsigned short* dst) {
  for (int o = 0; o < rows; ++o) {
    const int T = o >> 5, i = o & 31;
    const int sigma = 16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3);
    memcpy(dst + (size_t)o * K, src + (size_t)(32 * T + sigma) * K, (size_t)K * 2);
  }
}

int bottleneck_bf16_launch(const BottleneckProblem& p, hipStream_t stream) {
  PR_REQUIRE(p.x && p.y && p.w1 && p.w2 && p.w3 && p.b1 && p.b2 && p.b3, "bottleneck: null argument");
  PR_REQUIRE(p.planes == 64, "bottleneck: 64 planes only (got %d)", p.planes);
  PR_REQUIRE(p.W >= 1 && p.W <= 63 && p.H >= 1, "bottleneck: map %dx%d unsupported (width 1..63)", p.H, p.W);
  const long M = (long)p.B * p.H * p.W;
  PR_REQUIRE(M * 512 < (1L << 31), "bottleneck: tensor too large for one launch (%ld pixels)", M);
  if (M == 0) return PR_OK;
  BnArgs a;
  a.x = reinterpret_cast<const unsigned short*>(p.x); a.y = reinterpret_cast<unsigned short*>(p.y);
  a.w1 = reinterpret_cast<const unsigned short*>(p.w1); a.w2 = reinterpret_cast<const unsigned short*>(p.w2);
  a.w3 = reinterpret_cast<const unsigned short*>(p.w3);
  a.b1 = p.b1; a.b2 = p.b2; a.b3 = p.b3;
  a.x_bytes = (unsigned)(M * 512);
  a.H = p.H; a.W = p.W; a.HW = p.H * p.W; a.M = (int)M; a.nblocks = (int)ceil_div(M, 64L);
  int dev = 0;
  PR_HIP(hipGetDevice(&dev));
  if (!g_num_cus[dev & 63]) {
    int n = 0;
    PR_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    g_num_cus[dev & 63] = n > 0 ? n : 256;
  }
  const int grid = std::min(g_num_cus[dev & 63], a.nblocks);
  void (*kern)(const BnArgs) = bottleneck64_bf16<0>;
#ifdef PR_TIMING_HOOKS
  if (const char* e = getenv("POSERISK_BN_DBG")) {
    switch (atoi(e)) {
      case 1: kern = bottleneck64_bf16<1>; break;
      case 2: kern = bottleneck64_bf16<2>; break;
      case 3: kern = bottleneck64_bf16<3>; break;
      case 4: kern = bottleneck64_bf16<4>; break;
      case 7: kern = bottleneck64_bf16<7>; break;
      case 9: kern = bottleneck64_bf16<9>; break;
      case 13: kern = bottleneck64_bf16<13>; break;
      case 16: kern = bottleneck64_bf16<16>; break;
    }
    PR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
  }
#endif
  static std::atomic<uint64_t> attr_done{0};
  PR_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(bottleneck64_bf16<0>), kLdsBytes, attr_done));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), kLdsBytes, stream, a);
  return check_launch("bottleneck64_bf16");
}

}  // namespace pr
