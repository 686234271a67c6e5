// A whole layer3 Bottleneck (plain block: 1024 -> 256 -> 256 -> 1024 channels on a map of at most 224 pixels, 14x14 in the
// encoder) as ONE bf16 kernel, ONE FRAME PER WORKGROUP:
//   t1 = relu(conv1x1(x, W1) + b1),  t2 = relu(conv3x3(t1, W2) + b2),  y = relu(conv1x1(t2, W3) + b3 + x)
// (SPIN models/hmr.py Bottleneck.forward; call site lib/core/base.py:220.)
//
// As three launches a block of layer3 takes 166 us at B = 256 (39 + 66 + 61), 3.7 x its HBM floor (x in, y out: 206 MB):
// t1 and t2 go to HBM and come back, every launch has its own prologue, tail and epilogue.  A frame of this stage is 196
// pixels: its t1 and t2 (196 x 256 bf16, rows of 528 bytes) fit LDS WHOLE, so a workgroup takes a frame and nothing of the
// 3x3 convolution crosses a workgroup: no halo, no recomputation, and at B = 256 exactly one frame per CU.
//   phase 1  conv1: sixteen 64-channel slices of x (28 KB) and W1 (32 KB) pass through two LDS stages by LDS-DMA; wave
//            (cp, ph) keeps the accumulators of channel tiles 2 cp, 2 cp + 1 x pixel tiles 4 ph .. (8 tiles, 128 VGPRs):
//            per k-step it reads 2 weight and up to 4 pixel fragments for 8 MFMAs.  Then t1 = relu(. + b1) goes to LDS.
//   phase 2  conv2: W2's fragments come straight from L2 into REGISTERS in the order the MFMAs take them (host:
//            bottleneck256_pack_w2_frags_bf16; 72 stages of 32 k, 1 KB per instruction, a ring of three stages), so the
//            phase has no barrier and no DMA and the LDS pipe carries only t1's fragments.  A tap is a row shift in t1 plus a
//            per-lane mask (masked lanes read a zero row).  t2 = relu(. + b2) overwrites the dead t1.
//   phase 3  conv3 + b3 + x + ReLU: wave w owns output channels 128 w .. 128 w + 127 as two halves of 64; per half W3's
//            fragments sit in 128 VGPRs (fragment order, from L2), the frame's pixel tiles pass by: t2 fragments from LDS,
//            the residual by LDS-DMA as 8 rows x 128 bytes per instruction into a per-wave buffer a tile ahead, finished
//            pieces written back in place and the buffer stored as 8 rows x 128 bytes per instruction (a vector-memory
//            instruction costs its CU ~2 clocks per 128-byte line it touches: scripts/micro/t_store_pattern.hip).
// Transposed MFMAs throughout (weights are the A operand, rows permuted by sigma on the host: a lane is a pixel holding 16
// consecutive channels).  Same products in the same k order as the three separate launches (conv2's k slice-major:
// conv_k_index_bf16), t1 and t2 rounded to bf16 where those launches store them: bit-identical
// (tests/test_hip_parity.py::test_bottleneck256_bf16_*).
#include <algorithm>
#include <cstdio>
#include <vector>

#include "conv_igemm.h"

namespace pr {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using i16x2 = __attribute__((ext_vector_type(2))) short;
typedef __attribute__((address_space(3))) void lds_void;

[[maybe_unused]] constexpr unsigned kOOB = 0x80000000u;
constexpr int kC = 1024, kP = 256;
constexpr int kPT = 7;                                     // pixel tiles of 32 per frame at most
constexpr int kMaxPix = 32 * kPT;
constexpr int kRowT = 2 * kP + 16;                         // bytes of a t1 / t2 row: 256 channels + 16 bytes of padding (conflict-free b128 reads)
constexpr int kStageX = kMaxPix * 128;                     // phase 1: an x slice [224][128 B] ...
constexpr int kStage1 = kStageX + kP * 128;                // ... and a W1 slice [256][128 B]
constexpr int kOffT = 0;                                   // t1 / t2 [224][528]
constexpr int kOffR = ((kMaxPix * kRowT + 1023) / 1024) * 1024;   // phase 3: eight per-wave residual / output buffers [32][128 B]
constexpr int kOffZ = kOffR + 8 * 4096;                    // zero row
constexpr int kOffB = kOffZ + 1024;                        // b1 (256), b2 (256), b3 (1024) floats
constexpr int kLds = kOffB + (kP + kP + kC) * 4;
static_assert(2 * kStage1 <= kOffZ, "phase 1's stages must end below the zero row and the biases");
static_assert(kLds <= 160 * 1024, "LDS");

struct Bn3Args {
  const unsigned short* x;
  unsigned short* y;
  const unsigned short *w1, *w2, *w3;      // [256][1024] rows permuted by sigma per 32; conv2 and conv3 in fragment order (below)
  const float *b1, *b2, *b3;
  unsigned x_bytes;
  int H, W, HW, B;
  unsigned long long* stamps;   // timing builds only (-DPR_TIMING_HOOKS, POSERISK_B256_STAMPS): s_memtime at the phase boundaries of every workgroup's first frame
};

// Two values of an epilogue at once: sums as v_pk_add_f32, the ReLU on the ROUNDED pair as one v_pk_max_i16 (a bf16 is
// negative exactly when its bits are a negative int16 and rounding keeps the sign: round(relu(v)) == relu(round(v))).
__device__ inline unsigned relu_pack2(f32x2 v) {
  const i16x2 r = __builtin_bit_cast(i16x2, __builtin_convertvector(v, bf16x2));
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(r, i16x2{0, 0}));
}



__global__ __launch_bounds__(512) void bottleneck256_bf16(const Bn3Args a) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cp = wave & 3, ph = wave >> 2;   // conv1 / conv2: channel tiles 2 cp, 2 cp + 1; pixel tiles 4 ph ..
  const int i = lane & 31, h = lane >> 5;
  const int nt = (a.HW + 31) >> 5;           // pixel tiles of a frame

  const auto xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  const auto w1src = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.w1), 0, kP * kC * 2, 0x00020000);
  const auto ysrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)a.x_bytes, 0x00020000);

  // zero row, biases (visible after the first barrier)
  if (tid < 64) *reinterpret_cast<u32x4*>(smem + kOffZ + tid * 16) = u32x4{0u, 0u, 0u, 0u};
  float* lb1 = reinterpret_cast<float*>(smem + kOffB);
  float* lb2 = lb1 + kP;
  float* lb3 = lb2 + kP;
  if (tid < kP) {
    lb1[tid] = a.b1[tid];
    lb2[tid] = a.b2[tid];
  }
  lb3[tid] = a.b3[tid];
  lb3[tid + 512] = a.b3[tid + 512];
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

  // DMA geometry: a piece is 8 LDS rows of 128 bytes; lane l writes row l >> 3, slot l & 7, which holds logical slot dq
  const int dq = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
  // fragment reads of a swizzled [rows][128 B] stage: lane reads row (tile base + i), logical slot 2 ks + h
  int foff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) foff[ks] = i * 128 + (((2 * ks + h) ^ ((i >> 1) & 7)) << 4);
  const int rpiece = (lane & 7) ^ (lane >> 3);   // the 16-byte piece a lane moves in the 8 rows x 128 B shape (row & 7 == lane >> 3)

  auto STAMP = [&](int f, int k) {
#ifdef PR_TIMING_HOOKS
    if (a.stamps && f == (int)blockIdx.x && (threadIdx.x & 63) == 0) {
      unsigned long long* dst = a.stamps + ((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 8;
      dst[k] = __builtin_amdgcn_s_memtime();
      if (k == 0) dst[7] = __builtin_amdgcn_s_memrealtime();   // 100 MHz
    }
#endif
  };
  auto frame = [&](auto n_c, int f) {
    constexpr int N = decltype(n_c)::value;  // pixel tiles of this wave in phases 1 and 2: 4 ph .. 4 ph + N - 1
    const int m0 = f * a.HW;                 // the frame's first pixel

    // ================= phase 1: conv1, t1 row = pixel of the frame =================
    auto issue1 = [&](int s) {               // slice s (channels 64 s ..): this wave's pieces of x and of W1
      char* st = smem + (s & 1) * kStage1;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int piece = wave + 8 * j;      // 28 pieces of 8 pixels
        if (piece < kMaxPix / 8) {
          const int row = 8 * piece + (lane >> 3);
          const unsigned v = row < a.HW ? (unsigned)((m0 + row) * (2 * kC) + dq * 16) : kOOB;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, (lds_void*)(st + piece * 1024), 16, v, s * 128, 0, 0);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int piece = wave + 8 * j;      // 32 pieces of 8 weight rows
        const int row = 8 * piece + (lane >> 3);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w1src, (lds_void*)(st + kStageX + piece * 1024), 16,
                                                 (unsigned)(row * (2 * kC) + dq * 16), s * 128, 0, 0);
      }
    };
    f32x16 acc1[2][N > 0 ? N : 1];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int q = 0; q < N; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc1[c][q][e] = 0.f;
    STAMP(f, 0);
    __builtin_amdgcn_s_barrier();            // the previous frame's phase 3 has read t2 and its buffers; the stages may land on them
    asm volatile("" ::: "memory");
    issue1(0);
#pragma unroll 2
    for (int s = 0; s < kC / 64; ++s) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();          // slice s is in for everyone; everyone has read slice s - 1
      asm volatile("" ::: "memory");
      if (s + 1 < kC / 64) issue1(s + 1);
      const char* st = smem + (s & 1) * kStage1;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8 wf[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) wf[c] = *reinterpret_cast<const bf16x8*>(st + kStageX + (2 * cp + c) * 4096 + foff[ks]);
#pragma unroll
        for (int q = 0; q < N; ++q) {
          const bf16x8 xf = *reinterpret_cast<const bf16x8*>(st + (4 * ph + q) * 4096 + foff[ks]);
#pragma unroll
          for (int c = 0; c < 2; ++c) acc1[c][q] = mfma_bf16_step(wf[c], xf, acc1[c][q], ks);
        }
      }
    }
    STAMP(f, 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // everyone has read the last slice: t1 may overwrite the stages
    asm volatile("" ::: "memory");

    // phase 2's W2 fragments: stage st32 = 2 (4-slice * 9 + tap) + half, [72][cp 4][c 2][ks 2][lane 64] pieces of 8 k
    auto load_w2 = [&](int st32, bf16x8 (&f)[2][2]) {
      const u32x4* src = reinterpret_cast<const u32x4*>(a.w2) + (size_t)((st32 * 4 + cp) * 4) * 64 + lane;
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int k = 0; k < 2; ++k) f[c][k] = __builtin_bit_cast(bf16x8, src[(c * 2 + k) * 64]);
    };
    bf16x8 wq[3][2][2];
    load_w2(0, wq[0]);
    load_w2(1, wq[1]);
    // t1 = bf16(relu(acc1 + b1)): the lane's 16 registers are channels 32 (2 cp + c) + 16 h + r of pixel 32 (4 ph + q) + i
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const float* bp = lb1 + 32 * (2 * cp + c) + 16 * h;
#pragma unroll
      for (int q = 0; q < N; ++q) {
        unsigned pk[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) pk[e] = relu_pack2(f32x2{acc1[c][q][2 * e], acc1[c][q][2 * e + 1]} + f32x2{bp[2 * e], bp[2 * e + 1]});
        const int r = 32 * (4 * ph + q) + i;
        if (r < a.HW) {
          char* dst = smem + kOffT + r * kRowT + (32 * (2 * cp + c) + 16 * h) * 2;
          *reinterpret_cast<u32x4*>(dst) = u32x4{pk[0], pk[1], pk[2], pk[3]};
          *reinterpret_cast<u32x4*>(dst + 16) = u32x4{pk[4], pk[5], pk[6], pk[7]};
        }
      }
    }

    // ================= phase 2: conv2 over t1 =================
    // per-lane tap masks of this wave's pixel tiles: bit t = kh * 3 + kw set when the tap's pixel lies inside the image
    unsigned mask[N > 0 ? N : 1];
#pragma unroll
    for (int q = 0; q < N; ++q) {
      const int r = 32 * (4 * ph + q) + i;
      const int yy = r / a.W, xx = r - yy * a.W;
      unsigned mk = 0;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int y2 = yy + t / 3 - 1, x2 = xx + t % 3 - 1;
        if ((unsigned)y2 < (unsigned)a.H && (unsigned)x2 < (unsigned)a.W) mk |= 1u << t;
      }
      mask[q] = r < a.HW ? mk : 0u;
    }
    f32x16 acc2[2][N > 0 ? N : 1];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int q = 0; q < N; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[c][q][e] = 0.f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // t1 is complete
    asm volatile("" ::: "memory");
    STAMP(f, 2);
    auto stage2 = [&](int st32, const bf16x8 (&wf)[2][2]) {
      const int st = st32 >> 1, half = st32 & 1;
      const int sl = st / 9, tap = st - 9 * sl;
      const int kh = tap / 3, kw = tap - 3 * kh;
      const int shift = (kh - 1) * a.W + (kw - 1);
      int ta[N > 0 ? N : 1];                 // LDS byte address of the lane's t1 row for this tap (or the zero row)
#pragma unroll
      for (int q = 0; q < N; ++q) {
        // a masked lane reads zeros from the bank its own row would have used (a t1 row is 16 bytes past a multiple of 512:
        // row r starts at bank offset 16 r mod 256), so the zero reads do not collide with the other lanes' rows
        const int r = 32 * (4 * ph + q) + i + shift, o = sl * 128 + half * 64 + h * 16;
        ta[q] = ((mask[q] >> tap) & 1u) ? kOffT + r * kRowT + o : kOffZ + ((16 * r + o) & 255);
      }
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int q = 0; q < N; ++q) {
          const bf16x8 tf = *reinterpret_cast<const bf16x8*>(smem + ta[q] + k * 32);
#pragma unroll
          for (int c = 0; c < 2; ++c) acc2[c][q] = mfma_bf16_step(wf[c][k], tf, acc2[c][q], k);
        }
    };
#pragma unroll 1
    for (int p = 0; p < 24; ++p) {           // stages 3 p, 3 p + 1, 3 p + 2: the register ring's three buffers by name
      load_w2(3 * p + 2, wq[2]);
      stage2(3 * p, wq[0]);
      if (p < 23) load_w2(3 * p + 3, wq[0]);
      stage2(3 * p + 1, wq[1]);
      if (p < 23) load_w2(3 * p + 4, wq[1]);
      stage2(3 * p + 2, wq[2]);
    }
    STAMP(f, 3);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // everyone has read t1: t2 may overwrite it
    asm volatile("" ::: "memory");
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const float* bp = lb2 + 32 * (2 * cp + c) + 16 * h;
#pragma unroll
      for (int q = 0; q < N; ++q) {
        unsigned pk[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) pk[e] = relu_pack2(f32x2{acc2[c][q][2 * e], acc2[c][q][2 * e + 1]} + f32x2{bp[2 * e], bp[2 * e + 1]});
        const int r = 32 * (4 * ph + q) + i;
        if (r < a.HW) {
          char* dst = smem + kOffT + r * kRowT + (32 * (2 * cp + c) + 16 * h) * 2;
          *reinterpret_cast<u32x4*>(dst) = u32x4{pk[0], pk[1], pk[2], pk[3]};
          *reinterpret_cast<u32x4*>(dst + 16) = u32x4{pk[4], pk[5], pk[6], pk[7]};
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // t2 is complete
    asm volatile("" ::: "memory");
    STAMP(f, 4);
  };

  // ================= phase 3: conv3 + b3 + x + ReLU, this wave's 128 output channels as two halves of 64 =================
  auto phase3 = [&](int f) {
    const int m0 = f * a.HW;
    char* buf = smem + kOffR + wave * 4096;
    for (int half = 0; half < 2; ++half) {
      const int cq = 2 * wave + half;        // 64-channel group: channels 64 cq .. 64 cq + 63
      const int csoff = 128 * cq;            // its byte offset in a row of x / y
      bf16x8 w3f[2][16];                     // W3 fragments [cq 16][c 2][ks 16][lane 64]
      {
        const u32x4* src = reinterpret_cast<const u32x4*>(a.w3) + (size_t)cq * 32 * 64 + lane;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int ks = 0; ks < 16; ++ks) w3f[c][ks] = __builtin_bit_cast(bf16x8, src[(c * 16 + ks) * 64]);
      }
      auto dma_res = [&](int pt) {           // the residual of pixel tile pt: 4 instructions of 8 rows x 128 bytes
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int r = 32 * pt + 8 * q + (lane >> 3);
          const unsigned v = r < a.HW ? (unsigned)((m0 + r) * (2 * kC) + 16 * rpiece) : kOOB;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, (lds_void*)(buf + q * 1024), 16, v, csoff, 0, 0);
        }
      };
      dma_res(0);
      for (int pt = 0; pt < nt; ++pt) {
        f32x16 acc[2];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x8 tf[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) tf[k] = *reinterpret_cast<const bf16x8*>(smem + kOffT + (32 * pt + i) * kRowT + (4 * g + k) * 32 + h * 16);
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[c] = mfma_bf16_step(w3f[c][4 * g + k], tf[k], acc[c], k);
        }
        // the tile's residual has landed when at most the previous tile's 4 stores, issued behind its DMA, are outstanding
        if (pt > 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          char* p0 = buf + i * 128 + (((4 * c + 2 * h) ^ (i & 7)) << 4);
          char* p1 = buf + i * 128 + (((4 * c + 2 * h + 1) ^ (i & 7)) << 4);
          const u32x4 r0 = *reinterpret_cast<const u32x4*>(p0), r1 = *reinterpret_cast<const u32x4*>(p1);
          const float* bp = lb3 + 64 * cq + 32 * c + 16 * h;
          unsigned pk[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const unsigned r2 = e < 4 ? r0[e & 3] : r1[e & 3];
            f32x2 v = f32x2{acc[c][2 * e], acc[c][2 * e + 1]} + f32x2{bp[2 * e], bp[2 * e + 1]};
            v += f32x2{__uint_as_float(r2 << 16), __uint_as_float(r2 & 0xffff0000u)};
            pk[e] = relu_pack2(v);
          }
          *reinterpret_cast<u32x4*>(p0) = u32x4{pk[0], pk[1], pk[2], pk[3]};
          *reinterpret_cast<u32x4*>(p1) = u32x4{pk[4], pk[5], pk[6], pk[7]};
        }
        u32x4 o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = *reinterpret_cast<const u32x4*>(buf + q * 1024 + lane * 16);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the buffer has been read: the next tile's residual may land in it
        if (pt + 1 < nt) dma_res(pt + 1);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int r = 32 * pt + 8 * q + (lane >> 3);
          const unsigned yoff = r < a.HW ? (unsigned)((m0 + r) * (2 * kC) + 16 * rpiece) : kOOB;
          buffer_store_b128_sreg(o[q], ysrc, yoff, csoff);
        }
      }
      STAMP(f, 5 + half);
    }
  };

  for (int f = blockIdx.x; f < a.B; f += gridDim.x) {
    switch (std::min(std::max(nt - 4 * ph, 0), 4)) {
      case 4: frame(std::integral_constant<int, 4>{}, f); break;
      case 3: frame(std::integral_constant<int, 3>{}, f); break;
      case 2: frame(std::integral_constant<int, 2>{}, f); break;
      case 1: frame(std::integral_constant<int, 1>{}, f); break;
      default: frame(std::integral_constant<int, 0>{}, f); break;
    }
    phase3(f);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

}  // namespace

// (the fragment orders of w2 / w3 -- bottleneck256_pack_w2_frags_bf16 / _w3_ -- and bottleneck256_bf16_fits live in host_plan.cc)
static_assert(kMaxPix == 224, "host_plan.cc::bottleneck256_bf16_fits states this bound");

int bottleneck256_bf16_launch(const BottleneckProblem& p, hipStream_t stream) {
  PR_REQUIRE(p.x && p.y && p.w1 && p.w2 && p.w3 && p.b1 && p.b2 && p.b3, "bottleneck256: null argument");
  PR_REQUIRE(p.planes == kP && !p.first, "bottleneck256: a plain block with 256 planes (1024 channels)");
  PR_REQUIRE(bottleneck256_bf16_fits(p.H, p.W), "bottleneck256: a map of at most %d pixels (got %d x %d)", kMaxPix, p.H, p.W);
  const long M = (long)p.B * p.H * p.W;
  PR_REQUIRE(M >= 0 && M * kC * 2 < (1L << 31), "bottleneck256: tensor too large for one launch (%ld pixels)", M);
  if (M == 0) return PR_OK;
  Bn3Args a;
  a.x = reinterpret_cast<const unsigned short*>(p.x); a.y = reinterpret_cast<unsigned short*>(p.y);
  a.w1 = reinterpret_cast<const unsigned short*>(p.w1); a.w2 = reinterpret_cast<const unsigned short*>(p.w2);
  a.w3 = reinterpret_cast<const unsigned short*>(p.w3);
  a.b1 = p.b1; a.b2 = p.b2; a.b3 = p.b3;
  a.x_bytes = (unsigned)(M * kC * 2);
  a.H = p.H; a.W = p.W; a.HW = p.H * p.W; a.B = p.B;
  int cus = 256;
  PR_TRY(current_device_cus(&cus));
  static std::atomic<uint64_t> done{0};
  PR_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(bottleneck256_bf16), kLds, done));
  a.stamps = nullptr;
#ifdef PR_TIMING_HOOKS
  static unsigned long long* stamp_buf = nullptr;
  static int stamp_calls = 0;
  if (const char* path = getenv("POSERISK_B256_STAMPS")) {
    const size_t n = (size_t)256 * 8 * 8;
    if (!stamp_buf) PR_HIP(hipMalloc(&stamp_buf, n * 8));
    a.stamps = stamp_buf;
    if (++stamp_calls == 20) {
      PR_HIP(hipMemsetAsync(stamp_buf, 0, n * 8, stream));
      hipLaunchKernelGGL(bottleneck256_bf16, dim3(std::min(p.B, cus)), dim3(512), kLds, stream, a);
      std::vector<unsigned long long> host(n);
      PR_HIP(hipStreamSynchronize(stream));
      PR_HIP(hipMemcpy(host.data(), stamp_buf, n * 8, hipMemcpyDeviceToHost));
      if (FILE* fo = fopen(path, "wb")) {
        fwrite(host.data(), 8, n, fo);
        fclose(fo);
      }
      return check_launch("bottleneck256_bf16");
    }
  }
#endif
  hipLaunchKernelGGL(bottleneck256_bf16, dim3(std::min(p.B, cus)), dim3(512), kLds, stream, a);
  return check_launch("bottleneck256_bf16");
}

}  // namespace pr
