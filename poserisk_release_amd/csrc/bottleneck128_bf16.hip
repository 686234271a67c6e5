// A whole layer2 Bottleneck (plain block: 512 -> 128 -> 128 -> 512 channels on a W <= 31 map) as ONE persistent bf16 kernel:
//   t1 = relu(conv1x1(x, W1) + b1),  t2 = relu(conv3x3(t1, W2) + b2),  y = relu(conv1x1(t2, W3) + b3 + x)
// (SPIN models/hmr.py Bottleneck.forward; call site lib/core/base.py:220.)
//
// As three launches a block moves 820 MB at B = 256 (x twice, t1 and t2 written and read back) in 233 us; fused it moves x in
// (once more for the residual, out of L2 / the Infinity Cache) and y out.  layer1's structure (bottleneck_bf16.hip: every
// weight matrix resident, 64-pixel blocks) does not carry over: this block's weights are 544 KB.  What fits is to stream the
// weights through LDS ONCE PER CHUNK of up to 256 pixels, with the loops turned inside out so that a chunk's accumulators stay
// in registers while the weights pass:
//   phase 1  conv1 for the chunk's pixel tiles and one halo tile on either side (conv2 reaches W + 1 <= 32 pixels of the
//            flattened [B H W] index up and down): K-SLICE loop outside -- eight 64-channel slices of x (40 KB) and W1 (16 KB)
//            pass through two LDS stages, a wave keeps 5 accumulator tiles (pixel tiles q, q + 2, .. x one channel tile);
//            then t1 = relu(. + b1) goes to LDS once (rows of 272 bytes: any 32 consecutive rows are conflict free);
//   phase 2  conv2: STAGE loop outside -- the 18 (slice, tap) stages of W2 (16 KB each, packed slice-major:
//            conv_k_index_bf16) pass through a ring of four, two per barrier, a wave keeps 4 accumulator tiles; a tap is a row
//            shift in t1 plus a per-lane mask (masked lanes read a zero row); t2 = relu(. + b2) overwrites the dead t1;
//   phase 3  conv3 + b3 + x + ReLU: W3's rows in registers (64 VGPRs per wave), t2 fragments from LDS; the residual comes by
//            LDS-DMA and y leaves through the same per-wave LDS buffer, both as 8 rows x 128 bytes per instruction (see the
//            phase's comment: an instruction costs ~2 clocks per 128-byte line it touches).
// Every second workgroup opens with a short chunk (BottleneckProblem::lead_tiles), which takes the workgroups' memory-heavy
// phases out of step with each other.
// Transposed MFMAs throughout (weights are the A operand, rows permuted by sigma on the host: a lane is a pixel holding 16
// consecutive channels).  Same products in the same k order as the three separate launches, t1 and t2 rounded to bf16 where
// those launches store them: bit-identical (tests/test_hip_parity.py::test_bottleneck128_bf16_*).  25 % of conv1 is
// recomputed in the halo tiles.
#include <algorithm>
#include <cstdio>
#include <vector>

#include "conv_igemm.h"

namespace pr {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
typedef __attribute__((address_space(3))) void lds_void;

[[maybe_unused]] constexpr unsigned kOOB = 0x80000000u;
constexpr int kC = 512, kP = 128;           // block channels, planes
constexpr int kCT = 8;                      // pixel tiles (32 pixels) per chunk
constexpr int kRowT = 272;                  // bytes of a t1 / t2 row: 128 channels + 16 bytes of padding
constexpr int kTRows = 32 * (kCT + 2);      // t1 rows: the chunk and a halo tile on either side
constexpr int kOffT = 0;                                   // t1 / t2
constexpr int kStage1 = kTRows * 128 + kP * 128;           // phase 1: an x slice [320][128 B] + a W1 slice [128][128 B]
constexpr int kOffW2 = kTRows * kRowT;                     // phase 2: ring of four W2 stages [128][128 B], two per barrier
constexpr int kOffZ = kOffW2 + 4 * kP * 128;               // 512 zero bytes: what a masked conv2 tap reads
constexpr int kOffB = kOffZ + 512;                         // b1 (128), b2 (128), b3 (512) floats
constexpr int kLds = kOffB + (kP + kP + kC) * 4;
static_assert(2 * kStage1 <= kOffZ, "phase 1's stages must end below the zero row and the biases");
static_assert(kLds <= 160 * 1024, "LDS");

struct Bn2Args {
  const unsigned short* x;
  unsigned short* y;
  const unsigned short *w1, *w2, *w3;      // [128][512], [128][1152] (slice-major k), [512][128]; rows permuted by sigma per 32
  const float *b1, *b2, *b3;
  unsigned x_bytes;
  int H, W, HW, M, T, runs, lead;
  int dbg;                      // timing builds only (POSERISK_B128_DBG): 1 no output stores, 2 no residual loads, 4 no x DMA, 8 no W2 DMA, 16 no W1 DMA
  unsigned long long* stamps;   // timing builds only (-DPR_TIMING_HOOKS, POSERISK_B128_STAMPS): s_memtime at the phase boundaries of chunk 1
};

__device__ inline unsigned pack2(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}
// Two values of an epilogue at once: the sums as one v_pk_add_f32 each, the ReLU on the ROUNDED pair as one v_pk_max_i16
// (a bf16 is negative exactly when its bits are a negative int16, and rounding keeps the sign: round(relu(v)) == relu(round(v))
// for every finite v and both infinities, -0 included).  6 VALU operations per pair instead of 9.
using i16x2 = __attribute__((ext_vector_type(2))) short;
__device__ inline unsigned relu_pack2(f32x2 v) {
  const i16x2 r = __builtin_bit_cast(i16x2, __builtin_convertvector(v, bf16x2));
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(r, i16x2{0, 0}));
}

__global__ __launch_bounds__(512) void bottleneck128_bf16(const Bn2Args a) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int run = blockIdx.x;
  if (run >= a.runs) return;
  const int t_begin = (int)((long)a.T * run / a.runs), t_end = (int)((long)a.T * (run + 1) / a.runs);
  const int ntiles = t_end - t_begin;
  if (ntiles <= 0) return;
  // chunks: an optional short leading chunk (a.lead tiles, every second workgroup: takes the workgroups' memory-heavy
  // phases out of step), then the rest in equal chunks of at most kCT tiles
  const int lead = ((run & 1) && a.lead > 0 && ntiles > a.lead + kCT / 2) ? a.lead : 0;
  const int nrest = (ntiles - lead + kCT - 1) / kCT;
  const int cbase = (ntiles - lead) / nrest, cextra = (ntiles - lead) - cbase * nrest;   // rest chunk c has cbase + (c < cextra) tiles
  const int nchunks = nrest + (lead ? 1 : 0);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = wave & 3, hw = wave >> 2;   // conv1 / conv2: channel tile, parity of the pixel tiles this wave takes
  const int i = lane & 31, h = lane >> 5;

  const auto xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  const auto w1src = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.w1), 0, kP * kC * 2, 0x00020000);
  const auto w2src = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.w2), 0, kP * 9 * kP * 2, 0x00020000);
  const auto ysrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)a.x_bytes, 0x00020000);

  // zero row, biases (visible after the first barrier)
  if (tid < 32) *reinterpret_cast<u32x4*>(smem + kOffZ + tid * 16) = u32x4{0u, 0u, 0u, 0u};
  float* lb1 = reinterpret_cast<float*>(smem + kOffB);
  float* lb2 = lb1 + kP;
  float* lb3 = lb2 + kP;
  if (tid < kP) {
    lb1[tid] = a.b1[tid];
    lb2[tid] = a.b2[tid];
  }
  lb3[tid] = a.b3[tid];

  // W3 rows of this wave's 64 output channels as MFMA A fragments for the whole kernel (rows already in sigma order)
  bf16x8 w3f[2][8];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
      w3f[n][ks] = *reinterpret_cast<const bf16x8*>(a.w3 + (32 * (2 * wave + n) + i) * kP + 16 * ks + 8 * h);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // nothing of the set-up is counted among the rings' operations

  // DMA geometry: a piece is 8 LDS rows of 128 bytes; lane l writes row l >> 3, slot l & 7, which holds logical slot dq
  const int dq = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
  // fragment reads of a swizzled [rows][128 B] stage: lane reads row (tile base + i), logical slot 2 ks + h
  int foff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) foff[ks] = i * 128 + (((2 * ks + h) ^ ((i >> 1) & 7)) << 4);

#ifdef PR_TIMING_HOOKS
  const int dbg = a.dbg;
#else
  constexpr int dbg = 0;
#endif
#ifdef PR_TIMING_HOOKS
  if (dbg & 32)                          // start the workgroups in three groups a third of a chunk apart
    for (int z = 0; z < 4 * (int)(blockIdx.x % 3); ++z) __builtin_amdgcn_s_sleep(127);
  if (dbg & 128)                         // ... or in two groups half a chunk apart
    for (int z = 0; z < 6 * (int)(blockIdx.x & 1); ++z) __builtin_amdgcn_s_sleep(127);
#endif
  auto STAMP = [&](int c, int k) {
#ifdef PR_TIMING_HOOKS
    if (a.stamps && c < 4 && (threadIdx.x & 63) == 0) {
      unsigned long long* dst = a.stamps + (((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 4 + c) * 8;
      dst[k] = __builtin_amdgcn_s_memtime();
      if (k == 0) dst[7] = __builtin_amdgcn_s_memrealtime();   // 100 MHz: the shader clock is d(memtime) / d(memrealtime) x 100 MHz
    }
#endif
  };
  auto chunk_first = [&](int c) {
    if (lead) {
      if (c == 0) return t_begin;
      --c;
    }
    return t_begin + lead + c * cbase + (c < cextra ? c : cextra);
  };
  auto chunk_tiles = [&](int c) {
    if (lead) {
      if (c == 0) return lead;
      --c;
    }
    return cbase + (c < cextra ? 1 : 0);
  };

  auto chunk = [&](auto n2_c, int c_idx) {
    constexpr int N2 = decltype(n2_c)::value, N1 = N2 + 1;   // conv2 / conv1 pixel tiles of this wave
    const int n = chunk_tiles(c_idx), m0 = chunk_first(c_idx) * 32;

    const int csoff = 128 * wave;        // byte offset of this wave's first channel in a row of x / y
    const int rbuf = kOffW2 + wave * 8192;
    const int rpiece = (lane & 7) ^ (lane >> 3);            // the piece a lane moves in the row-contiguous shape (row & 7 == lane >> 3)
    auto dma_res = [&](int pt) {         // ALWAYS 4 instructions
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m = m0 + 32 * pt + 8 * q + (lane >> 3);
        const unsigned v = (pt < n && m < a.M && !(dbg & 2)) ? (unsigned)(m * (2 * kC) + 16 * rpiece) : kOOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, (lds_void*)(smem + rbuf + (pt & 1) * 4096 + q * 1024), 16, v, csoff, 0, 0);
      }
    };

    // ================= phase 1: t1 rows 0 .. 32 (n + 2) - 1 <-> pixels m0 - 32 + row =================
    auto issue1 = [&](int s) {           // slice s (channels 64 s ..): ALWAYS 5 pieces of x and 2 of W1
      char* st = smem + (s & 1) * kStage1;
      if (!(dbg & 4))
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int row = 8 * (wave + 8 * j) + (lane >> 3);
        const int m = m0 - 32 + row;
        const unsigned v = (row < 32 * (n + 2) && (unsigned)m < (unsigned)a.M) ? (unsigned)(m * (2 * kC) + dq * 16) : kOOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, (lds_void*)(st + (wave + 8 * j) * 1024), 16, v, s * 128, 0, 0);
      }
      if (!(dbg & 16))
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = 8 * (wave + 8 * j) + (lane >> 3);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w1src, (lds_void*)(st + kTRows * 128 + (wave + 8 * j) * 1024), 16,
                                                 (unsigned)(row * (2 * kC) + dq * 16), s * 128, 0, 0);
      }
    };
    f32x16 acc1[N1];
#pragma unroll
    for (int q = 0; q < N1; ++q)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc1[q][e] = 0.f;
    STAMP(c_idx, 0);
    __builtin_amdgcn_s_barrier();        // the previous chunk's phase 3 has read t2; the stages may land on it
    asm volatile("" ::: "memory");
    STAMP(c_idx, 1);
    issue1(0);
    for (int s = 0; s < kC / 64; ++s) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();      // slice s is in for everyone; everyone has read slice s - 1
      asm volatile("" ::: "memory");
      if (s + 1 < kC / 64) issue1(s + 1);
      const char* st = smem + (s & 1) * kStage1;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(st + kTRows * 128 + ct * 4096 + foff[ks]);
#pragma unroll
        for (int q = 0; q < N1; ++q) {
          const bf16x8 xf = *reinterpret_cast<const bf16x8*>(st + (hw + 2 * q) * 4096 + foff[ks]);
          acc1[q] = mfma_bf16_step(wf, xf, acc1[q], ks);
        }
      }
    }
    STAMP(c_idx, 2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();        // everyone has read the last slice: t1 may overwrite the stages
    asm volatile("" ::: "memory");
    // phase 2's first two W2 stages go out now (their ring lies above t1)
    auto issue2 = [&](int st) {          // stage st = (slice st / 9, tap st % 9): ALWAYS 2 pieces
      char* dst = smem + kOffW2 + (st & 3) * (kP * 128);
      if (!(dbg & 8))
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = 8 * (wave + 8 * j) + (lane >> 3);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w2src, (lds_void*)(dst + (wave + 8 * j) * 1024), 16,
                                                 (unsigned)(row * (2 * 9 * kP) + dq * 16), st * 128, 0, 0);
      }
    };
    issue2(0);
    issue2(1);
    {
      const float* bp = lb1 + 32 * ct + 16 * h;
#pragma unroll
      for (int q = 0; q < N1; ++q) {
        unsigned pk[8];
#pragma unroll
        for (int e = 0; e < 8; ++e)
          pk[e] = relu_pack2(f32x2{acc1[q][2 * e], acc1[q][2 * e + 1]} + f32x2{bp[2 * e], bp[2 * e + 1]});
        char* dst = smem + kOffT + (32 * (hw + 2 * q) + i) * kRowT + (32 * ct + 16 * h) * 2;
        *reinterpret_cast<u32x4*>(dst) = u32x4{pk[0], pk[1], pk[2], pk[3]};
        *reinterpret_cast<u32x4*>(dst + 16) = u32x4{pk[4], pk[5], pk[6], pk[7]};
      }
    }

    STAMP(c_idx, 3);
    // ================= phase 2: conv2 over t1 =================
    // per-lane tap masks of this wave's conv2 tiles: bit t = kh * 3 + kw set when the tap's pixel lies inside the image
    unsigned mask[N2 > 0 ? N2 : 1];
#pragma unroll
    for (int q = 0; q < N2; ++q) {
      const int m = m0 + 32 * (hw + 2 * q) + i;
      const int rem = m % a.HW, yy = rem / a.W, xx = rem - yy * a.W;
      unsigned mk = 0;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int y2 = yy + t / 3 - 1, x2 = xx + t % 3 - 1;
        if ((unsigned)y2 < (unsigned)a.H && (unsigned)x2 < (unsigned)a.W) mk |= 1u << t;
      }
      mask[q] = m < a.M ? mk : 0u;
    }
    int tbase[N2 > 0 ? N2 : 1];          // the rows' own LDS addresses (tap (1, 1), slice 0), this lane's k half
#pragma unroll
    for (int q = 0; q < N2; ++q) tbase[q] = kOffT + (32 + 32 * (hw + 2 * q) + i) * kRowT + h * 16;
    const int zlane = (16 * i + h * 16) & 255;
    f32x16 acc2[N2 > 0 ? N2 : 1];
#pragma unroll
    for (int q = 0; q < N2; ++q)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc2[q][e] = 0.f;
    // Nine intervals of TWO stages each (one barrier per 32 MFMAs of a wave; the overheads of an interval -- wait, barrier,
    // DMA issue, first fragment reads -- cost as much as 16 MFMAs): stages 2 p, 2 p + 1 are computed while 2 p + 2, 2 p + 3
    // land in the other half of the ring; they were issued behind the barrier of interval p, so the wait is for everything.
#pragma unroll 1
    for (int p = 0; p < 9; ++p) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();      // stages 2 p, 2 p + 1 (and, the first time, t1) are in for everyone; 2 p - 2, 2 p - 1 are read
      asm volatile("" ::: "memory");
      if (2 * p + 2 < 18) {
        issue2(2 * p + 2);
        issue2(2 * p + 3);
      }
#pragma unroll 1
      for (int u = 0; u < 2; ++u) {
        const int st = 2 * p + u;
        const int sl = st >= 9 ? 1 : 0, tap = st - 9 * sl;
        const int kh = tap / 3, kw = tap - 3 * kh;
        const int shift = (kh - 1) * a.W + (kw - 1);
        const char* ws = smem + kOffW2 + (st & 3) * (kP * 128) + ct * 4096;
        // LDS byte address of the lane's t1 row for this tap, or of zeros: a masked lane reads zeros from the bank its own row
        // would have used (a t1 row is 16 bytes past a multiple of 256: row r starts at bank offset 16 r mod 256, the same for
        // every pixel tile of the lane), so the zero reads do not collide with the other lanes' rows.  Branch-free (round 6:
        // hipcc turned the select into a divergent branch per row): one add per row, the tap's offset is wave-uniform.
        const int soff = shift * kRowT + sl * 128;
        const int zaddr = kOffZ + ((zlane + 16 * shift + sl * 128) & 255);
        int ta[N2 > 0 ? N2 : 1];
#pragma unroll
        for (int q = 0; q < N2; ++q) {
          const int keep = -(int)((mask[q] >> tap) & 1u);                          // all ones / zero
          ta[q] = zaddr + ((tbase[q] + soff - zaddr) & keep);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const bf16x8 wf = *reinterpret_cast<const bf16x8*>(ws + foff[ks]);
#pragma unroll
          for (int q = 0; q < N2; ++q) {
            const bf16x8 tf = *reinterpret_cast<const bf16x8*>(smem + ta[q] + ks * 32);
            acc2[q] = mfma_bf16_step(wf, tf, acc2[q], ks);
          }
        }
      }
    }
    STAMP(c_idx, 4);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();        // everyone has read t1: t2 may overwrite it (and the W2 ring is free: phase 3's buffers)
    asm volatile("" ::: "memory");
    dma_res(0);
    dma_res(1);
    {
      const float* bp = lb2 + 32 * ct + 16 * h;
#pragma unroll
      for (int q = 0; q < N2; ++q) {
        unsigned pk[8];
#pragma unroll
        for (int e = 0; e < 8; ++e)
          pk[e] = relu_pack2(f32x2{acc2[q][2 * e], acc2[q][2 * e + 1]} + f32x2{bp[2 * e], bp[2 * e + 1]});
        char* dst = smem + kOffT + (32 * (hw + 2 * q) + i) * kRowT + (32 * ct + 16 * h) * 2;
        *reinterpret_cast<u32x4*>(dst) = u32x4{pk[0], pk[1], pk[2], pk[3]};
        *reinterpret_cast<u32x4*>(dst + 16) = u32x4{pk[4], pk[5], pk[6], pk[7]};
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();        // t2 is complete
    asm volatile("" ::: "memory");
    STAMP(c_idx, 5);

    f32x2 b3r[2][8];                     // this wave's conv3 biases for the chunk (re-read per chunk: the registers are acc1 / acc2's)
#pragma unroll
    for (int nn = 0; nn < 2; ++nn)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float* bp = lb3 + 32 * (2 * wave + nn) + 16 * h + 2 * e;
        b3r[nn][e] = f32x2{bp[0], bp[1]};
      }
    // ================= phase 3: conv3 + b3 + x + ReLU, this wave's 64 output channels of every pixel tile =================
    // A vector-memory instruction costs the CU about two clocks per 128-byte line it touches (scripts/micro/
    // t_store_pattern.hip: 18 B/clk and CU for the accumulator layout's 32 rows per instruction, 54 B/clk for 8 rows of 128
    // contiguous bytes), and this phase moves 2 x 32 KB per pixel tile.  So neither the residual nor the output moves in the
    // accumulator layout: the residual comes by LDS-DMA as 8 rows x 128 bytes per instruction into a per-wave buffer
    // (two tiles ahead, the W2 ring is idle in this phase), the lanes read their pieces out of it, write the finished
    // pieces back IN PLACE, and the wave stores the buffer as 8 rows x 128 bytes per instruction.  16-byte piece p of a
    // row lies in slot p ^ (row & 7), which makes both access shapes conflict free.
    auto tile3 = [&](int pt) {
      f32x16 acc[2];
#pragma unroll
      for (int nn = 0; nn < 2; ++nn)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[nn][e] = 0.f;
      bf16x8 tf[8];
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) tf[ks] = *reinterpret_cast<const bf16x8*>(smem + kOffT + (32 * pt + i) * kRowT + ks * 32 + h * 16);
#pragma unroll
      for (int nn = 0; nn < 2; ++nn)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) acc[nn] = mfma_bf16_step(w3f[nn][ks], tf[ks], acc[nn], ks);
      // the tile's residual has landed when at most the operations issued behind its DMA are outstanding: the previous
      // tile's 4 stores (none before the first tile) and the next tile's 4 DMA instructions (none behind the last tile's)
      const int younger = (pt > 0 ? 4 : 0) + (pt + 1 < n ? 4 : 0);
      if (younger == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (younger == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      char* buf = smem + rbuf + (pt & 1) * 4096;
#pragma unroll
      for (int nn = 0; nn < 2; ++nn) {
        char* p0 = buf + i * 128 + (((4 * nn + 2 * h) ^ (i & 7)) << 4);
        char* p1 = buf + i * 128 + (((4 * nn + 2 * h + 1) ^ (i & 7)) << 4);
        const u32x4 r0 = *reinterpret_cast<const u32x4*>(p0), r1 = *reinterpret_cast<const u32x4*>(p1);
        unsigned pk[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const unsigned r2 = e < 4 ? r0[e & 3] : r1[e & 3];
          f32x2 v = f32x2{acc[nn][2 * e], acc[nn][2 * e + 1]} + b3r[nn][e];
          v += f32x2{__uint_as_float(r2 << 16), __uint_as_float(r2 & 0xffff0000u)};
          pk[e] = relu_pack2(v);
        }
        *reinterpret_cast<u32x4*>(p0) = u32x4{pk[0], pk[1], pk[2], pk[3]};
        *reinterpret_cast<u32x4*>(p1) = u32x4{pk[4], pk[5], pk[6], pk[7]};
      }
      u32x4 o[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] = *reinterpret_cast<const u32x4*>(buf + q * 1024 + lane * 16);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m = m0 + 32 * pt + 8 * q + (lane >> 3);
        const unsigned yoff = (m < a.M && !(dbg & 1)) ? (unsigned)(m * (2 * kC) + 16 * rpiece) : kOOB;
        buffer_store_b128_sreg(o[q], ysrc, yoff, csoff);
      }
    };
    for (int pt = 0; pt < n; ++pt) {
      tile3(pt);
      if (pt + 2 < n) dma_res(pt + 2);   // into the buffer whose rows the stores above have just read
    }
    STAMP(c_idx, 6);
  };

  for (int c = 0; c < nchunks; ++c) {
    const int n = chunk_tiles(c);
    switch ((n - hw + 1) / 2) {          // conv2 pixel tiles of this wave: hw, hw + 2, .. < n
      case 4: chunk(std::integral_constant<int, 4>{}, c); break;
      case 3: chunk(std::integral_constant<int, 3>{}, c); break;
      case 2: chunk(std::integral_constant<int, 2>{}, c); break;
      case 1: chunk(std::integral_constant<int, 1>{}, c); break;
      default: chunk(std::integral_constant<int, 0>{}, c); break;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

}  // namespace

int bottleneck128_bf16_launch(const BottleneckProblem& p, hipStream_t stream) {
  PR_REQUIRE(p.x && p.y && p.w1 && p.w2 && p.w3 && p.b1 && p.b2 && p.b3, "bottleneck128: null argument");
  PR_REQUIRE(p.planes == kP && !p.first, "bottleneck128: a plain block with 128 planes (512 channels)");
  PR_REQUIRE(p.W >= 1 && p.W <= 31 && p.H >= 1, "bottleneck128: map width 1..31 (got %d)", p.W);
  const long M = (long)p.B * p.H * p.W;
  PR_REQUIRE(M >= 0 && M * kC * 2 < (1L << 31), "bottleneck128: tensor too large for one launch (%ld pixels)", M);
  if (M == 0) return PR_OK;
  Bn2Args a;
  a.x = reinterpret_cast<const unsigned short*>(p.x); a.y = reinterpret_cast<unsigned short*>(p.y);
  a.w1 = reinterpret_cast<const unsigned short*>(p.w1); a.w2 = reinterpret_cast<const unsigned short*>(p.w2);
  a.w3 = reinterpret_cast<const unsigned short*>(p.w3);
  a.b1 = p.b1; a.b2 = p.b2; a.b3 = p.b3;
  a.x_bytes = (unsigned)(M * kC * 2);
  a.H = p.H; a.W = p.W; a.HW = p.H * p.W; a.M = (int)M; a.T = (int)ceil_div(M, 32L);
  int cus = 256;
  PR_TRY(current_device_cus(&cus));
  a.runs = std::min(cus, std::max(a.T / 4, 1));
  a.stamps = nullptr;
  a.dbg = 0;
  a.lead = std::max(p.lead_tiles, 0);   // 2 by default: 190 -> 181 us at B=256
  static std::atomic<uint64_t> done{0};
  PR_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(bottleneck128_bf16), kLds, done));
#ifdef PR_TIMING_HOOKS
  static unsigned long long* stamp_buf = nullptr;
  static int stamp_calls = 0;
  if (const char* e = getenv("POSERISK_B128_DBG")) a.dbg = atoi(e);
  if (a.dbg & 64) a.runs = std::max(a.runs / 2, 1);
  if (const char* path = getenv("POSERISK_B128_STAMPS")) {
    const size_t n = (size_t)256 * 8 * 4 * 8;
    if (!stamp_buf) PR_HIP(hipMalloc(&stamp_buf, n * 8));
    a.stamps = stamp_buf;
    if (++stamp_calls == 20) {
      PR_HIP(hipMemsetAsync(stamp_buf, 0, n * 8, stream));
      hipLaunchKernelGGL(bottleneck128_bf16, dim3(a.runs), dim3(512), kLds, stream, a);
      std::vector<unsigned long long> host(n);
      PR_HIP(hipStreamSynchronize(stream));
      PR_HIP(hipMemcpy(host.data(), stamp_buf, n * 8, hipMemcpyDeviceToHost));
      if (FILE* f = fopen(path, "wb")) {
        fwrite(host.data(), 8, n, f);
        fclose(f);
      }
      return check_launch("bottleneck128_bf16");
    }
  }
#endif
  hipLaunchKernelGGL(bottleneck128_bf16, dim3(a.runs), dim3(512), kLds, stream, a);
  return check_launch("bottleneck128_bf16");
}

}  // namespace pr
