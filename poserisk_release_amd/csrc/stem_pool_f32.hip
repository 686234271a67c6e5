// The fp32 encoder's stem in ONE kernel: conv1 (7x7 / stride 2 as the 4x4 / stride-1 convolution over the 12-channel
// space-to-depth image, hmr.hip) + bn1 (folded) + ReLU + MaxPool2d(3, 2, 1).  Replaces two launches of SPIN's HMR stem
// (call site lib/core/base.py:220): the 112x112x64 map (205 MB at B=64) is neither written nor read back, and the
// convolution no longer pays the tile kernel's operand traffic -- its weights live in registers.
//
//   workgroup = (image, band of 7 pooled rows) -> 15 conv rows (one shared with the band above: 7 % recomputed), 4 waves;
//     64 images x 8 bands = 512 workgroups = two per CU at B=64.
//   wave w owns output channels 16w .. 16w+15 for the whole width: a conv row is 7 pixel tiles x 16 channels on
//     v_mfma_f32_16x16x4_f32, 7 independent accumulators, K = 16 taps x 12 channels regrouped into 13 steps of 12 k
//     (the half-empty taps of the zero row / column share MFMAs): 39 MFMAs per tile.
//   B operand (weights): the wave's 16 channels x 156 k = 39 values per lane, loaded once, kept in registers.
//   A operand: the four space-to-depth rows a conv row reads sit in an LDS ring of six (LDS-DMA, whole rows of 5 376 B
//     behind a two-pixel zero margin; rows outside the image arrive as zeros from the descriptor's range check, so the
//     tap loop has no masks).  Lane (pixel m, k-group g) reads channels 3g .. 3g+2 of pixel m + tw of row y + th - 2:
//     three 4-byte LDS reads per tap and tile, placed BETWEEN the MFMAs (profiles/r04_experiments.txt: in front of them
//     they would stop the wave's MFMA stream); the 64 lanes of a read hit 64 different banks (pixels 48 B apart).
//   Pooling in registers: a lane's accumulator is 4 consecutive pixels of one channel; the horizontal 3-window takes one
//     neighbour value by ds_bpermute, the vertical one carries two partial rows in registers.  The pooled row leaves as
//     64-byte pieces (16 channels of a pixel) per lane group.
// The k order inside a pixel's fmaf chain is (step, j = 0..2, k-group g = 0..3), steps as listed at the weights -- fixed, so a
// frame's bits do not depend on its batch or position; it is NOT the tile kernel's order (channel-ascending inside a tap).
#include "conv_igemm.h"

namespace pr {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __attribute__((address_space(3))) void lds_void;

constexpr int kH = 112, kC = 12, kHP = 56;
constexpr int kBand = 7;                       // pooled rows per workgroup
constexpr int kRowBytes = kH * kC * 4;         // 5 376 B of one space-to-depth row
constexpr int kMargin = 2 * kC * 4;            // two zero pixels in front of the row (columns -2, -1)
constexpr int kSlot = 6400;                    // >= kMargin + 6 x 1 KB DMA pieces (the sixth runs past the row: zeros = the right margin)
constexpr int kRing = 6;
constexpr int kLds = kRing * kSlot;
constexpr unsigned kOOB = 0x80000000u;

struct SArgs {
  const float* x;      // [B][112][112][12]
  const float* w;      // packed [64][192], k = (th * 4 + tw) * 12 + c
  const float* bias;   // [64]
  float* y;            // [B][56][56][64]
  unsigned x_bytes;
  int B;
  int exp;             // timing builds only (POSERISK_STEM_EXP): 1 no fragment reads, 2 no pooling / stores, 4 no LDS-DMA after the first rows
};
#ifdef PR_TIMING_HOOKS
#define PR_ST_EXP(bit) (a.exp & (bit))
#else
#define PR_ST_EXP(bit) 0
#endif

__global__ __launch_bounds__(256) void stem_pool_f32(const SArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int img = blockIdx.x / (kHP / kBand), band = blockIdx.x % (kHP / kBand);
  const int r0 = band * kBand;                 // first pooled row
  const int y0 = 2 * r0 - 1;                   // first conv row (-1 for the top band: outside, contributes nothing)
  const int m = lane & 15, g = lane >> 4;

  // zero margins of every ring slot once (the DMA never writes the left one; the right one it rewrites with zeros)
  for (int i = tid; i < kRing * (kMargin / 4); i += 256) {
    const int slot = i / (kMargin / 4), o = i % (kMargin / 4);
    reinterpret_cast<float*>(smem + slot * kSlot)[o] = 0.f;
  }

  // ---- weights of this wave's 16 channels, K regrouped: 13 steps x 3 MFMAs instead of 16 taps x 3 ------------------
  // The 7x7 kernel sits in the 8x8 window of the 4x4 taps with a zero row in front and a zero column in front: of tap row
  // th = 0 only the sub-pixels di = 1 (k-groups 2, 3) carry weights, of tap column tw = 0 only dj = 1 (k-groups 1, 3).  An
  // MFMA sums over the four k-groups g, so a half-empty tap wastes half of it; instead
  //   steps 0 .. 8   the nine taps th, tw >= 1:   lane g = sub-pixel g of the tap (as before);
  //   steps 9, 10    tap row 0, taps (0, 2P), (0, 2P + 1):  lane g = sub-pixel 2 + (g & 1) of tap column 2P + (g >> 1);
  //   steps 11, 12   tap column 0: rows 1 and 2 in one step (lane g = sub-pixel 2 (g & 1) + 1 of tap row 1 + (g >> 1)),
  //                  row 3 alone (g >> 1 = 1 multiplies zeros);
  // 39 MFMAs per pixel tile instead of 48 (23 % of the 192 k were zeros).  b[step][j] is the weight of colour j.
  constexpr int kSteps = 13;
  float b[kSteps][3];
  {
    const float* wr = a.w + (size_t)(16 * wave + m) * 192;
#pragma unroll
    for (int st = 0; st < kSteps; ++st) {
      int tap, gg;
      bool live = true;
      if (st < 9) { tap = (1 + st / 3) * 4 + 1 + st % 3; gg = g; }
      else if (st < 11) { tap = 2 * (st - 9) + (g >> 1); gg = 2 + (g & 1); }
      else if (st == 11) { tap = (1 + (g >> 1)) * 4; gg = 2 * (g & 1) + 1; }
      else { tap = 3 * 4; gg = 2 * (g & 1) + 1; live = (g >> 1) == 0; }
#pragma unroll
      for (int j = 0; j < 3; ++j) b[st][j] = live ? wr[tap * 12 + 3 * gg + j] : 0.f;
    }
  }
  const float bias = a.bias[16 * wave + m];

  // ---- input rows by LDS-DMA: streamed row s (s = 0 .. 2 * kBand + 3) is image row y0 - 2 + s, slot s % 6 ------------
  const auto xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  constexpr int kRows = 2 * kBand + 1 + 3;     // 15 conv rows read 18 input rows
  auto issue_row = [&](int s) {                // pieces w and w + 4 of the row's six (wave-uniform)
    const int yy = y0 - 2 + s;
    const bool in = (unsigned)yy < (unsigned)kH;
    const unsigned base = in ? (unsigned)(((img * kH + yy) * kH) * kC * 4) : kOOB;
    char* dst = smem + (s % kRing) * kSlot + kMargin;
#pragma unroll
    for (int p = wave; p < 6; p += 4) {
      const unsigned off = (unsigned)(p * 1024 + lane * 16);
      // bytes past the row's end must read as zeros too: out of range by the vector offset
      const unsigned vo = (in && off < (unsigned)kRowBytes) ? base + off : kOOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, (lds_void*)(dst + p * 1024), 16, vo, 0, 0, 0);
    }
  };
#pragma unroll
  for (int s = 0; s < kRing; ++s) issue_row(s);

  // lane part of the A address: pixel m of a tile, channels 3g ..; tap (th, tw) adds slot(th) * kSlot + tw * 48,
  // tile T adds T * 16 * 48 (immediate)
  const int a_pix = kMargin - 2 * kC * 4 + m * (kC * 4);            // pixel p sits at byte kMargin + p * 48; tap column = p + tw - 2
  const int a_lane = a_pix + g * 12;                                 // steps 0 .. 8: sub-pixel g
  const int a_top = a_pix + (g >> 1) * (kC * 4) + (2 + (g & 1)) * 12;   // steps 9, 10: tap column 2P + (g >> 1), sub-pixel 2 + (g & 1)
  const int a_left = a_pix + (2 * (g & 1) + 1) * 12;                 // steps 11, 12: sub-pixel 2 (g & 1) + 1 (+ the row, below)

  float carry1[14], carry2[14];                // horizontally pooled rows: 2r-1 (carry1), max(2r-1, 2r) (carry2)
#pragma unroll
  for (int i = 0; i < 14; ++i) carry1[i] = carry2[i] = 0.f;

  for (int i = 0; i < 2 * kBand + 1; ++i) {    // conv row y = y0 + i reads streamed rows i .. i + 3
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // (lgkmcnt: the margins' zeros, first iteration)
    __builtin_amdgcn_s_barrier();              // rows <= i + 5 issued, rows <= i + 3 landed long ago; conv row i - 1 is finished
    asm volatile("" ::: "memory");
    if (i > 0 && i + 5 < kRows && !PR_ST_EXP(4)) issue_row(i + 5);        // into the slot of row i - 1
    const int y = y0 + i;
    float h[14];
    if ((unsigned)y < (unsigned)kH) {
      f32x4 acc[7];
#pragma unroll
      for (int t = 0; t < 7; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      float av[2][7][3];
      // lane address of a step's fragments (tile 0, colour 0)
      const int row12 = (g >> 1) ? ((i + 2) % kRing) * kSlot : ((i + 1) % kRing) * kSlot;   // step 11: tap row 1 or 2 by lane
      auto step_ptr = [&](int st) -> const char* {
        if (st < 9) return smem + ((i + 1 + st / 3) % kRing) * kSlot + (1 + st % 3) * (kC * 4) + a_lane;
        if (st < 11) return smem + (i % kRing) * kSlot + 2 * (st - 9) * (kC * 4) + a_top;
        if (st == 11) return smem + row12 + a_left;
        return smem + ((i + 3) % kRing) * kSlot + a_left;
      };
      {
        const char* p0 = step_ptr(0);
#pragma unroll
        for (int t = 0; t < 7; ++t)
#pragma unroll
          for (int j = 0; j < 3; ++j) av[0][t][j] = *reinterpret_cast<const float*>(p0 + t * 16 * kC * 4 + j * 4);
      }
#pragma unroll
      for (int st = 0; st < kSteps; ++st) {
        const int cur = st & 1;
        const char* pn = step_ptr(st + 1 < kSteps ? st + 1 : st);
        // 21 MFMAs (j-major: consecutive ones hit different accumulators) with the next step's 21 reads between them
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int t = 0; t < 7; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur][t][j], b[st][j], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (st + 1 < kSteps && !PR_ST_EXP(1)) {
              av[cur ^ 1][t][j] = *reinterpret_cast<const float*>(pn + t * 16 * kC * 4 + j * 4);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
      }
      // bias + ReLU, then the horizontal window: pooled column q = 2u + e, u = 4t + g, covers conv columns 2q-1 .. 2q+1
      float last[8];                           // last[t + 1] = this lane's pixel 3 of tile t; last[0] = 0 (column -1)
      last[0] = 0.f;
#pragma unroll
      for (int t = 0; t < 7; ++t) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t][e] = fmaxf(acc[t][e] + bias, 0.f);
        last[t + 1] = acc[t][3];
      }
#pragma unroll
      for (int t = 0; t < 7; ++t) {
        // conv column 4u - 1: pixel 3 of lane group g - 1 in this tile, or of group 3 in the tile before (g = 0)
        const int src = ((lane - 16) & 63) << 2;
        const float same = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, last[t + 1])));
        const float prev = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, last[t])));
        const float left = g == 0 ? prev : same;
        h[2 * t] = fmaxf(fmaxf(left, acc[t][0]), acc[t][1]);
        h[2 * t + 1] = fmaxf(fmaxf(acc[t][1], acc[t][2]), acc[t][3]);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 14; ++q) h[q] = 0.f;
    }
    // vertical window: even i = conv row 2r - 1 of pooled row r = r0 + i / 2 and (i > 0) conv row 2r' + 1 of r' = r - 1
    if (PR_ST_EXP(2)) {
      asm volatile("" :: "v"(h[0]), "v"(h[5]), "v"(h[13]));
    } else if ((i & 1) == 0) {
      if (i > 0) {
        const int r = r0 + i / 2 - 1;
        float* yr = a.y + (((size_t)img * kHP + r) * kHP) * 64 + 16 * wave + m;
#pragma unroll
        for (int t = 0; t < 7; ++t)
#pragma unroll
          for (int e = 0; e < 2; ++e) yr[(size_t)(8 * t + 2 * g + e) * 64] = fmaxf(carry2[2 * t + e], h[2 * t + e]);
      }
#pragma unroll
      for (int q = 0; q < 14; ++q) carry1[q] = h[q];
    } else {
#pragma unroll
      for (int q = 0; q < 14; ++q) carry2[q] = fmaxf(carry1[q], h[q]);
    }
  }
#endif
}

}  // namespace

int stem_pool_f32_launch(const float* x_s2d, const float* w, const float* bias, float* y, int B, hipStream_t stream) {
  PR_REQUIRE(x_s2d && w && bias && y && B >= 0, "stem_pool_f32: null argument");
  const size_t xb = (size_t)B * kH * kH * kC * 4;
  PR_REQUIRE(xb < (1ull << 31), "stem_pool_f32: %d frames are too many for one launch", B);
  if (B == 0) return PR_OK;
  SArgs a;
  a.x = x_s2d; a.w = w; a.bias = bias; a.y = y; a.x_bytes = (unsigned)xb; a.B = B;
  a.exp = 0;
#ifdef PR_TIMING_HOOKS
  if (const char* e = getenv("POSERISK_STEM_EXP")) a.exp = atoi(e);
#endif
  static std::atomic<uint64_t> attr_done{0};
  PR_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(stem_pool_f32), kLds, attr_done));
  hipLaunchKernelGGL(stem_pool_f32, dim3(B * (kHP / kBand)), dim3(256), kLds, stream, a);
  return check_launch("stem_pool_f32");
}

}  // namespace pr
