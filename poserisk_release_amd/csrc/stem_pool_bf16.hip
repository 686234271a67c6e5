// The bf16 encoder's stem in ONE kernel: conv1 (7x7 / stride 2 as a 4x4 / stride-1 convolution on the 2x2 space-to-depth
// image, hmr.hip) + folded BatchNorm + ReLU + MaxPool2d(3, 2, 1)   (SPIN models/hmr.py: conv1, bn1, relu, maxpool; call
// site lib/core/base.py:220).  As two launches the 112x112x64 map is written (411 MB at B=256) and read back by the
// pool; here it lives in an LDS ring of six rows and only the 56x56 pooled map leaves the chip.
//
//   * A workgroup (16 waves, four per SIMD: the LDS and MFMA latencies hide behind each other's work) owns a band of pooled rows of one image (the whole image at B >= 256, a quarter at B <= 64):
//     conv rows 2 py0 - 1 .. , computed one
//     row pair per iteration; wave (pt, ct, row) computes pixels 32 pt .. 32 pt + 31 x channels 32 ct .. 32 ct + 31 of one row:
//     16 MFMAs (one per tap: 16 channels = one k-step), the wave's weight fragments resident in 64 VGPRs.
//   * Transposed MFMAs (weights = first operand, weight-fragment lane i reads row sigma(i)): a lane is a PIXEL and holds
//     16 consecutive channels, written to the conv-row ring as two 16-byte chunks.
//   * The input rows stream through a 16-row LDS ring by LDS-DMA, 12+ rows ahead.  A ring row holds the two 16-byte
//     halves of a pixel in two planes with pixel p at slot p + 2: rows outside the image and the two pixels left of it
//     arrive as zeros from the DMA's range check, so the tap loop has NO masks -- tap (th, tw) of pixel x reads slot
//     x + tw of row y + th - 2.
//   * The kernel is bound by the LDS pipe (per row pair 256 KB of tap reads, 32 KB of conv-ring writes, 64 KB of pool
//     reads against 2 k cycles of MFMAs), so every 16-byte access is laid out for the LDS's lane groups (guide: ds_read_b128
//     4 x 16 lanes over 64 banks, ds_write_b128 8 x 8 lanes over 32): tap reads walk a plane (16 consecutive slots = all banks),
//     conv-ring pixels are 144 bytes apart (8 consecutive pixels = 8 different bank groups; 128-byte pixels: an 8-way
//     conflict on every write), the pool's threads are dealt so that a lane group reads two whole pixels 1152 bytes apart.
//     SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE was 0.50 before (scripts/pmc_lds_all.sh).
//   * Pooling: 448 threads take (pooled pixel, 8 channels) each: nine 16-byte reads from the ring, v_pk_max_i16 (the map is
//     non-negative after the ReLU: bf16 order = signed 16-bit order, frame_kernels.hip), one coalesced 16-byte store.
//   * One barrier per iteration (six conv-row slots: the rows being written never alias the rows being pooled).
// Same products in the same k order as conv_dma_bf16's 4x4-tap path (tap-major, 16 channels per MFMA), so the fused
// result equals the two launches bit for bit.
#include <algorithm>

#include "conv_igemm.h"

namespace pr {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using i16x8 = __attribute__((ext_vector_type(8))) short;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
typedef __attribute__((address_space(3))) void lds_void;

[[maybe_unused]] constexpr unsigned kOOB = 0x80000000u;
constexpr int kInRows = 16;                      // input-row ring
constexpr int kInRow = 4096;                     // two planes of 128 pixel slots x 16 bytes
constexpr int kCvRows = 6;                       // conv-row ring
constexpr int kCvPix = 144;                      // a conv-ring pixel: 64 channels bf16 + 16 bytes of padding (conflict-free 16-byte writes)
constexpr int kCvRow = 112 * kCvPix;
constexpr int kOffIn = 0;
constexpr int kOffCv = kOffIn + kInRows * kInRow;
constexpr int kOffBias = kOffCv + kCvRows * kCvRow;
constexpr int kStemLds = kOffBias + 256;
static_assert(kStemLds <= 160 * 1024, "LDS budget");

struct SPArgs {
  const unsigned short* x;    // [B][H][H][16] bf16 (space-to-depth image)
  const unsigned short* w;    // [64][256] bf16, k = (th * 4 + tw) * 16 + c
  const float* bias;          // [64]
  unsigned short* y;          // [B][H/2][H/2][64] bf16
  unsigned x_bytes;
  int B, H, bands, band_rows;   // workgroups per image, pooled rows per workgroup
};

__device__ inline unsigned pack2(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}

__global__ __launch_bounds__(1024) void stem_pool_bf16(const SPArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int H = a.H, HP = H >> 1;
  const int img = blockIdx.x / a.bands, band = blockIdx.x - img * a.bands;
  const int py0 = band * a.band_rows;
  const int npy = min(a.band_rows, HP - py0);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, h = lane >> 5;
  const int pt = wave & 3, ct = (wave >> 2) & 1, rsel = wave >> 3;   // pixel tile, channel tile, row of the iteration's pair
  const int px = 32 * pt + i;                      // the lane's pixel in a conv row (>= H: computed, never kept)

  // weight fragments of this wave's channel tile: MFMA row i <-> channel 32 ct + sigma(i), so that register r of lane half
  // h is channel 32 ct + 16 h + r; tap ks = th * 4 + tw is one k-step (16 channels), lane half h its channels 8 h ..
  const int wrow = 32 * ct + 16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3);
  bf16x8 wf[16];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) wf[ks] = *reinterpret_cast<const bf16x8*>(a.w + wrow * 256 + 16 * ks + 8 * h);
  if (tid < 64) *reinterpret_cast<float*>(smem + kOffBias + tid * 4) = a.bias[tid];

  // ---- input ring: row r of the image -> slot r & 15; one row = four 1 KB DMA pieces (plane, 64-slot half) ---------------
  const auto xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  auto issue_piece = [&](int r, int piece) {        // piece = 2 * plane + half, wave-uniform
    const int plane = piece >> 1, slot = 64 * (piece & 1) + lane, p = slot - 2;
    const bool ok = (unsigned)r < (unsigned)H && (unsigned)p < (unsigned)H;
    const unsigned voff = ok ? (unsigned)((((img * H + r) * H + p) * 16 + 8 * plane) * 2) : kOOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, (lds_void*)(smem + kOffIn + (r & (kInRows - 1)) * kInRow + piece * 1024), 16,
                                             voff, 0, 0, 0);
  };
  const int r_first = 2 * py0 - 3;                 // first input row the band touches (conv row 2 py0 - 1 reads y - 2 ..)
  // prologue: 15 rows = 60 pieces over the 16 waves
  for (int q = wave; q < 60; q += 16) issue_piece(r_first + (q >> 2), q & 3);

  const float* bp = reinterpret_cast<const float*>(smem + kOffBias) + 32 * ct + 16 * h;

  // conv row y (0 <= y < H) -> conv ring slot y % 6
  auto conv_row = [&](int y) {
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    // four waves per SIMD hide the LDS latency: one tap row (4 fragments) at a time
#pragma unroll
    for (int th = 0; th < 4; ++th) {
      // lane (i, h) reads slot px + tw of plane h: the 16 lanes of a ds_read_b128 group read 16 different 16-byte slots of
      // one plane = all 64 banks once
      const char* row = smem + kOffIn + ((y + th - 2) & (kInRows - 1)) * kInRow + h * 2048 + px * 16;
      bf16x8 bf[4];
#pragma unroll
      for (int tw = 0; tw < 4; ++tw) bf[tw] = *reinterpret_cast<const bf16x8*>(row + tw * 16);
#pragma unroll
      for (int tw = 0; tw < 4; ++tw) acc = mfma_bf16_step(wf[4 * th + tw], bf[tw], acc, tw);
    }
    unsigned pk[8];
#pragma unroll
    for (int e = 0; e < 8; ++e)
      pk[e] = pack2(fmaxf(acc[2 * e] + bp[2 * e], 0.f), fmaxf(acc[2 * e + 1] + bp[2 * e + 1], 0.f));
    if (px < H) {
      // 144-byte pixels: the eight lanes of a ds_write_b128 group (consecutive pixels) write eight different 16-byte bank groups
      // (with 128-byte pixels all eight met on the same banks)
      char* dst = smem + kOffCv + (y % kCvRows) * kCvRow + px * kCvPix + 64 * ct + 32 * h;
      *reinterpret_cast<u32x4*>(dst) = u32x4{pk[0], pk[1], pk[2], pk[3]};
      *reinterpret_cast<u32x4*>(dst + 16) = u32x4{pk[4], pk[5], pk[6], pk[7]};
    }
  };

  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (2 * py0 - 1 >= 0) {                           // the band's halo row (the pool's row above its first pair)
    if (rsel == 0) conv_row(2 * py0 - 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                   // its input rows are about to be refilled
    asm volatile("" ::: "memory");
  }

  // pooling: (pooled pixel, 16-byte chunk slot) per thread, eight pixels per wave, dealt so that each 16-lane group of a
  // ds_read_b128 ({0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32) reads ALL chunks of two pixels four pooled
  // pixels apart: 2 x 128 contiguous bytes 1152 = 4.5 x 256 bytes apart = every bank once
  const int quad = (lane >> 2) & 7, gpos = 4 * (quad >> 1) + (lane & 3);
  const int ggid = 2 * (lane >> 5) + (__builtin_popcount(quad) & 1);
  const int pp = 8 * wave + ggid + 4 * (gpos >> 3), c8 = gpos & 7;
  for (int it = 0; it < npy; ++it) {
    const int py = py0 + it;
    // Refill the two ring rows that conv(it - 1) was the last to read (needed five iterations from now): waves 0-7, ALWAYS
    // one piece per wave and iteration, also past the band's last row (zeros or the next rows, a few KB) -- the counted
    // wait below relies on this fixed pattern.
    if (wave < 8) issue_piece(2 * py + 12 + (wave >> 2), wave & 3);
    conv_row(2 * py + rsel);
    // The rows of the NEXT iteration's conv must have landed: they were issued at least four iterations ago (or in the
    // prologue, drained above); vector-memory operations retire in issue order, so a wave's three youngest refills and
    // three youngest pooled stores may stay in flight (waves 0-6 issue both, wave 7 refills only, waves 8-15 neither).
    if (wave < 7) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    else if (wave == 7) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (pp < HP) {
      const short lowest = (short)-32768;           // -0: below every value of the (non-negative) map
      i16x8 m = {lowest, lowest, lowest, lowest, lowest, lowest, lowest, lowest};
#pragma unroll
      for (int dr = -1; dr <= 1; ++dr) {
        const int r = 2 * py + dr;
        if (r < 0) continue;                        // (r <= H - 1 always: H is even)
        const char* row = smem + kOffCv + (r % kCvRows) * kCvRow + c8 * 16;
#pragma unroll
        for (int dc = -1; dc <= 1; ++dc) {
          const int q = 2 * pp + dc;
          if (q >= 0) m = __builtin_elementwise_max(m, *reinterpret_cast<const i16x8*>(row + q * kCvPix));
        }
      }
      *reinterpret_cast<i16x8*>(a.y + (((long)img * HP + py) * HP + pp) * 64 + c8 * 8) = m;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may still be in flight when the workgroup's LDS is released
#endif
}

}  // namespace

int stem_pool_bf16_launch(const void* x_s2d, const void* w, const float* bias, void* y, int B, int H, hipStream_t stream) {
  PR_REQUIRE(x_s2d && w && bias && y, "stem_pool: null argument");
  PR_REQUIRE(H >= 2 && H <= 112 && H % 2 == 0 && B >= 0, "stem_pool: map %dx%d unsupported (even, 2..112)", H, H);
  const size_t xb = (size_t)B * H * H * 32;
  PR_REQUIRE(xb < (1ull << 31), "stem_pool: tensor too large for one launch");
  if (B == 0) return PR_OK;
  SPArgs a;
  a.x = reinterpret_cast<const unsigned short*>(x_s2d); a.w = reinterpret_cast<const unsigned short*>(w);
  a.bias = bias; a.y = reinterpret_cast<unsigned short*>(y);
  a.x_bytes = (unsigned)xb; a.B = B; a.H = H;
  // One workgroup per CU and as few workgroups per image as still fill the chip: a band costs a prologue (weights, 15 input
  // rows) and one recomputed halo row, so B >= 256 runs whole images, smaller batches up to four bands per image.
  int cus = 256;
  PR_TRY(current_device_cus(&cus));
  const int HP = H / 2;
  int bands = std::min(std::max(1, ceil_div(std::max(cus, 1), B)), std::min(4, HP));
  a.band_rows = ceil_div(HP, bands);
  a.bands = ceil_div(HP, a.band_rows);
  static std::atomic<uint64_t> attr_done{0};
  PR_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(stem_pool_bf16), kStemLds, attr_done));
  hipLaunchKernelGGL(stem_pool_bf16, dim3(B * a.bands), dim3(1024), kStemLds, stream, a);
  return check_launch("stem_pool_bf16");
}

}  // namespace pr
