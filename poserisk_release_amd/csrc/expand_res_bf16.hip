// A Bottleneck's conv3 (1x1 expansion) + folded BatchNorm + residual + ReLU as a persistent bf16 kernel with the WEIGHTS IN
// REGISTERS:   y[M][N] = relu(t[M][K] . W[N][K]^T + bias[N] + res[M][N])      (SPIN models/hmr.py Bottleneck.forward:
// out = relu(bn3(conv3(out)) + identity); call site lib/core/base.py:220).  layer2: K = 128, N = 512; layer3: K = 256,
// N = 1024 as two column blocks of 512 (two workgroups read the same t); and layer2's FIRST conv3 with its downsample branch
// as a second, strided pixel source in the same K loop (K = 128 + 256, N = 512, no residual).
//
// On the tile kernel this layer is all epilogue: two K-steps per 128x64 tile, then an LDS transpose, two barriers and a
// residual read that only starts once the tile is done (3.6 TB/s).  Here a workgroup of eight waves walks a contiguous run
// of 64-pixel blocks; wave w keeps the W rows of its 64 output channels as MFMA A fragments for the whole kernel (K/16 x 2
// tiles x 4 = 64 VGPRs, 128 for layer3: no weight traffic at all), t streams through an LDS ring by LDS-DMA three blocks
// ahead, the residual of the next block is requested into registers as the stores free them, and y leaves straight from the
// accumulators (transposed MFMAs as in bottleneck_bf16.hip: a lane is a pixel holding 16 consecutive channels; lane i of a
// weight fragment reads row sigma(i)).  One barrier per block.  Same products in the same k order and the same epilogue
// arithmetic ((acc + bias) + res) as conv_dma_bf16: bit-identical.
#include <algorithm>

#include "conv_igemm.h"

namespace pr {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
typedef __attribute__((address_space(3))) void lds_void;

[[maybe_unused]] constexpr unsigned kOOB = 0x80000000u;

struct ExArgs {
  const unsigned short* t;     // [M][K1]
  const unsigned short* w;     // [N][K1 + K2] (conv_pack_weights_bf16 layout, rows in channel order, the two matrices side by side)
  const float* bias;           // [N]
  const unsigned short* res;   // [M][N] or unused
  unsigned short* y;           // [M][N]
  unsigned t_bytes, y_bytes;
  int M, nblocks, relu;
  // second source (a first block's downsample branch summed in conv3's K loop): x2 [B][H2][W2][K2], output pixel (img, ho,
  // wo) reads x2 (img, ho * stride2, wo * stride2)
  const unsigned short* x2;
  unsigned x2_bytes;
  int HoWo, Wo, H2, W2, stride2;
};

__device__ inline unsigned pack2(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}

// KS1 / KS2 = k-steps (16 channels) of t / of the second source; TPW = tiles of 32 channels per wave, a workgroup covers
// 256 TPW output channels; NS = column blocks: N = 256 TPW NS, workgroup w takes column block (w >> 3) % NS of pixel-block
// run (w / (8 NS)) * 8 + (w & 7) -- the NS workgroups that read the same pixels sit 8 apart, on one XCD.  RES: + residual.
template <int KS1, int KS2, int TPW, int NS, bool RES>
__global__ __launch_bounds__(512) void expand_res_bf16(const ExArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int KS = KS1 + KS2, K1 = 16 * KS1, K2 = 16 * KS2, K = K1 + K2, NW = 256 * TPW, N = NW * NS;
  constexpr int SL1 = K1 / 64, SL = K / 64;        // 64-channel slices (8 KB in LDS) per block: SL1 of t, the rest of x2
  constexpr int BLK = SL * 8192;
  constexpr int kSlots = BLK <= 32768 ? 4 : 3;     // t ring: blocks of 64 pixels
  constexpr int kAhead = kSlots - 1;               // blocks in flight ahead of the one being computed
  constexpr int KH = 8;                            // k-steps whose pixel fragments are in registers at a time
  static_assert(KS % KH == 0 && K1 % 64 == 0 && K2 % 64 == 0, "whole slices, whole fragment groups");
  constexpr int NONDMA = (RES ? 8 : 4) * TPW;      // per block and wave: 4 TPW stores (+ 4 TPW residual loads)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int cbk = NS == 1 ? 0 : (int)(blockIdx.x >> 3) % NS;
  const unsigned run = NS == 1 ? blockIdx.x : (blockIdx.x / (8 * NS)) * 8 + (blockIdx.x & 7), runs = gridDim.x / NS;
  const int b0 = (int)(run * (unsigned)a.nblocks / runs);
  const int b1 = (int)((run + 1u) * (unsigned)a.nblocks / runs);
  if (b0 >= b1) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, h = lane >> 5;
  float* lbias = reinterpret_cast<float*>(smem + kSlots * BLK);
  for (int c = tid; c < NW; c += 512) lbias[c] = a.bias[NW * cbk + c];

  // W rows of this wave's tiles: MFMA row i <-> channel NW cbk + 32 (TPW wave + n) + sigma(i)
  const int wrow = 16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3);
  bf16x8 wf[TPW][KS];
#pragma unroll
  for (int n = 0; n < TPW; ++n)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      wf[n][ks] = *reinterpret_cast<const bf16x8*>(a.w + (NW * cbk + 32 * (TPW * wave + n) + wrow) * K + 16 * ks + 8 * h);

  // pixel ring: block b -> slot (b - b0) % kSlots, SL slices of [64 pixels][128 B], 16-byte chunks XOR-swizzled on the source
  // side; a slice is eight 1 KB DMA groups of 8 pixels, wave w issues group w of every slice
  const auto tsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.t), 0, (int)a.t_bytes, 0x00020000);
  [[maybe_unused]] const auto xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(KS2 ? a.x2 : a.t), 0,
                                                                         KS2 ? (int)a.x2_bytes : 0, 0x00020000);
  [[maybe_unused]] const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(RES ? a.res : a.t), 0,
                                                                         RES ? (int)a.y_bytes : 0, 0x00020000);
  const auto ysrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)a.y_bytes, 0x00020000);
  const int dq = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
  auto issue_block = [&](int b) {                  // ALWAYS SL pieces
    const int m = b * 64 + 8 * wave + (lane >> 3);
    const bool ok = b < b1 && m < a.M;
    const unsigned voff = ok ? (unsigned)(m * (2 * K1) + dq * 16) : kOOB;
    char* slot = smem + ((b - b0) % kSlots) * BLK;
#pragma unroll
    for (int s = 0; s < SL1; ++s)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(tsrc, (lds_void*)(slot + s * 8192 + wave * 1024), 16, voff, s * 128, 0, 0);
    if constexpr (KS2 > 0) {
      unsigned voff2 = kOOB;
      if (ok) {
        const int img = m / a.HoWo, rem = m - img * a.HoWo;
        const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
        voff2 = (unsigned)((((img * a.H2 + ho * a.stride2) * a.W2 + wo * a.stride2) * K2) * 2 + dq * 16);
      }
#pragma unroll
      for (int s = SL1; s < SL; ++s)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, (lds_void*)(slot + s * 8192 + wave * 1024), 16, voff2, (s - SL1) * 128, 0, 0);
    }
  };
  // residual of (block b, pixel tile pt) in the epilogue's layout: per tile n two 16-byte pieces per lane.  ALWAYS 2 TPW
  // loads (rows >= M and blocks >= b1 read as zero through the range check): the counted wait below relies on it.
  const int csoff = (NW * cbk + 32 * TPW * wave) * 2;       // byte offset of this wave's first channel in a row of y / res
  [[maybe_unused]] u32x4 rr[2][TPW][2];
  auto load_res = [&](int b, int pt) {
    if constexpr (RES) {
      const int m = b * 64 + 32 * pt + i;
      const unsigned voff = (b < b1 && m < a.M) ? (unsigned)(m * (2 * N) + 32 * h) : kOOB;
#pragma unroll
      for (int n = 0; n < TPW; ++n) {
        rr[pt][n][0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, csoff + 64 * n, 0);
        rr[pt][n][1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + 16, csoff + 64 * n, 0);
      }
    }
  };

  int pfoff[2][4];
#pragma unroll
  for (int pt = 0; pt < 2; ++pt)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int prow = 32 * pt + i;
      pfoff[pt][kk] = prow * 128 + (((2 * kk + h) ^ ((prow >> 1) & 7)) << 4);
    }

  for (int d = 0; d < kAhead; ++d) issue_block(b0 + d);
  load_res(b0, 0);
  load_res(b0, 1);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // once: the first blocks, the first residual, the bias
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // One block.  Everything the wave issues per block is fixed -- SL DMA pieces FIRST, then per pixel tile 2 TPW stores and
  // (RES) the 2 TPW loads of the NEXT block's residual for that tile, into the registers the stores have just freed: a block
  // of lead, with one buffer -- so the wait for this block's DMA pieces leaves a counted number of younger operations in flight.
  for (int b = b0; b < b1; ++b) {
    if (b > b0) {
      // younger than block b's DMA pieces (issued kAhead blocks ago, in front of that iteration's loads and stores): that
      // iteration's other operations and everything of the (kAhead - 1) iterations since
      constexpr int younger = NONDMA + (kAhead - 1) * (SL + NONDMA);
      static_assert(younger <= 63, "vmcnt range");
      if (b - b0 >= kAhead) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(younger) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();    // block b has landed for everyone; everyone is done reading block b - 1
      asm volatile("" ::: "memory");
    }
    issue_block(b + kAhead);           // into the slot of block b - 1
    asm volatile("" ::: "memory");     // the DMA pieces stay the iteration's FIRST vector-memory operations (the count above)
    const char* slot = smem + ((b - b0) % kSlots) * BLK;
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      f32x16 acc[TPW];
#pragma unroll
      for (int n = 0; n < TPW; ++n)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
#pragma unroll
      for (int k0 = 0; k0 < KS; k0 += KH) {
        bf16x8 tf[KH];
#pragma unroll
        for (int ks = 0; ks < KH; ++ks)
          tf[ks] = *reinterpret_cast<const bf16x8*>(slot + ((k0 + ks) >> 2) * 8192 + pfoff[pt][(k0 + ks) & 3]);
#pragma unroll
        for (int n = 0; n < TPW; ++n)
#pragma unroll
          for (int ks = 0; ks < KH; ++ks) acc[n] = mfma_bf16_step(wf[n][k0 + ks], tf[ks], acc[n], ks);
      }
      const int m = b * 64 + 32 * pt + i;
      const unsigned yoff = m < a.M ? (unsigned)(m * (2 * N) + 32 * h) : kOOB;
#pragma unroll
      for (int n = 0; n < TPW; ++n) {
        const float* bp = lbias + 32 * (TPW * wave + n) + 16 * h;
        unsigned pk[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float v0 = acc[n][2 * e] + bp[2 * e], v1 = acc[n][2 * e + 1] + bp[2 * e + 1];
          if constexpr (RES) {
            const unsigned r2 = rr[pt][n][e >> 2][e & 3];
            v0 += __uint_as_float(r2 << 16);
            v1 += __uint_as_float(r2 & 0xffff0000u);
          }
          if (a.relu) {
            v0 = fmaxf(v0, 0.f);
            v1 = fmaxf(v1, 0.f);
          }
          pk[e] = pack2(v0, v1);
        }
        buffer_store_b128_sreg(u32x4{pk[0], pk[1], pk[2], pk[3]}, ysrc, yoff, csoff + 64 * n);
        buffer_store_b128_sreg(u32x4{pk[4], pk[5], pk[6], pk[7]}, ysrc, yoff + 16, csoff + 64 * n);
      }
      asm volatile("" ::: "memory");
      load_res(b + 1, pt);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may still be in flight when the workgroup's LDS is released
#endif
}

template <int KS1, int KS2, int TPW, int NS, bool RES>
int launch_expand(const ExArgs& a, int cus, hipStream_t stream) {
  // a multiple of 8 NS workgroups (column-block partners 8 apart), at most one per CU and one run per pixel block
  const int per = NS == 1 ? 1 : 8 * NS;
  int grid = std::min(std::max(cus, 1), a.nblocks * NS) / per * per;
  if (grid == 0) grid = per;
  constexpr int SL = (KS1 + KS2) / 4, BLK = SL * 8192;
  constexpr int lds = (BLK <= 32768 ? 4 : 3) * BLK + 256 * TPW * 4;
  static std::atomic<uint64_t> attr_done{0};
  PR_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(expand_res_bf16<KS1, KS2, TPW, NS, RES>), lds, attr_done));
  hipLaunchKernelGGL((expand_res_bf16<KS1, KS2, TPW, NS, RES>), dim3(grid), dim3(512), lds, stream, a);
  return check_launch("expand_res_bf16");
}

}  // namespace

// (expand_res_bf16_fits / expand_dual_bf16_fits: host_plan.cc, the plan routes by them)

int expand_res_bf16_launch(const void* t, const void* w, const float* bias, const void* res, void* y, long M, int K, int N,
                           int relu, hipStream_t stream) {
  PR_REQUIRE(t && w && bias && res && y, "expand_res: null argument");
  PR_REQUIRE(expand_res_bf16_fits(K, N), "expand_res: K = 128, N = 512 or K = 256, N = 1024 only (got %d, %d)", K, N);
  PR_REQUIRE(M >= 0 && M * 2 * N < (1L << 31), "expand_res: tensor too large for one launch (%ld rows)", M);
  if (M == 0) return PR_OK;
  ExArgs a{};
  a.t = reinterpret_cast<const unsigned short*>(t); a.w = reinterpret_cast<const unsigned short*>(w); a.bias = bias;
  a.res = reinterpret_cast<const unsigned short*>(res); a.y = reinterpret_cast<unsigned short*>(y);
  a.t_bytes = (unsigned)(M * 2 * K); a.y_bytes = (unsigned)(M * 2 * N);
  a.M = (int)M; a.nblocks = (int)ceil_div(M, 64L); a.relu = relu;
  int cus = 256;
  PR_TRY(current_device_cus(&cus));
  return K == 128 ? launch_expand<8, 0, 2, 1, true>(a, cus, stream) : launch_expand<16, 0, 2, 2, true>(a, cus, stream);
}

// A first block's conv3 with its downsample branch in the K loop, no residual: y = act(t . W3^T + x2[::s, ::s] . Wd^T + bias),
// w = [N][K1 + K2] (the two folded matrices side by side).  layer2: K1 = 128, K2 = 256, N = 512 (two column blocks of 256).
int expand_dual_bf16_launch(const void* t, const void* x2, const void* w, const float* bias, void* y, int B, int Ho, int Wo, int H2,
                            int W2, int stride2, int K1, int K2, int N, int relu, hipStream_t stream) {
  PR_REQUIRE(t && x2 && w && bias && y, "expand_dual: null argument");
  PR_REQUIRE(expand_dual_bf16_fits(K1, K2, N), "expand_dual: K = 128 + 256, N = 512 only (got %d + %d, %d)", K1, K2, N);
  PR_REQUIRE(stride2 > 0 && (H2 - 1) / stride2 + 1 == Ho && (W2 - 1) / stride2 + 1 == Wo,
             "expand_dual: second source %dx%d / stride %d does not land on the %dx%d output", H2, W2, stride2, Ho, Wo);
  const long M = (long)B * Ho * Wo;
  const size_t x2b = (size_t)B * H2 * W2 * K2 * 2;
  PR_REQUIRE(M >= 0 && M * 2 * N < (1L << 31) && x2b < (1ull << 31), "expand_dual: tensor too large for one launch (%ld rows)", M);
  if (M == 0) return PR_OK;
  ExArgs a{};
  a.t = reinterpret_cast<const unsigned short*>(t); a.w = reinterpret_cast<const unsigned short*>(w); a.bias = bias;
  a.res = nullptr; a.y = reinterpret_cast<unsigned short*>(y);
  a.t_bytes = (unsigned)(M * 2 * K1); a.y_bytes = (unsigned)(M * 2 * N);
  a.M = (int)M; a.nblocks = (int)ceil_div(M, 64L); a.relu = relu;
  a.x2 = reinterpret_cast<const unsigned short*>(x2); a.x2_bytes = (unsigned)x2b;
  a.HoWo = Ho * Wo; a.Wo = Wo; a.H2 = H2; a.W2 = W2; a.stride2 = stride2;
  int cus = 256;
  PR_TRY(current_device_cus(&cus));
  return launch_expand<8, 16, 1, 2, false>(a, cus, stream);
}

}  // namespace pr
