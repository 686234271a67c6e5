// A Bottleneck's 3x3 convolution and the 1x1 expansion behind it as ONE kernel (fp32):
//
//   t2 = relu(conv3x3(t1) + b2)            64 -> 64 channels, stride 1, pad 1      (SPIN Bottleneck conv2 + bn2 + relu)
//   y  = relu(t2 * W3^T + b3 + x)          64 -> N3 channels + residual            (conv3 + bn3 + add + relu)
//
// for the layer1 blocks of ResNet-50 (call site lib/core/base.py:220), where conv2's 64 output channels are exactly
// one 64x64 tile: the workgroup that has computed a tile of t2 holds ALL the channels its rows need for conv3.  The
// tile goes to LDS in the operand layout (never to HBM) and is multiplied by W3 in N3/64 column chunks on the same
// MFMA loop, W3 streaming through a two-stage LDS ring by LDS-DMA.  Measured motivation (profiles/r02_conv3_sweep_b64.txt):
// as its own launch conv3 has K = 64, i.e. two K-steps per tile, and runs at 58 TFLOP/s however it is tiled -- every
// workgroup of a CU waits for its loads and its stores at the same time; behind an 18-K-step 3x3 main loop those
// phases overlap the other workgroups' MFMAs, and t2 (51 MB per layer at B=64) is neither written nor read back.
//
// Same data path as conv_dma.hip (LDS-DMA staging, source-side XOR swizzle, zero fill by the buffer range check,
// one barrier per K-step); the arithmetic per output element is the same ascending-k fmaf chain as the separate
// kernels', so a frame's bits do not depend on its batch or position, and t2 is rounded to fp32 exactly where the
// separate launches round it (the store to HBM).
#include <cstdlib>

#include "conv_igemm.h"

namespace pr {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __attribute__((address_space(3))) void lds_void;

constexpr int BK = kConvBK;
[[maybe_unused]] constexpr unsigned kOOB = 0x80000000u;

struct FArgs {
  const float* x;      // t1 [B,H,W,Cin]
  const float* w2;     // [64][9*Cin] packed, BN folded
  const float* bias2;  // [64]
  const float* w3;     // [N3][64] packed, BN folded
  const float* bias3;  // [N3]
  const float* res;    // [M][N3] or nullptr
  float* y;            // [M][N3]
  unsigned x_bytes, w2_bytes, w3_bytes;
  int H, W, Cin, log2Cin, HoWo, M, nk, N3, relu3;
};

constexpr int kStage = 2 * 64 * 128;          // one main-loop stage: A 8 KB + B 8 KB
constexpr int kT2 = 0;                        // t2 tile as GEMM2's A operand: K-step 0 at +0, K-step 1 at +8192
constexpr int kRing = 2 * 8192;               // W3 ring: 2 stages of 64 rows x 128 B
constexpr int kCt = 2 * kStage;               // output staging [64][68] floats, behind the main loop's stages
constexpr int kCtStride = 68;
constexpr int kLdsStaged = kCt + 64 * kCtStride * 4;   // 49 KB: 3 workgroups per CU
constexpr int kLdsDirect = 2 * kStage;                  // 32 KB: 5 workgroups per CU, like the plain 64x64 kernel

// DIRECT: GEMM 2's outputs leave straight from the accumulator fragments (per store instruction two 128-byte row
// segments) instead of through an LDS transpose: no staging buffer (32 KB of LDS instead of 49 KB, five resident
// workgroups instead of three) and two barriers fewer per column chunk.
template <bool DIRECT>
__global__ __launch_bounds__(256, DIRECT ? 5 : 3) void conv3x3_conv1x1_f32(const FArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nb = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nb >> 3, rr = nb & 7;
  const int tile_m = (xcd < rr ? xcd * (q8 + 1) : rr * (q8 + 1) + (xcd - rr) * q8) + (bid >> 3);
  const int m0 = tile_m * 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // ---- DMA source addressing (conv_dma.hip): wave w stages row groups w and w + 4, lane = row 8g + (lane>>3),
  // physical chunk lane&7 = logical chunk q
  const int q = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
  const auto xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  const auto w2src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w2), 0, (int)a.w2_bytes, 0x00020000);
  const auto w3src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w3), 0, (int)a.w3_bytes, 0x00020000);
  int a_base[2], a_hi0[2], a_wi0[2];
  unsigned b2_off[2], b3_off[2];
  const int K2 = a.nk * BK;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = 8 * (wave + 4 * i) + (lane >> 3);
    const int m = m0 + r;
    if (m < a.M) {
      const int img = m / a.HoWo, rem = m - img * a.HoWo;
      const int ho = rem / a.W, wo = rem - ho * a.W;
      a_hi0[i] = ho - 1;
      a_wi0[i] = wo - 1;
      a_base[i] = (((img * a.H + a_hi0[i]) * a.W + a_wi0[i]) * a.Cin + q * 4) * 4;
    } else {
      a_hi0[i] = -(1 << 28);
      a_wi0[i] = 0;
      a_base[i] = (int)kOOB;
    }
    b2_off[i] = (unsigned)((r * K2 + q * 4) * 4);
    b3_off[i] = (unsigned)((r * 64 + q * 4) * 4);
  }

  auto issue = [&](int kt, int buf) {
    char* stage = smem + buf * kStage;
    const int k0 = kt * BK;
    const int tap = k0 >> a.log2Cin, ci0 = k0 & (a.Cin - 1);
    const int kh = tap / 3, kw = tap - kh * 3;
    const int koff = ((kh * a.W + kw) * a.Cin + ci0) * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool ok = (unsigned)(a_hi0[i] + kh) < (unsigned)a.H && (unsigned)(a_wi0[i] + kw) < (unsigned)a.W;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, (lds_void*)(stage + (wave + 4 * i) * 1024), 16,
                                               ok ? (unsigned)(a_base[i] + koff) : kOOB, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w2src, (lds_void*)(stage + 8192 + (wave + 4 * i) * 1024), 16, b2_off[i],
                                               kt * 128, 0, 0);
  };
  // W3 step s = 2 * chunk + kstep: rows [64 chunk, 64 chunk + 64) of W3, k in [32 kstep, 32 kstep + 32)
  auto issue3 = [&](int s) {
    char* stage = smem + kRing + (s & 1) * 8192;
    const int soff = ((s >> 1) * 64 * 64 + (s & 1) * BK) * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w3src, (lds_void*)(stage + (wave + 4 * i) * 1024), 16, b3_off[i], soff, 0, 0);
  };

  const int frow = lane & 31, fh = lane >> 5, fsw = (frow >> 1) & 7;
  int foff[BK / 8];
#pragma unroll
  for (int kk = 0; kk < BK / 8; ++kk) foff[kk] = frow * 128 + (((2 * kk + fh) ^ fsw) << 4);

  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  auto compute = [&](const char* Ab, const char* Bb) {
    f32x4 af[BK / 8], bf[BK / 8];
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      af[kk] = *reinterpret_cast<const f32x4*>(Ab + foff[kk]);
      bf[kk] = *reinterpret_cast<const f32x4*>(Bb + foff[kk]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk][j], bf[kk][j], acc, 0, 0, 0);
  };

  // ---- GEMM 1: the 3x3 convolution, K = 9 Cin ------------------------------------------------------------
  issue(0, 0);
  for (int kt = 0; kt < a.nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + 1 < a.nk) issue(kt + 1, (kt + 1) & 1);
    const char* st = smem + (kt & 1) * kStage;
    compute(st + wm * 32 * 128, st + 8192 + wn * 32 * 128);
  }

  // ---- t2 = relu(acc + b2) into LDS as GEMM 2's A operand (the stage layout: 128-byte rows, swizzled chunks) ----
  __syncthreads();      // every wave has finished reading the stages
  issue3(0);            // W3's first stage lands while the tile is written
  {
    const int c = lane & 31;
    const float b2 = a.bias2[wn * 32 + c];
    char* t2 = smem + kT2 + wn * 8192;      // this wave's 32 columns are K-step wn of GEMM 2
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int r = wm * 32 + 4 * (lane >> 5) + (e & 3) + 8 * (e >> 2);
      const float v = fmaxf(acc[e] + b2, 0.f);
      *reinterpret_cast<float*>(t2 + r * 128 + ((((c >> 2) ^ ((r >> 1) & 7))) << 4) + (c & 3) * 4) = v;
    }
  }

  // ---- GEMM 2: N3/64 column chunks of y = t2 * W3^T, two K-steps each ------------------------------------
  float* Ct = reinterpret_cast<float*>(smem + kCt);
  const int nsteps = (a.N3 >> 6) * 2;
  f32x4 rpre[4];
  float rfrag[16];
  const int col_l = lane & 31, row_h = 4 * (lane >> 5);
  // DIRECT: residual and output through range-checked buffer descriptors, one 32-bit lane offset for both
  const int yz_bytes = a.M * a.N3 * 4;
  [[maybe_unused]] const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res ? a.res : a.y), 0, yz_bytes, 0x00020000);
  [[maybe_unused]] const auto ysrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, yz_bytes, 0x00020000);
  [[maybe_unused]] const int frag_off = ((m0 + wm * 32 + row_h) * a.N3 + wn * 32 + col_l) * 4;
  for (int s = 0; s < nsteps; ++s) {
    // lgkmcnt: the t2 tile's ds_writes (s = 0) must have landed before the barrier lets other waves read them
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();      // W3 stage s landed; (s = 0) the t2 tile is complete; stage s-1 is consumed
    asm volatile("" ::: "memory");
    if (s + 1 < nsteps) issue3(s + 1);
    const int n0 = (s >> 1) * 64;
    if ((s & 1) == 0) {
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
      if (DIRECT && a.res) {      // in the fragment layout: acc[e] <-> (row wm*32 + row_h + (e&3) + 8(e>>2), col wn*32 + col_l)
#pragma unroll
        for (int e = 0; e < 16; ++e)    // rows >= M are beyond the descriptor's range: they read as zero
          rfrag[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                   rsrc, frag_off, (n0 + ((e & 3) + 8 * (e >> 2)) * a.N3) * 4, 0));
      }
      if (!DIRECT && a.res) {       // this chunk's residual, requested now, used after the second K-step
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int idx = tid + i * 256;
          const int r = idx >> 4, cc = idx & 15;
          const int row = m0 + r;
          const f32x4 z = {0.f, 0.f, 0.f, 0.f};
          rpre[i] = row < a.M ? *reinterpret_cast<const f32x4*>(a.res + (long)row * a.N3 + n0 + cc * 4) : z;
        }
      }
    }
    compute(smem + kT2 + (s & 1) * 8192 + wm * 32 * 128, smem + kRing + (s & 1) * 8192 + wn * 32 * 128);
    if (DIRECT && (s & 1)) {
      const float b3 = a.bias3[n0 + wn * 32 + col_l];
#pragma unroll
      for (int e = 0; e < 16; ++e) {    // stores to rows >= M fall outside the descriptor's range and are dropped
        float v = acc[e] + b3;
        if (a.res) v += rfrag[e];
        if (a.relu3) v = fmaxf(v, 0.f);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ysrc, frag_off,
                                                  (n0 + ((e & 3) + 8 * (e >> 2)) * a.N3) * 4, 0);
      }
    }
    if (!DIRECT && (s & 1)) {
      // chunk epilogue through LDS: whole 128-byte lines leave the workgroup (conv_dma.hip)
      {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int r = wm * 32 + row_h + (e & 3) + 8 * (e >> 2);
          Ct[r * kCtStride + wn * 32 + col_l] = acc[e];
        }
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int idx = tid + i * 256;
        const int r = idx >> 4, cc = idx & 15;
        const int row = m0 + r, col = n0 + cc * 4;
        if (row >= a.M) continue;
        f32x4 v = *reinterpret_cast<const f32x4*>(&Ct[r * kCtStride + cc * 4]);
        v += *reinterpret_cast<const f32x4*>(a.bias3 + col);
        if (a.res) v += rpre[i];
        if (a.relu3) {
          v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
        }
        *reinterpret_cast<f32x4*>(a.y + (long)row * a.N3 + col) = v;
      }
      // the next chunk's Ct writes come after two more barriers (steps s+1, s+2): every read above is done by then
    }
  }
#endif
}

int ilog2_exact_f(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return (1 << l) == v ? l : -1;
}

}  // namespace

int conv_fused3_launch(const ConvProblem& p, hipStream_t stream) {
  PR_REQUIRE(p.precision == 0, "conv_fused3: fp32 only");
  PR_REQUIRE(p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && p.Cout == 64 && p.Ho == p.H && p.Wo == p.W,
             "conv_fused3: the first convolution must be 3x3 / stride 1 / pad 1 with 64 output channels");
  const int l2 = ilog2_exact_f(p.Cin);
  PR_REQUIRE(l2 >= 0 && p.Cin % BK == 0, "conv_fused3: Cin must be a power of two >= 32 (%d)", p.Cin);
  PR_REQUIRE(p.w3 && p.bias && p.bias3 && p.y3 && p.N3 > 0 && p.N3 % 64 == 0 && !p.x2 && p.groups == 1,
             "conv_fused3: needs both biases, W3 and an output with N3 %% 64 == 0 (%d)", p.N3);
  const size_t xb = (size_t)p.B * p.H * p.W * p.Cin * 4;
  PR_REQUIRE(xb < (1ull << 31) && (size_t)p.M() * p.N3 * 4 < (1ull << 31), "conv_fused3: tensor too large for one launch");
  FArgs fa;
  fa.x = p.x; fa.w2 = p.w; fa.bias2 = p.bias; fa.w3 = p.w3; fa.bias3 = p.bias3; fa.res = p.res3; fa.y = p.y3;
  fa.x_bytes = (unsigned)xb; fa.w2_bytes = (unsigned)((size_t)64 * p.Kpad() * 4); fa.w3_bytes = (unsigned)((size_t)p.N3 * 64 * 4);
  fa.H = p.H; fa.W = p.W; fa.Cin = p.Cin; fa.log2Cin = l2; fa.HoWo = p.H * p.W; fa.M = p.M(); fa.nk = p.Kpad() / BK;
  fa.N3 = p.N3; fa.relu3 = p.relu3;
  if (fa.M == 0) return PR_OK;
  static const int staged = [] { const char* e = getenv("POSERISK_FUSED3_STAGED"); return e ? atoi(e) : 0; }();   // A/B timing
  if (staged) {
    static std::atomic<uint64_t> attr_done{0};
    PR_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(conv3x3_conv1x1_f32<false>), kLdsStaged, attr_done));
    hipLaunchKernelGGL(conv3x3_conv1x1_f32<false>, dim3(ceil_div(fa.M, 64)), dim3(256), kLdsStaged, stream, fa);
  } else {
    hipLaunchKernelGGL(conv3x3_conv1x1_f32<true>, dim3(ceil_div(fa.M, 64)), dim3(256), kLdsDirect, stream, fa);
  }
  return check_launch("conv3x3_conv1x1_f32");
}

}  // namespace pr
