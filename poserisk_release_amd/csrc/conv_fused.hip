// A Bottleneck's 3x3 convolution and the 1x1 expansion behind it as ONE kernel (fp32, and a bf16 twin below):
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
constexpr int kLds = 2 * kStage;              // 32 KB: five workgroups per CU, like the plain 64x64 kernel

// GEMM 2's outputs leave straight from the accumulator fragments (per store instruction two 128-byte row segments)
// through range-checked buffer descriptors.  Measured against an LDS-transposed epilogue like conv_dma.hip's (whole
// 16-byte chunks per lane, +17 KB of LDS = three resident workgroups instead of five, two more barriers per column
// chunk): 193 us vs 215 us per layer1 block at B=64 (separate launches: 238 us).
__global__ __launch_bounds__(256, 5) void conv3x3_conv1x1_f32(const FArgs a) {
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
  const int nsteps = (a.N3 >> 6) * 2;
  float rfrag[16];
  const int col_l = lane & 31, row_h = 4 * (lane >> 5);
  // residual and output through range-checked buffer descriptors, one 32-bit lane offset for both
  const int yz_bytes = a.M * a.N3 * 4;
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res ? a.res : a.y), 0, yz_bytes, 0x00020000);
  const auto ysrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, yz_bytes, 0x00020000);
  const int frag_off = ((m0 + wm * 32 + row_h) * a.N3 + wn * 32 + col_l) * 4;
  // Rows >= M of a ragged last tile are dropped by making their VECTOR offset the out-of-range sentinel: fragment row
  // (e & 3) + 8 (e >> 2) is valid while it is < mlim.  (LLVM documents the scalar offset, which carries the row term
  // below, as outside the range check; gfx950 measured does include it -- scripts/micro/t_soffset.hip -- and this code
  // relies on neither.)
  const int mlim = a.M - (m0 + wm * 32 + row_h);
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
      if (a.res) {      // in the fragment layout: acc[e] <-> (row wm*32 + row_h + (e&3) + 8(e>>2), col wn*32 + col_l)
#pragma unroll
        for (int e = 0; e < 16; ++e)    // rows >= M get the out-of-range vector offset: they read as zero
          rfrag[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                   rsrc, (e & 3) + 8 * (e >> 2) < mlim ? frag_off : (int)kOOB,
                                                   (n0 + ((e & 3) + 8 * (e >> 2)) * a.N3) * 4, 0));
      }
    }
    compute(smem + kT2 + (s & 1) * 8192 + wm * 32 * 128, smem + kRing + (s & 1) * 8192 + wn * 32 * 128);
    if (s & 1) {
      const float b3 = a.bias3[n0 + wn * 32 + col_l];
#pragma unroll
      for (int e = 0; e < 16; ++e) {    // stores to rows >= M get the out-of-range vector offset and are dropped
        float v = acc[e] + b3;
        if (a.res) v += rfrag[e];
        if (a.relu3) v = fmaxf(v, 0.f);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ysrc,
                                              (e & 3) + 8 * (e >> 2) < mlim ? frag_off : (int)kOOB,
                                              (n0 + ((e & 3) + 8 * (e >> 2)) * a.N3) * 4, 0);
      }
    }
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// Row-panel form of a short-K 1x1 convolution (fp32): y = act(x * W^T [+ x2 * W2^T] + b + res), K <= 256.
//
// The GEMM-2 loop above as a kernel of its own, for the conv3 layers whose K is only a few K-steps (layer2: K = 128;
// a first block's conv3 + downsample: K = 64 + 64).  As 64x64 tiles those layers spend most of a tile's life in its
// prologue and epilogue (profiles/r02_conv3_sweep_b64.txt: 75 TFLOP/s at K = 128, 58 at K = 64, whatever the tile);
// here a workgroup loads its 64 rows of x ONCE (all of K: nk stages of 8 KB), keeps them in LDS, and walks over
// `chunks` 64-column chunks of W streaming through the two-stage ring: one address set-up, the ring always one stage
// ahead, the next chunk's residual requested while this chunk multiplies, outputs straight from the fragments.
// Same ascending-k fmaf chain per output element as the tile kernel: the same bits.
// ---------------------------------------------------------------------------------------------------------------
struct PArgs {
  const float* x;      // [M][K1]: the rows of a 1x1 / stride-1 convolution's input
  const float* x2;     // optional second source [B,H2,W2,Cin2], sampled at (ho * stride2, wo * stride2): K-steps [nk1, nk)
  const float* w;      // [N][K] packed
  const float* bias;
  const float* res;    // [M][N] or nullptr
  float* y;
  unsigned x_bytes, x2_bytes, w_bytes;
  int M, N, K, nk, nk1, HoWo, Wo, H2, W2, Cin2, stride2;
  int chunks, nsplit, relu;
};

template <bool DUAL>
__global__ __launch_bounds__(256, 3) void conv1x1_panel_f32(const PArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nb = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nb >> 3, rr = nb & 7;
  const int item = (xcd < rr ? xcd * (q8 + 1) : rr * (q8 + 1) + (xcd - rr) * q8) + (bid >> 3);
  const int panel = item / a.nsplit, part = item - panel * a.nsplit;
  const int m0 = panel * 64, nbase = part * a.chunks * 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int q = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
  const auto xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  [[maybe_unused]] const auto x2src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(DUAL ? a.x2 : a.x), 0,
                                                                          DUAL ? (int)a.x2_bytes : 0, 0x00020000);
  const auto wsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, (int)a.w_bytes, 0x00020000);
  char* ring = smem + a.nk * 8192;

  // ---- the row panel: all of K for 64 rows, nk stages in the operand layout ------------------------------------
  const int K1 = a.nk1 * BK;
  unsigned b_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = 8 * (wave + 4 * i) + (lane >> 3);
    const int m = m0 + r;
    unsigned base = kOOB, base2 = kOOB;
    if (m < a.M) {
      base = (unsigned)((m * K1 + q * 4) * 4);
      if (DUAL) {
        const int img = m / a.HoWo, rem = m - img * a.HoWo;
        const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
        base2 = (unsigned)((((img * a.H2 + ho * a.stride2) * a.W2 + wo * a.stride2) * a.Cin2 + q * 4) * 4);
      }
    }
    for (int kt = 0; kt < a.nk; ++kt) {
      lds_void* dst = (lds_void*)(smem + kt * 8192 + (wave + 4 * i) * 1024);
      if (DUAL && kt >= a.nk1) __builtin_amdgcn_raw_ptr_buffer_load_lds(x2src, dst, 16, base2, (kt - a.nk1) * 128, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, dst, 16, base, kt * 128, 0, 0);
    }
    b_off[i] = (unsigned)((r * a.K + q * 4) * 4);
  }
  // W step s = chunk * nk + kt: rows [nbase + 64 chunk, +64), k in [32 kt, 32 kt + 32)
  auto issue_w = [&](int chunk, int kt, int buf) {
    const int soff = ((nbase + chunk * 64) * a.K + kt * BK) * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrc, (lds_void*)(ring + buf * 8192 + (wave + 4 * i) * 1024), 16, b_off[i],
                                               soff, 0, 0);
  };
  issue_w(0, 0, 0);

  const int frow = lane & 31, fh = lane >> 5, fsw = (frow >> 1) & 7;
  int foff[BK / 8];
#pragma unroll
  for (int kk = 0; kk < BK / 8; ++kk) foff[kk] = frow * 128 + (((2 * kk + fh) ^ fsw) << 4);
  f32x16 acc;
  float rfrag[16];
  const int col_l = lane & 31, row_h = 4 * (lane >> 5);
  const int yz_bytes = a.M * a.N * 4;
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res ? a.res : a.y), 0, yz_bytes, 0x00020000);
  const auto ysrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, yz_bytes, 0x00020000);
  const int frag_off = ((m0 + wm * 32 + row_h) * a.N + nbase + wn * 32 + col_l) * 4;
  const int mlim = a.M - (m0 + wm * 32 + row_h);   // fragment rows < mlim exist (see conv3x3_conv1x1_f32: out-of-range rows get the sentinel as their vector offset)

  int chunk = 0, kt = 0;
  const int nsteps = a.chunks * a.nk;
  for (int s = 0; s < nsteps; ++s) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();      // stage s (and, at s = 0, the panel) has landed; stage s-1 is consumed
    asm volatile("" ::: "memory");
    {
      int nchunk = chunk, nkt = kt + 1;
      if (nkt == a.nk) { nkt = 0; ++nchunk; }
      if (s + 1 < nsteps) issue_w(nchunk, nkt, (s + 1) & 1);
    }
    const int n0 = chunk * 64;
    if (kt == 0) {
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
      if (a.res) {
#pragma unroll
        for (int e = 0; e < 16; ++e)
          rfrag[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                   rsrc, (e & 3) + 8 * (e >> 2) < mlim ? frag_off : (int)kOOB,
                                                   (n0 + ((e & 3) + 8 * (e >> 2)) * a.N) * 4, 0));
      }
    }
    {
      const char* Ab = smem + kt * 8192 + wm * 32 * 128;
      const char* Bb = ring + (s & 1) * 8192 + wn * 32 * 128;
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
    }
    if (kt == a.nk - 1) {
      const float b = a.bias ? a.bias[nbase + n0 + wn * 32 + col_l] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float v = acc[e] + b;
        if (a.res) v += rfrag[e];
        if (a.relu) v = fmaxf(v, 0.f);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ysrc,
                                              (e & 3) + 8 * (e >> 2) < mlim ? frag_off : (int)kOOB,
                                              (n0 + ((e & 3) + 8 * (e >> 2)) * a.N) * 4, 0);
      }
      kt = 0;
      ++chunk;
    } else {
      ++kt;
    }
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// bf16 twin (BASELINE config 3): v_mfma_f32_32x32x16_bf16, fp32 accumulate, bf16 tensors.  A 128-byte LDS row holds
// 64 k-values, so conv2's whole K per tap is one K-step, the t2 tile (rounded to bf16 where the separate launch stores
// it) is ONE GEMM-2 K-step of 8 KB and each 64-column chunk of W3 one 8 KB ring stage.  A lane's output column is two
// bytes wide, so here the chunk does leave through an LDS transpose (16-byte stores of 8 channels, whole lines per
// row; conv_dma_bf16.hip); the staging tile is unpadded (its reads are fully contiguous) and the kernel needs exactly
// 40 KB of LDS: four workgroups per CU.
// ---------------------------------------------------------------------------------------------------------------
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u16x8 = __attribute__((ext_vector_type(8))) unsigned short;

struct FArgsB {
  const unsigned short* x;
  const unsigned short* w2;
  const float* bias2;
  const unsigned short* w3;
  const float* bias3;
  const unsigned short* res;
  unsigned short* y;
  unsigned x_bytes, w2_bytes, w3_bytes;
  int H, W, Cin, log2Cin, HoWo, M, nk, N3, relu3;
};

constexpr int kBStage = 2 * 64 * 128;      // A 8 KB + B 8 KB
constexpr int kBT2 = 0;                    // 8 KB
constexpr int kBRing = 8192;               // 2 x 8 KB
constexpr int kBCt = 8192 + 16384;         // [64][64] floats = 16 KB
constexpr int kBLds = kBCt + 64 * 64 * 4;  // 40960 <= the main loop's 2 stages (32768)? no: 40 KB, four per CU

__device__ inline unsigned short f2bf(float f) {
  const __bf16 h = (__bf16)f;
  return __builtin_bit_cast(unsigned short, h);
}

__global__ __launch_bounds__(256, 4) void conv3x3_conv1x1_bf16(const FArgsB a) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BKB = 64;
  const int nb = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nb >> 3, rr = nb & 7;
  const int tile_m = (xcd < rr ? xcd * (q8 + 1) : rr * (q8 + 1) + (xcd - rr) * q8) + (bid >> 3);
  const int m0 = tile_m * 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int q = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
  const auto xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  const auto w2src = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.w2), 0, (int)a.w2_bytes, 0x00020000);
  const auto w3src = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.w3), 0, (int)a.w3_bytes, 0x00020000);
  int a_base[2], a_hi0[2], a_wi0[2];
  unsigned b2_off[2], b3_off[2];
  const int K2 = a.nk * BKB;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = 8 * (wave + 4 * i) + (lane >> 3);
    const int m = m0 + r;
    if (m < a.M) {
      const int img = m / a.HoWo, rem = m - img * a.HoWo;
      const int ho = rem / a.W, wo = rem - ho * a.W;
      a_hi0[i] = ho - 1;
      a_wi0[i] = wo - 1;
      a_base[i] = (((img * a.H + a_hi0[i]) * a.W + a_wi0[i]) * a.Cin + q * 8) * 2;
    } else {
      a_hi0[i] = -(1 << 28);
      a_wi0[i] = 0;
      a_base[i] = (int)kOOB;
    }
    b2_off[i] = (unsigned)((r * K2 + q * 8) * 2);
    b3_off[i] = (unsigned)((r * 64 + q * 8) * 2);
  }
  auto issue = [&](int kt, int buf) {
    char* stage = smem + buf * kBStage;
    const int tap = kt % 9, ci0 = (kt / 9) * BKB;     // slice-major K (conv_k_index_bf16)
    const int kh = tap / 3, kw = tap - kh * 3;
    const int koff = ((kh * a.W + kw) * a.Cin + ci0) * 2;
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
  auto issue3 = [&](int nc) {     // rows [64 nc, 64 nc + 64) of W3, all 64 k
    char* stage = smem + kBRing + (nc & 1) * 8192;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w3src, (lds_void*)(stage + (wave + 4 * i) * 1024), 16, b3_off[i],
                                               nc * 64 * 64 * 2, 0, 0);
  };

  const int frow = lane & 31, fh = lane >> 5, fsw = (frow >> 1) & 7;
  int foff[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) foff[kk] = frow * 128 + (((2 * kk + fh) ^ fsw) << 4);
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  auto compute = [&](const char* Ab, const char* Bb) {
    bf16x8 af[4], bf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      af[kk] = *reinterpret_cast<const bf16x8*>(Ab + foff[kk]);
      bf[kk] = *reinterpret_cast<const bf16x8*>(Bb + foff[kk]);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) acc = mfma_bf16_step(af[kk], bf[kk], acc, kk);
  };

  issue(0, 0);
  for (int kt = 0; kt < a.nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + 1 < a.nk) issue(kt + 1, (kt + 1) & 1);
    const char* st = smem + (kt & 1) * kBStage;
    compute(st + wm * 32 * 128, st + 8192 + wn * 32 * 128);
  }

  __syncthreads();
  issue3(0);
  {
    const int c = wn * 32 + (lane & 31);       // t2 channel = GEMM 2's k
    const float b2 = a.bias2[c];
    char* t2 = smem + kBT2;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int r = wm * 32 + 4 * (lane >> 5) + (e & 3) + 8 * (e >> 2);
      *reinterpret_cast<unsigned short*>(t2 + r * 128 + ((((c >> 3) ^ ((r >> 1) & 7))) << 4) + (c & 7) * 2) =
          f2bf(fmaxf(acc[e] + b2, 0.f));
    }
  }

  float* Ct = reinterpret_cast<float*>(smem + kBCt);
  const int nchunks = a.N3 >> 6;
  u16x8 rpre[2];
  for (int nc = 0; nc < nchunks; ++nc) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();      // W3 chunk nc landed; (nc = 0) t2 complete; the previous chunk's Ct reads are done
    asm volatile("" ::: "memory");
    if (nc + 1 < nchunks) issue3(nc + 1);
    const int n0 = nc * 64;
    if (a.res) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int idx = tid + i * 256;
        const int r = idx >> 3, cc = idx & 7;
        const int row = m0 + r;
        const u16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        rpre[i] = row < a.M ? *reinterpret_cast<const u16x8*>(a.res + (long)row * a.N3 + n0 + cc * 8) : z;
      }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    compute(smem + kBT2 + wm * 32 * 128, smem + kBRing + (nc & 1) * 8192 + wn * 32 * 128);
    {
      const int col_l = lane & 31, row_h = 4 * (lane >> 5);
#pragma unroll
      for (int e = 0; e < 16; ++e) Ct[(wm * 32 + row_h + (e & 3) + 8 * (e >> 2)) * 64 + wn * 32 + col_l] = acc[e];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + i * 256;
      const int r = idx >> 3, cc = idx & 7;
      const int row = m0 + r, col = n0 + cc * 8;
      if (row >= a.M) continue;
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(&Ct[r * 64 + cc * 8]);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(&Ct[r * 64 + cc * 8 + 4]);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.bias3 + col);
      const f32x4 b1 = *reinterpret_cast<const f32x4*>(a.bias3 + col + 4);
      float v[8] = {v0[0] + b0[0], v0[1] + b0[1], v0[2] + b0[2], v0[3] + b0[3],
                    v1[0] + b1[0], v1[1] + b1[1], v1[2] + b1[2], v1[3] + b1[3]};
      if (a.res) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += __uint_as_float((unsigned)rpre[i][e] << 16);
      }
      u16x8 out;
#pragma unroll
      for (int e = 0; e < 8; ++e) out[e] = f2bf(a.relu3 ? fmaxf(v[e], 0.f) : v[e]);
      *reinterpret_cast<u16x8*>(a.y + (long)row * a.N3 + col) = out;
    }
    // the next chunk's Ct writes follow the barrier at the top of the loop: every read above is done by then
  }
#endif
}

// Row-panel form, bf16 twin (see conv1x1_panel_f32): K-steps of 64, the rows' whole K resident (nk stages of 8 KB), W
// streaming through the ring, each 64-column chunk leaving through the LDS transpose (16-byte stores of 8 channels).
struct PArgsB {
  const unsigned short* x;
  const unsigned short* x2;
  const unsigned short* w;
  const float* bias;
  const unsigned short* res;
  unsigned short* y;
  unsigned x_bytes, x2_bytes, w_bytes;
  int M, N, K, nk, nk1, HoWo, Wo, H2, W2, Cin2, stride2;
  int chunks, nsplit, relu;
};

template <bool DUAL>
__global__ __launch_bounds__(256, 2) void conv1x1_panel_bf16(const PArgsB a) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BKB = 64;
  const int nb = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nb >> 3, rr = nb & 7;
  const int item = (xcd < rr ? xcd * (q8 + 1) : rr * (q8 + 1) + (xcd - rr) * q8) + (bid >> 3);
  const int panel = item / a.nsplit, part = item - panel * a.nsplit;
  const int m0 = panel * 64, nbase = part * a.chunks * 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int q = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
  const auto xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  [[maybe_unused]] const auto x2src = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(DUAL ? a.x2 : a.x), 0,
                                                                          DUAL ? (int)a.x2_bytes : 0, 0x00020000);
  const auto wsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.w), 0, (int)a.w_bytes, 0x00020000);
  char* ring = smem + a.nk * 8192;
  float* Ct = reinterpret_cast<float*>(ring + 16384);

  const int K1 = a.nk1 * BKB;
  unsigned b_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = 8 * (wave + 4 * i) + (lane >> 3);
    const int m = m0 + r;
    unsigned base = kOOB, base2 = kOOB;
    if (m < a.M) {
      base = (unsigned)((m * K1 + q * 8) * 2);
      if (DUAL) {
        const int img = m / a.HoWo, rem = m - img * a.HoWo;
        const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
        base2 = (unsigned)((((img * a.H2 + ho * a.stride2) * a.W2 + wo * a.stride2) * a.Cin2 + q * 8) * 2);
      }
    }
    for (int kt = 0; kt < a.nk; ++kt) {
      lds_void* dst = (lds_void*)(smem + kt * 8192 + (wave + 4 * i) * 1024);
      if (DUAL && kt >= a.nk1) __builtin_amdgcn_raw_ptr_buffer_load_lds(x2src, dst, 16, base2, (kt - a.nk1) * 128, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, dst, 16, base, kt * 128, 0, 0);
    }
    b_off[i] = (unsigned)((r * a.K + q * 8) * 2);
  }
  auto issue_w = [&](int chunk, int kt, int buf) {
    const int soff = ((nbase + chunk * 64) * a.K + kt * BKB) * 2;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrc, (lds_void*)(ring + buf * 8192 + (wave + 4 * i) * 1024), 16, b_off[i],
                                               soff, 0, 0);
  };
  issue_w(0, 0, 0);

  const int frow = lane & 31, fh = lane >> 5, fsw = (frow >> 1) & 7;
  int foff[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) foff[kk] = frow * 128 + (((2 * kk + fh) ^ fsw) << 4);
  f32x16 acc;
  u16x8 rpre[2];
  int chunk = 0, kt = 0;
  const int nsteps = a.chunks * a.nk;
  for (int s = 0; s < nsteps; ++s) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    {
      int nchunk = chunk, nkt = kt + 1;
      if (nkt == a.nk) { nkt = 0; ++nchunk; }
      if (s + 1 < nsteps) issue_w(nchunk, nkt, (s + 1) & 1);
    }
    const int n0 = nbase + chunk * 64;
    if (kt == 0) {
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
      if (a.res) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int idx = tid + i * 256;
          const int r = idx >> 3, cc = idx & 7;
          const int row = m0 + r;
          const u16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
          rpre[i] = row < a.M ? *reinterpret_cast<const u16x8*>(a.res + (long)row * a.N + n0 + cc * 8) : z;
        }
      }
    }
    {
      const char* Ab = smem + kt * 8192 + wm * 32 * 128;
      const char* Bb = ring + (s & 1) * 8192 + wn * 32 * 128;
      bf16x8 af[4], bf[4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        af[kk] = *reinterpret_cast<const bf16x8*>(Ab + foff[kk]);
        bf[kk] = *reinterpret_cast<const bf16x8*>(Bb + foff[kk]);
      }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) acc = mfma_bf16_step(af[kk], bf[kk], acc, kk);
    }
    if (kt == a.nk - 1) {
      {
        const int col_l = lane & 31, row_h = 4 * (lane >> 5);
#pragma unroll
        for (int e = 0; e < 16; ++e) Ct[(wm * 32 + row_h + (e & 3) + 8 * (e >> 2)) * 64 + wn * 32 + col_l] = acc[e];
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int idx = tid + i * 256;
        const int r = idx >> 3, cc = idx & 7;
        const int row = m0 + r, col = n0 + cc * 8;
        if (row >= a.M) continue;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(&Ct[r * 64 + cc * 8]);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(&Ct[r * 64 + cc * 8 + 4]);
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        if (a.bias) {
          const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.bias + col);
          const f32x4 b1 = *reinterpret_cast<const f32x4*>(a.bias + col + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] += b0[e];
            v[4 + e] += b1[e];
          }
        }
        if (a.res) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += __uint_as_float((unsigned)rpre[i][e] << 16);
        }
        u16x8 out;
#pragma unroll
        for (int e = 0; e < 8; ++e) out[e] = f2bf(a.relu ? fmaxf(v[e], 0.f) : v[e]);
        *reinterpret_cast<u16x8*>(a.y + (long)row * a.N + col) = out;
      }
      // the next chunk's Ct writes follow at least one more barrier (the top of the next step)
      kt = 0;
      ++chunk;
    } else {
      ++kt;
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
  PR_REQUIRE(p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && p.Cout == 64 && p.Ho == p.H && p.Wo == p.W,
             "conv_fused3: the first convolution must be 3x3 / stride 1 / pad 1 with 64 output channels");
  const int l2 = ilog2_exact_f(p.Cin);
  PR_REQUIRE(l2 >= 0 && p.Cin % (p.precision == 1 ? 64 : BK) == 0, "conv_fused3: Cin must be a power of two >= %d (%d)",
             p.precision == 1 ? 64 : BK, p.Cin);
  PR_REQUIRE(p.w3 && p.bias && p.bias3 && p.y3 && p.N3 > 0 && p.N3 % 64 == 0 && !p.x2 && p.groups == 1,
             "conv_fused3: needs both biases, W3 and an output with N3 %% 64 == 0 (%d)", p.N3);
  const size_t xb = (size_t)p.B * p.H * p.W * p.Cin * (p.precision == 1 ? 2 : 4);
  PR_REQUIRE(xb < (1ull << 31) && (size_t)p.M() * p.N3 * 4 < (1ull << 31), "conv_fused3: tensor too large for one launch");
  if (p.precision == 1) {
    FArgsB fb;
    const int Kp = conv_kpad_bf16(p.K());
    fb.x = reinterpret_cast<const unsigned short*>(p.x); fb.w2 = reinterpret_cast<const unsigned short*>(p.w);
    fb.bias2 = p.bias; fb.w3 = reinterpret_cast<const unsigned short*>(p.w3); fb.bias3 = p.bias3;
    fb.res = reinterpret_cast<const unsigned short*>(p.res3); fb.y = reinterpret_cast<unsigned short*>(p.y3);
    fb.x_bytes = (unsigned)xb; fb.w2_bytes = (unsigned)((size_t)64 * Kp * 2); fb.w3_bytes = (unsigned)((size_t)p.N3 * 64 * 2);
    fb.H = p.H; fb.W = p.W; fb.Cin = p.Cin; fb.log2Cin = l2; fb.HoWo = p.H * p.W; fb.M = p.M(); fb.nk = Kp / 64;
    fb.N3 = p.N3; fb.relu3 = p.relu3;
    if (fb.M == 0) return PR_OK;
    hipLaunchKernelGGL(conv3x3_conv1x1_bf16, dim3(ceil_div(fb.M, 64)), dim3(256), kBLds, stream, fb);
    return check_launch("conv3x3_conv1x1_bf16");
  }
  FArgs fa;
  fa.x = p.x; fa.w2 = p.w; fa.bias2 = p.bias; fa.w3 = p.w3; fa.bias3 = p.bias3; fa.res = p.res3; fa.y = p.y3;
  fa.x_bytes = (unsigned)xb; fa.w2_bytes = (unsigned)((size_t)64 * p.Kpad() * 4); fa.w3_bytes = (unsigned)((size_t)p.N3 * 64 * 4);
  fa.H = p.H; fa.W = p.W; fa.Cin = p.Cin; fa.log2Cin = l2; fa.HoWo = p.H * p.W; fa.M = p.M(); fa.nk = p.Kpad() / BK;
  fa.N3 = p.N3; fa.relu3 = p.relu3;
  if (fa.M == 0) return PR_OK;
  hipLaunchKernelGGL(conv3x3_conv1x1_f32, dim3(ceil_div(fa.M, 64)), dim3(256), kLds, stream, fa);
  return check_launch("conv3x3_conv1x1_f32");
}

// Row-panel launch of a short-K 1x1 convolution (see conv1x1_panel_f32): same arguments as conv_dma_launch's 1x1 path.
int conv_panel_launch(const ConvProblem& p, hipStream_t stream) {
  const int K2 = p.x2 ? p.Cin2 : 0, K = p.Cin + K2;
  const int bk = p.precision == 1 ? 64 : BK, el = p.precision == 1 ? 2 : 4, kmax = p.precision == 1 ? 512 : 256;
  PR_REQUIRE(p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && p.groups == 1 && !p.w3,
             "conv_panel: 1x1 / stride-1 convolutions only");
  PR_REQUIRE(p.Cin % bk == 0 && K2 % bk == 0 && K <= kmax && p.Cout % 64 == 0,
             "conv_panel: K %d (<= %d, multiple of %d), N %d", K, kmax, bk, p.Cout);
  const size_t xb = (size_t)p.M() * p.Cin * el, x2b = p.x2 ? (size_t)p.B * p.H2 * p.W2 * p.Cin2 * el : 0;
  PR_REQUIRE(xb < (1ull << 31) && x2b < (1ull << 31) && (size_t)p.M() * p.Cout * 4 < (1ull << 31),
             "conv_panel: tensor too large for one launch");
  if (p.x2)
    PR_REQUIRE((p.H2 - 1) / p.stride2 + 1 == p.Ho && (p.W2 - 1) / p.stride2 + 1 == p.Wo,
               "conv_panel: second source %dx%d / stride %d does not land on the %dx%d output", p.H2, p.W2, p.stride2, p.Ho, p.Wo);
  if (p.precision == 1) {
    PArgsB pb;
    pb.x = reinterpret_cast<const unsigned short*>(p.x); pb.x2 = reinterpret_cast<const unsigned short*>(p.x2);
    pb.w = reinterpret_cast<const unsigned short*>(p.w); pb.bias = p.bias;
    pb.res = reinterpret_cast<const unsigned short*>(p.res); pb.y = reinterpret_cast<unsigned short*>(p.y);
    pb.x_bytes = (unsigned)xb; pb.x2_bytes = (unsigned)x2b; pb.w_bytes = (unsigned)((size_t)p.Cout * K * 2);
    pb.M = p.M(); pb.N = p.Cout; pb.K = K; pb.nk = K / 64; pb.nk1 = p.Cin / 64;
    pb.HoWo = p.Ho * p.Wo; pb.Wo = p.Wo; pb.H2 = p.H2; pb.W2 = p.W2; pb.Cin2 = p.Cin2; pb.stride2 = p.stride2;
    pb.relu = p.relu;
    if (pb.M == 0) return PR_OK;
    const int panels = ceil_div(pb.M, 64), nchunks = p.Cout / 64;
    int nsplit = 1;
    while (nsplit * 2 <= nchunks && nchunks % (nsplit * 2) == 0 && panels * nsplit < 1536) nsplit *= 2;
    pb.nsplit = nsplit;
    pb.chunks = nchunks / nsplit;
    // the attribute is set once per device to what the LARGEST admissible K needs (the launch itself asks for its own K's)
    const size_t lds = (size_t)pb.nk * 8192 + 16384 + 16384, lds_max = (size_t)(kmax / bk) * 8192 + 16384 + 16384;
    if (p.x2) {
      static std::atomic<uint64_t> attr_done{0};
      PR_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(conv1x1_panel_bf16<true>), lds_max, attr_done));
      hipLaunchKernelGGL(conv1x1_panel_bf16<true>, dim3(panels * nsplit), dim3(256), lds, stream, pb);
    } else {
      static std::atomic<uint64_t> attr_done{0};
      PR_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(conv1x1_panel_bf16<false>), lds_max, attr_done));
      hipLaunchKernelGGL(conv1x1_panel_bf16<false>, dim3(panels * nsplit), dim3(256), lds, stream, pb);
    }
    return check_launch("conv1x1_panel_bf16");
  }
  PArgs pa;
  pa.x = p.x; pa.x2 = p.x2; pa.w = p.w; pa.bias = p.bias; pa.res = p.res; pa.y = p.y;
  pa.x_bytes = (unsigned)xb; pa.x2_bytes = (unsigned)x2b; pa.w_bytes = (unsigned)((size_t)p.Cout * K * 4);
  pa.M = p.M(); pa.N = p.Cout; pa.K = K; pa.nk = K / BK; pa.nk1 = p.Cin / BK;
  pa.HoWo = p.Ho * p.Wo; pa.Wo = p.Wo; pa.H2 = p.H2; pa.W2 = p.W2; pa.Cin2 = p.Cin2; pa.stride2 = p.stride2;
  pa.relu = p.relu;
  if (pa.M == 0) return PR_OK;
  // work items: row panels x column parts, enough of them to balance the 256 CUs (about 6 per CU where the layer allows)
  const int panels = ceil_div(pa.M, 64), nchunks = p.Cout / 64;
  int nsplit = 1;
  while (nsplit * 2 <= nchunks && nchunks % (nsplit * 2) == 0 && panels * nsplit < 1536) nsplit *= 2;
  pa.nsplit = nsplit;
  pa.chunks = nchunks / nsplit;
  const size_t lds = (size_t)pa.nk * 8192 + 16384, lds_max = (size_t)(kmax / bk) * 8192 + 16384;
  if (p.x2) {
    static std::atomic<uint64_t> attr_done{0};
    PR_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(conv1x1_panel_f32<true>), lds_max, attr_done));
    hipLaunchKernelGGL(conv1x1_panel_f32<true>, dim3(panels * nsplit), dim3(256), lds, stream, pa);
  } else {
    static std::atomic<uint64_t> attr_done{0};
    PR_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(conv1x1_panel_f32<false>), lds_max, attr_done));
    hipLaunchKernelGGL(conv1x1_panel_f32<false>, dim3(panels * nsplit), dim3(256), lds, stream, pa);
  }
  return check_launch("conv1x1_panel_f32");
}

}  // namespace pr
