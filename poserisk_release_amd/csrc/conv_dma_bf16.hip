// bf16 implicit-GEMM convolution (fp32 accumulate) on v_mfma_f32_16x16x32_bf16 (round 6; 32 x 32 output tiles as four
// 16 x 16 quadrants, common.h Acc32 -- the sums keep the bits of the 32x32x16 form it was written on), LDS-DMA data path.
// Twin of conv_dma.hip (see there for the data path, swizzle and zero-fill notes); differences:
//   * elements are 2 bytes: a 128-byte LDS row holds BK = 64 k-values, one 16-byte chunk = 8 bf16 = the
//     whole A (or B) fragment of one lane for one MFMA (lane group g takes k = 8g..8g+7 of a 32-k step), so one
//     ds_read_b128 feeds an MFMA that does 16x the work of the fp32 one: this kernel is bound by
//     L2->LDS bandwidth and HBM, not by the matrix pipe, and wants the big tiles;
//   * activations, residual and output are bf16 (round-to-nearest-even on store), bias stays fp32;
//   * the stem pads Cin 3 -> 8 (one pixel = one 16-byte chunk).
//
// Same math as conv_igemm.hip, different data path:
//   HBM/L2 --buffer_load_dwordx4 ... lds--> LDS   (no VGPR staging, no ds_write; rows outside the
//        image or past K are addressed out of the buffer's range, so the hardware writes zeros)
//   LDS rows are 128 B (BK = 32 floats) and unpadded, because one DMA wave-instruction writes
//   64 lanes x 16 B = 8 whole rows contiguously.  Bank conflicts are removed by an XOR swizzle
//   applied on the SOURCE side: the 16-byte chunk stored at physical slot p of row r is logical
//   chunk p ^ ((r >> 1) & 7); ds_read_b128 applies the same XOR (guide rule 21).
//   Per K-step and per lane the address work is: nothing for 1x1 convs (the scalar offset
//   advances), one bounds test per row for 3x3 convs (tap decode is scalar), a per-lane tap
//   decode only for the 7x7 stem (Cin = 4).
//   One barrier per K-step: wait own DMA (vmcnt(0)) -> barrier -> issue next DMA -> MFMAs.
#include "conv_igemm.h"

namespace pr {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __attribute__((address_space(3))) void lds_void;

constexpr int BK = 64;                // 64 bf16 = 128 B per row per stage
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u16x8 = __attribute__((ext_vector_type(8))) unsigned short;
[[maybe_unused]] constexpr unsigned kOOB = 0x80000000u;  // voffset sentinel: beyond any buffer we accept (< 2 GiB)

struct DArgs {
  const unsigned short* x;   // bf16 bits
  const unsigned short* w;
  const float* bias;
  const unsigned short* res;
  unsigned short* y;
  unsigned x_bytes, w_bytes;
  int H, W, Cin, log2Cin, Ho, Wo, HoWo, Cout, stride, pad;
  int M, K, Kpad, nk;
  int tiles_n;
  int relu;
  // second A-operand source (DUAL kernels, 1x1 only): see conv_dma.hip
  const unsigned short* x2;
  unsigned x2_bytes;
  int H2, W2, Cin2, stride2, nk1;
};

__device__ inline float bf16_to_f32(unsigned short b) { return __uint_as_float((unsigned)b << 16); }
__device__ inline unsigned short f32_to_bf16(float f) {  // round-to-nearest-even (v_cvt_pk_bf16_f32 keeps NaN a NaN)
  const __bf16 h = (__bf16)f;
  return __builtin_bit_cast(unsigned short, h);
}

// TAP: 0 = 1x1 kernel (k = ci), 1 = one tap per K-step (Cin % 64 == 0), 2 = per-lane tap (Cin < 64)
template <int BM, int BN, int WAVES_M, int WAVES_N, int KS, int TAP, bool DUAL>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void conv_dma_bf16(const DArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)  // the host pass only needs the launch stub (LDS address-space casts
                                     // and gfx950 builtins in the body do not type-check there)
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int MI = WM / 32, NI = WN / 32;
  constexpr int GA = BM / 8, GB = BN / 8;        // 8-row DMA groups per tile
  constexpr int IA = GA / NW, IB = GB / NW;      // DMA instructions per wave per K-step
  static_assert(GA % NW == 0 && GB % NW == 0 && NW % 2 == 0, "DMA groups must split evenly over waves");
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int nb = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nb >> 3, rr = nb & 7;
  const int logical = (xcd < rr ? xcd * (q8 + 1) : rr * (q8 + 1) + (xcd - rr) * q8) + (bid >> 3);
  const int tile_n = logical % a.tiles_n, tile_m = logical / a.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // ---- DMA source addressing -------------------------------------------------------------
  // Wave w issues groups g = w + NW*i (same parity as w, NW even), lane covers row 8g + (lane>>3)
  // and physical chunk lane&7, i.e. logical chunk q = (lane&7) ^ ((4g + (lane>>4)) & 7).
  const int q = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
  const auto xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  const auto wsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.w), 0, (int)a.w_bytes, 0x00020000);

  int a_base[IA];  // byte offset of (img, hi0, wi0, ci = 4q); TAP 2: ci = 0
  int a_hi0[IA], a_wi0[IA];
  [[maybe_unused]] int a_base2[IA];   // DUAL: the same rows in the second source
  [[maybe_unused]] const auto xsrc2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(DUAL ? a.x2 : a.x), 0,
                                                                          DUAL ? (int)a.x2_bytes : 0, 0x00020000);
#pragma unroll
  for (int i = 0; i < IA; ++i) {
    const int r = 8 * (wave + NW * i) + (lane >> 3);
    const int m = m0 + r;
    a_base2[i] = (int)kOOB;
    if (m < a.M) {
      const int img = m / a.HoWo, rem = m - img * a.HoWo;
      const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
      a_hi0[i] = ho * a.stride - a.pad;
      a_wi0[i] = wo * a.stride - a.pad;
      a_base[i] = (((img * a.H + a_hi0[i]) * a.W + a_wi0[i]) * a.Cin + (TAP == 2 ? 0 : q * 8)) * 2;
      if (DUAL) a_base2[i] = (((img * a.H2 + ho * a.stride2) * a.W2 + wo * a.stride2) * a.Cin2 + q * 8) * 2;
    } else {
      a_hi0[i] = -(1 << 28);
      a_wi0[i] = 0;
      a_base[i] = (int)kOOB;
    }
  }
  unsigned b_off[IB];
#pragma unroll
  for (int i = 0; i < IB; ++i) {
    const int r = 8 * (wave + NW * i) + (lane >> 3);
    b_off[i] = (unsigned)(((n0 + r) * a.Kpad + q * 8) * 2);
  }

  auto issue = [&](int kt, int buf) {
    char* stage = smem + buf * STAGE;
    if (TAP == 0) {
      if (DUAL && kt >= a.nk1) {      // wave-uniform: the K-steps of the second source
        const int soff = (kt - a.nk1) * 128;
#pragma unroll
        for (int i = 0; i < IA; ++i)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc2, (lds_void*)(stage + (wave + NW * i) * 1024), 16,
                                                   (unsigned)a_base2[i], soff, 0, 0);
      } else {
        const int soff = kt * 128;
#pragma unroll
        for (int i = 0; i < IA; ++i)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, (lds_void*)(stage + (wave + NW * i) * 1024), 16,
                                                   (unsigned)a_base[i], soff, 0, 0);
      }
    } else if (TAP == 1) {
      // slice-major K (conv_k_index_bf16): K-step kt is tap kt % KS^2 of the 64-channel slice kt / KS^2
      const int tap = kt % (KS * KS), ci0 = (kt / (KS * KS)) * BK;
      const int kh = tap / KS, kw = tap - kh * KS;
      // the tap offset goes into the (range-checked) vector offset: a_base alone is negative for
      // rows whose window starts in the padding
      const int koff = ((kh * a.W + kw) * a.Cin + ci0) * 2;
#pragma unroll
      for (int i = 0; i < IA; ++i) {
        const bool ok = (unsigned)(a_hi0[i] + kh) < (unsigned)a.H && (unsigned)(a_wi0[i] + kw) < (unsigned)a.W;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, (lds_void*)(stage + (wave + NW * i) * 1024), 16,
                                                 ok ? (unsigned)(a_base[i] + koff) : kOOB, 0, 0, 0);
      }
    } else {
      const int k = kt * BK + q * 8;
      const int tap = k >> a.log2Cin, ci = k & (a.Cin - 1);
      const int kh = tap / KS, kw = tap - kh * KS;
      const int koff = ((kh * a.W + kw) * a.Cin + ci) * 2;
#pragma unroll
      for (int i = 0; i < IA; ++i) {
        const bool ok = k < a.K && (unsigned)(a_hi0[i] + kh) < (unsigned)a.H &&
                        (unsigned)(a_wi0[i] + kw) < (unsigned)a.W;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, (lds_void*)(stage + (wave + NW * i) * 1024), 16,
                                                 ok ? (unsigned)(a_base[i] + koff) : kOOB, 0, 0, 0);
      }
    }
    const int wsoff = kt * 128;
#pragma unroll
    for (int i = 0; i < IB; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrc, (lds_void*)(stage + A_BYTES + (wave + NW * i) * 1024), 16,
                                               b_off[i], wsoff, 0, 0);
  };

  // ---- fragment read addressing (swizzled) ---------------------------------------------------
  // v_mfma_f32_16x16x32_bf16 (common.h, Acc32): lane (j = lane & 15, g = lane >> 4) reads row (tile row base + 16 t + j),
  // k 32 s + 8 g .. + 7 of the 64-k stage row: logical 16-byte chunk 4 s + g, physical chunk (4 s + g) ^ ((row >> 1) & 7);
  // row bases are multiples of 32, so the XOR term is (j >> 1) & 7.  A ds_read_b128's 16-lane groups take rows {0-3, 12-15}
  // of one chunk and rows {4-11} of its neighbour: 16 different 16-byte slots of the 256-byte bank row, conflict-free.
  const int fj = frag_row(lane), fg = frag_kblock(lane), fsw = (fj >> 1) & 7;
  int foff[2][2];                          // [k-step of 32][row half t]
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int t = 0; t < 2; ++t) foff[s][t] = (16 * t + fj) * 128 + (((4 * s + fg) ^ fsw) << 4);

  Acc32 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc32_zero(acc[mi][ni]);

  auto compute = [&](int buf) {
    const char* Ab = smem + buf * STAGE + wm * WM * 128;
    const char* Bb = smem + buf * STAGE + A_BYTES + wn * WN * 128;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 af[MI][2], bf[NI][2];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int t = 0; t < 2; ++t) af[mi][t] = *reinterpret_cast<const bf16x8*>(Ab + mi * 32 * 128 + foff[s][t]);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int t = 0; t < 2; ++t) bf[ni][t] = *reinterpret_cast<const bf16x8*>(Bb + ni * 32 * 128 + foff[s][t]);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) mfma_bf16_32x32x32(acc[mi][ni], af[mi][0], af[mi][1], bf[ni][0], bf[ni][1]);
    }
  };

  issue(0, 0);
  for (int kt = 0; kt < a.nk; ++kt) {
    // own DMA of stage kt has landed; after the barrier everyone's has, and everyone has finished
    // reading the other buffer (stage kt-1), so it may be refilled.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + 1 < a.nk) issue(kt + 1, (kt + 1) & 1);
    compute(kt & 1);
  }

  // ---- epilogue through LDS --------------------------------------------------------------------
  // A lane's fragment holds one column of bf16 output per row: stored directly that is 64 bytes per
  // row per instruction (half an HBM line), and the conv3 layers, which write 4x what they read, ran at
  // 1.5 TB/s.  Instead the fp32 tile goes to LDS ([BM][BN+4] floats, reusing the stage buffers) and is
  // read back row-wise: each thread owns 8 consecutive columns, loads the residual as one 16-byte
  // chunk, adds bias/residual in fp32, applies ReLU, rounds to bf16 and stores 16 bytes, so a row of
  // BN columns leaves as whole 128-byte lines.
  constexpr int CT_STRIDE = BN + 4;
  // the fp32 tile of a 256 x 256 configuration (266 KB) does not fit LDS: it goes through in EP passes of BM / EP rows
  constexpr int EP = (size_t)BM * CT_STRIDE * 4 > 160 * 1024 ? 2 : 1, RM = BM / EP;
  static_assert(WAVES_M % EP == 0 && RM % WM == 0, "an epilogue pass takes whole waves");
  float* Ct = reinterpret_cast<float*>(smem);
  constexpr int CPR = BN / 8;  // 16-byte output chunks per row
#pragma unroll
  for (int ep = 0; ep < EP; ++ep) {
    __syncthreads();  // every wave has finished reading the stage buffers (the previous pass's rows)
    if (EP == 1 || wm / (WAVES_M / EP) == ep) {
      const int col_l = acc_col(lane), row_h = 4 * acc_half(lane);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const f32x16 regs = acc32_regs(acc[mi][ni]);      // the 32x32x16 register layout on lanes (acc_col, acc_half)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int r = wm * WM - ep * RM + mi * 32 + row_h + (e & 3) + 8 * (e >> 2);
            Ct[r * CT_STRIDE + wn * WN + ni * 32 + col_l] = regs[e];
          }
        }
    }
    __syncthreads();
    for (int idx = tid; idx < RM * CPR; idx += NW * 64) {
      const int r = idx / CPR, cc = idx - r * CPR;
      const int row = m0 + ep * RM + r, col = n0 + cc * 8;
      if (row >= a.M) continue;
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(&Ct[r * CT_STRIDE + cc * 8]);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(&Ct[r * CT_STRIDE + cc * 8 + 4]);
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
      const long o = (long)row * a.Cout + col;
      if (a.res) {
        const u16x8 rr = *reinterpret_cast<const u16x8*>(a.res + o);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bf16_to_f32(rr[e]);
      }
      u16x8 out;
#pragma unroll
      for (int e = 0; e < 8; ++e) out[e] = f32_to_bf16(a.relu ? fmaxf(v[e], 0.f) : v[e]);
      *reinterpret_cast<u16x8*>(a.y + o) = out;
    }
  }
#endif  // __HIP_DEVICE_COMPILE__
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int KS, int TAP, bool DUAL = false>
int launch_one_bf16(const DArgs& da, int grid, hipStream_t stream) {
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr size_t lds_stage = (size_t)2 * (BM + BN) * 128, lds_epi_whole = (size_t)BM * (BN + 4) * 4;
  constexpr size_t lds_epi = lds_epi_whole > 160 * 1024 ? lds_epi_whole / 2 : lds_epi_whole;   // two passes (the kernel's EP)
  constexpr size_t lds = lds_stage > lds_epi ? lds_stage : lds_epi;
  void (*kern)(const DArgs) = conv_dma_bf16<BM, BN, WAVES_M, WAVES_N, KS, TAP, DUAL>;
  static std::atomic<uint64_t> attr_done{0};  // per instantiation, one bit per device
  PR_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, attr_done));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, stream, da);
  return check_launch("conv_dma_bf16");
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
int launch_dma_bf16(const DArgs& da, int ks, int tap, int grid, hipStream_t stream) {
  if (tap == 0 && da.x2) return launch_one_bf16<BM, BN, WAVES_M, WAVES_N, 1, 0, true>(da, grid, stream);
  if (tap == 0) return launch_one_bf16<BM, BN, WAVES_M, WAVES_N, 1, 0>(da, grid, stream);
  if (tap == 1 && ks == 3) return launch_one_bf16<BM, BN, WAVES_M, WAVES_N, 3, 1>(da, grid, stream);
  if (tap == 2 && ks == 7) return launch_one_bf16<BM, BN, WAVES_M, WAVES_N, 7, 2>(da, grid, stream);
  if (tap == 2 && ks == 3) return launch_one_bf16<BM, BN, WAVES_M, WAVES_N, 3, 2>(da, grid, stream);
  if (tap == 2 && ks == 4)     // the stem after space-to-depth: 4x4 taps of 16 (12 real) channels
    return launch_one_bf16<BM, BN, WAVES_M, WAVES_N, 4, 2>(da, grid, stream);
  set_error("conv_dma: unsupported kernel size %d / tap mode %d", ks, tap);
  return PR_ERR_INVALID;
}

int ilog2_exact_b(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return (1 << l) == v ? l : -1;
}

}  // namespace

int conv_dma_bf16_launch(const ConvProblem& p, int BM, int BN, hipStream_t stream, int threads) {
  PR_REQUIRE(p.KH == p.KW, "conv: square kernels only");
  PR_REQUIRE(p.Cin % 8 == 0 && p.Cout % BN == 0, "conv: bad channels Cin=%d Cout=%d (tile N %d)", p.Cin, p.Cout, BN);
  const int K2 = p.x2 ? p.Cin2 : 0;   // second source (1x1): its channels extend the K loop
  const int Kpad1 = ceil_div(p.K(), BK) * BK, Kpad = Kpad1 + K2;
  const size_t xb = (size_t)p.B * p.H * p.W * p.Cin * 2, wb = (size_t)p.Cout * Kpad * 2;
  const size_t x2b = p.x2 ? (size_t)p.B * p.H2 * p.W2 * p.Cin2 * 2 : 0;
  PR_REQUIRE(xb < (1ull << 31) && wb < (1ull << 31) && (size_t)p.M() * p.Cout < (1ull << 31),
             "conv: tensor too large for one launch (%zu input bytes)", xb);
  const int l2 = ilog2_exact_b(p.Cin);
  int tap;
  if (p.KH == 1 && p.pad == 0) tap = 0;
  else if (p.Cin % BK == 0) tap = 1;     // slice-major K: no power of two needed
  else tap = 2;
  PR_REQUIRE(tap != 2 || l2 >= 0, "conv: k>1 with Cin %% 64 != 0 needs a power-of-two Cin (%d)", p.Cin);
  PR_REQUIRE(tap != 0 || p.Cin % BK == 0, "conv: bf16 1x1 path needs Cin %% 64 == 0 (%d)", p.Cin);
  if (p.x2) {
    PR_REQUIRE(tap == 0 && p.groups == 1 && p.Cin2 % BK == 0 && p.stride2 > 0 && x2b < (1ull << 31),
               "conv: a second source needs a 1x1 conv and Cin2 %% 64 == 0 (%d)", p.Cin2);
    PR_REQUIRE((p.H2 - 1) / p.stride2 + 1 == p.Ho && (p.W2 - 1) / p.stride2 + 1 == p.Wo,
               "conv: second source %dx%d / stride %d does not land on the %dx%d output", p.H2, p.W2, p.stride2, p.Ho, p.Wo);
  }
  DArgs da;
  da.x = reinterpret_cast<const unsigned short*>(p.x); da.w = reinterpret_cast<const unsigned short*>(p.w);
  da.bias = p.bias; da.res = reinterpret_cast<const unsigned short*>(p.res); da.y = reinterpret_cast<unsigned short*>(p.y);
  da.x_bytes = (unsigned)xb; da.w_bytes = (unsigned)wb;
  da.H = p.H; da.W = p.W; da.Cin = p.Cin; da.log2Cin = l2 < 0 ? 0 : l2;
  da.Ho = p.Ho; da.Wo = p.Wo; da.HoWo = p.Ho * p.Wo; da.Cout = p.Cout; da.stride = p.stride; da.pad = p.pad;
  da.M = p.M(); da.K = p.K() + K2; da.Kpad = Kpad; da.nk = Kpad / BK;
  da.x2 = reinterpret_cast<const unsigned short*>(p.x2); da.x2_bytes = (unsigned)x2b;
  da.H2 = p.H2; da.W2 = p.W2; da.Cin2 = p.Cin2; da.stride2 = p.stride2; da.nk1 = Kpad1 / BK;
  da.tiles_n = p.Cout / BN;
  da.relu = p.relu;
  if (da.M == 0) return PR_OK;
  const int grid = ceil_div(da.M, BM) * da.tiles_n;
  const int key = BM * 1000 + BN + (threads == 512 && BM == 128 ? 500000 : 0) + (threads == 128 ? 900000 : 0);
  switch (key) {
    case 128128: return launch_dma_bf16<128, 128, 2, 2>(da, p.KH, tap, grid, stream);
    case 628128: return launch_dma_bf16<128, 128, 4, 2>(da, p.KH, tap, grid, stream);
    case 628064: return launch_dma_bf16<128, 64, 4, 2>(da, p.KH, tap, grid, stream);
    case 964064: return launch_dma_bf16<64, 64, 2, 1>(da, p.KH, tap, grid, stream);
    case 1028064: return launch_dma_bf16<128, 64, 2, 1>(da, p.KH, tap, grid, stream);
    case 964128: return launch_dma_bf16<64, 128, 1, 2>(da, p.KH, tap, grid, stream);
    case 128064: return launch_dma_bf16<128, 64, 2, 2>(da, p.KH, tap, grid, stream);
    case 64064: return launch_dma_bf16<64, 64, 2, 2>(da, p.KH, tap, grid, stream);
    case 256128: return launch_dma_bf16<256, 128, 4, 2>(da, p.KH, tap, grid, stream);
    case 64128: return launch_dma_bf16<64, 128, 2, 2>(da, p.KH, tap, grid, stream);
    case 256064: return launch_dma_bf16<256, 64, 4, 2>(da, p.KH, tap, grid, stream);
    case 64256: return launch_dma_bf16<64, 256, 2, 2>(da, p.KH, tap, grid, stream);
    case 256256: return launch_dma_bf16<256, 256, 4, 2>(da, p.KH, tap, grid, stream);   // bf16 only (round 6: experiment 6)
  }
  set_error("conv_dma: no %dx%d tile", BM, BN);
  return PR_ERR_INVALID;
}

}  // namespace pr
