// fp32 implicit-GEMM convolution, LDS-DMA variant (the encoder's production kernel).
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
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "conv_igemm.h"

namespace pr {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __attribute__((address_space(3))) void lds_void;

constexpr int BK = kConvBK;           // 32 floats = 128 B per row per stage
#if defined(PR_EXPERIMENT) && PR_EXPERIMENT == 10
constexpr bool kOldKLoop = true;      // A/B builds: round 3's K loop (bursts of reads and DMA in front of the MFMAs)
#else
constexpr bool kOldKLoop = false;
#endif
[[maybe_unused]] constexpr unsigned kOOB = 0x80000000u;  // voffset sentinel: beyond any buffer we accept (< 2 GiB)

struct DArgs {
  const float* x;
  const float* w;
  const float* bias;
  const float* res;
  float* y;
  unsigned x_bytes, w_bytes;
  int H, W, Cin, log2Cin, Ho, Wo, HoWo, Cout, stride, pad;
  unsigned cin_magic;   // TAP 2: k / Cin = (k * cin_magic) >> 20 for every k < Kpad (Cin need not be a power of two)
  int M, K, Kpad, nk;
  int tiles_n;
  int relu;
  int groups, tiles_per_group;   // independent GEMMs in one launch: tile index -> (group, tile_m, tile_n)
  int n_full;   // blocks [0, n_full) compute whole BMxBN tiles; the rest are quarter-tile blocks (conv_tail_quarter)
  int n_tail;   // quarter-tile work items (4 per remaining tile); the grid is padded to a multiple of 8
  // Second A-operand source (DUAL kernels, 1x1 only): K-steps [nk1, nk) read row m's channels from x2, a
  // [B,H2,W2,Cin2] tensor sampled with stride2 (a Bottleneck's downsample branch summed into conv3's K loop).
  const float* x2;
  unsigned x2_bytes;
  int H2, W2, Cin2, stride2, nk1;
  // Split-K (SPLIT kernels): a tile's K-steps are dealt to `splitk` workgroups (work item = tile * splitk + part,
  // part p takes K-steps [p * nk_part, (p+1) * nk_part)); every part stores its raw fp32 partial tile into `slab`
  // ([tiles * splitk][BM * BN]) and draws a ticket from `tickets[tile]` (zeroed by the launcher on the stream); the part
  // that draws the last one sums the slabs IN PART ORDER and runs the epilogue.  Nobody waits for anybody.
  float* slab;
  int* tickets;
  int splitk, nk_part;
  unsigned long long* stamps;   // timing builds only (-DPR_TIMING_HOOKS, POSERISK_CONV_STAMPS): s_memrealtime per workgroup
};

#ifdef PR_TIMING_HOOKS
#define PR_CONV_STAMP(k) \
  if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime()
#else
#define PR_CONV_STAMP(k)
#endif

#if defined(__HIP_DEVICE_COMPILE__)
// Tile index -> (tile_m, tile_n) and, for grouped launches, the group's operand bases (x_bytes / w_bytes are per group).
struct TileRef {
  int tile_m, tile_n;
  const float* x;
  const float* w;
  float* y;
};
__device__ __forceinline__ TileRef tile_ref(const DArgs& a, int tile) {
  TileRef t;
  int g = 0;
  if (a.groups > 1) {
    g = tile / a.tiles_per_group;
    tile -= g * a.tiles_per_group;
  }
  t.tile_n = tile % a.tiles_n;
  t.tile_m = tile / a.tiles_n;
  t.x = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.x) + (size_t)g * a.x_bytes);
  t.w = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.w) + (size_t)g * a.w_bytes);
  t.y = a.y + (size_t)g * a.M * a.Cout;
  return t;
}

// Quarter-tile path for the tiles that do not fill a whole round of the 256 CUs.
//
// A launch of T tiles lasts ceil(T/256) tile-times although the CUs carry T/256 on average (784 tiles:
// 16 CUs get a fourth tile, the launch lasts 4 units for 3.06 of work).  The remainder tiles are cut into
// four 32x32 quarters, each computed by one 4-wave workgroup whose waves own 16x16 outputs on
// v_mfma_f32_16x16x4_f32, so the remainder spreads over 4x as many CUs in units of a quarter.
// Results are bit-identical to the whole-tile path: both MFMAs are ascending-k fmaf chains
// (scripts/micro/t_mfma_chain.hip) and the k order fed here is the whole-tile path's order
// (per 8 k-values: 0,4,1,5 | 2,6,3,7), so a frame's bits do not depend on which path its rows take.
// Same LDS-DMA staging and swizzle as the main path with 8 KB stages (32 A rows + 32 B rows), four of
// them in a ring, because a quarter's K-step has only 8 MFMAs to hide the DMA latency behind.
template <int KS, int TAP, bool DUAL>
__device__ __forceinline__ void conv_tail_quarter(const DArgs& a, int item, char* smem) {
  constexpr int NW = 4, NST = 4, STAGE = 64 * 128, A_BYTES = 32 * 128;
  // A quarter's wave has a quarter of a whole-tile wave's MFMAs per K-step but the same number of K-steps and
  // barriers; at equal priority the SIMD hands it one turn per turn of its neighbours and it finishes last.
  __builtin_amdgcn_s_setprio(3);
  const int quarter = item & 3;
  const TileRef tr = tile_ref(a, a.n_full + (item >> 2));
  const int m0 = tr.tile_m * 64 + 32 * (quarter >> 1), n0 = tr.tile_n * 64 + 32 * (quarter & 1);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int q = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
  const auto xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(tr.x), 0, (int)a.x_bytes, 0x00020000);
  const auto wsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(tr.w), 0, (int)a.w_bytes, 0x00020000);
  const int r = 8 * wave + (lane >> 3);   // this lane's row of the 32-row stage (A and B alike)
  int a_base, a_hi0, a_wi0;
  [[maybe_unused]] int a_base2 = (int)kOOB;
  [[maybe_unused]] const auto xsrc2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(DUAL ? a.x2 : a.x), 0,
                                                                          DUAL ? (int)a.x2_bytes : 0, 0x00020000);
  {
    const int m = m0 + r;
    if (m < a.M) {
      const int img = m / a.HoWo, rem = m - img * a.HoWo;
      const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
      a_hi0 = ho * a.stride - a.pad;
      a_wi0 = wo * a.stride - a.pad;
      a_base = (((img * a.H + a_hi0) * a.W + a_wi0) * a.Cin + (TAP == 2 ? 0 : q * 4)) * 4;
      if (DUAL) a_base2 = (((img * a.H2 + ho * a.stride2) * a.W2 + wo * a.stride2) * a.Cin2 + q * 4) * 4;
    } else {
      a_hi0 = -(1 << 28);
      a_wi0 = 0;
      a_base = (int)kOOB;
    }
  }
  const unsigned b_off = (unsigned)(((n0 + r) * a.Kpad + q * 4) * 4);

  auto issue = [&](int kt) {
    char* stage = smem + (kt & (NST - 1)) * STAGE;
    lds_void* adst = (lds_void*)(stage + wave * 1024);
    if (TAP == 0) {
      if (DUAL && kt >= a.nk1)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc2, adst, 16, (unsigned)a_base2, (kt - a.nk1) * 128, 0, 0);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, adst, 16, (unsigned)a_base, kt * 128, 0, 0);
    } else {
      const int k = kt * BK + (TAP == 2 ? q * 4 : 0);
      const int tap = TAP == 2 ? (int)(((unsigned)k * a.cin_magic) >> 20) : k >> a.log2Cin;
      const int ci = TAP == 2 ? k - tap * a.Cin : k & (a.Cin - 1);
      const int kh = tap / KS, kw = tap - kh * KS;
      const int koff = ((kh * a.W + kw) * a.Cin + ci) * 4;
      const bool ok = (TAP == 1 || k < a.K) && (unsigned)(a_hi0 + kh) < (unsigned)a.H &&
                      (unsigned)(a_wi0 + kw) < (unsigned)a.W;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, adst, 16, ok ? (unsigned)(a_base + koff) : kOOB, 0, 0, 0);
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrc, (lds_void*)(stage + A_BYTES + wave * 1024), 16, b_off, kt * 128,
                                             0, 0);
  };

  // fragment reads: lane group g = lane>>4 supplies, per 8 k-values, elements g>>1 and 2+(g>>1) of 16-byte chunk
  // (g&1).  The whole chunk is read (one ds_read_b128, conflict-free under the row swizzle like the whole tile's
  // reads) and the two elements are picked in registers: 128-byte rows are exactly the 32 banks of the 4-byte reads, so
  // two-dword reads (ds_read2_b32) of 16 rows x 2 chunks could never touch more than 8 banks -- a 4-way conflict on
  // every read (SQ_LDS_BANK_CONFLICT 1.8 M per launch against 50 k without quarters), which made the LDS, not the
  // matrix pipe, the quarter workgroups' bound.
  const int g = lane >> 4, fr = lane & 15;
  const int arow = (wave >> 1) * 16 + fr, brow = (wave & 1) * 16 + fr;
  const bool hi = (g >> 1) != 0;
  int aoff[BK / 8], boff[BK / 8];
#pragma unroll
  for (int kk = 0; kk < BK / 8; ++kk) {
    aoff[kk] = arow * 128 + (((2 * kk + (g & 1)) ^ ((arow >> 1) & 7)) << 4);
    boff[kk] = A_BYTES + brow * 128 + (((2 * kk + (g & 1)) ^ ((brow >> 1) & 7)) << 4);
  }

  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int nk = a.nk;
  struct Frags { float a0[BK / 8], a1[BK / 8], b0[BK / 8], b1[BK / 8]; };
  auto read_frags = [&](int kt, Frags& f) {
    const char* st = smem + (kt & (NST - 1)) * STAGE;
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      const f32x4 va = *reinterpret_cast<const f32x4*>(st + aoff[kk]);
      const f32x4 vb = *reinterpret_cast<const f32x4*>(st + boff[kk]);
      f.a0[kk] = hi ? va[1] : va[0]; f.a1[kk] = hi ? va[3] : va[2];
      f.b0[kk] = hi ? vb[1] : vb[0]; f.b1[kk] = hi ? vb[3] : vb[2];
    }
  };
  // One K-step.  Two DMA instructions per stage and wave, completing in order: stages kt+1, kt+2 are in
  // flight while stage kt is multiplied; the fragments of stage kt+1 are read right after the barrier so
  // that their LDS latency runs under the 8 (dependent) MFMAs of stage kt.
  auto step = [&](int kt, const Frags& cur, Frags& nxt) {
    if (kt + 1 < nk) {
      if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // stage kt+1 is visible to everyone, and everyone has consumed stage kt-1 (its buffer is refilled next)
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kt + 3 < nk) issue(kt + 3);
      read_frags(kt + 1, nxt);
    }
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.a0[kk], cur.b0[kk], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.a1[kk], cur.b1[kk], acc, 0, 0, 0);
    }
  };
  issue(0);
  if (nk > 1) issue(1);
  if (nk > 2) issue(2);
  if (nk > 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (nk > 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  Frags f0, f1;
  read_frags(0, f0);
  for (int kt = 0; kt < nk; kt += 2) {
    step(kt, f0, f1);
    if (kt + 1 < nk) step(kt + 1, f1, f0);
  }

  // epilogue straight from the fragments (a few percent of the launch's outputs): acc[j] = D[4g + j][fr]
  const int col = n0 + (wave & 1) * 16 + fr;
  const float bias = a.bias ? a.bias[col] : 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = m0 + (wave >> 1) * 16 + 4 * g + j;
    if (row >= a.M) continue;
    const long o = (long)row * a.Cout + col;
    float v = acc[j];
    if (a.bias) v += bias;
    if (a.res) v += a.res[o];
    if (a.relu) v = fmaxf(v, 0.f);
    tr.y[o] = v;
  }
}
#endif

// TAP: 0 = 1x1 kernel (k = ci), 1 = one tap per K-step (Cin % 32 == 0), 2 = per-lane tap (Cin < 32)
template <int BM, int BN, int WAVES_M, int WAVES_N, int KS, int TAP, bool DUAL, bool SPLIT = false>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void conv_dma_f32(const DArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)  // the host pass only needs the launch stub (LDS address-space casts
                                     // and gfx950 builtins in the body do not type-check there)
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int MI = WM / 32, NI = WN / 32;
  constexpr int GA = BM / 8, GB = BN / 8;        // 8-row DMA groups per tile
  constexpr int DW = NW;                         // waves that issue the DMA
  constexpr int IA = GA / DW, IB = GB / DW;      // DMA instructions per wave per K-step
  static_assert(GA % DW == 0 && GB % DW == 0 && DW % 2 == 0, "DMA groups must split evenly over waves");
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int nb = a.n_full, bid = blockIdx.x;
  PR_CONV_STAMP(0);
#ifdef PR_TIMING_HOOKS
  if (a.stamps && threadIdx.x == 0) {
    a.stamps[(size_t)blockIdx.x * 8 + 4] = __builtin_amdgcn_s_memtime();
    a.stamps[(size_t)blockIdx.x * 8 + 6] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));     // HW_REG_HW_ID
    a.stamps[(size_t)blockIdx.x * 8 + 7] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));    // HW_REG_XCC_ID
  }
#endif
  if constexpr (BM == 64 && BN == 64 && NW == 4) {
    if (bid >= nb) {
      // quarter-tile blocks: XCD-major like the whole tiles, so the four quarters of a tile share an L2
      const int t = bid - nb, per_xcd = (int)(gridDim.x - nb) >> 3;
      const int item = (t & 7) * per_xcd + (t >> 3);
      if (item < a.n_tail) conv_tail_quarter<KS, TAP, DUAL>(a, item, smem);
      return;
    }
  }
  const int xcd = bid & 7, q8 = nb >> 3, rr = nb & 7;
  const int logical = (xcd < rr ? xcd * (q8 + 1) : rr * (q8 + 1) + (xcd - rr) * q8) + (bid >> 3);
  const int part = SPLIT ? logical % a.splitk : 0;
  const int tile_id = SPLIT ? logical / a.splitk : logical;
  const TileRef tr = tile_ref(a, tile_id);
  const int m0 = tr.tile_m * BM, n0 = tr.tile_n * BN;
  const int k_begin = SPLIT ? min(part * a.nk_part, a.nk) : 0;
  const int k_end = SPLIT ? min(k_begin + a.nk_part, a.nk) : a.nk;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int dw = wave;

  // ---- DMA source addressing -------------------------------------------------------------
  // DMA wave w issues groups g = w + DW*i (same parity as w, DW even), lane covers row 8g + (lane>>3)
  // and physical chunk lane&7, i.e. logical chunk q = (lane&7) ^ ((4g + (lane>>4)) & 7).
  const int q = (lane & 7) ^ ((4 * (dw & 1) + (lane >> 4)) & 7);
  const auto xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(tr.x), 0, (int)a.x_bytes, 0x00020000);
  const auto wsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(tr.w), 0, (int)a.w_bytes, 0x00020000);

  int a_base[IA];  // byte offset of (img, hi0, wi0, ci = 4q); TAP 2: ci = 0
  int a_hi0[IA], a_wi0[IA];
  [[maybe_unused]] int a_base2[IA];   // DUAL: the same rows in the second source
  [[maybe_unused]] const auto xsrc2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(DUAL ? a.x2 : a.x), 0,
                                                                          DUAL ? (int)a.x2_bytes : 0, 0x00020000);
#pragma unroll
  for (int i = 0; i < IA; ++i) {
    const int r = 8 * (dw + DW * i) + (lane >> 3);
    const int m = m0 + r;
    a_base2[i] = (int)kOOB;
    if (m < a.M) {
      const int img = m / a.HoWo, rem = m - img * a.HoWo;
      const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
      a_hi0[i] = ho * a.stride - a.pad;
      a_wi0[i] = wo * a.stride - a.pad;
      a_base[i] = (((img * a.H + a_hi0[i]) * a.W + a_wi0[i]) * a.Cin + (TAP == 2 ? 0 : q * 4)) * 4;
      if (DUAL) a_base2[i] = (((img * a.H2 + ho * a.stride2) * a.W2 + wo * a.stride2) * a.Cin2 + q * 4) * 4;
    } else {
      a_hi0[i] = -(1 << 28);
      a_wi0[i] = 0;
      a_base[i] = (int)kOOB;
    }
  }
  unsigned b_off[IB];
#pragma unroll
  for (int i = 0; i < IB; ++i) {
    const int r = 8 * (dw + DW * i) + (lane >> 3);
    b_off[i] = (unsigned)(((n0 + r) * a.Kpad + q * 4) * 4);
  }

  auto issue = [&](int kt, int buf) {
    char* stage = smem + buf * STAGE;
    if (TAP == 0) {
      if (DUAL && kt >= a.nk1) {      // wave-uniform: the K-steps of the second source
        const int soff = (kt - a.nk1) * 128;
#pragma unroll
        for (int i = 0; i < IA; ++i)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc2, (lds_void*)(stage + (dw + DW * i) * 1024), 16,
                                                   (unsigned)a_base2[i], soff, 0, 0);
      } else {
        const int soff = kt * 128;
#pragma unroll
        for (int i = 0; i < IA; ++i)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, (lds_void*)(stage + (dw + DW * i) * 1024), 16,
                                                   (unsigned)a_base[i], soff, 0, 0);
      }
    } else if (TAP == 1) {
      const int k0 = kt * BK;
      const int tap = k0 >> a.log2Cin, ci0 = k0 & (a.Cin - 1);
      const int kh = tap / KS, kw = tap - kh * KS;
      // the tap offset goes into the (range-checked) vector offset: a_base alone is negative for
      // rows whose window starts in the padding
      const int koff = ((kh * a.W + kw) * a.Cin + ci0) * 4;
#pragma unroll
      for (int i = 0; i < IA; ++i) {
        const bool ok = (unsigned)(a_hi0[i] + kh) < (unsigned)a.H && (unsigned)(a_wi0[i] + kw) < (unsigned)a.W;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, (lds_void*)(stage + (dw + DW * i) * 1024), 16,
                                                 ok ? (unsigned)(a_base[i] + koff) : kOOB, 0, 0, 0);
      }
    } else {
      const int k = kt * BK + q * 4;
      const int tap = (int)(((unsigned)k * a.cin_magic) >> 20), ci = k - tap * a.Cin;
      const int kh = tap / KS, kw = tap - kh * KS;
      const int koff = ((kh * a.W + kw) * a.Cin + ci) * 4;
#pragma unroll
      for (int i = 0; i < IA; ++i) {
        const bool ok = k < a.K && (unsigned)(a_hi0[i] + kh) < (unsigned)a.H &&
                        (unsigned)(a_wi0[i] + kw) < (unsigned)a.W;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, (lds_void*)(stage + (dw + DW * i) * 1024), 16,
                                                 ok ? (unsigned)(a_base[i] + koff) : kOOB, 0, 0, 0);
      }
    }
    const int wsoff = kt * 128;
#pragma unroll
    for (int i = 0; i < IB; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrc, (lds_void*)(stage + A_BYTES + (dw + DW * i) * 1024), 16,
                                               b_off[i], wsoff, 0, 0);
  };

  // One DMA instruction of stage kt: piece < IA is this wave's A group `piece`, the rest its B groups (the interleaved K
  // loop below places the pieces one by one between MFMAs; the address arithmetic of `issue` per piece).
  auto issue_piece = [&](int kt, int buf, int piece) {
    char* stage = smem + buf * STAGE;
    if (piece >= IA) {
      const int i = piece - IA;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrc, (lds_void*)(stage + A_BYTES + (dw + DW * i) * 1024), 16, b_off[i], kt * 128,
                                               0, 0);
      return;
    }
    const int i = piece;
    lds_void* dst = (lds_void*)(stage + (dw + DW * i) * 1024);
    if (TAP == 0) {
      if (DUAL && kt >= a.nk1)      // wave-uniform: the K-steps of the second source
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc2, dst, 16, (unsigned)a_base2[i], (kt - a.nk1) * 128, 0, 0);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, dst, 16, (unsigned)a_base[i], kt * 128, 0, 0);
    } else if (TAP == 1) {
      const int k0 = kt * BK;
      const int tap = k0 >> a.log2Cin, ci0 = k0 & (a.Cin - 1);
      const int kh = tap / KS, kw = tap - kh * KS;
      const int koff = ((kh * a.W + kw) * a.Cin + ci0) * 4;
      const bool ok = (unsigned)(a_hi0[i] + kh) < (unsigned)a.H && (unsigned)(a_wi0[i] + kw) < (unsigned)a.W;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, dst, 16, ok ? (unsigned)(a_base[i] + koff) : kOOB, 0, 0, 0);
    } else {
      const int k = kt * BK + q * 4;
      const int tap = (int)(((unsigned)k * a.cin_magic) >> 20), ci = k - tap * a.Cin;
      const int kh = tap / KS, kw = tap - kh * KS;
      const int koff = ((kh * a.W + kw) * a.Cin + ci) * 4;
      const bool ok = k < a.K && (unsigned)(a_hi0[i] + kh) < (unsigned)a.H && (unsigned)(a_wi0[i] + kw) < (unsigned)a.W;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, dst, 16, ok ? (unsigned)(a_base[i] + koff) : kOOB, 0, 0, 0);
    }
  };

  // ---- fragment read addressing (swizzled) ---------------------------------------------------
  // lane reads row (tile row base + lane&31), k-floats 8kk + 4h .. +3 (h = lane>>5): logical chunk
  // 2kk + h, physical chunk (2kk + h) ^ ((row >> 1) & 7); row bases are multiples of 32.
  const int frow = lane & 31, fh = lane >> 5, fsw = (frow >> 1) & 7;
  int foff[BK / 8];
#pragma unroll
  for (int kk = 0; kk < BK / 8; ++kk) foff[kk] = frow * 128 + (((2 * kk + fh) ^ fsw) << 4);

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  auto compute = [&](int buf) {
    const char* Ab = smem + buf * STAGE + wm * WM * 128;
    const char* Bb = smem + buf * STAGE + A_BYTES + wn * WN * 128;
    // All fragment reads of the stage are issued before the first MFMA (8 x ds_read_b128 for a 32x32
    // wave tile): left to itself the compiler recycles one register set and waits lgkmcnt(0) in front of
    // every group of 4 MFMAs, exposing the LDS latency four times per K-step.
    f32x4 af[BK / 8][MI], bf[BK / 8][NI];
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) af[kk][mi] = *reinterpret_cast<const f32x4*>(Ab + mi * 32 * 128 + foff[kk]);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) bf[kk][ni] = *reinterpret_cast<const f32x4*>(Bb + ni * 32 * 128 + foff[kk]);
    }
    __builtin_amdgcn_sched_barrier(0);  // keep the reads above, the MFMAs below (hipcc sinks them otherwise)
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk][mi][j], bf[kk][ni][j], acc[mi][ni], 0, 0, 0);
  };

  // ---- the K loop's step, interleaved form -------------------------------------------------------------------
  // A wave issues in order: a burst of 8 ds_read_b128 (or 4 LDS-DMA instructions) in front of its MFMAs keeps its next
  // MFMA from issuing for as long as the burst takes to issue, and with 3 - 5 waves per SIMD the matrix pipe then idles
  // ~15 % of the main loop (scripts/micro/t_mfma_rate.hip: barrier + 8 reads + 16 MFMAs per iteration runs the pipe at
  // 0.80 / 0.87 / 0.92 with 1 / 2 / 3 workgroups per CU; the same reads placed one by one BETWEEN the MFMAs 0.99 at any
  // occupancy, with the 4 DMA instructions between MFMAs as well 0.84 / 0.91 / 0.93 against 0.75 / 0.84 / 0.87).  So the
  // fragments of stage kt + 1 are read into a second register set between the MFMAs of stage kt, and the DMA of stage
  // kt + 2 goes (first, for the longest lead) between them too -- into stage kt's buffer, which every wave has finished
  // reading before the barrier (lgkmcnt(0) in front of it).  Same MFMAs in the same order: the same bits.
  struct Frags {
    f32x4 a[BK / 8][MI], b[BK / 8][NI];
  };
  auto read_piece = [&](int buf, int piece, Frags& f) {      // piece = kk * (MI + NI) + r
    const int kk = piece / (MI + NI), r = piece % (MI + NI);
    const char* base = smem + buf * STAGE;
    if (r < MI) f.a[kk][r] = *reinterpret_cast<const f32x4*>(base + wm * WM * 128 + r * 32 * 128 + foff[kk]);
    else f.b[kk][r - MI] = *reinterpret_cast<const f32x4*>(base + A_BYTES + wn * WN * 128 + (r - MI) * 32 * 128 + foff[kk]);
  };
  constexpr int N_RD = (BK / 8) * (MI + NI), N_DMA = IA + IB;
  // PAR = kt & 1: stage kt sits in `cur` (registers), stage kt + 1 in LDS buffer !PAR, stage kt + 2 goes to buffer PAR.
  // do_dma / do_rd are wave-uniform (stage kt + 2 / kt + 1 exists): scalar branches around the pieces, so that ONE loop
  // body serves the whole K loop (peeled tails made the compiler keep a second copy of the accumulators).
  auto step = [&](auto par_c, int kt, bool do_dma, bool do_rd, const Frags& cur, Frags& nxt) {
    constexpr int PAR = decltype(par_c)::value;
    if (do_rd) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // own pieces of stage kt + 1 have landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // own reads of stage kt (buffer PAR) are complete
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int n = 0;
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[kk][mi][j], cur.b[kk][ni][j], acc[mi][ni], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (n < N_DMA) {
              if (do_dma) issue_piece(kt + 2, PAR, n);
              __builtin_amdgcn_sched_barrier(0);
            } else if (n - N_DMA < N_RD) {
              // unconditional: behind the last stage this reads a dead buffer into registers nobody uses (a branch here
              // makes the compiler's lgkmcnt bookkeeping conservative: it then waits for THESE reads in front of MFMAs)
              read_piece(PAR ^ 1, n - N_DMA, nxt);
              __builtin_amdgcn_sched_barrier(0);
            }
            ++n;
          }
    constexpr int N_MF = (BK / 8) * 4 * MI * NI;
    static_assert(N_MF >= N_RD + N_DMA, "more DMA pieces and fragment reads than MFMAs to put them between");
  };

  // The residual chunks this thread will need in the epilogue are requested before the main loop, so
  // they land while the MFMAs run (small tiles only: 4 chunks = 16 VGPRs per thread).
  constexpr int CPR = BN / 4;                      // 16-byte output chunks per row
  constexpr int RCH = (BM * CPR) / (NW * 64);      // chunks per thread
  constexpr bool kPrefetchRes = RCH <= 4 || (BM == 64 && BN == 256);   // the whole-row tile has registers to spare
  f32x4 rpre[kPrefetchRes ? RCH : 1];
  if (kPrefetchRes && !SPLIT && a.res) {
#pragma unroll
    for (int i = 0; i < RCH; ++i) {
      const int idx = tid + i * NW * 64;
      const int r = idx / CPR, cc = idx - r * CPR;
      const int row = m0 + r;
      f32x4 z = {0.f, 0.f, 0.f, 0.f};
      rpre[i] = row < a.M ? *reinterpret_cast<const f32x4*>(a.res + (long)row * a.Cout + n0 + cc * 4) : z;
    }
  }

  if (SPLIT || MI * NI > 2 || kOldKLoop || (k_end & 1)) {     // (an odd number of K-steps: no encoder layer has one)
  if (k_begin < k_end) issue(k_begin, k_begin & 1);
  for (int kt = k_begin; kt < k_end; ++kt) {
    // own DMA of stage kt has landed; after the barrier everyone's has, and everyone has finished
    // reading the other buffer (stage kt-1), so it may be refilled.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + 1 < k_end) issue(kt + 1, (kt + 1) & 1);
#ifdef PR_TIMING_HOOKS
    if (kt == k_begin) PR_CONV_STAMP(1);
#endif
    compute(kt & 1);
  }
  } else {
    // k_begin = 0 here (no split): stage kt lives in LDS buffer kt & 1 and in register set kt & 1
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    Frags f0, f1;
    issue(0, 0);
    if (k_end > 1) {
      issue(1, 1);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IA + IB) : "memory");     // stage 0 has landed (stage 1 may be in flight)
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    PR_CONV_STAMP(1);
#pragma unroll
    for (int pc = 0; pc < N_RD; ++pc) read_piece(0, pc, f0);
    for (int kt = 0; kt < k_end; kt += 2) {       // k_end is even here
      step(P0{}, kt, kt + 2 < k_end, true, f0, f1);
      step(P1{}, kt + 1, kt + 3 < k_end, kt + 2 < k_end, f1, f0);
    }
  }
  PR_CONV_STAMP(2);

  // ---- epilogue through LDS --------------------------------------------------------------------
  // The fp32 tile goes to LDS ([BM][BN+4] floats, reusing the stage buffers) and is read back row-wise:
  // each thread owns 4 consecutive columns, loads the residual as one 16-byte chunk, adds bias and
  // residual, applies ReLU and stores 16 bytes, so rows leave as whole 128-byte lines and the
  // residual read is as coalesced as the store (fragment-layout stores are 128 bytes per row per
  // instruction and needed the residual prefetched into 16 registers per 32x32 block).
  constexpr int CT_STRIDE = BN + 4;
  float* Ct = reinterpret_cast<float*>(smem);
  __syncthreads();  // every wave has finished reading the stage buffers
  {
    const int col_l = lane & 31, row_h = 4 * (lane >> 5);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int r = wm * WM + mi * 32 + row_h + (e & 3) + 8 * (e >> 2);
          Ct[r * CT_STRIDE + wn * WN + ni * 32 + col_l] = acc[mi][ni][e];
        }
  }
  __syncthreads();
  if constexpr (SPLIT) {
    // ---- split-K hand-over (guide, section 5 "in-launch split-K reduction"): plain 16-byte slab stores -> every wave
    // drains its stores -> barrier -> ONE agent-scope release -> ticket.  The part that draws the last ticket acquires
    // once and sums the slabs in part order (so the sum does not depend on who arrived when).
    float* mine = a.slab + ((size_t)tile_id * a.splitk + part) * (BM * BN);
#pragma unroll
    for (int i = 0; i < RCH; ++i) {
      const int idx = tid + i * NW * 64;
      const int r = idx / CPR, cc = idx - r * CPR;
      *reinterpret_cast<f32x4*>(mine + r * BN + cc * 4) = *reinterpret_cast<const f32x4*>(&Ct[r * CT_STRIDE + cc * 4]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* flag = reinterpret_cast<int*>(smem + BM * CT_STRIDE * 4);   // one word behind the staging tile
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      *flag = __hip_atomic_fetch_add(a.tickets + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (*flag != a.splitk - 1) return;       // not the last part of this tile: done (nobody waits)
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const float* base = a.slab + (size_t)tile_id * a.splitk * (BM * BN);
#pragma unroll
    for (int i = 0; i < RCH; ++i) {
      const int idx = tid + i * NW * 64;
      const int r = idx / CPR, cc = idx - r * CPR;
      f32x4 v = *reinterpret_cast<const f32x4*>(base + r * BN + cc * 4);
      for (int sp = 1; sp < a.splitk; ++sp) v += *reinterpret_cast<const f32x4*>(base + (size_t)sp * (BM * BN) + r * BN + cc * 4);
      *reinterpret_cast<f32x4*>(&Ct[r * CT_STRIDE + cc * 4]) = v;      // each thread rewrites only its own chunks
    }
  }
#pragma unroll
  for (int i = 0; i < RCH; ++i) {
    const int idx = tid + i * NW * 64;
    const int r = idx / CPR, cc = idx - r * CPR;
    const int row = m0 + r, col = n0 + cc * 4;
    if (row >= a.M) continue;
    f32x4 v = *reinterpret_cast<const f32x4*>(&Ct[r * CT_STRIDE + cc * 4]);
    if (a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + col);
    const long o = (long)row * a.Cout + col;
    if (a.res) v += (kPrefetchRes && !SPLIT) ? rpre[i] : *reinterpret_cast<const f32x4*>(a.res + o);
    if (a.relu) {
      v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
    }
    *reinterpret_cast<f32x4*>(tr.y + o) = v;
  }
  PR_CONV_STAMP(3);
#ifdef PR_TIMING_HOOKS
  if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 8 + 5] = __builtin_amdgcn_s_memtime();
#endif
#endif  // __HIP_DEVICE_COMPILE__
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int KS, int TAP, bool DUAL = false, bool SPLIT = false>
int launch_one(const DArgs& da, int grid, hipStream_t stream) {
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr size_t lds_stage = (size_t)2 * (BM + BN) * 128, lds_epi = (size_t)BM * (BN + 4) * 4 + (SPLIT ? 16 : 0);
  constexpr size_t lds = lds_stage > lds_epi ? lds_stage : lds_epi;
  void (*kern)(const DArgs) = conv_dma_f32<BM, BN, WAVES_M, WAVES_N, KS, TAP, DUAL, SPLIT>;
  static std::atomic<uint64_t> attr_done{0};  // per instantiation, one bit per device
  PR_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, attr_done));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, stream, da);
  return check_launch("conv_dma_f32");
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
int launch_dma(const DArgs& da, int ks, int tap, int grid, hipStream_t stream) {
  if (da.splitk > 1) {
    // split-K launches come from the encoder plan only (64x64 tile, 1x1 or one-tap-per-step 3x3, single source)
    if constexpr (BM == 64 && BN == 64 && WAVES_M == 2 && WAVES_N == 2) {
      if (tap == 0 && !da.x2) return launch_one<BM, BN, WAVES_M, WAVES_N, 1, 0, false, true>(da, grid, stream);
      if (tap == 1 && ks == 3) return launch_one<BM, BN, WAVES_M, WAVES_N, 3, 1, false, true>(da, grid, stream);
    }
    set_error("conv_dma: split-K runs on the 64x64 tile, 1x1 (one source) or 3x3 with Cin %% 32 == 0");
    return PR_ERR_INVALID;
  }
  if (tap == 0 && da.x2) {
    // dual-source launches come from the encoder plan only, which runs fp32 on the 64x64 tile
    if constexpr (BM == 64 && BN == 64 && WAVES_M == 2 && WAVES_N == 2)
      return launch_one<BM, BN, WAVES_M, WAVES_N, 1, 0, true>(da, grid, stream);
    set_error("conv_dma: dual-source launches run on the 64x64 tile only");
    return PR_ERR_INVALID;
  }
  if (tap == 0) return launch_one<BM, BN, WAVES_M, WAVES_N, 1, 0>(da, grid, stream);
  if (tap == 1 && ks == 3) return launch_one<BM, BN, WAVES_M, WAVES_N, 3, 1>(da, grid, stream);
  if (tap == 2 && ks == 7) return launch_one<BM, BN, WAVES_M, WAVES_N, 7, 2>(da, grid, stream);
  if (tap == 2 && ks == 3) return launch_one<BM, BN, WAVES_M, WAVES_N, 3, 2>(da, grid, stream);
  if (tap == 2 && ks == 4) {   // the stem after space-to-depth: 4x4 taps of 12 channels
    if constexpr (BM == 64 && BN == 64 && WAVES_M == 2 && WAVES_N == 2)
      return launch_one<BM, BN, WAVES_M, WAVES_N, 4, 2>(da, grid, stream);
  }
  set_error("conv_dma: unsupported kernel size %d / tap mode %d", ks, tap);
  return PR_ERR_INVALID;
}

int ilog2_exact(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return (1 << l) == v ? l : -1;
}

}  // namespace

int conv_dma_launch(const ConvProblem& p, int BM, int BN, hipStream_t stream, int threads) {
  PR_REQUIRE(p.KH == p.KW, "conv: square kernels only");
  PR_REQUIRE(p.Cin % 4 == 0 && p.Cout % BN == 0, "conv: bad channels Cin=%d Cout=%d (tile N %d)", p.Cin, p.Cout, BN);
  const int K2 = p.x2 ? p.Cin2 : 0;   // second source (1x1): its channels extend the K loop
  const size_t xb = (size_t)p.B * p.H * p.W * p.Cin * 4, wb = (size_t)p.Cout * (p.Kpad() + K2) * 4;
  const size_t x2b = p.x2 ? (size_t)p.B * p.H2 * p.W2 * p.Cin2 * 4 : 0;
  PR_REQUIRE(xb < (1ull << 31) && wb < (1ull << 31) && (size_t)p.M() * p.Cout < (1ull << 31),
             "conv: tensor too large for one launch (%zu input bytes)", xb);
  const int l2 = ilog2_exact(p.Cin);
  int tap;
  if (p.KH == 1 && p.pad == 0) tap = 0;
  else if (p.Cin % BK == 0 && l2 >= 0) tap = 1;
  else tap = 2;
  PR_REQUIRE(tap != 1 || l2 >= 0, "conv: the one-tap-per-K-step path needs power-of-two Cin (%d)", p.Cin);
  PR_REQUIRE(tap != 0 || p.Cin % BK == 0, "conv: 1x1 path needs Cin %% 32 == 0 (%d)", p.Cin);
  unsigned magic = (1u << 20) / (unsigned)p.Cin + 1;
  if (tap == 2)
    for (int k = 0; k < p.Kpad() + BK; k += 4)
      PR_REQUIRE((int)(((unsigned)k * magic) >> 20) == k / p.Cin, "conv: Cin %d / K %d outside the tap decode's range", p.Cin, p.K());
  if (p.x2) {
    PR_REQUIRE(tap == 0 && p.groups == 1 && p.Cin2 % BK == 0 && p.stride2 > 0 && x2b < (1ull << 31),
               "conv: a second source needs a 1x1 conv and Cin2 %% 32 == 0 (%d)", p.Cin2);
    PR_REQUIRE((p.H2 - 1) / p.stride2 + 1 == p.Ho && (p.W2 - 1) / p.stride2 + 1 == p.Wo,
               "conv: second source %dx%d / stride %d does not land on the %dx%d output", p.H2, p.W2, p.stride2, p.Ho, p.Wo);
  }
  DArgs da;
  da.x = p.x; da.w = p.w; da.bias = p.bias; da.res = p.res; da.y = p.y;
  da.x_bytes = (unsigned)xb; da.w_bytes = (unsigned)wb;
#ifdef PR_TIMING_HOOKS
  // Ablation builds only (POSERISK_CXXFLAGS=-DPR_TIMING_HOOKS; never in the shipped library): a zero-record descriptor
  // drops that operand's loads (zeros are written to LDS) while the instruction stream stays; results are wrong.
  if (const char* e = getenv("POSERISK_DEBUG_DROP")) {
    if (atoi(e) & 1) da.x_bytes = 0;
    if (atoi(e) & 2) da.w_bytes = 0;
  }
#endif
  da.H = p.H; da.W = p.W; da.Cin = p.Cin; da.log2Cin = l2 < 0 ? 0 : l2; da.cin_magic = magic;
  da.Ho = p.Ho; da.Wo = p.Wo; da.HoWo = p.Ho * p.Wo; da.Cout = p.Cout; da.stride = p.stride; da.pad = p.pad;
  da.M = p.M(); da.K = p.K() + K2; da.Kpad = p.Kpad() + K2; da.nk = da.Kpad / BK;
  da.x2 = p.x2; da.x2_bytes = (unsigned)x2b; da.H2 = p.H2; da.W2 = p.W2; da.Cin2 = p.Cin2; da.stride2 = p.stride2;
  da.nk1 = p.Kpad() / BK;
  da.tiles_n = p.Cout / BN;
  da.relu = p.relu;
  if (da.M == 0) return PR_OK;
  da.groups = p.groups;
  da.tiles_per_group = ceil_div(da.M, BM) * da.tiles_n;
  if (p.groups > 1)
    PR_REQUIRE(tap == 0 && !p.bias && !p.res && !p.relu, "conv: grouped launches are plain 1x1 GEMMs");
  int grid = da.tiles_per_group * p.groups;
  da.n_full = grid;
  da.n_tail = 0;
  da.splitk = 1; da.nk_part = da.nk; da.slab = nullptr; da.tickets = nullptr;
  if (p.splitk > 1) {
    PR_REQUIRE(BM == 64 && BN == 64 && p.groups == 1 && p.split_slab && p.split_tickets,
               "conv: split-K needs the 64x64 tile and a workspace");
    da.splitk = p.splitk;
    da.nk_part = ceil_div(da.nk, p.splitk);
    da.slab = p.split_slab;
    da.tickets = p.split_tickets;
    PR_HIP(hipMemsetAsync(p.split_tickets, 0, (size_t)grid * sizeof(int), stream));
    grid *= p.splitk;
    da.n_full = grid;
  }
  // Tile quantisation (256 CUs): the tiles beyond the last whole round of 256 run as quarter tiles when that
  // shortens the launch (conv_tail_quarter; same bits).  ConvTuning::tail = 0 turns it off for A/B timing.
  if (p.tune.tail && BM == 64 && BN == 64 && threads == 256 && da.splitk == 1) {
    // A quarter block needs as many K-steps as a whole tile and each of them costs it a DMA round trip, so it
    // only disappears behind the whole tiles when they run for at least two rounds (measured: 784 tiles
    // 152 -> 137 us, 392 tiles 154 -> 175 us); with at most 64 tiles every quarter gets a CU to itself.
    const int min_rounds = p.tune.tail_min_rounds, max_rem = p.tune.tail_max_rem;
    const int rem = grid % 256, rounds = grid / 256;
    if ((rounds >= min_rounds && rem > 0 && rem <= max_rem) || grid <= 64) {
      da.n_full = grid - rem;
      da.n_tail = 4 * rem;
      grid = da.n_full + ceil_div(da.n_tail, 8) * 8;
    }
  }
  da.stamps = nullptr;
#ifdef PR_TIMING_HOOKS
  // Timing builds: the 20th launch of a process records per-workgroup s_memrealtime stamps (0 entry, 1 first stage landed,
  // 2 main loop done, 3 stores issued; 4 / 5 s_memtime at entry / exit; 6 HW_ID, 7 XCC_ID) and writes them to the file.
  static unsigned long long* stamp_buf = nullptr;
  static int stamp_calls = 0;
  const char* stamp_path = getenv("POSERISK_CONV_STAMPS");
  const bool stamp_now = stamp_path && grid <= 16384 && ++stamp_calls == 20;
  if (stamp_now) {
    if (!stamp_buf) PR_HIP(hipMalloc(&stamp_buf, (size_t)16384 * 8 * 8));
    PR_HIP(hipMemsetAsync(stamp_buf, 0, (size_t)16384 * 8 * 8, stream));
    da.stamps = stamp_buf;
  }
  struct StampDump {
    bool on; const char* path; int grid; hipStream_t s; unsigned long long* buf;
    ~StampDump() {
      if (!on) return;
      std::vector<unsigned long long> host((size_t)grid * 8);
      (void)hipStreamSynchronize(s);
      (void)hipMemcpy(host.data(), buf, host.size() * 8, hipMemcpyDeviceToHost);
      if (FILE* fo = fopen(path, "wb")) { fwrite(host.data(), 8, host.size(), fo); fclose(fo); }
    }
  } stamp_dump{stamp_now, stamp_path, grid, stream, stamp_buf};
#endif
  const int key = BM * 1000 + BN + (threads == 512 && BM == 128 ? 500000 : 0) + (threads == 128 ? 900000 : 0);
  switch (key) {
    case 128128: return launch_dma<128, 128, 2, 2>(da, p.KH, tap, grid, stream);
    case 628128: return launch_dma<128, 128, 4, 2>(da, p.KH, tap, grid, stream);
    case 628064: return launch_dma<128, 64, 4, 2>(da, p.KH, tap, grid, stream);
    case 964064: return launch_dma<64, 64, 2, 1>(da, p.KH, tap, grid, stream);
    case 1028064: return launch_dma<128, 64, 2, 1>(da, p.KH, tap, grid, stream);
    case 964128: return launch_dma<64, 128, 1, 2>(da, p.KH, tap, grid, stream);
    case 128064: return launch_dma<128, 64, 2, 2>(da, p.KH, tap, grid, stream);
    case 64064: return launch_dma<64, 64, 2, 2>(da, p.KH, tap, grid, stream);
    case 256128: return launch_dma<256, 128, 4, 2>(da, p.KH, tap, grid, stream);
    case 64128: return launch_dma<64, 128, 2, 2>(da, p.KH, tap, grid, stream);
    case 256064: return launch_dma<256, 64, 4, 2>(da, p.KH, tap, grid, stream);
    case 64256: return launch_dma<64, 256, 2, 2>(da, p.KH, tap, grid, stream);
  }
  set_error("conv_dma: no %dx%d tile", BM, BN);
  return PR_ERR_INVALID;
}

}  // namespace pr
