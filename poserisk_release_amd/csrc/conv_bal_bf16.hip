// bf16 implicit-GEMM convolution with the work dealt to the CUs by hand: one persistent workgroup per CU, each with an
// equal share of the pixels.  (SPIN models/hmr.py Bottleneck conv1 / conv2 + BatchNorm + ReLU; call site
// lib/core/base.py:220.)
//
// Why: the tile kernel (conv_dma_bf16.hip) launches one workgroup per 128x128 tile and the dispatcher deals them round
// robin.  At B = 256 every stage of the network has 49 * 2^k rows per CU, so a launch is 3.06 (or 1.53, 6.12) tiles per
// CU: 16 CUs carry a fourth tile while 240 idle through it, and an MFMA-bound launch lasts 4 tile times for 3.06 of work
// (scripts/micro/t_wg_placement.hip: balance 0.766).  Here the pixel tiles (32 pixels) are cut into contiguous runs of
// T/runs +- 1 tiles, one run per workgroup, and a workgroup walks its run in chunks of up to 8 (12) pixel tiles against a
// block of 256 (128) output channels: twice the tile kernel's operand reuse, which is what lets ONE workgroup per CU keep
// the matrix pipes fed out of 160 KB of LDS.
//
// Structure (eight waves, two per SIMD):
//   * a wave owns <= 4 (3) pixel tiles x 2 channel tiles: <= 8 accumulator tiles, 6 fragment reads per 8 MFMAs;
//   * operands reach LDS by LDS-DMA in stages of 32 k-values (64-byte rows, 16 rows per 1 KB piece, XOR swizzle on the
//     source side: the 16-byte slot p of row r holds logical slot p ^ ((r >> 2) & 3), which makes the ds_read_b128 of a
//     32-row fragment conflict free); a ring of R stages runs R - 2 stages ahead behind counted vmcnt waits, one barrier
//     per stage; the ring never drains: the stages of the next chunk follow the last stage of this one;
//   * waves 4-7 (the SIMD partners of waves 0-3) run HALF A STAGE behind: when one partner waits at the barrier, or for
//     the first fragments of a freshly landed stage, the other has its MFMAs queued.  In lockstep the two stall together
//     (MI355X guide, 'Two waves per SIMD', item 9);
//   * the MFMAs are transposed (weights are the A operand, their rows permuted by sigma at the DMA's source side): a lane
//     ends up with 16 consecutive channels of one pixel, so bias / ReLU / the bf16 rounding happen in registers and y
//     leaves as 16-byte stores, no LDS transpose, while the ring keeps loading the next chunk.
// Same products, same k order, same epilogue arithmetic as conv_dma_bf16: bit-identical outputs
// (tests/test_hip_parity.py::test_conv_bal_bf16_equals_tile_kernel).
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
typedef __attribute__((address_space(3))) void lds_void;

[[maybe_unused]] constexpr unsigned kOOB = 0x80000000u;
constexpr int kStageBytes = 512 * 64;   // 512 rows (pixels + channels) of 32 k-values

struct BalArgs {
  const unsigned short* x;
  const unsigned short* w;
  const float* bias;
  unsigned short* y;
  unsigned x_bytes, w_bytes, y_bytes;
  int H, W, Cin, log2Cin, Ho, Wo, HoWo, Cout, stride, pad;
  int M, Kpad, ns;       // ns = stages of 32 k-values per chunk
  int T, NB, runs;       // pixel tiles of 32, channel blocks, pixel runs (= workgroups / NB)
  int relu;
  // second pixel source (1x1 only; a first block's downsample branch in conv3's K loop): stages [ns1, ns) read row m's
  // channels from x2, a [B,H2,W2,Cin2] tensor sampled at (ho * stride2, wo * stride2); ns1 = ns without one
  const unsigned short* x2;
  unsigned x2_bytes;
  int H2, W2, Cin2, stride2, ns1;
  unsigned long long* stamps;   // timing builds only (-DPR_TIMING_HOOKS, POSERISK_BAL_STAMPS): s_memtime at six points of intervals 8 .. 23
};

__device__ inline unsigned pack2(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}

__device__ inline void wait_vm(int n) {   // s_waitcnt vmcnt(n), n in {0, 1, 4, 8, 12} (anything else: 0, which is stricter)
  switch (n) {
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// PQ pixel groups x CP = 8 / PQ channel pairs of waves.  PQ = 2: chunks of <= 8 pixel tiles x 256 channels;
// PQ = 4: <= 12 pixel tiles x 128 channels.  TAP: 0 = 1x1 kernel, 1 = KS x KS kernel with one tap per stage.
// R = ring stages.  DBG (timing builds only, -DPR_TIMING_HOOKS; results are wrong when set): 1 no LDS-DMA after the
// prologue, 2 no MFMAs, 4 no fragment reads, 8 no stores, 16 no stagger (waves 4-7 in step with waves 0-3).
// PAIR: the two stages of a 64-k group (the two 64-byte halves of the same 128-byte lines) are issued together, their
// pieces alternating, every second interval -- a line's second half then meets its first half in the vector cache (or in
// its miss queue) instead of being fetched from L2 again a whole stage later, when 64 KB of other lines have passed
// through the 32 KB cache.  Needs R = 5.
template <int PQ, int KS, int TAP, int R, int DBG = 0, bool PAIR = false>
__global__ __launch_bounds__(512) void conv_bal_bf16(const BalArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int CP = 8 / PQ;                  // channel pairs (64 channels each)
  constexpr int PXT = PQ == 2 ? 4 : 3;        // pixel tiles per group
  constexpr int GROWS = 32 * PXT;             // LDS rows of a pixel group
  constexpr int PXR = PQ * GROWS;             // pixel rows of a stage (256 / 384)
  constexpr int CHR = 64 * CP;                // channel rows (256 / 128)
  static_assert(PXR + CHR == 512, "a stage is 512 rows");
  constexpr int MAXC = PQ * PXT;              // pixel tiles per chunk
  constexpr int PXP = PXR / 128, CHP = CHR / 128;   // DMA pieces (16 rows) per wave per stage
  constexpr int D = R - 2;                    // stages in flight ahead of the one waves 0-3 compute
  static_assert(!PAIR || R == 5, "paired stages need a ring of five");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wg = blockIdx.x;
  const int cb = (wg >> 3) % a.NB;            // channel block; partners (same pixels) sit 8 apart: one XCD
  const int run = (wg / (8 * a.NB)) * 8 + (wg & 7);
  if (run >= a.runs) return;
  const int t_begin = (int)((long)a.T * run / a.runs), t_end = (int)((long)a.T * (run + 1) / a.runs);
  const int ntiles = t_end - t_begin;
  if (ntiles <= 0) return;
  const int nchunks = (ntiles + MAXC - 1) / MAXC;
  const int cbase = ntiles / nchunks, cextra = ntiles - cbase * nchunks;   // chunk c has cbase + (c < cextra) tiles

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = PQ == 2 ? wave >> 2 : wave >> 1;      // pixel group
  const int cp = PQ == 2 ? wave & 3 : wave & 1;         // channel pair
  const bool lag = (DBG & 16) ? false : wave >= 4;
  const int i = lane & 31, h = lane >> 5;
  const int n0 = cb * CHR;
  auto STAMP = [&](int t, int k) {
#ifdef PR_TIMING_HOOKS
    if (a.stamps && t >= 8 && t < 24 && (threadIdx.x & 63) == 0)
      a.stamps[(((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + (t - 8)) * 6 + k] = __builtin_amdgcn_s_memtime();
#endif
  };

  const auto xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  const auto wsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.w), 0, (int)a.w_bytes, 0x00020000);
  [[maybe_unused]] const auto xsrc2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.x2 ? a.x2 : a.x), 0,
                                                                          a.x2 ? (int)a.x2_bytes : 0, 0x00020000);
  const auto ysrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)a.y_bytes, 0x00020000);

  // ---- DMA source addressing.  A piece is 16 LDS rows of 64 bytes; lane l writes row l >> 2, slot l & 3 of its piece,
  // which holds logical slot q.  Wave w issues pixel pieces w + 8 j and channel pieces w + 8 j.
  const int q = (lane & 3) ^ ((lane >> 4) & 3);
  unsigned b_off[CHP];
#pragma unroll
  for (int j = 0; j < CHP; ++j) {
    const int r = 16 * (wave + 8 * j) + (lane >> 2);     // LDS row of the channel part <-> MFMA row r & 31 of tile r >> 5
    const int ri = r & 31;
    const int ch = n0 + (r & ~31) + 16 * ((ri >> 2) & 1) + 4 * (ri >> 3) + (ri & 3);   // sigma
    b_off[j] = (unsigned)((ch * a.Kpad + q * 8) * 2);
  }
  float bias_r[2][16];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int e = 0; e < 16; ++e) bias_r[c][e] = a.bias ? a.bias[n0 + 64 * cp + 32 * c + 16 * h + e] : 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing of the set-up may be counted among the ring's operations

  // chunk geometry: chunk c covers tiles [first, first + n); group g takes cnt(g) of them from start(g)
  auto chunk_first = [&](int c) { return t_begin + c * cbase + (c < cextra ? c : cextra); };
  auto chunk_tiles = [&](int c) { return cbase + (c < cextra ? 1 : 0); };
  auto grp_cnt = [&](int n, int g) { return n / PQ + (g < n % PQ ? 1 : 0); };
  auto grp_start = [&](int n, int g) { return g * (n / PQ) + (g < n % PQ ? g : n % PQ); };

  // ---- issue side: runs D stages ahead of the compute side, through the chunks of the run without a gap
  int a_base[PXP], a_hi0[PXP], a_wi0[PXP];
  [[maybe_unused]] int a_base2[PXP];     // the same rows in the second source (TAP 0)
  auto setup_rows = [&](int c) {
    const int first = chunk_first(c), n = chunk_tiles(c);
#pragma unroll
    for (int j = 0; j < PXP; ++j) {
      const int row = 16 * (wave + 8 * j) + (lane >> 2);
      const int g = row / GROWS, local = row - g * GROWS;
      const int m = (first + grp_start(n, g)) * 32 + local;
      const bool valid = (local >> 5) < grp_cnt(n, g) && m < a.M;
      if (valid) {
        const int img = m / a.HoWo, rem = m - img * a.HoWo;
        const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
        a_hi0[j] = ho * a.stride - a.pad;
        a_wi0[j] = wo * a.stride - a.pad;
        a_base[j] = (((img * a.H + a_hi0[j]) * a.W + a_wi0[j]) * a.Cin + q * 8) * 2;
        if (TAP == 0) a_base2[j] = a.x2 ? (((img * a.H2 + ho * a.stride2) * a.W2 + wo * a.stride2) * a.Cin2 + q * 8) * 2 : (int)kOOB;
      } else {
        a_hi0[j] = -(1 << 28);
        a_wi0[j] = 0;
        a_base[j] = (int)kOOB;
        if (TAP == 0) a_base2[j] = (int)kOOB;
      }
    }
  };
  int is_chunk = 0, is_stage = 0, is_buf = 0;
  auto issue_next = [&]() {              // ALWAYS PXP + CHP = 4 pieces while stages remain (invalid rows read as zeros)
    if (is_chunk >= nchunks) return;
    if (!((DBG & 1) && (is_chunk > 0 || is_stage >= D))) {
      char* stage = smem + is_buf * kStageBytes;
      if (TAP == 0) {
        if (is_stage < a.ns1) {
#pragma unroll
          for (int j = 0; j < PXP; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, (lds_void*)(stage + (wave + 8 * j) * 1024), 16, (unsigned)a_base[j],
                                                     is_stage * 64, 0, 0);
        } else {                         // wave-uniform: the stages of the second source
#pragma unroll
          for (int j = 0; j < PXP; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc2, (lds_void*)(stage + (wave + 8 * j) * 1024), 16, (unsigned)a_base2[j],
                                                     (is_stage - a.ns1) * 64, 0, 0);
        }
      } else {
        // slice-major K (conv_k_index_bf16): stage s is half s & 1 of tap (s / 2) % KS^2 of the 64-channel slice s / (2 KS^2)
        const int g64 = is_stage >> 1;
        const int tap = g64 % (KS * KS), ci0 = (g64 / (KS * KS)) * 64 + (is_stage & 1) * 32;
        const int kh = tap / KS, kw = tap - kh * KS;
        const int koff = ((kh * a.W + kw) * a.Cin + ci0) * 2;
#pragma unroll
        for (int j = 0; j < PXP; ++j) {
          const bool ok = (unsigned)(a_hi0[j] + kh) < (unsigned)a.H && (unsigned)(a_wi0[j] + kw) < (unsigned)a.W;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, (lds_void*)(stage + (wave + 8 * j) * 1024), 16,
                                                   ok ? (unsigned)(a_base[j] + koff) : kOOB, 0, 0, 0);
        }
      }
#pragma unroll
      for (int j = 0; j < CHP; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrc, (lds_void*)(stage + PXR * 64 + (wave + 8 * j) * 1024), 16, b_off[j],
                                                 is_stage * 64, 0, 0);
    }
    is_buf = is_buf + 1 == R ? 0 : is_buf + 1;
    if (++is_stage == a.ns) {
      is_stage = 0;
      if (++is_chunk < nchunks) setup_rows(is_chunk);
    }
  };

  // PAIR: stages is_stage and is_stage + 1 (same tap, adjacent 64-byte halves) into buffers is_buf, is_buf + 1; 8 pieces,
  // alternating, so that stage is_stage's last piece has exactly ONE piece behind it
  auto issue_pair = [&]() {
    if (is_chunk >= nchunks) return;
    if (!((DBG & 1) && (is_chunk > 0 || is_stage >= 2))) {
      char* st0 = smem + is_buf * kStageBytes;
      char* st1 = smem + (is_buf + 1 >= R ? is_buf + 1 - R : is_buf + 1) * kStageBytes;
      int kh = 0, kw = 0, koff = 0;
      if (TAP != 0) {
        const int g64 = is_stage >> 1;
        const int tap = g64 % (KS * KS), ci0 = (g64 / (KS * KS)) * 64;   // is_stage is even here
        kh = tap / KS;
        kw = tap - kh * KS;
        koff = ((kh * a.W + kw) * a.Cin + ci0) * 2;
      }
#pragma unroll
      for (int j = 0; j < PXP; ++j) {
        unsigned v;
        int so;
        if (TAP == 0) {
          v = (unsigned)a_base[j];
          so = is_stage * 64;
        } else {
          const bool ok = (unsigned)(a_hi0[j] + kh) < (unsigned)a.H && (unsigned)(a_wi0[j] + kw) < (unsigned)a.W;
          v = ok ? (unsigned)(a_base[j] + koff) : kOOB;
          so = 0;
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, (lds_void*)(st0 + (wave + 8 * j) * 1024), 16, v, so, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, (lds_void*)(st1 + (wave + 8 * j) * 1024), 16, v, so + 64, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < CHP; ++j) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrc, (lds_void*)(st0 + PXR * 64 + (wave + 8 * j) * 1024), 16, b_off[j],
                                                 is_stage * 64, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrc, (lds_void*)(st1 + PXR * 64 + (wave + 8 * j) * 1024), 16, b_off[j],
                                                 is_stage * 64 + 64, 0, 0);
      }
    }
    is_buf = is_buf + 2 >= R ? is_buf + 2 - R : is_buf + 2;
    is_stage += 2;
    if (is_stage == a.ns) {
      is_stage = 0;
      if (++is_chunk < nchunks) setup_rows(is_chunk);
    }
  };

  // ---- compute side
  int foff[2];     // lane reads row (tile base + i), logical slot 2 kk + h
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) foff[kk] = i * 64 + (((2 * kk + h) ^ ((i >> 2) & 3)) << 4);
  const int px_row0 = grp * GROWS * 64, ch_row0 = PXR * 64 + cp * 64 * 64;

  struct Frags {
    bf16x8 w[2], p[PXT];
    int sel;   // the k-step the fragments belong to (read by experiment builds only)
  };
  f32x16 acc[PXT][2];
  Frags fa, fb;
  int g_stage = 0;                      // global stage (over the chunks of the run) the next interval computes
  const int S = nchunks * a.ns;

  // One chunk with NP pixel tiles in this wave: a.ns + 1 intervals, one barrier each.  Waves 0-3 compute stage s in
  // interval s and store in the last one; waves 4-7 compute the second half of stage s - 1 and the first half of stage s
  // in interval s, and the last half stage, then store, in the last one.  In interval s everyone issues the stage D
  // ahead into the buffer of stage s - 2, which waves 4-7 finished reading in interval s - 1.  Before the barrier of
  // stage s a wave waits for its own pieces of that stage: all but the (D - 1) stages issued behind it.  (The epilogue's
  // stores are not counted: the next wait then covers them as well, once per chunk.)
  auto chunk = [&](auto np_c, int c_idx) {
    constexpr int NP = decltype(np_c)::value;
    auto read_frags = [&](Frags& f, int buf, int kk) {
      const char* st = smem + buf * kStageBytes;
      f.sel = kk;
      if (DBG & 4) {
#pragma unroll
        for (int c = 0; c < 2; ++c) f.w[c] = __builtin_bit_cast(bf16x8, u32x4{(unsigned)lane, (unsigned)kk, (unsigned)c, 2u});
#pragma unroll
        for (int pt = 0; pt < NP; ++pt) f.p[pt] = __builtin_bit_cast(bf16x8, u32x4{(unsigned)lane, (unsigned)pt, (unsigned)buf, 2u});
        return;
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) f.w[c] = *reinterpret_cast<const bf16x8*>(st + ch_row0 + c * (32 * 64) + foff[kk]);
#pragma unroll
      for (int pt = 0; pt < NP; ++pt) f.p[pt] = *reinterpret_cast<const bf16x8*>(st + px_row0 + pt * (32 * 64) + foff[kk]);
    };
    auto mfmas = [&](const Frags& f) {
#pragma unroll
      for (int pt = 0; pt < NP; ++pt)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          if (DBG & 2) asm volatile("" ::"v"(f.w[c]), "v"(f.p[pt]));
          else acc[pt][c] = mfma_bf16_step(f.w[c], f.p[pt], acc[pt][c], f.sel);
        }
    };
    auto gate = [&]() {                 // the wait for this wave's pieces of stage g_stage, then the barrier
      const int ahead = S - 1 - g_stage;
      STAMP(g_stage, 0);
      if (PAIR) {
        // an even stage has its partner's last piece behind it; an odd one the eight pieces of the next pair, issued one
        // interval ago (if there is a next pair)
        wait_vm(ahead < 0 ? 0 : (g_stage & 1) ? (ahead > 0 ? 8 : 0) : 1);
      } else {
        wait_vm(4 * (ahead < D - 1 ? (ahead < 0 ? 0 : ahead) : D - 1));
      }
      STAMP(g_stage, 1);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      STAMP(g_stage, 2);
    };
#pragma unroll
    for (int pt = 0; pt < NP; ++pt)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[pt][c][e] = 0.f;
    int buf = g_stage % R;
    if (!lag) {
      for (int s = 0; s < a.ns; ++s) {
        gate();
        if (!PAIR) issue_next();
        else if (!(s & 1)) issue_pair();
        STAMP(g_stage, 3);
        read_frags(fa, buf, 0);
        read_frags(fb, buf, 1);
        STAMP(g_stage, 4);
        mfmas(fa);
        mfmas(fb);
        STAMP(g_stage, 5);
        ++g_stage;
        buf = buf + 1 == R ? 0 : buf + 1;
      }
      gate();
    } else {
      for (int s = 0; s < a.ns; ++s) {
        gate();
        read_frags(fb, buf, 0);          // stage s is in since this barrier
        if (s > 0) mfmas(fa);            // second half of stage s - 1: its fragments were read before the barrier
        STAMP(g_stage, 3);
        if (!PAIR) issue_next();
        else if (!(s & 1)) issue_pair();
        STAMP(g_stage, 4);
        read_frags(fa, buf, 1);          // for the next interval
        mfmas(fb);
        STAMP(g_stage, 5);
        ++g_stage;
        buf = buf + 1 == R ? 0 : buf + 1;
      }
      gate();
      mfmas(fa);
    }
    // epilogue: lane (i, h) holds channels n0 + 64 cp + 32 c + 16 h .. + 15 of pixel (first tile of the group + pt) * 32 + i
    const int n = chunk_tiles(c_idx);
    const int m0 = (chunk_first(c_idx) + grp_start(n, grp)) * 32 + i;
#pragma unroll
    for (int pt = 0; pt < NP; ++pt) {
      const int m = m0 + 32 * pt;
      const unsigned yoff = m < a.M ? (unsigned)(m * (2 * a.Cout) + 32 * h) : kOOB;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        unsigned pk[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float v0 = acc[pt][c][2 * e] + bias_r[c][2 * e], v1 = acc[pt][c][2 * e + 1] + bias_r[c][2 * e + 1];
          if (a.relu) {
            v0 = fmaxf(v0, 0.f);
            v1 = fmaxf(v1, 0.f);
          }
          pk[e] = pack2(v0, v1);
        }
        const int soff = (n0 + 64 * cp + 32 * c) * 2;
        if (DBG & 8) {
          asm volatile("" ::"v"(pk[0]), "v"(pk[1]), "v"(pk[2]), "v"(pk[3]), "v"(pk[4]), "v"(pk[5]), "v"(pk[6]), "v"(pk[7]));
        } else {
          buffer_store_b128_sreg(u32x4{pk[0], pk[1], pk[2], pk[3]}, ysrc, yoff, soff);
          buffer_store_b128_sreg(u32x4{pk[4], pk[5], pk[6], pk[7]}, ysrc, yoff + 16, soff);
        }
      }
    }
  };

  setup_rows(0);
  if (PAIR) {
    issue_pair();
  } else {
    for (int d = 0; d < D; ++d) issue_next();
  }
  for (int c = 0; c < nchunks; ++c) {
    switch (grp_cnt(chunk_tiles(c), grp)) {
      case 4:
        if constexpr (PXT >= 4) chunk(std::integral_constant<int, (PXT >= 4 ? 4 : 0)>{}, c);
        break;
      case 3: chunk(std::integral_constant<int, 3>{}, c); break;
      case 2: chunk(std::integral_constant<int, 2>{}, c); break;
      case 1: chunk(std::integral_constant<int, 1>{}, c); break;
      default: chunk(std::integral_constant<int, 0>{}, c); break;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

int ilog2_exact_c(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return (1 << l) == v ? l : -1;
}

template <int PQ, int R, int DBG = 0, bool PAIR = false>
int launch_bal(const BalArgs& a, int ks, int grid, hipStream_t stream) {
  constexpr int lds = R * kStageBytes;
  void (*k)(const BalArgs) = ks == 1 ? conv_bal_bf16<PQ, 1, 0, R, DBG, PAIR> : conv_bal_bf16<PQ, 3, 1, R, DBG, PAIR>;
  static std::atomic<uint64_t> done1{0}, done3{0};
  PR_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(k), lds, ks == 1 ? done1 : done3));
  hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, stream, a);
  return check_launch("conv_bal_bf16");
}

}  // namespace

bool conv_bal_bf16_fits(const ConvProblem& p) {
  if (p.precision != 1 || p.groups != 1 || p.w3 || p.res || p.KH != p.KW) return false;
  if (p.Cout % 128 || p.Cin % 64) return false;
  if (p.x2 && (p.KH != 1 || p.Cin2 % 64 || p.stride2 <= 0)) return false;   // a second source: 1x1 only
  if (p.KH == 1) return p.pad == 0;
  return p.KH == 3;
}

// Where it pays.  MEASURED inside the encoder at B = 256, same box, per layer (profiles/r03_conv_bal.txt): the layers with
// 256-channel blocks and runs of >= 6 pixel tiles gain 7-13 % (layer3's conv1 / conv2, the first conv1 of layer3 and
// layer4); with 3 tiles per run the weights a workgroup streams per chunk outweigh its pixels (layer4: 0.8x the tile
// kernel stand-alone).  Of the layers with 128 output channels only layer2's first conv1 gains (4 %); the others read
// a tensor the expansion kernel has just written and lose 2-12 % against the tile kernel's order of tiles.
// The outputs are the tile kernel's bit for bit, so the choice may depend on the batch.
bool conv_bal_bf16_pays(const ConvProblem& p, int cus) {
  if (!conv_bal_bf16_fits(p)) return false;
  const int nb = p.Cout / (p.Cout % 256 == 0 ? 256 : 128);
  const int runs = std::max(cus / (8 * nb), 1) * 8;
  if (ceil_div(p.M(), 32) < 6 * runs) return false;
  return p.Cout % 256 == 0 || (p.KH == 1 && p.Cin <= 256);
}

int conv_bal_bf16_launch(const ConvProblem& p, hipStream_t stream, int variant) {
  PR_REQUIRE(conv_bal_bf16_fits(p), "conv_bal_bf16: bf16, no residual, 1x1 (pad 0) or 3x3, Cin %% 64 == 0, "
             "Cout %% 128 == 0; got %dx%d Cin=%d Cout=%d", p.KH, p.KW, p.Cin, p.Cout);
  const int K2 = p.x2 ? p.Cin2 : 0;
  const int K = p.K(), Kpad1 = ceil_div(K, 64) * 64, Kpad = Kpad1 + K2;
  const size_t xb = (size_t)p.B * p.H * p.W * p.Cin * 2, wb = (size_t)p.Cout * Kpad * 2, yb = (size_t)p.M() * p.Cout * 2;
  const size_t x2b = p.x2 ? (size_t)p.B * p.H2 * p.W2 * p.Cin2 * 2 : 0;
  PR_REQUIRE(xb < (1ull << 31) && wb < (1ull << 31) && yb < (1ull << 31) && x2b < (1ull << 31),
             "conv_bal_bf16: tensor too large for one launch (%zu bytes)", std::max(xb, yb));
  if (p.x2)
    PR_REQUIRE((p.H2 - 1) / p.stride2 + 1 == p.Ho && (p.W2 - 1) / p.stride2 + 1 == p.Wo,
               "conv_bal_bf16: second source %dx%d / stride %d does not land on the %dx%d output", p.H2, p.W2, p.stride2, p.Ho, p.Wo);
  if (p.M() == 0) return PR_OK;
  // channel blocks of 256 where the layer has them (variant 1 forces blocks of 128)
  const bool wide = p.Cout % 256 == 0 && variant != 1;
  BalArgs a;
  a.x = reinterpret_cast<const unsigned short*>(p.x); a.w = reinterpret_cast<const unsigned short*>(p.w); a.bias = p.bias;
  a.y = reinterpret_cast<unsigned short*>(p.y);
  a.x_bytes = (unsigned)xb; a.w_bytes = (unsigned)wb; a.y_bytes = (unsigned)yb;
  a.H = p.H; a.W = p.W; a.Cin = p.Cin; a.log2Cin = std::max(ilog2_exact_c(p.Cin), 0);
  a.Ho = p.Ho; a.Wo = p.Wo; a.HoWo = p.Ho * p.Wo; a.Cout = p.Cout; a.stride = p.stride; a.pad = p.pad;
  a.M = p.M(); a.Kpad = Kpad; a.ns = Kpad / 32;
  a.T = ceil_div(a.M, 32); a.NB = p.Cout / (wide ? 256 : 128); a.relu = p.relu;
  a.x2 = reinterpret_cast<const unsigned short*>(p.x2); a.x2_bytes = (unsigned)x2b;
  a.H2 = p.H2; a.W2 = p.W2; a.Cin2 = p.Cin2; a.stride2 = p.stride2; a.ns1 = Kpad1 / 32;
  int cus = 256;
  PR_TRY(current_device_cus(&cus));
  // workgroups: a multiple of 8 NB (channel-block partners sit 8 apart, on one XCD), at most one per CU, and no more
  // pixel runs than pixel tiles
  const int per = 8 * a.NB;
  int groups = std::max(cus / per, 1);
  groups = std::min(groups, std::max(a.T / 8, 1));
  a.runs = groups * 8;
  const int grid = groups * per;
  a.stamps = nullptr;
#ifdef PR_TIMING_HOOKS
  static unsigned long long* stamp_buf = nullptr;
  static int stamp_calls = 0;
  const char* stamp_path = getenv("POSERISK_BAL_STAMPS");
  const size_t stamp_n = (size_t)256 * 8 * 16 * 6;
  if (stamp_path) {
    if (!stamp_buf) PR_HIP(hipMalloc(&stamp_buf, stamp_n * 8));
    PR_HIP(hipMemsetAsync(stamp_buf, 0, stamp_n * 8, stream));
    a.stamps = stamp_buf;
    if (++stamp_calls == 20) {   // a warm launch in the middle of the timing loop
      const int st = wide ? launch_bal<2, 4>(a, p.KH, grid, stream) : launch_bal<4, 4>(a, p.KH, grid, stream);
      std::vector<unsigned long long> host(stamp_n);
      PR_HIP(hipStreamSynchronize(stream));
      PR_HIP(hipMemcpy(host.data(), stamp_buf, stamp_n * 8, hipMemcpyDeviceToHost));
      if (FILE* f = fopen(stamp_path, "wb")) {
        fwrite(host.data(), 8, stamp_n, f);
        fclose(f);
      }
      return st;
    }
  }
  if (const char* e = getenv("POSERISK_BAL_DBG")) {
    switch (atoi(e)) {
      case 1: return wide ? launch_bal<2, 5, 1>(a, p.KH, grid, stream) : launch_bal<4, 5, 1>(a, p.KH, grid, stream);
      case 2: return wide ? launch_bal<2, 5, 2>(a, p.KH, grid, stream) : launch_bal<4, 5, 2>(a, p.KH, grid, stream);
      case 4: return wide ? launch_bal<2, 5, 4>(a, p.KH, grid, stream) : launch_bal<4, 5, 4>(a, p.KH, grid, stream);
      case 5: return wide ? launch_bal<2, 5, 5>(a, p.KH, grid, stream) : launch_bal<4, 5, 5>(a, p.KH, grid, stream);
      case 6: return wide ? launch_bal<2, 5, 6>(a, p.KH, grid, stream) : launch_bal<4, 5, 6>(a, p.KH, grid, stream);
      case 8: return wide ? launch_bal<2, 5, 8>(a, p.KH, grid, stream) : launch_bal<4, 5, 8>(a, p.KH, grid, stream);
      case 13: return wide ? launch_bal<2, 5, 13>(a, p.KH, grid, stream) : launch_bal<4, 5, 13>(a, p.KH, grid, stream);
      case 16: return wide ? launch_bal<2, 5, 16>(a, p.KH, grid, stream) : launch_bal<4, 5, 16>(a, p.KH, grid, stream);
      case 104: return wide ? launch_bal<2, 4>(a, p.KH, grid, stream) : launch_bal<4, 4>(a, p.KH, grid, stream);
      case 106: return wide ? launch_bal<2, 5, 6, true>(a, p.KH, grid, stream) : launch_bal<4, 5, 6, true>(a, p.KH, grid, stream);
      default: break;
    }
  }
#endif
  if (p.tune.bal_stages == 6 && !p.x2) return wide ? launch_bal<2, 5, 0, true>(a, p.KH, grid, stream) : launch_bal<4, 5, 0, true>(a, p.KH, grid, stream);
  // ring of 4 stages (128 KB) by default: 5 (all 160 KB) measured the same stand-alone and in the pipeline
  if (p.tune.bal_stages == 5) return wide ? launch_bal<2, 5>(a, p.KH, grid, stream) : launch_bal<4, 5>(a, p.KH, grid, stream);
  return wide ? launch_bal<2, 4>(a, p.KH, grid, stream) : launch_bal<4, 4>(a, p.KH, grid, stream);
}

}  // namespace pr
