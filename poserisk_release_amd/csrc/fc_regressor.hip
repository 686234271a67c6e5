// The regressor's fully connected layers (SPIN models/hmr.py: fc1, fc2, decpose/decshape/deccam; call site
// lib/core/base.py:220): y[M][N] = x[M][K] * W[N][K]^T + bias + res with M = the frames of a batch.
//
// As 64x64 MFMA tiles these GEMMs are one long dependent chain per output: K = 2048 is 64 K-steps on 64 of the 256 CUs,
// 13-16 us per launch and ten launches per batch (138 us of a 4.4 ms step at B=64, rocprofv3 kernel trace).  Here an
// output tile is 16x16 (v_mfma_f32_16x16x4_f32, 8-10 cycles per k instead of 32), a workgroup's four waves each take a
// quarter of K straight from L2/HBM into registers (no LDS staging: nothing is shared between waves, so no barriers;
// 16-byte loads, two sets of four iterations in flight) and meet once in LDS, summed in wave order.  The split is fixed
// (four waves, whatever the batch), so a frame's bits do not depend on its batch or position.
#include "conv_igemm.h"
#include "frame_kernels.h"

namespace pr {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

// grid (N / (16 NT), ceil(M / (16 MT))), 256 threads.  K % 64 == 0, N % (16 NT) == 0; x rows are K floats apart, y / res rows N floats.
// MT x NT output tiles of 16x16 per workgroup (round 5, built to test whether the launch is traffic-bound at M = 256, where the
// one-tile form is 1 024 workgroups that each stream 32 rows of K floats for ONE tile: 134 MB of L2 traffic per K = 1024
// launch, 13 us, ten launches per batch; it is not -- see the launcher).  A wave's x fragments are shared by NT weight
// fragments and the other way round: 16 (MT + NT) rows for MT NT tiles.  Every output is still its wave's ascending-k chain over a quarter of K and the same
// four-term sum, so the bits do not depend on (MT, NT) -- and therefore not on the batch the shape is chosen by.
template <int MT, int NT>
__global__ __launch_bounds__(256) void fc_rows16_f32(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, const float* res, float* y, int M,
                                                     int N, int K) {
  __shared__ float red[4][MT * NT][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.y * (16 * MT), n0 = blockIdx.x * (16 * NT);
  const int kq = K >> 2;                                     // this wave's share of K
  const float* xp[MT];
  const float* wp[NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = min(m0 + 16 * mt + r, M - 1);            // rows past M repeat the last one; they are not stored
    xp[mt] = x + (size_t)row * K + wave * kq + 4 * g;
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) wp[nt] = w + (size_t)(n0 + 16 * nt + r) * K + wave * kq + 4 * g;
  const int iters = kq >> 4;                                 // 16 k per iteration: lane group g holds k = 16 j + 4 g + t
  constexpr int U = MT * NT > 4 ? 2 : 4;                     // iterations in flight per register set
  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 a0[U][MT], b0[U][NT], a1[U][MT], b1[U][NT];
  auto load = [&](f32x4 (*a)[MT], f32x4 (*b)[NT], int j0) {
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (j0 + u < iters) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[u][mt] = *reinterpret_cast<const f32x4*>(xp[mt] + 16 * (j0 + u));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[u][nt] = *reinterpret_cast<const f32x4*>(wp[nt] + 16 * (j0 + u));
      }
  };
  auto fma = [&](const f32x4 (*a)[MT], const f32x4 (*b)[NT], int j0) {
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (j0 + u < iters) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][mt][t], b[u][nt][t], acc[mt][nt], 0, 0, 0);
      }
  };
  load(a0, b0, 0);
  for (int j0 = 0; j0 < iters; j0 += 2 * U) {
    load(a1, b1, j0 + U);
    fma(a0, b0, j0);
    load(a0, b0, j0 + 2 * U);
    fma(a1, b1, j0 + U);
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int i = 0; i < 4; ++i) red[wave][mt * NT + nt][i * 64 + lane] = acc[mt][nt][i];
  __syncthreads();
  // wave v finishes tiles v, v + 4, ...: the four partial sums in wave order, bias, residual
  for (int tile = wave; tile < MT * NT; tile += 4) {
    const int mt = tile / NT, nt = tile - mt * NT;
    const int n = n0 + 16 * nt + r;
    const float b = bias ? bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + 16 * mt + 4 * g + i;                // acc[i] = D[4 g + i][r]
      if (m >= M) continue;
      float v = ((red[0][tile][i * 64 + lane] + red[1][tile][i * 64 + lane]) + red[2][tile][i * 64 + lane]) + red[3][tile][i * 64 + lane];
      v += b;
      if (res) v += res[(size_t)m * N + n];
      y[(size_t)m * N + n] = v;
    }
  }
}

}  // namespace

// shape: 0 = by the batch (below), else 10 MT + NT (tests and A/B timing: POSERISK_FC_SHAPE)
int launch_fc_rows16(const float* x, const float* w, const float* bias, const float* res, float* y, int M, int N, int K,
                     hipStream_t s, int shape) {
  PR_REQUIRE(x && w && y && M >= 0 && N > 0 && N % 16 == 0 && K > 0 && K % 64 == 0,
             "fc: needs K %% 64 == 0 and N %% 16 == 0 (M %d, N %d, K %d)", M, N, K);
  if (M == 0) return PR_OK;
  // One tile per workgroup whatever the batch.  MEASURED (profiles/r05_experiments.txt section 4): 2 x 2 / 4 x 2 / 4 x 1 / 2 x 1 /
  // 1 x 2 tiles halve the operand traffic of a 256-frame launch and change nothing end to end -- bf16 B = 256 reads 90.1 -
  // 90.8 k frames/s for every shape (one batch in flight 85.3 - 85.6 k; 4 x 2 loses 1 %), fp32 B = 256 and B = 64 likewise: the
  // ten launches overlap the other lane's convolutions and are latency-, not traffic-bound.  The shapes stay (same bits,
  // POSERISK_FC_SHAPE) for A/B runs on other batch sizes.
  int MT = 1, NT = 1;
  if (shape > 0) { MT = shape / 10; NT = shape % 10; }
  PR_REQUIRE((MT == 1 || MT == 2 || MT == 4) && (NT == 1 || NT == 2) && N % (16 * NT) == 0, "fc: tile shape %d x %d for N %d", MT, NT, N);
  const dim3 grid(N / (16 * NT), ceil_div(M, 16 * MT));
#define PR_FC_CASE(A, B) \
  if (MT == A && NT == B) hipLaunchKernelGGL((fc_rows16_f32<A, B>), grid, dim3(256), 0, s, x, w, bias, res, y, M, N, K)
  PR_FC_CASE(1, 1); else PR_FC_CASE(2, 1); else PR_FC_CASE(4, 1); else PR_FC_CASE(1, 2); else PR_FC_CASE(2, 2); else PR_FC_CASE(4, 2);
#undef PR_FC_CASE
  return check_launch("fc_rows16_f32");
}

}  // namespace pr
