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

// grid (N/16, ceil(M/16)), 256 threads.  K % 64 == 0, N % 16 == 0; x rows are K floats apart, y / res rows N floats.
__global__ __launch_bounds__(256) void fc_rows16_f32(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, const float* res, float* y, int M,
                                                     int N, int K) {
  __shared__ float red[4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 16;
  const int kq = K >> 2;                                     // this wave's share of K
  const int row = min(m0 + r, M - 1);                        // rows past M repeat the last one; they are not stored
  const float* xp = x + (size_t)row * K + wave * kq + 4 * g;
  const float* wp = w + (size_t)(n0 + r) * K + wave * kq + 4 * g;
  const int iters = kq >> 4;                                 // 16 k per iteration: lane group g holds k = 16 j + 4 g + t
  constexpr int U = 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  f32x4 a0[U], b0[U], a1[U], b1[U];
  auto load = [&](f32x4* a, f32x4* b, int j0) {
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (j0 + u < iters) {
        a[u] = *reinterpret_cast<const f32x4*>(xp + 16 * (j0 + u));
        b[u] = *reinterpret_cast<const f32x4*>(wp + 16 * (j0 + u));
      }
  };
  auto fma = [&](const f32x4* a, const f32x4* b, int j0) {
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (j0 + u < iters) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][t], b[u][t], acc, 0, 0, 0);
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
  for (int i = 0; i < 4; ++i) red[wave][i * 64 + lane] = acc[i];
  __syncthreads();
  if (wave == 0) {
    const int n = n0 + r;
    const float b = bias ? bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + 4 * g + i;                          // acc[i] = D[4 g + i][r]
      if (m >= M) continue;
      float v = ((red[0][i * 64 + lane] + red[1][i * 64 + lane]) + red[2][i * 64 + lane]) + red[3][i * 64 + lane];
      v += b;
      if (res) v += res[(size_t)m * N + n];
      y[(size_t)m * N + n] = v;
    }
  }
}

}  // namespace

int launch_fc_rows16(const float* x, const float* w, const float* bias, const float* res, float* y, int M, int N, int K,
                     hipStream_t s) {
  PR_REQUIRE(x && w && y && M >= 0 && N > 0 && N % 16 == 0 && K > 0 && K % 64 == 0,
             "fc: needs K %% 64 == 0 and N %% 16 == 0 (M %d, N %d, K %d)", M, N, K);
  if (M == 0) return PR_OK;
  hipLaunchKernelGGL(fc_rows16_f32, dim3(N / 16, ceil_div(M, 16)), dim3(256), 0, s, x, w, bias, res, y, M, N, K);
  return check_launch("fc_rows16_f32");
}

}  // namespace pr
