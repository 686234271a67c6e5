// Winograd F(2x2,3x3) for the encoder's 3x3 / stride-1 / pad-1 fp32 convolutions (SPIN Bottleneck conv2 of
// layer2..layer4; call site lib/core/base.py:220).
//
//   V_k[p][ci]  = (B^T d B)_k        one 4x4 input patch d per 2x2 output tile p, k = 4i + j          (pass 1)
//   M_k[p][co]  = sum_ci V_k[p][ci] U_k[co][ci]       16 independent GEMMs, ONE grouped launch of the
//                                                     fp32 MFMA kernel (conv_dma.hip, groups = 16)     (pass 2)
//   y           = A^T M A + bias, ReLU                2x2 outputs per tile                              (pass 3)
//   U_k         = (G g G^T)_k        per (co, ci), folded with the BatchNorm scale in double on the host
//
// 36 multiplies per tile and channel pair become 16: 2.25x fewer MFMA FLOPs (1.72x at 7x7, where the 4x4 tiles
// cover 8x8).  The encoder is matrix-pipe / power bound, so the two extra streaming passes
// (V and M: 4x the activation's bytes each way) are paid from idle HBM bandwidth.  The transforms are adds in a
// fixed order and every tile is computed on its own, so a frame's bits still do not depend on its batch or
// position.  fp32 error of the whole encoder with these layers in Winograd form: within reordering noise of the
// direct form (3e-6 on the rotation matrices, measured against the CPU oracle).
#include <algorithm>

#include "conv_igemm.h"

namespace pr {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

struct WinoArgs {
  const float* x;     // [B,H,W,C]
  float* v;           // [16][P][C]
  const float* m;     // [16][P][Cout]
  const float* bias;  // [Cout] or nullptr
  float* y;           // [B,H,W,Cout]
  int B, H, W, C, Cout, th, tw, relu;
  long P;
};

// One thread per (tile p, 4 channels): 16 x 16-byte loads (zero outside the image), 32 + 32 vector adds,
// 16 x 16-byte stores; consecutive threads cover consecutive channels, so every V_k row is written whole.
__global__ __launch_bounds__(256) void wino_input_transform(const WinoArgs a) {
  const int c4n = a.C >> 2;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= a.P * c4n) return;
  const long p = idx / c4n;
  const int c = (int)(idx - p * c4n) * 4;
  const int tx = (int)(p % a.tw);
  const long q = p / a.tw;
  const int ty = (int)(q % a.th), img = (int)(q / a.th);
  const int h0 = 2 * ty - 1, w0 = 2 * tx - 1;
  f32x4 d[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int hi = h0 + i, wi = w0 + j;
      const bool ok = (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      d[i][j] = ok ? *reinterpret_cast<const f32x4*>(a.x + (((long)img * a.H + hi) * a.W + wi) * a.C + c) : z;
    }
  // t = B^T d (rows), v = t B (columns);  B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
  f32x4 t[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    t[0][j] = d[0][j] - d[2][j];
    t[1][j] = d[1][j] + d[2][j];
    t[2][j] = d[2][j] - d[1][j];
    t[3][j] = d[1][j] - d[3][j];
  }
  const long gs = a.P * a.C;   // floats per V_k
  float* out = a.v + p * a.C + c;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    *reinterpret_cast<f32x4*>(out + (4 * i + 0) * gs) = t[i][0] - t[i][2];
    *reinterpret_cast<f32x4*>(out + (4 * i + 1) * gs) = t[i][1] + t[i][2];
    *reinterpret_cast<f32x4*>(out + (4 * i + 2) * gs) = t[i][2] - t[i][1];
    *reinterpret_cast<f32x4*>(out + (4 * i + 3) * gs) = t[i][1] - t[i][3];
  }
}

// One thread per (tile p, 4 output channels): y = A^T m A + bias, ReLU;  A^T = [1 1 1 0; 0 1 -1 -1].
__global__ __launch_bounds__(256) void wino_output_transform(const WinoArgs a) {
  const int c4n = a.Cout >> 2;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= a.P * c4n) return;
  const long p = idx / c4n;
  const int c = (int)(idx - p * c4n) * 4;
  const int tx = (int)(p % a.tw);
  const long q = p / a.tw;
  const int ty = (int)(q % a.th), img = (int)(q / a.th);
  const long gs = a.P * a.Cout;
  const float* in = a.m + p * a.Cout + c;
  f32x4 m[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) m[i][j] = *reinterpret_cast<const f32x4*>(in + (4 * i + j) * gs);
  f32x4 s[2][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    s[0][j] = (m[0][j] + m[1][j]) + m[2][j];
    s[1][j] = (m[1][j] - m[2][j]) - m[3][j];
  }
  f32x4 b = {0.f, 0.f, 0.f, 0.f};
  if (a.bias) b = *reinterpret_cast<const f32x4*>(a.bias + c);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ho = 2 * ty + i;
    if (ho >= a.H) continue;
    f32x4 o[2];
    o[0] = ((s[i][0] + s[i][1]) + s[i][2]) + b;
    o[1] = ((s[i][1] - s[i][2]) - s[i][3]) + b;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int wo = 2 * tx + j;
      if (wo >= a.W) continue;
      f32x4 v = o[j];
      if (a.relu) {
        v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
      }
      *reinterpret_cast<f32x4*>(a.y + (((long)img * a.H + ho) * a.W + wo) * a.Cout + c) = v;
    }
  }
}

long wino_tiles(const ConvProblem& p) { return (long)p.B * ((p.H + 1) / 2) * ((p.W + 1) / 2); }

}  // namespace

size_t conv_winograd_work_floats(const ConvProblem& p) {
  return (size_t)16 * wino_tiles(p) * ((size_t)p.Cin + p.Cout);
}

void conv_winograd_pack_weights(const float* w, const double* scale, int Cout, int Cin, float* out) {
  // G = [1 0 0; 1/2 1/2 1/2; 1/2 -1/2 1/2; 0 0 1];  U = G g G^T in double, one rounding to fp32
  static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
  for (int o = 0; o < Cout; ++o)
    for (int ci = 0; ci < Cin; ++ci) {
      const float* g = w + ((size_t)o * Cin + ci) * 9;
      const double sc = scale ? scale[o] : 1.0;
      double t[4][3];
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 3; ++j)
          t[i][j] = G[i][0] * ((double)g[j] * sc) + G[i][1] * ((double)g[3 + j] * sc) + G[i][2] * ((double)g[6 + j] * sc);
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
          const double u = t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2];
          out[((size_t)(4 * i + j) * Cout + o) * Cin + ci] = (float)u;
        }
    }
}

int conv_winograd_launch(const ConvProblem& p, const float* u, float* work, hipStream_t stream) {
  PR_REQUIRE(p.precision == 0 && p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && !p.res,
             "winograd: 3x3 / stride 1 / pad 1 fp32 convolutions without residual only");
  PR_REQUIRE(p.Cin % kConvBK == 0 && p.Cout % 64 == 0, "winograd: Cin %% 32 and Cout %% 64 (got %d, %d)", p.Cin, p.Cout);
  PR_REQUIRE(p.x && p.y && u && work, "winograd: null tensor");
  if (p.B == 0) return PR_OK;
  WinoArgs a;
  a.x = p.x; a.bias = p.bias; a.y = p.y;
  a.B = p.B; a.H = p.H; a.W = p.W; a.C = p.Cin; a.Cout = p.Cout; a.relu = p.relu;
  a.th = (p.H + 1) / 2; a.tw = (p.W + 1) / 2;
  a.P = wino_tiles(p);
  a.v = work;
  float* m = work + (size_t)16 * a.P * p.Cin;
  a.m = m;
  PR_REQUIRE(a.P * std::max(p.Cin, p.Cout) < (1L << 29), "winograd: %ld tiles are too many for one launch", a.P);
  const long n_in = a.P * (p.Cin / 4), n_out = a.P * (p.Cout / 4);
  hipLaunchKernelGGL(wino_input_transform, dim3((unsigned)ceil_div(n_in, 256L)), dim3(256), 0, stream, a);
  PR_TRY(check_launch("wino_input_transform"));
  ConvProblem g;
  g.x = work; g.w = u; g.bias = nullptr; g.res = nullptr; g.y = m;
  g.B = (int)a.P; g.H = g.W = g.Ho = g.Wo = 1; g.Cin = p.Cin; g.Cout = p.Cout;
  g.KH = g.KW = 1; g.stride = 1; g.pad = 0; g.relu = 0; g.precision = 0;
  g.groups = 16;
  PR_TRY(conv_dma_launch(g, 64, 64, stream, 256));
  hipLaunchKernelGGL(wino_output_transform, dim3((unsigned)ceil_div(n_out, 256L)), dim3(256), 0, stream, a);
  return check_launch("wino_output_transform");
}

}  // namespace pr
