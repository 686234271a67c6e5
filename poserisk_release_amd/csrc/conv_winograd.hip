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
//
// Two output-tile sizes: F(2x2,3x3) as above (16 products per 4 outputs) and F(4x4,3x3) (m = 4: 6x6 patches, 36
// products per 16 outputs: 4x fewer MFMA FLOPs than the direct form and V/M only 2.25x the activation's size).  The
// larger transform's constants (Lavin & Gray: B^T up to 5, A^T up to 8, G down to 1/24) cost nothing measurable here:
// against an fp64 run of the same network the fp32 encoder is 2.4e-6 off on the rotation matrices in direct form and
// 1.8e-6 with F(4x4,3x3) layers.
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
  const long idx = xcd_contiguous_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x;  // overlapping patches: one L2
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
  const long idx = xcd_contiguous_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x;  // overlapping patches: one L2
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


using f32x2 = __attribute__((ext_vector_type(2))) float;

// ---- F(4x4,3x3): one thread per (tile, 2 channels); 36 values live, both passes in place ---------------------
// Two sets of interpolation points (Cook-Toom on 0, +-a, +-b, infinity):
//   PTS 0 (conv form 4): a = 1, b = 2 -- Lavin & Gray's matrices,
//     B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   PTS 1 (conv form 5): a = 11/16, b = 3/2 -- points nearly reciprocal to each other keep every entry of B^T and A^T
//     within [1/3, 3.4] (Lavin's reach 5 and 8), which halves the fp32 error of a layer (scripts/wino_points.py: rms
//     error of one 3x3 layer against fp64, in units of the direct fp32 convolution's: 10.7 -> 5.4; F(2x2): 2.6).  All
//     constants are dyadic rationals, exact in fp32, so A^T [(G g) * (B^T d)] is the convolution exactly:
//     row(0) = [a^2 b^2, 0, -(a^2+b^2), 0, 1, 0]   row(+-a) = [0, -+a b^2, -b^2, +-a, 1, 0]
//     row(+-b) = [0, -+b a^2, -a^2, +-b, 1, 0]      row(inf) = [0, a^2 b^2, 0, -(a^2+b^2), 0, 1]
// kWa = 11/16, kWb = 3/2: host_plan.h (the weights' G matrix is built from the same constants)
constexpr float kWa2 = kWa * kWa, kWb2 = kWb * kWb, kWab2 = kWa2 * kWb2, kWs2 = kWa2 + kWb2;

template <int PTS, typename V>
__device__ __forceinline__ void bt6(V& a0, V& a1, V& a2, V& a3, V& a4, V& a5) {
  const V d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5;
  if constexpr (PTS == 0) {
    a0 = 4.f * d0 - 5.f * d2 + d4;
    a1 = (d3 + d4) - 4.f * (d1 + d2);
    a2 = 4.f * (d1 - d2) + (d4 - d3);
    a3 = 2.f * (d3 - d1) + (d4 - d2);
    a4 = 2.f * (d1 - d3) + (d4 - d2);
    a5 = 4.f * d1 - 5.f * d3 + d5;
  } else {
    const V ea = d4 - kWb2 * d2, oa = kWa * (d3 - kWb2 * d1);   // even / odd parts of the rows of +-a
    const V eb = d4 - kWa2 * d2, ob = kWb * (d3 - kWa2 * d1);   // ... of +-b
    a0 = (kWab2 * d0 - kWs2 * d2) + d4;
    a1 = ea + oa;
    a2 = ea - oa;
    a3 = eb + ob;
    a4 = eb - ob;
    a5 = (kWab2 * d1 - kWs2 * d3) + d5;
  }
}

// V = f32x2 or f32x4: channels per thread (POSERISK_WINO_VEC; round 6 A/B of 16-byte accesses in the two passes)
template <int PTS, typename V>
__global__ __launch_bounds__(256) void wino43_input_transform(const WinoArgs a) {
  constexpr int NC = sizeof(V) / 4;
  const int c2n = a.C / NC;
  const long idx = xcd_contiguous_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x;  // overlapping patches: one L2
  if (idx >= a.P * c2n) return;
  const long p = idx / c2n;
  const int c = (int)(idx - p * c2n) * NC;
  const int tx = (int)(p % a.tw);
  const long q = p / a.tw;
  const int ty = (int)(q % a.th), img = (int)(q / a.th);
  const int h0 = 4 * ty - 1, w0 = 4 * tx - 1;
  V d[6][6];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int hi = h0 + i, wi = w0 + j;
      const bool ok = (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
      V z;
#pragma unroll
      for (int e = 0; e < NC; ++e) z[e] = 0.f;
      d[i][j] = ok ? *reinterpret_cast<const V*>(a.x + (((long)img * a.H + hi) * a.W + wi) * a.C + c) : z;
    }
#pragma unroll
  for (int j = 0; j < 6; ++j) bt6<PTS>(d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]);   // B^T d
#pragma unroll
  for (int i = 0; i < 6; ++i) bt6<PTS>(d[i][0], d[i][1], d[i][2], d[i][3], d[i][4], d[i][5]);   // (B^T d) B
  const long gs = a.P * a.C;
  float* out = a.v + p * a.C + c;
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) *reinterpret_cast<V*>(out + (6 * i + j) * gs) = d[i][j];
}

//   A^T = [1 1 1 1 1 0; 0 a -a b -b 0; 0 a^2 a^2 b^2 b^2 0; 0 a^3 -a^3 b^3 -b^3 1]    (PTS 0: a = 1, b = 2)
template <int PTS, typename V>
__device__ __forceinline__ void at6(const V m0, const V m1, const V m2, const V m3, const V m4, const V m5, V& o0, V& o1, V& o2, V& o3) {
  const V s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
  o0 = (m0 + s12) + s34;
  if constexpr (PTS == 0) {
    o1 = d12 + 2.f * d34;
    o2 = s12 + 4.f * s34;
    o3 = (d12 + 8.f * d34) + m5;
  } else {
    o1 = kWa * d12 + kWb * d34;
    o2 = kWa2 * s12 + kWb2 * s34;
    o3 = ((kWa2 * kWa) * d12 + (kWb2 * kWb) * d34) + m5;
  }
}

template <int PTS, typename V>
__global__ __launch_bounds__(256) void wino43_output_transform(const WinoArgs a) {
  constexpr int NC = sizeof(V) / 4;
  const int c2n = a.Cout / NC;
  const long idx = xcd_contiguous_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x;  // overlapping patches: one L2
  if (idx >= a.P * c2n) return;
  const long p = idx / c2n;
  const int c = (int)(idx - p * c2n) * NC;
  const int tx = (int)(p % a.tw);
  const long q = p / a.tw;
  const int ty = (int)(q % a.th), img = (int)(q / a.th);
  const long gs = a.P * a.Cout;
  const float* in = a.m + p * a.Cout + c;
  V s[4][6];   // A^T m, column by column
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    V m[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) m[i] = *reinterpret_cast<const V*>(in + (6 * i + j) * gs);
    at6<PTS>(m[0], m[1], m[2], m[3], m[4], m[5], s[0][j], s[1][j], s[2][j], s[3][j]);
  }
  V b;
#pragma unroll
  for (int e = 0; e < NC; ++e) b[e] = 0.f;
  if (a.bias) b = *reinterpret_cast<const V*>(a.bias + c);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ho = 4 * ty + i;
    V o[4];
    at6<PTS>(s[i][0], s[i][1], s[i][2], s[i][3], s[i][4], s[i][5], o[0], o[1], o[2], o[3]);
    if (ho >= a.H) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int wo = 4 * tx + j;
      if (wo >= a.W) continue;
      V v = o[j] + b;
      if (a.relu) {
#pragma unroll
        for (int e = 0; e < NC; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      *reinterpret_cast<V*>(a.y + (((long)img * a.H + ho) * a.W + wo) * a.Cout + c) = v;
    }
  }
}

long wino_tiles(const ConvProblem& p, int m) { return (long)p.B * ((p.H + m - 1) / m) * ((p.W + m - 1) / m); }

}  // namespace

size_t conv_winograd_work_floats(const ConvProblem& p, int form) {
  const int m = conv_winograd_tile(form);
  return (size_t)(m + 2) * (m + 2) * wino_tiles(p, m) * ((size_t)p.Cin + p.Cout);
}

int conv_winograd_launch(const ConvProblem& p, const float* u, float* work, int form, hipStream_t stream) {
  PR_REQUIRE(form == 2 || form == 4 || form == 5, "winograd: form %d (2 = F(2x2), 4 = F(4x4), 5 = F(4x4) on the points 0, +-11/16, +-3/2)", form);
  const int m_out = conv_winograd_tile(form);
  PR_REQUIRE(p.precision == 0 && p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && !p.res,
             "winograd: 3x3 / stride 1 / pad 1 fp32 convolutions without residual only");
  PR_REQUIRE(p.Cin % kConvBK == 0 && p.Cout % 64 == 0, "winograd: Cin %% 32 and Cout %% 64 (got %d, %d)", p.Cin, p.Cout);
  PR_REQUIRE(p.x && p.y && u && work, "winograd: null tensor");
  if (p.B == 0) return PR_OK;
  const int n2 = (m_out + 2) * (m_out + 2);
  WinoArgs a;
  a.x = p.x; a.bias = p.bias; a.y = p.y;
  a.B = p.B; a.H = p.H; a.W = p.W; a.C = p.Cin; a.Cout = p.Cout; a.relu = p.relu;
  a.th = (p.H + m_out - 1) / m_out; a.tw = (p.W + m_out - 1) / m_out;
  a.P = wino_tiles(p, m_out);
  a.v = work;
  float* m = work + (size_t)n2 * a.P * p.Cin;
  a.m = m;
  PR_REQUIRE(a.P * std::max(p.Cin, p.Cout) < (1L << 29), "winograd: %ld tiles are too many for one launch", a.P);
  const bool wide = m_out == 4 && p.tune.wino_vec == 4;      // F(4x4): four channels per thread instead of two (same bits: elementwise)
  const int per = m_out == 4 ? (wide ? 4 : 2) : 4;   // channels per thread
  const long n_in = a.P * (p.Cin / per), n_out = a.P * (p.Cout / per);
  if (form == 5 && wide) hipLaunchKernelGGL((wino43_input_transform<1, f32x4>), dim3((unsigned)ceil_div(n_in, 256L)), dim3(256), 0, stream, a);
  else if (form == 5) hipLaunchKernelGGL((wino43_input_transform<1, f32x2>), dim3((unsigned)ceil_div(n_in, 256L)), dim3(256), 0, stream, a);
  else if (m_out == 4 && wide) hipLaunchKernelGGL((wino43_input_transform<0, f32x4>), dim3((unsigned)ceil_div(n_in, 256L)), dim3(256), 0, stream, a);
  else if (m_out == 4) hipLaunchKernelGGL((wino43_input_transform<0, f32x2>), dim3((unsigned)ceil_div(n_in, 256L)), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(wino_input_transform, dim3((unsigned)ceil_div(n_in, 256L)), dim3(256), 0, stream, a);
  PR_TRY(check_launch("wino_input_transform"));
  ConvProblem g;
  g.x = work; g.w = u; g.bias = nullptr; g.res = nullptr; g.y = m;
  g.B = (int)a.P; g.H = g.W = g.Ho = g.Wo = 1; g.Cin = p.Cin; g.Cout = p.Cout;
  g.KH = g.KW = 1; g.stride = 1; g.pad = 0; g.relu = 0; g.precision = 0;
  g.groups = n2;
  g.tune = p.tune;
  PR_REQUIRE(p.Cout % p.tune.wino_bn == 0, "winograd: Cout %d is not a multiple of the grouped GEMM's tile N %d", p.Cout, p.tune.wino_bn);
  // K = 128 / 256 (layer2, layer3): weights resident in registers (decided by the layer's shape, never by the batch)
  if (p.tune.wino_regw && conv_regw_f32_fits(g)) PR_TRY(conv_regw_f32_launch(g, stream));
  else PR_TRY(conv_dma_launch(g, p.tune.wino_bm, p.tune.wino_bn, stream, 256));
  if (form == 5 && wide) hipLaunchKernelGGL((wino43_output_transform<1, f32x4>), dim3((unsigned)ceil_div(n_out, 256L)), dim3(256), 0, stream, a);
  else if (form == 5) hipLaunchKernelGGL((wino43_output_transform<1, f32x2>), dim3((unsigned)ceil_div(n_out, 256L)), dim3(256), 0, stream, a);
  else if (m_out == 4 && wide) hipLaunchKernelGGL((wino43_output_transform<0, f32x4>), dim3((unsigned)ceil_div(n_out, 256L)), dim3(256), 0, stream, a);
  else if (m_out == 4) hipLaunchKernelGGL((wino43_output_transform<0, f32x2>), dim3((unsigned)ceil_div(n_out, 256L)), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(wino_output_transform, dim3((unsigned)ceil_div(n_out, 256L)), dim3(256), 0, stream, a);
  return check_launch("wino_output_transform");
}

}  // namespace pr
