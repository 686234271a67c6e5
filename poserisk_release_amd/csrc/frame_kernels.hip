// Per-frame, non-GEMM kernels of the hot path (all HBM/latency bound, no MFMA):
//   layout change NCHW->NHWC4, max-pool, global average pool, regressor state init/finalise
//   (rot6d -> rotmat), rotmat -> axis-angle -> Euler degrees, REBA, RULA.
#include "frame_kernels.h"

#include <cfloat>
#include <cmath>

namespace pr {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

// ---------------------------------------------------------------------------------------------
// Encoder plumbing
// ---------------------------------------------------------------------------------------------

// x[B,3,H,W] -> y[B,H,W,4] (4th channel zero).  One thread per pixel; the three plane reads are
// coalesced dwords, the write is one dwordx4.
__global__ void nchw3_to_nhwc4(const float* __restrict__ x, float* __restrict__ y, long npix_total,
                               int hw) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix_total) return;
  const long b = i / hw, p = i - b * hw;
  const float* src = x + b * 3 * hw + p;
  f32x4 v = {src[0], src[hw], src[2 * (long)hw], 0.f};
  *reinterpret_cast<f32x4*>(y + i * 4) = v;
}

// Space-to-depth for the stem: crops NCHW f32[B,3,H,W] -> [B,H/2,W/2,12], channel (2 di + dj) * 3 + c of pixel (i, j)
// = x[c][2i + di][2j + dj].  The 7x7 / stride-2 stem then is a 4x4 / stride-1 convolution over 12 channels whose
// K = 192 is a whole number of 32-deep K-steps (the NHWC4 form pads K = 196 to 224), hmr.hip packs the weights to match.
__global__ void nchw3_to_s2d12(const float* __restrict__ x, float* __restrict__ y, long npix_total, int H2, int W2) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix_total) return;
  const int hw2 = H2 * W2;
  const long b = i / hw2;
  const int p = (int)(i - b * hw2), pi = p / W2, pj = p - pi * W2;
  const long plane = 4L * hw2;
  const float* src = x + b * 3 * plane + (2L * pi) * (2 * W2) + 2 * pj;
  float v[12];
#pragma unroll
  for (int di = 0; di < 2; ++di)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float2 t = *reinterpret_cast<const float2*>(src + c * plane + di * (2 * W2));
      v[(2 * di) * 3 + c] = t.x;
      v[(2 * di + 1) * 3 + c] = t.y;
    }
  float* dst = y + i * 12;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    f32x4 o = {v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]};
    *reinterpret_cast<f32x4*>(dst + 4 * k) = o;
  }
}

// MaxPool2d(3, stride 2, pad 1) on NHWC, 4 channels per thread (-inf padding like PyTorch).
__global__ void maxpool3x3s2_nhwc(const float* __restrict__ x, float* __restrict__ y, int B, int H,
                                  int W, int C, int Ho, int Wo) {
  const int c4n = C / 4;
  const long total = (long)B * Ho * Wo * c4n;
  // vertically adjacent outputs share an input row: keep neighbouring blocks on one XCD's L2
  const long i = xcd_contiguous_block(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c4 = (int)(i % c4n);
  long r = i / c4n;
  const int wo = (int)(r % Wo);
  r /= Wo;
  const int ho = (int)(r % Ho);
  const int b = (int)(r / Ho);
  f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int hi = ho * 2 - 1 + kh;
    if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int wi = wo * 2 - 1 + kw;
      if ((unsigned)wi >= (unsigned)W) continue;
      const f32x4 v = *reinterpret_cast<const f32x4*>(x + (((long)b * H + hi) * W + wi) * C + c4 * 4);
      m[0] = fmaxf(m[0], v[0]);
      m[1] = fmaxf(m[1], v[1]);
      m[2] = fmaxf(m[2], v[2]);
      m[3] = fmaxf(m[3], v[3]);
    }
  }
  *reinterpret_cast<f32x4*>(y + i * 4) = m;
}

// AvgPool2d(7) on [B,7,7,C] NHWC -> [B,C].
__global__ void avgpool_nhwc(const float* __restrict__ x, float* __restrict__ y, int B, int HW, int C) {
  const int c4n = C / 4;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * c4n) return;
  const int c4 = (int)(i % c4n);
  const long b = i / c4n;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  const float* p = x + b * HW * C + c4 * 4;
  // seven loads in flight, then their adds in pixel order: the sum's order (and bits) are those of the plain loop, whose 49
  // dependent 16-byte loads were 49 memory latencies in a row (14.7 us at B=64 for 25.7 MB)
  for (int k0 = 0; k0 < HW; k0 += 7) {
    f32x4 v[7];
#pragma unroll
    for (int e = 0; e < 7; ++e) v[e] = k0 + e < HW ? *reinterpret_cast<const f32x4*>(p + (long)(k0 + e) * C) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 7; ++e)
      if (k0 + e < HW) s += v[e];
  }
  const float d = (float)HW;
  f32x4 o = {s[0] / d, s[1] / d, s[2] / d, s[3] / d};
  *reinterpret_cast<f32x4*>(y + i * 4) = o;
}

// ---- bf16 encoder plumbing (precision = 1) -----------------------------------------------------------
using u16x8 = __attribute__((ext_vector_type(8))) unsigned short;
__device__ inline float bf2f(unsigned short b) { return __uint_as_float((unsigned)b << 16); }
__device__ inline unsigned short f2bf(float f) {
  const __bf16 h = (__bf16)f;
  return __builtin_bit_cast(unsigned short, h);
}

// x f32[B,3,H,W] -> y bf16[B,H,W,8] (channels 3..7 zero): one 16-byte chunk per pixel.
__global__ void nchw3_to_nhwc8_bf16(const float* __restrict__ x, unsigned short* __restrict__ y,
                                    long npix_total, int hw) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix_total) return;
  const long b = i / hw, p = i - b * hw;
  const float* src = x + b * 3 * hw + p;
  u16x8 v = {f2bf(src[0]), f2bf(src[hw]), f2bf(src[2 * (long)hw]), 0, 0, 0, 0, 0};
  *reinterpret_cast<u16x8*>(y + i * 8) = v;
}

// bf16 twin of nchw3_to_s2d12: 16 channels per pixel (12 real + 4 zero), 32 bytes.
__global__ void nchw3_to_s2d16_bf16(const float* __restrict__ x, unsigned short* __restrict__ y, long npix_total,
                                    int H2, int W2) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix_total) return;
  const int hw2 = H2 * W2;
  const long b = i / hw2;
  const int p = (int)(i - b * hw2), pi = p / W2, pj = p - pi * W2;
  const long plane = 4L * hw2;
  const float* src = x + b * 3 * plane + (2L * pi) * (2 * W2) + 2 * pj;
  unsigned short v[16];
#pragma unroll
  for (int k = 12; k < 16; ++k) v[k] = 0;
#pragma unroll
  for (int di = 0; di < 2; ++di)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float2 t = *reinterpret_cast<const float2*>(src + c * plane + di * (2 * W2));
      v[(2 * di) * 3 + c] = f2bf(t.x);
      v[(2 * di + 1) * 3 + c] = f2bf(t.y);
    }
  u16x8 o0 = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]}, o1 = {v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15]};
  *reinterpret_cast<u16x8*>(y + i * 16) = o0;
  *reinterpret_cast<u16x8*>(y + i * 16 + 8) = o1;
}

// MaxPool2d(3, 2, 1) on bf16 NHWC, 8 channels (16 bytes) per thread, for the ENCODER's input: the stem's ReLU output,
// i.e. non-negative values (and possibly -0 or a NaN).  On that domain the order of bf16 values is the order of their bit
// patterns as SIGNED 16-bit integers (-0 = 0x8000 is the smallest, a positive NaN 0x7fc0 the largest, so it propagates
// like torch's max-pool), so the maximum is four v_pk_max_i16 per tap instead of sixteen conversions and maxima --
// the float form of this kernel was VALU-bound at 2.3 TB/s (profiles/r02_bench_b256_bf16_lanes1_kernel_stats.csv).
using i16x8 = __attribute__((ext_vector_type(8))) short;
__global__ void maxpool3x3s2_nhwc_bf16(const unsigned short* __restrict__ x, unsigned short* __restrict__ y,
                                       int B, int H, int W, int C, int Ho, int Wo) {
  const int c8n = C / 8;
  const long total = (long)B * Ho * Wo * c8n;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c8 = (int)(i % c8n);
  long r = i / c8n;
  const int wo = (int)(r % Wo);
  r /= Wo;
  const int ho = (int)(r % Ho);
  const int b = (int)(r / Ho);
  const short lowest = (short)-32768;     // -0: below every value of the domain
  i16x8 m = {lowest, lowest, lowest, lowest, lowest, lowest, lowest, lowest};
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int hi = ho * 2 - 1 + kh;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int wi = wo * 2 - 1 + kw;
      if ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W) {
        const i16x8 v = *reinterpret_cast<const i16x8*>(x + (((long)b * H + hi) * W + wi) * C + c8 * 8);
        m = __builtin_elementwise_max(m, v);
      }
    }
  }
  *reinterpret_cast<i16x8*>(y + i * 8) = m;
}

// AvgPool2d(7) on bf16 [B,HW,C] -> f32 [B,C] (the regressor runs in fp32).
__global__ void avgpool_nhwc_bf16(const unsigned short* __restrict__ x, float* __restrict__ y, int B, int HW,
                                  int C) {
  const int c8n = C / 8;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * c8n) return;
  const int c8 = (int)(i % c8n);
  const long b = i / c8n;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const unsigned short* p = x + b * HW * C + c8 * 8;
  for (int k0 = 0; k0 < HW; k0 += 7) {          // seven loads in flight, adds in pixel order (see avgpool_nhwc)
    u16x8 v[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) v[j] = k0 + j < HW ? *reinterpret_cast<const u16x8*>(p + (long)(k0 + j) * C) : u16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 7; ++j)
      if (k0 + j < HW) {
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += bf2f(v[j][e]);
      }
  }
  const float d = (float)HW;
#pragma unroll
  for (int e = 0; e < 8; ++e) y[i * 8 + e] = s[e] / d;
}

__global__ void f32_to_bf16_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = f2bf(x[i]);
}
__global__ void bf16_to_f32_kernel(const unsigned short* __restrict__ x, float* __restrict__ y, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = bf2f(x[i]);
}

// ---------------------------------------------------------------------------------------------
// Crop front-end: bbox -> affine -> cv2.warpAffine(INTER_LINEAR, BORDER_CONSTANT) -> ToTensor
// _img_utils.py:53-101, 219-252, 259-266; data/demo_dataset.py:58-74.  OpenCV's 8-bit bilinear warp is
// fixed-point (AB_BITS 10, INTER_BITS 5, coefficients scaled by 2^15); the same integers are formed here.
// ---------------------------------------------------------------------------------------------
// A thread owns column x of kCropRows rows (y0, y0 + 28, ...) of one crop: the inverse matrix (three double divisions) and the
// column terms are formed once per thread instead of once per pixel -- a wave executes what its busiest lane executes, so
// per-pixel copies of that arithmetic were half of the kernel's instructions.  Every pixel sees the same operations in the same
// order as before (same bits); stores stay coalesced (a wave = 64 consecutive x).
constexpr int kCropRows = 8;
__global__ void crop_frames_kernel(const unsigned char* __restrict__ frames, int F, int H, int W, int bgr,
                                   const int* __restrict__ frame_idx, const float* __restrict__ bboxes, int N,
                                   float scale, float* __restrict__ crops, int* __restrict__ status) {
  constexpr int S = 224, kPer = S * S / kCropRows;
  static_assert(S % kCropRows == 0, "rows per thread");
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)N * kPer) return;
  const int n = (int)(i / kPer), q = (int)(i - (long)n * kPer);
  const int y0 = q / S, x = q - y0 * S;
  // A frame index outside [0, F) (a tracker result that does not belong to these frames) must not become an
  // out-of-range read: the crop is zero-filled and flagged.
  const int fi = frame_idx ? frame_idx[n] : n;
  const bool bad_frame = (unsigned)fi >= (unsigned)F;
  if (status && q == 0) status[n] = bad_frame ? 1 : 0;
  if (bad_frame) {
#pragma unroll
    for (int k = 0; k < kCropRows; ++k)
#pragma unroll
      for (int c = 0; c < 3; ++c) crops[((long)n * 3 + c) * S * S + (y0 + k * (S / kCropRows)) * S + x] = 0.f;
    return;
  }
  const float* bb = bboxes + (long)n * 4;
  // gen_trans_from_patch_cv (rot = 0): control points are stored as float32
  const double cx = (double)bb[0], cy = (double)bb[1];
  const float down = (float)(((double)bb[3] * (double)scale) * 0.5);
  const float right = (float)(((double)bb[2] * (double)scale) * 0.5);
  const double sx0 = (double)(float)cx, sy0 = (double)(float)cy;
  const double sy1 = (double)(float)(cy + (double)down), sx2 = (double)(float)(cx + (double)right);
  const double a = 112.0 / (sx2 - sx0), d = 112.0 / (sy1 - sy0);
  double M[6] = {a, 0.0, 112.0 - a * sx0, 0.0, d, 112.0 - d * sy0};
  // cv::warpAffine inverts the forward matrix in place
  double D = M[0] * M[4] - M[1] * M[3];
  D = D != 0 ? 1.0 / D : 0.0;
  const double A11 = M[4] * D, A22 = M[0] * D;
  M[0] = A11; M[1] *= -D; M[3] *= -D; M[4] = A22;
  const double b1 = -M[0] * M[2] - M[1] * M[5], b2 = -M[3] * M[2] - M[4] * M[5];
  M[2] = b1; M[5] = b2;
  const long adelta = __double2ll_rn(M[0] * (double)x * 1024.0), bdelta = __double2ll_rn(M[3] * (double)x * 1024.0);
  const unsigned char* img = frames + (long)fi * H * W * 3;
  const long frame_bytes = (long)F * H * W * 3;
#pragma unroll
  for (int k = 0; k < kCropRows; ++k) {
    const int y = y0 + k * (S / kCropRows), p = y * S + x;
    const long X0 = __double2ll_rn((M[1] * (double)y + M[2]) * 1024.0) + 16;
    const long Y0 = __double2ll_rn((M[4] * (double)y + M[5]) * 1024.0) + 16;
    const long X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
    long sxl = X >> 5, syl = Y >> 5;
    sxl = sxl < -32768 ? -32768 : (sxl > 32767 ? 32767 : sxl);
    syl = syl < -32768 ? -32768 : (syl > 32767 ? 32767 : syl);
    const int sx = (int)sxl, sy = (int)syl, fx = (int)(X & 31), fy = (int)(Y & 31);
    // 32x32 coefficient table of initInterTab2D: (32-fx)(32-fy)*32 ...; entry (0,0) is (32767, 0, 0, 1)
    int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
    if (fx == 0 && fy == 0) {
      w00 = 32767;
      w11 = 1;
    }
    const bool y0ok = (unsigned)sy < (unsigned)H, y1ok = (unsigned)(sy + 1) < (unsigned)H;
    const bool x0ok = (unsigned)sx < (unsigned)W, x1ok = (unsigned)(sx + 1) < (unsigned)W;
    // The two pixels of a source row are six adjacent bytes: one unaligned 8-byte load per row where both pixels and the
    // two bytes behind them lie inside the frames (everywhere but at the image borders and the buffer's last bytes), byte
    // loads with the border zeros otherwise -- the same bytes either way (twelve byte gathers per pixel were the kernel).
    unsigned long long r0 = 0, r1 = 0;
    const long o0 = ((long)sy * W + sx) * 3, o1 = o0 + (long)W * 3;
    if (y0ok && x0ok && x1ok && (long)fi * H * W * 3 + o0 + 8 <= frame_bytes) {
      __builtin_memcpy(&r0, img + o0, 8);
    } else if (y0ok) {
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        if (x0ok) r0 |= (unsigned long long)img[o0 + e] << (8 * e);
        if (x1ok) r0 |= (unsigned long long)img[o0 + 3 + e] << (8 * (3 + e));
      }
    }
    if (y1ok && x0ok && x1ok && (long)fi * H * W * 3 + o1 + 8 <= frame_bytes) {
      __builtin_memcpy(&r1, img + o1, 8);
    } else if (y1ok) {
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        if (x0ok) r1 |= (unsigned long long)img[o1 + e] << (8 * e);
        if (x1ok) r1 |= (unsigned long long)img[o1 + 3 + e] << (8 * (3 + e));
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int ch = bgr ? 2 - c : c;  // cv2.imread is BGR; cvtColor(BGR2RGB) at demo_dataset.py:59
      const int p00 = (int)((r0 >> (8 * ch)) & 0xff), p01 = (int)((r0 >> (8 * (3 + ch))) & 0xff);
      const int p10 = (int)((r1 >> (8 * ch)) & 0xff), p11 = (int)((r1 >> (8 * (3 + ch))) & 0xff);
      int v = (p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + (1 << 14)) >> 15;
      v = v < 0 ? 0 : (v > 255 ? 255 : v);
      crops[((long)n * 3 + c) * S * S + p] = (float)v / 255.0f;  // ToTensor
    }
  }
}

// state[B,192] <- [init_pose(144) | init_shape(10) | init_cam(3) | 0...]
__global__ void regressor_state_init(const float* __restrict__ init157, float* __restrict__ state, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * kStateStride) return;
  const int c = i % kStateStride;
  state[i] = c < 157 ? init157[c] : 0.f;
}

__device__ inline void rot6d_one(const float* p6, float* R) {
  // SPIN utils/geometry.py rot6d_to_rotmat: x.view(-1,3,2): a1 = x[:, :, 0], a2 = x[:, :, 1]
  const float a1x = p6[0], a1y = p6[2], a1z = p6[4];
  const float a2x = p6[1], a2y = p6[3], a2z = p6[5];
  const float n1 = fmaxf(sqrtf(a1x * a1x + a1y * a1y + a1z * a1z), 1e-12f);  // F.normalize eps
  const float b1x = a1x / n1, b1y = a1y / n1, b1z = a1z / n1;
  const float d = b1x * a2x + b1y * a2y + b1z * a2z;
  const float ux = a2x - d * b1x, uy = a2y - d * b1y, uz = a2z - d * b1z;
  const float n2 = fmaxf(sqrtf(ux * ux + uy * uy + uz * uz), 1e-12f);
  const float b2x = ux / n2, b2y = uy / n2, b2z = uz / n2;
  const float b3x = b1y * b2z - b1z * b2y, b3y = b1z * b2x - b1x * b2z, b3z = b1x * b2y - b1y * b2x;
  // stack((b1,b2,b3), dim=-1): columns
  R[0] = b1x; R[1] = b2x; R[2] = b3x;
  R[3] = b1y; R[4] = b2y; R[5] = b3y;
  R[6] = b1z; R[7] = b2z; R[8] = b3z;
}

// One thread per (frame, joint) for the rotations; threads with joint < 13 also copy betas/cam.
__global__ void regressor_finalize(const float* __restrict__ state, int stride, float* __restrict__ rotmat,
                                   float* __restrict__ betas, float* __restrict__ cam,
                                   float* __restrict__ pose6d, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * 24) return;
  const int b = i / 24, j = i % 24;
  const float* s = state + (long)b * stride;
  if (rotmat) {
    float R[9];
    rot6d_one(s + j * 6, R);
#pragma unroll
    for (int e = 0; e < 9; ++e) rotmat[(long)i * 9 + e] = R[e];
  }
  if (pose6d) {
#pragma unroll
    for (int e = 0; e < 6; ++e) pose6d[(long)b * 144 + j * 6 + e] = s[j * 6 + e];
  }
  if (j < 10 && betas) betas[b * 10 + j] = s[144 + j];
  if (j >= 10 && j < 13 && cam) cam[b * 3 + (j - 10)] = s[154 + (j - 10)];
}

__global__ void rot6d_kernel(const float* __restrict__ pose6d, float* __restrict__ rotmat, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float R[9];
  rot6d_one(pose6d + i * 6, R);
#pragma unroll
  for (int e = 0; e < 9; ++e) rotmat[i * 9 + e] = R[e];
}

// ---------------------------------------------------------------------------------------------
// rotmat -> axis-angle (OpenCV Rodrigues semantics) -> Euler degrees   coord_utils.py:24-30, 69-95
// ---------------------------------------------------------------------------------------------

// Orthogonal polar factor (= U*Vt of the SVD OpenCV takes) by Newton iteration X <- (X + X^-T)/2.
__device__ inline void polar_factor(double* X) {
  for (int it = 0; it < 12; ++it) {
    double C[9];
    C[0] = X[4] * X[8] - X[5] * X[7]; C[1] = X[5] * X[6] - X[3] * X[8]; C[2] = X[3] * X[7] - X[4] * X[6];
    C[3] = X[7] * X[2] - X[8] * X[1]; C[4] = X[8] * X[0] - X[6] * X[2]; C[5] = X[6] * X[1] - X[7] * X[0];
    C[6] = X[1] * X[5] - X[2] * X[4]; C[7] = X[2] * X[3] - X[0] * X[5]; C[8] = X[0] * X[4] - X[1] * X[3];
    const double det = X[0] * C[0] + X[1] * C[1] + X[2] * C[2];
    if (!(fabs(det) > 1e-300)) break;  // singular / NaN: leave as is
    const double inv = 1.0 / det;
    double diff = 0.0;
#pragma unroll
    for (int e = 0; e < 9; ++e) {
      const double y = 0.5 * (X[e] + C[e] * inv);
      diff = fmax(diff, fabs(y - X[e]));
      X[e] = y;
    }
    if (diff < 1e-16) break;
  }
}

__device__ inline void rodrigues_mat2vec(const float* Rf, float* out) {
  double R[9];
  bool bad = false;
#pragma unroll
  for (int e = 0; e < 9; ++e) {
    R[e] = (double)Rf[e];
    bad |= !(fabs(R[e]) < 100.0);  // OpenCV checkRange(-100,100); also catches NaN/inf
  }
  if (bad) {
    out[0] = out[1] = out[2] = 0.f;
    return;
  }
  polar_factor(R);
  double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
  const double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
  double c = (R[0] + R[4] + R[8] - 1.0) * 0.5;
  c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
  double theta = acos(c);
  if (s < 1e-5) {
    if (c > 0) {
      rx = ry = rz = 0.0;
    } else {
      rx = sqrt(fmax((R[0] + 1.0) * 0.5, 0.0));
      ry = sqrt(fmax((R[4] + 1.0) * 0.5, 0.0)) * (R[1] < 0 ? -1.0 : 1.0);
      rz = sqrt(fmax((R[8] + 1.0) * 0.5, 0.0)) * (R[2] < 0 ? -1.0 : 1.0);
      if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && ((R[5] > 0) != (ry * rz > 0))) rz = -rz;
      theta /= sqrt(rx * rx + ry * ry + rz * rz);
      rx *= theta; ry *= theta; rz *= theta;
    }
  } else {
    const double vth = (1.0 / (2.0 * s)) * theta;
    rx *= vth; ry *= vth; rz *= vth;
  }
  out[0] = (float)rx; out[1] = (float)ry; out[2] = (float)rz;
}

// float32 arithmetic exactly as numpy float32 scalars do it: one rounding per operation, never
// contracted into an FMA (HIP's __fmul_rn/__fadd_rn are plain operators and do get contracted).
__device__ inline float mul_f32(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ inline float add_f32(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}

__device__ inline void rodrigues_vec2mat(const float* v, float* Rf) {
  double rx = (double)v[0], ry = (double)v[1], rz = (double)v[2];
  const double theta = sqrt(rx * rx + ry * ry + rz * rz);
  if (theta < DBL_EPSILON) {
    Rf[0] = 1.f; Rf[1] = 0.f; Rf[2] = 0.f; Rf[3] = 0.f; Rf[4] = 1.f; Rf[5] = 0.f; Rf[6] = 0.f; Rf[7] = 0.f; Rf[8] = 1.f;
    return;
  }
  const double c = cos(theta), s = sin(theta), c1 = 1.0 - c, it = 1.0 / theta;
  rx *= it; ry *= it; rz *= it;
  Rf[0] = (float)(c + c1 * rx * rx);        Rf[1] = (float)(c1 * rx * ry - s * rz);   Rf[2] = (float)(c1 * rx * rz + s * ry);
  Rf[3] = (float)(c1 * rx * ry + s * rz);   Rf[4] = (float)(c + c1 * ry * ry);        Rf[5] = (float)(c1 * ry * rz - s * rx);
  Rf[6] = (float)(c1 * rx * rz - s * ry);   Rf[7] = (float)(c1 * ry * rz + s * rx);   Rf[8] = (float)(c + c1 * rz * rz);
}

constexpr int kEulerFramesPerBlock = 8;

// One thread per (frame, joint); 8 frames per 192-thread block.
__global__ __launch_bounds__(kEulerFramesPerBlock * 24) void pose_to_euler_kernel(
    const float* __restrict__ rotmat, int N, float* __restrict__ axis_angle, double* __restrict__ euler,
    int32_t* __restrict__ status) {
  __shared__ int flags[kEulerFramesPerBlock];
  const int t = threadIdx.x, lf = t / 24;
  if (t < kEulerFramesPerBlock) flags[t] = 0;
  __syncthreads();
  const long frame = (long)blockIdx.x * kEulerFramesPerBlock + lf;
  const long i = frame * 24 + (t % 24);
  if (frame < N) {
    float aa[3], R[9];
    if (rotmat) {
      rodrigues_mat2vec(rotmat + i * 9, aa);  // coord_utils.py:27
      axis_angle[i * 3 + 0] = aa[0]; axis_angle[i * 3 + 1] = aa[1]; axis_angle[i * 3 + 2] = aa[2];
    } else {  // axis-angle given (axis_angle_to_euler_angle called on its own, coord_utils.py:83)
      aa[0] = axis_angle[i * 3 + 0]; aa[1] = axis_angle[i * 3 + 1]; aa[2] = axis_angle[i * 3 + 2];
    }
    rodrigues_vec2mat(aa, R);               // coord_utils.py:86 (float32 matrix out)
    int flag = 0;
    // isRotationMatrix (coord_utils.py:62-67) in float32, no contraction
    float n2 = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        float d = mul_f32(R[0 * 3 + a], R[0 * 3 + b]);
        d = add_f32(d, mul_f32(R[1 * 3 + a], R[1 * 3 + b]));
        d = add_f32(d, mul_f32(R[2 * 3 + a], R[2 * 3 + b]));
        const float e = add_f32(a == b ? 1.f : 0.f, -d);
        n2 = add_f32(n2, mul_f32(e, e));
      }
    if (!(sqrtf(n2) < 1e-6f)) flag |= 1;
    // rotationMatrixToEulerAngles (coord_utils.py:69-81): f32 products/sum, double sqrt/atan2
    const float sy2 = add_f32(mul_f32(R[0], R[0]), mul_f32(R[3], R[3]));
    const double sy = sqrt((double)sy2);
    double ex, ey, ez;
    if (!(sy < 1e-6)) {
      ex = atan2((double)R[7], (double)R[8]);
      ey = atan2(-(double)R[6], sy);
      ez = atan2((double)R[3], (double)R[0]);
    } else {
      ex = atan2(-(double)R[5], (double)R[4]);
      ey = atan2(-(double)R[6], sy);
      ez = 0.0;
    }
    // euler_to_rotMat(yaw=ez, pitch=ey, roll=ex) and the signed-sum round-trip check (:87-91)
    const double cz = cos(ez), sz = sin(ez), cy = cos(ey), syy = sin(ey), cx = cos(ex), sx = sin(ex);
    const double M[9] = {cz * cy, cz * syy * sx - sz * cx, cz * syy * cx + sz * sx,
                         sz * cy, sz * syy * sx + cz * cx, sz * syy * cx - cz * sx,
                         -syy,    cy * sx,                 cy * cx};
    double dsum = 0.0;
#pragma unroll
    for (int e = 0; e < 9; ++e) dsum += (double)R[e] - M[e];
    if (dsum > 0.1) flag |= 2;
    euler[i * 3 + 0] = ex * 180.0 / M_PI;
    euler[i * 3 + 1] = ey * 180.0 / M_PI;
    euler[i * 3 + 2] = ez * 180.0 / M_PI;
    if (flag) atomicOr(&flags[lf], flag);
  }
  __syncthreads();
  if (status && t < kEulerFramesPerBlock) {
    const long f = (long)blockIdx.x * kEulerFramesPerBlock + t;
    if (f < N) status[f] = flags[t];
  }
}

// ---------------------------------------------------------------------------------------------
// REBA / RULA: one thread per frame.  Rule order, strict inequalities and the quirks of the
// reference are kept; table values are reba.py:13-43 / rula.py:13-58.
// ---------------------------------------------------------------------------------------------
__constant__ int8_t kRebaA[5][3][4] = {
    {{1, 2, 3, 4}, {1, 2, 3, 4}, {3, 3, 5, 6}}, {{2, 3, 4, 5}, {3, 4, 5, 6}, {4, 5, 6, 7}},
    {{2, 4, 5, 6}, {4, 5, 6, 7}, {5, 6, 7, 8}}, {{3, 5, 6, 7}, {5, 6, 7, 8}, {6, 7, 8, 9}},
    {{4, 6, 7, 8}, {6, 7, 8, 9}, {7, 8, 9, 9}}};
__constant__ int8_t kRebaB[6][2][3] = {{{1, 2, 2}, {1, 2, 3}}, {{1, 2, 3}, {2, 3, 4}}, {{3, 4, 5}, {4, 5, 5}},
                                      {{4, 5, 5}, {5, 6, 7}}, {{6, 7, 8}, {7, 8, 8}}, {{7, 8, 8}, {8, 9, 9}}};
__constant__ int8_t kRebaC[12][12] = {
    {1, 1, 1, 2, 3, 3, 4, 5, 6, 7, 7, 7},          {1, 2, 2, 3, 4, 4, 5, 6, 6, 7, 7, 8},
    {2, 3, 3, 3, 4, 5, 6, 7, 7, 8, 8, 8},          {3, 4, 4, 4, 5, 6, 7, 8, 8, 9, 9, 9},
    {4, 4, 4, 5, 6, 7, 8, 8, 9, 9, 9, 9},          {6, 6, 6, 7, 8, 8, 9, 9, 10, 10, 10, 10},
    {7, 7, 7, 8, 9, 9, 9, 10, 10, 11, 11, 11},     {8, 8, 8, 9, 10, 10, 10, 10, 10, 11, 11, 11},
    {9, 9, 9, 10, 10, 10, 11, 11, 11, 12, 12, 12}, {10, 10, 10, 11, 11, 11, 11, 12, 12, 12, 12, 12},
    {11, 11, 11, 11, 12, 12, 12, 12, 12, 12, 12, 12}, {12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12}};
__constant__ int8_t kRulaA[6][3][4][2] = {
    {{{1, 2}, {2, 2}, {2, 3}, {3, 3}}, {{2, 2}, {2, 2}, {3, 3}, {3, 3}}, {{2, 3}, {3, 3}, {3, 3}, {4, 4}}},
    {{{2, 3}, {3, 3}, {3, 4}, {4, 4}}, {{3, 3}, {3, 3}, {3, 4}, {4, 4}}, {{3, 4}, {4, 4}, {4, 4}, {5, 5}}},
    {{{3, 3}, {4, 4}, {4, 4}, {5, 5}}, {{3, 4}, {4, 4}, {4, 4}, {5, 5}}, {{4, 4}, {4, 4}, {4, 5}, {5, 5}}},
    {{{4, 4}, {4, 4}, {4, 5}, {5, 5}}, {{4, 4}, {4, 4}, {4, 5}, {5, 5}}, {{4, 4}, {4, 5}, {5, 5}, {6, 6}}},
    {{{5, 5}, {5, 5}, {5, 6}, {6, 7}}, {{5, 6}, {6, 6}, {6, 7}, {7, 7}}, {{6, 6}, {6, 7}, {7, 7}, {7, 8}}},
    {{{7, 7}, {7, 7}, {7, 8}, {8, 9}}, {{8, 8}, {8, 8}, {8, 9}, {9, 9}}, {{9, 9}, {9, 9}, {9, 9}, {9, 9}}}};
__constant__ int8_t kRulaB[6][6][2] = {
    {{1, 3}, {2, 3}, {3, 4}, {5, 5}, {6, 6}, {7, 7}}, {{2, 3}, {2, 3}, {4, 5}, {5, 5}, {6, 7}, {7, 7}},
    {{3, 3}, {3, 4}, {4, 5}, {5, 5}, {6, 7}, {7, 7}}, {{5, 5}, {5, 6}, {6, 7}, {7, 7}, {7, 7}, {8, 8}},
    {{7, 7}, {7, 7}, {7, 8}, {8, 8}, {8, 8}, {8, 8}}, {{8, 8}, {8, 8}, {8, 8}, {8, 9}, {9, 9}, {9, 9}}};
__constant__ int8_t kRulaC[7][7] = {{1, 2, 3, 3, 4, 5, 5}, {2, 2, 3, 4, 4, 5, 5}, {3, 3, 3, 4, 4, 5, 6},
                                   {3, 3, 3, 4, 5, 6, 6}, {4, 4, 4, 5, 6, 7, 7}, {5, 5, 6, 6, 7, 7, 7},
                                   {5, 5, 6, 7, 7, 7, 7}};

enum Joint {
  PELVIS, L_HIP, R_HIP, TORSO, L_KNEE, R_KNEE, SPINE, L_ANKLE, R_ANKLE, CHEST, L_TOE, R_TOE, NECK,
  L_THORAX, R_THORAX, HEAD, L_SHOULDER, R_SHOULDER, L_ELBOW, R_ELBOW, L_WRIST, R_WRIST, L_HAND, R_HAND
};

__device__ inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ inline double pymax(double a, double b) { return b > a ? b : a; }  // Python max(a, b)

// shared by REBA and RULA (reba.py:337-356, rula.py:290-309)
__device__ inline void lower_arm_bending(const double* p, int& l, int& r) {
  double a = pymax(p[L_ELBOW * 3 + 1], p[L_ELBOW * 3 + 2]);
  if (a > -100 && a < -60) l = 1;
  else if (a < -100 || (a > -60 && a < 0)) l = 2;
  else l = 1;
  a = pymax(p[R_ELBOW * 3 + 1], p[R_ELBOW * 3 + 2]);
  if (a > 60 && a < 100) r = 1;
  else if (a > 100 || (a > 0 && a < 60)) r = 2;
  else r = 1;
}
__device__ inline int shoulder_rise(double a) {  // reba.py:245-260, rula.py:201-217
  if (fabs(a) < 10) return 0;
  else if (fabs(a) >= 10) return 1;
  return 0;
}
__device__ inline int within10_pair(double a, double b) {  // "both <10 -> 0, any >10 -> 1, else 0"
  if (fabs(a) < 10 && fabs(b) < 10) return 0;
  else if (fabs(a) > 10 || fabs(b) > 10) return 1;
  return 0;
}
__device__ inline int over10(double a) {
  if (fabs(a) < 10) return 0;
  else if (fabs(a) > 10) return 1;
  return 0;
}

__device__ inline int reba_open_branch(double a2) {  // reba.py:213-219 / :232-238 (Q13)
  if (fabs(a2) < 20) return 1;
  else if (a2 > 20 || a2 < 70) return 2;
  else if (a2 > 70) return 2;
  else if (a2 > -70 && a2 < -20) return 4;
  else if (a2 < -70) return 4;
  return 1;
}

__global__ void reba_kernel(const double* __restrict__ euler, int N, pr_reba_info info,
                            int32_t* __restrict__ out) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= N) return;
  const double* p = euler + (long)f * 72;
#define ANG(j, k) p[(j)*3 + (k)]
  // ---- group A (reba.py:106-119)
  int trunk = 0, neck = 0, leg = 0;
  {
    const double a = ANG(TORSO, 0);  // trunk_bending :140-148
    if (fabs(a) < 5) trunk += 1;
    else if ((a > 5 && a < 20) || (a > -20 && a < -5)) trunk += 2;
    else if ((a > 20 && a < 60) || (a < -20)) trunk += 3;
    else if (a > 60) trunk += 4;
    else trunk += 1;
    trunk += over10(ANG(TORSO, 1));  // trunk_twist :158-164
    // trunk_side_bending :150-156 always 0 (Q11)
  }
  {
    const double a = ANG(NECK, 0);  // neck_bending :166-172 (Q12)
    if (a > -5 && a < 20) neck += 1;
    else if (a < 20 || a < -5) neck += 2;
    else neck += 1;
    neck += within10_pair(ANG(NECK, 2), ANG(NECK, 1));  // neck_twist :174-181
  }
  {
    int s1, s2;  // leg_bending :183-201
    double a = ANG(L_KNEE, 0);
    if (a < 30) s1 = 0;
    else if (a > 30 && a < 60) s1 = 1;
    else if (a > 60 && info.sitting > 0) s1 = 2;
    else s1 = 0;
    a = ANG(R_KNEE, 0);
    if (a < 30) s2 = 0;
    else if (a > 30 && a < 60) s2 = 1;
    else if (a > 60 && info.sitting > 0) s2 = 2;
    else s2 = 0;
    leg = info.legs_bilateral + (s1 > s2 ? s1 : s2);
  }
  trunk = clampi(trunk, 1, 5);
  neck = clampi(neck, 1, 3);
  leg = clampi(leg, 1, 4);
  int score_a = kRebaA[trunk - 1][neck - 1][leg - 1] + info.load_force;

  // ---- group B (reba.py:122-138)
  const double l2 = ANG(L_SHOULDER, 2), l1 = ANG(L_SHOULDER, 1), l0 = ANG(L_SHOULDER, 0);
  const double r2 = ANG(R_SHOULDER, 2), r1 = ANG(R_SHOULDER, 1), r0 = ANG(R_SHOULDER, 0);
  int ua_l, ua_r;
  // upper_arm_bending :203-243
  if (l2 > -110 && l2 < -20) {
    if (fabs(l1) < 20) ua_l = 1;
    else if (l1 > 20 || (l1 > -45 && l1 < -20)) ua_l = 2;
    else if (l1 > -90 && l1 <= -45) ua_l = 3;
    else if (l1 < -90) ua_l = 4;
    else ua_l = 1;
  } else if (l2 > -20) ua_l = reba_open_branch(l1);
  else ua_l = 1;
  ua_l -= info.arm_supported_l;
  if (r2 > 20 && r2 < 110) {
    if (fabs(r1) < 20) ua_r = 1;
    else if (r1 < -20 || (r1 > 20 && r1 <= 45)) ua_r = 2;
    else if (r1 > 45 && r1 <= 90) ua_r = 3;
    else if (r1 > 90) ua_r = 4;
    else ua_r = 1;
  } else if (l2 > -20) ua_r = reba_open_branch(l1);  // LEFT angles on the right arm (Q13)
  else ua_r = 1;
  ua_r -= info.arm_supported_r;
  ua_l += shoulder_rise(ANG(L_THORAX, 2));
  ua_r += shoulder_rise(ANG(R_THORAX, 2));
  {  // upper_arm_abducted_rotated :292-335
    int s1, s2;
    if (l2 > -110 && l2 < -20) {
      if (l2 < 45 && fabs(l0) < 10) s1 = 0;
      else if (l2 > 45 || fabs(l0) > 10) s1 = 1;
      else s1 = 0;
    } else if (l2 > -20) {
      if (fabs(l1) < 20) s1 = 1;
      else if (l1 > 20 || l1 < 70) s1 = 1;
      else if (l1 > 70) s1 = 0;
      else if (l1 > -70 && l1 < -20) s1 = 1;
      else if (l1 < -70) s1 = 0;
      else s1 = 0;
      if (fabs(l0) > 10) s1 += 1;
    } else s1 = 0;
    if (r2 > 20 && r2 < 110) {
      if (r2 > 45 && fabs(r0) < 10) s2 = 0;
      else if (r2 < 45 || fabs(r0) > 10) s2 = 1;
      else s2 = 0;
    } else if (r2 < 20) {
      if (fabs(r1) < 20) s2 = 1;
      else if (r1 > -70 && r1 < -20) s2 = 1;
      else if (r1 < -70) s2 = 0;
      else if (r1 > 20 && r1 < 70) s2 = 1;
      else if (r1 > 70) s2 = 0;
      else s2 = 0;
      if (fabs(r0) > 10) s1 += 1;  // increments the LEFT score (Q14)
    } else s2 = 0;
    ua_l += s1;
    ua_r += s2;
  }
  int la_l, la_r;
  lower_arm_bending(p, la_l, la_r);
  int wr_l, wr_r;
  {  // wrist_bending :358-373 + wrist_side_bending_or_twisted :375-392
    double a = ANG(L_WRIST, 2);
    if (fabs(a) < 15) wr_l = 1; else if (fabs(a) > 15) wr_l = 2; else wr_l = 1;
    a = ANG(R_WRIST, 2);
    if (fabs(a) < 15) wr_r = 1; else if (fabs(a) > 15) wr_r = 2; else wr_r = 1;
    wr_l += within10_pair(ANG(L_WRIST, 1), ANG(L_WRIST, 0));
    wr_r += within10_pair(ANG(R_WRIST, 1), ANG(R_WRIST, 0));
  }
  ua_l = clampi(ua_l, 1, 6); ua_r = clampi(ua_r, 1, 6);
  la_l = clampi(la_l, 1, 2); la_r = clampi(la_r, 1, 2);
  wr_l = clampi(wr_l, 1, 3); wr_r = clampi(wr_r, 1, 3);
  const int b_l = kRebaB[ua_l - 1][la_l - 1][wr_l - 1], b_r = kRebaB[ua_r - 1][la_r - 1][wr_r - 1];
  int score_b = (b_l > b_r ? b_l : b_r) + info.coupling;
  score_a = clampi(score_a, 1, 12);
  score_b = clampi(score_b, 1, 12);
  int32_t* o = out + (long)f * 10;
  o[0] = kRebaC[score_a - 1][score_b - 1] + info.activity;
  o[1] = trunk; o[2] = neck; o[3] = leg;
  o[4] = ua_l; o[5] = ua_r; o[6] = la_l; o[7] = la_r; o[8] = wr_l; o[9] = wr_r;
#undef ANG
}

__global__ void rula_kernel(const double* __restrict__ euler, int N, pr_rula_info info,
                            int32_t* __restrict__ out) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= N) return;
  const double* p = euler + (long)f * 72;
#define ANG(j, k) p[(j)*3 + (k)]
  const double l2 = ANG(L_SHOULDER, 2), l1 = ANG(L_SHOULDER, 1);
  const double r2 = ANG(R_SHOULDER, 2), r1 = ANG(R_SHOULDER, 1);
  int ua_l, ua_r;
  // upper_arm_bending rula.py:158-199
  if (l2 > -70 && l2 < 110) {
    if (fabs(l1) < 20) ua_l = 1;
    else if (l1 > 20 || (l1 > -45 && l1 < -20)) ua_l = 2;
    else if (l1 > -90 && l1 <= -45) ua_l = 3;
    else if (l1 < -90) ua_l = 4;
    else ua_l = 1;
  } else if (l2 > -20) {
    if (fabs(l1) < 20) ua_l = 1;
    else if (l1 > 20 && l1 < 70) ua_l = 2;
    else if (l1 > 70) ua_l = 2;
    else if (l1 > -70 && l1 < -20) ua_l = 4;
    else if (l1 < -70) ua_l = 4;
    else ua_l = 1;
  } else ua_l = 1;
  ua_l -= info.arm_supported_l;
  ua_r = 0;
  if (r2 > -70 && r2 < 110) {
    if (fabs(r1) < 20) { /* rula.py:183 assigns the angle, the score stays 0 (Q15) */ }
    else if (r1 < -20 || (r1 > 20 && r1 <= 45)) ua_r = 2;
    else if (r1 > 45 && r1 <= 90) ua_r = 3;
    else if (r1 > 90) ua_r = 4;
    else ua_r = 1;
  } else if (r2 < 20) {
    if (fabs(r1) < 20) ua_r = 1;
    else if (r1 > -70 && r1 < -20) ua_r = 2;
    else if (r1 < -70) ua_r = 2;
    else if (r1 > 20 && r1 < 70) ua_r = 4;
    else if (r1 > 70) ua_r = 4;
    else ua_r = 1;
  } else ua_r = 1;
  ua_r -= info.arm_supported_r;
  ua_l += shoulder_rise(ANG(L_THORAX, 2));
  ua_r += shoulder_rise(ANG(R_THORAX, 2));
  {  // upper_arm_abducted rula.py:249-288
    int s1 = 0, s2 = 0;
    if (l2 > -110 && l2 < -20) {
      if (l2 < 45) s1 = 0; else if (l2 > 45) s1 = 1; else s1 = 0;
    } else if (l2 > -20) {
      if (fabs(l1) < 20) s1 = 1;
      else if (l1 > 20 && l1 < 70) s1 = 1;
      else if (l1 > 70) s1 = 0;
      else if (l1 > -70 && l1 < -20) s1 = 1;
      else if (l1 < -70) s1 = 0;
      else s1 = 0;
    } else s1 = 0;
    if (r2 > 20 && r2 < 110) {
      if (r2 > 45) s2 = 0; else if (r2 < 45) s2 = 1; else s2 = 0;
    } else if (r2 < 20) {
      if (fabs(r1) < 20) s2 = 1;
      else if (r1 > -70 && r1 < -20) s2 = 1;
      else if (r1 < -70) s2 = 0;
      else if (r1 > 20 && r1 < 70) s2 = 1;
      else if (r1 > 70) s2 = 0;
      else s2 = 0;
    }  // no trailing else in the reference: s2 stays 0
    ua_l += s1;
    ua_r += s2;
  }
  int la_l, la_r;
  lower_arm_bending(p, la_l, la_r);
  {  // bent_from_midline_or_out_to_side rula.py:311-326 (Q16)
    double a = ANG(L_THORAX, 0);
    if (a < 10 || (a > -45 && a < -10)) la_l += 0;
    else if (a > 10 || a < -45) la_l += 1;
    a = ANG(R_THORAX, 0);
    if (a > -10 || (a > 10 && a < 45)) la_r += 0;
    else if (a < -10 || a > 45) la_r += 1;
  }
  int wr_l, wr_r, wt_l, wt_r;
  {
    double a = fabs(ANG(L_WRIST, 2));  // wrist_bending rula.py:328-346
    if (a < 1) wr_l = 1; else if (a > 1 && a < 15) wr_l = 2; else if (a > 15) wr_l = 3; else wr_l = 1;
    a = fabs(ANG(R_WRIST, 2));
    if (a < 1) wr_r = 1; else if (a > 1 && a < 15) wr_r = 2; else if (a > 15) wr_r = 3; else wr_r = 1;
    wr_l += over10(ANG(L_WRIST, 1));   // wrist_side_bending :348-363
    wr_r += over10(ANG(R_WRIST, 1));
    a = fabs(ANG(L_WRIST, 0));         // wrist_twist :365-380
    if (a < 45) wt_l = 1; else if (a > 45) wt_l = 2; else wt_l = 1;
    a = fabs(ANG(R_WRIST, 0));
    if (a < 45) wt_r = 1; else if (a > 45) wt_r = 2; else wt_r = 1;
  }
  ua_l = clampi(ua_l, 1, 6); ua_r = clampi(ua_r, 1, 6);
  la_l = clampi(la_l, 1, 3); la_r = clampi(la_r, 1, 3);
  wr_l = clampi(wr_l, 1, 4); wr_r = clampi(wr_r, 1, 4);
  wt_l = clampi(wt_l, 1, 2); wt_r = clampi(wt_r, 1, 2);
  const int a_l = kRulaA[ua_l - 1][la_l - 1][wr_l - 1][wt_l - 1] + info.a_muscle_l + info.a_load_l;
  const int a_r = kRulaA[ua_r - 1][la_r - 1][wr_r - 1][wt_r - 1] + info.a_muscle_r + info.a_load_r;
  int score_a = a_l > a_r ? a_l : a_r;

  // ---- group B rula.py:143-156
  int neck, trunk, leg;
  {
    double a = ANG(NECK, 0);  // neck_bending :406-414
    if (a > -5 && a < 10) neck = 1;
    else if (a > 10 && a < 20) neck = 2;
    else if (a > 20) neck = 3;
    else if (a < -5) neck = 4;
    else neck = 1;
    neck += within10_pair(ANG(NECK, 2), ANG(NECK, 1));  // :416-422
    a = ANG(TORSO, 0);  // trunk_bending :382-390
    if (fabs(a) < 5) trunk = 1;
    else if (a > 5 && a < 20) trunk = 2;
    else if (a > 20 && a < 60) trunk = 3;
    else if (a > 60) trunk = 4;
    else trunk = 1;
    trunk += over10(ANG(TORSO, 1));  // trunk_twisted :399-404
    trunk += over10(ANG(TORSO, 2));  // trunk_side_bending :392-397
    leg = info.legs_bilateral;
  }
  neck = clampi(neck, 1, 6);
  trunk = clampi(trunk, 1, 6);
  leg = clampi(leg, 1, 2);
  int score_b = kRulaB[neck - 1][trunk - 1][leg - 1] + info.b_muscle + info.b_load;
  score_a = clampi(score_a, 1, 7);
  score_b = clampi(score_b, 1, 7);
  int32_t* o = out + (long)f * 12;
  o[0] = kRulaC[score_a - 1][score_b - 1];
  o[1] = ua_l; o[2] = ua_r; o[3] = la_l; o[4] = la_r; o[5] = wr_l; o[6] = wr_r;
  o[7] = wt_l; o[8] = wt_r; o[9] = neck; o[10] = trunk; o[11] = leg;
#undef ANG
}

inline unsigned blocks_for(long n, int bs) { return (unsigned)((n + bs - 1) / bs); }

}  // namespace

int launch_nchw3_to_nhwc4(const float* x, float* y, int B, int H, int W, hipStream_t s) {
  const long n = (long)B * H * W;
  if (n == 0) return PR_OK;
  hipLaunchKernelGGL(nchw3_to_nhwc4, dim3(blocks_for(n, 256)), dim3(256), 0, s, x, y, n, H * W);
  return check_launch("nchw3_to_nhwc4");
}
int launch_nchw3_to_s2d12(const float* x, float* y, int B, int H, int W, hipStream_t s) {
  const long n = (long)B * (H / 2) * (W / 2);
  if (n == 0) return PR_OK;
  hipLaunchKernelGGL(nchw3_to_s2d12, dim3(blocks_for(n, 256)), dim3(256), 0, s, x, y, n, H / 2, W / 2);
  return check_launch("nchw3_to_s2d12");
}
int launch_nchw3_to_s2d16_bf16(const float* x, void* y, int B, int H, int W, hipStream_t s) {
  const long n = (long)B * (H / 2) * (W / 2);
  if (n == 0) return PR_OK;
  hipLaunchKernelGGL(nchw3_to_s2d16_bf16, dim3(blocks_for(n, 256)), dim3(256), 0, s, x, (unsigned short*)y, n, H / 2, W / 2);
  return check_launch("nchw3_to_s2d16_bf16");
}
int launch_maxpool(const float* x, float* y, int B, int H, int W, int C, hipStream_t s) {
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long n = (long)B * Ho * Wo * (C / 4);
  if (n == 0) return PR_OK;
  hipLaunchKernelGGL(maxpool3x3s2_nhwc, dim3(blocks_for(n, 256)), dim3(256), 0, s, x, y, B, H, W, C, Ho, Wo);
  return check_launch("maxpool3x3s2_nhwc");
}
int launch_avgpool(const float* x, float* y, int B, int HW, int C, hipStream_t s) {
  const long n = (long)B * (C / 4);
  if (n == 0) return PR_OK;
  hipLaunchKernelGGL(avgpool_nhwc, dim3(blocks_for(n, 64)), dim3(64), 0, s, x, y, B, HW, C);
  return check_launch("avgpool_nhwc");
}
int launch_nchw3_to_nhwc8_bf16(const float* x, void* y, int B, int H, int W, hipStream_t s) {
  const long n = (long)B * H * W;
  if (n == 0) return PR_OK;
  hipLaunchKernelGGL(nchw3_to_nhwc8_bf16, dim3(blocks_for(n, 256)), dim3(256), 0, s, x, (unsigned short*)y, n, H * W);
  return check_launch("nchw3_to_nhwc8_bf16");
}
int launch_maxpool_bf16(const void* x, void* y, int B, int H, int W, int C, hipStream_t s) {
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long n = (long)B * Ho * Wo * (C / 8);
  if (n == 0) return PR_OK;
  hipLaunchKernelGGL(maxpool3x3s2_nhwc_bf16, dim3(blocks_for(n, 256)), dim3(256), 0, s, (const unsigned short*)x,
                     (unsigned short*)y, B, H, W, C, Ho, Wo);
  return check_launch("maxpool3x3s2_nhwc_bf16");
}
int launch_avgpool_bf16(const void* x, float* y, int B, int HW, int C, hipStream_t s) {
  const long n = (long)B * (C / 8);
  if (n == 0) return PR_OK;
  hipLaunchKernelGGL(avgpool_nhwc_bf16, dim3(blocks_for(n, 64)), dim3(64), 0, s, (const unsigned short*)x, y, B, HW, C);
  return check_launch("avgpool_nhwc_bf16");
}
int launch_f32_to_bf16(const float* x, void* y, long n, hipStream_t s) {
  if (n == 0) return PR_OK;
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, x, (unsigned short*)y, n);
  return check_launch("f32_to_bf16_kernel");
}
int launch_bf16_to_f32(const void* x, float* y, long n, hipStream_t s) {
  if (n == 0) return PR_OK;
  hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, (const unsigned short*)x, y, n);
  return check_launch("bf16_to_f32_kernel");
}
int launch_crop_frames(const unsigned char* frames, int F, int H, int W, int bgr, const int* frame_idx,
                       const float* bboxes, int N, float scale, float* crops, int* status, hipStream_t s) {
  const long n = (long)N * 224 * 224 / kCropRows;       // a thread writes kCropRows pixels of one column
  if (n == 0) return PR_OK;
  hipLaunchKernelGGL(crop_frames_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, frames, F, H, W, bgr,
                     frame_idx, bboxes, N, scale, crops, status);
  return check_launch("crop_frames_kernel");
}
int launch_state_init(const float* init157, float* state, int B, hipStream_t s) {
  const long n = (long)B * kStateStride;
  if (n == 0) return PR_OK;
  hipLaunchKernelGGL(regressor_state_init, dim3(blocks_for(n, 256)), dim3(256), 0, s, init157, state, B);
  return check_launch("regressor_state_init");
}
int launch_regressor_finalize(const float* state, float* rotmat, float* betas, float* cam, float* pose6d,
                              int B, hipStream_t s) {
  const long n = (long)B * 24;
  if (n == 0) return PR_OK;
  hipLaunchKernelGGL(regressor_finalize, dim3(blocks_for(n, 192)), dim3(192), 0, s, state, kStateStride,
                     rotmat, betas, cam, pose6d, B);
  return check_launch("regressor_finalize");
}
int launch_rot6d(const float* pose6d, float* rotmat, long n, hipStream_t s) {
  if (n == 0) return PR_OK;
  hipLaunchKernelGGL(rot6d_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, pose6d, rotmat, n);
  return check_launch("rot6d_kernel");
}
int launch_pose_to_euler(const float* rotmat, int N, float* aa, double* euler, int32_t* status,
                         hipStream_t s) {
  if (N == 0) return PR_OK;
  hipLaunchKernelGGL(pose_to_euler_kernel, dim3(blocks_for(N, kEulerFramesPerBlock)),
                     dim3(kEulerFramesPerBlock * 24), 0, s, rotmat, N, aa, euler, status);
  return check_launch("pose_to_euler_kernel");
}
int launch_reba(const double* euler, int N, const pr_reba_info& info, int32_t* out, hipStream_t s) {
  if (N == 0) return PR_OK;
  hipLaunchKernelGGL(reba_kernel, dim3(blocks_for(N, 64)), dim3(64), 0, s, euler, N, info, out);
  return check_launch("reba_kernel");
}
int launch_rula(const double* euler, int N, const pr_rula_info& info, int32_t* out, hipStream_t s) {
  if (N == 0) return PR_OK;
  hipLaunchKernelGGL(rula_kernel, dim3(blocks_for(N, 64)), dim3(64), 0, s, euler, N, info, out);
  return check_launch("rula_kernel");
}

}  // namespace pr
