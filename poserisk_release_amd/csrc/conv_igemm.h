// Implicit-GEMM convolution on the fp32-input MFMA (v_mfma_f32_32x32x2_f32), NHWC activations.
//
//   y[m][n] = act( sum_k A[m][k] * Wp[n][k] + bias[n] + res[m][n] )
//   m = (image, ho, wo)   n = output channel   k = (kh, kw, ci), ci fastest
//
// Replaces the conv/BN/ReLU stack of SPIN's HMR (call site lib/core/base.py:220); BN is
// folded into Wp / bias by the caller (hmr.hip).  Also used as the plain GEMM for the
// regressor's fully connected layers (KH = KW = 1, H = W = 1).
#pragma once
#include "common.h"
#include "host_plan.h"

namespace pr {

// kConvBK (floats of K per LDS stage) and the host-side weight packers: host_plan.h

// A/B switches of the conv launches.  Defaults are the measured best.  They are read from the environment ONCE per handle
// (conv_tuning_from_env, at pr_hmr_create; the stand-alone test entries read them per call) and travel in the
// ConvProblem, so two handles of one process can differ and nothing is latched per process.
struct ConvTuning {
  int force_cfg = -1;        // POSERISK_CONV_CFG=<index>: one tile configuration wherever it fits
  int tail = 1;              // POSERISK_CONV_TAIL=0: no quarter tiles for the remainder of a launch
  int tail_min_rounds = 2;   // POSERISK_TAIL_MIN_ROUNDS
  int tail_max_rem = 128;    // POSERISK_TAIL_MAX_REM
  int wino_vec = 2;          // POSERISK_WINO_VEC: channels per thread of the F(4x4) transform passes (2 or 4; same bits)
  int wino_bm = 64, wino_bn = 64;   // POSERISK_WINO_TILE=<BM>x<BN>: tile of the Winograd forms' grouped GEMM (A/B timing)
  int wino_regw = 1;         // POSERISK_WINO_REGW=0: the grouped GEMM of a Winograd layer with K = 128 / 256 on the tile kernel
                             // instead of the register-resident-weights kernel (conv_regw_f32.hip)
  int regw_per_cu = 2;       // POSERISK_REGW_PER_CU: persistent workgroups per CU of conv1x1_regw_f32 (A/B timing)
  int regw_wt = 0, regw_wnb = 0; // POSERISK_REGW_WT / POSERISK_REGW_WNB: the same for the grouped GEMMs of a Winograd layer
  int regw_t = 0, regw_nb = 0;   // POSERISK_REGW_T / POSERISK_REGW_NB: 16-pixel tiles and 64-channel blocks per unit of that kernel (0 = its defaults; same bits)
  int bal_stages = 4;        // POSERISK_BAL_STAGES=5: conv_bal_bf16's LDS ring of 5 stages (all 160 KB) instead of 4 (128 KB)
};
ConvTuning conv_tuning_from_env();

struct ConvProblem {
  const float* x;     // [B,H,W,Cin]   Cin % 4 == 0
  const float* w;     // packed [Cout][Kpad], k = (kh*KW + kw)*Cin + ci, zero padded to Kpad
  const float* bias;  // [Cout] or nullptr
  const float* res;   // [M,Cout] or nullptr
  float* y;           // [M,Cout]
  int B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad;
  int relu;
  int precision = 0;  // 0 = fp32 tensors, 1 = bf16 tensors (fp32 accumulate, fp32 bias)
  // groups > 1 (fp32 LDS-DMA kernel, 1x1 only, no bias/residual/ReLU): `groups` independent GEMMs in one launch,
  // x = [groups][M][Cin], w = [groups][Cout][Kpad], y = [groups][M][Cout]  (the 16 products of a Winograd conv)
  int groups = 1;
  // Second A-operand source (LDS-DMA kernels, 1x1 convs only): y = act(x*W1 + x2*W2 + bias + res) in ONE K loop,
  // x2 = [B,H2,W2,Cin2] sampled at (ho*stride2, wo*stride2); w = [Cout][Kpad + Cin2] (the two weight matrices side by
  // side).  This is how a Bottleneck's downsample branch is summed into its conv3 (no downsample tensor in HBM).
  const float* x2 = nullptr;
  int H2 = 0, W2 = 0, Cin2 = 0, stride2 = 1;
  // A 1x1 convolution applied to this convolution's output inside the same kernel (conv_fused.hip; this conv must be
  // 3x3 / stride 1 / pad 1 with Cout = 64 and a bias, its ReLU is applied): y3[M][N3] = act(relu(y) * w3^T + bias3 +
  // res3).  y itself is not written; `res` and `y` above are unused.
  const float* w3 = nullptr;      // packed [N3][64]
  const float* bias3 = nullptr;
  const float* res3 = nullptr;
  float* y3 = nullptr;
  int N3 = 0, relu3 = 0;
  // Split-K (fp32 LDS-DMA kernel, 64x64 tile): the K-steps of every tile are dealt to `splitk` workgroups, partial
  // tiles meet in split_slab (floats: tiles * splitk * 4096) and are summed in part order by the workgroup that draws
  // the last of split_tickets[tile] (ints: tiles; zeroed by the launch).  A fixed property of a LAYER (never chosen
  // from the batch size), so a frame's bits do not depend on its batch.
  int splitk = 1;
  float* split_slab = nullptr;
  int* split_tickets = nullptr;
  ConvTuning tune;
  int M() const { return B * Ho * Wo; }
  int K() const { return KH * KW * Cin; }
  int Kpad() const { return ceil_div(K(), kConvBK) * kConvBK; }
  double flops() const { return 2.0 * (double)M() * (Cout * (K() + (x2 ? Cin2 : 0)) + (w3 ? (double)N3 * Cout : 0.0)); }
};

int conv_num_tile_cfgs();
const char* conv_tile_cfg_name(int cfg);
// Picks a tile configuration for the problem (chip-filling heuristic).
int conv_pick_tile_cfg(const ConvProblem& p);
// Asynchronous launch on `stream`.  cfg from conv_pick_tile_cfg or an explicit index.
int conv_launch(const ConvProblem& p, int cfg, hipStream_t stream);

// 3x3 conv + the 1x1 conv behind it in one kernel (conv_fused.hip); reached through conv_launch when p.w3 is set.
int conv_fused3_launch(const ConvProblem& p, hipStream_t stream);

// Row-panel form of a short-K (<= 256) fp32 1x1 convolution, optionally with a second source (conv_fused.hip); reached
// through conv_launch with cfg == kConvCfgPanel.
int conv_panel_launch(const ConvProblem& p, hipStream_t stream);

// LDS-DMA kernel family (conv_dma.hip); reached through conv_launch with cfg >= 6.
int conv_dma_launch(const ConvProblem& p, int BM, int BN, hipStream_t stream, int threads = 0);

// bf16 twin (conv_dma_bf16.hip): x, w, res, y of the ConvProblem point at bf16 data (cast to float* only
// to share the struct); weights packed by conv_pack_weights_bf16 (K padded to a multiple of 64).
int conv_dma_bf16_launch(const ConvProblem& p, int BM, int BN, hipStream_t stream, int threads = 0);
int conv_tile_dims(int cfg, int* BM, int* BN);

// Winograd F(m x m, 3x3), m = 2 or 4, for 3x3 / stride 1 / pad 1 fp32 convolutions (conv_winograd.hip), n = m + 2:
//   V[n*n][P][Cin] = B^T d B per n x n input patch,  M_k = V_k U_k^T (n*n grouped GEMMs on the MFMA kernel),
//   y = A^T M A + bias (+ReLU);  P = B * ceil(H/m) * ceil(W/m) output tiles of m x m.
// U is packed by conv_winograd_pack_weights ([n*n][Cout][Cin], BN scale folded in double); `work` holds V then M:
// conv_winograd_work_floats(p, m) floats.  2.25x (m = 2) / 4x (m = 4) fewer MFMA FLOPs than the direct form, two
// extra streaming passes.
// `form`: 2 = F(2x2,3x3), 4 = F(4x4,3x3) on Lavin & Gray's points 0, +-1, +-2, 5 = F(4x4,3x3) on 0, +-11/16, +-3/2 (half the
// fp32 error of form 4 at the same cost; conv_winograd.hip).
size_t conv_winograd_work_floats(const ConvProblem& p, int form);
int conv_winograd_launch(const ConvProblem& p, const float* u, float* work, int form, hipStream_t stream);

// A whole layer1 Bottleneck (conv1 1x1 -> conv2 3x3 -> conv3 1x1 + residual, 64 planes, stride 1) as one persistent
// bf16 kernel (bottleneck_bf16.hip).  x, y: [B,H,W,256] bf16; w1 [64][256], w2 [64][576] (k = tap * 64 + c), w3 [256][64]
// bf16 with BatchNorm folded and rows permuted by bottleneck_pack_rows_bf16; biases fp32 in channel order.
// first: the stage's first block -- x is [B,H,W,64], w1 [64][64], conv3 and the downsample branch are one GEMM over
// [t2 | x]: w3 [256][128] (the two folded matrices side by side), b3 the two folded biases summed; no residual.
struct BottleneckProblem {
  const void* x = nullptr;
  void* y = nullptr;
  const void *w1 = nullptr, *w2 = nullptr, *w3 = nullptr;
  const float *b1 = nullptr, *b2 = nullptr, *b3 = nullptr;
  int B = 0, H = 0, W = 0, planes = 64;
  bool first = false;
  // bottleneck128_bf16 only: pixel tiles of the short chunk every second workgroup opens with (0 = none).  It takes the
  // workgroups' memory-heavy phases out of step with each other; the arithmetic of a pixel does not depend on it.
  int lead_tiles = 2;
  double flops() const { return 2.0 * B * H * W * (double)planes * planes * (first ? 1 + 9 + 8 : 4 + 9 + 4); }
};
int bottleneck_bf16_launch(const BottleneckProblem& p, hipStream_t stream);   // planes 64 (layer1) or 128 (layer2, plain blocks)
// layer2's plain blocks (bottleneck128_bf16.hip): x, y [B,H,W,512] bf16, W <= 31; w1 [128][512], w2 [128][1152] (slice-major k),
// w3 [512][128], rows permuted by bottleneck_pack_rows_bf16.
int bottleneck128_bf16_launch(const BottleneckProblem& p, hipStream_t stream);
// layer3's plain blocks, one frame per workgroup (bottleneck256_bf16.hip): x, y [B,H,W,1024] bf16, H W <= 224; w1 [256][1024]
// rows permuted by bottleneck_pack_rows_bf16; w2 ([256][2304], slice-major k) and w3 ([1024][256]) permuted likewise and then
// packed into MFMA fragment order by the two functions below.
int bottleneck256_bf16_launch(const BottleneckProblem& p, hipStream_t stream);

// The bf16 encoder's stem in one kernel (stem_pool_bf16.hip): 4x4 / stride-1 convolution (window y-2 .. y+1) over the
// 16-channel space-to-depth image x_s2d [B,H,H,16] + bias + ReLU + MaxPool2d(3,2,1) -> y [B,H/2,H/2,64]; w = the stem's
// packed bf16 weights [64][256] (k = tap * 16 + c).  H even, <= 112.
int stem_pool_bf16_launch(const void* x_s2d, const void* w, const float* bias, void* y, int B, int H, hipStream_t stream);

// The fp32 encoder's stem in one kernel (stem_pool_f32.hip): the same convolution over the 12-channel space-to-depth image
// x_s2d f32[B,112,112,12] (w = the stem's packed fp32 weights [64][192], k = tap * 12 + c) + bias + ReLU + MaxPool2d(3,2,1)
// -> y f32[B,56,56,64].  Weights in registers, input rows in an LDS ring, pooling in registers.
int stem_pool_f32_launch(const float* x_s2d, const float* w, const float* bias, float* y, int B, hipStream_t stream);

// A Bottleneck's 1x1 expansion + bias + residual + ReLU with the weights resident in registers (expand_res_bf16.hip):
// y[M][N] = act(t[M][K] . w[N][K]^T + bias + res), bf16 tensors, w in conv_pack_weights_bf16 layout.  K = 128, N = 512
// (layer2) or K = 256, N = 1024 (layer3, two workgroups per run of pixels).
// ... and a first block's conv3 with its downsample branch as a second, strided source in the same K loop, no residual:
// y = act(t . W3^T + x2[::s, ::s] . Wd^T + bias), w = [N][K1 + K2].  K1 = 128, K2 = 256, N = 512 (layer2).
int expand_dual_bf16_launch(const void* t, const void* x2, const void* w, const float* bias, void* y, int B, int Ho, int Wo, int H2,
                            int W2, int stride2, int K1, int K2, int N, int relu, hipStream_t stream);
int expand_res_bf16_launch(const void* t, const void* w, const float* bias, const void* res, void* y, long M, int K, int N,
                           int relu, hipStream_t stream);
// kConvCfgExpand (300, host_plan.h): conv_launch routes a matching bf16 1x1 + residual problem to that kernel

// bf16 convolution with the pixels dealt evenly to one persistent workgroup per CU (conv_bal_bf16.hip): 1x1 or 3x3,
// Cin % 64 == 0, Cout % 128 == 0, optional bias / ReLU, no residual; weights in conv_pack_weights_bf16 layout.
// variant 0: channel blocks of 256 where Cout allows, else 128; variant 1: blocks of 128.
bool conv_bal_bf16_fits(const ConvProblem& p);
bool conv_bal_bf16_pays(const ConvProblem& p, int cus);   // the measured rule for choosing it over the tile kernel
int conv_bal_bf16_launch(const ConvProblem& p, hipStream_t stream, int variant = 0);
// kConvCfgBalanced (301, host_plan.h): conv_launch: 301 = variant 0, 302 = variant 1

// fp32 1x1 / stride-1 convolution with Cin = 128 or 256 and the weights resident in registers (conv_regw_f32.hip): optional
// bias / residual / ReLU, weights in conv_pack_weights layout ([Cout][Cin]); its own fixed k order (not the tile kernel's bits).
bool conv_regw_f32_fits(const ConvProblem& p);
int conv_regw_f32_launch(const ConvProblem& p, hipStream_t stream);
// kConvCfgRegW (400, host_plan.h): conv_launch routes a matching problem to that kernel

}  // namespace pr
