// The device-free half of pr_hmr_create: everything between the caller's weight blob and what the kernels launch.
// (SPIN models/hmr.py as lib/core/base.py:81-84 builds and loads it: hmr(cfg.SPIN.SMPL_MEAN_PARAMS) + load_state_dict.)
//
//   * the canonical blob's layout and size (include/poserisk_hip.h);
//   * eval-mode BatchNorm folded into every convolution, in double;
//   * weight packing for every kernel family: [Cout][K] fp32 and bf16 rows, the space-to-depth stem, the Winograd-domain
//     weights U = G g G^T, the row permutation of the transposed-MFMA kernels, the fragment orders of bottleneck256_bf16;
//   * the 53-convolution execution plan: which launch carries which layer, buffer rotation, fusions, kernel routing;
//   * workspace sizes per sub-batch, and what a forward of B frames launches.
//
// Nothing here includes a HIP header: device memory is reached through PlanSink.  The library's sink allocates and copies
// with HIP (hmr.hip); tests/native/host_plan_check.cc builds this file with g++ -fsanitize=address,undefined and a sink
// over host memory (SURVEY.md section 5 asks for a sanitizer build of the host C++; this code otherwise only ever runs
// behind pr_hmr_create, which needs a GPU).
#pragma once
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "host_common.h"

namespace pr {

constexpr int kConvBK = 32;        // floats of K per LDS stage of the fp32 kernels: packed fp32 rows are padded to it
constexpr int kStateStride = 192;  // regressor state row: pose6d(144) | betas(10) | cam(3) | zero pad
constexpr int kImg = 224;
constexpr double kBnEps = 1e-5;
constexpr int kNumConv = 53;
constexpr int kHmrMaxChunks = 8;

// conv_launch routes (conv_igemm.h lists the kernels behind them)
constexpr int kConvCfgPanel = 100;
constexpr int kConvCfgExpand = 300;
constexpr int kConvCfgBalanced = 301;
constexpr int kConvCfgRegW = 400;

// ---- weight packing (pure host) --------------------------------------------------------------------------------
// PyTorch OIHW float weights (+ optional per-output-channel scale, applied in double) -> packed [Cout][Kpad], k = (kh KW +
// kw) cin_pad + ci, zero padded to a multiple of kConvBK.
void conv_pack_weights(const float* w_oihw, const double* scale, int Cout, int Cin_real, int cin_pad, int KH, int KW,
                       float* out_packed);
// bf16 twin: K padded to a multiple of 64, k order by conv_k_index_bf16.
void conv_pack_weights_bf16(const float* w_oihw, const double* scale, int Cout, int Cin_real, int cin_pad, int KH, int KW,
                            unsigned short* out_packed);
int conv_kpad_bf16(int K);
// Position of (tap, ci) in a packed bf16 weight row.  Kernels with more than one tap and Cin % 64 == 0 run their K loop
// SLICE-major: k = (ci / 64) * taps * 64 + tap * 64 + ci % 64 -- all taps of a 64-channel slice before the next slice, so
// a kernel can keep a slice's pixel block in LDS across the taps (bottleneck_bf16.hip does for its one slice; the
// multi-slice form was built and measured in round 3, profiles/r03_experiments.txt); one slice (Cin = 64) is the plain
// tap-major order.  Otherwise (the stem) k = tap * Cin + ci.
inline int conv_k_index_bf16(int tap, int ci, int taps, int cin_pad) {
  if (taps > 1 && cin_pad % 64 == 0) return (ci >> 6) * taps * 64 + tap * 64 + (ci & 63);
  return tap * cin_pad + ci;
}
unsigned short f32_to_bf16_host(float f);

// Winograd F(m x m, 3x3): `form` 2 = F(2x2,3x3), 4 = F(4x4,3x3) on Lavin & Gray's points 0, +-1, +-2, 5 = F(4x4,3x3) on
// 0, +-kWa, +-kWb (half the fp32 error of form 4 at the same cost; conv_winograd.hip).  U = G g G^T in double with the
// BatchNorm scale folded first, one rounding to fp32, layout [(m+2)^2][Cout][Cin].
constexpr float kWa = 11.f / 16.f, kWb = 3.f / 2.f;
inline int conv_winograd_tile(int form) { return form == 2 ? 2 : 4; }
void conv_winograd_pack_weights(const float* w_oihw, const double* scale, int Cout, int Cin, int form, float* out_u);

// Packed weight rows for the transposed MFMAs of the whole-Bottleneck kernels: row 32 T + i of the packed matrix is output
// channel 32 T + sigma(i), sigma(i) = 16 ((i >> 2) & 1) + 4 (i >> 3) + (i & 3).  `src` is [rows][K] (rows % 32 == 0).
void bottleneck_pack_rows_bf16(const unsigned short* src, int rows, int K, unsigned short* dst);
// bottleneck256_bf16.hip's fragment orders of conv2's [256][2304] and conv3's [1024][256] permuted rows
void bottleneck256_pack_w2_frags_bf16(const unsigned short* rows, unsigned short* dst);
void bottleneck256_pack_w3_frags_bf16(const unsigned short* rows, unsigned short* dst);

// shape predicates of the kernels the plan routes to (pure; the kernels' own launchers re-check them)
bool expand_res_bf16_fits(int K, int N);
bool expand_dual_bf16_fits(int K1, int K2, int N);
bool bottleneck256_bf16_fits(int H, int W);

// ---- the plan -------------------------------------------------------------------------------------------------
struct ConvSpec {
  int Cin_real, Cin, Cout, k, stride, pad, H, W;  // input H,W
  int relu;
  int in_buf, out_buf, res_buf;  // activation buffer ids (res_buf < 0: none)
  float* w = nullptr;            // device, packed
  float* bias = nullptr;         // device
  float* u = nullptr;            // device, Winograd-domain weights [(m+2)^2][Cout][Cin] (3x3 stride-1 layers of layer2..4)
  int wino_m = 0;                // Winograd output tile (2 or 4), 0 = direct form
  int wino_form = 0;             // ... and the form it belongs to (2, 4, or 5 = F(4x4) on the points 0, +-11/16, +-3/2)
  int cfg = -1;
  int layer = 0;                 // index among the 53 convolutions of the network (execution order), for the profile
  int stage = 0;                 // ResNet stage 0..3 (layer1..layer4); the stem counts as stage 0
  // A first Bottleneck's downsample branch summed into its conv3 (one K loop over [conv2 output | block input],
  // conv_igemm.h ConvProblem::x2): the block input's buffer, channels, size and the branch's stride.
  int in2_buf = -1, Cin2 = 0, H2 = 0, stride2 = 1, layer2 = -1;
  // The block's conv3 applied inside this (3x3, 64-channel) convolution's kernel (conv_fused.hip): packed weights and
  // bias, output channels, residual and output buffers.
  float* w3 = nullptr;
  float* bias3 = nullptr;
  int N3 = 0, res3_buf = -1, out3_buf = -1;
  // The stem after space-to-depth: a 4x4 / stride-1 convolution over 12 channels of the 112x112 map whose window starts
  // two pixels up-left (pad 2) and ends one pixel down-right, so the output size is given, not derived; its algorithmic
  // work stays the 7x7 convolution's.
  int out_hw = 0;
  double macs_fixed = 0;
  int splitk = 1;                // K-steps of every tile dealt to this many workgroups (a property of the layer)
  // A whole Bottleneck in one kernel (bottleneck_bf16.hip; bf16 layer1 blocks without a downsample branch): this spec is
  // the block (in_buf -> out_buf, Cin = Cout = 4 * planes); w / bias are conv1's, w2b / bias2b conv2's, w3 / bias3 conv3's
  // (rows permuted by bottleneck_pack_rows_bf16).
  int bneck_planes = 0;
  bool bneck_first = false;      // the stage's first block: 64-channel input, downsample branch in conv3's K loop
  float* w2b = nullptr;
  float* bias2b = nullptr;
  int Ho() const { return out_hw ? out_hw : (H + 2 * pad - k) / stride + 1; }
  int Wo() const { return out_hw ? out_hw : (W + 2 * pad - k) / stride + 1; }
  double macs_per_frame() const {
    if (bneck_planes)      // 1x1 (4P -> P, first block P -> P) + 3x3 (P -> P) + 1x1 (P -> 4P) (+ the first block's P -> 4P branch)
      return (double)H * W * bneck_planes * bneck_planes * (bneck_first ? 18.0 : 17.0);
    return macs_fixed > 0 ? macs_fixed : (double)Ho() * Wo() * (Cout * (Cin_real * k * k + Cin2) + (double)N3 * Cout);
  }
  // Multiply-adds the matrix pipes really execute per frame: the packed K (zero padding included) for direct layers,
  // (m+2)^2 products per m x m output tile for a Winograd layer.
  double mfma_macs_per_frame(int k_step) const {
    if (bneck_planes) return macs_per_frame();
    if (wino_m) {
      const double tiles = (double)((H + wino_m - 1) / wino_m) * ((W + wino_m - 1) / wino_m);
      return tiles * (wino_m + 2) * (wino_m + 2) * Cin * Cout;
    }
    const int kp = (k * k * Cin + k_step - 1) / k_step * k_step;
    return (double)Ho() * Wo() * (Cout * (double)(kp + Cin2) + (double)N3 * Cout);
  }
};

struct FcSpec {  // y[B,N] = x[B,K] * W^T (+bias) (+res)
  int K = 0, N = 0;
  float* w = nullptr;
  float* bias = nullptr;
};

// Where the plan's constants go.  upload: a device copy of `bytes` host bytes; zeros: a zero-filled device buffer.
struct PlanSink {
  virtual int upload(const void* host, size_t bytes, float** out) = 0;
  virtual int zeros(size_t bytes, float** out) = 0;
  virtual ~PlanSink() = default;
};

// Settings + plan of one handle.  pr_hmr (hmr.hip) derives from it and adds the device state.
struct HmrPlan {
  int max_batch = 0;
  int precision = 0;  // 0 = fp32 encoder, 1 = bf16 encoder (fp32 accumulate); the regressor is always fp32
  int conv_form = PR_CONV_FORM_BUILTIN_DEFAULT;  // fp32 encoder: 0 = every conv direct, 2 / 4 / 5 = a Winograd form, or a digit per stage
  int stage_form[4] = {0, 5, 5, 5};  // the form per ResNet stage (layer1 stays direct: 64 channels)
  int wino_min_c = 128;
  bool fuse_downsample = true;  // first Bottlenecks: conv3 and the downsample branch as one dual-source GEMM
  bool fuse_conv3 = true;       // layer1 blocks 1, 2: conv2 (3x3, 64 channels) and conv3 in one kernel
  bool expand_regs = true;      // bf16 encoder: layer2's / layer3's conv3 + residual with the weights in registers (expand_res_bf16.hip)
  bool balanced = true;         // bf16 encoder: the evenly dealt persistent kernel where it pays (conv_bal_bf16.hip)
  int cus = 256;
  bool fuse_stem = true;        // conv1 + bn1 + relu + maxpool in one kernel (stem_pool_f32.hip / stem_pool_bf16.hip; needs stem_s2d)
  bool fuse_bottleneck = true;  // bf16 encoder, layer1 blocks 1, 2: the whole Bottleneck in one persistent kernel
  bool fuse_bottleneck2 = true; // bf16 encoder, layer2's plain blocks likewise (bottleneck128_bf16.hip)
  bool fuse_bottleneck3 = true; // bf16 encoder, layer3's plain blocks as one launch each when the batch fills the CUs (bottleneck256_bf16.hip)
  int b128_lead = 2;            // ... and the short chunk every second workgroup of that kernel opens with (A/B: POSERISK_B128_LEAD)
  bool stem_s2d = true;         // the 7x7 / stride-2 stem as a 4x4 / stride-1 convolution on the space-to-depth input
  bool regw = true;             // fp32: 1x1 / stride-1 layers with K = 128 / 256 on conv1x1_regw_f32 (weights in registers)
  int panel_max_k = 128;        // 1x1 / stride-1 expansions (conv3) with K up to this run as row panels (conv_fused.hip)
  int splitk = 1;               // fp32: split-K factor of the 7x7-map layers with 512 output channels; measured slower: off
  int fc_tiles = 0;             // POSERISK_FC_TILES=1: the regressor's FC layers on the 64x64 conv tiles (round 1's form)
  int fc_shape = 0;             // POSERISK_FC_SHAPE=<10 MT + NT>: output tiles per workgroup of fc_rows16_f32 (0 = by the batch; same bits)
  std::vector<ConvSpec> convs;
  FcSpec fc1x, fc1s, fc2, dec;
  float* init157 = nullptr;
  size_t wino_floats_per_frame = 0;
  int final_buf = 0;
  // A plain layer3 block as ONE launch (a frame per workgroup) beside its three ordinary launches `first .. first + 2` of the
  // plan: taken per sub-batch when its frames fill whole rounds of CUs (fused_pays), bit-identical either way.
  struct FusedBlock {
    size_t first;
    ConvSpec blk;
  };
  std::vector<FusedBlock> fused3;
  // regressor workspaces (device, zero-filled)
  float* xf = nullptr;       // [B,2048]
  float* h_static = nullptr; // [B,1024]
  float* h1 = nullptr;       // [B,1024]
  float* h2 = nullptr;       // [B,1024]
  float* state = nullptr;    // [B,192]
};

size_t hmr_weight_floats();
// conv_form as pr_hmr_create accepts it (-1 default, 0, 2, 4, 5 or three digits of those)
bool hmr_conv_form_valid(int conv_form);
// Resolves the form (PR_CONV_FORM_DEFAULT -> the built-in default, moved by POSERISK_WINOGRAD) into plan->conv_form /
// stage_form and reads every POSERISK_* A/B switch of the plan from the environment -- once per handle.
void hmr_plan_configure(HmrPlan* plan, int precision, int conv_form, int max_batch);
// blob -> BN-folded packed weights (through `sink`) + the launch plan.  PR_OK or PR_ERR_INVALID (message set).
int hmr_plan_build(HmrPlan* plan, const float* blob, size_t n_floats, PlanSink& sink);

// A frame per workgroup pays when the sub-batch's frames fill whole rounds of CUs (bottleneck256_bf16.hip)
inline bool hmr_fused3_pays(int b, int cus) {
  const int rounds = (b + cus - 1) / cus;
  return b > 0 && (long)b * 100 >= (long)rounds * cus * 85;
}
// Sub-batch capacity for n concurrent sub-batches: a sub-batch is also the unit of one conv launch, whose tensors must stay
// under 2 GiB (the DMA kernels' out-of-range sentinel): 512 frames x 56x56x256 fp32 = 1.6 GB
inline int hmr_chunk_cap(int max_batch, int n_chunks) { return std::min(ceil_div(max_batch, n_chunks), 512); }
// Element counts of one sub-batch's workspaces: act[0] (the layout-changed input), act[1..5] (rotating feature maps), the
// Winograd V / M array, the split-K slab and tickets (0 = none).  bf16 maps hold the same elements in half the floats.
struct HmrChunkSizes {
  size_t act0_floats, act_floats, wino_floats, slab_floats, tickets;
};
HmrChunkSizes hmr_chunk_sizes(const HmrPlan& plan, int chunk_cap);
// What one forward of B frames launches (pr_hmr_plan_counts): event brackets (a Winograd layer counts once) and how many
// of them are Winograd layers.
// `serial`: the passes run one after the other on the caller's stream (profile mode, or one sub-batch stream).
void hmr_plan_counts(const HmrPlan& plan, int B, int chunk_cap, int n_chunks, bool serial, int* conv_launches, int* winograd_layers);
// How pr_hmr_forward cuts B frames into sub-batches (the one place that decides it): serial passes of at most chunk_cap
// frames when one sub-batch runs at a time (also when a concurrent share would exceed chunk_cap), else n contiguous
// shares [c B / n, (c + 1) B / n) on the sub-batch streams.  Returns the number of sub-batches (<= 4096 / 1), their sizes
// in `sizes` (frames; room for max(n_chunks, ceil(B / chunk_cap)) entries), *concurrent = whether they run side by side.
int hmr_split_batch(int B, int chunk_cap, int n_chunks, bool serial, int* sizes, int max_sizes, bool* concurrent);

}  // namespace pr
