// pr_hmr: SPIN HMR (ResNet-50 encoder + 3-iteration regressor + rot6d->rotmat) on gfx950.
// Replaces models.hmr / spin_model(batch)  (lib/core/base.py:81-84, :220).
//
// Host side: parse the canonical weight blob (include/poserisk_hip.h), fold eval-mode BatchNorm
// into the convolutions in double precision, pack weights as [Cout][K] for the implicit-GEMM
// kernel, build the 53-conv execution plan over NHWC activation buffers kept resident in HBM.
// The regressor's fc1 is split into its constant part (pooled features, computed once) and
// its state part (157 inputs, recomputed per iteration).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "conv_igemm.h"
#include "frame_kernels.h"

namespace pr {
namespace {

constexpr int kImg = 224;
constexpr double kBnEps = 1e-5;
constexpr int kNumConv = 53;

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
  int K, N;
  float* w = nullptr;
  float* bias = nullptr;
};

}  // namespace
}  // namespace pr

struct pr_hmr {
  int device = 0;
  int max_batch = 0;
  int precision = 0;  // 0 = fp32 encoder, 1 = bf16 encoder (fp32 accumulate); the regressor is always fp32
  int conv_form = PR_CONV_FORM_BUILTIN_DEFAULT;  // fp32 encoder: 0 = every conv direct, 2 / 4 = Winograd F(2x2,3x3) / F(4x4,3x3), or a digit per stage
  int stage_form[4] = {0, 5, 5, 5};  // the form per ResNet stage (layer1 stays direct: 64 channels)
  int wino_min_c = 128;
  bool fuse_downsample = true;  // first Bottlenecks: conv3 and the downsample branch as one dual-source GEMM
  bool fuse_conv3 = true;       // layer1 blocks 1, 2: conv2 (3x3, 64 channels) and conv3 in one kernel
  pr::ConvTuning tune;          // tile-choice / quarter-tile switches of the conv launches (read once, at create)
  int fc_tiles = 0;             // POSERISK_FC_TILES=1: the regressor's FC layers on the 64x64 conv tiles (round 1's form)
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
  int splitk = 1;               // fp32: split-K factor of the 7x7-map layers with 512 output channels (392 tiles at B=64); measured slower (below): off
  float* split_slab[8] = {};      // per sub-batch chunk (kMaxChunks)
  int* split_tickets[8] = {};
  std::vector<pr::ConvSpec> convs;
  pr::FcSpec fc1x, fc1s, fc2, dec;
  float* init157 = nullptr;
  std::vector<float*> dev_allocs;
  // activation buffers per frame chunk: 0 = NHWC4 input, 1..5 = rotating feature maps.
  // The batch is cut into n_chunks contiguous sub-batches that run the encoder on their own
  // HIP streams (frames are independent).  Measured on MI355X at B=64 this lock-step form is SLOWER
  // than one stream (sub-batch kernels are smaller and all streams run the same layer at once), so the
  // default is 1; what does pay is whole batches in flight on different streams, which the caller
  // drives (pipeline.FramePipeline lanes).  Kept because it is bit-identical and lets a batch exceed
  // one sub-batch's workspace.
  static constexpr int kMaxChunks = 8;
  int n_chunks = 1;
  int chunk_cap = 0;
  float* act[kMaxChunks][6] = {};
  float* wino_work[kMaxChunks] = {};  // V and M of the Winograd layers (conv_winograd.hip)
  size_t wino_floats_per_frame = 0;
  hipStream_t streams[kMaxChunks] = {};
  hipEvent_t ev_fork = nullptr;
  hipEvent_t ev_join[kMaxChunks] = {};
  std::vector<float*> act_allocs;
  float* xf = nullptr;       // [B,2048]
  float* h_static = nullptr; // [B,1024]
  float* h1 = nullptr;       // [B,1024]
  float* h2 = nullptr;       // [B,1024]
  float* state = nullptr;    // [B,192]
  int final_buf = 0;
  // profiling
  bool profile = false;
  // A plain layer3 block as ONE launch (a frame per workgroup) beside its three ordinary launches `first .. first + 2` of the
  // plan: taken per sub-batch when its frames fill whole rounds of CUs (fused_pays), bit-identical either way.
  struct FusedBlock {
    size_t first;
    pr::ConvSpec blk;
  };
  std::vector<FusedBlock> fused3;
  std::vector<float> prof_ms;
  std::vector<int> prof_n;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
  std::vector<int> pending_layer;
};

namespace pr {
namespace {

struct BlobReader {
  const float* p;
  size_t left;
  const float* take(size_t n) {
    if (n > left) return nullptr;
    const float* r = p;
    p += n;
    left -= n;
    return r;
  }
};

size_t hmr_weight_floats() {
  size_t n = 64 * 3 * 49 + 4 * 64;
  int inpl = 64;
  const int planes[4] = {64, 128, 256, 512}, blocks[4] = {3, 4, 6, 3};
  for (int L = 0; L < 4; ++L)
    for (int b = 0; b < blocks[L]; ++b) {
      const int pl = planes[L];
      n += (size_t)pl * inpl + 4 * pl;
      n += (size_t)pl * pl * 9 + 4 * pl;
      n += (size_t)pl * 4 * pl + 4 * pl * 4;
      if (b == 0) n += (size_t)pl * 4 * inpl + 4 * pl * 4;
      inpl = pl * 4;
    }
  n += (size_t)1024 * 2205 + 1024 + (size_t)1024 * 1024 + 1024;
  n += (size_t)144 * 1024 + 144 + 10 * 1024 + 10 + 3 * 1024 + 3 + 144 + 10 + 3;
  return n;
}

int upload(pr_hmr* h, const std::vector<float>& host, float** out) {
  float* d = nullptr;
  PR_HIP(hipMalloc(&d, host.size() * sizeof(float)));
  h->dev_allocs.push_back(d);
  PR_HIP(hipMemcpy(d, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
  *out = d;
  return PR_OK;
}

int dev_alloc(pr_hmr* h, size_t floats, float** out) {
  float* d = nullptr;
  PR_HIP(hipMalloc(&d, std::max<size_t>(floats, 4) * sizeof(float)));
  h->dev_allocs.push_back(d);
  PR_HIP(hipMemset(d, 0, std::max<size_t>(floats, 4) * sizeof(float)));
  *out = d;
  return PR_OK;
}

// One convolution of the blob with its BatchNorm (gamma, beta, mean, var): the raw filter and the BN folded, in
// double, into a per-output-channel scale and bias.
struct FoldedConv {
  const float* w = nullptr;
  std::vector<double> scale, bias;
};

int read_conv_bn(BlobReader& br, int Cout, int Cin_real, int k, FoldedConv* out) {
  out->w = br.take((size_t)Cout * Cin_real * k * k);
  const float* g = br.take(Cout);
  const float* be = br.take(Cout);
  const float* mu = br.take(Cout);
  const float* var = br.take(Cout);
  PR_REQUIRE(out->w && g && be && mu && var, "hmr: weight blob too short");
  out->scale.resize(Cout);
  out->bias.resize(Cout);
  for (int o = 0; o < Cout; ++o) {
    const double s = (double)g[o] / std::sqrt((double)var[o] + kBnEps);
    out->scale[o] = s;
    out->bias[o] = (double)be[o] - (double)mu[o] * s;
  }
  return PR_OK;
}

// Packed K extent of one convolution's weight rows in the handle's precision.
int packed_k(const pr_hmr* h, int K) { return h->precision == 1 ? conv_kpad_bf16(K) : ceil_div(K, kConvBK) * kConvBK; }

// Folded weights of one or two convolutions (two: a conv3 and the downsample branch summed into it) -> device
// rows [Cout][Kpad(f1) + Kpad(f2)] in the handle's precision.
int upload_packed(pr_hmr* h, const ConvSpec& spec, const FoldedConv& f1, const FoldedConv* f2, float** out) {
  const int K1 = packed_k(h, spec.k * spec.k * spec.Cin), K2 = f2 ? packed_k(h, spec.Cin2) : 0;
  if (h->precision == 1) {
    std::vector<unsigned short> a((size_t)spec.Cout * K1), b((size_t)spec.Cout * K2), packed((size_t)spec.Cout * (K1 + K2));
    conv_pack_weights_bf16(f1.w, f1.scale.data(), spec.Cout, spec.Cin_real, spec.Cin, spec.k, spec.k, a.data());
    if (f2) conv_pack_weights_bf16(f2->w, f2->scale.data(), spec.Cout, spec.Cin2, spec.Cin2, 1, 1, b.data());
    for (int o = 0; o < spec.Cout; ++o) {
      memcpy(&packed[(size_t)o * (K1 + K2)], &a[(size_t)o * K1], (size_t)K1 * 2);
      if (K2) memcpy(&packed[(size_t)o * (K1 + K2) + K1], &b[(size_t)o * K2], (size_t)K2 * 2);
    }
    std::vector<float> as_f((packed.size() + 1) / 2);
    memcpy(as_f.data(), packed.data(), packed.size() * 2);
    return upload(h, as_f, out);
  }
  std::vector<float> a((size_t)spec.Cout * K1), b((size_t)spec.Cout * K2), packed((size_t)spec.Cout * (K1 + K2));
  conv_pack_weights(f1.w, f1.scale.data(), spec.Cout, spec.Cin_real, spec.Cin, spec.k, spec.k, a.data());
  if (f2) conv_pack_weights(f2->w, f2->scale.data(), spec.Cout, spec.Cin2, spec.Cin2, 1, 1, b.data());
  for (int o = 0; o < spec.Cout; ++o) {
    memcpy(&packed[(size_t)o * (K1 + K2)], &a[(size_t)o * K1], (size_t)K1 * 4);
    if (K2) memcpy(&packed[(size_t)o * (K1 + K2) + K1], &b[(size_t)o * K2], (size_t)K2 * 4);
  }
  return upload(h, packed, out);
}

// conv weight + its BatchNorm -> packed folded weights and bias on device.  `second` (a conv3 whose block has a
// downsample branch, fused form): the branch's conv + BatchNorm follow in the blob and are summed into this conv.
int add_conv(pr_hmr* h, BlobReader& br, ConvSpec spec, bool second = false) {
  FoldedConv f1, f2;
  std::vector<float> s2d_w;
  if (spec.out_hw) {
    // blob: conv1.weight [64][3][7][7].  8x8 window starting at original pixel (2 ho - 4, 2 wo - 4), i.e. the 7x7
    // kernel with a zero row / column in front; tap (th, tw) of the 4x4 kernel covers original rows 2 th + di:
    //   W2[o][(2 di + dj) * 3 + c][th][tw] = W[o][c][2 th + di - 1][2 tw + dj - 1]   (zero outside 0..6)
    PR_TRY(read_conv_bn(br, spec.Cout, 3, 7, &f1));
    s2d_w.assign((size_t)spec.Cout * 12 * 16, 0.f);
    for (int o = 0; o < spec.Cout; ++o)
      for (int c = 0; c < 3; ++c)
        for (int kh = 0; kh < 7; ++kh)
          for (int kw = 0; kw < 7; ++kw) {
            const int th = (kh + 1) >> 1, di = (kh + 1) & 1, tw = (kw + 1) >> 1, dj = (kw + 1) & 1;
            s2d_w[(((size_t)o * 12 + (2 * di + dj) * 3 + c) * 4 + th) * 4 + tw] = f1.w[(((size_t)o * 3 + c) * 7 + kh) * 7 + kw];
          }
    f1.w = s2d_w.data();
  } else
  PR_TRY(read_conv_bn(br, spec.Cout, spec.Cin_real, spec.k, &f1));
  if (second) PR_TRY(read_conv_bn(br, spec.Cout, spec.Cin2, 1, &f2));
  std::vector<float> bias(spec.Cout);
  for (int o = 0; o < spec.Cout; ++o) bias[o] = (float)(f1.bias[o] + (second ? f2.bias[o] : 0.0));
  PR_TRY(upload_packed(h, spec, f1, second ? &f2 : nullptr, &spec.w));
  PR_TRY(upload(h, bias, &spec.bias));
  const float* w = f1.w;
  const std::vector<double>& scale = f1.scale;
  // 3x3 / stride 1 with >= 128 channels (layer2..layer4): Winograd F(2x2,3x3).  layer1 (64 channels at 56x56)
  // stays direct: its 16 GEMMs would have K = 64 and the V/M passes cost more than the MFMAs they save.
  // The form is a property of the handle (pr_hmr_create's conv_form), so one process can hold several.
  const int use_wino = h->stage_form[spec.stage];
  if (use_wino && h->precision == 0 && spec.k == 3 && spec.stride == 1 && spec.pad == 1 && spec.Cin >= h->wino_min_c &&
      spec.Cin == spec.Cin_real) {
    const int m = conv_winograd_tile(use_wino), n2 = (m + 2) * (m + 2);
    std::vector<float> u((size_t)n2 * spec.Cout * spec.Cin);
    conv_winograd_pack_weights(w, scale.data(), spec.Cout, spec.Cin, use_wino, u.data());
    PR_TRY(upload(h, u, &spec.u));
    spec.wino_m = m;
    spec.wino_form = use_wino;
    const size_t tiles = (size_t)((spec.H + m - 1) / m) * ((spec.W + m - 1) / m);
    h->wino_floats_per_frame = std::max(h->wino_floats_per_frame, n2 * tiles * ((size_t)spec.Cin + spec.Cout));
  }
  // bf16: a 128 -> 512 / 256 -> 1024 expansion with residual (conv3 of layer2's and layer3's plain blocks) on the
  // register-resident-weights kernel (the fp32 twin was built and lost: profiles/r03_experiments.txt)
  if (h->precision == 1 && h->expand_regs && spec.k == 1 && spec.stride == 1 && expand_res_bf16_fits(spec.Cin, spec.Cout) &&
      spec.res_buf >= 0 && spec.in2_buf < 0 && !second)
    spec.cfg = kConvCfgExpand;
  // ... and layer2's FIRST conv3 with its downsample branch as the second source of the same kernel
  if (h->precision == 1 && h->expand_regs && spec.k == 1 && spec.stride == 1 && second && spec.in2_buf >= 0 && spec.res_buf < 0 &&
      expand_dual_bf16_fits(spec.Cin, spec.Cin2, spec.Cout))
    spec.cfg = kConvCfgExpand;
  // fp32 1x1 layers with K = 128 or 256 (layer1's conv1, layer2's and layer3's conv3): weights resident in registers.
  // 256 -> 64 at 56x56: 66 us against the tile kernel's 70.5; the wider ones 1 - 3 us ahead or level (profiles/r04_experiments.txt 5)
  if (h->precision == 0 && h->regw && spec.k == 1 && spec.stride == 1 && spec.in2_buf < 0 && !second &&
      (spec.Cin == 128 || spec.Cin == 256) && spec.Cin == spec.Cin_real && spec.Cout % 64 == 0 && spec.cfg < 0)
    spec.cfg = kConvCfgRegW;
  // short-K expansions (layer2's conv3: K = 128; a first block's conv3 + downsample: 64 + 64) as row panels
  {
    const int bk = h->precision == 1 ? 64 : kConvBK;
    if (spec.k == 1 && spec.stride == 1 && spec.Cout > spec.Cin && spec.Cin + spec.Cin2 <= h->panel_max_k &&
        spec.Cin % bk == 0 && spec.Cin2 % bk == 0 && spec.cfg < 0)
      spec.cfg = kConvCfgPanel;
  }
  // The 7x7-map layers with 512 output channels are 49 B / 64 x 8 = 392 tiles at B=64: 1.53 per CU, the launch lasts as
  // long as a CU with two.  Their K (2048 / 4608) is long, so it is dealt to `splitk` workgroups per tile (conv_dma.hip).
  // Decided by the layer's shape only -- never by the batch -- so a frame's bits do not depend on its batch.
  // MEASURED (B=64, POSERISK_SPLITK=2|3|4|6): 3x3/2 layer 158 -> 171 / 177 / 174 / 186 us, the two 1x1 layers 68 -> 89 /
  // 99 / 105 / 133 us: the ticket zeroing launch, the slab round trip and above all one agent-scope release (an L2
  // write-back) per workgroup cost more than the better balance returns.  Off by default (splitk = 1).
  if (h->precision == 0 && h->splitk > 1 && !spec.wino_m && spec.cfg < 0 && spec.in2_buf < 0 && !spec.w3 && spec.Cout == 512 &&
      spec.Ho() == 7 && spec.Cin % kConvBK == 0 && (spec.k == 1 || spec.k == 3))
    spec.splitk = h->splitk;
  h->convs.push_back(spec);
  return PR_OK;
}

// Linear weight [N,K_real] (+bias) -> packed [Npad][Kpad] using columns [col0, col0+K_real) of the
// source row of length src_cols.
int make_fc(pr_hmr* h, const float* w, const float* b, int N, int src_cols, int col0, int K_real,
            int Kpad, int Npad, FcSpec* out) {
  std::vector<float> packed((size_t)Npad * Kpad, 0.f), bias(Npad, 0.f);
  for (int n = 0; n < N; ++n) {
    for (int k = 0; k < K_real; ++k) packed[(size_t)n * Kpad + k] = w[(size_t)n * src_cols + col0 + k];
    if (b) bias[n] = b[n];
  }
  out->K = Kpad;
  out->N = Npad;
  PR_TRY(upload(h, packed, &out->w));
  PR_TRY(upload(h, bias, &out->bias));
  return PR_OK;
}

int build(pr_hmr* h, const float* blob, size_t n_floats) {
  BlobReader br{blob, n_floats};
  // stem: conv1 7x7/2 (input padded to 4 channels) -> act[1]; maxpool -> act[2]
  ConvSpec c1{3, h->precision == 1 ? 8 : 4, 64, 7, 2, 3, kImg, kImg, 1, 0, 1, -1};
  if (h->stem_s2d) {
    c1 = ConvSpec{12, h->precision == 1 ? 16 : 12, 64, 4, 1, 2, kImg / 2, kImg / 2, 1, 0, 1, -1};
    c1.out_hw = kImg / 2;
    c1.macs_fixed = (double)(kImg / 2) * (kImg / 2) * 64 * 3 * 49;
  }
  PR_TRY(add_conv(h, br, c1));
  int cur = 2, H = 56, inpl = 64, layer = 1;
  const int planes[4] = {64, 128, 256, 512}, blocks[4] = {3, 4, 6, 3};
  for (int L = 0; L < 4; ++L)
    for (int b = 0; b < blocks[L]; ++b) {
      const int pl = planes[L];
      const int stride = (b == 0 && L > 0) ? 2 : 1;
      // pick 4 free buffers among 1..5 other than cur
      int fr[4], nf = 0;
      for (int i = 1; i <= 5 && nf < 4; ++i)
        if (i != cur) fr[nf++] = i;
      const int t1 = fr[0], t2 = fr[1], ds = fr[2], outb = fr[3];
      const int Ho = H / stride;
      ConvSpec a{inpl, inpl, pl, 1, 1, 0, H, H, 1, cur, t1, -1};
      ConvSpec bb{pl, pl, pl, 3, stride, 1, H, H, 1, t1, t2, -1};
      ConvSpec cc{pl, pl, pl * 4, 1, 1, 0, Ho, Ho, 1, t2, outb, b == 0 ? ds : cur};
      a.stage = bb.stage = cc.stage = L;
      if (h->precision == 1 && ((L == 0 && h->fuse_bottleneck && (b > 0 || h->fuse_downsample)) ||
                                (L == 1 && b > 0 && h->fuse_bottleneck2))) {
        // conv1 -> conv2 -> conv3 + identity of this block as ONE launch (bottleneck_bf16.hip): the folded weight
        // matrices in the kernel's layout, one spec; the launch is reported under conv3's index.  The first block's
        // downsample branch rides in conv3's K loop ([t2 | x], as in the dual-source GEMM), its bias summed in double.
        const bool first = b == 0;
        FoldedConv f1, f2, f3, fd;
        PR_TRY(read_conv_bn(br, pl, inpl, 1, &f1));
        PR_TRY(read_conv_bn(br, pl, pl, 3, &f2));
        PR_TRY(read_conv_bn(br, pl * 4, pl, 1, &f3));
        if (first) PR_TRY(read_conv_bn(br, pl * 4, inpl, 1, &fd));
        ConvSpec blk{inpl, inpl, pl * 4, 1, 1, 0, H, H, 1, cur, outb, -1};
        blk.stage = L;
        blk.bneck_planes = pl;
        blk.bneck_first = first;
        auto packed16 = [&](const FoldedConv& f, int Cout, int Cin, int k) {
          std::vector<unsigned short> a16((size_t)Cout * conv_kpad_bf16(k * k * Cin));
          conv_pack_weights_bf16(f.w, f.scale.data(), Cout, Cin, Cin, k, k, a16.data());
          return a16;
        };
        auto upload_rows = [&](const std::vector<unsigned short>& a16, int Cout, float** out) -> int {
          const int K = (int)(a16.size() / Cout);
          std::vector<unsigned short> p16(a16.size());
          bottleneck_pack_rows_bf16(a16.data(), Cout, K, p16.data());
          std::vector<float> as_f((p16.size() + 1) / 2);
          memcpy(as_f.data(), p16.data(), p16.size() * 2);
          return upload(h, as_f, out);
        };
        auto bias_of = [&](const FoldedConv& f, const FoldedConv* g, float** out) -> int {
          std::vector<float> bv(f.bias.size());
          for (size_t o = 0; o < bv.size(); ++o) bv[o] = (float)(f.bias[o] + (g ? g->bias[o] : 0.0));
          return upload(h, bv, out);
        };
        PR_TRY(upload_rows(packed16(f1, pl, inpl, 1), pl, &blk.w));
        PR_TRY(upload_rows(packed16(f2, pl, pl, 3), pl, &blk.w2b));
        if (first) {
          const std::vector<unsigned short> a3 = packed16(f3, pl * 4, pl, 1), ad = packed16(fd, pl * 4, inpl, 1);
          std::vector<unsigned short> both((size_t)pl * 4 * (pl + inpl));
          for (int o = 0; o < pl * 4; ++o) {
            memcpy(&both[(size_t)o * (pl + inpl)], &a3[(size_t)o * pl], (size_t)pl * 2);
            memcpy(&both[(size_t)o * (pl + inpl) + pl], &ad[(size_t)o * inpl], (size_t)inpl * 2);
          }
          PR_TRY(upload_rows(both, pl * 4, &blk.w3));
        } else {
          PR_TRY(upload_rows(packed16(f3, pl * 4, pl, 1), pl * 4, &blk.w3));
        }
        PR_TRY(bias_of(f1, nullptr, &blk.bias));
        PR_TRY(bias_of(f2, nullptr, &blk.bias2b));
        PR_TRY(bias_of(f3, first ? &fd : nullptr, &blk.bias3));
        layer += first ? 3 : 2;  // conv1, conv2 (and the downsample branch) report no launch of their own
        blk.layer = layer++;
        h->convs.push_back(blk);
        cur = outb;
        H = Ho;
        inpl = pl * 4;
        continue;
      }
      const bool alt3 = h->precision == 1 && h->fuse_bottleneck3 && L == 2 && b > 0 && bottleneck256_bf16_fits(H, H);
      if (alt3) {
        // the block's folded weights once more, in bottleneck256_bf16's layouts (the reader is rewound for the three specs below)
        BlobReader again = br;
        FoldedConv f1, f2, f3;
        PR_TRY(read_conv_bn(again, pl, inpl, 1, &f1));
        PR_TRY(read_conv_bn(again, pl, pl, 3, &f2));
        PR_TRY(read_conv_bn(again, pl * 4, pl, 1, &f3));
        pr_hmr::FusedBlock fb;
        fb.first = h->convs.size();
        fb.blk = ConvSpec{inpl, inpl, pl * 4, 1, 1, 0, H, H, 1, cur, outb, -1};
        fb.blk.stage = L;
        fb.blk.bneck_planes = pl;
        auto rows16 = [&](const FoldedConv& f, int Cout, int Cin, int k) {
          std::vector<unsigned short> a16((size_t)Cout * conv_kpad_bf16(k * k * Cin)), p16(a16.size());
          conv_pack_weights_bf16(f.w, f.scale.data(), Cout, Cin, Cin, k, k, a16.data());
          bottleneck_pack_rows_bf16(a16.data(), Cout, (int)(a16.size() / Cout), p16.data());
          return p16;
        };
        auto upload16 = [&](const std::vector<unsigned short>& p16, float** out) -> int {
          std::vector<float> as_f((p16.size() + 1) / 2);
          memcpy(as_f.data(), p16.data(), p16.size() * 2);
          return upload(h, as_f, out);
        };
        auto bias_of = [&](const FoldedConv& f, float** out) -> int {
          std::vector<float> bv(f.bias.size());
          for (size_t o = 0; o < bv.size(); ++o) bv[o] = (float)f.bias[o];
          return upload(h, bv, out);
        };
        PR_TRY(upload16(rows16(f1, pl, inpl, 1), &fb.blk.w));
        {
          const std::vector<unsigned short> r2 = rows16(f2, pl, pl, 3), r3 = rows16(f3, pl * 4, pl, 1);
          std::vector<unsigned short> g2(r2.size()), g3(r3.size());
          bottleneck256_pack_w2_frags_bf16(r2.data(), g2.data());
          bottleneck256_pack_w3_frags_bf16(r3.data(), g3.data());
          PR_TRY(upload16(g2, &fb.blk.w2b));
          PR_TRY(upload16(g3, &fb.blk.w3));
        }
        PR_TRY(bias_of(f1, &fb.blk.bias));
        PR_TRY(bias_of(f2, &fb.blk.bias2b));
        PR_TRY(bias_of(f3, &fb.blk.bias3));
        h->fused3.push_back(fb);
      }
      a.layer = layer++;
      bb.layer = layer++;
      PR_TRY(add_conv(h, br, a));
      PR_TRY(add_conv(h, br, bb));
      if (b == 0 && h->fuse_downsample) {
        // relu(bn3(conv3(t2)) + bn_d(conv_d(x))) as ONE GEMM over K = [t2's channels | x's channels]: the downsample
        // tensor is never written or read back (execution order: the branch is layer n, conv3 layer n + 1)
        cc.res_buf = -1;
        cc.in2_buf = cur; cc.Cin2 = inpl; cc.H2 = H; cc.stride2 = stride;
        cc.layer2 = layer++;
        cc.layer = layer++;
        PR_TRY(add_conv(h, br, cc, true));
      } else if (b == 0) {
        // blob order is conv3/bn3 then downsample; execution order is downsample before conv3
        const size_t mark = h->convs.size();
        PR_TRY(add_conv(h, br, cc));
        ConvSpec dd{inpl, inpl, pl * 4, 1, stride, 0, H, H, 0, cur, ds, -1};
        dd.stage = L;
        PR_TRY(add_conv(h, br, dd));
        std::swap(h->convs[mark], h->convs[mark + 1]);
        h->convs[mark].layer = layer++;
        h->convs[mark + 1].layer = layer++;
      } else if (L == 0 && h->fuse_conv3) {
        // conv2's 64 output channels are one tile: conv3 + residual + ReLU run on it inside conv2's kernel, and the
        // 64-channel map between them never reaches HBM (conv_fused.hip)
        FoldedConv f3;
        PR_TRY(read_conv_bn(br, cc.Cout, cc.Cin_real, 1, &f3));
        ConvSpec& f = h->convs.back();      // conv2, just added
        std::vector<float> bias3(cc.Cout);
        for (int o = 0; o < cc.Cout; ++o) bias3[o] = (float)f3.bias[o];
        PR_TRY(upload_packed(h, cc, f3, nullptr, &f.w3));
        PR_TRY(upload(h, bias3, &f.bias3));
        f.N3 = cc.Cout; f.res3_buf = cc.res_buf; f.out3_buf = cc.out_buf;
        f.layer2 = f.layer;
        f.layer = layer++;
      } else {
        cc.layer = layer++;
        PR_TRY(add_conv(h, br, cc));
      }
      if (alt3) {
        PR_REQUIRE(h->convs.size() == h->fused3.back().first + 3, "hmr: a plain layer3 block is three launches of the plan");
        h->fused3.back().blk.layer = h->convs.back().layer;   // reported under conv3's index, as the other whole-block kernels
      }
      cur = outb;
      H = Ho;
      inpl = pl * 4;
    }
  h->final_buf = cur;
  PR_REQUIRE(layer == kNumConv && (int)h->convs.size() ==
                                      kNumConv - (h->fuse_downsample ? 4 : 0) -
                                          (h->precision == 1 && h->fuse_bottleneck ? (h->fuse_downsample ? 6 : 4) : h->fuse_conv3 ? 2 : 0) -
                                          (h->precision == 1 && h->fuse_bottleneck2 ? 6 : 0),
             "hmr: planned %d convolutions in %zu launches, expected %d", layer, h->convs.size(), kNumConv);

  const float* fc1w = br.take((size_t)1024 * 2205);
  const float* fc1b = br.take(1024);
  const float* fc2w = br.take((size_t)1024 * 1024);
  const float* fc2b = br.take(1024);
  const float* dpw = br.take((size_t)144 * 1024);
  const float* dpb = br.take(144);
  const float* dsw = br.take((size_t)10 * 1024);
  const float* dsb = br.take(10);
  const float* dcw = br.take((size_t)3 * 1024);
  const float* dcb = br.take(3);
  const float* ip = br.take(144);
  const float* is = br.take(10);
  const float* ic = br.take(3);
  PR_REQUIRE(fc1w && fc1b && fc2w && fc2b && dpw && dpb && dsw && dsb && dcw && dcb && ip && is && ic,
             "hmr: weight blob too short");
  PR_REQUIRE(br.left == 0, "hmr: weight blob has %zu trailing floats", br.left);
  PR_TRY(make_fc(h, fc1w, fc1b, 1024, 2205, 0, 2048, 2048, 1024, &h->fc1x));
  PR_TRY(make_fc(h, fc1w, nullptr, 1024, 2205, 2048, 157, kStateStride, 1024, &h->fc1s));
  PR_TRY(make_fc(h, fc2w, fc2b, 1024, 1024, 0, 1024, 1024, 1024, &h->fc2));
  {
    std::vector<float> decw((size_t)157 * 1024), decb(157);
    std::copy(dpw, dpw + (size_t)144 * 1024, decw.begin());
    std::copy(dsw, dsw + (size_t)10 * 1024, decw.begin() + (size_t)144 * 1024);
    std::copy(dcw, dcw + (size_t)3 * 1024, decw.begin() + (size_t)154 * 1024);
    std::copy(dpb, dpb + 144, decb.begin());
    std::copy(dsb, dsb + 10, decb.begin() + 144);
    std::copy(dcb, dcb + 3, decb.begin() + 154);
    PR_TRY(make_fc(h, decw.data(), decb.data(), 157, 1024, 0, 1024, 1024, kStateStride, &h->dec));
  }
  {
    std::vector<float> init(160, 0.f);
    std::copy(ip, ip + 144, init.begin());
    std::copy(is, is + 10, init.begin() + 144);
    std::copy(ic, ic + 3, init.begin() + 154);
    PR_TRY(upload(h, init, &h->init157));
  }

  // workspaces
  const size_t B = (size_t)h->max_batch;
  PR_TRY(dev_alloc(h, B * 2048, &h->xf));
  PR_TRY(dev_alloc(h, B * 1024, &h->h_static));
  PR_TRY(dev_alloc(h, B * 1024, &h->h1));
  PR_TRY(dev_alloc(h, B * 1024, &h->h2));
  PR_TRY(dev_alloc(h, B * kStateStride, &h->state));
  h->prof_ms.assign(kNumConv, 0.f);
  h->prof_n.assign(kNumConv, 0);
  return PR_OK;
}

// (Re)allocate the activation buffers for n sub-batches and create their streams / events.
int set_chunks(pr_hmr* h, int n) {
  PR_REQUIRE(n >= 1 && n <= pr_hmr::kMaxChunks, "hmr: stream count %d out of range 1..%d", n, pr_hmr::kMaxChunks);
  PR_HIP(hipDeviceSynchronize());
  for (float* p : h->act_allocs) (void)hipFree(p);
  h->act_allocs.clear();
  h->n_chunks = n;
  // a sub-batch is also the unit of one conv launch, whose tensors must stay under 2 GiB (the DMA kernel's
  // out-of-range sentinel): 512 frames x 56x56x256 fp32 = 1.6 GB
  h->chunk_cap = std::min(ceil_div(h->max_batch, n), 512);
  const size_t cb = (size_t)h->chunk_cap;
  const size_t fmap = (size_t)112 * 112 * 64;  // == 56*56*256, the largest feature map per frame
  for (int c = 0; c < n; ++c) {
    for (int i = 0; i <= 5; ++i) {
      // element counts; bf16 buffers hold the same number of elements in half the bytes (input: 8 channels)
      size_t floats = i == 0 ? cb * kImg * kImg * 4 : cb * fmap;
      if (h->precision == 1 && i > 0) floats = (floats + 1) / 2;
      float* d = nullptr;
      PR_HIP(hipMalloc(&d, floats * sizeof(float)));
      h->act_allocs.push_back(d);
      h->act[c][i] = d;
    }
    if (h->wino_floats_per_frame) {
      float* d = nullptr;
      PR_HIP(hipMalloc(&d, cb * h->wino_floats_per_frame * sizeof(float)));
      h->act_allocs.push_back(d);
      h->wino_work[c] = d;
    }
    {
      size_t slab = 0, tickets = 0;
      for (const ConvSpec& cs : h->convs)
        if (cs.splitk > 1) {
          const size_t tiles = (size_t)ceil_div((int)(cb * cs.Ho() * cs.Wo()), 64) * (cs.Cout / 64);
          slab = std::max(slab, tiles * cs.splitk * 4096);
          tickets = std::max(tickets, tiles);
        }
      if (slab) {
        float* d = nullptr;
        PR_HIP(hipMalloc(&d, slab * sizeof(float)));
        h->act_allocs.push_back(d);
        h->split_slab[c] = d;
        PR_HIP(hipMalloc(&d, tickets * sizeof(int)));
        h->act_allocs.push_back(d);
        h->split_tickets[c] = reinterpret_cast<int*>(d);
      }
    }
    if (n > 1 && !h->streams[c]) PR_HIP(hipStreamCreateWithFlags(&h->streams[c], hipStreamNonBlocking));
    if (n > 1 && !h->ev_join[c]) PR_HIP(hipEventCreateWithFlags(&h->ev_join[c], hipEventDisableTiming));
  }
  if (n > 1 && !h->ev_fork) PR_HIP(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
  return PR_OK;
}

ConvProblem conv_problem(const pr_hmr* h, const ConvSpec& c, int chunk, int B) {
  ConvProblem p;
  p.x = h->act[chunk][c.in_buf];
  p.w = c.w;
  p.bias = c.bias;
  p.res = c.res_buf >= 0 ? h->act[chunk][c.res_buf] : nullptr;
  p.y = h->act[chunk][c.out_buf];
  p.B = B; p.H = c.H; p.W = c.W; p.Cin = c.Cin; p.Ho = c.Ho(); p.Wo = c.Wo(); p.Cout = c.Cout;
  p.KH = p.KW = c.k; p.stride = c.stride; p.pad = c.pad; p.relu = c.relu;
  p.precision = h->precision;
  p.tune = h->tune;
  if (c.in2_buf >= 0) {
    p.x2 = h->act[chunk][c.in2_buf];
    p.H2 = p.W2 = c.H2; p.Cin2 = c.Cin2; p.stride2 = c.stride2;
  }
  if (c.splitk > 1) {
    p.splitk = c.splitk;
    p.split_slab = h->split_slab[chunk];
    p.split_tickets = h->split_tickets[chunk];
  }
  if (c.w3 && c.out3_buf >= 0) {
    p.w3 = c.w3; p.bias3 = c.bias3; p.N3 = c.N3; p.relu3 = 1;
    p.res3 = c.res3_buf >= 0 ? h->act[chunk][c.res3_buf] : nullptr;
    p.y3 = h->act[chunk][c.out3_buf];
  }
  return p;
}

int fc_launch(const pr_hmr* h, const FcSpec& fc, const float* x, const float* res, float* y, int B, bool use_bias,
              hipStream_t s) {
  // A/B switch (pr_hmr::fc_tiles): the layer on the 64x64 conv tiles (round 1's form) instead of fc_regressor.hip
  if (!h->fc_tiles) return launch_fc_rows16(x, fc.w, use_bias ? fc.bias : nullptr, res, y, B, fc.N, fc.K, s);
  ConvProblem p;
  p.x = x; p.w = fc.w; p.bias = use_bias ? fc.bias : nullptr; p.res = res; p.y = y;
  p.B = B; p.H = p.W = p.Ho = p.Wo = 1; p.Cin = fc.K; p.Cout = fc.N;
  p.KH = p.KW = 1; p.stride = 1; p.pad = 0; p.relu = 0;
  p.tune = h->tune;
  return conv_launch(p, conv_pick_tile_cfg(p), s);
}


// One sub-batch of the encoder: where it reads, where it writes, which stream it runs on.
struct ChunkRun {
  int chunk;
  const float* x;
  int b;
  float* xf_out;
  hipStream_t s;
};

// Encoder over n sub-batches: layout change, 53 convs, max-pool, global average pool -> xf[b,2048].
// Launches are issued layer by layer across the sub-batches so that all streams advance together
// (issuing one whole sub-batch after another would stagger them by the host's enqueue time).
int encode_chunks(pr_hmr* h, const ChunkRun* runs, int n) {
  const bool bf = h->precision == 1;
  for (int i = 0; i < n; ++i) {
    if (h->stem_s2d) {
      if (bf) PR_TRY(launch_nchw3_to_s2d16_bf16(runs[i].x, h->act[runs[i].chunk][0], runs[i].b, kImg, kImg, runs[i].s));
      else PR_TRY(launch_nchw3_to_s2d12(runs[i].x, h->act[runs[i].chunk][0], runs[i].b, kImg, kImg, runs[i].s));
    } else {
      if (bf) PR_TRY(launch_nchw3_to_nhwc8_bf16(runs[i].x, h->act[runs[i].chunk][0], runs[i].b, kImg, kImg, runs[i].s));
      else PR_TRY(launch_nchw3_to_nhwc4(runs[i].x, h->act[runs[i].chunk][0], runs[i].b, kImg, kImg, runs[i].s));
    }
  }
  // A frame per workgroup pays when the sub-batch's frames fill whole rounds of CUs: one round lasts as long for 1 frame as
  // for `cus` (stand-alone at B=256: 137 us against 165 us for the three launches).
  auto fused_pays = [&](int b) {
    const int rounds = (b + h->cus - 1) / h->cus;
    return b > 0 && (long)b * 100 >= (long)rounds * h->cus * 85;
  };
  size_t skip_until[pr_hmr::kMaxChunks] = {};
  for (size_t ci = 0; ci < h->convs.size(); ++ci) {
    ConvSpec& c = h->convs[ci];
    const int li = c.layer;
    const pr_hmr::FusedBlock* alt = nullptr;
    for (const pr_hmr::FusedBlock& fb : h->fused3)
      if (fb.first == ci) alt = &fb;
    for (int i = 0; i < n; ++i) {
      const ChunkRun& r = runs[i];
      if (ci < skip_until[i]) continue;      // the block's other two launches: done by the whole-block kernel
      if (alt && fused_pays(r.b)) {
        BottleneckProblem bp;
        bp.x = h->act[r.chunk][alt->blk.in_buf]; bp.y = h->act[r.chunk][alt->blk.out_buf];
        bp.w1 = alt->blk.w; bp.w2 = alt->blk.w2b; bp.w3 = alt->blk.w3;
        bp.b1 = alt->blk.bias; bp.b2 = alt->blk.bias2b; bp.b3 = alt->blk.bias3;
        bp.B = r.b; bp.H = alt->blk.H; bp.W = alt->blk.W; bp.planes = alt->blk.bneck_planes; bp.first = false;
        if (h->profile) {
          hipEvent_t e0, e1;
          PR_HIP(hipEventCreate(&e0));
          PR_HIP(hipEventCreate(&e1));
          PR_HIP(hipEventRecord(e0, r.s));
          PR_TRY(bottleneck_bf16_launch(bp, r.s));
          PR_HIP(hipEventRecord(e1, r.s));
          h->pending.emplace_back(e0, e1);
          h->pending_layer.push_back(alt->blk.layer);
        } else {
          PR_TRY(bottleneck_bf16_launch(bp, r.s));
        }
        skip_until[i] = ci + 3;
        continue;
      }
      // a whole-Bottleneck spec (bneck_planes) is launched from its own fields: it has no conv3 output buffer of its own
      // (out3_buf = -1), so no ConvProblem is built for it
      ConvProblem p = c.bneck_planes ? ConvProblem{} : conv_problem(h, c, r.chunk, r.b);
      int cfg = c.cfg >= 0 || c.bneck_planes ? c.cfg : conv_pick_tile_cfg(p);
      if (bf && h->balanced && c.cfg < 0 && !c.bneck_planes && !c.u && conv_bal_bf16_pays(p, h->cus)) cfg = kConvCfgBalanced;
      // a Winograd layer is three launches (transform, 16 grouped GEMMs, transform); it is timed as one conv
      const bool stem_pool = ci == 0 && h->stem_s2d && h->fuse_stem;   // the stem and its max-pool as one launch
      auto go = [&]() -> int {
        if (stem_pool && !bf)
          return stem_pool_f32_launch(h->act[r.chunk][0], c.w, c.bias, h->act[r.chunk][2], r.b, r.s);
        if (stem_pool)
          return stem_pool_bf16_launch(h->act[r.chunk][0], c.w, c.bias, h->act[r.chunk][2], r.b, kImg / 2, r.s);
        if (c.bneck_planes) {
          BottleneckProblem bp;
          bp.x = h->act[r.chunk][c.in_buf]; bp.y = h->act[r.chunk][c.out_buf];
          bp.w1 = c.w; bp.w2 = c.w2b; bp.w3 = c.w3; bp.b1 = c.bias; bp.b2 = c.bias2b; bp.b3 = c.bias3;
          bp.B = r.b; bp.H = c.H; bp.W = c.W; bp.planes = c.bneck_planes; bp.first = c.bneck_first;
          bp.lead_tiles = h->b128_lead;
          return bottleneck_bf16_launch(bp, r.s);
        }
        return c.u ? conv_winograd_launch(p, c.u, h->wino_work[r.chunk], c.wino_form, r.s) : conv_launch(p, cfg, r.s);
      };
      if (h->profile) {
        hipEvent_t e0, e1;
        PR_HIP(hipEventCreate(&e0));
        PR_HIP(hipEventCreate(&e1));
        PR_HIP(hipEventRecord(e0, r.s));
        PR_TRY(go());
        PR_HIP(hipEventRecord(e1, r.s));
        h->pending.emplace_back(e0, e1);
        h->pending_layer.push_back(li);
      } else {
        PR_TRY(go());
      }
      if (ci == 0 && !stem_pool) {
        if (bf) PR_TRY(launch_maxpool_bf16(h->act[r.chunk][1], h->act[r.chunk][2], r.b, 112, 112, 64, r.s));
        else PR_TRY(launch_maxpool(h->act[r.chunk][1], h->act[r.chunk][2], r.b, 112, 112, 64, r.s));
      }
    }
  }
  for (int i = 0; i < n; ++i) {
    if (bf) PR_TRY(launch_avgpool_bf16(h->act[runs[i].chunk][h->final_buf], runs[i].xf_out, runs[i].b, 49, 2048, runs[i].s));
    else PR_TRY(launch_avgpool(h->act[runs[i].chunk][h->final_buf], runs[i].xf_out, runs[i].b, 49, 2048, runs[i].s));
  }
  return PR_OK;
}

}  // namespace
}  // namespace pr

extern "C" {

size_t pr_hmr_weight_floats(void) { return pr::hmr_weight_floats(); }
int pr_hmr_num_conv_layers(void) { return pr::kNumConv; }

int pr_hmr_create(int device, const float* weights_host, size_t n_floats, int max_batch, int precision,
                  int conv_form, pr_hmr_t** out) {
  using namespace pr;
  PR_REQUIRE(out && weights_host, "pr_hmr_create: null argument");
  PR_REQUIRE(max_batch > 0 && max_batch <= 4096, "pr_hmr_create: max_batch %d out of range", max_batch);
  PR_REQUIRE(precision == 0 || precision == 1, "pr_hmr_create: precision %d unknown (0 = fp32, 1 = bf16 encoder)", precision);
  auto form_ok = [](int f) { return f == 0 || f == 2 || f == 4 || f == 5; };
  PR_REQUIRE(conv_form == PR_CONV_FORM_DEFAULT || form_ok(conv_form) ||
                 (conv_form >= 100 && conv_form <= 555 && form_ok(conv_form / 100) && form_ok(conv_form / 10 % 10) &&
                  form_ok(conv_form % 10)),
             "pr_hmr_create: conv_form %d unknown (-1 default, 0 direct, 2 F(2x2,3x3), 4 F(4x4,3x3), 5 F(4x4,3x3) on the "
             "points 0, +-11/16, +-3/2, or three digits of those for layer2 / layer3 / layer4)", conv_form);
  PR_REQUIRE(n_floats == hmr_weight_floats(), "pr_hmr_create: blob has %zu floats, expected %zu", n_floats,
             hmr_weight_floats());
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_error("pr_hmr_create: no HIP device visible");
    return PR_ERR_NO_DEVICE;
  }
  PR_REQUIRE(device >= 0 && device < ndev, "pr_hmr_create: device %d of %d", device, ndev);
  DeviceGuard g(device);
  std::unique_ptr<pr_hmr> h(new pr_hmr);
  h->device = device;
  h->max_batch = max_batch;
  h->precision = precision;
  if (conv_form == PR_CONV_FORM_DEFAULT) {
    // POSERISK_WINOGRAD in the environment only moves the default (A/B runs of unmodified callers); an explicit
    // conv_form always wins
    conv_form = PR_CONV_FORM_BUILTIN_DEFAULT;
    if (const char* e = getenv("POSERISK_WINOGRAD")) {
      const int v = atoi(e);
      conv_form = (v == 2 || v == 4 || v == 5 || (v >= 100 && v <= 555)) ? v : 0;
    }
  }
  h->conv_form = conv_form;
  for (int st = 1; st < 4; ++st) {
    const int f = conv_form >= 100 ? (st == 1 ? conv_form / 100 : st == 2 ? conv_form / 10 % 10 : conv_form % 10) : conv_form;
    h->stage_form[st] = form_ok(f) ? f : 0;
  }
  // A/B switches: every one is read here, once per handle, into a handle field (nothing is latched per process)
  h->tune = conv_tuning_from_env();
  if (const char* e = getenv("POSERISK_FC_TILES")) h->fc_tiles = atoi(e);
  if (const char* e = getenv("POSERISK_WINOGRAD_MIN_C")) h->wino_min_c = atoi(e);
  if (const char* e = getenv("POSERISK_FUSE_DOWNSAMPLE")) h->fuse_downsample = atoi(e) != 0;   // A/B timing only
  if (const char* e = getenv("POSERISK_FUSE_CONV3")) h->fuse_conv3 = atoi(e) != 0;             // A/B timing only
  if (const char* e = getenv("POSERISK_STEM_S2D")) h->stem_s2d = atoi(e) != 0;                 // A/B timing only
  if (const char* e = getenv("POSERISK_FUSE_BOTTLENECK")) h->fuse_bottleneck = atoi(e) != 0;   // A/B timing only
  if (const char* e = getenv("POSERISK_FUSE_BOTTLENECK2")) h->fuse_bottleneck2 = atoi(e) != 0; // A/B timing only
  if (const char* e = getenv("POSERISK_FUSE_STEM")) h->fuse_stem = atoi(e) != 0;               // A/B timing only
  if (const char* e = getenv("POSERISK_EXPAND_REGS")) h->expand_regs = atoi(e) != 0;           // A/B timing only
  if (const char* e = getenv("POSERISK_FUSE_BOTTLENECK3")) h->fuse_bottleneck3 = atoi(e) != 0; // A/B timing only
  if (const char* e = getenv("POSERISK_B128_LEAD")) h->b128_lead = atoi(e);                    // A/B timing only
  if (const char* e = getenv("POSERISK_BALANCED")) h->balanced = atoi(e) != 0;                 // A/B timing only
  (void)hipDeviceGetAttribute(&h->cus, hipDeviceAttributeMultiprocessorCount, h->device);
  if (precision == 1) h->panel_max_k = 0;   // bf16: off until measured (POSERISK_PANEL_MAX_K)
  if (const char* e = getenv("POSERISK_PANEL_MAX_K")) h->panel_max_k = atoi(e);                // A/B timing only (0 = off)
  if (const char* e = getenv("POSERISK_REGW")) h->regw = atoi(e) != 0;                         // A/B timing only
  if (const char* e = getenv("POSERISK_SPLITK")) h->splitk = std::max(1, std::min(atoi(e), 8));    // A/B timing only (1 = off)
  int st = build(h.get(), weights_host, n_floats);
  if (st == PR_OK) {
    int n = 1;  // sub-batch streams: 1 unless POSERISK_HMR_STREAMS / pr_hmr_set_streams ask for more
    if (const char* e = getenv("POSERISK_HMR_STREAMS")) n = atoi(e);
    n = std::max(1, std::min(n, (int)pr_hmr::kMaxChunks));
    st = set_chunks(h.get(), std::min(n, max_batch));
  }
  if (st != PR_OK) {
    for (float* p : h->dev_allocs) (void)hipFree(p);
    for (float* p : h->act_allocs) (void)hipFree(p);
    return st;
  }
  *out = h.release();
  return PR_OK;
}

int pr_hmr_destroy(pr_hmr_t* h) {
  if (!h) return PR_OK;
  pr::DeviceGuard g(h->device);
  for (auto& pe : h->pending) {
    (void)hipEventDestroy(pe.first);
    (void)hipEventDestroy(pe.second);
  }
  for (float* p : h->dev_allocs) (void)hipFree(p);
  for (float* p : h->act_allocs) (void)hipFree(p);
  for (int c = 0; c < pr_hmr::kMaxChunks; ++c) {
    if (h->streams[c]) (void)hipStreamDestroy(h->streams[c]);
    if (h->ev_join[c]) (void)hipEventDestroy(h->ev_join[c]);
  }
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  delete h;
  return PR_OK;
}

int pr_hmr_set_concurrency(pr_hmr_t* h, int n_in_flight) {
  PR_REQUIRE(h, "pr_hmr_set_concurrency: null handle");
  PR_REQUIRE(n_in_flight >= 1, "pr_hmr_set_concurrency: %d handles in flight", n_in_flight);
  if (!getenv("POSERISK_REGW_PER_CU")) h->tune.regw_per_cu = n_in_flight >= 2 ? 1 : 2;   // the env is the A/B override
  return PR_OK;
}

int pr_hmr_set_streams(pr_hmr_t* h, int n_streams) {
  PR_REQUIRE(h, "pr_hmr_set_streams: null handle");
  pr::DeviceGuard g(h->device);
  return pr::set_chunks(h, std::min(n_streams, h->max_batch));
}

int pr_hmr_forward(pr_hmr_t* h, const float* x_dev, int B, float* rotmat_dev, float* betas_dev,
                   float* cam_dev, float* xf_dev, float* pose6d_dev, void* stream) {
  using namespace pr;
  PR_REQUIRE(B >= 0, "pr_hmr_forward: negative batch");
  if (B == 0) return PR_OK;
  PR_REQUIRE(h && x_dev, "pr_hmr_forward: null argument");
  if (B > h->max_batch) {
    set_error("pr_hmr_forward: batch %d exceeds max_batch %d", B, h->max_batch);
    return PR_ERR_CAPACITY;
  }
  if (B == 0) return PR_OK;
  hipStream_t s = (hipStream_t)stream;
  // Profiling runs serially on the caller's stream so that each conv's event bracket is its own time.
  int nch = h->profile ? 1 : std::min(h->n_chunks, B);
  if (nch > 1 && ceil_div(B, nch) > h->chunk_cap) nch = 1;  // larger than the concurrent buffers: serial passes
  const size_t frame = (size_t)3 * kImg * kImg;
  if (nch == 1) {
    // one sub-batch at a time on the caller's stream (more than one pass if B exceeds a chunk's buffers)
    for (int b0 = 0; b0 < B; b0 += h->chunk_cap) {
      ChunkRun r{0, x_dev + b0 * frame, std::min(h->chunk_cap, B - b0), h->xf + (size_t)b0 * 2048, s};
      PR_TRY(encode_chunks(h, &r, 1));
    }
  } else {
    ChunkRun runs[pr_hmr::kMaxChunks];
    PR_HIP(hipEventRecord(h->ev_fork, s));
    for (int c = 0; c < nch; ++c) {
      const int b0 = (int)((long)c * B / nch), b1 = (int)((long)(c + 1) * B / nch);
      runs[c] = ChunkRun{c, x_dev + b0 * frame, b1 - b0, h->xf + (size_t)b0 * 2048, h->streams[c]};
      PR_HIP(hipStreamWaitEvent(h->streams[c], h->ev_fork, 0));
    }
    PR_TRY(encode_chunks(h, runs, nch));
    for (int c = 0; c < nch; ++c) {
      PR_HIP(hipEventRecord(h->ev_join[c], h->streams[c]));
      PR_HIP(hipStreamWaitEvent(s, h->ev_join[c], 0));
    }
  }
  if (xf_dev) PR_HIP(hipMemcpyAsync(xf_dev, h->xf, (size_t)B * 2048 * sizeof(float), hipMemcpyDeviceToDevice, s));
  if (!rotmat_dev && !betas_dev && !cam_dev && !pose6d_dev) return PR_OK;
  // regressor: h_static = xf*W1x^T + b1 once; 3 x { h1 = state*W1s^T + h_static; h2 = h1*W2^T + b2;
  //                                               state += h2*Wdec^T + bdec }
  PR_TRY(launch_state_init(h->init157, h->state, B, s));
  PR_TRY(fc_launch(h, h->fc1x, h->xf, nullptr, h->h_static, B, true, s));
  for (int it = 0; it < 3; ++it) {
    PR_TRY(fc_launch(h, h->fc1s, h->state, h->h_static, h->h1, B, false, s));
    PR_TRY(fc_launch(h, h->fc2, h->h1, nullptr, h->h2, B, true, s));
    PR_TRY(fc_launch(h, h->dec, h->h2, h->state, h->state, B, true, s));
  }
  PR_TRY(launch_regressor_finalize(h->state, rotmat_dev, betas_dev, cam_dev, pose6d_dev, B, s));
  return PR_OK;
}

int pr_hmr_conv_form(pr_hmr_t* h) { return h ? h->conv_form : PR_ERR_INVALID; }

int pr_hmr_plan_counts(pr_hmr_t* h, int B, int* conv_launches, int* winograd_layers) {
  using namespace pr;
  PR_REQUIRE(h && B > 0 && B <= h->max_batch, "pr_hmr_plan_counts: need a handle and a batch within its capacity");
  // as encode_chunks walks the plan for one sub-batch of min(B, chunk_cap) frames (the passes of a larger batch repeat it)
  const int b = std::min(B, h->chunk_cap);
  const int rounds = (b + h->cus - 1) / h->cus;
  const bool fused3 = (long)b * 100 >= (long)rounds * h->cus * 85;
  int launches = 0, wino = 0;
  size_t skip_until = 0;
  for (size_t ci = 0; ci < h->convs.size(); ++ci) {
    if (ci < skip_until) continue;
    bool alt = false;
    for (const pr_hmr::FusedBlock& fb : h->fused3) alt = alt || fb.first == ci;
    if (alt && fused3) {
      skip_until = ci + 3;
      ++launches;
      continue;
    }
    ++launches;
    if (h->convs[ci].u) ++wino;
  }
  const int passes = (B + h->chunk_cap - 1) / h->chunk_cap;
  if (conv_launches) *conv_launches = launches * passes;
  if (winograd_layers) *winograd_layers = wino * passes;
  return PR_OK;
}

int pr_hmr_profile_enable(pr_hmr_t* h, int on) {
  PR_REQUIRE(h, "pr_hmr_profile_enable: null handle");
  h->profile = on != 0;
  return PR_OK;
}

int pr_hmr_profile_read(pr_hmr_t* h, float* ms, int* launches, double* flops_per_frame, double* mfma_flops_per_frame,
                        int n_layers) {
  using namespace pr;
  PR_REQUIRE(h && n_layers == kNumConv, "pr_hmr_profile_read: need %d layers", kNumConv);
  for (size_t i = 0; i < h->pending.size(); ++i) {
    float t = 0.f;
    PR_HIP(hipEventSynchronize(h->pending[i].second));
    PR_HIP(hipEventElapsedTime(&t, h->pending[i].first, h->pending[i].second));
    h->prof_ms[h->pending_layer[i]] += t;
    h->prof_n[h->pending_layer[i]] += 1;
    (void)hipEventDestroy(h->pending[i].first);
    (void)hipEventDestroy(h->pending[i].second);
  }
  h->pending.clear();
  h->pending_layer.clear();
  for (int i = 0; i < kNumConv; ++i) {
    if (ms) ms[i] = h->prof_ms[i];
    if (launches) launches[i] = h->prof_n[i];
    if (flops_per_frame) flops_per_frame[i] = 0.0;   // a downsample branch fused into its conv3 is counted there
    if (mfma_flops_per_frame) mfma_flops_per_frame[i] = 0.0;
    h->prof_ms[i] = 0.f;
    h->prof_n[i] = 0;
  }
  for (const ConvSpec& c : h->convs) {
    if (flops_per_frame) flops_per_frame[c.layer] = 2.0 * c.macs_per_frame();
    if (mfma_flops_per_frame) {
      mfma_flops_per_frame[c.layer] = 2.0 * c.mfma_macs_per_frame(h->precision == 1 ? 64 : kConvBK);
      // the fused fp32 stem multiplies 13 steps x 12 k per output (stem_pool_f32.hip), not the 192 of the 4x4 x 12 taps
      if (c.out_hw && h->precision == 0 && h->stem_s2d && h->fuse_stem)
        mfma_flops_per_frame[c.layer] = 2.0 * (double)c.Ho() * c.Wo() * c.Cout * 156.0;
    }
  }
  return PR_OK;
}

}  // extern "C"
