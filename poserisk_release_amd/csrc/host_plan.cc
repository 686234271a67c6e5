// Device-free half of pr_hmr_create (see host_plan.h): blob layout, BatchNorm folding, weight packing, the launch plan.
// Compiled into libposerisk_hip.so by hipcc as plain C++ and, for tests/native/host_plan_check.cc, by g++ with
// -fsanitize=address,undefined.  No HIP header may be included here.
#include "host_plan.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

namespace pr {

// ---- weight packing -------------------------------------------------------------------------------------------------
int conv_kpad_bf16(int K) { return ceil_div(K, 64) * 64; }

unsigned short f32_to_bf16_host(float f) {  // round-to-nearest-even; NaN stays NaN
  unsigned u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

void conv_pack_weights_bf16(const float* w, const double* scale, int Cout, int Cin_real, int cin_pad, int KH,
                            int KW, unsigned short* out) {
  const int Kpad = conv_kpad_bf16(KH * KW * cin_pad);
  for (int o = 0; o < Cout; ++o) {
    unsigned short* row = out + (size_t)o * Kpad;
    for (int k = 0; k < Kpad; ++k) row[k] = 0;
    const double s = scale ? scale[o] : 1.0;
    for (int ci = 0; ci < Cin_real; ++ci)
      for (int kh = 0; kh < KH; ++kh)
        for (int kw = 0; kw < KW; ++kw)
          row[conv_k_index_bf16(kh * KW + kw, ci, KH * KW, cin_pad)] =
              f32_to_bf16_host((float)((double)w[(((size_t)o * Cin_real + ci) * KH + kh) * KW + kw] * s));
  }
}

void conv_pack_weights(const float* w, const double* scale, int Cout, int Cin_real, int cin_pad,
                       int KH, int KW, float* out) {
  const int K = KH * KW * cin_pad;
  const int Kpad = ceil_div(K, kConvBK) * kConvBK;
  for (int o = 0; o < Cout; ++o) {
    float* row = out + (size_t)o * Kpad;
    for (int k = 0; k < Kpad; ++k) row[k] = 0.f;
    const double s = scale ? scale[o] : 1.0;
    for (int ci = 0; ci < Cin_real; ++ci)
      for (int kh = 0; kh < KH; ++kh)
        for (int kw = 0; kw < KW; ++kw) {
          const double v = (double)w[(((size_t)o * Cin_real + ci) * KH + kh) * KW + kw] * s;
          row[(kh * KW + kw) * cin_pad + ci] = (float)v;
        }
  }
}

void conv_winograd_pack_weights(const float* w, const double* scale, int Cout, int Cin, int form, float* out) {
  // U = G g G^T in double, one rounding to fp32;  layout [(m+2)^2][Cout][Cin]
  static const double G2[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
  static const double G4[6][3] = {{1.0 / 4, 0, 0},          {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                  {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6},  {0, 0, 1}};
  // form 5: G[j] = [1, p_j, p_j^2] / N_j on the points 0, +-a, +-b (N_0 = a^2 b^2, N_a = 2 a^2 (a^2 - b^2),
  // N_b = 2 b^2 (b^2 - a^2)), last row [0, 0, 1]; with a = 1, b = 2 these are G4's rows
  double G5[6][3];
  {
    const double a = kWa, b = kWb, a2 = a * a, b2 = b * b, Na = 2 * a2 * (a2 - b2), Nb = 2 * b2 * (b2 - a2);
    const double rows[6][3] = {{1 / (a2 * b2), 0, 0}, {1 / Na, a / Na, a2 / Na}, {1 / Na, -a / Na, a2 / Na},
                               {1 / Nb, b / Nb, b2 / Nb}, {1 / Nb, -b / Nb, b2 / Nb}, {0, 0, 1}};
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 3; ++j) G5[i][j] = rows[i][j];
  }
  const int m = conv_winograd_tile(form);
  const int n = m + 2;
  const double(*G)[3] = form == 5 ? G5 : m == 4 ? G4 : G2;
  for (int o = 0; o < Cout; ++o)
    for (int ci = 0; ci < Cin; ++ci) {
      const float* g = w + ((size_t)o * Cin + ci) * 9;
      const double sc = scale ? scale[o] : 1.0;
      double t[6][3];
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < 3; ++j)
          t[i][j] = G[i][0] * ((double)g[j] * sc) + G[i][1] * ((double)g[3 + j] * sc) + G[i][2] * ((double)g[6 + j] * sc);
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
          const double u = t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2];
          out[((size_t)(n * i + j) * Cout + o) * Cin + ci] = (float)u;
        }
    }
}

// Packed weight rows for the transposed MFMAs: row 32 T + i of the packed matrix is output channel 32 T + sigma(i),
// sigma(i) = 16 ((i >> 2) & 1) + 4 (i >> 3) + (i & 3), so that accumulator register r of lane half h (MFMA row
// (r & 3) + 8 (r >> 2) + 4 h) is channel 32 T + 16 h + r.  `src` is [rows][K] (rows % 32 == 0).
void bottleneck_pack_rows_bf16(const unsigned short* src, int rows, int K, unsigned short* dst) {
  for (int o = 0; o < rows; ++o) {
    const int T = o >> 5, i = o & 31;
    const int sigma = 16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3);
    memcpy(dst + (size_t)o * K, src + (size_t)(32 * T + sigma) * K, (size_t)K * 2);
  }
}

// conv2's weights [256][2304] (rows permuted by bottleneck_pack_rows_bf16, k slice-major) -> the order phase 2's waves of
// bottleneck256_bf16 load them in: [stage 72 of 32 k][channel-tile pair 4][tile of the pair 2][k-step 2][lane 64] pieces of
// 8 k; lane (i, h) of k-step ks holds row 32 (2 cp + c) + i, k = 32 st32 + 16 ks + 8 h .. + 7.
void bottleneck256_pack_w2_frags_bf16(const unsigned short* rows, unsigned short* dst) {
  constexpr int kP = 256;
  for (int st = 0; st < 72; ++st)
    for (int cp = 0; cp < 4; ++cp)
      for (int c = 0; c < 2; ++c)
        for (int ks = 0; ks < 2; ++ks)
          for (int l = 0; l < 64; ++l) {
            const int i = l & 31, h = l >> 5;
            const unsigned short* src = rows + (size_t)(32 * (2 * cp + c) + i) * (9 * kP) + 32 * st + 16 * ks + 8 * h;
            std::copy(src, src + 8, dst + (((((size_t)st * 4 + cp) * 2 + c) * 2 + ks) * 64 + l) * 8);
          }
}

// conv3's weights [1024][256] (rows permuted) -> [64-channel group 16][tile 2][k-step 16][lane 64] pieces of 8 k
void bottleneck256_pack_w3_frags_bf16(const unsigned short* rows, unsigned short* dst) {
  constexpr int kP = 256;
  for (int cq = 0; cq < 16; ++cq)
    for (int c = 0; c < 2; ++c)
      for (int ks = 0; ks < 16; ++ks)
        for (int l = 0; l < 64; ++l) {
          const int i = l & 31, h = l >> 5;
          const unsigned short* src = rows + (size_t)(32 * (2 * cq + c) + i) * kP + 16 * ks + 8 * h;
          std::copy(src, src + 8, dst + ((((size_t)cq * 2 + c) * 16 + ks) * 64 + l) * 8);
        }
}

bool expand_res_bf16_fits(int K, int N) { return (K == 128 && N == 512) || (K == 256 && N == 1024); }
bool expand_dual_bf16_fits(int K1, int K2, int N) { return K1 == 128 && K2 == 256 && N == 512; }
bool bottleneck256_bf16_fits(int H, int W) { return H >= 1 && W >= 1 && H * W <= 224; }   // 32 x 7 pixel tiles: kMaxPix of the kernel

// ---- the plan -----------------------------------------------------------------------------------------------------------
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

size_t hmr_weight_floats_impl() {
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

struct Ctx {
  HmrPlan* h;
  PlanSink* sink;
};

int upload(Ctx& cx, const std::vector<float>& host, float** out) {
  return cx.sink->upload(host.data(), host.size() * sizeof(float), out);
}

int dev_alloc(Ctx& cx, size_t floats, float** out) { return cx.sink->zeros(std::max<size_t>(floats, 4) * sizeof(float), out); }

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
int packed_k(const HmrPlan* h, int K) { return h->precision == 1 ? conv_kpad_bf16(K) : ceil_div(K, kConvBK) * kConvBK; }

// Folded weights of one or two convolutions (two: a conv3 and the downsample branch summed into it) -> device
// rows [Cout][Kpad(f1) + Kpad(f2)] in the handle's precision.
int upload_packed(Ctx& cx, const ConvSpec& spec, const FoldedConv& f1, const FoldedConv* f2, float** out) {
  HmrPlan* const h = cx.h;
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
    return upload(cx, as_f, out);
  }
  std::vector<float> a((size_t)spec.Cout * K1), b((size_t)spec.Cout * K2), packed((size_t)spec.Cout * (K1 + K2));
  conv_pack_weights(f1.w, f1.scale.data(), spec.Cout, spec.Cin_real, spec.Cin, spec.k, spec.k, a.data());
  if (f2) conv_pack_weights(f2->w, f2->scale.data(), spec.Cout, spec.Cin2, spec.Cin2, 1, 1, b.data());
  for (int o = 0; o < spec.Cout; ++o) {
    memcpy(&packed[(size_t)o * (K1 + K2)], &a[(size_t)o * K1], (size_t)K1 * 4);
    if (K2) memcpy(&packed[(size_t)o * (K1 + K2) + K1], &b[(size_t)o * K2], (size_t)K2 * 4);
  }
  return upload(cx, packed, out);
}

// conv weight + its BatchNorm -> packed folded weights and bias on device.  `second` (a conv3 whose block has a
// downsample branch, fused form): the branch's conv + BatchNorm follow in the blob and are summed into this conv.
int add_conv(Ctx& cx, BlobReader& br, ConvSpec spec, bool second = false) {
  HmrPlan* const h = cx.h;
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
  PR_TRY(upload_packed(cx, spec, f1, second ? &f2 : nullptr, &spec.w));
  PR_TRY(upload(cx, bias, &spec.bias));
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
    PR_TRY(upload(cx, u, &spec.u));
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
int make_fc(Ctx& cx, const float* w, const float* b, int N, int src_cols, int col0, int K_real,
            int Kpad, int Npad, FcSpec* out) {
  std::vector<float> packed((size_t)Npad * Kpad, 0.f), bias(Npad, 0.f);
  for (int n = 0; n < N; ++n) {
    for (int k = 0; k < K_real; ++k) packed[(size_t)n * Kpad + k] = w[(size_t)n * src_cols + col0 + k];
    if (b) bias[n] = b[n];
  }
  out->K = Kpad;
  out->N = Npad;
  PR_TRY(upload(cx, packed, &out->w));
  PR_TRY(upload(cx, bias, &out->bias));
  return PR_OK;
}

int build(Ctx& cx, const float* blob, size_t n_floats) {
  HmrPlan* const h = cx.h;
  BlobReader br{blob, n_floats};
  // stem: conv1 7x7/2 (input padded to 4 channels) -> act[1]; maxpool -> act[2]
  ConvSpec c1{3, h->precision == 1 ? 8 : 4, 64, 7, 2, 3, kImg, kImg, 1, 0, 1, -1};
  if (h->stem_s2d) {
    c1 = ConvSpec{12, h->precision == 1 ? 16 : 12, 64, 4, 1, 2, kImg / 2, kImg / 2, 1, 0, 1, -1};
    c1.out_hw = kImg / 2;
    c1.macs_fixed = (double)(kImg / 2) * (kImg / 2) * 64 * 3 * 49;
  }
  PR_TRY(add_conv(cx, br, c1));
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
          return upload(cx, as_f, out);
        };
        auto bias_of = [&](const FoldedConv& f, const FoldedConv* g, float** out) -> int {
          std::vector<float> bv(f.bias.size());
          for (size_t o = 0; o < bv.size(); ++o) bv[o] = (float)(f.bias[o] + (g ? g->bias[o] : 0.0));
          return upload(cx, bv, out);
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
        HmrPlan::FusedBlock fb;
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
          return upload(cx, as_f, out);
        };
        auto bias_of = [&](const FoldedConv& f, float** out) -> int {
          std::vector<float> bv(f.bias.size());
          for (size_t o = 0; o < bv.size(); ++o) bv[o] = (float)f.bias[o];
          return upload(cx, bv, out);
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
      PR_TRY(add_conv(cx, br, a));
      PR_TRY(add_conv(cx, br, bb));
      if (b == 0 && h->fuse_downsample) {
        // relu(bn3(conv3(t2)) + bn_d(conv_d(x))) as ONE GEMM over K = [t2's channels | x's channels]: the downsample
        // tensor is never written or read back (execution order: the branch is layer n, conv3 layer n + 1)
        cc.res_buf = -1;
        cc.in2_buf = cur; cc.Cin2 = inpl; cc.H2 = H; cc.stride2 = stride;
        cc.layer2 = layer++;
        cc.layer = layer++;
        PR_TRY(add_conv(cx, br, cc, true));
      } else if (b == 0) {
        // blob order is conv3/bn3 then downsample; execution order is downsample before conv3
        const size_t mark = h->convs.size();
        PR_TRY(add_conv(cx, br, cc));
        ConvSpec dd{inpl, inpl, pl * 4, 1, stride, 0, H, H, 0, cur, ds, -1};
        dd.stage = L;
        PR_TRY(add_conv(cx, br, dd));
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
        PR_TRY(upload_packed(cx, cc, f3, nullptr, &f.w3));
        PR_TRY(upload(cx, bias3, &f.bias3));
        f.N3 = cc.Cout; f.res3_buf = cc.res_buf; f.out3_buf = cc.out_buf;
        f.layer2 = f.layer;
        f.layer = layer++;
      } else {
        cc.layer = layer++;
        PR_TRY(add_conv(cx, br, cc));
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
  PR_TRY(make_fc(cx, fc1w, fc1b, 1024, 2205, 0, 2048, 2048, 1024, &h->fc1x));
  PR_TRY(make_fc(cx, fc1w, nullptr, 1024, 2205, 2048, 157, kStateStride, 1024, &h->fc1s));
  PR_TRY(make_fc(cx, fc2w, fc2b, 1024, 1024, 0, 1024, 1024, 1024, &h->fc2));
  {
    std::vector<float> decw((size_t)157 * 1024), decb(157);
    std::copy(dpw, dpw + (size_t)144 * 1024, decw.begin());
    std::copy(dsw, dsw + (size_t)10 * 1024, decw.begin() + (size_t)144 * 1024);
    std::copy(dcw, dcw + (size_t)3 * 1024, decw.begin() + (size_t)154 * 1024);
    std::copy(dpb, dpb + 144, decb.begin());
    std::copy(dsb, dsb + 10, decb.begin() + 144);
    std::copy(dcb, dcb + 3, decb.begin() + 154);
    PR_TRY(make_fc(cx, decw.data(), decb.data(), 157, 1024, 0, 1024, 1024, kStateStride, &h->dec));
  }
  {
    std::vector<float> init(160, 0.f);
    std::copy(ip, ip + 144, init.begin());
    std::copy(is, is + 10, init.begin() + 144);
    std::copy(ic, ic + 3, init.begin() + 154);
    PR_TRY(upload(cx, init, &h->init157));
  }

  // workspaces
  const size_t B = (size_t)h->max_batch;
  PR_TRY(dev_alloc(cx, B * 2048, &h->xf));
  PR_TRY(dev_alloc(cx, B * 1024, &h->h_static));
  PR_TRY(dev_alloc(cx, B * 1024, &h->h1));
  PR_TRY(dev_alloc(cx, B * 1024, &h->h2));
  PR_TRY(dev_alloc(cx, B * kStateStride, &h->state));
  return PR_OK;
}

}  // namespace

size_t hmr_weight_floats() { return hmr_weight_floats_impl(); }

bool hmr_conv_form_valid(int conv_form) {
  auto form_ok = [](int f) { return f == 0 || f == 2 || f == 4 || f == 5; };
  return conv_form == PR_CONV_FORM_DEFAULT || form_ok(conv_form) ||
         (conv_form >= 100 && conv_form <= 555 && form_ok(conv_form / 100) && form_ok(conv_form / 10 % 10) && form_ok(conv_form % 10));
}

void hmr_plan_configure(HmrPlan* h, int precision, int conv_form, int max_batch) {
  auto form_ok = [](int f) { return f == 0 || f == 2 || f == 4 || f == 5; };
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
  if (const char* e = getenv("POSERISK_FC_TILES")) h->fc_tiles = atoi(e);
  if (const char* e = getenv("POSERISK_FC_SHAPE")) h->fc_shape = atoi(e);                      // A/B timing only (same bits)
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
  if (precision == 1) h->panel_max_k = 0;   // bf16: off until measured (POSERISK_PANEL_MAX_K)
  if (const char* e = getenv("POSERISK_PANEL_MAX_K")) h->panel_max_k = atoi(e);                // A/B timing only (0 = off)
  if (const char* e = getenv("POSERISK_REGW")) h->regw = atoi(e) != 0;                         // A/B timing only
  if (const char* e = getenv("POSERISK_SPLITK")) h->splitk = std::max(1, std::min(atoi(e), 8));    // A/B timing only (1 = off)
}

int hmr_plan_build(HmrPlan* plan, const float* blob, size_t n_floats, PlanSink& sink) {
  PR_REQUIRE(plan && blob, "hmr plan: null argument");
  PR_REQUIRE(n_floats == hmr_weight_floats(), "pr_hmr_create: blob has %zu floats, expected %zu", n_floats, hmr_weight_floats());
  PR_REQUIRE(plan->max_batch > 0 && plan->max_batch <= 4096, "pr_hmr_create: max_batch %d out of range", plan->max_batch);
  Ctx cx{plan, &sink};
  return build(cx, blob, n_floats);
}

HmrChunkSizes hmr_chunk_sizes(const HmrPlan& plan, int chunk_cap) {
  HmrChunkSizes z{};
  const size_t cb = (size_t)chunk_cap;
  const size_t fmap = (size_t)112 * 112 * 64;  // == 56*56*256, the largest feature map per frame
  // element counts; bf16 buffers hold the same number of elements in half the bytes (input: 8 channels)
  z.act0_floats = cb * kImg * kImg * 4;
  z.act_floats = plan.precision == 1 ? (cb * fmap + 1) / 2 : cb * fmap;
  z.wino_floats = cb * plan.wino_floats_per_frame;
  for (const ConvSpec& cs : plan.convs)
    if (cs.splitk > 1) {
      const size_t tiles = (size_t)ceil_div((int)(cb * cs.Ho() * cs.Wo()), 64) * (cs.Cout / 64);
      z.slab_floats = std::max(z.slab_floats, tiles * cs.splitk * 4096);
      z.tickets = std::max(z.tickets, tiles);
    }
  return z;
}

int hmr_split_batch(int B, int chunk_cap, int n_chunks, bool serial, int* sizes, int max_sizes, bool* concurrent) {
  int nch = serial ? 1 : std::min(n_chunks, B);
  if (nch > 1 && ceil_div(B, nch) > chunk_cap) nch = 1;   // larger than the concurrent buffers: serial passes
  int n = 0;
  if (nch <= 1) {
    for (int b0 = 0; b0 < B && n < max_sizes; b0 += chunk_cap) sizes[n++] = std::min(chunk_cap, B - b0);
  } else {
    for (int c = 0; c < nch && n < max_sizes; ++c) sizes[n++] = (int)((long)(c + 1) * B / nch) - (int)((long)c * B / nch);
  }
  if (concurrent) *concurrent = nch > 1;
  return n;
}

void hmr_plan_counts(const HmrPlan& plan, int B, int chunk_cap, int n_chunks, bool serial, int* conv_launches, int* winograd_layers) {
  // as encode_chunks walks the plan, once per sub-batch pr_hmr_forward really runs: the last serial pass is shorter and
  // concurrent shares are B / n frames each, and whether a whole-block kernel pays is decided per sub-batch
  std::vector<int> sizes((size_t)std::max(n_chunks, ceil_div(std::max(B, 1), std::max(chunk_cap, 1))) + 1);
  const int n = hmr_split_batch(B, chunk_cap, n_chunks, serial, sizes.data(), (int)sizes.size(), nullptr);
  int launches = 0, wino = 0;
  for (int i = 0; i < n; ++i) {
    const bool fused3 = hmr_fused3_pays(sizes[i], plan.cus);
    size_t skip_until = 0;
    for (size_t ci = 0; ci < plan.convs.size(); ++ci) {
      if (ci < skip_until) continue;
      bool alt = false;
      for (const HmrPlan::FusedBlock& fb : plan.fused3) alt = alt || fb.first == ci;
      ++launches;
      if (alt && fused3) {
        skip_until = ci + 3;
        continue;
      }
      if (plan.convs[ci].u) ++wino;
    }
  }
  if (conv_launches) *conv_launches = launches;
  if (winograd_layers) *winograd_layers = wino;
}

}  // namespace pr
