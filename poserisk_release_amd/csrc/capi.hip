// C-ABI entry points that are thin wrappers (error state, stand-alone kernels, the per-batch
// driver).  pr_hmr_* lives in hmr.hip, pr_smpl_* in smpl.hip.
#include <vector>

#include "conv_igemm.h"
#include "frame_kernels.h"

namespace pr {
namespace {
thread_local std::string g_last_error;
thread_local bool g_stream_declared = false;
thread_local hipStream_t g_declared_stream = nullptr;
}
void declared_stream(bool* declared, hipStream_t* s) {
  *declared = g_stream_declared;
  *s = g_declared_stream;
}
void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_last_error = buf;
}
}  // namespace pr

#define PR_STR_(x) #x
#define PR_STR(x) PR_STR_(x)

extern "C" {

const char* pr_last_error(void) { return pr::g_last_error.c_str(); }
int pr_abi_version(void) { return 10; }

int pr_declare_stream(void* stream, int declared) {
  pr::g_stream_declared = declared != 0;
  pr::g_declared_stream = declared ? (hipStream_t)stream : nullptr;
  return PR_OK;
}

// What this binary is: the shipped build says "release"; ablation / experiment builds (POSERISK_CXXFLAGS) name their macros,
// so that a bench record taken on one cannot be mistaken for the shipped library's.
const char* pr_build_info(void) {
  return "gfx950"
#ifdef PR_TIMING_HOOKS
         " +PR_TIMING_HOOKS(results may be wrong)"
#endif
#ifdef PR_EXPERIMENT
         " +PR_EXPERIMENT=" PR_STR(PR_EXPERIMENT)
#endif
#if !defined(PR_TIMING_HOOKS) && !defined(PR_EXPERIMENT)
         " release"
#endif
      ;
}

int pr_rot6d_to_rotmat(const float* pose6d_dev, int N, float* rotmat_dev, void* stream) {
  PR_REQUIRE(pose6d_dev && rotmat_dev && N >= 0, "pr_rot6d_to_rotmat: bad argument");
  return pr::launch_rot6d(pose6d_dev, rotmat_dev, (long)N * 24, (hipStream_t)stream);
}

int pr_pose_to_euler(const float* rotmat_dev, int N, float* axis_angle_dev, double* euler_deg_dev,
                     int32_t* status_dev, void* stream) {
  PR_REQUIRE(rotmat_dev && axis_angle_dev && euler_deg_dev && N >= 0, "pr_pose_to_euler: bad argument");
  return pr::launch_pose_to_euler(rotmat_dev, N, axis_angle_dev, euler_deg_dev, status_dev,
                                  (hipStream_t)stream);
}

int pr_axis_angle_to_euler(const float* axis_angle_dev, int N, double* euler_deg_dev, int32_t* status_dev,
                           void* stream) {
  PR_REQUIRE(axis_angle_dev && euler_deg_dev && N >= 0, "pr_axis_angle_to_euler: bad argument");
  return pr::launch_pose_to_euler(nullptr, N, const_cast<float*>(axis_angle_dev), euler_deg_dev, status_dev,
                                  (hipStream_t)stream);
}

int pr_reba(const double* euler_deg_dev, int N, const pr_reba_info* info, int32_t* out_dev, void* stream) {
  PR_REQUIRE(euler_deg_dev && info && out_dev && N >= 0, "pr_reba: bad argument");
  return pr::launch_reba(euler_deg_dev, N, *info, out_dev, (hipStream_t)stream);
}

int pr_rula(const double* euler_deg_dev, int N, const pr_rula_info* info, int32_t* out_dev, void* stream) {
  PR_REQUIRE(euler_deg_dev && info && out_dev && N >= 0, "pr_rula: bad argument");
  return pr::launch_rula(euler_deg_dev, N, *info, out_dev, (hipStream_t)stream);
}

int pr_crop_frames(const uint8_t* frames_dev, int F, int H, int W, int bgr, const int32_t* frame_idx_dev,
                   const float* bboxes_dev, int N, float scale, float* crops_dev, int32_t* status_dev,
                   void* stream) {
  PR_REQUIRE(frames_dev && bboxes_dev && crops_dev, "pr_crop_frames: null argument");
  PR_REQUIRE(F > 0 && H > 0 && W > 0 && H < 32768 && W < 32768 && N >= 0 && scale > 0, "pr_crop_frames: bad geometry");
  PR_REQUIRE(frame_idx_dev || N <= F, "pr_crop_frames: %d boxes for %d frames without a frame index", N, F);
  return pr::launch_crop_frames(frames_dev, F, H, W, bgr, frame_idx_dev, bboxes_dev, N, scale, crops_dev,
                                status_dev, (hipStream_t)stream);
}

int pr_conv_num_tile_cfgs(void) { return pr::conv_num_tile_cfgs(); }

int pr_conv2d_nhwc(int device, const void* x_dev, const float* w_host, const float* bias_host,
                   const void* res_dev, void* y_dev, int B, int H, int W, int Cin, int Cin_real, int Cout,
                   int KH, int KW, int stride, int pad, int relu, int tile_cfg, int precision, int repeats,
                   float* ms_out, void* stream) {
  using namespace pr;
  PR_REQUIRE(x_dev && w_host && y_dev, "pr_conv2d_nhwc: null argument");
  PR_REQUIRE(precision == 0 || precision == 1, "pr_conv2d_nhwc: precision %d unknown", precision);
  PR_REQUIRE(precision == 0 || Cin % 8 == 0, "pr_conv2d_nhwc: bf16 needs Cin %% 8 == 0");
  PR_REQUIRE(Cin_real > 0 && Cin_real <= Cin && Cout % 64 == 0, "pr_conv2d_nhwc: bad channels");
  PR_REQUIRE(stride > 0 && pad >= 0 && KH > 0 && KW > 0, "pr_conv2d_nhwc: bad geometry");
  DeviceGuard g(device);
  hipStream_t s = (hipStream_t)stream;
  PR_TRY(refuse_if_capturing(s, "stand-alone test entry"));   // allocates and synchronises: never inside a capture
  ConvProblem p;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad;
  p.Ho = (H + 2 * pad - KH) / stride + 1;
  p.Wo = (W + 2 * pad - KW) / stride + 1;
  p.relu = relu;
  PR_REQUIRE(p.Ho > 0 && p.Wo > 0, "pr_conv2d_nhwc: empty output");
  p.precision = precision;
  p.tune = conv_tuning_from_env();
  // device scratch of this call; freed on every return path
  struct Scratch {
    float *wd = nullptr, *bd = nullptr, *work = nullptr, *slab = nullptr;
    int* tickets = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ~Scratch() {
      if (slab) (void)hipFree(slab);
      if (tickets) (void)hipFree(tickets);
      if (wd) (void)hipFree(wd);
      if (bd) (void)hipFree(bd);
      if (work) (void)hipFree(work);
      if (e0) (void)hipEventDestroy(e0);
      if (e1) (void)hipEventDestroy(e1);
    }
  } sc;
  float *&wd = sc.wd, *&bd = sc.bd, *&work = sc.work;
  const bool wino = tile_cfg == -2 || tile_cfg == -4 || tile_cfg == -5;
  const int wino_m = -tile_cfg;      // the Winograd form (2, 4, 5)
  if (wino) {
    PR_REQUIRE(precision == 0 && KH == 3 && KW == 3 && stride == 1 && pad == 1 && !res_dev && Cin == Cin_real &&
                   Cin % 32 == 0,
               "pr_conv2d_nhwc: tile_cfg -2 / -4 / -5 (Winograd) is for fp32 3x3 / stride 1 / pad 1 without residual, Cin %% 32 == 0");
    const int wn = conv_winograd_tile(wino_m) + 2;
    std::vector<float> u((size_t)wn * wn * Cout * Cin);
    conv_winograd_pack_weights(w_host, nullptr, Cout, Cin, wino_m, u.data());
    PR_HIP(hipMalloc(&wd, u.size() * sizeof(float)));
    PR_HIP(hipMemcpy(wd, u.data(), u.size() * sizeof(float), hipMemcpyHostToDevice));
    PR_HIP(hipMalloc(&work, std::max<size_t>(conv_winograd_work_floats(p, wino_m), 4) * sizeof(float)));
  } else if (precision == 1) {
    std::vector<unsigned short> packed((size_t)Cout * conv_kpad_bf16(p.K()));
    conv_pack_weights_bf16(w_host, nullptr, Cout, Cin_real, Cin, KH, KW, packed.data());
    PR_HIP(hipMalloc(&wd, packed.size() * 2));
    PR_HIP(hipMemcpy(wd, packed.data(), packed.size() * 2, hipMemcpyHostToDevice));
  } else {
    std::vector<float> packed((size_t)Cout * p.Kpad());
    conv_pack_weights(w_host, nullptr, Cout, Cin_real, Cin, KH, KW, packed.data());
    PR_HIP(hipMalloc(&wd, packed.size() * sizeof(float)));
    PR_HIP(hipMemcpy(wd, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  if (bias_host) {
    PR_HIP(hipMalloc(&bd, Cout * sizeof(float)));
    PR_HIP(hipMemcpy(bd, bias_host, Cout * sizeof(float), hipMemcpyHostToDevice));
  }
  p.x = (const float*)x_dev; p.w = wd; p.bias = bd; p.res = (const float*)res_dev; p.y = (float*)y_dev;
  int cfg = tile_cfg >= 0 ? tile_cfg : conv_pick_tile_cfg(p);
  if (tile_cfg > 200 && tile_cfg <= 208) {      // 64x64 tile with the K-steps of every tile dealt to tile_cfg - 200 workgroups
    PR_REQUIRE(precision == 0, "pr_conv2d_nhwc: split-K is fp32 only");
    cfg = 8;
    p.splitk = tile_cfg - 200;
    const size_t tiles = (size_t)ceil_div(p.M(), 64) * (Cout / 64);
    PR_HIP(hipMalloc(&sc.slab, tiles * p.splitk * 4096 * sizeof(float)));
    PR_HIP(hipMalloc(&sc.tickets, tiles * sizeof(int)));
    p.split_slab = sc.slab;
    p.split_tickets = sc.tickets;
  }
  auto go = [&]() -> int { return wino ? conv_winograd_launch(p, wd, work, wino_m, s) : conv_launch(p, cfg, s); };
  int st = go();
  if (st == PR_OK && repeats > 0 && ms_out) {
    hipEvent_t &e0 = sc.e0, &e1 = sc.e1;
    PR_HIP(hipEventCreate(&e0));
    PR_HIP(hipEventCreate(&e1));
    PR_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < repeats && st == PR_OK; ++i) st = go();
    PR_HIP(hipEventRecord(e1, s));
    PR_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    PR_HIP(hipEventElapsedTime(&ms, e0, e1));
    *ms_out = ms / repeats;
  }
  hipError_t e = hipStreamSynchronize(s);  // the scratch buffers must outlive the launches
  if (st != PR_OK) return st;
  PR_HIP(e);
  return PR_OK;
}

int pr_conv1x1_dual_nhwc(int device, const void* x1_dev, const float* w1_host, const void* x2_dev, const float* w2_host,
                         const float* bias_host, void* y_dev, int B, int Ho, int Wo, int Cin1, int H2, int W2, int Cin2,
                         int stride2, int Cout, int relu, int tile_cfg, int precision, void* stream) {
  using namespace pr;
  PR_REQUIRE(x1_dev && w1_host && x2_dev && w2_host && y_dev, "pr_conv1x1_dual_nhwc: null argument");
  PR_REQUIRE(precision == 0 || precision == 1, "pr_conv1x1_dual_nhwc: precision %d unknown", precision);
  const int kq = precision == 1 ? 64 : kConvBK;
  PR_REQUIRE(Cin1 > 0 && Cin2 > 0 && Cin1 % kq == 0 && Cin2 % kq == 0 && Cout % 64 == 0 && stride2 > 0,
             "pr_conv1x1_dual_nhwc: channels must be multiples of %d (Cout of 64)", kq);
  DeviceGuard g(device);
  hipStream_t s = (hipStream_t)stream;
  PR_TRY(refuse_if_capturing(s, "stand-alone test entry"));   // allocates and synchronises: never inside a capture
  ConvProblem p;
  p.B = B; p.H = p.Ho = Ho; p.W = p.Wo = Wo; p.Cin = Cin1; p.Cout = Cout; p.KH = p.KW = 1; p.stride = 1; p.pad = 0;
  p.relu = relu; p.precision = precision;
  p.tune = conv_tuning_from_env();
  p.H2 = H2; p.W2 = W2; p.Cin2 = Cin2; p.stride2 = stride2;
  struct Scratch {
    float *wd = nullptr, *bd = nullptr;
    ~Scratch() {
      if (wd) (void)hipFree(wd);
      if (bd) (void)hipFree(bd);
    }
  } sc;
  const size_t K = (size_t)Cin1 + Cin2;
  if (precision == 1) {
    std::vector<unsigned short> a((size_t)Cout * Cin1), b((size_t)Cout * Cin2), packed((size_t)Cout * K);
    conv_pack_weights_bf16(w1_host, nullptr, Cout, Cin1, Cin1, 1, 1, a.data());
    conv_pack_weights_bf16(w2_host, nullptr, Cout, Cin2, Cin2, 1, 1, b.data());
    for (int o = 0; o < Cout; ++o) {
      memcpy(&packed[o * K], &a[(size_t)o * Cin1], (size_t)Cin1 * 2);
      memcpy(&packed[o * K + Cin1], &b[(size_t)o * Cin2], (size_t)Cin2 * 2);
    }
    PR_HIP(hipMalloc(&sc.wd, packed.size() * 2));
    PR_HIP(hipMemcpy(sc.wd, packed.data(), packed.size() * 2, hipMemcpyHostToDevice));
  } else {
    std::vector<float> packed((size_t)Cout * K);
    for (int o = 0; o < Cout; ++o) {
      memcpy(&packed[o * K], w1_host + (size_t)o * Cin1, (size_t)Cin1 * 4);
      memcpy(&packed[o * K + Cin1], w2_host + (size_t)o * Cin2, (size_t)Cin2 * 4);
    }
    PR_HIP(hipMalloc(&sc.wd, packed.size() * 4));
    PR_HIP(hipMemcpy(sc.wd, packed.data(), packed.size() * 4, hipMemcpyHostToDevice));
  }
  if (bias_host) {
    PR_HIP(hipMalloc(&sc.bd, Cout * sizeof(float)));
    PR_HIP(hipMemcpy(sc.bd, bias_host, Cout * sizeof(float), hipMemcpyHostToDevice));
  }
  p.x = (const float*)x1_dev; p.x2 = (const float*)x2_dev; p.w = sc.wd; p.bias = sc.bd; p.res = nullptr; p.y = (float*)y_dev;
  const int cfg = tile_cfg >= 0 ? tile_cfg : conv_pick_tile_cfg(p);
  const int st = conv_launch(p, cfg, s);
  const hipError_t e = hipStreamSynchronize(s);
  if (st != PR_OK) return st;
  PR_HIP(e);
  return PR_OK;
}

int pr_conv3x3_conv1x1_nhwc(int device, const void* x_dev, const float* w2_host, const float* b2_host,
                            const float* w3_host, const float* b3_host, const void* res_dev, void* y_dev, int B, int H,
                            int W, int Cin, int N3, int relu3, int precision, void* stream) {
  using namespace pr;
  PR_REQUIRE(x_dev && w2_host && b2_host && w3_host && b3_host && y_dev, "pr_conv3x3_conv1x1_nhwc: null argument");
  PR_REQUIRE(precision == 0 || precision == 1, "pr_conv3x3_conv1x1_nhwc: precision %d unknown", precision);
  PR_REQUIRE(Cin >= (precision ? 64 : 32) && (Cin & (Cin - 1)) == 0 && N3 > 0 && N3 % 64 == 0,
             "pr_conv3x3_conv1x1_nhwc: Cin must be a power of two >= 32 (bf16: 64), N3 a multiple of 64");
  DeviceGuard g(device);
  hipStream_t s = (hipStream_t)stream;
  PR_TRY(refuse_if_capturing(s, "stand-alone test entry"));   // allocates and synchronises: never inside a capture
  struct Scratch {
    float* p[4] = {nullptr, nullptr, nullptr, nullptr};
    ~Scratch() {
      for (float* q : p)
        if (q) (void)hipFree(q);
    }
  } sc;
  ConvProblem p;
  p.B = B; p.H = p.Ho = H; p.W = p.Wo = W; p.Cin = Cin; p.Cout = 64; p.KH = p.KW = 3; p.stride = 1; p.pad = 1; p.relu = 1;
  p.precision = precision;
  p.tune = conv_tuning_from_env();
  // weights in the handle's precision: floats, or bf16 bit patterns carried in a float vector
  std::vector<float> w2p, w3p;
  if (precision == 1) {
    std::vector<unsigned short> a((size_t)64 * conv_kpad_bf16(p.K())), b((size_t)N3 * 64);
    conv_pack_weights_bf16(w2_host, nullptr, 64, Cin, Cin, 3, 3, a.data());
    conv_pack_weights_bf16(w3_host, nullptr, N3, 64, 64, 1, 1, b.data());
    w2p.resize((a.size() + 1) / 2);
    w3p.resize((b.size() + 1) / 2);
    memcpy(w2p.data(), a.data(), a.size() * 2);
    memcpy(w3p.data(), b.data(), b.size() * 2);
  } else {
    w2p.resize((size_t)64 * p.Kpad());
    conv_pack_weights(w2_host, nullptr, 64, Cin, Cin, 3, 3, w2p.data());
    w3p.assign(w3_host, w3_host + (size_t)N3 * 64);
  }
  const size_t sizes[4] = {w2p.size(), 64, w3p.size(), (size_t)N3};
  const float* src[4] = {w2p.data(), b2_host, w3p.data(), b3_host};
  for (int i = 0; i < 4; ++i) {
    PR_HIP(hipMalloc(&sc.p[i], sizes[i] * sizeof(float)));
    PR_HIP(hipMemcpy(sc.p[i], src[i], sizes[i] * sizeof(float), hipMemcpyHostToDevice));
  }
  p.x = (const float*)x_dev; p.w = sc.p[0]; p.bias = sc.p[1]; p.w3 = sc.p[2]; p.bias3 = sc.p[3];
  p.res3 = (const float*)res_dev; p.y3 = (float*)y_dev;
  p.N3 = N3; p.relu3 = relu3;
  const int st = conv_launch(p, 8, s);
  const hipError_t e = hipStreamSynchronize(s);
  if (st != PR_OK) return st;
  PR_HIP(e);
  return PR_OK;
}

int pr_bottleneck_nhwc(int device, const void* x_dev, const float* w1_host, const float* b1_host, const float* w2_host,
                       const float* b2_host, const float* w3_host, const float* b3_host, const float* wd_host,
                       const float* bd_host, void* y_dev, int B, int H, int W, int repeats, float* ms_out, void* stream) {
  using namespace pr;
  PR_REQUIRE(x_dev && w1_host && b1_host && w2_host && b2_host && w3_host && b3_host && y_dev, "pr_bottleneck_nhwc: null argument");
  PR_REQUIRE(!wd_host == !bd_host, "pr_bottleneck_nhwc: the downsample branch needs both its weight and its bias");
  PR_REQUIRE(B >= 0 && H > 0 && W > 0, "pr_bottleneck_nhwc: bad geometry");
  const bool first = wd_host != nullptr;
  const int cin = first ? 64 : 256, k3 = first ? 128 : 64;
  DeviceGuard g(device);
  hipStream_t s = (hipStream_t)stream;
  PR_TRY(refuse_if_capturing(s, "stand-alone test entry"));   // allocates and synchronises: never inside a capture
  struct Scratch {
    void* p[6] = {};
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ~Scratch() {
      for (void* q : p)
        if (q) (void)hipFree(q);
      if (e0) (void)hipEventDestroy(e0);
      if (e1) (void)hipEventDestroy(e1);
    }
  } sc;
  // bf16 weights in the encoder's packed layout (k = tap * Cin + c; a first block's conv3 and downsample matrices side
  // by side), rows permuted for the transposed MFMAs
  std::vector<unsigned short> a1((size_t)64 * cin), a2((size_t)64 * 576), a3((size_t)256 * 64), ad((size_t)256 * 64),
      a3d((size_t)256 * k3), p1(a1.size()), p2(a2.size()), p3(a3d.size());
  conv_pack_weights_bf16(w1_host, nullptr, 64, cin, cin, 1, 1, a1.data());
  conv_pack_weights_bf16(w2_host, nullptr, 64, 64, 64, 3, 3, a2.data());
  conv_pack_weights_bf16(w3_host, nullptr, 256, 64, 64, 1, 1, a3.data());
  if (first) conv_pack_weights_bf16(wd_host, nullptr, 256, 64, 64, 1, 1, ad.data());
  for (int o = 0; o < 256; ++o) {
    memcpy(&a3d[(size_t)o * k3], &a3[(size_t)o * 64], 128);
    if (first) memcpy(&a3d[(size_t)o * k3 + 64], &ad[(size_t)o * 64], 128);
  }
  std::vector<float> b3(b3_host, b3_host + 256);
  if (first)
    for (int o = 0; o < 256; ++o) b3[o] = (float)((double)b3_host[o] + (double)bd_host[o]);
  bottleneck_pack_rows_bf16(a1.data(), 64, cin, p1.data());
  bottleneck_pack_rows_bf16(a2.data(), 64, 576, p2.data());
  bottleneck_pack_rows_bf16(a3d.data(), 256, k3, p3.data());
  const void* src[6] = {p1.data(), p2.data(), p3.data(), b1_host, b2_host, b3.data()};
  const size_t bytes[6] = {p1.size() * 2, p2.size() * 2, p3.size() * 2, 64 * 4, 64 * 4, 256 * 4};
  for (int i = 0; i < 6; ++i) {
    PR_HIP(hipMalloc(&sc.p[i], bytes[i]));
    PR_HIP(hipMemcpy(sc.p[i], src[i], bytes[i], hipMemcpyHostToDevice));
  }
  BottleneckProblem p;
  p.x = x_dev; p.y = y_dev; p.w1 = sc.p[0]; p.w2 = sc.p[1]; p.w3 = sc.p[2];
  p.b1 = (const float*)sc.p[3]; p.b2 = (const float*)sc.p[4]; p.b3 = (const float*)sc.p[5];
  p.B = B; p.H = H; p.W = W; p.first = first;
  int st = bottleneck_bf16_launch(p, s);
  if (st == PR_OK && repeats > 0 && ms_out) {
    PR_HIP(hipEventCreate(&sc.e0));
    PR_HIP(hipEventCreate(&sc.e1));
    PR_HIP(hipEventRecord(sc.e0, s));
    for (int i = 0; i < repeats && st == PR_OK; ++i) st = bottleneck_bf16_launch(p, s);
    PR_HIP(hipEventRecord(sc.e1, s));
    PR_HIP(hipEventSynchronize(sc.e1));
    float ms = 0.f;
    PR_HIP(hipEventElapsedTime(&ms, sc.e0, sc.e1));
    *ms_out = ms / repeats;
  }
  const hipError_t e = hipStreamSynchronize(s);
  if (st != PR_OK) return st;
  PR_HIP(e);
  return PR_OK;
}

int pr_bottleneck128_nhwc(int device, const void* x_dev, const float* w1_host, const float* b1_host, const float* w2_host,
                          const float* b2_host, const float* w3_host, const float* b3_host, void* y_dev, int B, int H, int W,
                          int repeats, float* ms_out, void* stream) {
  using namespace pr;
  PR_REQUIRE(x_dev && w1_host && b1_host && w2_host && b2_host && w3_host && b3_host && y_dev, "pr_bottleneck128_nhwc: null argument");
  PR_REQUIRE(B >= 0 && H > 0 && W > 0, "pr_bottleneck128_nhwc: bad geometry");
  DeviceGuard g(device);
  hipStream_t s = (hipStream_t)stream;
  PR_TRY(refuse_if_capturing(s, "stand-alone test entry"));   // allocates and synchronises: never inside a capture
  struct Scratch {
    void* p[6] = {};
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ~Scratch() {
      for (void* q : p)
        if (q) (void)hipFree(q);
      if (e0) (void)hipEventDestroy(e0);
      if (e1) (void)hipEventDestroy(e1);
    }
  } sc;
  // bf16 weights in the encoder's packed layout (conv2's K slice-major), rows permuted for the transposed MFMAs
  std::vector<unsigned short> a1((size_t)128 * 512), a2((size_t)128 * 1152), a3((size_t)512 * 128), p1(a1.size()), p2(a2.size()),
      p3(a3.size());
  conv_pack_weights_bf16(w1_host, nullptr, 128, 512, 512, 1, 1, a1.data());
  conv_pack_weights_bf16(w2_host, nullptr, 128, 128, 128, 3, 3, a2.data());
  conv_pack_weights_bf16(w3_host, nullptr, 512, 128, 128, 1, 1, a3.data());
  bottleneck_pack_rows_bf16(a1.data(), 128, 512, p1.data());
  bottleneck_pack_rows_bf16(a2.data(), 128, 1152, p2.data());
  bottleneck_pack_rows_bf16(a3.data(), 512, 128, p3.data());
  const void* src[6] = {p1.data(), p2.data(), p3.data(), b1_host, b2_host, b3_host};
  const size_t bytes[6] = {p1.size() * 2, p2.size() * 2, p3.size() * 2, 128 * 4, 128 * 4, 512 * 4};
  for (int i = 0; i < 6; ++i) {
    PR_HIP(hipMalloc(&sc.p[i], bytes[i]));
    PR_HIP(hipMemcpy(sc.p[i], src[i], bytes[i], hipMemcpyHostToDevice));
  }
  BottleneckProblem p;
  p.x = x_dev; p.y = y_dev; p.w1 = sc.p[0]; p.w2 = sc.p[1]; p.w3 = sc.p[2];
  p.b1 = (const float*)sc.p[3]; p.b2 = (const float*)sc.p[4]; p.b3 = (const float*)sc.p[5];
  p.B = B; p.H = H; p.W = W; p.planes = 128; p.first = false;
  int st = bottleneck_bf16_launch(p, s);
  if (st == PR_OK && repeats > 0 && ms_out) {
    PR_HIP(hipEventCreate(&sc.e0));
    PR_HIP(hipEventCreate(&sc.e1));
    PR_HIP(hipEventRecord(sc.e0, s));
    for (int i = 0; i < repeats && st == PR_OK; ++i) st = bottleneck_bf16_launch(p, s);
    PR_HIP(hipEventRecord(sc.e1, s));
    PR_HIP(hipEventSynchronize(sc.e1));
    float ms = 0.f;
    PR_HIP(hipEventElapsedTime(&ms, sc.e0, sc.e1));
    *ms_out = ms / repeats;
  }
  const hipError_t e = hipStreamSynchronize(s);
  if (st != PR_OK) return st;
  PR_HIP(e);
  return PR_OK;
}

int pr_bottleneck256_nhwc(int device, const void* x_dev, const float* w1_host, const float* b1_host, const float* w2_host,
                          const float* b2_host, const float* w3_host, const float* b3_host, void* y_dev, int B, int H, int W,
                          int repeats, float* ms_out, void* stream) {
  using namespace pr;
  PR_REQUIRE(x_dev && w1_host && b1_host && w2_host && b2_host && w3_host && b3_host && y_dev, "pr_bottleneck256_nhwc: null argument");
  PR_REQUIRE(B >= 0 && H > 0 && W > 0, "pr_bottleneck256_nhwc: bad geometry");
  DeviceGuard g(device);
  hipStream_t s = (hipStream_t)stream;
  PR_TRY(refuse_if_capturing(s, "stand-alone test entry"));   // allocates and synchronises: never inside a capture
  struct Scratch {
    void* p[6] = {};
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ~Scratch() {
      for (void* q : p)
        if (q) (void)hipFree(q);
      if (e0) (void)hipEventDestroy(e0);
      if (e1) (void)hipEventDestroy(e1);
    }
  } sc;
  // bf16 weights in the encoder's packed layout (conv2's K slice-major), rows permuted for the transposed MFMAs
  // (conv2 and conv3 further into MFMA fragment order: bottleneck256_bf16.hip)
  std::vector<unsigned short> a1((size_t)256 * 1024), a2((size_t)256 * 2304), a3((size_t)1024 * 256), p1(a1.size()), p2(a2.size()),
      p3(a3.size()), rows(a2.size());
  conv_pack_weights_bf16(w1_host, nullptr, 256, 1024, 1024, 1, 1, a1.data());
  conv_pack_weights_bf16(w2_host, nullptr, 256, 256, 256, 3, 3, a2.data());
  conv_pack_weights_bf16(w3_host, nullptr, 1024, 256, 256, 1, 1, a3.data());
  bottleneck_pack_rows_bf16(a1.data(), 256, 1024, p1.data());
  bottleneck_pack_rows_bf16(a2.data(), 256, 2304, rows.data());
  bottleneck256_pack_w2_frags_bf16(rows.data(), p2.data());
  bottleneck_pack_rows_bf16(a3.data(), 1024, 256, rows.data());
  bottleneck256_pack_w3_frags_bf16(rows.data(), p3.data());
  const void* src[6] = {p1.data(), p2.data(), p3.data(), b1_host, b2_host, b3_host};
  const size_t bytes[6] = {p1.size() * 2, p2.size() * 2, p3.size() * 2, 256 * 4, 256 * 4, 1024 * 4};
  for (int i = 0; i < 6; ++i) {
    PR_HIP(hipMalloc(&sc.p[i], bytes[i]));
    PR_HIP(hipMemcpy(sc.p[i], src[i], bytes[i], hipMemcpyHostToDevice));
  }
  BottleneckProblem p;
  p.x = x_dev; p.y = y_dev; p.w1 = sc.p[0]; p.w2 = sc.p[1]; p.w3 = sc.p[2];
  p.b1 = (const float*)sc.p[3]; p.b2 = (const float*)sc.p[4]; p.b3 = (const float*)sc.p[5];
  p.B = B; p.H = H; p.W = W; p.planes = 256; p.first = false;
  int st = bottleneck_bf16_launch(p, s);
  if (st == PR_OK && repeats > 0 && ms_out) {
    PR_HIP(hipEventCreate(&sc.e0));
    PR_HIP(hipEventCreate(&sc.e1));
    PR_HIP(hipEventRecord(sc.e0, s));
    for (int i = 0; i < repeats && st == PR_OK; ++i) st = bottleneck_bf16_launch(p, s);
    PR_HIP(hipEventRecord(sc.e1, s));
    PR_HIP(hipEventSynchronize(sc.e1));
    float ms = 0.f;
    PR_HIP(hipEventElapsedTime(&ms, sc.e0, sc.e1));
    *ms_out = ms / repeats;
  }
  const hipError_t e = hipStreamSynchronize(s);
  if (st != PR_OK) return st;
  PR_HIP(e);
  return PR_OK;
}

int pr_stem_pool_nhwc(int device, const void* x_dev, const float* w_host, const float* bias_host, void* y_dev, int B, int H,
                      int repeats, float* ms_out, void* stream) {
  using namespace pr;
  PR_REQUIRE(x_dev && w_host && bias_host && y_dev, "pr_stem_pool_nhwc: null argument");
  DeviceGuard g(device);
  hipStream_t s = (hipStream_t)stream;
  PR_TRY(refuse_if_capturing(s, "stand-alone test entry"));   // allocates and synchronises: never inside a capture
  struct Scratch {
    void* p[2] = {};
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ~Scratch() {
      for (void* q : p)
        if (q) (void)hipFree(q);
      if (e0) (void)hipEventDestroy(e0);
      if (e1) (void)hipEventDestroy(e1);
    }
  } sc;
  std::vector<unsigned short> packed((size_t)64 * 256);
  conv_pack_weights_bf16(w_host, nullptr, 64, 16, 16, 4, 4, packed.data());      // k = (th * 4 + tw) * 16 + c
  PR_HIP(hipMalloc(&sc.p[0], packed.size() * 2));
  PR_HIP(hipMemcpy(sc.p[0], packed.data(), packed.size() * 2, hipMemcpyHostToDevice));
  PR_HIP(hipMalloc(&sc.p[1], 64 * 4));
  PR_HIP(hipMemcpy(sc.p[1], bias_host, 64 * 4, hipMemcpyHostToDevice));
  int st = stem_pool_bf16_launch(x_dev, sc.p[0], (const float*)sc.p[1], y_dev, B, H, s);
  if (st == PR_OK && repeats > 0 && ms_out) {
    PR_HIP(hipEventCreate(&sc.e0));
    PR_HIP(hipEventCreate(&sc.e1));
    PR_HIP(hipEventRecord(sc.e0, s));
    for (int i = 0; i < repeats && st == PR_OK; ++i) st = stem_pool_bf16_launch(x_dev, sc.p[0], (const float*)sc.p[1], y_dev, B, H, s);
    PR_HIP(hipEventRecord(sc.e1, s));
    PR_HIP(hipEventSynchronize(sc.e1));
    float ms = 0.f;
    PR_HIP(hipEventElapsedTime(&ms, sc.e0, sc.e1));
    *ms_out = ms / repeats;
  }
  const hipError_t e = hipStreamSynchronize(s);
  if (st != PR_OK) return st;
  PR_HIP(e);
  return PR_OK;
}

int pr_stem_pool_f32_nhwc(int device, const float* x_dev, const float* w_host, const float* bias_host, float* y_dev, int B,
                          int repeats, float* ms_out, void* stream) {
  using namespace pr;
  PR_REQUIRE(x_dev && w_host && bias_host && y_dev, "pr_stem_pool_f32_nhwc: null argument");
  DeviceGuard g(device);
  hipStream_t s = (hipStream_t)stream;
  PR_TRY(refuse_if_capturing(s, "stand-alone test entry"));   // allocates and synchronises: never inside a capture
  struct Scratch {
    void* p[2] = {};
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ~Scratch() {
      for (void* q : p)
        if (q) (void)hipFree(q);
      if (e0) (void)hipEventDestroy(e0);
      if (e1) (void)hipEventDestroy(e1);
    }
  } sc;
  // the kernel shares MFMAs between the half-empty taps of the 7x7 kernel's zero row / column (stem_pool_f32.hip): weights
  // that are not a 7x7 kernel in the 8x8 window would be computed wrong, so they are refused
  for (int o = 0; o < 64; ++o)
    for (int c12 = 0; c12 < 12; ++c12)
      for (int t = 0; t < 4; ++t) {
        const int sub = c12 / 3;                                              // 2 di + dj
        const float top = w_host[((o * 12 + c12) * 4 + 0) * 4 + t], left = w_host[((o * 12 + c12) * 4 + t) * 4 + 0];
        PR_REQUIRE(((sub >> 1) != 0 || top == 0.f) && ((sub & 1) != 0 || left == 0.f),
                   "pr_stem_pool_f32_nhwc: the weights must be a 7x7 kernel in the 4x4 taps' 8x8 window (zero for tap row 0 / "
                   "sub-row 0 and for tap column 0 / sub-column 0); output channel %d, channel %d is not", o, c12);
      }
  std::vector<float> packed((size_t)64 * 192);
  conv_pack_weights(w_host, nullptr, 64, 12, 12, 4, 4, packed.data());      // k = (th * 4 + tw) * 12 + c
  PR_HIP(hipMalloc(&sc.p[0], packed.size() * 4));
  PR_HIP(hipMemcpy(sc.p[0], packed.data(), packed.size() * 4, hipMemcpyHostToDevice));
  PR_HIP(hipMalloc(&sc.p[1], 64 * 4));
  PR_HIP(hipMemcpy(sc.p[1], bias_host, 64 * 4, hipMemcpyHostToDevice));
  int st = stem_pool_f32_launch(x_dev, (const float*)sc.p[0], (const float*)sc.p[1], y_dev, B, s);
  if (st == PR_OK && repeats > 0 && ms_out) {
    PR_HIP(hipEventCreate(&sc.e0));
    PR_HIP(hipEventCreate(&sc.e1));
    PR_HIP(hipEventRecord(sc.e0, s));
    for (int i = 0; i < repeats && st == PR_OK; ++i) st = stem_pool_f32_launch(x_dev, (const float*)sc.p[0], (const float*)sc.p[1], y_dev, B, s);
    PR_HIP(hipEventRecord(sc.e1, s));
    PR_HIP(hipEventSynchronize(sc.e1));
    float ms = 0.f;
    PR_HIP(hipEventElapsedTime(&ms, sc.e0, sc.e1));
    *ms_out = ms / repeats;
  }
  const hipError_t e = hipStreamSynchronize(s);
  if (st != PR_OK) return st;
  PR_HIP(e);
  return PR_OK;
}

int pr_frames_forward(pr_hmr_t* hmr, pr_smpl_t* smpl, const float* x_dev, int B,
                      const pr_reba_info* reba_info, const pr_rula_info* rula_info,
                      const pr_frames_out* out, void* stream) {
  PR_REQUIRE(B >= 0, "pr_frames_forward: negative batch");
  if (B == 0) return PR_OK;  // an empty batch is legal (empty tensors have null data pointers)
  PR_REQUIRE(hmr && smpl && x_dev && out, "pr_frames_forward: null argument");
  PR_REQUIRE(out->rotmat && out->axis_angle && out->euler_deg && out->joint_cam,
             "pr_frames_forward: rotmat, axis_angle, euler_deg and joint_cam are required");
  PR_REQUIRE((!out->reba || reba_info) && (!out->rula || rula_info), "pr_frames_forward: score output without info");
  // base.py:220        encoder + regressor
  PR_TRY(pr_hmr_forward(hmr, x_dev, B, out->rotmat, out->betas, out->cam, nullptr, nullptr, stream));
  // base.py:225-229    rotmat -> axis-angle -> Euler degrees (per frame, per joint)
  PR_TRY(pr_pose_to_euler(out->rotmat, B, out->axis_angle, out->euler_deg, out->status, stream));
  // base.py:239        joint_cam (mutates axis_angle's root rows, as the reference does)
  PR_TRY(pr_smpl_joint_cam(smpl, out->axis_angle, B, out->joint_cam, out->verts, stream));
  // base.py:151,168    scorers
  if (out->reba) PR_TRY(pr_reba(out->euler_deg, B, reba_info, out->reba, stream));
  if (out->rula) PR_TRY(pr_rula(out->euler_deg, B, rula_info, out->rula, stream));
  return PR_OK;
}

}  // extern "C"
