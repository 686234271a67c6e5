// pr_hmr: SPIN HMR (ResNet-50 encoder + 3-iteration regressor + rot6d->rotmat) on gfx950.
// Replaces models.hmr / spin_model(batch)  (lib/core/base.py:81-84, :220).
//
// The host side that needs no device -- parsing the canonical weight blob (include/poserisk_hip.h), folding eval-mode
// BatchNorm into the convolutions in double, packing weights for every kernel family, the 53-conv execution plan over NHWC
// activation buffers, the regressor's fc1 split into its constant part (pooled features, computed once) and its state part
// (157 inputs, recomputed per iteration) -- is host_plan.cc (plain C++, also built under ASan + UBSan by tests/native).
// Here: device memory, workspaces, streams, the launch sequence.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "conv_igemm.h"
#include "frame_kernels.h"

struct pr_hmr : pr::HmrPlan {
  // settings, the 53-convolution plan, packed-weight pointers and regressor workspaces: pr::HmrPlan (host_plan.h, built by
  // hmr_plan_build -- device-free code that tests/native runs under ASan + UBSan); what follows is the device state
  int device = 0;
  pr::ConvTuning tune;          // tile-choice / quarter-tile switches of the conv launches (read once, at create)
  float* split_slab[8] = {};      // per sub-batch chunk (kMaxChunks)
  int* split_tickets[8] = {};
  std::vector<float*> dev_allocs;
  // activation buffers per frame chunk: 0 = NHWC4 input, 1..5 = rotating feature maps.
  // The batch is cut into n_chunks contiguous sub-batches that run the encoder on their own
  // HIP streams (frames are independent).  Measured on MI355X at B=64 this lock-step form is SLOWER
  // than one stream (sub-batch kernels are smaller and all streams run the same layer at once), so the
  // default is 1; what does pay is whole batches in flight on different streams, which the caller
  // drives (pipeline.FramePipeline lanes).  Kept because it is bit-identical and lets a batch exceed
  // one sub-batch's workspace.
  static constexpr int kMaxChunks = pr::kHmrMaxChunks;
  int n_chunks = 1;
  int chunk_cap = 0;
  float* act[kMaxChunks][6] = {};
  float* wino_work[kMaxChunks] = {};  // V and M of the Winograd layers (conv_winograd.hip)
  hipStream_t streams[kMaxChunks] = {};
  hipEvent_t ev_fork = nullptr;
  hipEvent_t ev_join[kMaxChunks] = {};
  std::vector<float*> act_allocs;
  // profiling
  bool profile = false;
  std::vector<float> prof_ms;
  std::vector<int> prof_n;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
  std::vector<int> pending_layer;
};

namespace pr {
namespace {

// hmr_plan_build's constants land in device memory, owned by the handle
struct DeviceSink : PlanSink {
  pr_hmr* h;
  explicit DeviceSink(pr_hmr* handle) : h(handle) {}
  int upload(const void* host, size_t bytes, float** out) override {
    float* d = nullptr;
    PR_HIP(hipMalloc(&d, std::max<size_t>(bytes, 16)));
    h->dev_allocs.push_back(d);
    PR_HIP(hipMemcpy(d, host, bytes, hipMemcpyHostToDevice));
    *out = d;
    return PR_OK;
  }
  int zeros(size_t bytes, float** out) override {
    float* d = nullptr;
    PR_HIP(hipMalloc(&d, std::max<size_t>(bytes, 16)));
    h->dev_allocs.push_back(d);
    PR_HIP(hipMemset(d, 0, std::max<size_t>(bytes, 16)));
    *out = d;
    return PR_OK;
  }
};

// (Re)allocate the activation buffers for n sub-batches and create their streams / events.
int set_chunks(pr_hmr* h, int n) {
  PR_REQUIRE(n >= 1 && n <= pr_hmr::kMaxChunks, "hmr: stream count %d out of range 1..%d", n, pr_hmr::kMaxChunks);
  PR_HIP(hipDeviceSynchronize());
  for (float* p : h->act_allocs) (void)hipFree(p);
  h->act_allocs.clear();
  h->n_chunks = n;
  // a sub-batch is also the unit of one conv launch, whose tensors must stay under 2 GiB (the DMA kernel's
  // out-of-range sentinel): 512 frames x 56x56x256 fp32 = 1.6 GB
  h->chunk_cap = hmr_chunk_cap(h->max_batch, n);
  const HmrChunkSizes z = hmr_chunk_sizes(*h, h->chunk_cap);   // element counts per sub-batch (host_plan.cc)
  for (int c = 0; c < n; ++c) {
    for (int i = 0; i <= 5; ++i) {
      float* d = nullptr;
      PR_HIP(hipMalloc(&d, (i == 0 ? z.act0_floats : z.act_floats) * sizeof(float)));
      h->act_allocs.push_back(d);
      h->act[c][i] = d;
    }
    if (z.wino_floats) {
      float* d = nullptr;
      PR_HIP(hipMalloc(&d, z.wino_floats * sizeof(float)));
      h->act_allocs.push_back(d);
      h->wino_work[c] = d;
    }
    if (z.slab_floats) {
      float* d = nullptr;
      PR_HIP(hipMalloc(&d, z.slab_floats * sizeof(float)));
      h->act_allocs.push_back(d);
      h->split_slab[c] = d;
      PR_HIP(hipMalloc(&d, z.tickets * sizeof(int)));
      h->act_allocs.push_back(d);
      h->split_tickets[c] = reinterpret_cast<int*>(d);
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
  if (!h->fc_tiles) return launch_fc_rows16(x, fc.w, use_bias ? fc.bias : nullptr, res, y, B, fc.N, fc.K, s, h->fc_shape);
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
  auto fused_pays = [&](int b) { return hmr_fused3_pays(b, h->cus); };
  size_t skip_until[pr_hmr::kMaxChunks] = {};
  for (size_t ci = 0; ci < h->convs.size(); ++ci) {
    ConvSpec& c = h->convs[ci];
    const int li = c.layer;
    const pr::HmrPlan::FusedBlock* alt = nullptr;
    for (const pr::HmrPlan::FusedBlock& fb : h->fused3)
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
  PR_REQUIRE(hmr_conv_form_valid(conv_form),
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
  PR_TRY(refuse_under_declared_capture("pr_hmr_create"));
  std::unique_ptr<pr_hmr> h(new pr_hmr);
  h->device = device;
  // the conv form and every POSERISK_* A/B switch: read here, once per handle (nothing is latched per process)
  hmr_plan_configure(h.get(), precision, conv_form, max_batch);
  h->tune = conv_tuning_from_env();
  (void)hipDeviceGetAttribute(&h->cus, hipDeviceAttributeMultiprocessorCount, h->device);
  DeviceSink sink(h.get());
  int st = hmr_plan_build(h.get(), weights_host, n_floats, sink);
  h->prof_ms.assign(kNumConv, 0.f);
  h->prof_n.assign(kNumConv, 0);
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
  PR_TRY(pr::refuse_under_declared_capture("pr_hmr_destroy"));   // the handle stays valid: destroy it after the capture
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
  PR_TRY(pr::refuse_under_declared_capture("pr_hmr_set_streams"));
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
  // Profiling runs serially on the caller's stream so that each conv's event bracket is its own time.  The split is
  // hmr_split_batch's (host_plan.cc), which pr_hmr_plan_counts walks too.
  int sizes[4096];
  bool concurrent = false;
  const int nsub = hmr_split_batch(B, h->chunk_cap, h->n_chunks, h->profile != 0, sizes, 4096, &concurrent);
  const size_t frame = (size_t)3 * kImg * kImg;
  if (!concurrent) {
    // one sub-batch at a time on the caller's stream (more than one pass if B exceeds a chunk's buffers)
    for (int i = 0, b0 = 0; i < nsub; b0 += sizes[i], ++i) {
      ChunkRun r{0, x_dev + b0 * frame, sizes[i], h->xf + (size_t)b0 * 2048, s};
      PR_TRY(encode_chunks(h, &r, 1));
    }
  } else {
    const int nch = nsub;
    ChunkRun runs[pr_hmr::kMaxChunks];
    PR_HIP(hipEventRecord(h->ev_fork, s));
    for (int c = 0, b0 = 0; c < nch; b0 += sizes[c], ++c) {
      runs[c] = ChunkRun{c, x_dev + b0 * frame, sizes[c], h->xf + (size_t)b0 * 2048, h->streams[c]};
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
  hmr_plan_counts(*h, B, h->chunk_cap, h->n_chunks, h->profile != 0, conv_launches, winograd_layers);
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
