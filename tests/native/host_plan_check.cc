// CPU-only check of the device-free half of pr_hmr_create (poserisk_release_amd/csrc/host_plan.cc), built by
// tests/test_host_plan_native.py with  g++ -fsanitize=address,undefined -fno-sanitize-recover=all.
// ("sanitizer build of the host C++"; the code under test only ever runs behind pr_hmr_create, which
// needs a GPU, so nothing else exercises its index arithmetic off-device.)
//
//   host_plan_check <blob.f32> <manifest.json> <dump.bin>
//
// Builds the full ResNet-50 + regressor plan from the canonical weight blob (include/poserisk_hip.h; the SPIN state dict
// lib/core/base.py:83-84 loads) for max_batch in {1, 7, 64, 230, 256, 460}, both precisions, every conv form and every
// plan-shaping A/B switch, with a PlanSink over exact-size heap blocks, and checks for each plan:
//   * the 53 layer indices are carried exactly once, the buffer rotation delivers to every launch the tensor the ResNet
//     topology says it reads (symbolic dataflow), kernel routing matches the shapes the kernels take;
//   * algorithmic multiply-adds per frame == 4 087 136 256, executed >= algorithmic for direct forms;
//   * launches per forward (pr_hmr_plan_counts) and workspace sizes for 1, 2 and 8 sub-batches (every launch < 2 GiB).
// For the default fp32 and bf16 plans at B = 64 it writes a manifest of every upload (bytes, FNV-1a) and the raw bytes of
// the first upload of each size and of the regressor's nine to <dump.bin>: the Python test recomputes BN folding, packing and the Winograd G-transform
// with numpy from the same state dict (poserisk_release_amd/weights.py order) and compares.
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "../../poserisk_release_amd/csrc/host_plan.h"

namespace pr {
static std::string g_err;
void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
}
}  // namespace pr

using namespace pr;

static int g_fail = 0;
#define CHECK(cond, ...)                                   \
  do {                                                     \
    if (!(cond)) {                                         \
      fprintf(stderr, "CHECK FAILED %s:%d: ", __FILE__, __LINE__); \
      fprintf(stderr, __VA_ARGS__);                        \
      fprintf(stderr, "\n");                               \
      ++g_fail;                                            \
    }                                                      \
  } while (0)

struct Upload {
  size_t bytes;
  uint64_t fnv;
  bool zeros;
  void* ptr;
};

struct HostSink : PlanSink {
  std::vector<Upload> ups;
  int upload(const void* host, size_t bytes, float** out) override {
    void* p = malloc(bytes ? bytes : 1);          // exact size: ASan sees any read past a packed array
    memcpy(p, host, bytes);
    uint64_t h = 1469598103934665603ull;
    const unsigned char* b = static_cast<const unsigned char*>(host);
    for (size_t i = 0; i < bytes; ++i) h = (h ^ b[i]) * 1099511628211ull;
    ups.push_back({bytes, h, false, p});
    *out = static_cast<float*>(p);
    return PR_OK;
  }
  int zeros(size_t bytes, float** out) override {
    void* p = calloc(1, bytes ? bytes : 1);
    ups.push_back({bytes, 0, true, p});
    *out = static_cast<float*>(p);
    return PR_OK;
  }
  ~HostSink() override {
    for (Upload& u : ups) free(u.ptr);
  }
};

struct Flags {       // what the plan is expected to look like, from the switches
  bool fuse_downsample = true, fuse_conv3 = true, fuse_bneck = true, fuse_bneck2 = true, fuse_bneck3 = true;
};

static const int kPlanes[4] = {64, 128, 256, 512}, kBlocks[4] = {3, 4, 6, 3};

// Symbolic dataflow over the activation buffers: every launch must read the tensor the topology says.
static void check_dataflow(const HmrPlan& pl, const char* tag) {
  std::string holds[6];
  holds[0] = "input";
  size_t ci = 0;
  const auto& cv = pl.convs;
  CHECK(!cv.empty() && cv[0].in_buf == 0 && cv[0].layer == 0, "%s: the stem is not the first launch", tag);
  holds[1] = "stem";
  holds[2] = "in(0,0)";       // the max-pooled map (the fused stem writes it directly)
  ci = 1;
  std::set<int> layers = {0};
  auto take_layer = [&](int l) {
    CHECK(l >= 0 && l < kNumConv, "%s: layer index %d out of range", tag, l);
    CHECK(layers.insert(l).second, "%s: layer index %d carried twice", tag, l);
  };
  auto buf_ok = [&](int b, bool input) { return b >= (input ? 0 : 1) && b <= 5; };
  int inpl = 64, H = 56;
  for (int L = 0; L < 4; ++L)
    for (int b = 0; b < kBlocks[L]; ++b) {
      char xin[32], xout[32];
      snprintf(xin, sizeof xin, "in(%d,%d)", L, b);
      if (b + 1 < kBlocks[L]) snprintf(xout, sizeof xout, "in(%d,%d)", L, b + 1);
      else snprintf(xout, sizeof xout, "in(%d,0)", L + 1);
      const int pln = kPlanes[L], stride = (b == 0 && L > 0) ? 2 : 1, Ho = H / stride;
      CHECK(ci < cv.size(), "%s: plan ends inside block %d.%d", tag, L, b);
      if (ci >= cv.size()) return;
      const ConvSpec& c0 = cv[ci];
      for (const HmrPlan::FusedBlock& fb : pl.fused3)
        if (fb.first == ci) {
          CHECK(ci + 2 < cv.size() && fb.blk.in_buf == c0.in_buf && fb.blk.out_buf == cv[ci + 2].out_buf && fb.blk.layer == cv[ci + 2].layer,
                "%s: fused layer3 block at launch %zu does not span its three launches", tag, ci);
          CHECK(fb.blk.bneck_planes == pln && fb.blk.H == H && fb.blk.w && fb.blk.w2b && fb.blk.w3 && fb.blk.bias && fb.blk.bias2b && fb.blk.bias3,
                "%s: fused layer3 block incomplete", tag);
        }
      if (c0.bneck_planes) {
        CHECK(c0.bneck_planes == pln && c0.bneck_first == (b == 0) && c0.H == H && c0.Cin == inpl && c0.Cout == 4 * pln, "%s: whole-block spec %d.%d has the wrong shape", tag, L, b);
        CHECK(buf_ok(c0.in_buf, true) && buf_ok(c0.out_buf, false) && c0.in_buf != c0.out_buf, "%s: whole-block buffers", tag);
        CHECK(holds[c0.in_buf] == xin, "%s: block %d.%d reads buffer %d holding '%s'", tag, L, b, c0.in_buf, holds[c0.in_buf].c_str());
        CHECK(c0.w && c0.w2b && c0.w3 && c0.bias && c0.bias2b && c0.bias3, "%s: whole-block weights missing", tag);
        take_layer(c0.layer);
        // its conv1 / conv2 (/ downsample) indices are carried by this launch: they must stay unclaimed by others
        holds[c0.out_buf] = xout;
        ++ci;
      } else {
        // conv1
        CHECK(c0.k == 1 && c0.stride == 1 && c0.Cin == inpl && c0.Cout == pln && c0.H == H && c0.relu, "%s: conv1 of %d.%d", tag, L, b);
        CHECK(buf_ok(c0.in_buf, true) && buf_ok(c0.out_buf, false) && c0.in_buf != c0.out_buf && c0.res_buf < 0, "%s: conv1 buffers", tag);
        CHECK(holds[c0.in_buf] == xin, "%s: conv1 of %d.%d reads '%s'", tag, L, b, holds[c0.in_buf].c_str());
        take_layer(c0.layer);
        holds[c0.out_buf] = "t1";
        ++ci;
        CHECK(ci < cv.size(), "%s: plan ends behind conv1 of %d.%d", tag, L, b);
        if (ci >= cv.size()) return;
        const ConvSpec& c1 = cv[ci];
        CHECK(c1.k == 3 && c1.stride == stride && c1.pad == 1 && c1.Cin == pln && c1.Cout == pln && c1.H == H && c1.relu, "%s: conv2 of %d.%d", tag, L, b);
        CHECK(holds[c1.in_buf] == "t1" && c1.in_buf != c1.out_buf, "%s: conv2 of %d.%d reads '%s'", tag, L, b, holds[c1.in_buf].c_str());
        if (c1.wino_m) CHECK(c1.u && stride == 1 && pln >= pl.wino_min_c && pl.precision == 0 && (c1.wino_m == 2 || c1.wino_m == 4), "%s: Winograd spec", tag);
        ++ci;
        if (c1.w3) {           // conv3 inside conv2's kernel
          CHECK(L == 0 && b > 0 && c1.N3 == 4 * pln && c1.bias3 && buf_ok(c1.out3_buf, false) && c1.out3_buf != c1.in_buf, "%s: fused conv3 of %d.%d", tag, L, b);
          CHECK(c1.res3_buf >= 0 && holds[c1.res3_buf] == xin && c1.res3_buf != c1.out3_buf, "%s: fused conv3's residual reads '%s'", tag, c1.res3_buf >= 0 ? holds[c1.res3_buf].c_str() : "-");
          take_layer(c1.layer);
          take_layer(c1.layer2);
          holds[c1.out3_buf] = xout;
        } else {
          take_layer(c1.layer);
          holds[c1.out_buf] = "t2";
          CHECK(ci < cv.size(), "%s: plan ends behind conv2 of %d.%d", tag, L, b);
          if (ci >= cv.size()) return;
          const ConvSpec* c2 = &cv[ci];
          if (b == 0 && c2->in2_buf < 0) {      // separate downsample launch first
            CHECK(c2->k == 1 && c2->stride == stride && c2->Cin == inpl && c2->Cout == 4 * pln && !c2->relu && c2->H == H, "%s: downsample of %d.%d", tag, L, b);
            CHECK(holds[c2->in_buf] == xin && c2->in_buf != c2->out_buf, "%s: downsample reads '%s'", tag, holds[c2->in_buf].c_str());
            take_layer(c2->layer);
            holds[c2->out_buf] = "ds";
            ++ci;
            CHECK(ci < cv.size(), "%s: plan ends behind the downsample of %d.%d", tag, L, b);
            if (ci >= cv.size()) return;
            c2 = &cv[ci];
            CHECK(c2->res_buf >= 0 && holds[c2->res_buf] == "ds", "%s: conv3 of %d.0 adds '%s'", tag, L, c2->res_buf >= 0 ? holds[c2->res_buf].c_str() : "-");
          } else if (b == 0) {
            CHECK(c2->res_buf < 0 && holds[c2->in2_buf] == xin && c2->Cin2 == inpl && c2->H2 == H && c2->stride2 == stride && c2->in2_buf != c2->out_buf,
                  "%s: dual-source conv3 of %d.0 reads '%s'", tag, L, holds[c2->in2_buf].c_str());
            take_layer(c2->layer2);
          } else {
            CHECK(c2->res_buf >= 0 && holds[c2->res_buf] == xin, "%s: conv3 of %d.%d adds '%s'", tag, L, b, c2->res_buf >= 0 ? holds[c2->res_buf].c_str() : "-");
          }
          CHECK(c2->k == 1 && c2->stride == 1 && c2->Cin == pln && c2->Cout == 4 * pln && c2->H == Ho && c2->relu, "%s: conv3 of %d.%d", tag, L, b);
          CHECK(holds[c2->in_buf] == "t2" && c2->in_buf != c2->out_buf && c2->res_buf != c2->out_buf, "%s: conv3 of %d.%d reads '%s'", tag, L, b, holds[c2->in_buf].c_str());
          take_layer(c2->layer);
          holds[c2->out_buf] = xout;
          ++ci;
        }
      }
      inpl = 4 * pln;
      H = Ho;
    }
  CHECK(ci == cv.size(), "%s: %zu launches left over", tag, cv.size() - ci);
  CHECK(holds[pl.final_buf] == "in(4,0)", "%s: the average pool reads buffer %d holding '%s'", tag, pl.final_buf, holds[pl.final_buf].c_str());
  // indices carried by nobody must be exactly those folded into whole-block launches (conv1, conv2, downsample of such blocks)
  int folded = 0;
  for (const ConvSpec& c : cv)
    if (c.bneck_planes) folded += c.bneck_first ? 3 : 2;
  CHECK((int)layers.size() + folded == kNumConv, "%s: %zu layer indices carried + %d folded != %d", tag, layers.size(), folded, kNumConv);
}

static void check_routing_and_work(const HmrPlan& pl, const char* tag) {
  double macs = 0, mfma = 0;
  for (const ConvSpec& c : pl.convs) {
    macs += c.macs_per_frame();
    mfma += c.mfma_macs_per_frame(pl.precision == 1 ? 64 : kConvBK);
    CHECK(c.w && c.bias, "%s: layer %d has no weights", tag, c.layer);
    CHECK(c.cfg == -1 || c.cfg == kConvCfgPanel || c.cfg == kConvCfgExpand || c.cfg == kConvCfgRegW, "%s: layer %d routed to %d", tag, c.layer, c.cfg);
    if (c.cfg == kConvCfgRegW) CHECK(pl.precision == 0 && c.k == 1 && c.stride == 1 && (c.Cin == 128 || c.Cin == 256) && c.Cout % 64 == 0 && c.in2_buf < 0, "%s: regw route of layer %d", tag, c.layer);
    if (c.cfg == kConvCfgExpand) CHECK(pl.precision == 1 && c.k == 1 && c.stride == 1 && ((c.res_buf >= 0 && expand_res_bf16_fits(c.Cin, c.Cout)) || (c.in2_buf >= 0 && expand_dual_bf16_fits(c.Cin, c.Cin2, c.Cout))), "%s: expand route of layer %d", tag, c.layer);
    if (c.cfg == kConvCfgPanel) CHECK(c.k == 1 && c.stride == 1 && c.Cout > c.Cin && c.Cin + c.Cin2 <= pl.panel_max_k, "%s: panel route of layer %d", tag, c.layer);
    if (c.splitk > 1) CHECK(pl.precision == 0 && c.Cout == 512 && c.Ho() == 7, "%s: split-K on layer %d", tag, c.layer);
  }
  CHECK(macs == 4087136256.0, "%s: %.0f multiply-adds per frame, SURVEY.md 8d says 4 087 136 256", tag, macs);
  bool any_wino = false;
  for (const ConvSpec& c : pl.convs) any_wino = any_wino || c.wino_m;
  if (!any_wino) CHECK(mfma >= macs, "%s: executed %.0f < algorithmic %.0f without a Winograd layer", tag, mfma, macs);
  else CHECK(mfma < macs, "%s: Winograd layers execute fewer products", tag);
  CHECK(pl.fc1x.K == 2048 && pl.fc1x.N == 1024 && pl.fc1s.K == kStateStride && pl.fc1s.N == 1024 && pl.fc2.K == 1024 && pl.fc2.N == 1024 &&
            pl.dec.K == 1024 && pl.dec.N == kStateStride && pl.init157 && pl.xf && pl.h_static && pl.h1 && pl.h2 && pl.state,
        "%s: regressor plan", tag);
}

static void check_sizes_and_counts(const HmrPlan& pl, const char* tag, int want_launches, int want_wino) {
  for (int n : {1, 2, 8}) {
    if (n > pl.max_batch) continue;
    const int cap = hmr_chunk_cap(pl.max_batch, n);
    CHECK(cap >= 1 && cap <= 512 && (long)cap * n >= std::min(pl.max_batch, 512 * n), "%s: chunk_cap %d for %d sub-batches of %d", tag, cap, n, pl.max_batch);
    const HmrChunkSizes z = hmr_chunk_sizes(pl, cap);
    const size_t elem = pl.precision == 1 ? 2 : 4;
    CHECK(z.act_floats * 4 >= (size_t)cap * 112 * 112 * 64 * elem, "%s: feature-map buffer too small", tag);
    CHECK((size_t)cap * 112 * 112 * 64 * elem < (1ull << 31), "%s: a launch's tensor reaches 2 GiB at %d frames", tag, cap);
    CHECK(z.act0_floats >= (size_t)cap * 112 * 112 * (pl.precision == 1 ? 16 / 2 : 12), "%s: input buffer too small", tag);
    CHECK((z.wino_floats != 0) == (pl.wino_floats_per_frame != 0), "%s: Winograd workspace", tag);
    int launches = 0, wino = 0;
    hmr_plan_counts(pl, pl.max_batch, cap, n, false, &launches, &wino);
    const int passes = (pl.max_batch + cap - 1) / cap;
    if (want_launches >= 0 && n == 1) CHECK(launches == want_launches * passes && wino == want_wino * passes, "%s: %d launches / %d Winograd layers per forward, expected %d / %d x %d", tag, launches, wino, want_launches, want_wino, passes);
    // the split the counts walk is the one pr_hmr_forward runs: the shares cover the batch, none exceeds a sub-batch's buffers
    for (int B : {1, pl.max_batch / 2 + 1, pl.max_batch})
      for (bool serial : {false, true}) {
        std::vector<int> sizes(4096);
        bool conc = false;
        const int ns = hmr_split_batch(B, cap, n, serial, sizes.data(), 4096, &conc);
        long sum = 0;
        for (int i = 0; i < ns; ++i) {
          sum += sizes[i];
          CHECK(sizes[i] >= 1 && sizes[i] <= cap, "%s: share %d of %d frames (cap %d)", tag, i, sizes[i], cap);
        }
        CHECK(sum == B && ns >= 1 && (!conc || (ns == std::min(n, B) && !serial)), "%s: split of %d frames into %d (n %d, serial %d)", tag, B, ns, n, (int)serial);
      }
  }
}

// Launch counts where the sub-batches of one forward differ in size (round 5's advisor: the counts assumed every pass was
// min(B, chunk_cap) frames): the whole-block layer3 kernel is taken per sub-batch (hmr_fused3_pays), so a short last pass
// or halved shares launch more kernels than a full one.
static void check_counts_follow_the_split(const HmrPlan& bf16_plan, const char* tag) {
  int full = 0, part = 0, w = 0;
  const int cus = bf16_plan.cus;
  hmr_plan_counts(bf16_plan, cus, cus, 1, true, &full, &w);            // one pass of `cus` frames: the fused blocks pay
  hmr_plan_counts(bf16_plan, cus / 4, cus, 1, true, &part, &w);        // a quarter of the CUs: they do not
  CHECK(hmr_fused3_pays(cus, cus) && !hmr_fused3_pays(cus / 4, cus) && part > full, "%s: %d launches at %d frames, %d at %d", tag, full, cus, part, cus / 4);
  int mixed = 0;
  hmr_plan_counts(bf16_plan, cus + cus / 4, cus, 1, true, &mixed, &w); // a full pass and a short one
  CHECK(mixed == full + part, "%s: passes of %d + %d frames launch %d kernels, expected %d + %d", tag, cus, cus / 4, mixed, full, part);
  int halves = 0;
  hmr_plan_counts(bf16_plan, cus, cus, 2, false, &halves, &w);         // two concurrent shares of cus / 2 frames each
  int half = 0;
  hmr_plan_counts(bf16_plan, cus / 2, cus, 1, true, &half, &w);
  CHECK(halves == 2 * half, "%s: two shares launch %d kernels, one share %d", tag, halves, half);
  int prof = 0;
  hmr_plan_counts(bf16_plan, cus, cus, 2, true, &prof, &w);            // profile mode runs the same handle serially
  CHECK(prof == full, "%s: serial (profile) mode launches %d kernels, expected %d", tag, prof, full);
}

struct Built {
  HmrPlan plan;
  HostSink sink;
};

static bool build(Built& b, const std::vector<float>& blob, int precision, int form, int max_batch, const char* tag) {
  hmr_plan_configure(&b.plan, precision, form, max_batch);
  const int st = hmr_plan_build(&b.plan, blob.data(), blob.size(), b.sink);
  CHECK(st == PR_OK, "%s: hmr_plan_build failed: %s", tag, g_err.c_str());
  return st == PR_OK;
}

int main(int argc, char** argv) {
  if (argc < 4) {
    fprintf(stderr, "usage: host_plan_check <blob.f32> <manifest.json> <dump.bin>\n");
    return 2;
  }
  std::vector<float> blob(hmr_weight_floats());
  {
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(blob.data(), 4, blob.size(), f) != blob.size() || fgetc(f) != EOF) {
      fprintf(stderr, "blob %s does not hold exactly %zu floats\n", argv[1], blob.size());
      return 2;
    }
    fclose(f);
  }
  int plans = 0;
  // ---- refusals -----------------------------------------------------------------------------------------------------
  {
    Built b;
    hmr_plan_configure(&b.plan, 0, PR_CONV_FORM_DEFAULT, 64);
    CHECK(hmr_plan_build(&b.plan, blob.data(), blob.size() - 1, b.sink) == PR_ERR_INVALID && g_err.find("floats") != std::string::npos, "a short blob is refused by size");
    hmr_plan_configure(&b.plan, 0, PR_CONV_FORM_DEFAULT, 0);
    CHECK(hmr_plan_build(&b.plan, blob.data(), blob.size(), b.sink) == PR_ERR_INVALID, "max_batch 0 is refused");
    CHECK(hmr_conv_form_valid(-1) && hmr_conv_form_valid(0) && hmr_conv_form_valid(5) && hmr_conv_form_valid(244) && hmr_conv_form_valid(505) &&
              !hmr_conv_form_valid(3) && !hmr_conv_form_valid(1) && !hmr_conv_form_valid(243) && !hmr_conv_form_valid(1000) && !hmr_conv_form_valid(-2),
          "conv_form validation");
  }
  // ---- every conv form (fp32) at B = 64 ---------------------------------------------------------------------------------
  const int forms[] = {PR_CONV_FORM_DEFAULT, 0, 2, 4, 5, 244, 455, 505};
  for (int form : forms) {
    char tag[64];
    snprintf(tag, sizeof tag, "fp32 form %d B=64", form);
    Built b;
    if (!build(b, blob, 0, form, 64, tag)) continue;
    ++plans;
    check_dataflow(b.plan, tag);
    check_routing_and_work(b.plan, tag);
    int wino = 0;
    for (const ConvSpec& c : b.plan.convs) wino += c.wino_m ? 1 : 0;
    const int want = form == 0 ? 0 : form == 505 ? 5 : 10;     // 505 = form 5 in layer2 (3 layers) and layer4 (2), layer3 direct
    CHECK(wino == want, "%s: %d Winograd layers, expected %d", tag, wino, want);
    check_sizes_and_counts(b.plan, tag, 47, want);
    if (form == PR_CONV_FORM_DEFAULT) CHECK(b.plan.conv_form == PR_CONV_FORM_BUILTIN_DEFAULT, "default form resolves to %d", b.plan.conv_form);
  }
  // ---- batch sizes, both precisions ---------------------------------------------------------------------------------------
  for (int precision : {0, 1})
    for (int B : {1, 7, 64, 230, 256, 460}) {
      char tag[64];
      snprintf(tag, sizeof tag, "%s default B=%d", precision ? "bf16" : "fp32", B);
      Built b;
      if (!build(b, blob, precision, PR_CONV_FORM_DEFAULT, B, tag)) continue;
      ++plans;
      check_dataflow(b.plan, tag);
      check_routing_and_work(b.plan, tag);
      // bf16: 37 launches in the plan; layer3's five plain blocks collapse to one launch each when the batch fills the CUs
      const bool f3 = hmr_fused3_pays(std::min(B, 512), b.plan.cus);
      check_sizes_and_counts(b.plan, tag, precision ? (f3 ? 27 : 37) : 47, precision ? 0 : 10);
      if (precision) CHECK(b.plan.convs.size() == 37 && b.plan.fused3.size() == 5 && b.plan.wino_floats_per_frame == 0, "%s: %zu launches, %zu fused layer3 blocks", tag, b.plan.convs.size(), b.plan.fused3.size());
      if (precision && B == 460) check_counts_follow_the_split(b.plan, tag);
      if ((B == 64) && argc >= 4) {
        // manifest + raw bytes of the first upload of every size (the Python test recomputes them)
        FILE* mf = fopen(argv[2], precision ? "a" : "w");
        FILE* df = fopen(argv[3], precision ? "ab" : "wb");
        if (!mf || !df) { fprintf(stderr, "cannot write %s / %s\n", argv[2], argv[3]); return 2; }
        fprintf(mf, "%s{\"precision\": %d, \"uploads\": [", precision ? "" : "[", precision);
        std::set<size_t> seen;
        long pos = ftell(df);
        for (size_t i = 0; i < b.sink.ups.size(); ++i) {
          const Upload& u = b.sink.ups[i];
          long at = -1;
          // the first upload of every size, and the regressor's (the last nine before the zero-filled workspaces)
          const bool regressor = i + 5 + 9 >= b.sink.ups.size();
          if (!u.zeros && (seen.insert(u.bytes).second || regressor)) {
            at = pos;
            fwrite(u.ptr, 1, u.bytes, df);
            pos += (long)u.bytes;
          }
          fprintf(mf, "%s{\"i\": %zu, \"bytes\": %zu, \"fnv\": \"%016" PRIx64 "\", \"zeros\": %d, \"dump_at\": %ld}", i ? ", " : "", i, u.bytes, u.fnv, u.zeros ? 1 : 0, at);
        }
        fprintf(mf, "]}%s\n", precision ? "]" : ",");
        fclose(mf);
        fclose(df);
      }
    }
  // ---- every plan-shaping switch, one at a time (read once per handle by hmr_plan_configure) --------------------------------
  struct Sw { const char* name; const char* value; int precision; int launches; int wino; };
  const Sw switches[] = {
      {"POSERISK_FUSE_DOWNSAMPLE", "0", 0, 51, 10}, {"POSERISK_FUSE_CONV3", "0", 0, 49, 10}, {"POSERISK_STEM_S2D", "0", 0, 47, 10},
      {"POSERISK_REGW", "0", 0, 47, 10}, {"POSERISK_PANEL_MAX_K", "0", 0, 47, 10}, {"POSERISK_SPLITK", "2", 0, 47, 10},
      {"POSERISK_WINOGRAD", "244", 0, 47, 10}, {"POSERISK_WINOGRAD_MIN_C", "256", 0, 47, 7},
      {"POSERISK_FUSE_DOWNSAMPLE", "0", 1, -1, 0}, {"POSERISK_FUSE_BOTTLENECK", "0", 1, -1, 0}, {"POSERISK_FUSE_BOTTLENECK2", "0", 1, -1, 0},
      {"POSERISK_FUSE_BOTTLENECK3", "0", 1, -1, 0}, {"POSERISK_EXPAND_REGS", "0", 1, -1, 0}, {"POSERISK_STEM_S2D", "0", 1, -1, 0},
      {"POSERISK_PANEL_MAX_K", "128", 1, -1, 0},
  };
  for (const Sw& sw : switches) {
    char tag[96];
    snprintf(tag, sizeof tag, "%s %s=%s B=256", sw.precision ? "bf16" : "fp32", sw.name, sw.value);
    setenv(sw.name, sw.value, 1);
    Built b;
    const bool ok = build(b, blob, sw.precision, PR_CONV_FORM_DEFAULT, 256, tag);
    unsetenv(sw.name);
    if (!ok) continue;
    ++plans;
    check_dataflow(b.plan, tag);
    check_routing_and_work(b.plan, tag);
    check_sizes_and_counts(b.plan, tag, sw.launches, sw.wino);
    if (!strcmp(sw.name, "POSERISK_SPLITK")) {
      int n = 0;
      for (const ConvSpec& c : b.plan.convs) n += c.splitk > 1;
      const HmrChunkSizes z = hmr_chunk_sizes(b.plan, 256);
      CHECK(n >= 1 && z.slab_floats > 0 && z.tickets > 0, "%s: %d split layers, slab %zu", tag, n, z.slab_floats);
    }
    if (!strcmp(sw.name, "POSERISK_FUSE_BOTTLENECK3")) CHECK(b.plan.fused3.empty(), "%s: fused layer3 blocks still planned", tag);
  }
  printf("host_plan_check: %d plans built and checked, %d failures\n", plans, g_fail);
  return g_fail ? 1 : 0;
}
