// Host side of the convolution launches: tile-configuration table and choice, weight packing (fp32 and bf16), dispatch to
// the LDS-DMA kernels (conv_dma.hip, conv_dma_bf16.hip), the fused kernels (conv_fused.hip) and the row-panel form.
// The first-generation kernel that lived here (register-staged operands, padded LDS rows; tile configs 0-5) was retired
// in round 3: production ran on the LDS-DMA kernels since round 1 and the parity tests compare every configuration with
// torch, not with it.  The indices 0-5 stay reserved (they return PR_ERR_INVALID) so that 6.. keep their numbers.
#include "conv_igemm.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace pr {
namespace {

constexpr int BK = kConvBK;

struct TileCfg {
  int BM, BN, threads;
  const char* name;
  int blocks_per_cu;   // LDS-limited residency
};

constexpr int kNumRegCfg = 6;  // 0..5: the first-generation register-staged kernel, retired in round 3 (indices kept); 6..: LDS-DMA kernels (conv_dma.hip)
constexpr int kNumCfg = 19;
const TileCfg kCfgs[kNumCfg] = {
    {128, 128, 256, "reg_128x128x32_w2x2", 2},
    {128, 64, 256, "reg_128x64x32_w2x2", 2},
    {64, 64, 256, "reg_64x64x32_w2x2", 4},
    {256, 128, 512, "reg_256x128x32_w4x2", 1},
    {64, 128, 256, "reg_64x128x32_w2x2", 2},
    {256, 64, 512, "reg_256x64x32_w4x2", 1},
    {128, 128, 256, "dma_128x128x32_w2x2", 2},
    {128, 64, 256, "dma_128x64x32_w2x2", 3},
    {64, 64, 256, "dma_64x64x32_w2x2", 5},
    {256, 128, 512, "dma_256x128x32_w4x2", 1},
    {64, 128, 256, "dma_64x128x32_w2x2", 3},
    {256, 64, 512, "dma_256x64x32_w4x2", 2},
    {128, 128, 512, "dma_128x128x32_w4x2", 2},   // 8 waves per 128x128 tile (32x64 per wave)
    {128, 64, 512, "dma_128x64x32_w4x2", 3},     // 8 waves per 128x64 tile (32x32 per wave)
    {64, 64, 128, "dma_64x64x32_w2x1", 5},       // 2 waves per 64x64 tile (32x64 per wave)
    {128, 64, 128, "dma_128x64x32_w2x1", 3},     // 2 waves per 128x64 tile (64x64 per wave)
    {64, 128, 128, "dma_64x128x32_w1x2", 3},     // 2 waves per 64x128 tile (64x64 per wave)
    {64, 256, 256, "dma_64x256x32_w2x2", 2},     // whole 256-channel rows per tile (32x128 per wave): short-K conv3
    {256, 256, 512, "dma_256x256x64_w4x2_bf16", 1},   // bf16 ONLY (fp32's epilogue and registers do not take it): 64x128 per wave
};

int ilog2_exact(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return (1 << l) == v ? l : -1;
}

}  // namespace

ConvTuning conv_tuning_from_env() {
  ConvTuning t;
  if (const char* e = getenv("POSERISK_CONV_CFG")) t.force_cfg = atoi(e);
  if (const char* e = getenv("POSERISK_CONV_TAIL")) t.tail = atoi(e);
  if (const char* e = getenv("POSERISK_TAIL_MIN_ROUNDS")) t.tail_min_rounds = atoi(e);
  if (const char* e = getenv("POSERISK_TAIL_MAX_REM")) t.tail_max_rem = atoi(e);
  if (const char* e = getenv("POSERISK_WINO_VEC")) t.wino_vec = atoi(e) == 4 ? 4 : 2;
  if (const char* e = getenv("POSERISK_WINO_REGW")) t.wino_regw = atoi(e) != 0;
  if (const char* e = getenv("POSERISK_REGW_PER_CU")) { const int v = atoi(e); if (v >= 1 && v <= 4) t.regw_per_cu = v; }
  if (const char* e = getenv("POSERISK_REGW_T")) t.regw_t = atoi(e);
  if (const char* e = getenv("POSERISK_REGW_NB")) t.regw_nb = atoi(e);
  if (const char* e = getenv("POSERISK_REGW_WT")) t.regw_wt = atoi(e);
  if (const char* e = getenv("POSERISK_REGW_WNB")) t.regw_wnb = atoi(e);
  if (const char* e = getenv("POSERISK_WINO_TILE")) {
    int bm = 0, bn = 0;
    if (sscanf(e, "%dx%d", &bm, &bn) == 2 && (bm == 64 || bm == 128) && (bn == 64 || bn == 128)) { t.wino_bm = bm; t.wino_bn = bn; }
  }
  if (const char* e = getenv("POSERISK_BAL_STAGES")) t.bal_stages = atoi(e);   // 4, 5, or 6 = ring of five with paired stages
  return t;
}

int conv_num_tile_cfgs() { return kNumCfg; }
const char* conv_tile_cfg_name(int cfg) { return (cfg >= 0 && cfg < kNumCfg) ? kCfgs[cfg].name : "?"; }

int conv_pick_tile_cfg(const ConvProblem& p) {
  // Experiment hook (ConvTuning::force_cfg): one tile configuration wherever it fits.
  const int forced = p.tune.force_cfg;
  // (fp32 dual-source and split-K launches exist on the 64x64 tile only: they keep it)
  const bool fixed_tile = p.precision == 0 && (p.x2 || p.splitk > 1);
  if (forced >= 0 && forced < kNumCfg && !fixed_tile && p.Cout % kCfgs[forced].BN == 0 && p.M() >= kCfgs[forced].BM &&
      (p.precision == 1 || forced != 18))
    return forced;
  if (p.precision == 1) {
    // bf16: the MFMA is 16x faster, so the kernel lives on L2->LDS bandwidth and wants big tiles.  Per-layer times inside
    // the B=256 pipeline, every tile configuration in turn (gpurun_out/r02_layers256_bf16_cfg*.txt; round 1's isolated
    // sweep without residuals had put the 256x64 tile first): the 8-wave 128x128 tile (32x64 per wave) is the fastest on
    // every layer with >= 128 output channels (4.18 ms of conv per step against 4.49 with 256x64) except layer2's short-K
    // expansions with residual, where 128x64 wins (125 vs 140 us); 64-channel 1x1 layers take 128x64, the stem and
    // layer1's 3x3 keep 256x64 (all tiles within 1 %).
    if (p.Cout % 128 == 0 && p.M() >= 128) return (p.KH == 1 && !p.x2 && p.res && p.Cin <= 128) ? 13 : 12;
    if (p.M() >= 256) return p.KH == 1 ? 13 : 11;
    return 8;
  }
  // fp32: the 4-wave 64x64 LDS-DMA tile (5 workgroups per CU, quarter tiles for the remainder).  Sweeps of all 23
  // ResNet-50 shapes at B=64 and B=256 (profiles/r01_conv_tile_sweep_b64.txt, ..._b256_fp32.txt): it is the fastest or
  // within a few percent of the fastest configuration on every shape; an earlier cost model that weighed tile
  // efficiency against quantisation picked larger tiles at B=256 and lost 10 % of the conv time there.
  return 8;
}

int conv_tile_dims(int cfg, int* BM, int* BN) {
  if (cfg < 0 || cfg >= kNumCfg) return PR_ERR_INVALID;
  *BM = kCfgs[cfg].BM;
  *BN = kCfgs[cfg].BN;
  return PR_OK;
}

int conv_launch(const ConvProblem& p, int cfg, hipStream_t stream) {
  if (cfg == kConvCfgPanel) return conv_panel_launch(p, stream);
  if (cfg == kConvCfgRegW) return conv_regw_f32_launch(p, stream);
  if (cfg == kConvCfgBalanced || cfg == kConvCfgBalanced + 1) return conv_bal_bf16_launch(p, stream, cfg - kConvCfgBalanced);
  if (cfg == kConvCfgExpand) {
    PR_REQUIRE(p.precision == 1 && p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && p.bias && !p.w3 && p.groups == 1 &&
                   (p.x2 != nullptr) != (p.res != nullptr),
               "conv: tile cfg %d is the bf16 1x1 expansion with register-resident weights: + bias + residual, or + a second "
               "source without residual", cfg);
    if (p.x2)
      return expand_dual_bf16_launch(p.x, p.x2, p.w, p.bias, p.y, p.B, p.Ho, p.Wo, p.H2, p.W2, p.stride2, p.Cin, p.Cin2, p.Cout,
                                     p.relu, stream);
    return expand_res_bf16_launch(p.x, p.w, p.bias, p.res, p.y, (long)p.M(), p.Cin, p.Cout, p.relu, stream);
  }
  PR_REQUIRE(cfg >= 0 && cfg < kNumCfg, "conv: bad tile cfg %d", cfg);
  const TileCfg& t = kCfgs[cfg];
  if (p.w3) return conv_fused3_launch(p, stream);
  if (p.precision == 1) {
    PR_REQUIRE(cfg >= kNumRegCfg, "conv: bf16 runs on the LDS-DMA tile configs (>= %d) only", kNumRegCfg);
    return conv_dma_bf16_launch(p, t.BM, t.BN, stream, t.threads);
  }
  PR_REQUIRE(p.KH == p.KW, "conv: square kernels only (got %dx%d)", p.KH, p.KW);
  PR_REQUIRE(p.Cin % 4 == 0, "conv: Cin %% 4 != 0 (%d)", p.Cin);
  PR_REQUIRE(p.Cout % t.BN == 0, "conv: Cout %d not a multiple of tile N %d", p.Cout, t.BN);
  PR_REQUIRE(p.x && p.w && p.y, "conv: null tensor");
  PR_REQUIRE(cfg >= kNumRegCfg, "conv: tile cfg %d (%s) was the first-generation register-staged kernel, retired in round 3; "
             "the LDS-DMA kernels are cfgs %d..%d", cfg, t.name, kNumRegCfg, kNumCfg - 1);
  PR_REQUIRE(!(t.BM == 256 && t.BN == 256), "conv: tile cfg %d (%s) is a bf16-only tile", cfg, t.name);
  return conv_dma_launch(p, t.BM, t.BN, stream, t.threads);
}


}  // namespace pr
