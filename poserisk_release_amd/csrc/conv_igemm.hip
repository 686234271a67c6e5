// fp32 implicit-GEMM convolution for gfx950 on v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD).
//
// Data flow per workgroup (BM x BN output tile, BK = 32 floats of K per stage):
//   HBM/L2 --global_load_dwordx4--> VGPR --ds_write_b128--> LDS [rows][36 floats]
//        (144-byte rows: every 16-lane ds_read_b128 group lands on 16 distinct 16-B slots)
//   LDS --ds_read_b128--> 4 k-values per lane --> 4 MFMAs (lane half h feeds k = 8kk+4h+j)
//   the next stage's global loads are issued before the MFMAs of the current one and
//   written to the other LDS buffer after them (one barrier per stage).
// Epilogue: + folded-BN bias, + residual, ReLU, dword stores (32 lanes = 128 contiguous bytes).
// Workgroup ids are remapped so that the 8 XCDs each own a contiguous run of M-tiles
// (all N-tiles of an M-tile share one L2).
#include "conv_igemm.h"

#include <cstdlib>
#include <cstring>
#include <vector>

namespace pr {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int BK = kConvBK;
constexpr int LDS_STRIDE = BK + 4;  // floats
constexpr int CHUNKS = BK / 4;      // 16-byte chunks per row per stage

struct KArgs {
  const float* x;
  const float* w;
  const float* bias;
  const float* res;
  float* y;
  int H, W, Cin, log2Cin, Ho, Wo, HoWo, Cout, stride, pad;
  int M, K, Kpad, nk;
  int tiles_n;
  int relu;
};

template <int BM, int BN, int WAVES_M, int WAVES_N, int KS>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void conv_igemm_f32(const KArgs a) {
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int MI = WM / 32, NI = WN / 32;
  constexpr int RPP = NT / CHUNKS;  // rows staged per pass of the whole workgroup
  constexpr int A_LOADS = BM / RPP, B_LOADS = BN / RPP;
  static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be a multiple of rows-per-pass");
  static_assert(WM % 32 == 0 && WN % 32 == 0, "wave tile is made of 32x32 MFMA blocks");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                        // [2][BM][LDS_STRIDE]
  float* Bs = smem + 2 * BM * LDS_STRIDE;  // [2][BN][LDS_STRIDE]

  // XCD-aware (bijective) remap: blocks b and b+8 share an XCD; give each XCD a contiguous
  // run of logical tiles, N-tile fastest.
  const int nb = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q = nb >> 3, rr = nb & 7;
  const int logical = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
  const int tile_n = logical % a.tiles_n, tile_m = logical / a.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int c = tid & (CHUNKS - 1), r0 = tid / CHUNKS;

  // Per-thread A-row descriptors (constant over K).
  long a_base[A_LOADS];
  int a_hi0[A_LOADS], a_wi0[A_LOADS];
#pragma unroll
  for (int i = 0; i < A_LOADS; ++i) {
    const int m = m0 + r0 + i * RPP;
    if (m < a.M) {
      const int img = m / a.HoWo, rem = m - img * a.HoWo;
      const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
      a_hi0[i] = ho * a.stride - a.pad;
      a_wi0[i] = wo * a.stride - a.pad;
      a_base[i] = ((long)(img * a.H + a_hi0[i]) * a.W + a_wi0[i]) * a.Cin;
    } else {
      a_hi0[i] = -(1 << 28);  // never in [0,H)
      a_wi0[i] = 0;
      a_base[i] = 0;
    }
  }
  const float* wrow = a.w + (long)(n0 + r0) * a.Kpad + c * 4;

  f32x4 ra[A_LOADS], rb[B_LOADS];
  auto gload = [&](int kt) {
    const int k = kt * BK + c * 4;
    int kh = 0, kw = 0, ci = k;
    if (KS > 1) {
      const int tap = k >> a.log2Cin;
      ci = k & (a.Cin - 1);
      kh = tap / KS;
      kw = tap - kh * KS;
    }
    const bool kvalid = k < a.K;
    const long koff = ((long)kh * a.W + kw) * a.Cin + ci;
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
      const int hi = a_hi0[i] + kh, wi = a_wi0[i] + kw;
      const bool ok = kvalid && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok) v = *reinterpret_cast<const f32x4*>(a.x + a_base[i] + koff);
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i)
      rb[i] = *reinterpret_cast<const f32x4*>(wrow + (long)i * RPP * a.Kpad + kt * BK);
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i)
      *reinterpret_cast<f32x4*>(&As[(buf * BM + r0 + i * RPP) * LDS_STRIDE + c * 4]) = ra[i];
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i)
      *reinterpret_cast<f32x4*>(&Bs[(buf * BN + r0 + i * RPP) * LDS_STRIDE + c * 4]) = rb[i];
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  const int frag_off = (lane & 31) * LDS_STRIDE + (lane >> 5) * 4;
  auto compute = [&](int buf) {
    const float* Ab = As + (buf * BM + wm * WM) * LDS_STRIDE + frag_off;
    const float* Bb = Bs + (buf * BN + wn * WN) * LDS_STRIDE + frag_off;
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      f32x4 af[MI], bf[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        af[mi] = *reinterpret_cast<const f32x4*>(Ab + mi * 32 * LDS_STRIDE + kk * 8);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        bf[ni] = *reinterpret_cast<const f32x4*>(Bb + ni * 32 * LDS_STRIDE + kk * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi][j], bf[ni][j], acc[mi][ni], 0, 0, 0);
    }
  };

  gload(0);
  sstore(0);
  __syncthreads();
  for (int kt = 0; kt < a.nk; ++kt) {
    const int buf = kt & 1;
    const bool more = kt + 1 < a.nk;
    if (more) gload(kt + 1);
    compute(buf);
    if (more) sstore(buf ^ 1);
    __syncthreads();
  }

  // Epilogue.  C/D map of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5).
  const int col_l = lane & 31, row_h = 4 * (lane >> 5);
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int col = n0 + wn * WN + ni * 32 + col_l;
    const float bv = a.bias ? a.bias[col] : 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int rbase = m0 + wm * WM + mi * 32 + row_h;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = rbase + (e & 3) + 8 * (e >> 2);
        if (row < a.M) {
          const long o = (long)row * a.Cout + col;
          float v = acc[mi][ni][e] + bv;
          if (a.res) v += a.res[o];
          if (a.relu) v = fmaxf(v, 0.f);
          a.y[o] = v;
        }
      }
    }
  }
}

struct TileCfg {
  int BM, BN, threads;
  const char* name;
  int blocks_per_cu;   // LDS-limited residency
};

constexpr int kNumRegCfg = 6;  // 0..5: register-staged kernel of this file; 6..11: LDS-DMA kernel (conv_dma.hip)
constexpr int kNumCfg = 18;
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
};

template <int BM, int BN, int WAVES_M, int WAVES_N>
int launch_ks(const KArgs& ka, int ks, int grid, hipStream_t stream) {
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr size_t lds = (size_t)2 * (BM + BN) * LDS_STRIDE * sizeof(float);
  auto go = [&](auto kern) -> int {
    static std::atomic<uint64_t> attr_done{0};  // per instantiation (one per lambda instantiation), one bit per device
    PR_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, attr_done));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, stream, ka);
    return check_launch("conv_igemm_f32");
  };
  switch (ks) {
    case 1: return go(conv_igemm_f32<BM, BN, WAVES_M, WAVES_N, 1>);
    case 3: return go(conv_igemm_f32<BM, BN, WAVES_M, WAVES_N, 3>);
    case 7: return go(conv_igemm_f32<BM, BN, WAVES_M, WAVES_N, 7>);
    default: set_error("unsupported kernel size %d", ks); return PR_ERR_INVALID;
  }
}

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
  return t;
}

int conv_num_tile_cfgs() { return kNumCfg; }
const char* conv_tile_cfg_name(int cfg) { return (cfg >= 0 && cfg < kNumCfg) ? kCfgs[cfg].name : "?"; }

int conv_pick_tile_cfg(const ConvProblem& p) {
  // Experiment hook (ConvTuning::force_cfg): one tile configuration wherever it fits.
  const int forced = p.tune.force_cfg;
  if (forced >= 0 && forced < kNumCfg && p.Cout % kCfgs[forced].BN == 0 && p.M() >= kCfgs[forced].BM) return forced;
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
          row[(kh * KW + kw) * cin_pad + ci] =
              f32_to_bf16_host((float)((double)w[(((size_t)o * Cin_real + ci) * KH + kh) * KW + kw] * s));
  }
}

int conv_launch(const ConvProblem& p, int cfg, hipStream_t stream) {
  if (cfg == kConvCfgPanel) return conv_panel_launch(p, stream);
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
  const int l2 = ilog2_exact(p.Cin);
  if (cfg >= kNumRegCfg) return conv_dma_launch(p, t.BM, t.BN, stream, t.threads);
  PR_REQUIRE(p.KH == 1 || l2 >= 0, "conv: k>1 needs power-of-two Cin (%d)", p.Cin);
  PR_REQUIRE(!p.x2 && p.groups == 1, "conv: second source / groups run on the LDS-DMA tile configs (>= %d) only", kNumRegCfg);
  PR_REQUIRE((long)p.B * p.H * p.W * p.Cin < (1L << 31) && (long)p.M() * p.Cout < (1L << 31),
             "conv: tensor too large for one call");
  KArgs ka;
  ka.x = p.x; ka.w = p.w; ka.bias = p.bias; ka.res = p.res; ka.y = p.y;
  ka.H = p.H; ka.W = p.W; ka.Cin = p.Cin; ka.log2Cin = l2 < 0 ? 0 : l2;
  ka.Ho = p.Ho; ka.Wo = p.Wo; ka.HoWo = p.Ho * p.Wo; ka.Cout = p.Cout;
  ka.stride = p.stride; ka.pad = p.pad;
  ka.M = p.M(); ka.K = p.K(); ka.Kpad = p.Kpad(); ka.nk = ka.Kpad / BK;
  ka.tiles_n = p.Cout / t.BN;
  ka.relu = p.relu;
  if (ka.M == 0) return PR_OK;
  const int grid = ceil_div(ka.M, t.BM) * ka.tiles_n;
  switch (cfg) {
    case 0: return launch_ks<128, 128, 2, 2>(ka, p.KH, grid, stream);
    case 1: return launch_ks<128, 64, 2, 2>(ka, p.KH, grid, stream);
    case 2: return launch_ks<64, 64, 2, 2>(ka, p.KH, grid, stream);
    case 3: return launch_ks<256, 128, 4, 2>(ka, p.KH, grid, stream);
    case 4: return launch_ks<64, 128, 2, 2>(ka, p.KH, grid, stream);
    case 5: return launch_ks<256, 64, 4, 2>(ka, p.KH, grid, stream);
  }
  return PR_ERR_INVALID;
}

void conv_pack_weights(const float* w, const double* scale, int Cout, int Cin_real, int cin_pad,
                       int KH, int KW, float* out) {
  const int K = KH * KW * cin_pad;
  const int Kpad = ceil_div(K, BK) * BK;
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

}  // namespace pr
