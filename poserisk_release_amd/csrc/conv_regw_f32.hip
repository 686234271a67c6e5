// fp32 1x1 convolution with the WEIGHTS RESIDENT IN REGISTERS (K = 128 or 256): y = act(x W^T + bias + res).
// The encoder's short-K 1x1 layers (layer1's conv1 256 -> 64, layer2's conv3 128 -> 512, layer3's conv3 256 -> 1024; SPIN
// models/hmr.py Bottleneck, call site lib/core/base.py:220).
//
// Why: on the 64x64 tile kernel these layers are bound by the operand fetch -- 16 KB of L2 -> LDS per 1 024 MFMA cycles
// per workgroup, half of it weights (profiles/r04_experiments.txt).  A wave's 16 output channels x K weights
// are K / 4 values per lane: they are loaded once per workgroup and stay, only the activations stream (half the bytes, a
// quarter of the LDS-DMA instructions per MFMA).  Same construction as the fused stem (stem_pool_f32.hip).
//
//   Also the `groups` independent GEMMs of a Winograd layer (conv_winograd.hip: 36 products per layer; K = 128 on layer2,
//   256 on layer3): the tensors are flat over the groups, a unit never crosses one.
//   work unit = (block of 64 output channels, 32 consecutive pixels); a persistent workgroup of 4 waves takes a contiguous
//     run of units of ONE channel block where it can (the weights are reloaded when the block changes).
//   wave w: channels 64 nb + 16 w .. + 15, both 16-pixel tiles of the unit: two independent accumulators on
//     v_mfma_f32_16x16x4_f32 (a single dependent chain of that shape issues at 40 of 32 cycles).
//   A operand (weights): lane (m, g) holds W[16 wave + m][16 q + 4 g + j] for q < K / 16, j < 4 -- MFMA (q, j) sums
//     k = 16 q + 4 g + j over g.
//   B operand (pixels): the unit's 32 pixel rows (K floats each) sit in one of two LDS stages, 16-byte chunk c of row r in slot
//     c ^ (r & 15) (the LDS-DMA applies the XOR on its SOURCE address), so that the 16 lanes of a tile read 16 different
//     slots: lane (m, g) reads chunk 4 q + g of row m as one ds_read_b128 per four MFMAs, placed between the MFMAs; the DMA
//     of the next unit's rows goes between them too.  One barrier per unit.
//   Epilogue from the accumulators (which start at the bias): a lane holds 4 consecutive channels of one pixel; residual,
//     ReLU, one 16-byte buffer store per tile (the 4 lanes of a pixel make 64 contiguous bytes) -- in an MFMA slot of the
//     NEXT unit, the residual requested a unit ahead the same way.
// The k order inside an output's fmaf chain is (q, j, g) -- fixed, so a frame's bits do not depend on its batch; it is not
// the tile kernel's order.
//
// Round 5: NB channel blocks per pass.  With one 64-channel block per unit every block of a layer re-streams the layer's whole
// input through LDS (N / 64 = 8 - 16 passes: 2.9 GB per step of L2 -> LDS traffic at B = 64, a third of the encoder's reads;
// profiles/r05a_hbm_traffic_b64.json).  A wave now holds its 16 channels of NB consecutive blocks (NB x K / 4 weight registers:
// 128 at K = 256, NB = 2) and a unit is (NB x 64 channels, 16 T pixels): the pixel fragments read from LDS feed NB MFMAs each,
// the stage is filled once per NB blocks -- half (a quarter) of the LDS-DMA, the L2 reads and the fragment reads per MFMA.  An
// output's chain is still bias, then (q, j) ascending: the SAME BITS for every (T, NB).
#include "conv_igemm.h"

namespace pr {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
typedef __attribute__((address_space(3))) void lds_void;
constexpr unsigned kOOB = 0x80000000u;
// 16-pixel tiles per unit (T).  Measured in round 4 with 4 at K = 128 (64-pixel units, a unit's fixed costs once per 4 096
// MFMA cycles as at K = 256): no faster inside the loop and the shares get coarser (6 or 7 units per workgroup instead of 13 or
// 14): layer2's Winograd GEMM 44.3 -> 45.8 us, its conv3 70.1 -> 72.5 us.

struct RArgs {
  const float* x;       // [groups][M][K]
  const float* w;       // [groups][N][K]
  const float* bias;    // [N] or nullptr
  const float* res;     // [M][N] or nullptr
  float* y;             // [groups][M][N]
  unsigned x_bytes, y_bytes;
  int M, N, relu;
  int units, pp, nblk;  // pp = pixel groups of a GEMM (ceil(M / 32)), nblk = N / 64; unit u = (group * nblk + nb) * pp + pixel group
  int exp;              // timing builds only (POSERISK_REGW_EXP): 1 no output stores, 2 no LDS-DMA after the first unit, 4 no residual loads, 8 no barrier, 16 no fragment reads
  unsigned long long* stamps;   // timing builds only (-DPR_TIMING_HOOKS, POSERISK_REGW_STAMPS): per wave, s_memrealtime sums
};
#ifdef PR_TIMING_HOOKS
#define PR_RW_T() (a.stamps ? __builtin_amdgcn_s_memrealtime() : 0ull)
#define PR_RW_EXP(bit) (a.exp & (bit))
#else
#define PR_RW_T() 0ull
#define PR_RW_EXP(bit) 0
#endif

template <int K, int T, int NB>
__global__ __launch_bounds__(256) void conv1x1_regw_f32(const RArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  // T = 16-pixel tiles per unit, NB = 64-channel blocks per unit: T x NB accumulators per wave
  constexpr int kPx = 16 * T;                   // pixels per unit
  constexpr int ROW = K * 4;                    // bytes of a pixel row
  constexpr int STAGE = kPx * ROW;              // 16 / 32 KB
  constexpr int NQ = K / 16;                    // MFMA quads
  constexpr int NDMA = STAGE / 1024 / 4;        // LDS-DMA instructions per wave and unit (4 / 8)
  constexpr int NI = T * NB;                    // output items (16-byte stores) and residual loads per unit and wave
  static_assert(NDMA <= NQ && NI <= NQ && 4 * T * NB >= T + 3, "slots");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, g = lane >> 4;

  // this workgroup's run of units.  Unit u = (cb, pg): cb = u / pp = group * nblk + channel-block GROUP (NB blocks of 64; a.nblk
  // counts groups), pg = its pixel group (kPx rows).  With the tensors flat over the groups (x [groups * M][K], w [groups * N][K],
  // y [groups * M][N]) weight row = 64 NB cb + .., output column = 64 NB (cb % nblk) + .., first row = group * M + kPx pg, and
  // rows >= group * M + M are not this group's.
  const long G = gridDim.x;
  const int u0 = (int)((long)blockIdx.x * a.units / G), u1 = (int)((long)(blockIdx.x + 1) * a.units / G);
  if (u0 >= u1) return;

  const auto xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  // DMA piece i of this wave (i < NDMA): bytes [(wave * NDMA + i) * 1024, + 1024) of the stage; lane covers 16 of them.
  // row0 / lim: the unit's first flat row and its group's end
  auto issue_piece = [&](int row0, int lim, int stage, int i) {
    const int o = (wave * NDMA + i) * 1024 + lane * 16;
    const int r = o / ROW, s = (o % ROW) >> 4;
    const int c = s ^ (r & 15);
    const unsigned vo = (row0 + r < lim) ? (unsigned)((row0 + r) * ROW + c * 16) : kOOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, (lds_void*)(smem + stage * STAGE + (wave * NDMA + i) * 1024), 16, vo, 0, 0, 0);
  };
  auto unit_rows = [&](int u, int& row0, int& lim) {
    const int cb = u / a.pp, grp = cb / a.nblk;
    row0 = grp * a.M + (u - cb * a.pp) * kPx;
    lim = grp * a.M + a.M;
  };

  // A read addresses: row m of a tile, chunk 4 q + g -> slot (4 q + g) ^ m = (q >> 2) * 16 + ((4 (q & 3) + g) ^ m)
  int abase[4];
#pragma unroll
  for (int qq = 0; qq < 4; ++qq) abase[qq] = m * ROW + (((4 * qq + g) ^ m) << 4);

  float b[NB][NQ][4];
  [[maybe_unused]] unsigned long long t_begin = PR_RW_T(), t_w = 0, t_wait = 0, t_mfma = 0, n_units = 0;
  int nx_row0, nx_lim;                  // the unit whose rows are being fetched (one ahead of the one being multiplied)
  unit_rows(u0, nx_row0, nx_lim);
#pragma unroll
  for (int i = 0; i < NDMA; ++i) issue_piece(nx_row0, nx_lim, 0, i);
  int stage = 0;
  // The epilogue of unit u - 1 and the residual requests of unit u ride BETWEEN the MFMAs of unit u, one output (an add, a
  // max, a 4-byte buffer store) or one 4-byte buffer load per slot: a wave issues in order, so the same sixteen memory
  // instructions in a block behind the barrier kept its MFMAs from issuing for 0.9 us of a 4.3 us unit (per-wave stamps,
  // profiles/r04_experiments.txt section 5).  A row outside the unit's group gets the offset 2^31, which the buffer's
  // range check drops (stores) or answers with zero (loads); the "unit before the first" has no rows.
  const auto ysrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)a.y_bytes, 0x00020000);
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res ? a.res : a.y), 0, (int)a.y_bytes, 0x00020000);
  const bool has_res = a.res != nullptr;
  const unsigned n4 = (unsigned)a.N * 4u;
  // The weights are the MFMA's A operand and the pixels its B operand, so a lane's four accumulator values are four
  // consecutive CHANNELS (16 wave + 4 g ..) of ONE pixel (row 16 t + m of the unit): 16-byte stores and residual loads, two
  // of each per unit and wave, where pixels-as-A needed eight 4-byte ones.
  f32x4 pacc[NI] = {};                  // item i = nb * T + t: block nb of the unit, pixel tile t
  f32x4 prv[NI] = {};                   // residual of the unit whose accumulators sit in pacc: loaded a unit ahead
  unsigned p_off = 0;                   // that unit's byte offset of (row m, this lane's first channel of block 0)
  int p_rows = 0;                       // ... and how many of its kPx rows exist
  auto out_item = [&](int i) {
    const int nb = i / T, t = i % T;
    f32x4 v = pacc[i];
    if (has_res) v += prv[i];
    if (a.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
    const unsigned off = m < p_rows - 16 * t ? p_off + (unsigned)(16 * t) * n4 + (unsigned)nb * 256u : kOOB;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ysrc, off, 0, 0);
  };
  for (int ub = u0; ub < u1;) {
    // one channel block at a time: its weights are loaded here, OUTSIDE the unit loop (K / 16 16-byte loads per lane) and
    // waited for at once -- loaded under a condition inside the unit loop, the compiler's vmcnt bookkeeping made every
    // unit's MFMA loop wait for ALL outstanding vector-memory operations, the next unit's DMA included
    const int cb = ub / a.pp, grp = cb / a.nblk;
    const int ue = min(u1, (cb + 1) * a.pp);
    const int wrow = cb * (64 * NB) + 16 * wave + m;                 // weight row of block 0 (flat over the groups)
    const int ch4 = (cb - grp * a.nblk) * (64 * NB) + 16 * wave + 4 * g;   // this lane's first output channel of block 0
    const int g_row0 = grp * a.M;
    [[maybe_unused]] const unsigned long long tw0 = PR_RW_T();
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const f32x4* wr = reinterpret_cast<const f32x4*>(a.w + (size_t)(wrow + 64 * nb) * K) + g;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const f32x4 v = wr[q * 4];
        b[nb][q][0] = v[0]; b[nb][q][1] = v[1]; b[nb][q][2] = v[2]; b[nb][q][3] = v[3];
      }
    }
    f32x4 bias[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      bias[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (a.bias) bias[nb] = *reinterpret_cast<const f32x4*>(a.bias + ch4 + 64 * nb);
    }
    // the compiler's own waits for these loads must happen HERE (an empty asm that "uses" the registers): left to the first
    // use, they sit inside the unit loop as vmcnt(15) .. vmcnt(0) in front of every quad of EVERY unit and drain the next
    // unit's DMA and the residual loads with them
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) asm volatile("" : "+v"(b[nb][q][0]), "+v"(b[nb][q][1]), "+v"(b[nb][q][2]), "+v"(b[nb][q][3]));
      asm volatile("" : "+v"(bias[nb]));
    }
#ifdef PR_TIMING_HOOKS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t_w += PR_RW_T() - tw0;
#endif
    for (int u = ub; u < ue; ++u) {
      const int row0 = g_row0 + (u - cb * a.pp) * kPx;                // this unit's first flat row
      const int rows = min(kPx, g_row0 + a.M - row0);
      const unsigned off = (unsigned)(row0 + m) * n4 + (unsigned)ch4 * 4u;
      const bool more = u + 1 < u1;
      if (more) {
        if (u + 1 < ue) nx_row0 = row0 + kPx;                         // (nx_lim stays)
        else unit_rows(u + 1, nx_row0, nx_lim);
      }
      [[maybe_unused]] const unsigned long long t0 = PR_RW_T();
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // own DMA pieces of this unit have landed (issued a unit ago)
      // ... and so has the previous unit's residual: tell the compiler here (see the weights above), or it waits for
      // everything outstanding in front of each of the eight adds
#pragma unroll
      for (int i = 0; i < NI; ++i) asm volatile("" : "+v"(prv[i]));
      if (!PR_RW_EXP(8)) __builtin_amdgcn_s_barrier();                // everyone's have; everyone has finished the other stage
      asm volatile("" ::: "memory");
      [[maybe_unused]] const unsigned long long t1 = PR_RW_T();
      const char* st = smem + stage * STAGE;
      f32x4 acc[NI];
      f32x4 av[2][T];
#pragma unroll
      for (int t = 0; t < T; ++t) av[0][t] = *reinterpret_cast<const f32x4*>(st + t * 16 * ROW + abase[0]);
#pragma unroll
      for (int i = 0; i < NI; ++i) acc[i] = bias[i / T];
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int cur = q & 1;
        int n = 0;                      // MFMA slot within the quad
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int t = 0; t < T; ++t) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
              acc[nb * T + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[nb][q][j], av[cur][t][j], acc[nb * T + t], 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
              // one instruction of the rest behind each MFMA.  Slots of quad q, in MFMA order: 0 .. T - 1 the next quad's
              // fragment of tile n; T one LDS-DMA piece of the next unit; T + 1 item q of the PREVIOUS unit goes out; T + 2 the
              // request for the residual that will be added to item q of THIS unit a unit from now (into the registers the
              // output has just read)
              if (n < T && q + 1 < NQ && !PR_RW_EXP(16)) {
                av[cur ^ 1][n] = *reinterpret_cast<const f32x4*>(st + n * 16 * ROW + ((q + 1) >> 2) * 256 + abase[(q + 1) & 3]);
                __builtin_amdgcn_sched_barrier(0);
              } else if (n == T && q < NDMA) {
                if (more && !PR_RW_EXP(2)) issue_piece(nx_row0, nx_lim, stage ^ 1, q);
                __builtin_amdgcn_sched_barrier(0);
              } else if (n == T + 1 && q < NI) {
                if (!PR_RW_EXP(1)) out_item(q);
                __builtin_amdgcn_sched_barrier(0);
              } else if (n == T + 2 && q < NI && has_res && !PR_RW_EXP(4)) {
                prv[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                rsrc, m < rows - 16 * (q % T) ? off + (unsigned)(16 * (q % T)) * n4 + (unsigned)(q / T) * 256u : kOOB, 0, 0));
                __builtin_amdgcn_sched_barrier(0);
              }
              ++n;
            }
          }
        }
      }
#ifdef PR_TIMING_HOOKS
      if (a.stamps) {
        asm volatile("s_nop 0" : "+v"(acc[0]));
        t_wait += t1 - t0; t_mfma += PR_RW_T() - t1; ++n_units;
      }
#endif
#pragma unroll
      for (int i = 0; i < NI; ++i) pacc[i] = acc[i];
      p_off = off; p_rows = rows;
      stage ^= 1;
    }
    ub = ue;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < NI; ++i) asm volatile("" : "+v"(prv[i]));
#pragma unroll
  for (int i = 0; i < NI; ++i) out_item(i);
#ifdef PR_TIMING_HOOKS
  if (a.stamps && lane == 0) {
    unsigned long long* o = a.stamps + ((size_t)blockIdx.x * 4 + wave) * 8;
    o[0] = t_begin; o[1] = __builtin_amdgcn_s_memrealtime(); o[2] = t_w; o[3] = t_wait; o[4] = 0; o[5] = t_mfma; o[6] = n_units;
    o[7] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
  }
#endif
#endif
}

}  // namespace

bool conv_regw_f32_fits(const ConvProblem& p) {
  return p.precision == 0 && p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && !p.x2 && !p.w3 && p.groups >= 1 &&
         (p.groups == 1 || (!p.bias && !p.res && !p.relu)) && p.splitk == 1 && (p.Cin == 128 || p.Cin == 256) &&
         p.Cout % 64 == 0 && p.M() > 0;
}

int conv_regw_f32_launch(const ConvProblem& p, hipStream_t stream) {
  PR_REQUIRE(conv_regw_f32_fits(p), "conv_regw: fp32 1x1 / stride 1, one source, Cin 128 or 256, Cout %% 64 == 0, plain stores when grouped (got Cin %d, Cout %d)",
             p.Cin, p.Cout);
  PR_REQUIRE(p.x && p.w && p.y, "conv_regw: null tensor");
  const size_t xb = (size_t)p.groups * p.M() * p.Cin * 4, yb = (size_t)p.groups * p.M() * p.Cout * 4;
  PR_REQUIRE(xb < (1ull << 31) && yb < (1ull << 31), "conv_regw: tensor too large for one launch");
  RArgs a;
  a.x = p.x; a.w = p.w; a.bias = p.bias; a.res = p.res; a.y = p.y;
  a.x_bytes = (unsigned)xb; a.y_bytes = (unsigned)yb; a.M = p.M(); a.N = p.Cout; a.relu = p.relu;
  // (T, NB): 16-pixel tiles and 64-channel blocks per unit.  MEASURED (profiles/r05_experiments.txt section 2, B = 64, same
  // box): plain layers T = 1, NB = 2 -- layer2's conv3 71.0 -> 67.4 us, layer3's conv3 and 256 -> 128 level (63.3 / 110.7), three
  // batches in flight +0.35 %; T = 2, NB = 2 halves the units per workgroup and loses on layer3 (63.4 -> 66.7 us: 6 or 7 units
  // per workgroup instead of 12 or 13), NB = 4 at K = 128 needs 308 registers with T = 2 (84.7 us) and is level with NB = 2 at
  // T = 1.  The 36 GEMMs of a Winograd layer keep T = 2, NB = 1: NB = 2 costs layer3's 72 -> 77 us (T = 1) / 82 us (T = 2)
  // with one batch in flight and is level with three.  POSERISK_REGW_T / _NB (plain) and POSERISK_REGW_WT / _WNB (Winograd
  // GEMMs) move them for A/B runs.  Never chosen from the batch, and every (T, NB) gives the same bits anyway.
  const int nblk64 = p.Cout / 64;
  const bool grouped = p.groups > 1;       // the 36 GEMMs of a Winograd layer: their own pair of knobs
  int NB = grouped ? (p.tune.regw_wnb > 0 ? p.tune.regw_wnb : 1) : (p.tune.regw_nb > 0 ? p.tune.regw_nb : 2);
  while (NB > 1 && (nblk64 % NB != 0 || (p.Cin == 256 && NB > 2))) NB >>= 1;
  int T = grouped ? (p.tune.regw_wt > 0 ? p.tune.regw_wt : 2) : (p.tune.regw_t > 0 ? p.tune.regw_t : 1);
  if (T != 1 && T != 2) T = 2;
  const int px = 16 * T;
  a.pp = ceil_div(a.M, px);
  a.nblk = nblk64 / NB;
  a.units = a.pp * a.nblk * p.groups;
  int cus = 256;
  PR_TRY(current_device_cus(&cus));
  const size_t lds = (size_t)2 * px * p.Cin * 4;
  // two workgroups per CU (64 KB of LDS each at K = 256, T = 2), every one with an equal share of the units
  const int grid = std::min(a.units, p.tune.regw_per_cu * cus);
  a.stamps = nullptr;
  a.exp = 0;
#ifdef PR_TIMING_HOOKS
  if (const char* e = getenv("POSERISK_REGW_EXP")) a.exp = atoi(e);
  // timing builds: the 20th launch records, per wave, s_memrealtime (100 MHz) at entry / exit and the summed time spent on the
  // weight loads, at the unit's wait + barrier, in the epilogue + residual requests, and in the MFMA loop; units; HW_ID
  static unsigned long long* stamp_buf = nullptr;
  static int stamp_calls = 0;
  const char* stamp_path = getenv("POSERISK_REGW_STAMPS");
  const bool stamp_now = stamp_path && ++stamp_calls == 20;
  if (stamp_now) {
    if (!stamp_buf) PR_HIP(hipMalloc(&stamp_buf, (size_t)grid * 4 * 8 * 8));
    PR_HIP(hipMemsetAsync(stamp_buf, 0, (size_t)grid * 4 * 8 * 8, stream));
    a.stamps = stamp_buf;
  }
  struct StampDump {
    bool on; const char* path; int grid; hipStream_t s; unsigned long long* buf;
    ~StampDump() {
      if (!on) return;
      std::vector<unsigned long long> host((size_t)grid * 4 * 8);
      (void)hipStreamSynchronize(s);
      (void)hipMemcpy(host.data(), buf, host.size() * 8, hipMemcpyDeviceToHost);
      if (FILE* fo = fopen(path, "wb")) { fwrite(host.data(), 8, host.size(), fo); fclose(fo); }
    }
  } stamp_dump{stamp_now, stamp_path, grid, stream, stamp_buf};
#endif
  auto go = [&](auto kern, std::atomic<uint64_t>& done) -> int {
    PR_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, done));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a);
    return PR_OK;
  };
  static std::atomic<uint64_t> done[2][2][3] = {};
  const int ki = p.Cin == 256, ti = T - 1, ni = NB == 4 ? 2 : NB - 1;
#define PR_REGW_CASE(KK, TT, NN) \
  if (p.Cin == KK && T == TT && NB == NN) PR_TRY(go(conv1x1_regw_f32<KK, TT, NN>, done[ki][ti][ni]))
  PR_REGW_CASE(128, 1, 1); else PR_REGW_CASE(128, 1, 2); else PR_REGW_CASE(128, 1, 4);
  else PR_REGW_CASE(128, 2, 1); else PR_REGW_CASE(128, 2, 2); else PR_REGW_CASE(128, 2, 4);
  else PR_REGW_CASE(256, 1, 1); else PR_REGW_CASE(256, 1, 2);
  else PR_REGW_CASE(256, 2, 1); else PR_REGW_CASE(256, 2, 2);
  else { set_error("conv_regw: no kernel for K %d, T %d, NB %d", p.Cin, T, NB); return PR_ERR_INVALID; }
#undef PR_REGW_CASE
  return check_launch("conv1x1_regw_f32");
}

}  // namespace pr
