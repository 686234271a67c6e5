// fp32 1x1 convolution with the WEIGHTS RESIDENT IN REGISTERS (K = 128 or 256): y = act(x W^T + bias + res).
// The encoder's short-K 1x1 layers (layer1's conv1 256 -> 64, layer2's conv3 128 -> 512, layer3's conv3 256 -> 1024; SPIN
// models/hmr.py Bottleneck, call site lib/core/base.py:220).
//
// Why: on the 64x64 tile kernel these layers are bound by the operand fetch -- 16 KB of L2 -> LDS per 1 024 MFMA cycles
// per workgroup, half of it weights (profiles/r04_experiments.txt).  A wave's 16 output channels x K weights
// are K / 4 values per lane: they are loaded once per workgroup and stay, only the activations stream (half the bytes, a
// quarter of the LDS-DMA instructions per MFMA).  Same construction as the fused stem (stem_pool_f32.hip).
//
//   work unit = (block of 64 output channels, 32 consecutive pixels); a persistent workgroup of 4 waves takes a contiguous
//     run of units of ONE channel block where it can (the weights are reloaded when the block changes).
//   wave w: channels 64 nb + 16 w .. + 15, both 16-pixel tiles of the unit: two independent accumulators on
//     v_mfma_f32_16x16x4_f32 (a single dependent chain of that shape issues at 40 of 32 cycles).
//   B operand: lane (n, g) holds W[c][16 q + 4 g + j] for q < K / 16, j < 4 -- MFMA (q, j) sums k = 16 q + 4 g + j over g.
//   A operand: the unit's 32 pixel rows (K floats each) sit in one of two LDS stages, 16-byte chunk c of row r in slot
//     c ^ (r & 15) (the LDS-DMA applies the XOR on its SOURCE address), so that the 16 lanes of a tile read 16 different
//     slots: lane (m, g) reads chunk 4 q + g of row m as one ds_read_b128 per four MFMAs, placed between the MFMAs; the DMA
//     of the next unit's rows goes between them too.  One barrier per unit.
//   Epilogue from the accumulators (which start at the bias): a lane holds 4 consecutive pixels of one channel; residual,
//     ReLU, 4-byte buffer stores that the 16 lanes of a pixel make 64 contiguous bytes -- one output per MFMA slot of the
//     NEXT unit, the residual requested a unit ahead the same way.
// The k order inside an output's fmaf chain is (q, j, g) -- fixed, so a frame's bits do not depend on its batch; it is not
// the tile kernel's order.
#include "conv_igemm.h"

namespace pr {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __attribute__((address_space(3))) void lds_void;
constexpr unsigned kOOB = 0x80000000u;
constexpr int kPx = 32;               // pixels per unit

struct RArgs {
  const float* x;       // [M][K]
  const float* w;       // [N][K]
  const float* bias;    // [N] or nullptr
  const float* res;     // [M][N] or nullptr
  float* y;             // [M][N]
  unsigned x_bytes, y_bytes;
  int M, N, relu;
  int units, pp;        // pp = pixel groups (ceil(M / 32)); unit u = nb * pp + group
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

template <int K>
__global__ __launch_bounds__(256) void conv1x1_regw_f32(const RArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int ROW = K * 4;                    // bytes of a pixel row
  constexpr int STAGE = kPx * ROW;              // 16 / 32 KB
  constexpr int NQ = K / 16;                    // MFMA quads
  constexpr int NDMA = STAGE / 1024 / 4;        // LDS-DMA instructions per wave and unit (4 / 8)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, g = lane >> 4;

  // this workgroup's run of units
  const long G = gridDim.x;
  const int u0 = (int)((long)blockIdx.x * a.units / G), u1 = (int)((long)(blockIdx.x + 1) * a.units / G);
  if (u0 >= u1) return;

  const auto xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  // DMA piece i of this wave (i < NDMA): bytes [(wave * NDMA + i) * 1024, + 1024) of the stage; lane covers 16 of them
  auto issue_unit = [&](int u, int stage) {
    const int px0 = (u % a.pp) * kPx;
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
      const int o = (wave * NDMA + i) * 1024 + lane * 16;
      const int r = o / ROW, s = (o % ROW) >> 4;
      const int c = s ^ (r & 15);
      const unsigned vo = (px0 + r < a.M) ? (unsigned)((px0 + r) * ROW + c * 16) : kOOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, (lds_void*)(smem + stage * STAGE + (wave * NDMA + i) * 1024), 16, vo, 0, 0, 0);
    }
  };
  auto issue_piece = [&](int u, int stage, int i) {
    const int px0 = (u % a.pp) * kPx;
    const int o = (wave * NDMA + i) * 1024 + lane * 16;
    const int r = o / ROW, s = (o % ROW) >> 4;
    const int c = s ^ (r & 15);
    const unsigned vo = (px0 + r < a.M) ? (unsigned)((px0 + r) * ROW + c * 16) : kOOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, (lds_void*)(smem + stage * STAGE + (wave * NDMA + i) * 1024), 16, vo, 0, 0, 0);
  };

  // A read addresses: row m of a tile, chunk 4 q + g -> slot (4 q + g) ^ m = (q >> 2) * 16 + ((4 (q & 3) + g) ^ m)
  int abase[4];
#pragma unroll
  for (int qq = 0; qq < 4; ++qq) abase[qq] = m * ROW + (((4 * qq + g) ^ m) << 4);

  float b[NQ][4];
  [[maybe_unused]] unsigned long long t_begin = PR_RW_T(), t_w = 0, t_wait = 0, t_mfma = 0, n_units = 0;
  issue_unit(u0, 0);
  int stage = 0;
  // The epilogue of unit u - 1 and the residual requests of unit u ride BETWEEN the MFMAs of unit u, one output (an add, a
  // max, a 4-byte buffer store) or one 4-byte buffer load per slot: a wave issues in order, so the same sixteen memory
  // instructions in a block behind the barrier kept its MFMAs from issuing for 0.9 us of a 4.3 us unit (per-wave stamps,
  // profiles/r04_experiments.txt section 5).  Byte offsets go through the buffer's range check: a pixel >= M is an offset
  // >= M * N * 4 and is dropped / reads zero, and so is everything of the "unit before the first" (offset 2^31).
  const auto ysrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)a.y_bytes, 0x00020000);
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res ? a.res : a.y), 0, (int)a.y_bytes, 0x00020000);
  const bool has_res = a.res != nullptr;
  const unsigned n4 = (unsigned)a.N * 4u;
  f32x4 pacc0 = {0.f, 0.f, 0.f, 0.f}, pacc1 = {0.f, 0.f, 0.f, 0.f};
  float prv[2][4] = {};                 // residual of the unit whose accumulators sit in pacc: loaded a unit ahead
  unsigned p_lane = 0, p_row = kOOB;    // its lane offset (4 g rows + channel) and its first pixel's row offset
  auto out_item = [&](int t, int e) {   // pacc[t][e] = pixel p_px0 + 16 t + 4 g + e of channel p_ch
    float v = t ? pacc1[e] : pacc0[e];
    if (has_res) v += prv[t][e];
    if (a.relu) v = fmaxf(v, 0.f);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ysrc, p_lane + (p_row + (unsigned)(16 * t + e) * n4), 0, 0);
  };
  for (int ub = u0; ub < u1;) {
    // one channel block at a time: its weights are loaded here, OUTSIDE the unit loop (K / 16 16-byte loads per lane) and
    // waited for at once -- loaded under a condition inside the unit loop, the compiler's vmcnt bookkeeping made every
    // unit's MFMA loop wait for ALL outstanding vector-memory operations, the next unit's DMA included
    const int nb = ub / a.pp;
    const int ue = min(u1, (nb + 1) * a.pp);
    const int ch = nb * 64 + 16 * wave + m;
    const unsigned lane_off = (unsigned)(4 * g) * n4 + (unsigned)ch * 4u;
    [[maybe_unused]] const unsigned long long tw0 = PR_RW_T();
    {
      const f32x4* wr = reinterpret_cast<const f32x4*>(a.w + (size_t)ch * K) + g;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const f32x4 v = wr[q * 4];
        b[q][0] = v[0]; b[q][1] = v[1]; b[q][2] = v[2]; b[q][3] = v[3];
      }
    }
    float bias = a.bias ? a.bias[ch] : 0.f;
    // the compiler's own waits for these loads must happen HERE (an empty asm that "uses" the registers): left to the first
    // use, they sit inside the unit loop as vmcnt(15) .. vmcnt(0) in front of every quad of EVERY unit and drain the next
    // unit's DMA and the residual loads with them
#pragma unroll
    for (int q = 0; q < NQ; ++q) asm volatile("" : "+v"(b[q][0]), "+v"(b[q][1]), "+v"(b[q][2]), "+v"(b[q][3]));
    asm volatile("" : "+v"(bias));
#ifdef PR_TIMING_HOOKS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t_w += PR_RW_T() - tw0;
#endif
    for (int u = ub; u < ue; ++u) {
      const unsigned row = (unsigned)((u % a.pp) * kPx) * n4;
      [[maybe_unused]] const unsigned long long t0 = PR_RW_T();
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // own DMA pieces of this unit have landed (issued a unit ago)
      // ... and so has the previous unit's residual: tell the compiler here (see the weights above), or it waits for
      // everything outstanding in front of each of the eight adds
      asm volatile("" : "+v"(prv[0][0]), "+v"(prv[0][1]), "+v"(prv[0][2]), "+v"(prv[0][3]), "+v"(prv[1][0]), "+v"(prv[1][1]),
                        "+v"(prv[1][2]), "+v"(prv[1][3]));
      if (!PR_RW_EXP(8)) __builtin_amdgcn_s_barrier();                // everyone's have; everyone has finished the other stage
      asm volatile("" ::: "memory");
      [[maybe_unused]] const unsigned long long t1 = PR_RW_T();
      const char* st = smem + stage * STAGE;
      f32x4 acc0 = {bias, bias, bias, bias}, acc1 = acc0;             // a lane's four values are four pixels of ONE channel
      f32x4 av[2][2];
      av[0][0] = *reinterpret_cast<const f32x4*>(st + abase[0]);
      av[0][1] = *reinterpret_cast<const f32x4*>(st + 16 * ROW + abase[0]);
      const bool more = u + 1 < u1;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int cur = q & 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur][0][j], b[q][j], acc0, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          // behind the first MFMA of a pair: output q of the previous unit (j == 1), then the request for the residual that
          // will be added to output q of THIS unit a unit from now (j == 3: into the register the output just read)
          if (q < 8 && j == 1) {
            if (!PR_RW_EXP(1)) out_item(q >> 2, q & 3);
            __builtin_amdgcn_sched_barrier(0);
          } else if (q < 8 && j == 3 && has_res && !PR_RW_EXP(4)) {
            prv[q >> 2][q & 3] = __builtin_bit_cast(
                float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane_off + (row + (unsigned)(16 * (q >> 2) + (q & 3)) * n4), 0, 0));
            __builtin_amdgcn_sched_barrier(0);
          }
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur][1][j], b[q][j], acc1, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          // behind the second: the next quad's two reads, and one DMA piece of the next unit per quad while there are some
          if (j < 2 && q + 1 < NQ && !PR_RW_EXP(16)) {
            av[cur ^ 1][j] = *reinterpret_cast<const f32x4*>(st + j * 16 * ROW + ((q + 1) >> 2) * 256 + abase[(q + 1) & 3]);
            __builtin_amdgcn_sched_barrier(0);
          } else if (j == 2 && q < NDMA) {
            if (more && !PR_RW_EXP(2)) issue_piece(u + 1, stage ^ 1, q);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
#ifdef PR_TIMING_HOOKS
      if (a.stamps) {
        asm volatile("s_nop 0" : "+v"(acc0), "+v"(acc1));
        t_wait += t1 - t0; t_mfma += PR_RW_T() - t1; ++n_units;
      }
#endif
      pacc0 = acc0; pacc1 = acc1;
      p_lane = lane_off; p_row = row;
      stage ^= 1;
    }
    ub = ue;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("" : "+v"(prv[0][0]), "+v"(prv[0][1]), "+v"(prv[0][2]), "+v"(prv[0][3]), "+v"(prv[1][0]), "+v"(prv[1][1]),
                    "+v"(prv[1][2]), "+v"(prv[1][3]));
#pragma unroll
  for (int i = 0; i < 8; ++i) out_item(i >> 2, i & 3);
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
  return p.precision == 0 && p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && !p.x2 && !p.w3 && p.groups == 1 &&
         p.splitk == 1 && (p.Cin == 128 || p.Cin == 256) && p.Cout % 64 == 0 && p.M() > 0;
}

int conv_regw_f32_launch(const ConvProblem& p, hipStream_t stream) {
  PR_REQUIRE(conv_regw_f32_fits(p), "conv_regw: fp32 1x1 / stride 1, one source, Cin 128 or 256, Cout %% 64 == 0 (got Cin %d, Cout %d)",
             p.Cin, p.Cout);
  PR_REQUIRE(p.x && p.w && p.y, "conv_regw: null tensor");
  const size_t xb = (size_t)p.M() * p.Cin * 4;
  PR_REQUIRE(xb < (1ull << 31) && (size_t)p.M() * p.Cout * 4 < (1ull << 31), "conv_regw: tensor too large for one launch");
  RArgs a;
  a.x = p.x; a.w = p.w; a.bias = p.bias; a.res = p.res; a.y = p.y;
  a.x_bytes = (unsigned)xb; a.y_bytes = (unsigned)((size_t)p.M() * p.Cout * 4); a.M = p.M(); a.N = p.Cout; a.relu = p.relu;
  a.pp = ceil_div(a.M, kPx);
  a.units = a.pp * (p.Cout / 64);
  int cus = 256;
  PR_TRY(current_device_cus(&cus));
  const size_t lds = (size_t)2 * kPx * p.Cin * 4;
  // two workgroups per CU (64 KB of LDS each at K = 256), every one with an equal share of the units
  const int grid = std::min(a.units, 2 * cus);
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
  static std::atomic<uint64_t> done128{0}, done256{0};
  if (p.Cin == 128) {
    PR_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(conv1x1_regw_f32<128>), lds, done128));
    hipLaunchKernelGGL(conv1x1_regw_f32<128>, dim3(grid), dim3(256), lds, stream, a);
  } else {
    PR_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(conv1x1_regw_f32<256>), lds, done256));
    hipLaunchKernelGGL(conv1x1_regw_f32<256>, dim3(grid), dim3(256), lds, stream, a);
  }
  return check_launch("conv1x1_regw_f32");
}

}  // namespace pr
