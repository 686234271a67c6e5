// How fast does one SIMD issue v_mfma_f32_32x32x2_f32 from W co-resident waves, each running ONE dependent accumulator
// chain (the fp32 conv kernel's wave: one 32x32 tile) or TWO independent ones?  And with a barrier + LDS reads every 16
// MFMAs (the kernel's K-step skeleton without any global memory traffic)?
//   hipcc --offload-arch=gfx950 -O3 -o t_mfma_rate.bin t_mfma_rate.hip && ./t_mfma_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

// SKEL 3: the fragments of iteration it+1 are read (into a second register set) before the MFMAs of iteration it, as a
// kernel with a ring of three LDS stages could do: the LDS latency runs under the wave's own MFMAs.
// SKEL 4: as 2, plus 4 LDS-DMA instructions (buffer_load ... lds, 1 KB each, L2-resident source) per iteration and wave and
// the wait for the previous iteration's ones in front of the barrier -- the conv kernel's whole K-step.
// SKEL 5: 3 + the DMA of SKEL 4 (two stages ahead: vmcnt(4)).
template <int ACC, int SKEL>
__global__ __launch_bounds__(256) void k(float* out, int iters, const float* in) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  f32x16 acc[ACC];
  for (int a = 0; a < ACC; ++a) for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  float av = in[threadIdx.x], bv = in[threadIdx.x + 256];
  f32x4 af[4], bf[4];
  for (int i = 0; i < 4; ++i) { af[i] = f32x4{av, av, av, av}; bf[i] = f32x4{bv, bv, bv, bv}; }
  if constexpr (SKEL >= 3) {
    const int frow = lane & 31, fh = lane >> 5, fsw = (frow >> 1) & 7;
    const int wave = threadIdx.x >> 6;
    const auto src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, 1 << 20, 0x00020000);
    const unsigned voff = (unsigned)((blockIdx.x & 63) * 16384 + wave * 4096 + lane * 16);
    typedef __attribute__((address_space(3))) void lds_void;
    auto dma = [&](int stage) {
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(src, (lds_void*)(smem + stage * 16384 + (wave * 4 + i) * 1024), 16, voff, i * 1024, 0, 0);
    };
    auto rd = [&](int stage, f32x4* a4, f32x4* b4) {
      const char* base = smem + stage * 16384 + frow * 128;
      for (int i = 0; i < 4; ++i) {
        a4[i] = *reinterpret_cast<const f32x4*>(base + (((2 * i + fh) ^ fsw) << 4));
        b4[i] = *reinterpret_cast<const f32x4*>(base + 8192 + (((2 * i + fh) ^ fsw) << 4));
      }
    };
    auto mm = [&](const f32x4* a4, const f32x4* b4) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int a = 0; a < ACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i][j], b4[i][j], acc[a], 0, 0, 0);
    };
    if constexpr (SKEL >= 6 && SKEL <= 13) {
      // one LDS stage ahead in registers; the reads (and, SKEL 7, the DMA instructions) sit BETWEEN the MFMAs
      f32x4 af2[4], bf2[4];
      constexpr bool DMA = SKEL >= 7, DMA_FIRST = SKEL >= 8, RING3 = SKEL == 9;
      constexpr int SPREAD = SKEL >= 10 ? SKEL - 9 : 0;      // waves' DMA slots staggered by this many MFMA slots per wave
      if (DMA) { dma(0); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      __builtin_amdgcn_s_barrier();
      rd(0, af, bf);
      if (DMA) { dma(1); }
      if (RING3) { dma(2); }
      int st = 0;   // RING3: the LDS stage holding iteration `it`
      auto step = [&](auto wv, int it, const f32x4* ca, const f32x4* cb, f32x4* na, f32x4* nb) {
        if (DMA) { if (RING3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int s1 = RING3 ? (st + 1 == 3 ? 0 : st + 1) : ((it + 1) & 1);     // stage to read: iteration it+1
        const int sd = RING3 ? st : (it & 1);                                    // stage to refill: this iteration's (in registers)
        const char* base = smem + s1 * 16384 + frow * 128;
        const int r0 = DMA_FIRST ? 4 : 0, d0 = DMA_FIRST ? 0 : 8;
        // exact program order (sched_barrier(0) between all of them): MFMA, read, MFMA, read, ... MFMA, DMA, ...
#pragma unroll
        for (int n = 0; n < 16; ++n) {
          const int i = n >> 2, j = n & 3;
#pragma unroll
          for (int a = 0; a < ACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[i][j], cb[i][j], acc[a], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (SPREAD > 0) {
            // wave W issues its 4 DMA pieces behind MFMAs W * SPREAD .. W * SPREAD + 3; the 8 reads take the first 8 slots
            // that are not its DMA slots (W is a compile-time constant of this copy of the step)
            constexpr int W = decltype(wv)::value;
            const int d = n - W * SPREAD;
            if (d >= 0 && d < 4) {
              __builtin_amdgcn_raw_ptr_buffer_load_lds(src, (lds_void*)(smem + sd * 16384 + (W * 4 + d) * 1024), 16, voff, d * 1024, 0, 0);
            } else {
              const int m = d < 0 ? n : n - 4;
              if (m < 8) {
                if (m & 1) nb[m >> 1] = *reinterpret_cast<const f32x4*>(base + 8192 + (((2 * (m >> 1) + fh) ^ fsw) << 4));
                else na[m >> 1] = *reinterpret_cast<const f32x4*>(base + (((2 * (m >> 1) + fh) ^ fsw) << 4));
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          } else
          if (n >= r0 && n < r0 + 8) {
            const int m = n - r0;
            if (m & 1) nb[m >> 1] = *reinterpret_cast<const f32x4*>(base + 8192 + (((2 * (m >> 1) + fh) ^ fsw) << 4));
            else na[m >> 1] = *reinterpret_cast<const f32x4*>(base + (((2 * (m >> 1) + fh) ^ fsw) << 4));
            __builtin_amdgcn_sched_barrier(0);
          } else if (DMA && n >= d0 && n < d0 + 4) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(src, (lds_void*)(smem + sd * 16384 + (wave * 4 + (n - d0)) * 1024), 16, voff,
                                                     (n - d0) * 1024, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        st = s1;
      };
      auto loop = [&](auto wv) {
        for (int it = 0; it < iters; it += 2) {
          step(wv, it, af, bf, af2, bf2);
          step(wv, it + 1, af2, bf2, af, bf);
        }
      };
      const int wu = __builtin_amdgcn_readfirstlane(wave);
      if (SPREAD == 0 || wu == 0) loop(std::integral_constant<int, 0>{});
      else if (wu == 1) loop(std::integral_constant<int, 1>{});
      else if (wu == 2) loop(std::integral_constant<int, 2>{});
      else loop(std::integral_constant<int, 3>{});
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if constexpr (SKEL == 4) {
      dma(0);
      for (int it = 0; it < iters; ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        dma((it + 1) & 1);
        rd(it & 1, af, bf);
        __builtin_amdgcn_sched_barrier(0);
        mm(af, bf);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      f32x4 af2[4], bf2[4];
      int st = 0;
      if (SKEL == 5) { dma(0); dma(1); }
      if (SKEL == 5) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      rd(0, af, bf);
      for (int it = 0; it < iters; it += 2) {
        // even step: multiply (af, bf), read the next step's fragments into (af2, bf2)
        if (SKEL == 5) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        int s1 = st + 1 == 3 ? 0 : st + 1, s2 = s1 + 1 == 3 ? 0 : s1 + 1;
        if (SKEL == 5) dma(s2);
        rd(s1, af2, bf2);
        __builtin_amdgcn_sched_barrier(0);
        mm(af, bf);
        __builtin_amdgcn_sched_barrier(0);
        st = s1;
        if (SKEL == 5) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        s1 = st + 1 == 3 ? 0 : st + 1; s2 = s1 + 1 == 3 ? 0 : s1 + 1;
        if (SKEL == 5) dma(s2);
        rd(s1, af, bf);
        __builtin_amdgcn_sched_barrier(0);
        mm(af2, bf2);
        __builtin_amdgcn_sched_barrier(0);
        st = s1;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  } else
  for (int it = 0; it < iters; ++it) {
    if (SKEL >= 1) {
      __builtin_amdgcn_s_barrier();
      if (SKEL >= 2) {
        // the conv kernel's conflict-free fragment read: row lane&31, logical chunk 2i + (lane>>5), XOR-swizzled by the row
        const int frow = lane & 31, fh = lane >> 5, fsw = (frow >> 1) & 7;
        const char* base = smem + (it & 1) * 16384 + frow * 128;
        for (int i = 0; i < 4; ++i) {
          af[i] = *reinterpret_cast<const f32x4*>(base + (((2 * i + fh) ^ fsw) << 4));
          bf[i] = *reinterpret_cast<const f32x4*>(base + 8192 + (((2 * i + fh) ^ fsw) << 4));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int a = 0; a < ACC; ++a)
          acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][j], bf[i][j], acc[a], 0, 0, 0);
  }
  float s = 0.f;
  for (int a = 0; a < ACC; ++a) for (int e = 0; e < 16; ++e) s += acc[a][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// Producer / consumer split: waves 0-3 run skeleton 6 (barrier, MFMAs with the next iteration's 8 reads between them), NP extra
// waves issue the workgroup's 16 LDS-DMA instructions per iteration (16 / NP each, in a burst behind the barrier) and wait
// for them in front of the barrier, NST - 1 iterations ahead.
template <int NP, int NST>
__global__ __launch_bounds__(256 + 64 * NP) void kp(float* out, int iters, const float* in) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void lds_void;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int IPW = 16 / NP;
  if (wave >= 4) {
    const int pw = wave - 4;
    const auto src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, 1 << 20, 0x00020000);
    const unsigned voff = (unsigned)((blockIdx.x & 63) * 16384 + lane * 16);
    auto dma = [&](int stage) {
#pragma unroll
      for (int i = 0; i < IPW; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(src, (lds_void*)(smem + stage * 16384 + (pw * IPW + i) * 1024), 16, voff, (pw * IPW + i) * 1024, 0, 0);
    };
    for (int s = 0; s < NST; ++s) dma(s);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 1) * IPW) : "memory");
    __builtin_amdgcn_s_barrier();
    int buf = 0;
    for (int it = 0; it < iters; ++it) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * IPW) : "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      dma(buf);
      buf = buf + 1 == NST ? 0 : buf + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const int frow = lane & 31, fh = lane >> 5, fsw = (frow >> 1) & 7;
  f32x4 fa[2][4], fb[2][4];
  __builtin_amdgcn_s_barrier();
  for (int i = 0; i < 4; ++i) {
    fa[0][i] = *reinterpret_cast<const f32x4*>(smem + frow * 128 + (((2 * i + fh) ^ fsw) << 4));
    fb[0][i] = *reinterpret_cast<const f32x4*>(smem + 8192 + frow * 128 + (((2 * i + fh) ^ fsw) << 4));
  }
  int nb = 1;
  auto step = [&](const f32x4* ca, const f32x4* cb, f32x4* na, f32x4* nbr) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const char* base = smem + nb * 16384 + frow * 128;
#pragma unroll
    for (int n = 0; n < 16; ++n) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[n >> 2][n & 3], cb[n >> 2][n & 3], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (n < 8) {
        if (n & 1) nbr[n >> 1] = *reinterpret_cast<const f32x4*>(base + 8192 + (((2 * (n >> 1) + fh) ^ fsw) << 4));
        else na[n >> 1] = *reinterpret_cast<const f32x4*>(base + (((2 * (n >> 1) + fh) ^ fsw) << 4));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    nb = nb + 1 == NST ? 0 : nb + 1;
  };
  for (int it = 0; it < iters; it += 2) {
    step(fa[0], fb[0], fa[1], fb[1]);
    step(fa[1], fb[1], fa[0], fb[0]);
  }
  float sacc = 0.f;
  for (int e = 0; e < 16; ++e) sacc += acc[e];
  out[blockIdx.x * 256 + threadIdx.x] = sacc;
}

template <int NP, int NST>
void runp(int wgs_per_cu, int iters, float* out, float* in) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * wgs_per_cu, lds = NST * 16384;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kp<NP, NST>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((kp<NP, NST>), dim3(grid), dim3(256 + 64 * NP), lds, 0, out, iters, in);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("  %d workgroup(s)/CU, %d producer wave(s), %d LDS stages: %8.1f us   = %.3f of 157.3\n", wgs_per_cu, NP, NST, ms * 1e3,
         grid * 4.0 * iters * 16 * 4096 / (ms * 1e-3) / 1e12 / 157.3);
}

// B fragments straight from L2 into registers: per iteration and wave 2 LDS-DMA (A only) + 4 ds_read_b128 (A) + 4
// global_load_dwordx4 of 1 KB contiguous each (B, fragment-order weights), one stage ahead, all BETWEEN the MFMAs.
template <int DUMMY>
__global__ __launch_bounds__(256) void kg(float* out, int iters, const float* in) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void lds_void;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const int frow = lane & 31, fh = lane >> 5, fsw = (frow >> 1) & 7;
  const auto src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, 1 << 20, 0x00020000);
  const unsigned voff = (unsigned)((blockIdx.x & 63) * 16384 + wave * 2048 + lane * 16);
  const f32x4* gb = reinterpret_cast<const f32x4*>(in) + (blockIdx.x & 31) * 2048 + (wave & 1) * 1024 + lane;
  f32x4 fa[2][4], fb[2][4];
  auto dma = [&](int stage, int i) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(src, (lds_void*)(smem + stage * 8192 + (wave * 2 + i) * 1024), 16, voff, i * 1024, 0, 0);
  };
  dma(0, 0); dma(0, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int i = 0; i < 4; ++i) {
    fa[0][i] = *reinterpret_cast<const f32x4*>(smem + frow * 128 + (((2 * i + fh) ^ fsw) << 4));
    fb[0][i] = gb[i * 64];
  }
  dma(1, 0); dma(1, 1);
  auto step = [&](int it, const f32x4* ca, const f32x4* cb, f32x4* na, f32x4* nbr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (also waits for the B loads of this step: issued a step ago)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const char* base = smem + ((it + 1) & 1) * 8192 + frow * 128;
    const f32x4* g = gb + ((it + 1) & 7) * 256;
#pragma unroll
    for (int n = 0; n < 16; ++n) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[n >> 2][n & 3], cb[n >> 2][n & 3], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (n < 2) { dma(it & 1, n); __builtin_amdgcn_sched_barrier(0); }
      else if (n < 6) { nbr[n - 2] = g[(n - 2) * 64]; __builtin_amdgcn_sched_barrier(0); }
      else if (n < 10) { na[n - 6] = *reinterpret_cast<const f32x4*>(base + (((2 * (n - 6) + fh) ^ fsw) << 4)); __builtin_amdgcn_sched_barrier(0); }
    }
  };
  for (int it = 0; it < iters; it += 2) {
    step(it, fa[0], fb[0], fa[1], fb[1]);
    step(it + 1, fa[1], fb[1], fa[0], fb[0]);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float sacc = 0.f;
  for (int e = 0; e < 16; ++e) sacc += acc[e];
  out[blockIdx.x * 256 + threadIdx.x] = sacc;
}

void rung(int wgs_per_cu, int iters, float* out, float* in) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * wgs_per_cu, lds = 16384;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((kg<0>), dim3(grid), dim3(256), lds, 0, out, iters, in);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("  %d workgroup(s)/CU, A by LDS-DMA + LDS reads, B straight from L2 into registers: %8.1f us   = %.3f of 157.3\n", wgs_per_cu, ms * 1e3,
         grid * 4.0 * iters * 16 * 4096 / (ms * 1e-3) / 1e12 / 157.3);
}

template <int ACC, int SKEL>
void run(int wgs_per_cu, int iters, float* out, float* in) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * wgs_per_cu;
  const int lds = SKEL == 3 || SKEL == 5 || SKEL == 9 ? 49152 : 32768;   // 6, 7: two stages
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<ACC, SKEL>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<ACC, SKEL>), dim3(grid), dim3(256), lds, 0, out, iters, in);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = (double)wgs_per_cu * iters * 16 * ACC;       // one wave of every workgroup per SIMD
  const double cyc = mfma_per_simd * 64;                                     // issue cycles at 64 per MFMA
  printf("  %d workgroup(s)/CU, %d chain(s)/wave, skeleton %d: %8.1f us   %.1f TFLOP/s   = %.3f of 157.3   (MFMA cycles/SIMD %.0f -> %.3f GHz-equivalent)\n",
         wgs_per_cu, ACC, SKEL, ms * 1e3, grid * 4.0 * iters * 16 * ACC * 4096 / (ms * 1e-3) / 1e12,
         grid * 4.0 * iters * 16 * ACC * 4096 / (ms * 1e-3) / 1e12 / 157.3, cyc, cyc / (ms * 1e3) / 1e3);
}

int main() {
  float *out, *in;
  hipMalloc(&out, 256 * 8 * 256 * 4); hipMalloc(&in, 1 << 20); hipMemset(in, 0, 1 << 20);
  const int iters = 2048;   // x16 MFMAs per chain
  printf("skeleton 0: MFMAs only; 1: + s_barrier every 16 MFMAs per chain; 2: + 8 ds_read_b128 behind the barrier\n");
  for (int w : {1, 2, 3, 5}) run<1, 0>(w, iters, out, in);
  for (int w : {1, 2, 3}) run<2, 0>(w, iters / 2, out, in);
  for (int w : {1, 2, 3, 5}) run<1, 1>(w, iters, out, in);
  for (int w : {1, 2, 3, 5}) run<1, 2>(w, iters, out, in);
  for (int w : {1, 2, 3}) run<2, 2>(w, iters / 2, out, in);
  printf("skeleton 3: as 2 with the next iteration's fragments read before this one's MFMAs; 4: 2 + 4 LDS-DMA per wave and iteration (the conv kernel's K-step); 5: 3 + that DMA, two stages ahead\n");
  for (int w : {1, 2, 3}) run<1, 3>(w, iters, out, in);
  for (int w : {1, 2, 3, 4}) run<1, 4>(w, iters, out, in);
  for (int w : {1, 2, 3}) run<1, 5>(w, iters, out, in);
  printf("skeleton 6: one stage ahead in registers, its 8 reads BETWEEN this iteration's MFMAs; 7: 6 + the 4 LDS-DMA between MFMAs too (two LDS stages)\n");
  for (int w : {1, 2, 3, 4}) run<1, 6>(w, iters, out, in);
  for (int w : {1, 2, 3, 4}) run<1, 7>(w, iters, out, in);
  printf("skeleton 8: 7 with the DMA between the FIRST four MFMAs (the reads behind); 9: 8 with three LDS stages, DMA two iterations ahead (vmcnt(4))\n");
  for (int w : {1, 2, 3, 4}) run<1, 8>(w, iters, out, in);
  for (int w : {1, 2, 3}) run<1, 9>(w, iters, out, in);
  printf("skeleton 10 / 11 / 12 / 13: 8 with the four waves' DMA slots staggered by 1 / 2 / 3 / 4 MFMA slots per wave\n");
  for (int w : {1, 3}) run<1, 10>(w, iters, out, in);
  for (int w : {1, 3}) run<1, 11>(w, iters, out, in);
  for (int w : {1, 3}) run<1, 12>(w, iters, out, in);
  for (int w : {1, 3}) run<1, 13>(w, iters, out, in);
  for (int w : {1, 2, 3, 4}) rung(w, iters, out, in);
  printf("producer waves issue the DMA; consumers = skeleton 6\n");
  for (int w : {1, 2, 3}) runp<1, 2>(w, iters, out, in);
  for (int w : {1, 2, 3}) runp<2, 2>(w, iters, out, in);
  for (int w : {1, 2, 3}) runp<2, 3>(w, iters, out, in);
  for (int w : {1, 2, 3}) runp<4, 2>(w, iters, out, in);
  for (int w : {1, 2}) runp<4, 3>(w, iters, out, in);
  return 0;
}
