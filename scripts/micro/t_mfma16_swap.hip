// Round 6: a 32x32 output tile on FOUR v_mfma_f32_16x16x32_bf16 per 32 k instead of two v_mfma_f32_32x32x16_bf16 per 32 k,
// then 8 v_permlane32_swap_b32 that put the accumulators back into the 32x32x16 REGISTER layout, so that the kernels'
// epilogues stay as they are (csrc/common.h: Acc32 / acc32_regs).  Checks on exact small integers the lane maps that rests on:
//   16x16x32: lane (j = l & 15, g = l >> 4) holds A[row j][k = 8 g ..], B[k = 8 g ..][col j], D[row 4 g + r][col j];
//   v_permlane32_swap_b32 v0, v1: lanes 32-63 of v0 swap with lanes 0-31 of v1;
// with v0 = tile (rt, ct = 0)[r], v1 = tile (rt, ct = 1)[r]: register e = 8 rt + 4 (which of the two) + r of lane l is
// D[row (e & 3) + 8 (e >> 2) + 4 h][col i] with i = 16 (l >> 5) + (l & 15), h = (l >> 4) & 1 -- the 32x32x16 accumulator's
// registers on relabelled lanes (there: i = l & 31, h = l >> 5); no row permutation anywhere.  On random bf16 data: how far
// the two shapes' sums differ (the order inside an MFMA differs: close, not bit-identical).
// (The BUILTINS __builtin_amdgcn_permlane16_swap / permlane32_swap are miscompiled by hipcc 7.2 in this pattern -- their second
// result is dropped and calls are merged; gpurun_out/r06 notes -- hence the instruction in inline asm with its own wait states.)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
constexpr int K = 128;
__global__ void k32(const unsigned short* A, const unsigned short* B, float* D) {   // A[32][K], Bt[32][K] (column-major B)
  const int l = threadIdx.x, i = l & 31, h = l >> 5;
  f32x16 acc = {0};
  for (int ks = 0; ks < K / 16; ++ks) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(A + i * K + 16 * ks + 8 * h);
    const bf16x8 b = *reinterpret_cast<const bf16x8*>(B + i * K + 16 * ks + 8 * h);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  }
  for (int e = 0; e < 16; ++e) D[l * 16 + e] = acc[e];
}
__global__ void k16(const unsigned short* A, const unsigned short* B, float* D) {
  const int l = threadIdx.x, j = l & 15, g = l >> 4;
  f32x4 t[2][2] = {};
  for (int s = 0; s < K / 32; ++s) {
    bf16x8 a[2], b[2];
    for (int rt = 0; rt < 2; ++rt) a[rt] = *reinterpret_cast<const bf16x8*>(A + (16 * rt + j) * K + 32 * s + 8 * g);
    for (int ct = 0; ct < 2; ++ct) b[ct] = *reinterpret_cast<const bf16x8*>(B + (16 * ct + j) * K + 32 * s + 8 * g);
    for (int rt = 0; rt < 2; ++rt)
      for (int ct = 0; ct < 2; ++ct) t[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[rt], b[ct], t[rt][ct], 0, 0, 0);
  }
  // ONE asm statement per 32x32 tile: every accumulator is an input, so all 16x16x32 MFMAs of the tile have issued before
  // it; the pad in front is the XDL-write -> VALU-read wait (the compiler pads nothing for asm operands)
  float x[8], y[8];
  for (int rt = 0; rt < 2; ++rt)
    for (int r = 0; r < 4; ++r) { x[4 * rt + r] = t[rt][0][r]; y[4 * rt + r] = t[rt][1][r]; }
  asm volatile("s_nop 15\n\ts_nop 3\n\t"
               "v_permlane32_swap_b32 %0, %8\n\tv_permlane32_swap_b32 %1, %9\n\tv_permlane32_swap_b32 %2, %10\n\tv_permlane32_swap_b32 %3, %11\n\t"
               "v_permlane32_swap_b32 %4, %12\n\tv_permlane32_swap_b32 %5, %13\n\tv_permlane32_swap_b32 %6, %14\n\tv_permlane32_swap_b32 %7, %15"
               : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]),
                 "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), "+v"(y[7]));
  f32x16 o;
  for (int rt = 0; rt < 2; ++rt)
    for (int r = 0; r < 4; ++r) { o[8 * rt + r] = x[4 * rt + r]; o[8 * rt + 4 + r] = y[4 * rt + r]; }
  for (int e = 0; e < 16; ++e) D[l * 16 + e] = o[e];
}
static unsigned short bf(float v) { unsigned u; memcpy(&u, &v, 4); return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }
static float fb(unsigned short b) { unsigned u = (unsigned)b << 16; float v; memcpy(&v, &u, 4); return v; }
int main() {
  static unsigned short A[32 * K], B[32 * K];
  static float D32[1024], D16[1024];
  unsigned short *a, *b; float* d;
  (void)hipMalloc(&a, sizeof A); (void)hipMalloc(&b, sizeof B); (void)hipMalloc(&d, 4096);
  for (int pass = 0; pass < 2; ++pass) {
    srand(7 + pass);
    for (auto& v : A) v = bf(pass ? ((rand() % 2001) - 1000) / 1000.f * ldexpf(1.f, rand() % 8 - 4) : (float)(rand() % 17 - 8));
    for (auto& v : B) v = bf(pass ? ((rand() % 2001) - 1000) / 1000.f * ldexpf(1.f, rand() % 8 - 4) : (float)(rand() % 13 - 6));
    (void)hipMemcpy(a, A, sizeof A, hipMemcpyHostToDevice); (void)hipMemcpy(b, B, sizeof B, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, a, b, d); (void)hipMemcpy(D32, d, 4096, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, a, b, d); (void)hipMemcpy(D16, d, 4096, hipMemcpyDeviceToHost);
    int layout_bad = 0, differ = 0; double maxrel = 0, max32 = 0, max16 = 0;
    for (int l = 0; l < 64; ++l)
      for (int e = 0; e < 16; ++e) {
        const int i = 16 * (l >> 5) + (l & 15), h = (l >> 4) & 1, row = (e & 3) + 8 * (e >> 2) + 4 * h;   // the 16-path's lane labels
        const int l32 = 32 * h + i;                                                                      // the same (i, h) in the 32x32x16 kernel
        double ref = 0, mag = 0;
        for (int k = 0; k < K; ++k) { ref += (double)fb(A[row * K + k]) * fb(B[i * K + k]); mag += fabs((double)fb(A[row * K + k]) * fb(B[i * K + k])); }
        layout_bad += fabs(D16[l * 16 + e] - ref) > 1e-5 * mag + 1e-6;
        differ += memcmp(&D16[l * 16 + e], &D32[l32 * 16 + e], 4) != 0;
        maxrel = fmax(maxrel, fabs((double)D16[l * 16 + e] - D32[l32 * 16 + e]) / mag);
        max32 = fmax(max32, fabs(D32[l32 * 16 + e] - ref) / mag); max16 = fmax(max16, fabs(D16[l * 16 + e] - ref) / mag);
      }
    printf("%s data: elements in the wrong place %d of 1024; bits differing from the 32x32x16 form %d; max |d| / sum|a b| %.2e (32x32x16 vs fp64 %.2e, 16x16x32 vs fp64 %.2e)\n",
           pass ? "random bf16" : "exact integer", layout_bad, differ, maxrel, max32, max16);
  }
  return 0;
}
