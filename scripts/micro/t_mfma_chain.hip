// Do v_mfma_f32_32x32x2_f32 and v_mfma_f32_16x16x4_f32 accumulate k in the same order with the same
// rounding (an ascending-k fmaf chain)?  Prints the number of elements whose bits differ.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int K = 64;
__global__ void k32(const float* A, const float* B, float* D) {  // A[32][K], B[K][32]
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  f32x16 acc = {0};
  for (int i = 0; i < K / 2; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + 2 * i + h], B[(2 * i + h) * 32 + r], acc, 0, 0, 0);
  for (int j = 0; j < 16; ++j) D[((j / 4) * 8 + h * 4 + (j % 4)) * 32 + r] = acc[j];
}
__global__ void k16(const float* A, const float* B, float* D) {  // top-left 16x16 block
  const int l = threadIdx.x, r = l & 15, g = l >> 4;
  f32x4 acc = {0};
  for (int i = 0; i < K / 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r * K + 4 * i + g], B[(4 * i + g) * 32 + r], acc, 0, 0, 0);
  for (int j = 0; j < 4; ++j) D[(4 * g + j) * 32 + r] = acc[j];
}
int main() {
  static float A[32 * K], B[K * 32], D32[1024], D16[1024];
  srand(7);
  for (auto& v : A) v = ((rand() % 2001) - 1000) / 1000.f * ldexpf(1.f, rand() % 12 - 6);
  for (auto& v : B) v = ((rand() % 2001) - 1000) / 1000.f * ldexpf(1.f, rand() % 12 - 6);
  float *a, *b, *d;
  hipMalloc(&a, sizeof A); hipMalloc(&b, sizeof B); hipMalloc(&d, 4096);
  hipMemcpy(a, A, sizeof A, hipMemcpyHostToDevice); hipMemcpy(b, B, sizeof B, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, a, b, d); hipMemcpy(D32, d, 4096, hipMemcpyDeviceToHost);
  hipMemset(d, 0, 4096);
  hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, a, b, d); hipMemcpy(D16, d, 4096, hipMemcpyDeviceToHost);
  int d_32_16 = 0, d_32_fma = 0, d_16_fma = 0, d_32_mul = 0;
  for (int r = 0; r < 16; ++r)
    for (int c = 0; c < 16; ++c) {
      float f = 0.f, m = 0.f;
      for (int k = 0; k < K; ++k) { f = fmaf(A[r * K + k], B[k * 32 + c], f); volatile float p = A[r * K + k] * B[k * 32 + c]; m = m + p; }
      d_32_16 += memcmp(&D32[r * 32 + c], &D16[r * 32 + c], 4) != 0;
      d_32_fma += memcmp(&D32[r * 32 + c], &f, 4) != 0;
      d_16_fma += memcmp(&D16[r * 32 + c], &f, 4) != 0;
      d_32_mul += memcmp(&D32[r * 32 + c], &m, 4) != 0;
    }
  printf("of 256 elements: 32x32x2 vs 16x16x4 differ %d ; 32x32x2 vs fmaf chain %d ; 16x16x4 vs fmaf chain %d ; 32x32x2 vs mul+add chain %d\n",
         d_32_16, d_32_fma, d_16_fma, d_32_mul);
  return 0;
}
