// Calibration kernels for rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 (known byte counts, the access
// shapes the conv kernel uses): 16-B/lane LDS-DMA reads, 4-B/lane row-segment stores, 4-B/lane loads.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void lds_void;
using f32x4 = __attribute__((ext_vector_type(4))) float;

__global__ void read_dma16(const float* src, float* out, size_t n_floats) {   // every byte once via LDS-DMA
  extern __shared__ __attribute__((aligned(16))) char smem[];
#if defined(__HIP_DEVICE_COMPILE__)
  auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, (int)(n_floats * 4), 0x00020000);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  float acc = 0.f;
  for (size_t base = (size_t)blockIdx.x * 4096; base < n_floats * 4; base += (size_t)gridDim.x * 4096) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + wave * 1024), 16, (unsigned)(base + wave * 1024 + lane * 16), 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    acc += ((float*)smem)[threadIdx.x];
    __syncthreads();
  }
  if (acc == 123.456f) out[0] = acc;
#endif
}
__global__ void write_dword(float* dst, size_t n_floats) {   // 4 B per lane, 64 consecutive floats per wave
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_floats; i += (size_t)gridDim.x * blockDim.x) dst[i] = (float)i;
}
__global__ void write_dwordx4(float* dst, size_t n_floats) {
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n_floats; i += (size_t)gridDim.x * blockDim.x * 4) {
    f32x4 v = {(float)i, 1.f, 2.f, 3.f};
    *(f32x4*)(dst + i) = v;
  }
}
__global__ void read_dword(const float* src, float* out, size_t n_floats) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_floats; i += (size_t)gridDim.x * blockDim.x) acc += src[i];
  if (acc == 123.456f) out[0] = acc;
}
int main() {
  const size_t n = (size_t)256 << 20;  // 1 GiB of floats = 268M floats -> 1 GiB
  float *a, *b;
  if (hipMalloc(&a, n * 4) != hipSuccess || hipMalloc(&b, n * 4) != hipSuccess) return 1;
  hipMemset(a, 0, n * 4);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(read_dma16, dim3(2048), dim3(256), 4096, 0, a, b, n);
    hipLaunchKernelGGL(write_dword, dim3(2048), dim3(256), 0, 0, b, n);
    hipLaunchKernelGGL(write_dwordx4, dim3(2048), dim3(256), 0, 0, b, n);
    hipLaunchKernelGGL(read_dword, dim3(2048), dim3(256), 0, 0, a, b, n);
  }
  hipDeviceSynchronize();
  printf("each kernel moved %zu bytes\n", n * 4);
  return 0;
}
