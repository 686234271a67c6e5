#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
__global__ void k(const float* src, float* out, int nbytes) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // poison
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = -7.f;
  __syncthreads();
  auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
  // lane i loads 16 bytes at offset (i*16) for i<32, OOB for i>=32
  unsigned voff = threadIdx.x < 32 ? threadIdx.x * 16 : 0x7fffff00u;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += blockDim.x) out[i] = lds[i];
}
int main() {
  float *s, *o; hipMalloc(&s, 4096); hipMalloc(&o, 4096);
  float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = i + 1;
  hipMemcpy(s, h, 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, s, o, 4096);
  hipMemcpy(h, o, 1024, hipMemcpyDeviceToHost);
  printf("in-range: %g %g %g ... lane31: %g ; OOB lane32: %g %g lane63: %g\n", h[0], h[1], h[4], h[127], h[128], h[129], h[255]);
  return 0;
}
