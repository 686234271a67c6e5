// Which offsets does the range check of a raw buffer access cover on gfx950?  (hipcc --offload-arch=gfx950 t_soffset.hip)
// A 256-byte descriptor over the first quarter of a 1 KiB buffer; stores whose VECTOR offset is in range but whose SCALAR
// offset carries them past num_records, and the other way round.  LLVM's description of the raw buffer intrinsics puts
// the scalar offset outside the range check; MEASURED on MI355X / ROCm 7.2 (profiles/r03_t_soffset.txt): the stores of
// kind (a) are dropped too, i.e. the check covers voffset + soffset.  The kernels nevertheless put every
// may-be-out-of-range term into the vector offset (or the sentinel there), so they depend on neither reading.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* buf) {
  auto rsrc = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 256, 0x00020000);
  const unsigned lane = threadIdx.x;
  // (a) voffset in range (lane * 4 < 256), soffset = 512: address 512 + lane*4 is outside the 256-byte descriptor
  __builtin_amdgcn_raw_buffer_store_b32(0x11111111u, rsrc, lane * 4, 512, 0);
  // (b) voffset out of range (0x80000000), soffset = 0
  __builtin_amdgcn_raw_buffer_store_b32(0x22222222u, rsrc, 0x80000000u + lane * 4, 0, 0);
  // (c) voffset in range but voffset + soffset would wrap to an in-range address if the sum were checked
  __builtin_amdgcn_raw_buffer_store_b32(0x33333333u, rsrc, 768 + lane * 4, 0, 0);
}
int main() {
  unsigned *d, h[256];
  hipMalloc(&d, 1024);
  hipMemset(d, 0, 1024);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, (float*)d);
  hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
  int a = 0, b = 0, c = 0;
  for (int i = 0; i < 256; ++i) { a += h[i] == 0x11111111u; b += h[i] == 0x22222222u; c += h[i] == 0x33333333u; }
  printf("(a) in-range voffset + out-of-range soffset: %d of 64 stores landed (at byte %d..)\n", a, 512);
  printf("(b) out-of-range voffset: %d of 64 stores landed\n", b);
  printf("(c) voffset >= num_records, no soffset: %d of 64 stores landed\n", c);
  printf("=> the range check %s the scalar offset\n", a ? "does NOT cover" : "covers");
  return 0;
}
