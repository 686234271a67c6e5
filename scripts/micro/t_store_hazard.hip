// Does a VALU write of a buffer_store_dwordx4's first data register, issued right behind the store, reach memory?
// (hipcc --offload-arch=gfx950 t_store_hazard.hip)   The ISA lists "VMEM store of more than 64 bits followed by a write of
// its data VGPRs" as a hazard; hipcc pads it unless soffset is a register (LLVM GCNHazardRecognizer::createsVALUHazard).
// Here the sequence is fixed by inline assembly: data = A in v[20:23], store, then `v_mov_b32 v20, B` after `nops` wait
// states, soffset either an SGPR or the constant 0; every wave stores its own 1 KB again and again (back-pressure is what
// makes the store read its data late), and the host counts first dwords that came out as B.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <int NOPS, bool SREG>
__global__ __launch_bounds__(512) void k(unsigned* out, unsigned bytes, int iters) {
  const unsigned lane = threadIdx.x & 63, wave = blockIdx.x * 8 + (threadIdx.x >> 6);
  const unsigned long long base = (unsigned long long)out;
  const i32x4 rsrc = {(int)(unsigned)base, (int)((base >> 32) & 0xffff), (int)bytes, 0x00020000};
  const unsigned A = 0x11110000u + lane, B = 0xBAD00000u + lane;
  for (int it = 0; it < iters; ++it) {
    const unsigned voff = ((wave * (unsigned)iters + it) * 64 + lane) * 16;
    const int soff = SREG ? (it & 1) * 0 : 0;     // an SGPR holding 0 (not foldable: depends on it) / the literal
    if (SREG) {
      asm volatile(
          "v_mov_b32 v20, %[a]\n v_mov_b32 v21, %[a]\n v_mov_b32 v22, %[a]\n v_mov_b32 v23, %[a]\n s_nop 7\n"
          "buffer_store_dwordx4 v[20:23], %[voff], %[rsrc], %[soff] offen\n"
          ".rept %[nops]\n s_nop 0\n .endr\n"
          "v_mov_b32 v20, %[b]\n"
          ::[a] "v"(A), [b] "v"(B), [voff] "v"(voff), [rsrc] "s"(rsrc), [soff] "s"(soff), [nops] "n"(NOPS)
          : "v20", "v21", "v22", "v23", "memory");
    } else {
      asm volatile(
          "v_mov_b32 v20, %[a]\n v_mov_b32 v21, %[a]\n v_mov_b32 v22, %[a]\n v_mov_b32 v23, %[a]\n s_nop 7\n"
          "buffer_store_dwordx4 v[20:23], %[voff], %[rsrc], 0 offen\n"
          ".rept %[nops]\n s_nop 0\n .endr\n"
          "v_mov_b32 v20, %[b]\n"
          ::[a] "v"(A), [b] "v"(B), [voff] "v"(voff), [rsrc] "s"(rsrc), [nops] "n"(NOPS)
          : "v20", "v21", "v22", "v23", "memory");
    }
  }
}
// the same with a flat-encoded global store (what hipcc emits for a plain 16-byte `*p = v`)
template <int NOPS>
__global__ __launch_bounds__(512) void kg(unsigned* out, int iters) {
  const unsigned lane = threadIdx.x & 63, wave = blockIdx.x * 8 + (threadIdx.x >> 6);
  const unsigned A = 0x11110000u + lane, B = 0xBAD00000u + lane;
  for (int it = 0; it < iters; ++it) {
    unsigned* p = out + ((size_t)(wave * (unsigned)iters + it) * 64 + lane) * 4;
    asm volatile(
        "v_mov_b32 v20, %[a]\n v_mov_b32 v21, %[a]\n v_mov_b32 v22, %[a]\n v_mov_b32 v23, %[a]\n s_nop 7\n"
        "global_store_dwordx4 %[p], v[20:23], off\n"
        ".rept %[nops]\n s_nop 0\n .endr\n"
        "v_mov_b32 v20, %[b]\n"
        ::[a] "v"(A), [b] "v"(B), [p] "v"(p), [nops] "n"(NOPS)
        : "v20", "v21", "v22", "v23", "memory");
  }
}
template <int NOPS, bool SREG, bool GLOBAL = false>
void run(const char* what) {
  const int grid = 1024, iters = 64;
  const size_t n = (size_t)grid * 8 * iters * 64 * 4;
  unsigned* d;
  hipMalloc(&d, n * 4);
  hipMemset(d, 0, n * 4);
  if (GLOBAL) hipLaunchKernelGGL((kg<NOPS>), dim3(grid), dim3(512), 0, 0, d, iters);
  else hipLaunchKernelGGL((k<NOPS, SREG>), dim3(grid), dim3(512), 0, 0, d, (unsigned)(n * 4), iters);
  hipDeviceSynchronize();
  std::vector<unsigned> h(n);
  hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
  size_t bad = 0, other = 0;
  for (size_t i = 0; i < n; i += 4) {
    if ((h[i] & 0xffff0000u) == 0xBAD00000u) ++bad;
    else if ((h[i] & 0xffff0000u) != 0x11110000u) ++other;
  }
  printf("%-46s %8zu of %zu stores wrote the NEW first dword (%zu neither)\n", what, bad, n / 4, other);
  hipFree(d);
}
int main() {
  run<0, true>("soffset in an SGPR, VALU write right behind:");
  run<1, true>("soffset in an SGPR, 1 wait state between:");
  run<2, true>("soffset in an SGPR, 2 wait states between:");
  run<4, true>("soffset in an SGPR, 4 wait states between:");
  run<0, false>("soffset the constant 0, VALU write right behind:");
  run<1, false>("soffset the constant 0, 1 wait state between:");
  run<2, false>("soffset the constant 0, 2 wait states between:");
  run<0, false, true>("global_store_dwordx4, VALU write right behind:");
  run<1, false, true>("global_store_dwordx4, 1 wait state between:");
  run<2, false, true>("global_store_dwordx4, 2 wait states between:");
  return 0;
}
