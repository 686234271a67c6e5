// Where does the dispatcher put the workgroups of a grid that is a non-integer number of rounds?  (hipcc --offload-arch=gfx950)
// Each workgroup of `threads` threads and `lds` bytes of LDS records its XCC and HW_ID, spins ~30 us so that the whole grid
// is resident together (as the conv tile kernels' grids are), and the host histograms workgroups per CU.  The makespan of
// an MFMA-bound launch is set by the CU that carries the most workgroups, not by the mean.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
__global__ void k(unsigned* out, long spin) {
  extern __shared__ char smem[];
  if (threadIdx.x == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));     // HW_REG_HW_ID, 32 bits
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));   // HW_REG_XCC_ID
    out[2 * blockIdx.x] = hw;
    out[2 * blockIdx.x + 1] = xcc;
    smem[0] = 1;
  }
  const long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < spin) __builtin_amdgcn_s_sleep(8);
}
int main(int argc, char** argv) {
  const int cases[][3] = {{784, 256, 32768}, {3136, 256, 32768}, {392, 256, 32768}, {1568, 256, 32768}, {784, 512, 65536},
                          {1568, 512, 65536}, {392, 512, 65536}, {196, 512, 65536}, {640, 256, 32768}};
  for (auto& c : cases) {
    const int grid = c[0], threads = c[1], lds = c[2];
    unsigned* d; std::vector<unsigned> h(2 * grid);
    hipMalloc(&d, h.size() * 4);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(k, dim3(grid), dim3(threads), lds, 0, d, 3000L);   // 30 us at the 100 MHz memtime clock
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, int> per_cu;
    for (int b = 0; b < grid; ++b) per_cu[(h[2 * b + 1] & 0xf) << 16 | (h[2 * b] & 0xff00)]++;   // xcc | se, sh, cu bits
    std::map<int, int> hist;
    for (auto& kv : per_cu) hist[kv.second]++;
    printf("grid %5d x %3d threads, %5d B LDS: %zu CUs used; workgroups per CU:", grid, threads, lds, per_cu.size());
    int mx = 0;
    for (auto& kv : hist) { printf("  %d CUs x %d", kv.second, kv.first); mx = kv.first > mx ? kv.first : mx; }
    printf("   mean %.2f max %d -> balance %.3f\n", (double)grid / 256, mx, (double)grid / 256 / mx);
    hipFree(d);
  }
  return 0;
}
