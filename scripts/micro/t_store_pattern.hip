// What does the SHAPE of a wave's 16-byte stores cost?  (hipcc --offload-arch=gfx950 -O3)
// One 512-thread workgroup per CU writes its own run of 1 KiB rows (a [pixels][512] bf16 map), 32 rows per step, every wave
// four `buffer_store_dwordx4` per step -- the epilogue of the transposed-MFMA kernels -- in four lane -> address shapes:
//   0  as the kernels store today: lane (i, h) -> row i, bytes 128 w + 64 n + 32 h + 16 j   (16-byte pieces 32 bytes apart)
//   1  halves swapped by the weight-row permutation: bytes 128 w + 64 n + 32 j + 16 h        (32 contiguous bytes per row and instruction)
//   2  eight lanes per row: instruction q -> rows 8 q + l / 8, bytes 128 w + 16 (l % 8)      (128 contiguous bytes per row)
//   3  a whole row per instruction: rows 4 w + q, bytes 16 l                                   (1 KiB contiguous)
// and, for the read side (the residual), the same shapes as loads.  Prints GB/s chip-wide and bytes per clock and CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
template <int SHAPE, bool LOAD>
__global__ __launch_bounds__(512) void k(char* y, int rows_per_wg, unsigned* sink) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
  char* base = y + (size_t)blockIdx.x * rows_per_wg * 1024;
  u32x4 v = {threadIdx.x, blockIdx.x, 1u, 2u}, acc = {0u, 0u, 0u, 0u};
  for (int r0 = 0; r0 < rows_per_wg; r0 += 32) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = q >> 1, j = q & 1;
      size_t off;
      if (SHAPE == 0) off = (size_t)(r0 + i) * 1024 + 128 * w + 64 * n + 32 * h + 16 * j;
      else if (SHAPE == 1) off = (size_t)(r0 + i) * 1024 + 128 * w + 64 * n + 32 * j + 16 * h;
      else if (SHAPE == 2) off = (size_t)(r0 + 8 * q + (lane >> 3)) * 1024 + 128 * w + 16 * (lane & 7);
      else off = (size_t)(r0 + 4 * w + q) * 1024 + 16 * lane;
      if (LOAD) {
        const u32x4 t = *reinterpret_cast<const u32x4*>(base + off);
        acc += t;
      } else {
        v.x += q;
        *reinterpret_cast<u32x4*>(base + off) = v;
      }
    }
  }
  if (LOAD && acc.x == 0x12345u) sink[0] = acc.y;
}
template <int SHAPE, bool LOAD>
void run(char* d, int grid, int rows, unsigned* sink, double ghz) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((k<SHAPE, LOAD>), dim3(grid), dim3(512), 0, 0, d, rows, sink);
  hipEventRecord(e0);
  const int reps = 20;
  for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((k<SHAPE, LOAD>), dim3(grid), dim3(512), 0, 0, d, rows, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)grid * rows * 1024, s = ms / 1e3 / reps;
  printf("  %s shape %d, %3d workgroups x %5d rows: %7.1f us  %6.2f TB/s  %5.1f B/clk/CU (at %.1f GHz)\n", LOAD ? "load " : "store", SHAPE, grid,
         rows, s * 1e6, bytes / s / 1e12, bytes / s / grid / (ghz * 1e9), ghz);
}
int main() {
  const int rows = 24 * 32 * 4;   // 3 MiB per workgroup, 768 MiB at 256 workgroups: beyond the Infinity Cache
  char* d; unsigned* sink;
  hipMalloc(&d, (size_t)256 * rows * 1024);
  hipMalloc(&sink, 64);
  hipMemset(d, 1, (size_t)256 * rows * 1024);
  const double ghz = 2.0;
  for (int grid : {256, 64}) {
    run<0, false>(d, grid, rows, sink, ghz); run<1, false>(d, grid, rows, sink, ghz);
    run<2, false>(d, grid, rows, sink, ghz); run<3, false>(d, grid, rows, sink, ghz);
    run<0, true>(d, grid, rows, sink, ghz); run<1, true>(d, grid, rows, sink, ghz);
    run<2, true>(d, grid, rows, sink, ghz); run<3, true>(d, grid, rows, sink, ghz);
  }
  return 0;
}
