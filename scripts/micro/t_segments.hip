// What HBM rate does a conv3-like epilogue access pattern reach?  y[m][c] = relu(res[m][c] + 1) over [M][N] bf16 (or
// any 2-byte type), with the work cut the way the kernels cut it:
//   seg128:  a workgroup owns 64 rows x 64 columns (128-byte row segments), 16 B per thread, one tile at a time
//   seg512:  a workgroup owns 64 rows x 256 columns (whole 512-byte rows), 16 B per thread
//   stream:  plain grid-stride 16-byte copy
// build: hipcc -O3 --offload-arch=gfx950 t_segments.hip -o t_segments.bin ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
using u16x8 = __attribute__((ext_vector_type(8))) unsigned short;

__global__ __launch_bounds__(256) void seg(const unsigned short* res, unsigned short* y, int M, int N, int cols_per_wg) {
  const int cpr = cols_per_wg / 8;                       // 16-byte chunks per row of the workgroup's tile
  const int tiles_n = N / cols_per_wg;
  const long tile = blockIdx.x;
  const int tm = (int)(tile / tiles_n), tn = (int)(tile % tiles_n);
  for (int idx = threadIdx.x; idx < 64 * cpr; idx += 256) {
    const int r = idx / cpr, cc = idx % cpr;
    const long o = (long)(tm * 64 + r) * N + tn * cols_per_wg + cc * 8;
    if (tm * 64 + r >= M) continue;
    u16x8 v = *reinterpret_cast<const u16x8*>(res + o);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = v[e] + 1;
    *reinterpret_cast<u16x8*>(y + o) = v;
  }
}
__global__ __launch_bounds__(256) void stream(const unsigned short* res, unsigned short* y, long n) {
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += (long)gridDim.x * 256 * 8) {
    u16x8 v = *reinterpret_cast<const u16x8*>(res + i);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = v[e] + 1;
    *reinterpret_cast<u16x8*>(y + i) = v;
  }
}
int main() {
  const int M = 802816, N = 256;                          // layer1 at B=256: 411 MB per tensor
  const long n = (long)M * N;
  unsigned short *a, *b;
  if (hipMalloc(&a, n * 2) != hipSuccess || hipMalloc(&b, n * 2) != hipSuccess) return 1;
  hipMemset(a, 0, n * 2);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto time = [&](const char* name, auto launch) {
    launch(); launch();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-8s %7.1f us  %.2f TB/s (read + write)\n", name, ms * 100, 2.0 * n * 2 / (ms / 10 * 1e-3) / 1e12);
  };
  time("seg128", [&] { hipLaunchKernelGGL(seg, dim3((M / 64) * (N / 64)), dim3(256), 0, 0, a, b, M, N, 64); });
  time("seg256", [&] { hipLaunchKernelGGL(seg, dim3((M / 64) * (N / 128)), dim3(256), 0, 0, a, b, M, N, 128); });
  time("seg512", [&] { hipLaunchKernelGGL(seg, dim3((M / 64) * (N / 256)), dim3(256), 0, 0, a, b, M, N, 256); });
  time("stream", [&] { hipLaunchKernelGGL(stream, dim3(2048), dim3(256), 0, 0, a, b, n); });
  return 0;
}
