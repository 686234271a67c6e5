// Where a workgroup of smpl_skin_tile spends its time: 8 s_memtime stamps per workgroup (100 MHz clock), averaged.
// Build + run on the GPU box:  hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast-honor-pragmas \
//     -o scripts/micro/t_smpl_tile.bin scripts/micro/t_smpl_tile.hip && scripts/micro/t_smpl_tile.bin [B]
// Synthetic operands of the real sizes (V = 6890, 207 coefficients, 4 nonzero weights); values do not matter here.
#include "../../poserisk_release_amd/csrc/smpl.hip"

#include <cstdio>

namespace pr { void set_error(const char*, ...) {} }   // the library keeps its last error in capi.hip; not linked here

int main(int argc, char** argv) {
  using namespace pr;
  const int B = argc > 1 ? atoi(argv[1]) : 64;
  const int V = 6890, R = ceil_div(3 * V, 64) * 64 + 64, NP = 207, NPpad = 208, NB = 10, NNZ = 4;
  const int Bs = ceil_div(B, kFB) * kFB;
  auto dalloc = [](size_t n, int fill) { void* p = nullptr; (void)hipMalloc(&p, n); (void)hipMemset(p, fill, n); return p; };
  SkinArgs sa;
  sa.posedirs_T = (float*)dalloc((size_t)NPpad * R * 4, 0);
  sa.shapedirs_T = (float*)dalloc((size_t)NB * R * 4, 0);
  sa.v_template = (float*)dalloc((size_t)R * 4, 0);
  sa.ell_idx = (int*)dalloc((size_t)NNZ * V * 4, 0);
  sa.ell_w = (float*)dalloc((size_t)NNZ * V * 4, 0);
  sa.A = (float*)dalloc((size_t)Bs * kJ * 12 * 4, 0);
  sa.pm_T = (float*)dalloc((size_t)NPpad * Bs * 4, 0);
  sa.betas_T = (float*)dalloc((size_t)kMaxNB * Bs * 4, 0);
  sa.voff = (float*)dalloc((size_t)Bs * 3 * 4, 0);
  sa.verts = (float*)dalloc((size_t)Bs * V * 3 * 4, 0);
  sa.V = V; sa.R = R; sa.NP = NP; sa.NPpad = NPpad; sa.NB = NB; sa.NNZ = NNZ; sa.B = B; sa.Bs = Bs;
  sa.n_fg = ceil_div(B, kFB);
  sa.n_rt = ceil_div(ceil_div(V, 21 * kRS), 8) * 8;
  const int grid = sa.n_rt * sa.n_fg;
  sa.stamps = (unsigned long long*)dalloc((size_t)grid * 8 * 8, 0);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(smpl_skin_tile<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTileLds);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(smpl_skin_tile<4>, dim3(grid), dim3(kTileThreads), kTileLds, 0, sa);
  (void)hipEventRecord(e0, 0);
  for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(smpl_skin_tile<4>, dim3(grid), dim3(kTileThreads), kTileLds, 0, sa);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> st((size_t)grid * 8);
  (void)hipMemcpy(st.data(), sa.stamps, st.size() * 8, hipMemcpyDeviceToHost);
  unsigned long long t0 = ~0ull, t1 = 0;
  double ph[7] = {};
  for (int g = 0; g < grid; ++g) {
    t0 = std::min(t0, st[g * 8]); t1 = std::max(t1, st[g * 8 + 7]);
    for (int i = 0; i < 7; ++i) ph[i] += (double)(st[g * 8 + i + 1] - st[g * 8 + i]);
  }
  printf("B=%d grid=%d: %.2f us per launch; first start -> last end %.2f us (stamps at 100 MHz)\n", B, grid, ms / 20 * 1e3, (t1 - t0) / 100.0);
  const char* nm[7] = {"prologue issue", "barrier 1 (loads land)", "coefficient loop", "Red writes + finish-phase loads issue", "barrier 2", "transforms to LDS + barrier 3", "finish (blend, skin, store)"};
  for (int i = 0; i < 7; ++i) printf("  %-40s %.2f us\n", nm[i], ph[i] / grid / 100.0);
  return 0;
}
