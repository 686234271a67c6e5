"""The fp32 stem: one fused launch (stem_pool_f32) against conv (tile kernel, 4x4 taps) + max-pool, stand-alone at B=64."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from poserisk_release_amd import ops
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rng = np.random.default_rng(0)
x = torch.randn((B, 112, 112, 12), device=dev)
w7 = (rng.standard_normal((64, 3, 7, 7)) / np.sqrt(147)).astype(np.float32)
w = np.zeros((64, 12, 4, 4), np.float32)     # the 7x7 kernel in the 4x4 taps' 8x8 window (pr_hmr_create's layout)
for kh in range(7):
    for kw in range(7):
        th, di, tw, dj = (kh + 1) >> 1, (kh + 1) & 1, (kw + 1) >> 1, (kw + 1) & 1
        w[:, (2 * di + dj) * 3:(2 * di + dj) * 3 + 3, th, tw] = w7[:, :, kh, kw]
b = rng.standard_normal(64).astype(np.float32)
y, ms = ops.stem_pool_f32_nhwc(x, w, b, repeats=30)
alg = 2.0 * B * 112 * 112 * 64 * 147
print(f"stem_pool_f32, B={B}: {ms*1e3:.1f} us  ({alg/ms/1e9:.1f} TF algorithmic, {2.0*B*112*112*64*156/ms/1e9:.1f} TF executed)")
