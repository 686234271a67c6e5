"""The fp32 stem: one fused launch (stem_pool_f32) against conv (tile kernel, 4x4 taps) + max-pool, stand-alone at B=64."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from poserisk_release_amd import ops
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rng = np.random.default_rng(0)
x = torch.randn((B, 112, 112, 12), device=dev)
w = (rng.standard_normal((64, 12, 4, 4)) / np.sqrt(192)).astype(np.float32)
b = rng.standard_normal(64).astype(np.float32)
y, ms = ops.stem_pool_f32_nhwc(x, w, b, repeats=30)
alg = 2.0 * B * 112 * 112 * 64 * 147
print(f"stem_pool_f32, B={B}: {ms*1e3:.1f} us  ({alg/ms/1e9:.1f} TF algorithmic, {2.0*B*112*112*64*192/ms/1e9:.1f} TF executed)")
