"""A first block's conv3 + downsample (dual-source 1x1) at B=256 bf16: the tile kernel against the evenly dealt kernel (tile_cfg 301)
and, where it fits, the register-resident-weights kernel (300).  usage: exp_dual.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, time
from poserisk_release_amd import ops
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rng = np.random.default_rng(0)
for name, Ho, C1, C2, N, cfgs in (("layer2.0", 28, 128, 256, 512, (-1, 301, 300)), ("layer3.0", 14, 256, 512, 1024, (-1, 301)),
                                  ("layer4.0", 7, 512, 1024, 2048, (-1, 301))):
    t = torch.randn((B, Ho, Ho, C1), device=dev).bfloat16()
    x2 = torch.randn((B, 2 * Ho, 2 * Ho, C2), device=dev).bfloat16()
    w1 = (rng.standard_normal((N, C1)) / np.sqrt(C1)).astype(np.float32)
    w2 = (rng.standard_normal((N, C2)) / np.sqrt(C2)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    out = torch.empty((B, Ho, Ho, N), device=dev, dtype=torch.bfloat16)
    res = []
    for cfg in cfgs:
        for _ in range(3): ops.conv1x1_dual_nhwc(t, w1, x2, w2, bias, stride2=2, relu=True, tile_cfg=cfg, precision="bf16", out=out)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): ops.conv1x1_dual_nhwc(t, w1, x2, w2, bias, stride2=2, relu=True, tile_cfg=cfg, precision="bf16", out=out)
        torch.cuda.synchronize(); res.append((cfg, (time.perf_counter() - t0) / 20 * 1e6))
    print(name, "  ".join(f"cfg {c}: {u:7.1f} us (incl. host packing)" for c, u in res), flush=True)
