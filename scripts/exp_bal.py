"""The evenly dealt bf16 convolution (tile_cfg 301) against the tile kernel's best configuration on the encoder's shapes at
B=256.  usage: exp_bal.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from poserisk_release_amd import ops
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
shapes = [("layer2 conv1", 28, 512, 128, 1, 1, False), ("layer2 conv2", 28, 128, 128, 3, 1, False),
          ("layer2.0 conv1", 56, 256, 128, 1, 1, False), ("layer2.0 conv2", 56, 128, 128, 3, 2, False),
          ("layer3 conv1", 14, 1024, 256, 1, 1, False), ("layer3 conv2", 14, 256, 256, 3, 1, False),
          ("layer3.0 conv1", 28, 512, 256, 1, 1, False), ("layer3.0 conv2", 28, 256, 256, 3, 2, False),
                    ("layer3 conv3", 14, 256, 1024, 1, 1, True), ("layer2 conv3", 28, 128, 512, 1, 1, True),
          ("layer4 conv1", 7, 2048, 512, 1, 1, False), ("layer4 conv2", 7, 512, 512, 3, 1, False),
          ("layer4.0 conv1", 14, 1024, 512, 1, 1, False), ("layer4.0 conv2", 14, 512, 512, 3, 2, False)
]
rng = np.random.default_rng(0)
only = sys.argv[2].split(",") if len(sys.argv) > 2 and sys.argv[2] else None
for name, H, Cin, Cout, k, stride, with_res in shapes:
    if only and name not in only:
        continue
    x = torch.randn((B, H, H, Cin), device=dev).bfloat16()
    w = (rng.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
    bias = rng.standard_normal(Cout).astype(np.float32)
    Ho = (H + 2 * (k // 2) - k) // stride + 1
    res = torch.randn((B, Ho, Ho, Cout), device=dev).bfloat16() if with_res else None
    out = torch.empty((B, Ho, Ho, Cout), device=dev, dtype=torch.bfloat16)
    t = {}
    cfgs = (-1, 301, 302) if Cout % 256 == 0 else (-1, 301)
    if with_res:
        cfgs = (-1, 300)
    for cfg in cfgs:
        ops.conv2d_nhwc(x, w, bias, res, stride=stride, pad=k // 2, relu=True, tile_cfg=cfg, precision="bf16", out=out, repeats=3)
        _, ms = ops.conv2d_nhwc(x, w, bias, res, stride=stride, pad=k // 2, relu=True, tile_cfg=cfg, precision="bf16", out=out, repeats=30)
        t[cfg] = ms * 1e3
    gf = 2.0 * B * Ho * Ho * Cout * Cin * k * k / 1e9
    print(f"{name:15s} M={B*Ho*Ho:7d} N={Cout:4d} K={Cin*k*k:4d}   tile {t[-1]:7.1f} us ({gf/t[-1]:6.1f} TF)" + (f"   dealt {t[301]:7.1f} us ({gf/t[301]:6.1f} TF) x{t[-1]/t[301]:.2f}" if 301 in t else f"   weights-in-registers {t[300]:7.1f} us x{t[-1]/t[300]:.2f}") + (f"   128-blocks {t[302]:7.1f} us x{t[-1]/t[302]:.2f}" if 302 in t else ""), flush=True)
