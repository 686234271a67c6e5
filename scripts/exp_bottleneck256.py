"""The layer3 whole-Bottleneck kernel against the three launches it replaces, B=256 14x14 bf16 (stand-alone).  usage: exp_bottleneck256.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from poserisk_release_amd import ops
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rng = np.random.default_rng(0)
x = torch.randn((B, 14, 14, 1024), device=dev).bfloat16()
w1 = (rng.standard_normal((256, 1024)) / 32).astype(np.float32)
w2 = (rng.standard_normal((256, 256, 3, 3)) / 48).astype(np.float32)
w3 = (rng.standard_normal((1024, 256)) / 16).astype(np.float32)
b1, b2, b3 = (rng.standard_normal(n).astype(np.float32) * 0.5 for n in (256, 256, 1024))
y, _ = ops.bottleneck256_nhwc(x, w1, b1, w2, b2, w3, b3, repeats=5)
y, ms = ops.bottleneck256_nhwc(x, w1, b1, w2, b2, w3, b3, repeats=30)
t1, m1 = ops.conv2d_nhwc(x, w1.reshape(256, 1024, 1, 1), b1, None, relu=True, precision="bf16", repeats=30)
t2, m2 = ops.conv2d_nhwc(t1, w2, b2, None, pad=1, relu=True, precision="bf16", repeats=30)
y2, m3 = ops.conv2d_nhwc(t2, w3.reshape(1024, 256, 1, 1), b3, x, relu=True, tile_cfg=300, precision="bf16", repeats=30)
gf = 2.0 * B * 196 * (1024 * 256 + 2304 * 256 + 256 * 1024) / 1e9
nbad = int((y != y2).sum())
print(f"bottleneck256_bf16: {ms*1e3:.1f} us ({gf/ms/1e3:.0f} TFLOP/s, x+y {2*x.numel()*2/ms/1e9:.2f} TB/s)   separate launches: {m1*1e3:.1f} + {m2*1e3:.1f} + {m3*1e3:.1f} = {(m1+m2+m3)*1e3:.1f} us   equal: {bool(torch.equal(y, y2))} ({nbad} of {y.numel()} differ, max |d| {float((y.float()-y2.float()).abs().max()):.4f})")
