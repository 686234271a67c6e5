"""The layer2 whole-Bottleneck kernel against the three launches it replaces, B=256 28x28 bf16 (stand-alone).  usage: exp_bottleneck128.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from poserisk_release_amd import ops
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rng = np.random.default_rng(0)
x = torch.randn((B, 28, 28, 512), device=dev).bfloat16()
w1 = (rng.standard_normal((128, 512)) / 22).astype(np.float32)
w2 = (rng.standard_normal((128, 128, 3, 3)) / 34).astype(np.float32)
w3 = (rng.standard_normal((512, 128)) / 11).astype(np.float32)
b1, b2, b3 = (rng.standard_normal(n).astype(np.float32) * 0.5 for n in (128, 128, 512))
y, _ = ops.bottleneck128_nhwc(x, w1, b1, w2, b2, w3, b3, repeats=5)
y, ms = ops.bottleneck128_nhwc(x, w1, b1, w2, b2, w3, b3, repeats=30)
t1, m1 = ops.conv2d_nhwc(x, w1.reshape(128, 512, 1, 1), b1, None, relu=True, precision="bf16", repeats=30)
t2, m2 = ops.conv2d_nhwc(t1, w2, b2, None, pad=1, relu=True, precision="bf16", repeats=30)
y2, m3 = ops.conv2d_nhwc(t2, w3.reshape(512, 128, 1, 1), b3, x, relu=True, tile_cfg=300, precision="bf16", repeats=30)
gf = 2.0 * B * 784 * (512 * 128 + 1152 * 128 + 128 * 512) / 1e9
print(f"bottleneck128_bf16: {ms*1e3:.1f} us ({gf/ms/1e3:.0f} TFLOP/s, x+y {2*x.numel()*2/ms/1e9:.2f} TB/s)   separate launches: {m1*1e3:.1f} + {m2*1e3:.1f} + {m3*1e3:.1f} = {(m1+m2+m3)*1e3:.1f} us   equal: {bool(torch.equal(y, y2))}")
