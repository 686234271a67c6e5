"""Timing of a layer1 Bottleneck at the encoder's shape: the one-kernel form (csrc/bottleneck_bf16.hip) against the
two launches it replaces (conv1 on the tile kernel + conv2/conv3 fused).  `python scripts/exp_bottleneck.py [B]`."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poserisk_release_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.standard_normal((B, 56, 56, 256)).astype(np.float32)).to(dev).to(torch.bfloat16)
w1 = (rng.standard_normal((64, 256)) / 16).astype(np.float32)
w2 = (rng.standard_normal((64, 64, 3, 3)) / 24).astype(np.float32)
w3 = (rng.standard_normal((256, 64)) / 8).astype(np.float32)
b1, b2, b3 = (rng.standard_normal(n).astype(np.float32) * 0.5 for n in (64, 64, 256))
for rep in range(3):
    y, ms = ops.bottleneck_nhwc(x, w1, b1, w2, b2, w3, b3, repeats=20)
    gb = 2 * x.numel() * 2 / 1e9
    print(f"bottleneck64_bf16 B={B}: {ms * 1e3:8.1f} us   {gb / ms:7.2f} TB/s of x+y   "
          f"{2.0 * B * 3136 * 64 * 64 * 17 / ms / 1e9:7.1f} TFLOP/s", flush=True)
t1, ms1 = ops.conv2d_nhwc(x, w1.reshape(64, 256, 1, 1), b1, None, relu=True, tile_cfg=-1, precision="bf16", repeats=20)
print(f"conv1 alone (heuristic tile): {ms1 * 1e3:8.1f} us", flush=True)
y2 = ops.conv3x3_conv1x1_nhwc(t1, w2, b2, w3, b3, x, relu=True, precision="bf16")
print("equal to separate launches:", bool(torch.equal(y, y2)), int((y != y2).sum()), flush=True)
