"""Round 6, experiment 6: the 256 x 256 bf16 tile (cfg 18) against the 128 x 128 / 8-wave tile (cfg 12), stand-alone, on layer4's
shapes at B = 256 and on a shape with many tiles (how fast is the tile itself when the grid fills the chip?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from poserisk_release_amd import ops
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
for name, (B, H, Cin, Cout, k, s, p, res) in {
        "layer4 3x3 512->512 @7x7": (256, 7, 512, 512, 3, 1, 1, False),
        "layer4.0 3x3/2 512->512 @14->7": (256, 14, 512, 512, 3, 2, 1, False),
        "layer4 1x1 2048->512 @7x7": (256, 7, 2048, 512, 1, 1, 0, False),
        "layer4 1x1 512->2048 @7x7 + res": (256, 7, 512, 2048, 1, 1, 0, True),
        "many tiles: 1x1 512->512 @28x28": (256, 28, 512, 512, 1, 1, 0, False),
        "many tiles: 3x3 256->256 @14x14 x4 frames": (1024, 14, 256, 256, 3, 1, 1, False)}.items():
    x = torch.randn((B, H, H, Cin), device=dev).bfloat16()
    w = (rng.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    Ho = (H + 2 * p - k) // s + 1
    r = torch.randn((B, Ho, Ho, Cout), device=dev).bfloat16() if res else None
    out = []
    ys = {}
    for cfg in (12, 18, 12, 18):
        y, ms = ops.conv2d_nhwc(x, w, b, r, stride=s, pad=p, relu=True, tile_cfg=cfg, precision="bf16", repeats=30)
        ys[cfg] = y
        out.append(f"cfg {cfg}: {ms*1e3:6.1f} us {2.0*B*Ho*Ho*Cout*Cin*k*k/ms/1e9:6.0f} TF")
    print(f"{name:44s} tiles {-(-B*Ho*Ho//128)*(Cout//128):5d} / {-(-B*Ho*Ho//256)*(Cout//256):4d}   " + " | ".join(out) + f"   same bits: {bool(torch.equal(ys[12], ys[18]))}")
