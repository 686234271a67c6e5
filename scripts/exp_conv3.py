"""Tile sweep of the short-K conv3 layers WITH their residual (the tune script runs without one).
usage: exp_conv3.py [B] [fp32|bf16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from poserisk_release_amd import _lib, ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
prec = sys.argv[2] if len(sys.argv) > 2 else "fp32"
dev = torch.device("cuda", 0)
ncfg = _lib.load().pr_conv_num_tile_cfgs()
rng = np.random.default_rng(0)
for (H, Cin, Cout, res) in [(56, 64, 256, True), (56, 64, 256, False), (56, 256, 64, False), (28, 128, 512, True), (28, 512, 128, False),
                            (14, 256, 1024, True), (7, 512, 2048, True)]:
    x = torch.randn((B, H, H, Cin), device=dev)
    r = torch.randn((B, H, H, Cout), device=dev) if res else None
    w = (rng.standard_normal((Cout, Cin, 1, 1)) / np.sqrt(Cin)).astype(np.float32)
    flops = 2.0 * B * H * H * Cout * Cin
    nbytes = (B * H * H * (Cin + Cout * (2 if res else 1))) * (4 if prec == "fp32" else 2)
    out = []
    for cfg in list(range(6, ncfg)) + ([100] if Cin <= (256 if prec == 'fp32' else 512) and Cout > Cin else []):
        try:
            _, ms = ops.conv2d_nhwc(x, w, np.zeros(Cout, np.float32), r, relu=True, tile_cfg=cfg, repeats=20, precision=prec)
        except _lib.PoseRiskHipError:
            continue
        out.append((ms, cfg))
    print(f"H{H} {Cin}->{Cout} res={res}: " + " ".join(f"[{c}]{ms*1e3:.1f}us" for ms, c in sorted(out, key=lambda t: t[1])) +
          f"  best [{min(out)[1]}] {min(out)[0]*1e3:.1f} us = {flops/min(out)[0]/1e9:.0f} TF, {nbytes/min(out)[0]/1e9:.2f} TB/s", flush=True)
