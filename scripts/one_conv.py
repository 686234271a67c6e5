"""Run one conv shape/tile config repeatedly (for rocprofv3 --pmc / --kernel-trace on the GPU box).
usage: one_conv.py B H Cin Cout k stride pad cfg reps"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from poserisk_release_amd import ops

B, H, Cin, Cout, k, s, p, cfg, reps = [int(v) for v in sys.argv[1:10]]
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
Cr = Cin
Cp = max(4, Cin) if Cin >= 4 else 4
x = torch.randn((B, H, H, Cp), device=dev)
w = (rng.standard_normal((Cout, Cr, k, k)) / np.sqrt(Cr * k * k)).astype(np.float32)
y, ms = ops.conv2d_nhwc(x, w, None, None, stride=s, pad=p, relu=True, tile_cfg=cfg, repeats=reps)
Ho = (H + 2 * p - k) // s + 1
print(f"{ms*1e3:.1f} us  {2.0*B*Ho*Ho*Cout*Cr*k*k/ms/1e9:.1f} TF")
