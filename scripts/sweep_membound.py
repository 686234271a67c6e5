import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from poserisk_release_amd import _lib, ops
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
B = 64
# (name, H, Cin, Cout, k, s, p, res)
SH = [("L3 ds 64->256", 56, 64, 256, 1, 1, 0, False), ("L4 64->256+res", 56, 64, 256, 1, 1, 0, True),
      ("L5 256->64", 56, 256, 64, 1, 1, 0, False), ("L1 64->64", 56, 64, 64, 1, 1, 0, False),
      ("L14 128->512+res", 28, 128, 512, 1, 1, 0, True), ("L13 ds 256->512 s2", 56, 256, 512, 1, 2, 0, False),
      ("L11 256->128", 56, 256, 128, 1, 1, 0, False), ("L15 512->128", 28, 512, 128, 1, 1, 0, False),
      ("L27 256->1024+res", 14, 256, 1024, 1, 1, 0, True), ("L46 512->2048+res", 7, 512, 2048, 1, 1, 0, True)]
for name, H, Cin, Cout, k, s, p, res in SH:
    x = torch.randn((B, H, H, Cin), device=dev)
    w = (rng.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
    Ho = (H + 2 * p - k) // s + 1
    r = torch.randn((B, Ho, Ho, Cout), device=dev) if res else None
    out = []
    for cfg in range(6, 14):
        try:
            _, ms = ops.conv2d_nhwc(x, w, None, r, stride=s, pad=p, relu=True, tile_cfg=cfg, repeats=30)
            out.append(f"[{cfg}]{ms*1e3:6.1f}")
        except _lib.PoseRiskHipError:
            pass
    print(f"{name:22s} " + " ".join(out), flush=True)
