"""Phase times of bottleneck128_bf16's second chunk from a -DPR_TIMING_HOOKS build (POSERISK_B128_STAMPS=<file>): mean shader cycles between
0 chunk start | 1 after the first barrier | 2 end of phase 1's slices | 3 t1 written | 4 end of phase 2's stages | 5 t2 written + barrier | 6 end of phase 3"""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(256, 8, 8).astype(np.int64)
ok = (a[..., 0] > 0) & (a[..., 6] > 0)
names = ["barrier", "phase 1 (conv1, 8 slices)", "t1 write", "phase 2 (conv2, 18 stages)", "t2 write", "phase 3 (conv3 + residual)"]
for hw, ws in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
    g, m = a[:, ws], ok[:, ws]
    print(hw, " | ".join(f"{n} {np.mean((g[..., k + 1] - g[..., k])[m]):7.0f}" for k, n in enumerate(names)), f"| chunk {np.mean((g[..., 6] - g[..., 0])[m]):7.0f} cycles (n={int(m.sum())})")
