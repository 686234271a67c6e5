"""Phase times of bottleneck256_bf16's first frame per workgroup from a -DPR_TIMING_HOOKS build (POSERISK_B256_STAMPS=<file>): mean shader cycles
between 0 frame start | 1 end of phase 1's slices | 2 t1 written + barrier | 3 end of phase 2's stages | 4 t2 written + barrier | 5 phase 3 first half | 6 second half;
stamp 7 is s_memrealtime (100 MHz) at the start."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(256, 8, 8).astype(np.int64)
names = ["phase 1", "t1 write", "phase 2", "t2 write", "phase 3a", "phase 3b"]
ok = (a[..., 0] > 0) & (a[..., 6] > 0)
for hw, ws in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
    g, m = a[:, ws], ok[:, ws]
    print(hw, " | ".join(f"{n} {np.mean((g[..., k + 1] - g[..., k])[m]):6.0f}" for k, n in enumerate(names)),
          f"| frame {np.mean((g[..., 6] - g[..., 0])[m]):7.0f} cycles (n={int(m.sum())})")
