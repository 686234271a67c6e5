"""Phase times of bottleneck128_bf16's chunks from a -DPR_TIMING_HOOKS build (POSERISK_B128_STAMPS=<file>): mean shader cycles between
0 chunk start | 1 after the first barrier | 2 end of phase 1's slices | 3 t1 written | 4 end of phase 2's stages | 5 t2 written + barrier | 6 end of phase 3;
stamp 7 is s_memrealtime (100 MHz) at the chunk's start: the shader clock between two chunk starts."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(256, 8, 4, 8).astype(np.int64)
names = ["barrier", "phase 1", "t1 write", "phase 2", "t2 write", "phase 3"]
for c in range(4):
    g = a[:, :, c]
    ok = (g[..., 0] > 0) & (g[..., 6] > 0)
    if not ok.any():
        continue
    for hw, ws in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
        gg, m = g[:, ws], ok[:, ws]
        print(f"chunk {c} {hw}", " | ".join(f"{n} {np.mean((gg[..., k + 1] - gg[..., k])[m]):6.0f}" for k, n in enumerate(names)),
              f"| chunk {np.mean((gg[..., 6] - gg[..., 0])[m]):7.0f} cycles (n={int(m.sum())})")
both = (a[:, 0, 0, 0] > 0) & (a[:, 0, 3, 0] > 0)
if both.any():
    dt = (a[:, 0, 3, 0] - a[:, 0, 0, 0])[both].astype(float)
    dr = (a[:, 0, 3, 7] - a[:, 0, 0, 7])[both].astype(float)
    print(f"shader clock between the starts of chunks 0 and 3: {np.mean(dt / dr) * 0.1:.2f} GHz; workgroup from start of chunk 0 to end of its last chunk: "
          f"{np.mean((a[:, :, :, 6].max(axis=(1, 2)) - a[:, 0, 0, 0])[both]):.0f} cycles")
