"""Reads the s_memtime stamps of a -DPR_TIMING_HOOKS build (POSERISK_BN_STAMPS=<file>): per wave group, the mean cycles
between the phase boundaries of one iteration of bottleneck64_bf16.  k: 0 iteration start, 1 end of the first half's
work (group A: conv2 + t2 written), 2 before barrier b1 (A: after the DMA wait), 3 after b1, 4 end of the second half's work, 5
after barrier b0."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(256, 8, 16, 6).astype(np.int64)
ok = (a[..., 0] > 0) & (a[..., 5] > 0)
names = ["work1 (0->1)", "dma wait (1->2)", "barrier b1 (2->3)", "work2 (3->4)", "barrier b0 (4->5)", "to next start (5->0')"]
for grp, ws in (("A", slice(0, 4)), ("B", slice(4, 8))):
    g = a[:, ws]
    d = [g[..., 1] - g[..., 0], g[..., 2] - g[..., 1], g[..., 3] - g[..., 2], g[..., 4] - g[..., 3], g[..., 5] - g[..., 4]]
    nxt = g[:, :, 1:, 0] - g[:, :, :-1, 5]
    m = ok[:, ws]
    if grp == "B":
        names[0], names[1] = "dma issue (0->1)", "tile + dma wait (1->2)"
    print(f"group {grp}: " + ", ".join(f"{n} {np.mean(x[m]):7.0f}" for n, x in zip(names, d)) + f", {names[5]} {np.mean(nxt[m[:, :, 1:]]):7.0f}")
    print(f"   iteration total {np.mean((g[:, :, 1:, 0] - g[:, :, :-1, 0])[m[:, :, 1:]]):7.0f} cycles (s_memtime ticks)")
