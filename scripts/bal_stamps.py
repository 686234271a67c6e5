"""Reads the s_memtime stamps of a -DPR_TIMING_HOOKS build of conv_bal_bf16 (POSERISK_BAL_STAMPS=<file>): mean shader cycles between
six points of an interval, for waves 0-3 (lead) and 4-7 (half a stage behind).
lead: 0 before the DMA wait, 1 after it, 2 after the barrier, 3 after the DMA issue, 4 after the fragment reads are issued, 5 after the MFMAs are issued
lag : 0, 1, 2 the same, 3 after (first-half reads issued + carried MFMAs issued), 4 after the DMA issue, 5 after (next reads + MFMAs issued)"""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(256, 8, 16, 6).astype(np.int64)
for name, ws in (("lead", slice(0, 4)), ("lag ", slice(4, 8))):
    g = a[:, ws]
    ok = (g[..., 0] > 0) & (g[..., 5] > 0)
    d = [g[..., k + 1] - g[..., k] for k in range(5)]
    tot = g[:, :, 1:, 0] - g[:, :, :-1, 0]
    okt = ok[:, :, 1:] & ok[:, :, :-1]
    print(name, " ".join(f"{k}->{k+1} {np.mean(x[ok]):6.0f}" for k, x in enumerate(d)), f"| 5->0' {np.mean((g[:, :, 1:, 0] - g[:, :, :-1, 5])[okt]):6.0f} | interval {np.mean(tot[okt]):6.0f} cycles  (n={int(ok.sum())})")
