"""Per-workgroup timeline of one conv_dma_f32 launch from a -DPR_TIMING_HOOKS build (POSERISK_CONV_STAMPS=<file>):
s_memrealtime (100 MHz) at 0 entry | 1 first stage landed | 2 main loop done | 3 stores issued; s_memtime at entry / exit;
HW_ID; XCC_ID.  Prints when workgroups start and end relative to the first start, and how long they live.
    python scripts/conv_stamps.py <file>"""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8).astype(np.int64)
ok = (a[:, 0] > 0) & (a[:, 3] > 0)
a = a[ok]
t0 = a[:, 0].min()
us = lambda v: (v - t0) / 100.0
start, first, main, end = us(a[:, 0]), us(a[:, 1]), us(a[:, 2]), us(a[:, 3])
cu = ((a[:, 7] & 0xf) << 8) | ((a[:, 6] >> 8) & 0xf) | (((a[:, 6] >> 13) & 0x7) << 4)   # (xcc, se, cu)
q = lambda v: " ".join(f"{np.quantile(v, p):7.2f}" for p in (0, 0.1, 0.5, 0.9, 1.0))
print(f"{len(a)} workgroups on {len(set(cu.tolist()))} (xcc, se, cu) slots; microseconds after the first workgroup's entry: min p10 p50 p90 max")
print("  entry              ", q(start))
print("  first stage landed ", q(first), "  (prologue: ", q(first - start), ")")
print("  main loop done     ", q(main), "  (main loop: ", q(main - first), ")")
print("  stores issued      ", q(end), "  (epilogue: ", q(end - main), ")")
print("  lifetime           ", q(end - start))
clk = (a[:, 5] - a[:, 4]) / np.maximum(end - start, 1e-9) / 1e3
print(f"  s_memtime ticks per microsecond of lifetime / 1000: {np.median(clk):.3f}")
per = {}
for c, s_, e_ in zip(cu.tolist(), start.tolist(), end.tolist()):
    per.setdefault(c, []).append((s_, e_))
n_per = np.array([len(v) for v in per.values()])
last_end = np.array([max(e for _, e in v) for v in per.values()])
print(f"  workgroups per CU: min {n_per.min()} max {n_per.max()}; a CU's last end: ", q(last_end))
second = sorted(s_ for v in per.values() for s_, _ in sorted(v)[3:])
if second:
    print(f"  workgroups beyond a CU's first three start at: ", q(np.array(second)))
