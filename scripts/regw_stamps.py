"""Per-wave phase times of one conv1x1_regw_f32 launch from a -DPR_TIMING_HOOKS build (POSERISK_REGW_STAMPS=<file>): per wave
s_memrealtime (100 MHz) at entry / exit, summed time on weight loads | wait + barrier | epilogue + residual requests | MFMA loop,
units, HW_ID.    python scripts/regw_stamps.py <file>"""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8).astype(np.int64)
a = a[a[:, 0] > 0]
t0 = a[:, 0].min()
q = lambda v: " ".join(f"{np.quantile(v, p):8.2f}" for p in (0, 0.1, 0.5, 0.9, 1.0))
us = lambda v: v / 100.0
print(f"{len(a)} waves; microseconds: min p10 p50 p90 max")
print("  entry after first   ", q(us(a[:, 0] - t0)))
print("  exit after first    ", q(us(a[:, 1] - t0)))
print("  lifetime            ", q(us(a[:, 1] - a[:, 0])))
print("  weight loads        ", q(us(a[:, 2])))
print("  wait + barrier      ", q(us(a[:, 3])))
print("  epilogue + residual ", q(us(a[:, 4])))
print("  MFMA loop           ", q(us(a[:, 5])))
print("  units               ", q(a[:, 6]))
n = np.maximum(a[:, 6], 1)
print("  per unit: wait      ", q(us(a[:, 3]) / n))
print("  per unit: epilogue  ", q(us(a[:, 4]) / n))
print("  per unit: MFMA loop ", q(us(a[:, 5]) / n))
