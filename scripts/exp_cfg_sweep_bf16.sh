#!/bin/bash
# Per-layer conv time of the bf16 encoder (B = 256, one batch in flight) with every tile configuration of the tile kernel forced
# in turn wherever it fits (POSERISK_CONV_CFG; layers on other kernels are unaffected): is 128x128 / 8 waves still the best
# tile now that the tile kernel runs on v_mfma_f32_16x16x32_bf16?   gpurun -- 'bash scripts/exp_cfg_sweep_bf16.sh'
set -eo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/cfg_sweep
python3 scripts/layer_table.py 256 bf16 > gpurun_out/cfg_sweep/default.txt 2>/dev/null
for c in 6 7 8 9 10 11 12 13 17; do
  POSERISK_CONV_CFG=$c timeout -k 10 120 python3 scripts/layer_table.py 256 bf16 > gpurun_out/cfg_sweep/cfg$c.txt 2>/dev/null || echo "cfg $c failed"
done
python3 scripts/layer_table.py 256 bf16 > gpurun_out/cfg_sweep/default2.txt 2>/dev/null
python3 - <<'PY'
import glob, os
def rd(f):
    d = {}
    for ln in open(f):
        if ln.startswith("L"): d[ln[:3].replace(" ", "")] = float(ln[3:].split()[0])
        elif ln.startswith("total"): d["total"] = float(ln.split()[1]) * 1e3
    return d
base = rd("gpurun_out/cfg_sweep/default.txt"); base2 = rd("gpurun_out/cfg_sweep/default2.txt")
cfgs = sorted(glob.glob("gpurun_out/cfg_sweep/cfg*.txt"), key=lambda f: int(os.path.basename(f)[3:-4]))
tabs = {os.path.basename(f)[:-4]: rd(f) for f in cfgs}
print("layer  default default2 " + " ".join(f"{k:>7s}" for k in tabs))
for L in base:
    row = [tabs[k].get(L, float("nan")) for k in tabs]
    best = min(row + [base[L]])
    mark = "" if best >= base[L] - 1.5 else f"   <- best {best:.1f} ({list(tabs)[row.index(best)]})"
    print(f"{L:6s} {base[L]:7.1f} {base2.get(L, float('nan')):7.1f}  " + " ".join(f"{v:7.1f}" for v in row) + mark)
PY
