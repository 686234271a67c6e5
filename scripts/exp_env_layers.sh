#!/bin/bash
# Same-box A/B of one environment switch on the bf16 per-layer table (B = 256, one batch in flight): default / switch / default / switch.
#   gpurun -- 'bash scripts/exp_env_layers.sh POSERISK_BALANCED=0'
set -eo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
for i in 1 2; do
  python3 scripts/layer_table.py 256 bf16 > gpurun_out/ab_layers_base$i.txt 2>/dev/null
  env "$@" python3 scripts/layer_table.py 256 bf16 > gpurun_out/ab_layers_new$i.txt 2>/dev/null
done
python3 - <<'PY'
def rd(f):
    d = {}
    for ln in open(f):
        if ln.startswith("L"): d[ln[:3].replace(" ", "")] = float(ln[3:].split()[0])
        elif ln.startswith("total"): d["total"] = float(ln.split()[1]) * 1e3
    return d
o1, n1, o2, n2 = (rd(f"gpurun_out/ab_layers_{w}.txt") for w in ("base1", "new1", "base2", "new2"))
for k in o1:
    if k not in n1: print(k, "missing in the switched run"); continue
    d = (n1[k] + n2[k] - o1[k] - o2[k]) / 2
    if abs(d) > 1.0 or k == "total": print(f"{k:6s} {o1[k]:7.1f} {n1[k]:7.1f} {o2[k]:7.1f} {n2[k]:7.1f}  {d:+6.1f}")
PY
