#!/bin/bash
# Same-box A/B of two library builds (scripts/ab_libs/$1.so against scripts/ab_libs/$2.so): the bf16 per-layer table, A B A B.
set -eo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
for i in 1 2; do
  POSERISK_LIB_PATH=$PWD/scripts/ab_libs/$1.so python3 scripts/layer_table.py 256 bf16 > gpurun_out/ab_layers_base$i.txt 2>/dev/null
  POSERISK_LIB_PATH=$PWD/scripts/ab_libs/$2.so python3 scripts/layer_table.py 256 bf16 > gpurun_out/ab_layers_new$i.txt 2>/dev/null
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
    d = (n1[k] + n2[k] - o1[k] - o2[k]) / 2
    if abs(d) > 1.0 or k == "total": print(f"{k:6s} {o1[k]:7.1f} {n1[k]:7.1f} {o2[k]:7.1f} {n2[k]:7.1f}  {d:+6.1f}")
PY
