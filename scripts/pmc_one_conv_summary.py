"""Average per dispatch of every counter collected by scripts/pmc_one_conv.sh.  usage: pmc_one_conv_summary.py <name>"""
import csv, glob, os, sys
from collections import defaultdict
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "pmc_" + sys.argv[1])
acc, n, ns = defaultdict(float), defaultdict(int), []
for f in sorted(glob.glob(os.path.join(root, "p*", "pmc_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        if "conv_dma" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
            ns.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
avg = {k: acc[k] / n[k] for k in acc}
dur = sum(ns) / len(ns)
cyc = avg["GRBM_GUI_ACTIVE"] / 8
print(f"{sys.argv[1]}: duration {dur/1e3:.1f} us, clock {cyc/dur:.3f} GHz, elapsed cycles {cyc:.0f}")
for k, v in sorted(avg.items()):
    print(f"  {k:32s} {v:16.0f}   per (cycle x 1024 SIMDs): {v/(cyc*1024):.3f}   per (cycle x 256 CUs): {v/(cyc*256):.3f}")
