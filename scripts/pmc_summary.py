"""Turn the rocprofv3 passes of scripts/profile_round.sh into profiles/<tag>_hbm_traffic_<cfg>.json,
profiles/<tag>_pmc_mfma_busy_<cfg>.txt and profiles/<tag>_bench_<cfg>_lanes1_kernel_stats.csv (+ the two bench lines).
usage: pmc_summary.py <tag> [fp32|bf16] [gpurun_out]      cfg = b64 (fp32) | b256_bf16

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes read (calibrated with
scripts/micro/t_traffic.hip), WRITE_SIZE is exact.  Encoder convolutions are told from the regressor's FC
launches (same kernel) by their grid: >= 256 workgroups.
"""
import csv
import json
import os
import sys
from collections import defaultdict

import glob
import shutil

tag = sys.argv[1]
mode = sys.argv[2] if len(sys.argv) > 2 else "fp32"
root = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out")
REPO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
SFX = "b256_bf16" if mode == "bf16" else "b64"
BATCH = 256 if mode == "bf16" else 64
ELEM = 2 if mode == "bf16" else 4


def rows(counter_dir):
    hits = glob.glob(os.path.join(root, f"{tag}_pmc_{SFX}_{counter_dir}", "**", "*counter_collection.csv"), recursive=True)
    with open(hits[0]) as f:
        yield from csv.DictReader(f)


def kind(r):
    n = r["Kernel_Name"]
    if "conv_dma_f32" in n or "conv_dma_bf16" in n:
        # the regressor's FC layers run on the fp32 kernel with small grids
        return "conv" if int(r["Grid_Size"]) >= 256 * int(r["Workgroup_Size"]) or "bf16" in n else "fc"
    if "conv3x3_conv1x1" in n or "conv1x1_panel" in n or "wino" in n or "bottleneck64" in n or "bottleneck128" in n or "bottleneck256" in n or "stem_pool" in n or "expand_res" in n or "conv_bal" in n:
        return "conv"      # fused pairs, row panels and the transform passes of a Winograd layer: all conv-layer traffic
    if "fc_rows16" in n:
        return "fc"
    for k in ("smpl_skin", "maxpool3x3s2_nhwc", "nchw3_to_s2d", "nchw3_to_nhwc", "avgpool_nhwc", "smpl_pose"):
        if k in n:
            return k
    return None


# kernel statistics and bench lines of the same round
for src, dst in ((f"{tag}_bench_{SFX}_default.json", None), (f"{tag}_bench_{SFX}_lanes1_under_rocprof.json", None)):
    if os.path.exists(os.path.join(root, src)):
        shutil.copy(os.path.join(root, src), os.path.join(REPO, "profiles", src))
for f in glob.glob(os.path.join(root, f"{tag}_ktrace_{SFX}", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(REPO, "profiles", f"{tag}_bench_{SFX}_lanes1_kernel_stats.csv"))


# the headline mode (several batches in flight): GPU-busy fraction and kernel overlap from its kernel trace
for f in glob.glob(os.path.join(root, f"{tag}_ktrace_lanes_{SFX}", "**", "*kernel_trace.csv"), recursive=True):
    ev = []
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", r.get("Stream_Id", "?"))))
    ev.sort()
    # the timed region of `bench.py --warmup 5 --steps 30 --repeats 1`: from the 6th to the 36th layout-change launch
    # (one per step, the first kernel of a step); what follows in the trace is bench.py's one-batch-in-flight pass
    firsts = [e for e in ev if "nchw3_to" in e[2]]
    t_lo = firsts[5][0] if len(firsts) > 35 else ev[0][0]
    t_hi = firsts[35][0] if len(firsts) > 35 else ev[-1][1]
    ev = [e for e in ev if t_lo <= e[0] < t_hi]
    pts = sorted([(a, 1) for a, b, *_ in ev] + [(b, -1) for a, b, *_ in ev])
    span = pts[-1][0] - pts[0][0]
    depth, last, busy, multi = 0, pts[0][0], 0, 0
    for t, d in pts:
        if depth >= 1:
            busy += t - last
        if depth >= 2:
            multi += t - last
        depth += d
        last = t
    ksum = sum(b - a for a, b, *_ in ev)
    convs = [e for e in ev if any(k in e[2] for k in ("conv_dma", "conv3x3_conv1x1", "conv1x1_panel", "wino", "bottleneck64", "bottleneck128", "bottleneck256", "stem_pool", "expand_res", "conv_bal"))]
    fps = 30 * BATCH / ((t_hi - t_lo) * 1e-9)
    txt = (f"rocprofv3 --kernel-trace of the headline mode (bench.py default lanes, B={BATCH} {mode}), the 30 timed steps "
           f"(first kernel of step 6 to first kernel of step 36: {fps:.0f} frames/s under the profiler):\n"
           f"kernels {len(ev)} on {len({e[3] for e in ev})} queues, span {span / 1e6:.2f} ms\n"
           f"GPU busy (at least one kernel running): {busy / span:.3f} of the span\n"
           f"two or more kernels running at once:    {multi / span:.3f} of the span\n"
           f"sum of kernel durations / span:         {ksum / span:.3f}  (> 1: batches overlap; what the lanes buy)\n"
           f"conv-family kernel time / span:         {sum(b - a for a, b, *_ in convs) / span:.3f}\n"
           f"frames/s over that span from the conv launches' count: see {tag}_bench_{SFX}_lanes_under_rocprof.json\n")
    open(os.path.join(REPO, "profiles", f"{tag}_lanes_overlap_{SFX}.txt"), "w").write(txt)
    print(txt)
    src = os.path.join(root, f"{tag}_bench_{SFX}_lanes_under_rocprof.json")
    if os.path.exists(src):
        shutil.copy(src, os.path.join(REPO, "profiles", os.path.basename(src)))
for f in glob.glob(os.path.join(root, f"{tag}_ktrace_lanes_{SFX}", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(REPO, "profiles", f"{tag}_bench_{SFX}_lanes_kernel_stats.csv"))

tot = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(set))
for cdir in ("FETCH_SIZE", "WRITE_SIZE"):
    for r in rows(cdir):
        k = kind(r)
        if k:
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]].add(r["Dispatch_Id"])
out = {
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, scripts/profile_round.sh) on "
              f"`bench.py --lanes 1 --steps 4 --warmup 2` (all its timed regions), B={BATCH} {mode}",
    "correction": "FETCH_SIZE x2 (gfx950 reports half of coalesced reads; calibrated with scripts/micro/t_traffic.hip: "
                  "1 GiB read by 16-B LDS-DMA and by dword loads both report 524 300 KiB), WRITE_SIZE x1 (1 GiB of dword "
                  "or dwordx4 stores reports 1 048 576 KiB); the counters sit on the L2's memory side, so Infinity-Cache "
                  "hits are included",
}
# steps the profiled run made = launches of the once-per-step layout change (bench.py times several K-step regions)
STEPS = max(len(cnt["nchw3_to_s2d"]["FETCH_SIZE"]), len(cnt["maxpool3x3s2_nhwc"]["FETCH_SIZE"]), 1)
for k in tot:
    n = len(cnt[k]["FETCH_SIZE"])
    rd = tot[k]["FETCH_SIZE"] * 1024 * 2 / max(n, 1)
    wr = tot[k]["WRITE_SIZE"] * 1024 / max(len(cnt[k]["WRITE_SIZE"]), 1)
    out[f"{k}_launches_measured"] = n
    out[f"{k}_read_bytes_per_launch"] = round(rd)
    out[f"{k}_write_bytes_per_launch"] = round(wr)
    out[f"{k}_hbm_bytes_per_launch"] = round(rd + wr)
# the conv family also per conv LAYER (53 per step, however many launches carry them: a Winograd layer is three kernels, a
# whole-Bottleneck kernel is three layers) and per step
conv_rd = tot["conv"]["FETCH_SIZE"] * 1024 * 2 / STEPS
conv_wr = tot["conv"]["WRITE_SIZE"] * 1024 / STEPS
out["conv_launches_per_step"] = round(len(cnt["conv"]["FETCH_SIZE"]) / STEPS, 2)
out["conv_hbm_bytes_per_step"] = round(conv_rd + conv_wr)
out["conv_read_bytes_per_layer"] = round(conv_rd / 53)
out["conv_write_bytes_per_layer"] = round(conv_wr / 53)
out["conv_hbm_bytes_per_layer"] = round((conv_rd + conv_wr) / 53)
out["conv_algorithmic_write_bytes_per_layer"] = round(11113984 * ELEM * BATCH / 53)   # conv outputs per frame
out["conv_note"] = ("*_per_launch: per kernel launch of the conv family; *_per_layer: the step's bytes over its 53 conv layers, "
                    "however many launches carry them (the maps that no longer exist in HBM -- downsample outputs, t1 / t2 of the "
                    "whole-Bottleneck kernels -- are not written: compare with conv_algorithmic_write_bytes_per_layer, the "
                    "unfused figure).  Earlier files of this name divided reads by layers and writes by launches.")
out["smpl_algorithmic_bytes_per_launch"] = 19_350_000 + BATCH * 83_296
path = os.path.join(REPO, "profiles", f"{tag}_hbm_traffic_{SFX}.json")
json.dump(out, open(path, "w"), indent=1)
print(path, out.get("conv_hbm_bytes_per_launch"))

# MFMA busy: SQ_VALU_MFMA_BUSY_CYCLES summed over the chip / (elapsed cycles x 1024 SIMDs); elapsed cycles =
# GRBM_GUI_ACTIVE / 8 (one count per XCD).  Per dispatch the counters stay far below 2^31 (they saturate there).
per = defaultdict(dict)
for r in rows("MFMA"):
    if kind(r) == "conv" and "wino" not in r["Kernel_Name"]:
        per[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
        per[r["Dispatch_Id"]]["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
busy = sum(d["SQ_VALU_MFMA_BUSY_CYCLES"] for d in per.values())
cyc = sum(d["GRBM_GUI_ACTIVE"] / 8 for d in per.values())
ns = sum(d["ns"] for d in per.values())
sat = sum(1 for d in per.values() if d["SQ_VALU_MFMA_BUSY_CYCLES"] >= 2 ** 31)
PEAK = 2500.0 if mode == "bf16" else 157.3
txt = (f"rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE on `bench.py --lanes 1 --steps 4 --warmup 2`, B={BATCH} {mode} "
       f"(kernels serialised by the counter collection)\n"
       f"encoder conv launches: {len(per)} (saturated counters: {sat})\n"
       f"MFMA busy cycles / (elapsed cycles x 1024 SIMDs) = {busy / (cyc * 1024):.3f}\n"
       f"clock = GRBM_GUI_ACTIVE/8/duration = {cyc / ns:.3f} GHz\n"
       f"=> MFMA-busy-equivalent rate at that clock: {busy / (cyc * 1024) * PEAK * (cyc / ns) / 2.4:.1f} TFLOP/s "
       f"(spec peak {PEAK} at 2.4 GHz)\n")
path = os.path.join(REPO, "profiles", f"{tag}_pmc_mfma_busy_{SFX}.txt")
open(path, "w").write(txt)
print(txt)
