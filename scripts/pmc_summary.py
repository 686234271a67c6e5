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
        rs = list(csv.DictReader(f))
    classify(rs)
    yield from rs


SMALL = ("smpl_skin", "maxpool3x3s2_nhwc", "maxpool", "nchw3_to_s2d", "nchw3_to_nhwc", "avgpool_nhwc", "smpl_pose", "fc_rows16")
TRANSFORM = ("wino43_input_transform", "wino43_output_transform", "wino_input_transform", "wino_output_transform")


def classify(rs):
    """Conv family BY EXCLUSION, not by name: with one batch in flight a step's dispatches are, in dispatch order, the
    layout change (nchw3_to_*), then the encoder's convolution launches (whatever the kernels are called this round),
    then the global average pool.  Everything between the two that is not a pooling kernel is `conv`; Winograd transform
    passes are `conv` with r["_transform"] = True.  (Round 4's summary listed conv kernels by name and silently dropped a
    new one: 19 of 67 launches per step.)  Sets r["_kind"] on every row."""
    disp = {}
    for r in rs:
        disp.setdefault(int(r["Dispatch_Id"]), r["Kernel_Name"])
    inside = False
    kinds = {}
    for d in sorted(disp):
        n = disp[d]
        if "nchw3_to_" in n:
            inside = True
            kinds[d] = "nchw3_to_s2d" if "s2d" in n else "nchw3_to_nhwc"
            continue
        if "avgpool_nhwc" in n:
            inside = False
            kinds[d] = "avgpool_nhwc"
            continue
        if inside and "maxpool" in n:
            kinds[d] = "maxpool3x3s2_nhwc"
        elif inside and ("rocclr" in n or "at::native" in n):
            kinds[d] = None                     # a runtime copy / fill between the two (none today)
        elif inside:
            kinds[d] = "conv"
        else:
            kinds[d] = next((("fc" if k == "fc_rows16" else k) for k in SMALL if k in n), None)
            if kinds[d] is None and "conv_dma_f32" in n:
                kinds[d] = "fc"                 # POSERISK_FC_TILES=1: the regressor on the conv tiles, outside the encoder
    for r in rs:
        r["_kind"] = kinds[int(r["Dispatch_Id"])]
        r["_transform"] = any(t in r["Kernel_Name"] for t in TRANSFORM)


def kind(r):
    return r["_kind"]


def expected_conv_kernels():
    """What the LIBRARY says a step launches (bench.py's roofline object: event brackets of pr_hmr_profile_read per step +
    two transform passes per Winograd layer), from the bench line of the kernel-trace pass of the same round."""
    for name in (f"{tag}_bench_{SFX}_lanes1_under_rocprof.json", f"{tag}_bench_{SFX}_default.json"):
        for base in (root, os.path.join(REPO, "profiles")):
            p = os.path.join(base, name)
            if os.path.exists(p):
                try:
                    rf = json.load(open(p)).get("roofline") or {}
                except ValueError:
                    continue
                if "conv_kernels_per_step" in rf:
                    return rf["conv_kernels_per_step"], rf.get("winograd_layers"), p
    return None, None, None


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
    # conv family by exclusion (several queues interleave here, so by name of everything that is NOT one): a library kernel
    # that is none of the small per-frame kernels is a convolution launch
    NOT_CONV = SMALL + ("regressor_", "pose_to_euler", "reba_kernel", "rula_kernel", "rot6d", "smpl_flags", "crop_frames",
                        "f32_to_bf16", "bf16_to_f32")
    convs = [e for e in ev if "pr::" in e[2] and not any(k in e[2] for k in NOT_CONV)]
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
pk = defaultdict(lambda: defaultdict(float))           # per conv kernel name: counter sums
pkn = defaultdict(lambda: defaultdict(set))


def short(name):
    """Kernel name without namespaces and argument list, template arguments kept."""
    n = name.replace("pr::(anonymous namespace)::", "").replace("void ", "")
    depth, out_ = 0, []
    for ch in n:
        if ch == "<":
            depth += 1
        if ch == "(" and depth == 0:
            break
        if ch == ">":
            depth -= 1
        out_.append(ch)
    return "".join(out_).strip()


for cdir in ("FETCH_SIZE", "WRITE_SIZE"):
    for r in rows(cdir):
        k = kind(r)
        if k:
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]].add(r["Dispatch_Id"])
        if k == "conv":
            pk[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
            pkn[short(r["Kernel_Name"])][r["Counter_Name"]].add(r["Dispatch_Id"])
out = {
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, scripts/profile_round.sh) on "
              f"`bench.py --lanes 1 --steps 4 --warmup 2` (all its timed regions), B={BATCH} {mode}",
    "correction": "FETCH_SIZE x2 (gfx950 reports half of coalesced reads; calibrated with scripts/micro/t_traffic.hip: "
                  "1 GiB read by 16-B LDS-DMA and by dword loads both report 524 300 KiB), WRITE_SIZE x1 (1 GiB of dword "
                  "or dwordx4 stores reports 1 048 576 KiB); the counters sit on the L2's memory side, so Infinity-Cache "
                  "hits are included",
    "classification": "conv family = every dispatch between a step's layout change (nchw3_to_*) and its global average pool, "
                      "by position in the dispatch order, whatever the kernel is called (scripts/pmc_summary.py::classify)",
}
# steps the profiled run made = launches of the once-per-step layout change (bench.py times several K-step regions)
STEPS = max(len(cnt["nchw3_to_s2d"]["FETCH_SIZE"]), len(cnt["nchw3_to_nhwc"]["FETCH_SIZE"]), 1)

# ---- self-check: the counters must have seen exactly the launches the library says a step makes -----------------------
want, wino_layers, want_src = expected_conv_kernels()
for cname in ("FETCH_SIZE", "WRITE_SIZE"):
    got = len(cnt["conv"][cname]) / STEPS
    if want is None:
        print(f"WARNING: no bench line with roofline.conv_kernels_per_step found for {tag}/{SFX}: the conv-launch count "
              f"({got:g} per step in the {cname} pass) is NOT cross-checked", file=sys.stderr)
    elif abs(got - want) > 1e-9:
        raise SystemExit(f"pmc_summary: the {cname} pass holds {got:g} conv-family launches per step, the library reports "
                         f"{want} ({want_src}: event brackets + 2 transform passes per Winograd layer) -- a kernel is "
                         f"misclassified or the passes ran another configuration; refusing to write a summary")
out["conv_kernels_per_step_expected"] = want
out["winograd_layers"] = wino_layers
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
out["steps_measured"] = STEPS
out["conv_launches_per_step"] = round(len(cnt["conv"]["FETCH_SIZE"]) / STEPS, 2)
out["conv_hbm_bytes_per_step"] = round(conv_rd + conv_wr)
out["conv_read_bytes_per_step"] = round(conv_rd)
out["conv_write_bytes_per_step"] = round(conv_wr)
out["conv_read_bytes_per_layer"] = round(conv_rd / 53)
out["conv_write_bytes_per_layer"] = round(conv_wr / 53)
out["conv_hbm_bytes_per_layer"] = round((conv_rd + conv_wr) / 53)
out["conv_algorithmic_write_bytes_per_layer"] = round(11113984 * ELEM * BATCH / 53)   # conv outputs per frame
out["conv_algorithmic_write_bytes_per_step"] = 11113984 * ELEM * BATCH
out["conv_per_kernel_per_step"] = {
    name: {"launches": round(len(pkn[name]["FETCH_SIZE"]) / STEPS, 2),
           "read_bytes": round(pk[name]["FETCH_SIZE"] * 1024 * 2 / STEPS),
           "write_bytes": round(pk[name]["WRITE_SIZE"] * 1024 / STEPS)}
    for name in sorted(pk, key=lambda n: -(pk[n]["FETCH_SIZE"] * 2 + pk[n]["WRITE_SIZE"]))}
out["conv_note"] = ("*_per_launch: per kernel launch of the conv family; *_per_layer: the step's bytes over its 53 conv layers, "
                    "however many launches carry them (the maps that no longer exist in HBM -- downsample outputs, t1 / t2 of the "
                    "whole-Bottleneck kernels -- are not written: compare with conv_algorithmic_write_bytes_per_layer, the "
                    "unfused figure).  Earlier files of this name divided reads by layers and writes by launches; round 4's "
                    "file left conv1x1_regw_f32 out (5.66 GB per step where the counters held 9.69).")
out["smpl_algorithmic_bytes_per_launch"] = 19_350_000 + BATCH * 83_296
path = os.path.join(REPO, "profiles", f"{tag}_hbm_traffic_{SFX}.json")
json.dump(out, open(path, "w"), indent=1)
print(path, out.get("conv_hbm_bytes_per_launch"), "per step", out["conv_hbm_bytes_per_step"])

# MFMA busy: SQ_VALU_MFMA_BUSY_CYCLES summed over the chip / (elapsed cycles x 1024 SIMDs); elapsed cycles =
# GRBM_GUI_ACTIVE / 8 (one count per XCD).  Per dispatch the counters stay far below 2^31 (they saturate there).
per = defaultdict(dict)
for r in rows("MFMA"):
    if kind(r) == "conv":
        d = per[r["Dispatch_Id"]]
        d[r["Counter_Name"]] = float(r["Counter_Value"])
        d["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        d["name"] = short(r["Kernel_Name"])
        d["transform"] = r["_transform"]
msteps = max(len({r["Dispatch_Id"] for r in rows("MFMA") if kind(r) in ("nchw3_to_s2d", "nchw3_to_nhwc")}), 1)
if want is not None and abs(len(per) / msteps - want) > 1e-9:
    raise SystemExit(f"pmc_summary: the MFMA pass holds {len(per) / msteps:g} conv-family launches per step, the library "
                     f"reports {want} ({want_src}); refusing to write a summary")


def ratio(ds):
    ds = list(ds)
    b = sum(d["SQ_VALU_MFMA_BUSY_CYCLES"] for d in ds)
    c = sum(d["GRBM_GUI_ACTIVE"] / 8 for d in ds)
    t = sum(d["ns"] for d in ds)
    return b, c, t


gemm = [d for d in per.values() if not d["transform"]]
b1, c1, t1 = ratio(gemm)
b2, c2, t2 = ratio(per.values())
sat = sum(1 for d in per.values() if d["SQ_VALU_MFMA_BUSY_CYCLES"] >= 2 ** 31)
PEAK = 2500.0 if mode == "bf16" else 157.3
by = defaultdict(list)
for d in per.values():
    by[d["name"]].append(d)
lines = []
for name in sorted(by, key=lambda n: -sum(d["ns"] for d in by[n])):
    b, c, t = ratio(by[name])
    lines.append(f"  {name:<58s} {len(by[name]) / msteps:6.2f} launches/step {t / msteps / 1e3:9.1f} us/step  busy {b / (c * 1024):.3f}")
txt = (f"rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE on `bench.py --lanes 1 --steps 4 --warmup 2`, B={BATCH} {mode} "
       f"(kernels serialised by the counter collection); {msteps} steps\n"
       f"conv-family launches per step: {len(per) / msteps:g} = {len(gemm) / msteps:g} multiplying launches + "
       f"{(len(per) - len(gemm)) / msteps:g} Winograd transform passes; the library reports {want} "
       f"(event brackets + 2 per Winograd layer: cross-checked, the script refuses to write on a mismatch); saturated counters: {sat}\n"
       f"MFMA busy cycles / (elapsed cycles x 1024 SIMDs):\n"
       f"  (1) over the multiplying launches only (transform passes left out):      {b1 / (c1 * 1024):.3f}\n"
       f"  (2) over ALL conv-family time (transform passes at zero MFMA counted):   {b2 / (c2 * 1024):.3f}\n"
       f"  (3) per kernel:\n" + "\n".join(lines) + "\n"
       f"clock = GRBM_GUI_ACTIVE/8/duration = {c2 / t2:.3f} GHz\n"
       f"=> MFMA-busy-equivalent rate at that clock, figure (1): {b1 / (c1 * 1024) * PEAK * (c1 / t1) / 2.4:.1f} TFLOP/s; "
       f"figure (2): {b2 / (c2 * 1024) * PEAK * (c2 / t2) / 2.4:.1f} TFLOP/s (spec peak {PEAK} at 2.4 GHz)\n")
path = os.path.join(REPO, "profiles", f"{tag}_pmc_mfma_busy_{SFX}.txt")
open(path, "w").write(txt)
print(txt)
