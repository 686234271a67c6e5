"""Per-shape tile sweep of the encoder's 23 distinct conv shapes at a given batch (GPU box only).
Prints ms and TFLOP/s for every tile configuration; used to fill the per-layer table in hmr.hip."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from poserisk_release_amd import _lib, ops

# (H, Cin_real, Cin, Cout, k, stride, pad, count)
SHAPES = [
    (224, 3, 4, 64, 7, 2, 3, 1),
    (56, 64, 64, 64, 1, 1, 0, 1), (56, 64, 64, 64, 3, 1, 1, 3), (56, 64, 64, 256, 1, 1, 0, 4),
    (56, 256, 256, 64, 1, 1, 0, 2), (56, 256, 256, 128, 1, 1, 0, 1), (56, 128, 128, 128, 3, 2, 1, 1),
    (28, 128, 128, 512, 1, 1, 0, 4), (56, 256, 256, 512, 1, 2, 0, 1), (28, 512, 512, 128, 1, 1, 0, 3),
    (28, 128, 128, 128, 3, 1, 1, 3), (28, 512, 512, 256, 1, 1, 0, 1), (28, 256, 256, 256, 3, 2, 1, 1),
    (14, 256, 256, 1024, 1, 1, 0, 6), (28, 512, 512, 1024, 1, 2, 0, 1), (14, 1024, 1024, 256, 1, 1, 0, 5),
    (14, 256, 256, 256, 3, 1, 1, 5), (14, 1024, 1024, 512, 1, 1, 0, 1), (14, 512, 512, 512, 3, 2, 1, 1),
    (7, 512, 512, 2048, 1, 1, 0, 3), (14, 1024, 1024, 2048, 1, 2, 0, 1), (7, 2048, 2048, 512, 1, 1, 0, 2),
    (7, 512, 512, 512, 3, 1, 1, 2),
]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    prec = sys.argv[3] if len(sys.argv) > 3 else "fp32"
    dev = torch.device("cuda", 0)
    ncfg = _lib.load().pr_conv_num_tile_cfgs()
    rng = np.random.default_rng(0)
    total_best = 0.0
    total_heur = 0.0
    rows = []
    for (H, Cr, Cin, Cout, k, s, p, cnt) in SHAPES:
        if prec == 'bf16' and Cin == 4:
            Cin = 8
        x = torch.randn((B, H, H, Cin), device=dev)
        w = (rng.standard_normal((Cout, Cr, k, k)) / np.sqrt(Cr * k * k)).astype(np.float32)
        Ho = (H + 2 * p - k) // s + 1
        flops = 2.0 * B * Ho * Ho * Cout * Cr * k * k
        res = {}
        for cfg in ([-1] + list(range(ncfg))) if prec == 'fp32' else list(range(6, ncfg)):
            try:
                _, ms = ops.conv2d_nhwc(x, w, None, None, stride=s, pad=p, relu=True, tile_cfg=cfg, repeats=reps, precision=prec)
            except _lib.PoseRiskHipError:
                continue
            res[cfg] = ms
        best = min((v, c) for c, v in res.items() if c >= 0)
        total_best += best[0] * cnt
        total_heur += res.get(-1, best[0]) * cnt
        rows.append(dict(shape=[H, Cr, Cout, k, s], count=cnt, gflop=flops / 1e9, ms=res, best_cfg=best[1]))
        print(f"H{H:3d} Cin{Cr:4d} Cout{Cout:4d} k{k} s{s} x{cnt}: " +
              " ".join(f"[{c}]{v*1e3:7.1f}us/{flops/v/1e9:5.1f}TF" for c, v in sorted(res.items())) +
              f"  best={best[1]}", flush=True)
    tf = 8.174272512e9 * B
    print(f"B={B}: sum(best)={total_best:.3f} ms -> {tf/total_best/1e9:.1f} TF ; heuristic={total_heur:.3f} ms -> {tf/total_heur/1e9:.1f} TF")
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(rows, open(f"gpurun_out/tune_conv_B{B}_{prec}.json", "w"))


if __name__ == "__main__":
    main()
