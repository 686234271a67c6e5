"""Numerics and speed of the Winograd forms per ResNet stage (conv_form digits: layer2, layer3, layer4).
For each form: error of the fp32 HIP encoder + regressor against an fp64 run of the same network AND against the fp32
oracle (oracle/hmr_ref.py on torch-CPU: what the reference computes) on the trained-like stress weights
(tests/stress_weights.py) and on the He-normal synthetic weights, all joints; and the conv time per step at B=64.
    python scripts/exp_wino_forms.py [forms...]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
from oracle import hmr_ref
from poserisk_release_amd import synth
from poserisk_release_amd.hmr import HMR
from stress_weights import trained_like_state_dict
dev = torch.device("cuda", 0)
forms = [int(a) for a in sys.argv[1:]] or [0, 2, 4, 244, 424, 442, 224, 242, 422]
sets = {"stress": trained_like_state_dict(), "he": synth.hmr_state_dict(seed=1)}
n = 8
x = synth.crops(n, seed=3)
ref = {}
for name, sd in sets.items():
    m64 = hmr_ref.build(sd).double()
    with torch.no_grad():
        xf = m64.features(torch.from_numpy(x).double()); p6, b, c = m64.regress(xf)
        ref[name] = (xf, hmr_ref.rot6d_to_rotmat(p6).view(n, 24, 3, 3), b, c, p6)
        m32 = hmr_ref.build(sd)
        p6f, _, _ = m32.regress(m32.features(torch.from_numpy(x)))
        ref[name] += (hmr_ref.rot6d_to_rotmat(p6f).view(n, 24, 3, 3).double(), p6f.double())
        v = p6.view(n * 24, 3, 2)
        a1, a2 = v[:, :, 0], v[:, :, 1]
        b1 = a1 / a1.norm(dim=1, keepdim=True)
        u2 = a2 - (b1 * a2).sum(1, keepdim=True) * b1
        well = torch.minimum(a1.norm(dim=1), u2.norm(dim=1)) > 0.5
        print(f"[{name}] fp32 oracle vs fp64: rot {float((ref[name][5] - ref[name][1]).abs().max()):.2e} "
              f"p6 {float((ref[name][6] - p6).abs().max()):.2e}; joints with an ill-conditioned 6-D vector "
              f"(dropped by the test's mask): {float(1 - well.float().mean()):.3f}", flush=True)
xb = torch.rand((64, 3, 224, 224), device=dev)
for f in forms:
    row = f"form {f:3d}:"
    for name, sd in sets.items():
        m = HMR(max_batch=64, conv_form=f).to(dev); m.load_state_dict(sd)
        rot, betas, cam, xfg, p6g = m(torch.from_numpy(x).to(dev), return_features=True)
        xf, r, b, c, p6, r32, p632 = ref[name]
        dp = p6g.cpu().double() - p6
        d = xfg.cpu().double() - xf
        row += (f"  [{name}] xf max {float(d.abs().max() / xf.abs().max()):.2e} rms {float(d.pow(2).mean().sqrt() / xf.pow(2).mean().sqrt()):.2e}"
                f" p6 max {float(dp.abs().max()):.2e} rms {float(dp.pow(2).mean().sqrt()):.2e}"
                f" rot {float((rot.cpu().double() - r).abs().max()):.2e} betas {float((betas.cpu().double() - b).abs().max()):.2e}"
                f" | vs fp32 oracle: rot {float((rot.cpu().double() - r32).abs().max()):.2e} p6 {float((p6g.cpu().double() - p632).abs().max()):.2e}")
        if name == "he":
            for _ in range(3): m(xb)
            torch.cuda.synchronize()
            m.profile_enable(True)
            for _ in range(10): m(xb)
            torch.cuda.synchronize()
            ms, cnt, fl = m.profile_read(); m.profile_enable(False)
            row += f"  conv {ms.sum() / 10:.3f} ms/step ({fl.sum() * 64 / (ms.sum() / 10 * 1e-3) / 1e12:.1f} TF alg)"
        del m
    print(row, flush=True)
