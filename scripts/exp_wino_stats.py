"""Distribution of the fp32 encoder's error per Winograd form: >= 256 frames x >= 3 seeds of the
trained-like stress weights (tests/stress_weights.py), every joint, against an fp64 run of the same network and against
the fp32 oracle (oracle/hmr_ref.py on torch-CPU: what the reference computes).  Reported per form, pooled over seeds,
frames, joints and matrix elements: rms / p99 / max of
  rot      |rotmat(GPU) - rotmat(fp64)|                                   (includes rot6d's fp32 amplification on
                                                                           nearly degenerate 6-D vectors)
  rot<-p6  |rot6d_to_rotmat_fp64(pose6d(GPU)) - rotmat(fp64)|             (the encoder + regressor error PROPAGATED to
                                                                           the rotation, without fp32 rounding inside rot6d)
  p6       |pose6d(GPU) - pose6d(fp64)|
  xf       relative rms of the pooled features
and each as a ratio to the direct form's.  Conv time per step at B=64 (He-normal weights) beside it.

    python scripts/exp_wino_stats.py [--frames 256] [--seeds 5 6 7] [--forms 0 2 244 4 5 255]
"""
import argparse, json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
from oracle import hmr_ref
from poserisk_release_amd import synth
from poserisk_release_amd.hmr import HMR
from stress_weights import trained_like_state_dict

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=256)
ap.add_argument("--seeds", type=int, nargs="+", default=[5, 6, 7])
ap.add_argument("--forms", type=int, nargs="+", default=[0, 2, 244, 4, 5, 255])
ap.add_argument("--out", default=None)
args = ap.parse_args()
dev = torch.device("cuda", 0)
torch.set_num_threads(min(16, os.cpu_count() or 1))
n = args.frames


def stats(e):
    e = np.abs(np.asarray(e, np.float64)).ravel()
    return dict(rms=float(np.sqrt(np.mean(e * e))), p99=float(np.quantile(e, 0.99)), max=float(e.max()))


pool = {f: {k: [] for k in ("rot", "rot_p6", "p6", "rot_vs32", "p6_vs32", "xf_rel")} for f in args.forms}
oracle32 = {k: [] for k in ("rot", "p6")}
cond_all = []
for seed in args.seeds:
    t0 = time.time()
    sd = trained_like_state_dict(seed=seed)
    x = synth.crops(n, seed=100 + seed)
    m64, m32 = hmr_ref.build(sd).double(), hmr_ref.build(sd)
    xf64, p664, p632 = [], [], []
    with torch.no_grad():
        for i in range(0, n, 32):
            xb = torch.from_numpy(x[i:i + 32])
            f = m64.features(xb.double()); xf64.append(f); p664.append(m64.regress(f)[0])
            p632.append(m32.regress(m32.features(xb))[0])
        xf64, p664, p632 = torch.cat(xf64), torch.cat(p664), torch.cat(p632)
        r64 = hmr_ref.rot6d_to_rotmat(p664).view(n, 24, 3, 3)
        r32 = hmr_ref.rot6d_to_rotmat(p632).view(n, 24, 3, 3).double()
    v = p664.view(n * 24, 3, 2)
    a1, a2 = v[:, :, 0], v[:, :, 1]
    b1 = a1 / a1.norm(dim=1, keepdim=True)
    u2 = a2 - (b1 * a2).sum(1, keepdim=True) * b1
    cond_all.append(torch.minimum(a1.norm(dim=1), u2.norm(dim=1)).numpy())
    oracle32["rot"].append((r32 - r64).numpy()); oracle32["p6"].append((p632.double() - p664).numpy())
    print(f"[seed {seed}] references in {time.time() - t0:.0f} s; ill-conditioned joints (min norm < 0.5): "
          f"{float((cond_all[-1] < 0.5).mean()):.3f}", flush=True)
    xg = torch.from_numpy(x).to(dev)
    for f in args.forms:
        m = HMR(max_batch=64, conv_form=f).to(dev); m.load_state_dict(sd)
        rot, p6, xf = [], [], []
        for i in range(0, n, 64):
            r, _, _, xfg, p6g = m(xg[i:i + 64], return_features=True)
            rot.append(r.cpu()); p6.append(p6g.cpu()); xf.append(xfg.cpu())
        rot, p6, xf = torch.cat(rot).double(), torch.cat(p6).double(), torch.cat(xf).double()
        with torch.no_grad():
            rp = hmr_ref.rot6d_to_rotmat(p6).view(n, 24, 3, 3)
        P = pool[f]
        P["rot"].append((rot - r64).numpy()); P["rot_p6"].append((rp - r64).numpy()); P["p6"].append((p6 - p664).numpy())
        P["rot_vs32"].append((rot - r32).numpy()); P["p6_vs32"].append((p6 - p632.double()).numpy())
        P["xf_rel"].append(float((xf - xf64).pow(2).mean().sqrt() / xf64.pow(2).mean().sqrt()))
        del m

# conv time per step at B = 64 on the benign weights (the bench's), per form
sd_he = synth.hmr_state_dict(seed=1)
xb = torch.rand((64, 3, 224, 224), device=dev)
speed = {}
for f in args.forms:
    m = HMR(max_batch=64, conv_form=f).to(dev); m.load_state_dict(sd_he)
    for _ in range(3): m(xb)
    torch.cuda.synchronize()
    m.profile_enable(True)
    for _ in range(10): m(xb)
    torch.cuda.synchronize()
    ms, cnt, fl = m.profile_read(); m.profile_enable(False)
    speed[f] = float(ms.sum() / 10)
    del m

res = {"frames": n, "seeds": args.seeds, "samples_per_form": n * len(args.seeds) * 24 * 9,
       "ill_conditioned_frac": float((np.concatenate(cond_all) < 0.5).mean()),
       "fp32_oracle_vs_fp64": {k: stats(np.concatenate([a.ravel() for a in v])) for k, v in oracle32.items()}, "forms": {}}
for f in args.forms:
    P = pool[f]
    res["forms"][str(f)] = {k: stats(np.concatenate([a.ravel() for a in P[k]])) for k in ("rot", "rot_p6", "p6", "rot_vs32", "p6_vs32")}
    res["forms"][str(f)]["xf_rel_rms"] = float(np.mean(P["xf_rel"]))
    res["forms"][str(f)]["conv_ms_per_step_b64"] = speed[f]
d0 = res["forms"].get("0")
print(f"\n{n} frames x seeds {args.seeds} = {res['samples_per_form']} rotation-matrix elements per form; "
      f"{res['ill_conditioned_frac']:.3f} of the joints have a nearly degenerate 6-D vector")
o = res["fp32_oracle_vs_fp64"]
print(f"fp32 oracle vs fp64:   rot rms {o['rot']['rms']:.2e} p99 {o['rot']['p99']:.2e} max {o['rot']['max']:.2e} | p6 rms {o['p6']['rms']:.2e} p99 {o['p6']['p99']:.2e} max {o['p6']['max']:.2e}")
print(f"{'form':>5s} | {'rot vs fp64: rms p99 max':^34s} | {'rot<-p6 vs fp64: rms p99 max':^34s} | {'p6 vs fp64: rms p99 max':^34s} | xf rel rms | rot vs fp32 oracle: rms p99 max | conv ms")
for f in args.forms:
    e = res["forms"][str(f)]
    def col(k):
        s = e[k]; r = ""
        if d0: r = f" (x{s['rms'] / d0[k]['rms']:.2f} x{s['p99'] / d0[k]['p99']:.2f} x{s['max'] / d0[k]['max']:.2f})"
        return f"{s['rms']:.2e} {s['p99']:.2e} {s['max']:.2e}" + r
    print(f"{f:5d} | {col('rot')} | {col('rot_p6')} | {col('p6')} | {e['xf_rel_rms']:.2e} | {col('rot_vs32')} | {e['conv_ms_per_step_b64']:.3f}", flush=True)
if args.out:
    json.dump(res, open(args.out, "w"), indent=1)
