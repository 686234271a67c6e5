import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from poserisk_release_amd import synth
from poserisk_release_amd.hmr import HMR
from oracle import hmr_ref
sd = synth.hmr_state_dict(seed=1)
ref = hmr_ref.build(sd)
x = synth.crops(8, seed=0)
with torch.no_grad():
    xf = ref.features(torch.from_numpy(x)); p6, b, c = ref.regress(xf); r = hmr_ref.rot6d_to_rotmat(p6).view(8, 24, 3, 3)
m = HMR(max_batch=8).to("cuda:0"); m.load_state_dict(sd)
rot, betas, cam, xfg, p6g = m(torch.from_numpy(x).cuda(), return_features=True)
print("xf rel", float((xfg.cpu() - xf).abs().max() / xf.abs().max()), "rotmat", float((rot.cpu() - r).abs().max()),
      "betas", float((betas.cpu() - b).abs().max()), "cam", float((cam.cpu() - c).abs().max()))
