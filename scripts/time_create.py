import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from poserisk_release_amd import synth
from poserisk_release_amd.hmr import HMR
sd = synth.hmr_state_dict(seed=1)
t = time.time(); m = HMR(max_batch=64).to("cuda:0"); m.load_state_dict(sd); m._ensure(64); torch.cuda.synchronize(); print("first handle", round(time.time() - t, 2), "s")
t = time.time(); c = m.clone(); c.to("cuda:0")._ensure(64); torch.cuda.synchronize(); print("clone", round(time.time() - t, 2), "s")
