import sys, time; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch, numpy as np
from oracle import hmr_ref
from poserisk_release_amd import synth
from stress_weights import trained_like_state_dict
torch.set_num_threads(8)
t=time.time(); sd = trained_like_state_dict(seed=5); print("sd", time.time()-t)
x = torch.from_numpy(synth.crops(16, seed=3))
m64 = hmr_ref.build(sd).double(); m32 = hmr_ref.build(sd)
with torch.no_grad():
    t=time.time(); xf = m64.features(x.double()); print("fp64 16 frames", time.time()-t)
    t=time.time(); xf = m32.features(x); print("fp32 16 frames", time.time()-t)
