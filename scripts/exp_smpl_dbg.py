import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from poserisk_release_amd import synth
from poserisk_release_amd.smpl_layer import SMPLLayer
dev = torch.device("cuda", 0)
model = synth.smpl_model(V=6890, seed=2)
l = SMPLLayer(model, device=dev, max_batch=64); l._ensure()
for B in (16, 64):
    pose = torch.from_numpy(synth.poses(B, seed=1)).to(dev); betas = torch.from_numpy(synth.betas(B, seed=2)).to(dev)
    for dbg in (0, 1, 2, 3):
        os.environ["POSERISK_SMPL_DBG"] = str(dbg)
        for _ in range(3): l(pose, betas)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): l(pose, betas)
        e1.record(); torch.cuda.synchronize()
        print(f"B={B} dbg={dbg}: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us per forward", flush=True)
