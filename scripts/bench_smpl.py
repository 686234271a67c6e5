import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from poserisk_release_amd import synth
from poserisk_release_amd.smpl_layer import SMPLLayer
dev = torch.device("cuda", 0)
import sys as _s
MB = int(_s.argv[1]) if len(_s.argv) > 1 else 2048
layer = SMPLLayer(synth.smpl_model(V=6890, seed=2), device=dev, max_batch=MB)
for B in (1, 16, 64, 256, 2048):
    pose = torch.from_numpy(synth.poses(B, seed=1)).to(dev); betas = torch.from_numpy(synth.betas(B, seed=2)).to(dev)
    for _ in range(5): layer(pose, betas)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): layer(pose, betas)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    byts = 19.35e6 + B * 83296
    print(f"B={B}: {us:.1f} us per forward (flags+pose+skin+alloc), algorithmic {byts/1e6:.1f} MB -> {byts/us/1e6:.2f} TB/s", flush=True)
