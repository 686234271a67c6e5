"""Per-layer conv time inside the real pipeline (serial, hipEvent brackets).  usage: layer_table.py [B] [fp32|bf16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from poserisk_release_amd import synth, pipeline as pl
from poserisk_release_amd.hmr import HMR
from poserisk_release_amd.smpl_layer import SMPLLayer
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
prec = sys.argv[2] if len(sys.argv) > 2 else "fp32"
m = HMR(max_batch=B, precision=prec).to(dev); m.load_state_dict(synth.hmr_state_dict(seed=1))
layer = SMPLLayer(synth.smpl_model(V=6890, seed=2), device=dev, max_batch=B)
pipe = pl.FramePipeline(m, layer, synth.EXAMPLE_INFO, with_verts=True)
x = torch.rand((B, 3, 224, 224), device=dev)
for _ in range(5): pipe(x)
torch.cuda.synchronize()
m.profile_enable(True)
N = 20
for _ in range(N): pipe(x)
torch.cuda.synchronize()
ms, cnt, fl = m.profile_read()
tot, carried = 0, 0.0
for i in range(53):
    if cnt[i] == 0:      # a layer that ran inside a later launch (a downsample branch in its conv3, conv1 / conv2 of a block taken by a
        carried += fl[i] # whole-block kernel at this batch): its FLOP count with the launch that reports next
        continue
    t = ms[i] / cnt[i] * 1e3
    tot += t
    f = fl[i] + carried
    carried = 0.0
    print(f"L{i:2d} {t:8.1f} us  {f*B/(t*1e-6)/1e12:6.1f} TF  gflop={f*B/1e9:7.2f}")
print(f"total {tot/1e3:.3f} ms  -> {fl.sum()*B/(tot*1e-6)/1e12:.1f} TF")
