"""smpl_skin_tile against smpl_skin: same bits, and the time per forward.  usage: exp_smpl_tile.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from poserisk_release_amd import synth
from poserisk_release_amd.smpl_layer import SMPLLayer
dev = torch.device("cuda", 0)
model = synth.smpl_model(V=6890, seed=2)


def layer(tile, mb):
    os.environ["POSERISK_SMPL_TILE"] = str(tile)      # read by pr_smpl_create
    l = SMPLLayer(model, device=dev, max_batch=mb)
    l._ensure()
    return l


for mb in (64, 2048):
    old, new = layer(0, mb), layer(1, mb)
    for B in ((1, 7, 16, 64) if mb == 64 else (1, 16, 64, 100, 256, 2048)):
        pose = torch.from_numpy(synth.poses(B, seed=1)).to(dev); betas = torch.from_numpy(synth.betas(B, seed=2)).to(dev)
        vo, jo = old(pose, betas); vn, jn = new(pose, betas)
        torch.cuda.synchronize()
        same = torch.equal(vo, vn) and torch.equal(jo, jn)
        t = []
        for l in (old, new):
            for _ in range(5): l(pose, betas)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): l(pose, betas)
            e1.record(); torch.cuda.synchronize()
            t.append(e0.elapsed_time(e1) / 50 * 1e3)
        print(f"max_batch {mb} B={B}: same bits {same}  max|diff| {(vo - vn).abs().max().item():.3g}  old {t[0]:.1f} us  tile {t[1]:.1f} us", flush=True)
