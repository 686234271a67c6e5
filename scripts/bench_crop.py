"""Time the crop front-end (pr_crop_frames) and the frames -> scores path of BASELINE config 5 (no detector)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from poserisk_release_amd import ops, synth, pipeline as pl
from poserisk_release_amd.hmr import HMR
from poserisk_release_amd.smpl_layer import SMPLLayer
dev = torch.device("cuda", 0)
F, H, W, B = 256, 450, 800, 64          # the reference resizes videos to width 800 (funcs_utils.py:26-31)
frames = torch.randint(0, 256, (F, H, W, 3), dtype=torch.uint8, device=dev)
rng = np.random.default_rng(0)
bboxes = np.stack([rng.uniform(300, 500, F), rng.uniform(150, 300, F), rng.uniform(80, 200, F), rng.uniform(150, 400, F)], 1).astype(np.float32)
idx = np.arange(F, dtype=np.int32)
for _ in range(3): ops.crop_frames(frames, bboxes[:B], idx[:B])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(50): c = ops.crop_frames(frames, bboxes[:B], idx[:B])
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 50 * 1e3
print(f"pr_crop_frames: {B} crops from {H}x{W} frames: {us:.1f} us ({B*3*224*224*4/us/1e3:.0f} GB/s of crop output)")
m = HMR(max_batch=B).to(dev); m.load_state_dict(synth.hmr_state_dict(seed=1))
layer = SMPLLayer(synth.smpl_model(V=6890, seed=2), device=dev, max_batch=B)
pipe = pl.FramePipeline(m, layer, synth.EXAMPLE_INFO, lanes=3); pipe.prepare(B, dev)
def run(n):
    for s in range(n):
        lo = (s * B) % F
        pipe(ops.crop_frames(frames, bboxes[lo:lo + B], idx[lo:lo + B]))
    pipe.synchronize(); torch.cuda.synchronize()
run(6)
t = time.perf_counter(); run(40); dt = time.perf_counter() - t
print(f"frames -> crops -> pose -> SMPL joints -> REBA/RULA: {40*B/dt:.0f} frames/s (B={B}, 3 batches in flight)")
