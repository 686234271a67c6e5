"""BASELINE configs[4] without the detector, on one GPU: the crop front-end (pr_crop_frames) alone, frames -> scores with
the decoded frames RESIDENT in HBM, and frames -> scores with the FEED in the loop (feed.FrameFeed: pinned host ring of
uint8 frames -> upload -> crop -> pose / SMPL / REBA / RULA -> per-frame results back to pinned host memory), beside the
resident-crop rate bench.py measures.   usage: bench_crop.py [steps]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from poserisk_release_amd import ops, synth, pipeline as pl
from poserisk_release_amd.feed import FrameFeed
from poserisk_release_amd.hmr import HMR
from poserisk_release_amd.smpl_layer import SMPLLayer
dev = torch.device("cuda", 0)
F, H, W, B = 256, 450, 800, 64          # the reference resizes videos to width 800 (funcs_utils.py:26-31)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
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
sd, sm = synth.hmr_state_dict(seed=1), synth.smpl_model(V=6890, seed=2)
def make(lanes=3):
    m = HMR(max_batch=B).to(dev); m.load_state_dict(sd)
    p = pl.FramePipeline(m, SMPLLayer(sm, device=dev, max_batch=B), synth.EXAMPLE_INFO, lanes=lanes); p.prepare(B, dev)
    return p
pipe = make()
crops = torch.rand((B, 3, 224, 224), device=dev)
def timed(fn, n):
    fn(8); t = time.perf_counter(); fn(n); return n * B / (time.perf_counter() - t)
def resident_crops(n):
    for _ in range(n): pipe(crops)
    pipe.synchronize(); torch.cuda.synchronize()
r_crops = timed(resident_crops, steps)
print(f"crops resident in HBM (bench.py's workload, no vertices): {r_crops:.0f} frames/s (B={B}, 3 batches in flight)")
def resident_frames(n):
    for s in range(n):
        lo = (s * B) % F
        pipe(ops.crop_frames(frames, bboxes[lo:lo + B], idx[lo:lo + B]))
    pipe.synchronize(); torch.cuda.synchronize()
r_frames = timed(resident_frames, steps)
print(f"frames resident in HBM -> crops -> pose -> SMPL joints -> REBA/RULA: {r_frames:.0f} frames/s = {r_frames / r_crops:.3f} of the resident-crop rate")
# the feed in the loop: the host frames live in the ring's pinned slots (a decoder would write there)
feed = FrameFeed(pipe, B, (H, W), dev, depth=int(os.environ.get('FEED_DEPTH', 0)) or None,
                 keys=() if os.environ.get('FEED_KEYS') == 'none' else ('euler', 'joint_cam', 'axis_angle', 'reba', 'rula', 'status'))
host_frames = frames.cpu().numpy(); del frames
for s in feed.slots:
    s.h_frames.copy_(torch.from_numpy(host_frames[:B])); s.h_bbox.copy_(torch.from_numpy(bboxes[:B]))
sink = [0.0]
def fed(n, memcpy=False):
    pending = []
    for s in range(n):
        i, hf, hb = feed.acquire()
        if pending and pending[0] == i:
            r = feed.result(pending.pop(0)); sink[0] += float(r["euler"][0, 0, 0]) if "euler" in r else 0.0      # the consumer touches the host result
        if memcpy:                       # a decoder that does NOT write into the ring: one pageable -> pinned copy per batch
            lo = (s * B) % F
            np.copyto(hf, host_frames[lo:lo + B]); np.copyto(hb, bboxes[lo:lo + B])
        feed.submit(i); pending.append(i)
    for i in pending: feed.result(i)
    feed.synchronize()
r_feed = timed(fed, steps)
gb = B * H * W * 3 / 1e9
print(f"FEED IN THE LOOP (pinned ring of uint8 frames -> H2D -> crop -> batch -> per-frame records D2H, {len(feed.slots)} slots): "
      f"{r_feed:.0f} frames/s = {r_feed / r_crops:.3f} of the resident-crop rate, {r_feed / r_frames:.3f} of the resident-frame rate; "
      f"{r_feed / B * gb:.1f} GB/s of frames over PCIe")
feed._skip_upload = True
r_noup = timed(fed, steps)
feed._skip_upload = False
print(f"... the same ring WITHOUT the frame upload (boxes only; crops from the slot's device frames): {r_noup:.0f} frames/s = {r_noup / r_crops:.3f}")
r_feed_cp = timed(lambda n: fed(n, True), steps)
print(f"... plus one host memcpy of the batch's frames into the ring per step (single thread): {r_feed_cp:.0f} frames/s = "
      f"{r_feed_cp / r_crops:.3f} of the resident-crop rate")
# what each leg could carry alone
h = torch.empty((B, H, W, 3), dtype=torch.uint8).pin_memory(); d = torch.empty((B, H, W, 3), dtype=torch.uint8, device=dev)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(50): d.copy_(h, non_blocking=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 50
print(f"H2D alone, pinned, {gb*1e3:.0f} MB per batch: {gb / dt:.1f} GB/s = {B / dt:.0f} frames/s")
