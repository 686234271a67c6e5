"""Experiment: does running independent sub-batches on separate HIP streams fill tile-quantisation tails?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from poserisk_release_amd import synth
from poserisk_release_amd.hmr import HMR

dev = torch.device("cuda", 0)
sd = synth.hmr_state_dict(seed=1)
B = 64
x = torch.rand((B, 3, 224, 224), device=dev)

def bench(nsplit, iters=20):
    models = []
    for i in range(nsplit):
        m = HMR(max_batch=B // nsplit).to(dev); m.load_state_dict(sd); models.append(m)
    streams = [torch.cuda.Stream(dev) for _ in range(nsplit)]
    chunks = x.chunk(nsplit)
    def step():
        for m, s, c in zip(models, streams, chunks):
            with torch.cuda.stream(s):
                m(c)
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print(f"nsplit={nsplit}: {dt*1e3:.3f} ms/step  {B/dt:.0f} frames/s", flush=True)
    for m in models: m._release()

for n in (1, 2, 4):
    bench(n)
