import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from poserisk_release_amd import synth
from poserisk_release_amd.hmr import HMR
dev = torch.device("cuda", 0)
sd = synth.hmr_state_dict(seed=1)
B = 64
x = torch.rand((B, 3, 224, 224), device=dev)
def bench(nsplit, join, iters=20):
    models = []
    for i in range(nsplit):
        m = HMR(max_batch=B // nsplit).to(dev); m.load_state_dict(sd); m.set_streams(1); models.append(m)
    streams = [torch.cuda.Stream(dev) for _ in range(nsplit)]
    chunks = x.chunk(nsplit)
    main = torch.cuda.current_stream(dev)
    def step():
        if join:
            ev = main.record_event()
        for m, s, c in zip(models, streams, chunks):
            if join: s.wait_event(ev)
            with torch.cuda.stream(s):
                m(c)
            if join: main.wait_event(s.record_event())
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print(f"nsplit={nsplit} join={join}: {dt*1e3:.3f} ms/step  {B/dt:.0f} frames/s", flush=True)
    for m in models: m._release()
bench(1, False); bench(4, False); bench(4, True); bench(2, True)
