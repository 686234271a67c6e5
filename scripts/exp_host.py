import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from poserisk_release_amd import synth
from poserisk_release_amd.hmr import HMR
dev = torch.device("cuda", 0)
sd = synth.hmr_state_dict(seed=1)
B = 64
x = torch.rand((B, 3, 224, 224), device=dev)
for n in (1, 4):
    m = HMR(max_batch=B).to(dev); m.load_state_dict(sd); m.set_streams(n)
    for _ in range(3): m(x)
    torch.cuda.synchronize()
    host = []
    t0 = time.perf_counter()
    for _ in range(10):
        a = time.perf_counter(); m(x); host.append(time.perf_counter() - a)
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t0) / 10
    print(f"streams={n}: host enqueue {sum(host)/10*1e3:.3f} ms/step, wall {tot*1e3:.3f} ms/step", flush=True)
    # one isolated step (GPU idle before): latency
    a = time.perf_counter(); m(x); torch.cuda.synchronize(); print(f"   single step latency {1e3*(time.perf_counter()-a):.3f} ms")
    m._release()
