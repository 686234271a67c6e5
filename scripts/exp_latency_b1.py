import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from poserisk_release_amd import synth
from poserisk_release_amd.hmr import HMR
dev = torch.device('cuda', 0)
for prec in ('fp32', 'bf16'):
    m = HMR(max_batch=1, precision=prec).to(dev); m.load_state_dict(synth.hmr_state_dict(seed=1))
    x = torch.from_numpy(synth.crops(1, seed=3)).to(dev)
    for _ in range(20): m(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): m(x)
    e1.record(); torch.cuda.synchronize()
    gpu_ms = e0.elapsed_time(e1) / 200
    t0 = time.perf_counter()
    for _ in range(200):
        m(x); torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 200 * 1e3
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        m(x)
        torch.cuda.synchronize()
        with torch.cuda.graph(g): out = m(x)
    for _ in range(10): g.replay()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(200): g.replay()
    e1.record(); torch.cuda.synchronize()
    graph_ms = e0.elapsed_time(e1) / 200
    t0 = time.perf_counter()
    for _ in range(200):
        g.replay(); torch.cuda.synchronize()
    gwall = (time.perf_counter() - t0) / 200 * 1e3
    print(f"{prec} B=1 HMR forward: back-to-back {gpu_ms:.3f} ms/frame, with a sync per frame {wall:.3f} ms; hipGraph replay back-to-back {graph_ms:.3f}, with a sync {gwall:.3f}")
