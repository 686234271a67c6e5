"""Eager vs hipGraph replay of one whole batch (pr_frames_forward), one batch in flight.  usage: exp_graph.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from poserisk_release_amd import synth, pipeline as pl
from poserisk_release_amd.hmr import HMR
from poserisk_release_amd.smpl_layer import SMPLLayer
dev = torch.device("cuda", 0)
sd = synth.hmr_state_dict(seed=1); sm = synth.smpl_model(V=6890, seed=2)
for B in (1, 4, 16, 64):
    m = HMR(max_batch=B).to(dev); m.load_state_dict(sd)
    layer = SMPLLayer(sm, device=dev, max_batch=max(B, 16))
    pipe = pl.FramePipeline(m, layer, synth.EXAMPLE_INFO, with_verts=True)
    x = torch.rand((B, 3, 224, 224), device=dev)
    for _ in range(5): pipe(x)
    torch.cuda.synchronize()
    n = 200 if B <= 16 else 60
    t = time.perf_counter()
    for _ in range(n): pipe(x)
    torch.cuda.synchronize(); eager = (time.perf_counter() - t) / n
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): pipe(x)
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): g.replay()
    torch.cuda.synchronize(); graph = (time.perf_counter() - t) / n
    print(f"B={B}: eager {eager*1e3:.3f} ms/batch ({B/eager:.0f} f/s), graph {graph*1e3:.3f} ms/batch ({B/graph:.0f} f/s)", flush=True)
