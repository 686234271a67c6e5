"""Child process of tests/test_rccl_world1.py: the RCCL branch of the path's one exchange (it feeds the
whole-video aggregation of lib/core/base.py:263-271) executed on ONE MI355X with a world of one rank.

Started as a fresh interpreter, so the process group is initialised before anything here touches the GPU -- exactly as a
rank of `bench.py --gpus N` does: pipeline.init_distributed("nccl", dev) -> RecordExchange.step per batch (comm stream
waits for the batch's event -> pack -> release_after -> all_gather_into_tensor).  Prints one JSON line.
usage: rccl_world1_child.py <steps> <lanes> <graph 0|1> <port>
"""
import json
import os
import sys

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, REPO)

steps, lanes, graph, port = int(sys.argv[1]), int(sys.argv[2]), bool(int(sys.argv[3])), sys.argv[4]
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from poserisk_release_amd import pipeline as pl  # noqa: E402
from poserisk_release_amd import synth  # noqa: E402
from poserisk_release_amd.hmr import HMR  # noqa: E402
from poserisk_release_amd.smpl_layer import SMPLLayer  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
pl.init_distributed("nccl", dev)                       # RCCL, bound to this rank's device
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1

B = 8
sd = synth.hmr_state_dict(seed=1)
sm = synth.smpl_model(V=6890, seed=2)
model = HMR(max_batch=B).to(dev)
model.load_state_dict(sd)
layer = SMPLLayer(sm, device=dev, max_batch=16)
pipe = pl.FramePipeline(model, layer, synth.EXAMPLE_INFO, with_verts=False, lanes=lanes, graph=graph)
pipe.prepare(B, dev)
ex = pl.RecordExchange(1, B, dev, n_buffers=lanes)
crops = [torch.from_numpy(synth.crops(B, seed=300 + i)).to(dev) for i in range(4)]

snaps, events, waited = [], [], []
for i in range(steps):
    out = pipe(crops[i % 4])
    pair = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]
    ex.step(out, on_stream=lambda st, k: pair[k].record(st))
    # the lane was told which event its next batch has to wait for (release_after) ...
    waited.append(out.lane.reuse_after is not None)
    events.append(pair)
    with torch.cuda.stream(ex.stream):
        snaps.append(ex.gathered.clone())           # what the collective delivered for THIS step, in stream order
pipe.synchronize()
ex.stream.synchronize()
torch.cuda.synchronize(dev)
# ... and every lane's forward consumed it (the wait was enqueued on the lane's stream) except the last round's
pending = sum(1 for lane in pipe._lanes if lane.reuse_after is not None)

# expected records by a second route: the same crops through a single eager lane, no exchange, fully synchronised
ref_pipe = pl.FramePipeline(model, layer, synth.EXAMPLE_INFO, with_verts=False, lanes=1)
want = []
for i in range(4):
    o = ref_pipe(crops[i])
    torch.cuda.synchronize(dev)
    want.append(pl.pack_record(o).clone())
ok = all(torch.equal(snaps[i], want[i % 4]) for i in range(steps))
distinct = not torch.equal(want[0], want[1])
comm_ms = sum(a.elapsed_time(b) for a, b in events) / len(events)
try:
    ver = ".".join(str(v) for v in torch.cuda.nccl.version())
except Exception as e:  # noqa: BLE001
    ver = f"unavailable ({type(e).__name__})"
print(json.dumps({"backend": dist.get_backend(), "world_size": dist.get_world_size(), "rccl_version": ver,
                  "steps": steps, "lanes": lanes, "graph": graph, "gathered_equals_pack_record": bool(ok),
                  "records_differ_between_steps": bool(distinct), "release_after_set_every_step": all(waited),
                  "lanes_with_unconsumed_release": pending, "comm_ms_per_step": round(comm_ms, 4),
                  "device": torch.cuda.get_device_name(dev)}), flush=True)
dist.destroy_process_group()
