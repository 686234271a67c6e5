"""The per-batch driver of the hot path: crops -> everything `get_pose_estimation_results`
(lib/core/base.py:211-240) and the two scorers (base.py:151,168) produce, in one C-ABI call,
with no host synchronisation and no per-frame Python.

Multi-GPU: frames are independent, so every rank runs this on its own contiguous
shard; `gather_frames` is the single exchange before whole-video aggregation (base.py:263-271).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

# per-frame record gathered across ranks: rotmat 216 + betas 10 + cam 3 (the "SMPL params", 916 B)
RECORD_FLOATS = 216 + 10 + 3


class _Lane:
    """One in-flight batch: its own encoder / SMPL handles (workspaces), stream and output buffers."""

    def __init__(self, hmr_model, smpl_layer, stream):
        self.hmr, self.smpl, self.stream = hmr_model, smpl_layer, stream
        self.bufs = {}
        self.done = None
        self.reuse_after = None     # event a consumer on another stream recorded after reading this lane's outputs
        self.graph = None           # graph mode: (key, hipGraph of one pr_frames_forward, its input tensor, its outputs)


class BatchOut(dict):
    """The output tensors of one batch by name; `event` (batches in flight: recorded behind the batch on its lane's
    stream) and `lane` ride along as attributes so that iterating the dict yields tensors only.  `lane.blob` is the uint8
    device buffer all per-frame results (everything but the vertices) are views of; `lane.layout` maps a name to its
    (offset, bytes, per-frame shape, dtype) in it."""
    event = None
    lane = None


class FramePipeline:
    """crops -> everything the reference's loop produces, one C-ABI call per batch.

    lanes > 1 keeps that many WHOLE batches in flight on separate HIP streams (round-robin), each with
    its own workspaces: consecutive batches overlap on the GPU, which fills the tile-quantisation tails
    and launch gaps of the ~75 kernels per batch (+15 % frames/s at B=64, bit-identical results).  The
    returned tensors belong to the lane and are valid after `wait(out)` (or `synchronize()`); they are
    reused when the same lane comes round again.
    """

    def __init__(self, hmr_model, smpl_layer, add_info, with_verts=False, lanes=1, graph=False):
        """add_info: the dict of additional_information.json, or None to leave REBA / RULA out of the batch call
        (a caller that scores all frames afterwards with its own add_info, like Predictor).
        graph: every lane captures its batch (the ~90 launches of one pr_frames_forward) into a hipGraph the first time it
        sees a batch size and replays it afterwards -- one host call per batch instead of ~90 launches (the per-rank host
        budget when eight ranks share one host: DESIGN.md 6).  Same kernels, same order: the same bits.  The crops are copied
        into the lane's captured input tensor on the lane's stream (a device-to-device copy of the batch)."""
        self.graph = bool(graph)
        self.with_verts = with_verts
        self.with_scores = add_info is not None
        self._reba = _lib.reba_info_struct(add_info["REBA"]) if self.with_scores else None
        self._rula = _lib.rula_info_struct(add_info["RULA"]) if self.with_scores else None
        self.hmr, self.smpl = hmr_model, smpl_layer
        self._lanes = [_Lane(hmr_model, smpl_layer, None)]
        for _ in range(1, int(lanes)):
            self._lanes.append(_Lane(hmr_model.clone(), smpl_layer.clone(), None))
        for lane in self._lanes:
            lane.hmr.set_concurrency(len(self._lanes))      # persistent kernels leave room for the other lanes' kernels
        self._next = 0

    def prepare(self, B, dev):
        """Create every lane's handles, stream and buffers now (weight packing takes seconds per handle),
        so that no lane is built lazily inside a timed or latency-sensitive loop."""
        dev = torch.device(dev)
        for lane in self._lanes:
            if (len(self._lanes) > 1 or self.graph) and lane.stream is None:
                lane.stream = torch.cuda.Stream(dev)
            lane.hmr.to(dev)._ensure(B)
            lane.smpl.to(dev)._ensure()
            self._out(lane, B, dev)
        torch.cuda.synchronize(dev)

    def _out(self, lane, B, dev):
        key = (B, str(dev))
        if key not in lane.bufs:
            # every per-frame result is a view into ONE device blob (lane.blob, uint8), so that a consumer on the host needs one
            # device-to-host copy per batch (feed.FrameFeed), not one per tensor; the vertices (83 KB per frame) stay apart
            spec = [("rotmat", (24, 3, 3), torch.float32), ("betas", (10,), torch.float32), ("cam", (3,), torch.float32),
                    ("axis_angle", (24, 3), torch.float32), ("euler", (24, 3), torch.float64),
                    ("joint_cam", (24, 3), torch.float32), ("status", (), torch.int32),
                    # not written by pr_frames_forward: a front end that crops on the GPU (feed.FrameFeed) points
                    # pr_crop_frames' status here, so that a bad frame index travels with the batch's one read-back
                    ("crop_status", (), torch.int32)]
            if self.with_scores:
                spec += [("reba", (10,), torch.int32), ("rula", (12,), torch.int32)]
            offs, pos = {}, 0
            for name, shape, dt in spec:
                nbytes = B * int(np.prod(shape, dtype=np.int64)) * torch.empty((), dtype=dt).element_size()
                offs[name] = (pos, nbytes, shape, dt)
                pos = (pos + nbytes + 255) // 256 * 256
            blob = torch.zeros((max(pos, 256),), dtype=torch.uint8, device=dev)
            o = {name: blob[p0:p0 + nb].view(dt).view((B,) + shape) for name, (p0, nb, shape, dt) in offs.items()}
            lane.blob, lane.layout = blob, offs
            if self.with_verts:
                o["verts"] = torch.empty((B, lane.smpl.num_verts, 3), dtype=torch.float32, device=dev)
            lane.bufs = {key: o}  # keep one shape resident
        return lane.bufs[key]

    def next_stream(self, dev):
        """The stream the NEXT forward() will run its batch on (its lane's own stream, or the current stream for a single
        lane without graph mode): a producer that enqueues the batch's input there needs no extra stream and no event."""
        lane = self._lanes[self._next]
        if len(self._lanes) > 1 or self.graph:
            if lane.stream is None:
                lane.stream = torch.cuda.Stream(torch.device(dev))
            return lane.stream
        return torch.cuda.current_stream(torch.device(dev))

    def forward(self, crops):
        """crops f32[B,3,224,224] on the GPU -> dict of device tensors (reused across calls of equal B)."""
        if crops.device.type != "cuda":
            raise _lib.PoseRiskHipError("crops must be on the GPU")
        x = crops.contiguous().float()
        B = x.shape[0]
        dev = x.device
        lane = self._lanes[self._next]
        self._next = (self._next + 1) % len(self._lanes)
        multi = len(self._lanes) > 1 or self.graph      # graph mode: capture needs a stream of its own, even for one lane
        if multi and lane.stream is None:
            lane.stream = torch.cuda.Stream(dev)
        lane.hmr.to(dev)._ensure(B)
        lane.smpl.to(dev)._ensure()     # both handles exist BEFORE anything (a graph key, a capture) looks at them
        if getattr(lane.hmr, "_concurrency", 1) != len(self._lanes):   # the model may serve another pipeline too (bench.py)
            lane.hmr.set_concurrency(len(self._lanes))
        o = self._out(lane, B, dev)
        fo = _lib.FramesOut(o["rotmat"].data_ptr(), o["betas"].data_ptr(), o["cam"].data_ptr(),
                            o["axis_angle"].data_ptr(), o["euler"].data_ptr(), o["joint_cam"].data_ptr(),
                            o["verts"].data_ptr() if self.with_verts else None,
                            o["reba"].data_ptr() if self.with_scores else None,
                            o["rula"].data_ptr() if self.with_scores else None, o["status"].data_ptr())
        if multi:
            lane.stream.wait_stream(torch.cuda.current_stream(dev))   # crops were produced there
            stream = lane.stream
        else:
            stream = torch.cuda.current_stream(dev)
        if lane.reuse_after is not None:      # a side-stream reader of the previous outputs (release_after)
            stream.wait_event(lane.reuse_after)
            lane.reuse_after = None
        timing = getattr(self, "record_times", None)
        if timing is not None:                # measurement only (bench.py's lanes_overlap): events around this batch
            t0 = torch.cuda.Event(enable_timing=True)
            t0.record(stream)

        def launch(src, on):
            _lib.check(_lib.load().pr_frames_forward(lane.hmr.handle, lane.smpl.handle, src.data_ptr(), B,
                                                     C.byref(self._reba) if self.with_scores else None,
                                                     C.byref(self._rula) if self.with_scores else None, C.byref(fo),
                                                     on.cuda_stream), "pr_frames_forward")

        if self.graph:
            # the capture bakes in the handles' device pointers (activations, weights, workspaces) and the output blob's:
            # any of them being reallocated (a regrown or reloaded model, set_streams, new buffers) forces a recapture
            # ... and so does launch-shaping state that is not a pointer: the concurrency hint resizes the persistent kernels'
            # grids (same bits; a graph captured under another hint would replay the old grids and time the wrong setting)
            key = (B, str(dev), id(lane.hmr), lane.hmr.generation, id(lane.smpl), lane.smpl.generation,
                   lane.blob.data_ptr(), o["verts"].data_ptr() if self.with_verts else 0, self.with_scores,
                   getattr(lane.hmr, "_concurrency", 1), getattr(lane.hmr, "_streams", None))
            if lane.graph is None or lane.graph[0] != key:
                static_x = torch.empty_like(x)
                with torch.cuda.stream(stream):
                    static_x.copy_(x, non_blocking=True)
                    launch(static_x, stream)                  # eager once: every kernel loaded, every lazy allocation done
                stream.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=stream):
                    launch(static_x, torch.cuda.current_stream(dev))
                lane.graph = (key, g, static_x)
            _, g, static_x = lane.graph
            with torch.cuda.stream(stream):
                static_x.copy_(x, non_blocking=True)
                g.replay()
        else:
            launch(x, stream)
        if timing is not None:
            t1 = torch.cuda.Event(enable_timing=True)
            t1.record(stream)
            timing.append((t0, t1))
        o = BatchOut(o)
        o.lane = lane
        if multi:
            x.record_stream(stream)
            lane.done = o.event = stream.record_event()
        return o

    __call__ = forward

    @staticmethod
    def wait(out, stream=None):
        """Make `stream` (default: the current stream) wait for the batch that produced `out`."""
        ev = getattr(out, "event", None)
        if ev is not None:
            (stream or torch.cuda.current_stream()).wait_event(ev)
        elif stream is not None:                 # single lane: the batch ran on the current stream
            stream.wait_stream(torch.cuda.current_stream())

    @staticmethod
    def release_after(out, stream):
        """Tell the lane that produced `out` that `stream` (not the current stream) has read its tensors up to
        this point: the lane's next batch waits for that before overwriting them.  Readers on the current
        stream need nothing: a lane's next batch already waits for the current stream."""
        out.lane.reuse_after = stream.record_event()

    def synchronize(self):
        for lane in self._lanes:
            if lane.stream is not None:
                lane.stream.synchronize()


def shard_bounds(n_frames, world_size, rank):
    """Contiguous frame range of `rank`: [r*N/W, (r+1)*N/W)."""
    return (rank * n_frames) // world_size, ((rank + 1) * n_frames) // world_size


def world_and_rank(group=None):
    """(world_size, rank) of the initialised process group, (1, 0) without one."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def pack_record(out):
    """Per-frame SMPL-parameter record f32[B,229] = rotmat | betas | cam."""
    B = out["rotmat"].shape[0]
    return torch.cat([out["rotmat"].reshape(B, 216), out["betas"], out["cam"]], dim=1).contiguous()


def pack_record_into(out, dst):
    """Same, written into a preallocated f32[B,229] tensor (no allocation inside the timed loop)."""
    B = out["rotmat"].shape[0]
    dst[:, :216].copy_(out["rotmat"].reshape(B, 216))
    dst[:, 216:226].copy_(out["betas"])
    dst[:, 226:].copy_(out["cam"])
    return dst


def all_gather_rows(gathered, local, group=None):
    """gathered[W*n, ...] <- concatenation over ranks of local[n, ...]: RCCL `all_gather_into_tensor` on the
    nccl backend; gloo (CPU tests, single-GPU rehearsals) takes the list form."""
    import torch.distributed as dist
    if dist.get_backend(group) == "gloo":
        W = dist.get_world_size(group)
        parts = list(gathered.view((W, local.shape[0]) + tuple(local.shape[1:])).unbind(0))
        dist.all_gather(parts, local.contiguous(), group=group)
    else:
        dist.all_gather_into_tensor(gathered, local.contiguous(), group=group)
    return gathered


def init_distributed(backend, dev):
    """One process per GPU: `nccl` (= RCCL on ROCm, the collectives run over xGMI) bound to this rank's device, or `gloo` for
    CPU rehearsals.  Rendezvous on 127.0.0.1 unless the launcher says otherwise."""
    import os
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)


class RecordExchange:
    """The one exchange of the path, once per batch: every rank's per-frame SMPL-parameter records
    (f32[B, 229] = 916 B per frame) all-gathered into f32[W * B, 229], on a side stream so that it overlaps the next batch.

    Per step, in this order on the side stream: wait for the batch's event -> pack its record -> tell the lane its outputs
    have been read (its next batch may overwrite them) -> all_gather_into_tensor.  `records` is a ring as deep as the
    batches in flight, so a record is not rewritten while its collective may still read it.
    `stream` / `stream_context` are injectable for the CPU rehearsal of the call sequence (tests/test_host_cpu.py)."""

    def __init__(self, world, B, dev, n_buffers, stream=None, stream_context=None, group=None):
        self.world, self.B, self.group = world, B, group
        self.stream = stream if stream is not None else torch.cuda.Stream(dev)
        self._ctx = stream_context if stream_context is not None else torch.cuda.stream
        self.gathered = torch.empty((world * B, RECORD_FLOATS), dtype=torch.float32, device=dev)
        self.records = [torch.empty((B, RECORD_FLOATS), dtype=torch.float32, device=dev) for _ in range(max(int(n_buffers), 1))]
        self.steps = 0

    def step(self, out, on_stream=None):
        """Exchange the records of batch `out`; on_stream(stream) -> (before, after) hooks for timing events (bench.py)."""
        rec = self.records[self.steps % len(self.records)]
        self.steps += 1
        FramePipeline.wait(out, self.stream)
        with self._ctx(self.stream):
            if on_stream is not None:
                on_stream(self.stream, 0)
            pack_record_into(out, rec)
            FramePipeline.release_after(out, self.stream)   # the lane may overwrite `out` once this has run
            all_gather_rows(self.gathered, rec, self.group)
            if on_stream is not None:
                on_stream(self.stream, 1)
        return rec

    def last_record(self):
        return self.records[(self.steps - 1) % len(self.records)]


def gather_frames(local, n_total, group=None):
    """All-gather equal-sized (padded) per-rank tensors [n_pad, ...] and drop the padding.

    One collective per shard (RCCL over xGMI with the nccl backend, gloo on CPU); frame order is
    rank order because shards are contiguous.  `local` must already be padded to ceil(N/W) rows.
    """
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local[:n_total]
    W = dist.get_world_size(group)
    gathered = torch.empty((W * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    all_gather_rows(gathered, local, group)
    n_pad = local.shape[0]
    parts = []
    for r in range(W):
        lo, hi = shard_bounds(n_total, W, r)
        parts.append(gathered[r * n_pad: r * n_pad + (hi - lo)])
    return torch.cat(parts, dim=0)


def gather_padded(local, n_total, group=None):
    """`gather_frames` for a rank's UNPADDED shard rows: pads to ceil(N/W) first."""
    world, _ = world_and_rank(group)
    if world == 1:
        return local[:n_total]
    n_pad = -(-n_total // world)
    padded = torch.zeros((n_pad,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    return gather_frames(padded, n_total, group)
