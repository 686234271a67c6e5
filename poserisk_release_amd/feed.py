"""Host frames -> per-frame results with the FEED in the loop (BASELINE configs[4] without the detector).

The reference feeds its hot loop through a DataLoader of OpenCV crops, `batch.to(device)` and `.cpu().numpy()` per batch
(lib/core/base.py:219-233, data/demo_dataset.py:58-74).  Here the decoded frames of a batch cross PCIe once, as uint8, and
everything behind that is asynchronous:

    pinned host ring (uint8 frames + boxes, the decoder's destination)
      --copy stream (SDMA)-->  device frames
      --the batch's lane stream: pr_crop_frames -> pr_frames_forward (FramePipeline, `lanes` batches in flight) -> read-back of
        the per-frame records into pinned host memory (Euler degrees, joint_cam, axis-angle, REBA / RULA records, status)

ONE stream besides the lanes', ONE upload and ONE read-back per batch (frames and boxes share a pinned buffer; every per-frame
result of a lane is a view of one device blob); every hand-over is an event, the host never waits for the GPU except in
`result()`.  Measured (scripts/bench_crop.py -> profiles/r04_bench_feed.txt, B=64, three batches in flight): 0.92 of the
resident-crop rate; the same ring without the frame upload 0.975 -- the bound is the upload itself (69 MB per batch by SDMA
beside the kernels: -5 %), not the ring (-2.5 %), the host (a memcpy of the batch into the ring per step changes nothing) or
PCIe (56 GB/s alone, 16.6 GB/s used).
At 16 k frames/s the ring carries 17 GB/s of 450x800 frames up and 35 MB/s of results down (PCIe Gen5 x16: 63 GB/s).
"""
import numpy as np
import torch

from . import _lib
from .pipeline import FramePipeline

# status: the reference's two rotation asserts as bits (pr_pose_to_euler); crop_status: 1 = the frame index named no decoded
# frame (pr_crop_frames zero-fills that crop; the reference would raise from cv2.imread)
RESULT_KEYS = ("euler", "joint_cam", "axis_angle", "reba", "rula", "status", "crop_status")


class _Slot:
    pass


class FrameFeed:
    def __init__(self, pipe, batch, frame_hw, device, depth=None, keys=RESULT_KEYS, scale=1.2, bgr=False):
        """pipe: a FramePipeline (its lanes give the batches in flight); batch: frames per submission; frame_hw: (H, W) of the
        decoded frames; depth: ring slots (default lanes + 2: one being filled by the host, one in flight per lane, one being
        read back); keys: the per-frame results copied to the host."""
        self.pipe, self.B, self.dev = pipe, int(batch), torch.device(device)
        H, W = frame_hw
        self.keys = tuple(k for k in keys if k in ("euler", "joint_cam", "axis_angle", "status", "crop_status", "rotmat", "betas", "cam") or
                          (pipe.with_scores and k in ("reba", "rula")))
        self.scale, self.bgr = float(scale), int(bool(bgr))
        self.s_h2d = torch.cuda.Stream(self.dev)
        n = int(depth) if depth else len(pipe._lanes) + 2
        self.slots = []
        for _ in range(n):
            s = _Slot()
            # frames and boxes share one pinned buffer (and one device buffer): a whole batch is ONE upload
            nf = (self.B * H * W * 3 + 255) // 256 * 256
            s.h_buf = torch.empty((nf + self.B * 16,), dtype=torch.uint8).pin_memory()
            s.d_buf = torch.empty((nf + self.B * 16,), dtype=torch.uint8, device=self.dev)
            s.h_frames, s.d_frames = (b[:self.B * H * W * 3].view(self.B, H, W, 3) for b in (s.h_buf, s.d_buf))
            s.h_bbox, s.d_bbox = (b[nf:].view(torch.float32).view(self.B, 4) for b in (s.h_buf, s.d_buf))
            # zeros: a ragged last batch still runs the pipeline at the slot's full B (one shape resident, one graph), its
            # tail rows being whatever the slot held before
            s.crops = torch.zeros((self.B, 3, 224, 224), dtype=torch.float32, device=self.dev)
            s.h_blob, s.layout, s.layout_B = None, None, 0
            s.ev_h2d = s.ev_crop = s.ev_batch = s.ev_d2h = None
            s.n = 0
            self.slots.append(s)
        self.submitted = 0

    # ---- producer side -------------------------------------------------------------------------------------------
    def acquire(self):
        """-> (slot index, frames uint8[B,H,W,3] numpy view, bboxes f32[B,4] numpy view) of the next ring slot, pinned host
        memory for the decoder to write into.  Blocks only if that slot's previous upload has not left the host yet."""
        i = self.submitted % len(self.slots)
        s = self.slots[i]
        if s.ev_h2d is not None:
            s.ev_h2d.synchronize()
        return i, s.h_frames.numpy(), s.h_bbox.numpy()

    def submit(self, i, n=None):
        """Enqueue upload -> crop -> pose / SMPL / scores -> read-back of slot i's first n frames (default: a whole batch).
        Returns immediately."""
        s = self.slots[i]
        n = self.B if n is None else int(n)
        if not 0 < n <= self.B:
            raise ValueError(f"{n} frames in a slot of {self.B}")
        s.n = n
        # upload: the device frames may be overwritten once the crop kernel of the slot's previous round has read them
        if s.ev_crop is not None:
            self.s_h2d.wait_event(s.ev_crop)
        with torch.cuda.stream(self.s_h2d):
            if getattr(self, "_skip_upload", False):          # (scripts/bench_crop.py: what the upload itself costs the step)
                s.d_bbox[:n].copy_(s.h_bbox[:n], non_blocking=True)
            elif n == self.B:
                s.d_buf.copy_(s.h_buf, non_blocking=True)
            else:
                s.d_frames[:n].copy_(s.h_frames[:n], non_blocking=True)
                s.d_bbox[:n].copy_(s.h_bbox[:n], non_blocking=True)
            s.ev_h2d = self.s_h2d.record_event()
        # crop, batch and read-back on the lane's own stream; the crops tensor may be overwritten once the batch that read it
        # (maybe on another lane) has run
        st = self.pipe.next_stream(self.dev)
        st.wait_event(s.ev_h2d)
        if s.ev_batch is not None:
            st.wait_event(s.ev_batch)
        if s.ev_d2h is not None:
            st.wait_event(s.ev_d2h)                    # the slot's previous read-back (maybe on another lane) wrote h_blob
        # the crop kernel's status goes into the blob of the lane that will run this batch: it rides in the one read-back
        lane = self.pipe._lanes[self.pipe._next]
        crop_status = self.pipe._out(lane, self.B, self.dev)["crop_status"]
        if lane.reuse_after is not None:
            # crop_status lives in the lane's blob: a side-stream reader of the lane's previous outputs (release_after) must
            # be through before the crop kernel writes it, not only before the forward does
            st.wait_event(lane.reuse_after)
        with torch.cuda.stream(st):
            _, H, W, _ = s.d_frames.shape
            _lib.check(_lib.load().pr_crop_frames(s.d_frames.data_ptr(), n, H, W, self.bgr, None, s.d_bbox.data_ptr(), n,
                                                  self.scale, s.crops.data_ptr(), crop_status.data_ptr(), st.cuda_stream),
                       "pr_crop_frames")
            s.ev_crop = st.record_event()
            if n < self.B:
                # a ragged last batch: rows n.. still hold an earlier batch's crops -- blank them (and their crop status), so
                # that no stale frame re-enters the model and a reader of the whole blob sees zeros' results, not old frames'
                s.crops[n:].zero_()
                crop_status[n:].zero_()
            # always the slot's full B (a ragged last batch is sliced in result(); it costs a full batch's forward): the lane
            # keeps one shape resident, the slot's pinned blob is allocated once, a graph-mode lane replays one graph
            out = self.pipe(s.crops)                   # runs on `st` (its lane's stream = the current one here)
            assert out.lane is lane
            s.ev_batch = out.event if out.event is not None else st.record_event()
            # ONE device-to-host copy: every per-frame result of the lane is a view of lane.blob (pipeline.FramePipeline._out)
            if s.h_blob is None or s.h_blob.numel() != lane.blob.numel():
                s.h_blob = torch.empty((lane.blob.numel(),), dtype=torch.uint8).pin_memory()
            s.h_blob.copy_(lane.blob, non_blocking=True)
            s.layout, s.layout_B = lane.layout, out["status"].shape[0]
            s.ev_d2h = st.record_event()               # the lane's next batch is behind it on the same stream
        self.submitted += 1

    # ---- consumer side -------------------------------------------------------------------------------------------
    def result(self, i):
        """Wait for slot i's read-back -> {key: numpy view [n, ...]} (pinned host memory, valid until the slot is submitted
        again)."""
        s = self.slots[i]
        s.ev_d2h.synchronize()
        host = s.h_blob.numpy()
        out = {}
        for k in self.keys:
            p0, nb, shape, dt = s.layout[k]
            npdt = {torch.float32: np.float32, torch.float64: np.float64, torch.int32: np.int32}[dt]
            out[k] = host[p0:p0 + nb].view(npdt).reshape((s.layout_B,) + tuple(shape))[:s.n]
        return out

    def run(self, batches):
        """Convenience driver: `batches` yields (frames uint8[n,H,W,3], bboxes f32[n,4]) host arrays, n <= batch; yields the
        result dicts (copies) in order, keeping the ring full."""
        pending = []
        for frames, bboxes in batches:
            i, hf, hb = self.acquire()
            if pending and pending[0] == i:            # the ring is full: hand the oldest result out before reusing its slot
                yield {k: v.copy() for k, v in self.result(pending.pop(0)).items()}
            n = len(frames)
            np.copyto(hf[:n], frames)
            np.copyto(hb[:n], np.asarray(bboxes, np.float32))
            self.submit(i, n)
            pending.append(i)
        for i in pending:
            yield {k: v.copy() for k, v in self.result(i).items()}

    def synchronize(self):
        self.s_h2d.synchronize()
        self.pipe.synchronize()
        torch.cuda.synchronize(self.dev)
