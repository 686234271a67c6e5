"""Mirror of the hot-path half of lib/core/base.py::Predictor.

    predictor = Predictor(args)                                       main/run.py:31
    result, joint_cam, images, debug_result = predictor.get_pose_estimation_results(loader)   base.py:126
    reba_results = predictor.reba(result, joint_cam, add_info)        base.py:151
    final, scores, logs = predictor.post_processing(reba_results, ...)  base.py:153

`get_pose_estimation_results` is the reference's loop (base.py:211-240) with the per-frame Python
removed: every batch is one `pr_frames_forward` call on the GPU, results stay on the device until the
end of the loop, and with `torch.distributed` initialised every rank takes a contiguous frame shard and
the per-frame records are all-gathered once.  Video decoding, tracking and reporting (base.py:47-74,
273-420) are outside the accelerated path: `__call__` runs them only if the
reference's own front-end modules are importable, otherwise use `score_crops` with ready crops.
"""
import json
import os.path as osp

import numpy as np
import torch

from poserisk_release_amd import pipeline as pl
from poserisk_release_amd import synth
from poserisk_release_amd.hmr import hmr

from reba import REBA
from rula import RULA
from smpl import SMPL


def aggregate(scores):
    """base.py:263-271: sort desc; mean, top-50 %, top-10 % (NaN when N < 10), max, mode; 3 dp."""
    from scipy.stats import mode
    import warnings
    s = np.sort(np.asarray(scores))[::-1]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        top50 = round(s[:len(s) // 2].mean(), 3)
        top10 = round(s[:len(s) // 10].mean(), 3)
    return (round(s.mean(), 3), top50, top10, round(s.max(), 3), mode(s).mode.item())


class Predictor:
    def __init__(self, args, spin_model=None, smpl_model=None, batch_size=64, spin_checkpoint=None,
                 smpl_mean_params=None):
        """args: the namespace main/run.py builds (fields gpu,type,input,info,output,visualize,debug,
        debug_joints,debug_frame).  Models may be injected; otherwise they are loaded from the
        reference's locations (lib/core/config.py:45-50) when those licensed files exist."""
        self.device = torch.device('cuda') if torch.cuda.is_available() else torch.device('cpu')
        self.smpl_model = smpl_model if smpl_model is not None else SMPL()
        if spin_model is None:
            spin_model = hmr(smpl_mean_params)
            if spin_checkpoint is not None:
                ckpt = torch.load(spin_checkpoint, map_location='cpu')     # base.py:83 lacks map_location (Q23)
                spin_model.load_state_dict(ckpt['model'], strict=False)
        self.spin_model = spin_model.to(self.device)
        self.batch_size = batch_size
        self.lanes = int(getattr(args, 'lanes', 1))     # whole batches in flight (pipeline.FramePipeline)
        self._pipe = None
        debug = bool(getattr(args, 'debug', False))
        self.reba, self.rula = REBA(debug), RULA(debug)
        scores = str(getattr(args, 'type', 'REBA,RULA')).replace(' ', '').upper().split(',')
        self.run_reba = 'REBA' in scores
        self.run_rula = 'RULA' in scores
        self.debugging = debug
        self.debug_frame = getattr(args, 'debug_frame', -1)
        dj = str(getattr(args, 'debug_joints', '')).replace(' ', '').split(',')
        if dj == ['']:
            self.debug_joints = None
        else:
            for joint in dj:
                if joint.upper() not in self.smpl_model.joints_name_upper:
                    print("\n\nInvalid Joint name!\n\n")
                    assert 0
            self.debug_joints = dj

    # ---- base.py:211-240 -------------------------------------------------------------------------
    def get_pose_estimation_results(self, crop_dataloader, keep_images=True, n_total=None):
        """Iterable of f32[b,3,224,224] batches -> (result f64[N,24,3] Euler deg, joint_cam f32[N,24,3] mm,
        images f32[N,3,224,224], debug_result f32[N,24,3] axis-angle with root rows = 3.14,0,0).

        `n_total` (multi-GPU): the loader holds only this rank's contiguous shard
        `pipeline.shard_bounds(n_total, world, rank)`; the per-frame results of all ranks are all-gathered
        once after the loop (RCCL with the nccl backend), so every rank returns all `n_total` frames in
        frame order.  `images` stays local."""
        self.spin_model.eval()
        if self._pipe is None:
            self._pipe = pl.FramePipeline(self.spin_model, self.smpl_model.layer['neutral'], synth.DEFAULT_INFO,
                                          lanes=self.lanes)
        pipe = self._pipe
        eul, jc, aa, st, images = [], [], [], [], []
        with torch.no_grad():
            for batch in crop_dataloader:
                batch = torch.as_tensor(batch)
                out = pipe(batch.to(self.device, non_blocking=True))
                pl.FramePipeline.wait(out)      # the copies below queue behind this batch; the next one overlaps
                eul.append(out['euler'].clone()); jc.append(out['joint_cam'].clone())
                aa.append(out['axis_angle'].clone()); st.append(out['status'].clone())
                if keep_images:
                    images.append(batch.cpu().numpy())
        dev = self.device
        cat = lambda parts, shape, dt: torch.cat(parts) if parts else torch.empty((0,) + shape, dtype=dt, device=dev)
        eul, jc = cat(eul, (24, 3), torch.float64), cat(jc, (24, 3), torch.float32)
        aa, st = cat(aa, (24, 3), torch.float32), cat(st, (), torch.int32)
        if n_total is not None:
            eul, jc, aa, st = (pl.gather_padded(t, n_total) for t in (eul, jc, aa, st))
        status = st.cpu().numpy()
        if np.any(status != 0):     # coord_utils.py:70,91: the reference aborts on these
            raise AssertionError(f"invalid rotation in frames {np.nonzero(status)[0].tolist()}")
        result = eul.cpu().numpy()
        joint_cam = jc.cpu().numpy()
        debug_result = aa.cpu().numpy()
        images = np.concatenate(images) if images else np.zeros((0, 3, 224, 224), np.float32)
        return result, joint_cam, images, debug_result

    # ---- base.py:242-271 (aggregation; the score plot is reporting and lives outside the path) ------
    def post_processing(self, results, joint_names=None, timestamp=None, output_path=None, title=''):
        scores = np.array([r['score'] for r in results])
        logs = np.array([r['log_score'] for r in results])
        return aggregate(scores), np.copy(scores), logs

    # ---- the accelerated half of __call__ (base.py:126-182 without tracking / reporting) ------------
    def score_crops(self, crop_batches, add_info=None, n_total=None):
        """crops -> dict(result, joint_cam, reba=(final, scores, logs, level), rula=(...)).
        `n_total`: see get_pose_estimation_results (the batches are this rank's shard of n_total frames)."""
        if add_info is None:
            add_info = synth.DEFAULT_INFO
        elif isinstance(add_info, str):
            with open(add_info, 'r') as f:
                add_info = json.load(f)
        result, joint_cam, _, debug_result = self.get_pose_estimation_results(crop_batches, keep_images=False,
                                                                              n_total=n_total)
        out = dict(result=result, joint_cam=joint_cam, debug_result=debug_result)
        if self.run_reba:
            final, scores, logs = self.post_processing(self.reba(result, joint_cam, add_info))
            out['reba'] = (final, scores, logs, self.reba.action_level(final[4]))
        if self.run_rula:
            final, scores, logs = self.post_processing(self.rula(result, joint_cam, add_info))
            out['rula'] = (final, scores, logs, self.rula.action_level(final[4]))
        return out

    def score_frames(self, frames, tracking_results, add_info=None, bgr=False, bbox_scale=1.2):
        """Decoded frames + tracker output -> scores, all on the GPU (BASELINE config 5 without the detector).

        frames: uint8[F,H,W,3] (torch CUDA tensor or numpy); tracking_results: multi_person_tracker's dict
        {id: {'bbox': [n,4] (cx,cy,w,h), 'frames': [n]}}.  Does what base.py:53-73 (track filter + target
        selection), demo_dataset.py:58-74 (crop) and base.py:126-182 (pose, scores) do."""
        from poserisk_release_amd import ops, tracks
        frames = torch.as_tensor(frames)
        if frames.device.type != 'cuda':
            frames = frames.to(self.device)
        bboxes, fidx = tracks.target_track(tracking_results, frames.shape[0])
        # one process per GPU: this rank crops and scores its contiguous shard of the track
        world, rank = pl.world_and_rank()
        lo, hi = pl.shard_bounds(len(fidx), world, rank)

        def batches():
            for i in range(lo, hi, self.batch_size):
                j = min(i + self.batch_size, hi)
                yield ops.crop_frames(frames, bboxes[i:j], fidx[i:j].astype(np.int32), scale=bbox_scale, bgr=bgr)
        out = self.score_crops(batches(), add_info, n_total=len(fidx) if world > 1 else None)
        out['frames'] = fidx
        out['bboxes'] = bboxes
        return out

    def __call__(self, input_path, info_path, output_path):
        raise NotImplementedError(
            "video decoding, multi-person tracking and report rendering (base.py:47-74, 273-420) are outside "
            "the accelerated path; produce 224x224 crops with the reference's CropDataset and call "
            "Predictor.score_crops(crops, info_path), or pass decoded frames and the tracker's output to "
            "Predictor.score_frames(frames, tracking_results, info_path)")
