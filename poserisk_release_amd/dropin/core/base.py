"""Mirror of the hot-path half of lib/core/base.py::Predictor.

    predictor = Predictor(args)                                       main/run.py:31
    result, joint_cam, images, debug_result = predictor.get_pose_estimation_results(loader)   base.py:126
    reba_results = predictor.reba(result, joint_cam, add_info)        base.py:151
    final, scores, logs = predictor.post_processing(reba_results, ...)  base.py:153

`get_pose_estimation_results` is the reference's loop (base.py:211-240) with the per-frame Python
removed: every batch is one `pr_frames_forward` call on the GPU, results stay on the device until the
end of the loop, and with `torch.distributed` initialised every rank takes a contiguous frame shard and
the per-frame records are all-gathered once.  Video decoding and tracking (base.py:47-74) are outside the
accelerated path: `__call__` takes decoded frames + the tracker's dict, finds them in a
directory, or runs the reference's own front end when cv2 and multi_person_tracker are importable; it writes the
result text files, the score plots, the debug CSVs and (where cv2 is importable) the annotated mp4.
"""
import json
import os.path as osp

import numpy as np
import torch

from poserisk_release_amd import pipeline as pl
from poserisk_release_amd import synth
from poserisk_release_amd.hmr import hmr

from core.config import cfg
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


PRECISIONS = {'fp32': 'fp32', 'f32': 'fp32', 'float32': 'fp32', 'float': 'fp32',
              'bf16': 'bf16', 'bfloat16': 'bf16'}


def encoder_precision(args=None):
    """'fp32' | 'bf16': `args.dtype` when the caller's namespace has one (SURVEY.md section 5 asks for a --dtype next to
    the reference's flags; main/run.py:11-19 has none, so unmodified it is the config's), else cfg.SPIN.precision."""
    want = getattr(args, 'dtype', None) if args is not None else None
    if want is None:
        want = cfg.SPIN.get('precision', 'fp32')
    if isinstance(want, torch.dtype):
        want = str(want).replace('torch.', '')
    key = str(want).lower()
    if key not in PRECISIONS:
        raise ValueError(f"encoder precision {want!r} unknown: one of {sorted(set(PRECISIONS))} "
                         "(cfg.SPIN.precision / args.dtype)")
    return PRECISIONS[key]


def load_spin_model(smpl_mean_params=None, checkpoint=None, precision=None, conv_form=None):
    """base.py:81-84 with the reference's file locations (lib/core/config.py:45-47):

        spin_model = hmr(cfg.SPIN.SMPL_MEAN_PARAMS)
        checkpoint = torch.load(cfg.SPIN.checkpoint);  spin_model.load_state_dict(checkpoint['model'], strict=False)

    Both files are licensed downloads (reference README.md:36-37); a missing one is reported by name here, at
    construction, instead of as a KeyError inside the first forward."""
    mean_path = smpl_mean_params if smpl_mean_params is not None else cfg.SPIN.SMPL_MEAN_PARAMS
    ckpt_path = checkpoint if checkpoint is not None else cfg.SPIN.checkpoint
    for what, path, knob in (("SMPL mean parameters", mean_path, "cfg.SPIN.SMPL_MEAN_PARAMS"),
                             ("SPIN checkpoint", ckpt_path, "cfg.SPIN.checkpoint")):
        if isinstance(path, str) and not osp.isfile(path):
            raise FileNotFoundError(
                f"{what} not found at '{path}' ({knob}, lib/core/config.py:46-47; PoseRisk root taken as "
                f"'{cfg.root_dir}', override with $POSERISK_ROOT): download it as the reference's README.md:36-37 "
                "describes, or pass spin_model= / spin_checkpoint= / smpl_mean_params= to Predictor")
    model = hmr(mean_path, precision=precision if precision is not None else encoder_precision(),
                conv_form=conv_form if conv_form is not None else cfg.SPIN.get('conv_form', 'default'))
    import pickle
    try:
        ckpt = torch.load(ckpt_path, map_location='cpu')            # base.py:83 lacks map_location (Q23)
    except pickle.UnpicklingError:
        # torch >= 2.6 loads with weights_only=True by default; SPIN's checkpoint also pickles optimizer / scheduler
        # objects, which that mode refuses.  Only THAT refusal falls back to full unpickling (the reference's own
        # behaviour, base.py:83: the file is the user's licensed download); any other failure is raised as it is.
        ckpt = torch.load(ckpt_path, map_location='cpu', weights_only=False)
    if 'model' not in ckpt:
        raise KeyError(f"'{ckpt_path}' has no 'model' entry (base.py:84 reads checkpoint['model']); keys: {list(ckpt)[:8]}")
    missing, _ = model.load_state_dict(ckpt['model'], strict=False)
    if missing:     # the reference's strict=False would run on with torchvision's ImageNet init; there is none here
        raise KeyError(f"'{ckpt_path}': checkpoint['model'] lacks {len(missing)} of the encoder/regressor tensors, "
                       f"e.g. {missing[:4]}")
    return model


def default_information():
    """The `add_info` used when --info is not a file (base.py:140-142).  The reference points
    cfg.DATASET.default_information at lib/core/default_information.json while the file ships as
    main/default_information.json; either is read when present, else the same values built in."""
    for path in (cfg.DATASET.default_information, osp.join(cfg.root_dir, 'main', 'default_information.json')):
        if osp.isfile(path):
            with open(path, 'r') as f:
                return json.load(f)
    return synth.DEFAULT_INFO


class Predictor:
    def __init__(self, args, spin_model=None, smpl_model=None, batch_size=None, spin_checkpoint=None,
                 smpl_mean_params=None):
        """`Predictor(args)` as main/run.py:31 calls it (args fields gpu,type,input,info,output,visualize,debug,
        debug_joints,debug_frame): SMPL() from the reference's CWD-relative model directory (lib/utils/smpl.py:9),
        SPIN weights and mean parameters from cfg.SPIN.checkpoint / cfg.SPIN.SMPL_MEAN_PARAMS (base.py:80-84);
        FileNotFoundError names whichever file is absent.  Models may also be injected (tests, other locations).
        `batch_size`: frames per GPU call, default cfg.DATASET.hip_batch_size (64)."""
        self.device = torch.device('cuda') if torch.cuda.is_available() else torch.device('cpu')
        self.world_size = self._join_world(args)        # one process per GPU; may move self.device to this rank's GPU
        self.smpl_model = smpl_model if smpl_model is not None else SMPL()
        self.precision = encoder_precision(args)        # args.dtype, else cfg.SPIN.precision ('fp32' | 'bf16')
        if spin_model is None:
            spin_model = load_spin_model(smpl_mean_params, spin_checkpoint, precision=self.precision)
        self.spin_model = spin_model.to(self.device)
        self.batch_size = int(batch_size if batch_size is not None else cfg.DATASET.get('hip_batch_size', 64))
        # whole batches in flight (pipeline.FramePipeline): args.lanes, else cfg.DATASET.hip_lanes
        self.lanes = int(getattr(args, 'lanes', None) or cfg.DATASET.get('hip_lanes', 2))
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

    def _join_world(self, args):
        """One process per GPU.  The world size asked for: `args.world_size` when the namespace has one,
        else cfg.DATASET.hip_world_size, else (0) whatever the launcher exported as $WORLD_SIZE.  With more than one rank
        and no process group yet, this rank joins one over RCCL (`nccl`) on the GPU $LOCAL_RANK names -- main/run.py:26
        narrows CUDA_VISIBLE_DEVICES to --gpu first, so `--gpu 0,1,...` has to list a GPU per rank.  A mismatch between
        what was asked for and what the launcher started is an error, never a silent single-GPU run."""
        import os
        import torch.distributed as dist
        want = getattr(args, 'world_size', None)
        if want is None:
            want = cfg.DATASET.get('hip_world_size', 0)
        want = int(want or 0)
        launched = int(os.environ.get('WORLD_SIZE', '1'))
        if dist.is_available() and dist.is_initialized():
            have = dist.get_world_size()
            if want > 1 and want != have:
                raise RuntimeError(f"world size {want} asked for (args.world_size / cfg.DATASET.hip_world_size) but the "
                                   f"initialised process group has {have} ranks")
            return have
        if want > 1 and launched != want:
            raise RuntimeError(
                f"world size {want} asked for (args.world_size / cfg.DATASET.hip_world_size) but this process was started "
                f"with WORLD_SIZE={launched}: launch one rank per GPU, e.g. `python -m torch.distributed.run --nnodes=1 "
                f"--nproc-per-node {want} --master-addr 127.0.0.1 main/run.py --gpu {','.join(str(i) for i in range(want))} ...`")
        if launched <= 1:
            return 1
        if self.device.type != 'cuda':
            raise RuntimeError("several ranks need a GPU each (there is no CPU path)")
        local = int(os.environ.get('LOCAL_RANK', os.environ.get('RANK', '0')))
        if os.environ.get('POSERISK_SHARE_GPU') == '1':
            local = 0        # rehearsal only (a one-GPU box, gloo): every rank on cuda:0, like bench.py --share-gpu
        if local >= torch.cuda.device_count():
            raise RuntimeError(f"rank with LOCAL_RANK={local} sees {torch.cuda.device_count()} GPU(s): list one per rank in "
                               "--gpu (main/run.py:26 sets CUDA_VISIBLE_DEVICES from it)")
        self.device = torch.device('cuda', local)
        torch.cuda.set_device(self.device)
        pl.init_distributed(os.environ.get('POSERISK_DIST_BACKEND', 'nccl'), self.device)
        return dist.get_world_size()

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
            # the scorers run afterwards on all frames with the caller's add_info (base.py:151,168)
            self._pipe = pl.FramePipeline(self.spin_model, self.smpl_model.layer['neutral'], None, lanes=self.lanes)
        pipe = self._pipe
        eul, jc, aa, st, images = [], [], [], [], []
        # host batches cross PCIe from PINNED staging buffers (one per batch in flight + 1): a `.to(device, non_blocking=True)`
        # from pageable memory -- what a DataLoader without pin_memory yields -- is a synchronous copy that stalls the loop
        stage, n_stage, uploads = [], self.lanes + 1, 0
        with torch.no_grad():
            for batch in crop_dataloader:
                batch = torch.as_tensor(batch)
                if batch.device.type == 'cpu' and not batch.is_pinned():
                    k = uploads % n_stage
                    if len(stage) <= k:
                        stage.append([None, None])
                    buf, ev = stage[k]
                    if buf is None or buf.shape[0] < batch.shape[0] or buf.shape[1:] != batch.shape[1:] or buf.dtype != batch.dtype:
                        buf = torch.empty((max(batch.shape[0], self.batch_size),) + tuple(batch.shape[1:]), dtype=batch.dtype).pin_memory()
                        ev = None
                    if ev is not None:
                        ev.synchronize()                    # the upload that last used this buffer has left the host
                    buf[:batch.shape[0]].copy_(batch)
                    dbatch = buf[:batch.shape[0]].to(self.device, non_blocking=True)
                    stage[k] = [buf, torch.cuda.current_stream(self.device).record_event()]
                    uploads += 1
                else:
                    dbatch = batch.to(self.device, non_blocking=True)
                out = pipe(dbatch)
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
            add_info = default_information()
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

    def score_frames(self, frames, tracking_results, add_info=None, bgr=False, bbox_scale=None):
        """Decoded frames + tracker output -> scores, all on the GPU (BASELINE config 5 without the detector).

        frames: uint8[F,H,W,3] (torch CUDA tensor or numpy); tracking_results: multi_person_tracker's dict
        {id: {'bbox': [n,4] (cx,cy,w,h), 'frames': [n]}}.  Does what base.py:53-73 (track filter + target
        selection), demo_dataset.py:58-74 (crop) and base.py:126-182 (pose, scores) do."""
        from poserisk_release_amd import ops, tracks
        frames = torch.as_tensor(frames)
        if frames.device.type != 'cuda':
            frames = frames.to(self.device)
        if bbox_scale is None:
            bbox_scale = cfg.DATASET.bbox_scale                                   # base.py:121
        bboxes, fidx = tracks.target_track(tracking_results, frames.shape[0], cfg.DATASET.min_frame_ratio)
        if len(fidx) and (int(np.min(fidx)) < 0 or int(np.max(fidx)) >= frames.shape[0]):
            raise ValueError(f"tracking results name frames {int(np.min(fidx))}..{int(np.max(fidx))} but only "
                             f"{frames.shape[0]} frames were decoded (tracking.pkl does not belong to these frames)")
        # one process per GPU: this rank crops and scores its contiguous shard of the track
        world, rank = pl.world_and_rank()
        if world > 1:
            # the shard bounds and the gather's shapes follow from the track: a rank that selected another one (its own
            # tracker run, another file) would hang the collective or gather rows that are not its frames -- compare first
            import hashlib
            import torch.distributed as dist
            mine = (int(frames.shape[0]), len(fidx),
                    hashlib.sha256(np.ascontiguousarray(fidx).tobytes() + np.ascontiguousarray(bboxes).tobytes()).hexdigest())
            seen = [None] * world
            dist.all_gather_object(seen, mine)
            if any(s != seen[0] for s in seen):
                raise RuntimeError(f"the ranks do not hold the same target track (frames, track length, digest per rank: "
                                   f"{[s[:2] + (s[2][:8],) for s in seen]}): run the front end on one rank and broadcast its "
                                   "output (Predictor.__call__ does), or pass every rank the same frames and tracking dict")
        lo, hi = pl.shard_bounds(len(fidx), world, rank)

        def batches():
            for i in range(lo, hi, self.batch_size):
                j = min(i + self.batch_size, hi)
                yield ops.crop_frames(frames, bboxes[i:j], fidx[i:j].astype(np.int32), scale=bbox_scale, bgr=bgr)
        out = self.score_crops(batches(), add_info, n_total=len(fidx) if world > 1 else None)
        out['frames'] = fidx
        out['bboxes'] = bboxes
        return out

    # ---- base.py:273-282 ------------------------------------------------------------------------------------
    def visualize_joint_cam_mesh(self, debug_result, joint_cam, frames, output_path):
        """The --debug --debug_frame N outputs: `smpl_model.obj` (the frame's mesh in mm, zero betas, from the axis-angle
        with the root already overwritten, Q5) and `joint_3d.png`."""
        from poserisk_release_amd import reports
        idx = int(np.where(np.asarray(frames) == self.debug_frame)[0][0])
        pose = torch.as_tensor(debug_result[idx]).reshape(1, -1).float()
        verts, _ = self.smpl_model.layer['neutral'](pose, torch.zeros((1, 10)))
        mesh = verts.detach().cpu().numpy().astype(np.float32).reshape(-1, 3) * 1000
        reports.save_obj(mesh, self.smpl_model.face, osp.join(output_path, 'smpl_model.obj'))
        reports.save_joint_3d_plot(joint_cam[idx], self.smpl_model.skeleton, osp.join(output_path, 'joint_3d.png'),
                                   frame=self.debug_frame)

    # ---- main/run.py:31  predictor(args.input, args.info, args.output) -----------------------------------
    def load_front_end(self, input_path, output_path):
        """Decoded frames + tracker output for `input_path` -> (frames u8[F,H,W,3], bgr, fps, tracking dict).

        Video decoding and multi-person tracking are outside the accelerated path; this only finds them:
          1. a directory with `frames.npy` (uint8 [F,H,W,3] RGB) and `tracking.pkl` (multi_person_tracker's
             dict {id: {'bbox': [n,4] (cx,cy,w,h), 'frames': [n]}}), optional `fps.txt`;
          2. otherwise the reference's own front end when `cv2` and `multi_person_tracker` are importable:
             frames are decoded and resized as funcs_utils.get_images does (width <= 800, else height <= 450),
             written as JPEGs under <output>/tmp for the tracker (base.py:47-56) and read back, so the crops see
             the same JPEG-decoded pixels as the reference's CropDataset."""
        import pickle
        if osp.isdir(input_path) and osp.isfile(osp.join(input_path, 'frames.npy')):
            frames = np.load(osp.join(input_path, 'frames.npy'))
            with open(osp.join(input_path, 'tracking.pkl'), 'rb') as f:
                tracking = pickle.load(f)
            fps_file = osp.join(input_path, 'fps.txt')
            fps = float(open(fps_file).read()) if osp.isfile(fps_file) else 30.0
            return frames, False, fps, tracking
        try:
            import cv2
            from multi_person_tracker import MPT
        except ImportError as e:
            raise RuntimeError(
                f"{input_path!r} is not a directory with frames.npy + tracking.pkl, and the reference's front end "
                f"(cv2 video decoding, multi_person_tracker) is not importable here ({e}); decode and track outside, "
                "then call Predictor.score_frames(frames, tracking_results, info)") from e
        import os
        import shutil
        image_path = osp.join(output_path, 'tmp')
        shutil.rmtree(image_path, ignore_errors=True)
        os.makedirs(image_path, exist_ok=True)
        cap = cv2.VideoCapture(input_path)
        fps = cap.get(cv2.CAP_PROP_FPS)
        width, height = cap.get(cv2.CAP_PROP_FRAME_WIDTH), cap.get(cv2.CAP_PROP_FRAME_HEIGHT)
        if width > 800:
            width, height = 800, int(height * 800 / width)
        elif height > 450:
            width, height = int(width * 450 / height), 450
        n = 0
        while cap.isOpened():
            ok, frame = cap.read()
            if not ok:
                break
            cv2.imwrite(osp.join(image_path, '{0:09d}.jpg'.format(n)), cv2.resize(frame, (int(width), int(height))))
            n += 1
        cap.release()
        tracker = MPT(device=self.device, batch_size=8, display=False, detection_threshold=0.1, detector_type='yolo',
                      output_format='dict', yolo_img_size=416)
        tracking = tracker(image_path)
        frames = np.stack([cv2.imread(osp.join(image_path, '{0:09d}.jpg'.format(i))) for i in range(n)])
        shutil.rmtree(image_path, ignore_errors=True)
        return frames, True, fps, tracking

    def _front_end_on_rank0(self, input_path, output_path, world, rank):
        """Several ranks (one per GPU): the front end writes, runs the tracker in and deletes <output>/tmp, so it runs on
        rank 0 ONLY and its output is broadcast -- every rank then shards the SAME track (a tracker run per rank may select
        different tracks; one rank's rmtree would delete JPEGs another rank is still reading).  The metadata goes as one
        pickled object, the frames as one uint8 tensor (on this rank's GPU with RCCL, on the host with gloo)."""
        import torch.distributed as dist
        meta, err = [None], None
        frames = None
        if rank == 0:
            try:
                frames, bgr, fps, tracking = self.load_front_end(input_path, output_path)
                frames = torch.as_tensor(np.ascontiguousarray(frames))
                meta = [dict(shape=tuple(frames.shape), bgr=bool(bgr), fps=float(fps), tracking=tracking)]
            except Exception as e:               # the other ranks are waiting in the broadcast: tell them, then raise
                meta, err = [dict(error=f"{type(e).__name__}: {e}")], e
        dist.broadcast_object_list(meta, src=0)
        m = meta[0]
        if 'error' in m:
            raise err if err is not None else RuntimeError(f"rank 0's front end failed: {m['error']}")
        on_gpu = dist.get_backend() == 'nccl'
        if rank != 0:
            frames = torch.empty(m['shape'], dtype=torch.uint8, device=self.device if on_gpu else 'cpu')
        elif on_gpu:
            frames = frames.to(self.device)
        dist.broadcast(frames, src=0)
        return frames, m['bgr'], m['fps'], m['tracking']

    def __call__(self, input_path, info_path, output_path, frames=None, tracking_results=None, fps=30.0, bgr=False):
        """The reference's entry point (base.py:126-209) around the accelerated path: front end (given, found or
        the reference's own: `load_front_end`) -> crops, pose, scores on the GPU -> `reba_result.txt` /
        `rula_result.txt`, `<TITLE>_score.png`, `<TITLE>_video.mp4` (OpenCV drawing: only where cv2 is importable) and
        with `args.debug` the CSV logs under <output>/debug.  Returns the dict of `score_frames` plus `fps`."""
        import os
        from poserisk_release_amd import reports
        os.makedirs(output_path, exist_ok=True)
        world, rank = pl.world_and_rank()
        if frames is None or tracking_results is None:
            if world > 1:
                frames, bgr, fps, tracking_results = self._front_end_on_rank0(input_path, output_path, world, rank)
            else:
                frames, bgr, fps, tracking_results = self.load_front_end(input_path, output_path)
        if info_path and osp.isfile(str(info_path)):
            with open(info_path, 'r') as f:
                add_info = json.load(f)
        else:
            add_info = default_information()                   # base.py:140-142
        out = self.score_frames(frames, tracking_results, add_info, bgr=bgr)
        out['fps'] = fps
        if rank != 0:
            # every rank holds all frames' results after the gather; the reports (result txt, plots, mp4, debug CSVs and
            # OBJ) are ONE set of files in <output>: rank 0 writes them, the others would race it on the same names
            return out
        fidx = out['frames']
        timestamp = (0, fidx, int(frames.shape[0]))                # base.py:130 (torch tensor or numpy: both have .shape)
        debug_path = osp.join(output_path, 'debug')
        if self.debugging:
            os.makedirs(debug_path, exist_ok=True)
        if self.debugging and self.debug_frame is not None and self.debug_frame >= 0:
            # base.py:128-135: the --debug_frame branch dumps that frame's mesh and 3-D skeleton and stops there
            self.visualize_joint_cam_mesh(out['debug_result'], out['joint_cam'], fidx, debug_path)
            print("\n Debug files are saved in : ", debug_path)
            return out
        pose_str = reports.pose_to_str(out['result'])
        if self.debugging and self.debug_joints is not None:
            reports.save_pose_log_csv(debug_path, timestamp, pose_str, self.debug_joints, self.smpl_model.joints_name_upper)
        for title, scorer in (('REBA', self.reba), ('RULA', self.rula)):
            if title.lower() not in out:
                continue
            final, scores, logs, (level, name) = out[title.lower()]
            reports.save_score_plot(output_path, title, timestamp, scores)      # base.py:254-262
            if getattr(self, 'visualize', True):                                # base.py:156,173 (OpenCV only)
                fr = frames.cpu().numpy() if isinstance(frames, torch.Tensor) else np.asarray(frames)   # once, host side
                video = reports.write_annotated_video(output_path, title, fr if bgr else fr[..., ::-1], out['bboxes'],
                                                      timestamp, fps, scores, scorer.eval_items, logs)
                if video is None and not getattr(self, '_warned_no_cv2', False):
                    print("OpenCV (cv2) is not importable: the annotated mp4 is skipped, all other outputs are written")
                    self._warned_no_cv2 = True
            reports.write_result_txt(output_path, title, final, level, name)
            if self.debugging:
                reports.save_score_csv(debug_path, title, timestamp, scores, scorer.eval_items, logs, scorer.log)
        return out
