"""The per-frame hot path arranged exactly as the reference executes it.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates the driver loop
``Predictor.get_pose_estimation_results`` (``lib/core/base.py:211-240``) followed
by the two scorers (``base.py:151,168``): encoder in batches of ``cfg.DATASET.batch_size``
(= 8, ``lib/core/config.py:32``), then per-frame Python loops for the rotation
conversions, a batch-1 SMPL forward per frame, REBA and RULA.  This arrangement is
what ``bench.py`` times as the ``cpu_baseline`` ("port").  PINNED since round 6: the reference's own
``get_pose_estimation_results`` run on ``oracle.hmr_ref`` + its own ``SMPL_Layer`` (tests/golden/make_golden.py::
gen_driver_loop) gives this function's Euler angles and axis-angle array bit for bit and its joint_cam within 3e-3 mm
(tests/test_oracle_golden.py::test_pipeline_ref_matches_the_references_driver_loop).
"""
import time

import numpy as np
import torch

from . import coord_ref, reba_ref, rula_ref, smpl_ref


def run(hmr_model, smpl_model, crops, add_info, batch_size=8, timings=None, per_frame_scorers=False):
    """crops f32[N,3,224,224] in [0,1] -> dict(euler, axis_angle, joint_cam, rotmat, betas, cam, reba, rula).
    per_frame_scorers: score frame by frame, as the reference's classes do (`for pose in poses`, reba.py:50-81,
    rula.py:66-98), instead of one vectorised call over all frames -- the same results, the reference's arrangement
    (bench.py's cpu_baseline times it that way)."""
    t = dict(encoder=0.0, rot=0.0, smpl=0.0, score=0.0)
    eul, aa, rot, betas, cams = [], [], [], [], []
    crops = torch.as_tensor(crops)
    with torch.no_grad():
        for i in range(0, crops.shape[0], batch_size):
            t0 = time.perf_counter()
            r, b, c = hmr_model(crops[i:i + batch_size])
            r = r.numpy()
            t1 = time.perf_counter()
            for fr in r:                                   # base.py:225-229
                p = coord_ref.rot_to_angle(fr)
                aa.append(p)
                eul.append(coord_ref.axis_angle_to_euler_angle(p))
            t2 = time.perf_counter()
            rot.append(r); betas.append(b.numpy()); cams.append(c.numpy())
            t['encoder'] += t1 - t0
            t['rot'] += t2 - t1
    euler = np.stack(eul)
    axis_angle = np.stack(aa)
    t0 = time.perf_counter()
    joint_cam = coord_ref.get_joint_cam(axis_angle, lambda p, b: smpl_ref.smpl_forward(smpl_model, p, b))
    t1 = time.perf_counter()
    if per_frame_scorers:
        reba = np.concatenate([reba_ref.reba_packed(euler[f], add_info["REBA"]) for f in range(euler.shape[0])])
        rula = np.concatenate([rula_ref.rula_packed(euler[f], add_info["RULA"]) for f in range(euler.shape[0])])
    else:
        reba = reba_ref.reba_packed(euler, add_info["REBA"])
        rula = rula_ref.rula_packed(euler, add_info["RULA"])
    t2 = time.perf_counter()
    t['smpl'] += t1 - t0
    t['score'] += t2 - t1
    if timings is not None:
        timings.update(t)
    return dict(euler=euler, axis_angle=axis_angle, joint_cam=joint_cam, rotmat=np.concatenate(rot),
                betas=np.concatenate(betas), cam=np.concatenate(cams), reba=reba, rula=rula)
