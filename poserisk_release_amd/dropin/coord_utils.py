"""Mirror of lib/utils/coord_utils.py on the MI355X kernels (numpy in, numpy out, like the reference).

The per-joint Python loops of the reference (24 cv2.Rodrigues calls per frame, twice) become one
kernel launch per call; the reference's `assert`s (coord_utils.py:70,91) are raised from the
kernel's status bits.  For whole batches use poserisk_release_amd.ops.pose_to_euler directly.
"""
import numpy as np
import torch

from poserisk_release_amd import ops


def _dev():
    if not torch.cuda.is_available():
        from poserisk_release_amd._lib import PoseRiskHipError
        raise PoseRiskHipError("coord_utils needs an MI355X (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def _pad24(a, tail):
    """[J, *tail] -> one 24-joint frame (the kernels work on 24-joint frames)."""
    a = np.asarray(a, dtype=np.float32)
    J = a.shape[0]
    if J > 24:
        raise ValueError("at most 24 joints per call")
    out = np.zeros((1, 24) + tail, np.float32)
    if tail == (3, 3):
        out[0] = np.eye(3, dtype=np.float32)
    out[0, :J] = a
    return out, J


def rot_to_angle(rotmat):
    """coord_utils.py:24-30: f32[J,3,3] -> f32[J,3] axis-angle (OpenCV Rodrigues semantics)."""
    x, J = _pad24(rotmat, (3, 3))
    aa, _, _ = ops.pose_to_euler(torch.from_numpy(x).to(_dev()))
    return aa.cpu().numpy()[0, :J]


def batch_rot_to_euler(rotmat):
    """f32[N,24,3,3] -> (axis_angle f32[N,24,3], euler_deg f64[N,24,3]); raises like the reference."""
    aa, eul, st = ops.pose_to_euler(torch.as_tensor(rotmat, dtype=torch.float32).to(_dev()))
    if int(st.abs().sum()) != 0:
        raise AssertionError("invalid rotation (isRotationMatrix / Euler round trip), coord_utils.py:70,91")
    return aa.cpu().numpy(), eul.cpu().numpy()


def axis_angle_to_euler_angle(pose):
    """coord_utils.py:83-95: f32[J,3] axis-angle -> f64[J,3] Euler degrees (x, y, z)."""
    pose = np.asarray(pose, dtype=np.float32)
    J = pose.shape[0]
    if J > 24:
        raise ValueError("at most 24 joints per call")
    x = np.zeros((1, 24, 3), np.float32)
    x[0, :J] = pose
    eul, st = ops.axis_angle_to_euler(torch.from_numpy(x).to(_dev()))
    if int(st[0]) != 0:
        raise AssertionError("invalid rotation (isRotationMatrix / Euler round trip), coord_utils.py:70,91")
    return eul.cpu().numpy()[0, :J]


def get_joint_cam(poses, smpl_model):
    """coord_utils.py:7-21: f32[N,24,3] axis-angle -> f32[N,24,3] joints (mm, root-relative).
    Like the reference, the root row of every pose is overwritten IN PLACE with (3.14, 0, 0)."""
    layer = smpl_model.layer['neutral']
    p = np.asarray(poses)
    t = torch.from_numpy(np.ascontiguousarray(p, dtype=np.float32)).to(layer.device)
    jc = layer.joint_cam(t)
    if isinstance(poses, np.ndarray):
        poses[:, 0] = np.array([3.14, 0, 0], dtype=poses.dtype)   # Q5: visible to the caller
    return jc.cpu().numpy()
