"""Rotation conversions of the hot path, restated from ``lib/utils/coord_utils.py``.

TEST INFRASTRUCTURE (see oracle/__init__.py).  The Euler part is pinned by
tests/golden/euler.npz (reference run with oracle.rodrigues_cv as the ``cv2``
stub); the Rodrigues part inherits rodrigues_cv's "parity unpinned".

Dtype behaviour that parity depends on:
  * rotmat f32 -> rotvec f32 (OpenCV keeps depth)            coord_utils.py:24-30
  * rotvec f32 -> matrix f32; the Euler extraction squares / adds f32 scalars
    (numpy float32 arithmetic), takes sqrt/atan2 in double     coord_utils.py:69-81
  * the result is float64 degrees, order (x, y, z)             coord_utils.py:93
"""
import math

import numpy as np

from . import rodrigues_cv

ROOT_INIT = (3.14, 0.0, 0.0)  # coord_utils.py:10 -- 3.14, not pi (Q4)


class RotationAssertion(AssertionError):
    """Raised where the reference does ``assert`` / ``assert 0`` (coord_utils.py:70,91)."""


def rot_to_angle(rotmat):
    """coord_utils.py:24-30 -- per joint matrix -> axis-angle, f32[J,3,3] -> f32[J,3]."""
    return np.stack([rodrigues_cv.rotmat_to_rotvec(m) for m in rotmat])


def is_rotation_matrix(R):
    """coord_utils.py:62-67 -- Frobenius norm of I - R^T R below 1e-6, in R's dtype."""
    err = np.identity(3, dtype=R.dtype) - np.dot(R.T, R)
    return np.linalg.norm(err) < 1e-6


def rotation_matrix_to_euler(R):
    """coord_utils.py:69-81 -- ZYX Euler (x, y, z) in radians, float64."""
    if not is_rotation_matrix(R):
        raise RotationAssertion("not a rotation matrix")
    sy = math.sqrt(R[0, 0] * R[0, 0] + R[1, 0] * R[1, 0])  # products/sum in R.dtype
    if sy < 1e-6:
        ex = math.atan2(-R[1, 2], R[1, 1])
        ey = math.atan2(-R[2, 0], sy)
        ez = 0
    else:
        ex = math.atan2(R[2, 1], R[2, 2])
        ey = math.atan2(-R[2, 0], sy)
        ez = math.atan2(R[1, 0], R[0, 0])
    return np.array([ex, ey, ez])


def euler_to_rotmat(yaw, pitch, roll):
    """coord_utils.py:45-60 -- R = Rz(yaw) Ry(pitch) Rx(roll), float64."""
    cz, sz = np.cos(yaw), np.sin(yaw)
    cy, sy = np.cos(pitch), np.sin(pitch)
    cx, sx = np.cos(roll), np.sin(roll)
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    return np.dot(Rz, np.dot(Ry, Rx))


def axis_angle_to_euler_angle(pose):
    """coord_utils.py:83-95 -- f32[J,3] axis-angle -> f64[J,3] Euler degrees.

    Keeps the reference's round-trip check: the *signed* sum of (R - R') above
    0.1 aborts (coord_utils.py:90-91).
    """
    out = []
    for rv in pose:
        R = rodrigues_cv.rotvec_to_rotmat(rv)
        e = rotation_matrix_to_euler(R)
        R2 = euler_to_rotmat(e[2], e[1], e[0])
        if (R - R2).sum() > 0.1:
            raise RotationAssertion("euler round trip")
        out.append(e * 180 / math.pi)
    return np.stack(out)


def get_joint_cam(poses, smpl_forward):
    """coord_utils.py:7-21 -- per-frame SMPL joints, mm, root-relative.

    ``smpl_forward(pose f32[1,72], betas f32[1,10]) -> (verts, joints)`` is the
    neutral layer.  As in the reference the root row of every pose is
    overwritten IN PLACE with (3.14, 0, 0) (Q4/Q5), betas are zero, batch is 1.
    """
    zeros = np.zeros((1, 10), dtype=np.float32)
    out = []
    for pose in poses:
        pose[0] = np.asarray(ROOT_INIT, dtype=pose.dtype)
        _, joints = smpl_forward(np.asarray(pose, dtype=np.float32).reshape(1, 72), zeros)
        j = np.asarray(joints, dtype=np.float32).reshape(24, 3) * 1000
        out.append(j - j[0, None])
    return np.stack(out)
