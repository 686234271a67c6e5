"""OpenCV-semantics ``cv2.Rodrigues`` restated in numpy (float64 internally).

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED: opencv-python is
an unpinned dependency of the reference (``requirements.txt:10``) and is absent
from this build image, so this restates OpenCV's published algorithm
(calib3d ``cvRodrigues2``: SVD-orthonormalise, theta = acos((tr-1)/2), axis from
the skew part, special cases theta~0 / theta~pi; vector form
R = c*I + (1-c)*r*r^T + s*[r]_x) and anchors on the reference's call sites:

  * ``lib/utils/coord_utils.py:27``   cv2.Rodrigues(p)[0]      3x3 f32 -> 3 f32
  * ``lib/utils/coord_utils.py:86``   cv2.Rodrigues(angle)[0]  3 f32  -> 3x3 f32

The output depth equals the input depth (float32 in, float32 out) while the
arithmetic is double, exactly as the OpenCV routine does.
"""
import numpy as np

_DBL_EPS = np.finfo(np.float64).eps


def rotmat_to_rotvec(R):
    """3x3 rotation matrix -> rotation vector, OpenCV matrix->vector branch."""
    out_dtype = np.asarray(R).dtype if np.asarray(R).dtype in (np.float32, np.float64) else np.float64
    R = np.asarray(R, dtype=np.float64).reshape(3, 3)
    if not np.all(np.isfinite(R)) or np.any(np.abs(R) >= 100):
        return np.zeros(3, dtype=out_dtype)
    U, _, Vt = np.linalg.svd(R)
    R = U @ Vt
    r = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]], dtype=np.float64)
    s = np.sqrt((r[0] * r[0] + r[1] * r[1] + r[2] * r[2]) * 0.25)
    c = (R[0, 0] + R[1, 1] + R[2, 2] - 1.0) * 0.5
    c = 1.0 if c > 1.0 else (-1.0 if c < -1.0 else c)
    theta = np.arccos(c)
    if s < 1e-5:
        if c > 0:
            r = np.zeros(3)
        else:
            rx = np.sqrt(max((R[0, 0] + 1) * 0.5, 0.0))
            ry = np.sqrt(max((R[1, 1] + 1) * 0.5, 0.0)) * (-1.0 if R[0, 1] < 0 else 1.0)
            rz = np.sqrt(max((R[2, 2] + 1) * 0.5, 0.0)) * (-1.0 if R[0, 2] < 0 else 1.0)
            if abs(rx) < abs(ry) and abs(rx) < abs(rz) and ((R[1, 2] > 0) != (ry * rz > 0)):
                rz = -rz
            r = np.array([rx, ry, rz])
            r = r * (theta / np.sqrt(rx * rx + ry * ry + rz * rz))
    else:
        r = r * ((1.0 / (2.0 * s)) * theta)
    return r.astype(out_dtype)


def rotvec_to_rotmat(v):
    """Rotation vector -> 3x3 matrix, OpenCV vector->matrix branch."""
    out_dtype = np.asarray(v).dtype if np.asarray(v).dtype in (np.float32, np.float64) else np.float64
    r = np.asarray(v, dtype=np.float64).reshape(3)
    theta = np.sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2])
    if theta < _DBL_EPS:
        return np.eye(3, dtype=out_dtype)
    c, s = np.cos(theta), np.sin(theta)
    c1 = 1.0 - c
    r = r * (1.0 / theta)
    rrt = np.outer(r, r)
    r_x = np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]], dtype=np.float64)
    return (c * np.eye(3) + c1 * rrt + s * r_x).astype(out_dtype)


def Rodrigues(src):
    """Signature-compatible stand-in for ``cv2.Rodrigues``: returns (dst, jacobian).

    Matrix in -> (3,1) vector out; 3-vector in -> 3x3 matrix out (same depth).
    Used as the ``cv2`` stub when tests/golden/make_golden.py imports the
    reference's ``coord_utils``; the jacobian is never read by the
    reference so ``None`` is returned in its place.
    """
    a = np.asarray(src)
    if a.shape == (3, 3):
        return rotmat_to_rotvec(a).reshape(3, 1), None
    return rotvec_to_rotmat(a.reshape(3)), None
