"""SMPL linear-blend-skinning forward restated in numpy float32.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Pinned by tests/golden/smpl_*.npz,
produced by the reference's own ``SMPL_Layer.forward`` on seeded synthetic models.

Follows ``lib/smplpytorch/smplpytorch/pytorch/smpl_layer.py:65-158``,
``rodrigues_layer.py:13-52`` and ``tensutils.py:6-48``.
"""
import numpy as np

F32 = np.float32

# smpl_layer.py:60-62 -- kintree_table[0] of the SMPL model
SMPL_PARENTS = (-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21)


def batch_rodrigues(axisang):
    """rodrigues_layer.py:41-52 + quat2mat :13-38.  f32[N,3] -> f32[N,9] (row-major).

    Quirks kept (Q8): the norm is of (v + 1e-8) but the division uses v itself;
    the half-angle quaternion is re-normalised before expansion.
    """
    v = np.asarray(axisang, dtype=F32)
    norm = np.sqrt(((v + F32(1e-8)) ** 2).sum(axis=1, dtype=F32), dtype=F32)[:, None]
    axis = v / norm
    half = norm * F32(0.5)
    q = np.concatenate([np.cos(half, dtype=F32), np.sin(half, dtype=F32) * axis], axis=1).astype(F32)
    q = q / np.sqrt((q * q).sum(axis=1, dtype=F32), dtype=F32)[:, None]
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    w2, x2, y2, z2 = w * w, x * x, y * y, z * z
    wx, wy, wz = w * x, w * y, w * z
    xy, xz, yz = x * y, x * z, y * z
    two = F32(2)
    return np.stack([
        w2 + x2 - y2 - z2, two * xy - two * wz, two * wy + two * xz,
        two * wz + two * xy, w2 - x2 + y2 - z2, two * yz - two * wx,
        two * xz - two * wy, two * wx + two * yz, w2 - x2 - y2 + z2,
    ], axis=1).astype(F32)


class SMPLModel:
    """The five model constants + kinematic tree (what ``SMPL_Layer.__init__`` registers,
    smpl_layer.py:40-63), as plain float32 arrays."""

    def __init__(self, v_template, shapedirs, posedirs, J_regressor, weights,
                 parents=SMPL_PARENTS, model_betas=None):
        self.v_template = np.asarray(v_template, F32).reshape(-1, 3)          # [V,3]
        V = self.v_template.shape[0]
        self.shapedirs = np.asarray(shapedirs, F32).reshape(V, 3, -1)          # [V,3,10]
        self.posedirs = np.asarray(posedirs, F32).reshape(V, 3, -1)            # [V,3,207]
        self.J_regressor = np.asarray(J_regressor, F32)                        # [J,V]
        self.weights = np.asarray(weights, F32)                                # [V,J]
        self.parents = tuple(int(p) for p in parents)
        nb = self.shapedirs.shape[2]
        self.model_betas = np.zeros((1, nb), F32) if model_betas is None else np.asarray(model_betas, F32).reshape(1, nb)


def smpl_forward(model, pose, betas=None, trans=None, center_idx=None):
    """smpl_layer.py:65-158.  pose f32[B,72], betas f32[B,10] -> verts f32[B,V,3], joints f32[B,J,3]."""
    pose = np.asarray(pose, F32)
    B = pose.shape[0]
    J = len(model.parents)
    rot = np.concatenate([batch_rodrigues(pose[:, 3 * j:3 * j + 3]) for j in range(J)], axis=1)  # tensutils.py:6-19
    root_rot = rot[:, :9].reshape(B, 3, 3)
    rest = rot[:, 9:]
    pose_map = rest - np.tile(np.eye(3, dtype=F32).reshape(1, 9), (1, J - 1))              # tensutils.py:41-48

    # shape blend + joint regression, smpl_layer.py:85-95 (Q9: all-zero betas -> model betas, repeat)
    use_model_betas = betas is None or float(np.sqrt((np.asarray(betas, F32) ** 2).sum())) == 0.0
    b = model.model_betas if use_model_betas else np.asarray(betas, F32)
    v_shaped = model.v_template[None] + np.transpose(model.shapedirs @ b.T, (2, 0, 1))     # [Bb,V,3]
    joints0 = np.matmul(model.J_regressor, v_shaped)                                       # [Bb,J,3]
    if use_model_betas:
        joints0 = np.tile(joints0, (B, 1, 1))

    # pose blend, smpl_layer.py:97-99
    v_posed = v_shaped + np.transpose(model.posedirs @ pose_map.T, (2, 0, 1))              # [B,V,3]

    # kinematic chain, smpl_layer.py:102-119
    G = np.zeros((B, J, 4, 4), F32)
    G[:, :, 3, 3] = 1
    G[:, 0, :3, :3] = root_rot
    G[:, 0, :3, 3] = joints0[:, 0]
    for i in range(1, J):
        p = model.parents[i]
        local = np.zeros((B, 4, 4), F32)
        local[:, 3, 3] = 1
        local[:, :3, :3] = rest[:, (i - 1) * 9:i * 9].reshape(B, 3, 3)
        local[:, :3, 3] = joints0[:, i] - joints0[:, p]
        G[:, i] = np.matmul(G[:, p], local)

    # remove rest pose, smpl_layer.py:122-132:  A_i = G_i - pack(G_i @ [j_i; 0])
    jh = np.concatenate([joints0, np.zeros((B, J, 1), F32)], axis=2)[..., None]            # [B,J,4,1]
    A = G.copy()
    A[..., 3:4] = G[..., 3:4] - np.matmul(G, jh)

    # skinning, smpl_layer.py:134-144:  T = A . W^T ; verts = sum_k T[:, :, k] * [v_posed; 1]_k
    T = np.einsum('bjrc,vj->bvrc', A, model.weights).astype(F32)                           # [B,V,4,4]
    vh = np.concatenate([v_posed, np.ones((B, v_posed.shape[1], 1), F32)], axis=2)         # [B,V,4]
    verts = np.einsum('bvrc,bvc->bvr', T, vh).astype(F32)[:, :, :3]
    jtr = G[:, :, :3, 3].copy()                                                           # smpl_layer.py:145

    # smpl_layer.py:147-155
    if trans is None or float(np.sqrt((np.asarray(trans, F32) ** 2).sum())) == 0.0:
        if center_idx is not None:
            c = jtr[:, center_idx][:, None]
            jtr = jtr - c
            verts = verts - c
    else:
        t = np.asarray(trans, F32)[:, None]
        jtr = jtr + t
        verts = verts + t
    return verts.astype(F32), jtr.astype(F32)
