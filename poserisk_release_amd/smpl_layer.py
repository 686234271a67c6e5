"""Host-side mirror of `SMPL_Layer` (lib/smplpytorch/smplpytorch/pytorch/smpl_layer.py:12-158).

    layer = SMPLLayer(model_dict, gender='neutral')
    verts, joints = layer(pose[B,72], betas[B,10])          # metres

The arithmetic runs in libposerisk_hip.so (pr_smpl_*).  Inputs may be CPU tensors (the
reference keeps its SMPL layers on the CPU, lib/utils/smpl.py:44-45): they are moved to the
layer's GPU, and the results come back on the input's device.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


class SMPLLayer:
    def __init__(self, model, gender="neutral", center_idx=None, device=None, max_batch=256):
        """model: dict with v_template[V,3], shapedirs[V,3,NB], posedirs[V,3,207], J_regressor[J,V],
        weights[V,J], parents[J] (kintree_table[0]); optional 'model_betas'[NB], 'f' (faces)."""
        self.gender = gender
        self.center_idx = center_idx
        self._m = dict(v_template=_f32(model["v_template"]).reshape(-1, 3),
                       shapedirs=_f32(model["shapedirs"]), posedirs=_f32(model["posedirs"]),
                       J_regressor=_f32(model["J_regressor"]), weights=_f32(model["weights"]))
        V = self._m["v_template"].shape[0]
        self._m["shapedirs"] = self._m["shapedirs"].reshape(V, 3, -1)
        self._m["posedirs"] = self._m["posedirs"].reshape(V, 3, -1)
        self.num_verts = V
        self.kintree_parents = [int(p) for p in np.asarray(model["parents"]).reshape(-1)]
        self.num_joints = len(self.kintree_parents)
        self.num_betas = self._m["shapedirs"].shape[2]
        mb = model.get("model_betas")
        self._model_betas = _f32(mb).reshape(-1) if mb is not None else np.zeros(self.num_betas, np.float32)
        # buffers the rest of the reference reads (lib/utils/smpl.py:12-13)
        self.th_betas = torch.from_numpy(self._model_betas.copy()).unsqueeze(0)
        self.th_v_template = torch.from_numpy(self._m["v_template"].copy()).unsqueeze(0)
        self.th_shapedirs = torch.from_numpy(self._m["shapedirs"].copy())
        self.th_posedirs = torch.from_numpy(self._m["posedirs"].copy())
        self.th_J_regressor = torch.from_numpy(self._m["J_regressor"].copy())
        self.th_weights = torch.from_numpy(self._m["weights"].copy())
        faces = model.get("f")
        self.th_faces = torch.from_numpy(np.asarray(faces).astype(np.int64)) if faces is not None \
            else torch.zeros((0, 3), dtype=torch.long)
        self._device = torch.device(device) if device is not None else None
        self._handle = None
        self._max_batch = int(max_batch)

    def clone(self):
        """A second handle on the same model constants (own workspaces), one per pipeline lane."""
        m = dict(self._m, parents=self.kintree_parents, model_betas=self._model_betas, f=self.th_faces.numpy())
        return SMPLLayer(m, gender=self.gender, center_idx=self.center_idx, device=self._device,
                         max_batch=self._max_batch)

    def to(self, device):
        device = torch.device(device)
        if self._device != device:
            self._release()
        self._device = device
        return self

    @property
    def generation(self):
        return getattr(self, "_generation", 0)

    def _release(self):
        if self._handle is not None:
            dev = self._device if self._device is not None and self._device.type == "cuda" else None
            _lib.check(_lib.declare_stream(dev).pr_smpl_destroy(self._handle), "pr_smpl_destroy")   # refused under a capture
            self._handle = None
            self._generation = getattr(self, "_generation", 0) + 1

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _ensure(self):
        if self._handle is not None:
            return
        if self._device is None:
            self._device = torch.device("cuda", torch.cuda.current_device())
        if self._device.type != "cuda":
            raise _lib.PoseRiskHipError("SMPLLayer computes on an MI355X only (no CPU fallback)")
        m = self._m
        parents = np.ascontiguousarray(np.asarray(self.kintree_parents, dtype=np.int32))
        parents[0] = -1
        h = C.c_void_p()
        idx = self._device.index if self._device.index is not None else torch.cuda.current_device()
        _lib.check(_lib.declare_stream(self._device).pr_smpl_create(
            idx, m["v_template"].ctypes.data, m["shapedirs"].ctypes.data, m["posedirs"].ctypes.data,
            m["J_regressor"].ctypes.data, m["weights"].ctypes.data, parents.ctypes.data,
            self._model_betas.ctypes.data, self.num_verts, self.num_joints, self.num_betas,
            self._max_batch, C.byref(h)), "pr_smpl_create")
        self._handle = h
        self._generation = getattr(self, "_generation", 0) + 1    # see HMR.generation

    @property
    def handle(self):
        self._ensure()
        return self._handle

    @property
    def device(self):
        self._ensure()
        return self._device

    def _stage(self, t, B, width):
        """None / the reference's `torch.zeros(1)` placeholder -> None; else f32[B,width] on the GPU."""
        if t is None:
            return None
        t = torch.as_tensor(t)
        if t.numel() != B * width:
            if bool((t == 0).all()):
                return None
            raise ValueError(f"expected {B}x{width} values, got shape {tuple(t.shape)}")
        return t.to(self._device, torch.float32).reshape(B, width).contiguous()

    def forward(self, th_pose_axisang, th_betas=None, th_trans=None, return_verts=True):
        self._ensure()
        pose_in = torch.as_tensor(th_pose_axisang)
        out_dev = pose_in.device
        B = pose_in.shape[0]
        pose = pose_in.to(self._device, torch.float32).reshape(B, 72).contiguous()
        betas = self._stage(th_betas, B, self.num_betas)
        trans = self._stage(th_trans, B, 3)
        verts = torch.empty((B, self.num_verts, 3), dtype=torch.float32, device=self._device) if return_verts else None
        joints = torch.empty((B, self.num_joints, 3), dtype=torch.float32, device=self._device)
        stream = torch.cuda.current_stream(self._device).cuda_stream
        _lib.check(_lib.load().pr_smpl_forward(
            self._handle, pose.data_ptr(), betas.data_ptr() if betas is not None else None,
            trans.data_ptr() if trans is not None else None, B,
            -1 if self.center_idx is None else int(self.center_idx),
            verts.data_ptr() if verts is not None else None, joints.data_ptr(), stream), "pr_smpl_forward")
        if out_dev != self._device:
            return (verts.to(out_dev) if verts is not None else None), joints.to(out_dev)
        return verts, joints

    __call__ = forward

    def joint_cam(self, axis_angle, return_verts=False):
        """get_joint_cam (lib/utils/coord_utils.py:7-21) on f32[N,24,3] CUDA tensor; mutates its root rows."""
        self._ensure()
        if axis_angle.device != self._device or axis_angle.dtype != torch.float32 or not axis_angle.is_contiguous():
            raise ValueError("joint_cam needs a contiguous float32 tensor on the layer's GPU (it is mutated in place)")
        N = axis_angle.shape[0]
        jc = torch.empty((N, 24, 3), dtype=torch.float32, device=self._device)
        verts = torch.empty((N, self.num_verts, 3), dtype=torch.float32, device=self._device) if return_verts else None
        stream = torch.cuda.current_stream(self._device).cuda_stream
        _lib.check(_lib.load().pr_smpl_joint_cam(self._handle, axis_angle.data_ptr(), N, jc.data_ptr(),
                                                 verts.data_ptr() if verts is not None else None, stream),
                   "pr_smpl_joint_cam")
        return (jc, verts) if return_verts else jc
