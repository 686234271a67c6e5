"""Host-side mirror of SPIN's `models.hmr` as PoseRisk uses it (lib/core/base.py:23,81-84,212,220).

    model = hmr(cfg.SPIN.SMPL_MEAN_PARAMS).to(device)
    model.load_state_dict(checkpoint['model'], strict=False)
    model.eval()
    pred_rotmat, pred_betas, pred_camera = model(batch)      # under torch.no_grad()

Underneath, the forward runs the hand-written gfx950 kernels of libposerisk_hip.so through the
C ABI; torch supplies device memory and the current stream only.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib, weights


CONV_FORMS = {"default": -1, "direct": 0, "winograd2": 2, "winograd4": 4, "winograd5": 5, "winograd244": 244,
              "winograd255": 255, "winograd455": 455}


class HMR:
    def __init__(self, smpl_mean_params=None, pretrained=True, max_batch=64, precision="fp32", conv_form="default"):
        """conv_form (fp32 encoder): "direct" | "winograd2" | "winograd4" (F(4x4,3x3), Lavin & Gray's points) | "winograd5"
        (F(4x4,3x3) on the points 0, +-11/16, +-3/2: same cost, half the rounding error) | "winograd244" (F(2x2) in layer2,
        F(4x4) in layer3 and layer4) | "default" (= "winograd5") -- the form of the ten 3x3 / stride-1 layers with >= 128
        channels (pr_hmr_create, include/poserisk_hip.h); an int of three digits (e.g. 244) gives the form of layer2 /
        layer3 / layer4 separately."""
        # `pretrained` is accepted for signature compatibility; SPIN uses it to fetch torchvision's
        # ImageNet weights, which load_state_dict overwrites anyway (base.py:83-84).
        self._sd = {}
        if smpl_mean_params is not None:
            mp = np.load(smpl_mean_params) if isinstance(smpl_mean_params, str) else smpl_mean_params
            self._sd["init_pose"] = np.asarray(mp["pose"], np.float32).reshape(1, -1)
            self._sd["init_shape"] = np.asarray(mp["shape"], np.float32).reshape(1, -1)
            self._sd["init_cam"] = np.asarray(mp["cam"], np.float32).reshape(1, -1)
        self._device = torch.device("cpu")
        self._handle = None
        self._capacity = 0
        self._min_capacity = int(max_batch)
        self._precision = {"fp32": 0, "bf16": 1}[precision]
        self._conv_form = CONV_FORMS[conv_form] if isinstance(conv_form, str) else int(conv_form)
        self.training = False

    # ---- nn.Module-like surface used by base.py -------------------------------------------
    def to(self, device):
        device = torch.device(device)
        if device != self._device:
            self._release()
        self._device = device
        return self

    def cuda(self, device=None):
        return self.to(torch.device("cuda", torch.cuda.current_device() if device is None else device))

    def eval(self):
        self.training = False
        return self

    def load_state_dict(self, state_dict, strict=False):
        for k, v in state_dict.items():
            k = k[7:] if k.startswith("module.") else k
            self._sd[k] = weights._to_numpy(v).copy()
        self._release()
        missing = weights.missing_keys(self._sd)
        if strict and missing:
            raise RuntimeError(f"missing keys: {missing}")
        return missing, []

    def clone(self):
        """A second handle on the same weights (own workspaces), e.g. one per pipeline lane / stream."""
        m = HMR(max_batch=self._min_capacity, precision={0: "fp32", 1: "bf16"}[self._precision])
        m._conv_form = self._conv_form
        m._sd = self._sd          # host copies are read-only after load_state_dict
        m._device = self._device
        return m

    def state_dict(self):
        return {k: torch.from_numpy(v.copy()) for k, v in self._sd.items()}

    # ---- handle management --------------------------------------------------------------
    def _release(self):
        if self._handle is not None:
            dev = self._device if self._device.type == "cuda" else None
            # refused (handle kept) while the current stream is being captured: include/poserisk_hip.h, pr_declare_stream
            _lib.check(_lib.declare_stream(dev).pr_hmr_destroy(self._handle), "pr_hmr_destroy")
            self._handle = None
            self._generation = getattr(self, "_generation", 0) + 1
            self._capacity = 0

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _ensure(self, batch):
        if self._device.type != "cuda":
            raise _lib.PoseRiskHipError("HMR runs on an MI355X only: call .to('cuda') first "
                                        "(there is no CPU fallback)")
        if self._handle is not None and batch <= self._capacity:
            return
        self._release()
        lib = _lib.declare_stream(self._device)     # create / set_streams refuse under a hipGraph capture of this stream
        blob = weights.flatten_state_dict(self._sd)
        cap = max(batch, self._min_capacity)
        h = C.c_void_p()
        idx = self._device.index if self._device.index is not None else torch.cuda.current_device()
        _lib.check(lib.pr_hmr_create(idx, blob.ctypes.data, blob.size, cap, self._precision, self._conv_form,
                                     C.byref(h)),
                   "pr_hmr_create")
        self._handle, self._capacity = h, cap
        self._generation = getattr(self, "_generation", 0) + 1    # device pointers changed: captured hipGraphs are stale
        if getattr(self, "_streams", None):
            _lib.check(lib.pr_hmr_set_streams(h, self._streams), "pr_hmr_set_streams")
        if getattr(self, "_concurrency", 1) > 1:
            _lib.check(lib.pr_hmr_set_concurrency(h, self._concurrency), "pr_hmr_set_concurrency")

    @property
    def handle(self):
        return self._handle

    @property
    def generation(self):
        """Counts the (re)allocations of this model's device state (handle created, regrown, workspaces reallocated by
        set_streams): whatever baked its device pointers in -- a captured hipGraph -- is valid for one generation only."""
        return getattr(self, "_generation", 0)

    # ---- forward ------------------------------------------------------------------------
    def forward(self, x, return_features=False):
        if x.dim() != 4 or tuple(x.shape[1:]) != (3, 224, 224):
            raise ValueError(f"expected [B,3,224,224], got {tuple(x.shape)}")
        if x.device.type != "cuda":
            raise _lib.PoseRiskHipError("input batch must be on the GPU (base.py:219 does batch.to(device))")
        x = x.contiguous().float()
        B = x.shape[0]
        self._ensure(B)
        dev = x.device
        rotmat = torch.empty((B, 24, 3, 3), dtype=torch.float32, device=dev)
        betas = torch.empty((B, 10), dtype=torch.float32, device=dev)
        cam = torch.empty((B, 3), dtype=torch.float32, device=dev)
        xf = torch.empty((B, 2048), dtype=torch.float32, device=dev) if return_features else None
        p6 = torch.empty((B, 144), dtype=torch.float32, device=dev) if return_features else None
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(_lib.load().pr_hmr_forward(self._handle, x.data_ptr(), B, rotmat.data_ptr(), betas.data_ptr(),
                                              cam.data_ptr(), xf.data_ptr() if xf is not None else None,
                                              p6.data_ptr() if p6 is not None else None, stream),
                   "pr_hmr_forward")
        if return_features:
            return rotmat, betas, cam, xf, p6
        return rotmat, betas, cam

    __call__ = forward

    def set_streams(self, n):
        """Number of concurrent sub-batch streams inside the encoder (1..8); results do not depend on it."""
        self._streams = int(n)
        if self._handle is not None:
            _lib.check(_lib.declare_stream(self._device).pr_hmr_set_streams(self._handle, self._streams), "pr_hmr_set_streams")
            self._generation = getattr(self, "_generation", 0) + 1    # workspaces were reallocated

    def set_concurrency(self, n):
        """Hint: how many handles' forwards overlap on this device (pipeline lanes); the persistent kernels size their grids
        by it (pr_hmr_set_concurrency, include/poserisk_hip.h).  Results do not depend on it."""
        self._concurrency = max(1, int(n))
        if self._handle is not None:
            _lib.check(_lib.load().pr_hmr_set_concurrency(self._handle, self._concurrency), "pr_hmr_set_concurrency")

    # ---- per-layer conv timing for bench.py's roofline ---------------------------------------
    def profile_enable(self, on=True):
        _lib.check(_lib.load().pr_hmr_profile_enable(self._handle, int(on)), "pr_hmr_profile_enable")

    def conv_form_resolved(self):
        """The conv form the handle really runs (what "default" resolved to in this library / environment)."""
        self._ensure(1)
        return int(_lib.load().pr_hmr_conv_form(self._handle))

    def plan_counts(self, batch):
        """-> (conv launches, Winograd layers) of one forward of `batch` frames: kernels between the layout change and the
        average pool = launches + 2 * Winograd layers (pr_hmr_plan_counts)."""
        self._ensure(batch)
        n, w = C.c_int(0), C.c_int(0)
        _lib.check(_lib.load().pr_hmr_plan_counts(self._handle, int(batch), C.byref(n), C.byref(w)), "pr_hmr_plan_counts")
        return n.value, w.value

    def profile_read(self, with_mfma_flops=False):
        """-> (ms, launches, algorithmic FLOP per frame) per conv layer; with_mfma_flops adds the FLOP the matrix pipes
        execute (K padding, Winograd products)."""
        n = _lib.load().pr_hmr_num_conv_layers()
        ms = np.zeros(n, np.float32)
        cnt = np.zeros(n, np.int32)
        fl = np.zeros(n, np.float64)
        mf = np.zeros(n, np.float64)
        _lib.check(_lib.load().pr_hmr_profile_read(self._handle, ms.ctypes.data, cnt.ctypes.data, fl.ctypes.data,
                                                   mf.ctypes.data, n), "pr_hmr_profile_read")
        return (ms, cnt, fl, mf) if with_mfma_flops else (ms, cnt, fl)


def hmr(smpl_mean_params=None, pretrained=True, **kw):
    """Constructor with SPIN's name and positional argument (base.py:81)."""
    return HMR(smpl_mean_params, pretrained=pretrained, **kw)
