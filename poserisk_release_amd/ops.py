"""Thin torch-tensor wrappers over the stand-alone C-ABI kernels (rotation conversions, scorers,
the stand-alone conv used by parity tests and tile tuning)."""
import numpy as np
import torch

from . import _lib


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def _need_cuda(t, name):
    if t.device.type != "cuda":
        raise _lib.PoseRiskHipError(f"{name}: tensor must be on the GPU (no CPU fallback)")


def rot6d_to_rotmat(pose6d):
    """SPIN utils/geometry.py::rot6d_to_rotmat.  f32[N,144] -> f32[N,24,3,3]."""
    _need_cuda(pose6d, "rot6d_to_rotmat")
    p = pose6d.contiguous().float()
    N = p.shape[0]
    out = torch.empty((N, 24, 3, 3), dtype=torch.float32, device=p.device)
    _lib.check(_lib.load().pr_rot6d_to_rotmat(p.data_ptr(), N, out.data_ptr(), _stream(p.device)), "pr_rot6d_to_rotmat")
    return out


def pose_to_euler(rotmat):
    """rot_to_angle + axis_angle_to_euler_angle (lib/utils/coord_utils.py:24-30, 83-95) for a batch.
    f32[N,24,3,3] -> (axis_angle f32[N,24,3], euler_deg f64[N,24,3], status int32[N])."""
    _need_cuda(rotmat, "pose_to_euler")
    r = rotmat.contiguous().float()
    N = r.shape[0]
    aa = torch.empty((N, 24, 3), dtype=torch.float32, device=r.device)
    eul = torch.empty((N, 24, 3), dtype=torch.float64, device=r.device)
    st = torch.empty((N,), dtype=torch.int32, device=r.device)
    _lib.check(_lib.load().pr_pose_to_euler(r.data_ptr(), N, aa.data_ptr(), eul.data_ptr(), st.data_ptr(),
                                            _stream(r.device)), "pr_pose_to_euler")
    return aa, eul, st


def axis_angle_to_euler(axis_angle):
    """axis_angle_to_euler_angle (lib/utils/coord_utils.py:83-95) for a batch.
    f32[N,24,3] -> (euler_deg f64[N,24,3], status int32[N])."""
    _need_cuda(axis_angle, "axis_angle_to_euler")
    a = axis_angle.contiguous().float()
    N = a.shape[0]
    eul = torch.empty((N, 24, 3), dtype=torch.float64, device=a.device)
    st = torch.empty((N,), dtype=torch.int32, device=a.device)
    _lib.check(_lib.load().pr_axis_angle_to_euler(a.data_ptr(), N, eul.data_ptr(), st.data_ptr(), _stream(a.device)),
               "pr_axis_angle_to_euler")
    return eul, st


def reba(euler_deg, info):
    """REBA.__call__ arithmetic (lib/utils/reba.py:50-81).  f64[N,24,3] -> int32[N,10]."""
    _need_cuda(euler_deg, "reba")
    e = euler_deg.contiguous().double()
    N = e.shape[0]
    out = torch.empty((N, 10), dtype=torch.int32, device=e.device)
    s = _lib.reba_info_struct(info)
    _lib.check(_lib.load().pr_reba(e.data_ptr(), N, s, out.data_ptr(), _stream(e.device)), "pr_reba")
    return out


def rula(euler_deg, info):
    """RULA.__call__ arithmetic (lib/utils/rula.py:66-98).  f64[N,24,3] -> int32[N,12]."""
    _need_cuda(euler_deg, "rula")
    e = euler_deg.contiguous().double()
    N = e.shape[0]
    out = torch.empty((N, 12), dtype=torch.int32, device=e.device)
    s = _lib.rula_info_struct(info)
    _lib.check(_lib.load().pr_rula(e.data_ptr(), N, s, out.data_ptr(), _stream(e.device)), "pr_rula")
    return out


def _out(out, shape, dt, dev):
    """The output tensor: a fresh one, or the caller's (e.g. a view into a larger buffer with guard rows behind it)."""
    if out is None:
        return torch.empty(shape, dtype=dt, device=dev)
    if tuple(out.shape) != tuple(shape) or out.dtype != dt or not out.is_contiguous() or out.device != dev:
        raise ValueError(f"out must be a contiguous {dt} tensor of shape {tuple(shape)} on {dev}")
    return out


def conv2d_nhwc(x, w_oihw, bias=None, residual=None, stride=1, pad=0, relu=False, tile_cfg=-1, repeats=0,
                precision="fp32", out=None):
    """Stand-alone conv on NHWC (test / tuning entry).  x [B,H,W,Cin] CUDA (f32, or bf16 with
    precision="bf16"), w numpy OIHW.  Returns (y [B,Ho,Wo,Cout] in x's dtype, ms_per_launch or None)."""
    _need_cuda(x, "conv2d_nhwc")
    bf = precision == "bf16"
    dt = torch.bfloat16 if bf else torch.float32
    x = x.contiguous().to(dt)
    B, H, W, Cin = x.shape
    w = np.ascontiguousarray(w_oihw, dtype=np.float32)
    Cout, Cin_real, KH, KW = w.shape
    Ho = (H + 2 * pad - KH) // stride + 1
    Wo = (W + 2 * pad - KW) // stride + 1
    y = _out(out, (B, Ho, Wo, Cout), dt, x.device)
    b = np.ascontiguousarray(bias, dtype=np.float32) if bias is not None else None
    res = residual.contiguous().to(dt) if residual is not None else None
    ms = np.zeros(1, np.float32)
    idx = x.device.index if x.device.index is not None else torch.cuda.current_device()
    _lib.check(_lib.load().pr_conv2d_nhwc(
        idx, x.data_ptr(), w.ctypes.data, b.ctypes.data if b is not None else None,
        res.data_ptr() if res is not None else None, y.data_ptr(), B, H, W, Cin, Cin_real, Cout, KH, KW,
        stride, pad, int(relu), tile_cfg, 1 if bf else 0, repeats, ms.ctypes.data, _stream(x.device)), "pr_conv2d_nhwc")
    return y, (float(ms[0]) if repeats > 0 else None)


def conv1x1_dual_nhwc(x1, w1, x2, w2, bias=None, stride2=1, relu=False, tile_cfg=-1, precision="fp32", out=None):
    """relu(x1*W1 + x2[::stride2, ::stride2]*W2 + bias) as one dual-source GEMM (a first Bottleneck's conv3 with its
    downsample branch summed in).  x1 [B,Ho,Wo,C1], x2 [B,H2,W2,C2] CUDA, w1 [Cout,C1], w2 [Cout,C2] numpy."""
    _need_cuda(x1, "conv1x1_dual_nhwc")
    bf = precision == "bf16"
    dt = torch.bfloat16 if bf else torch.float32
    x1, x2 = x1.contiguous().to(dt), x2.contiguous().to(dt)
    B, Ho, Wo, C1 = x1.shape
    _, H2, W2, C2 = x2.shape
    w1 = np.ascontiguousarray(w1, dtype=np.float32).reshape(-1, C1)
    w2 = np.ascontiguousarray(w2, dtype=np.float32).reshape(-1, C2)
    Cout = w1.shape[0]
    y = _out(out, (B, Ho, Wo, Cout), dt, x1.device)
    b = np.ascontiguousarray(bias, dtype=np.float32) if bias is not None else None
    idx = x1.device.index if x1.device.index is not None else torch.cuda.current_device()
    _lib.check(_lib.load().pr_conv1x1_dual_nhwc(
        idx, x1.data_ptr(), w1.ctypes.data, x2.data_ptr(), w2.ctypes.data, b.ctypes.data if b is not None else None,
        y.data_ptr(), B, Ho, Wo, C1, H2, W2, C2, stride2, Cout, int(relu), tile_cfg, 1 if bf else 0,
        _stream(x1.device)), "pr_conv1x1_dual_nhwc")
    return y


def conv3x3_conv1x1_nhwc(x, w2, b2, w3, b3, residual=None, relu=True, precision="fp32", out=None):
    """relu?(relu(conv3x3(x, w2) + b2) * w3^T + b3 + residual) in one kernel (a layer1 Bottleneck's conv2 + conv3).
    x [B,H,W,Cin] CUDA (f32, or bf16 with precision="bf16"), w2 [64,Cin,3,3], w3 [N3,64] numpy -> [B,H,W,N3]."""
    _need_cuda(x, "conv3x3_conv1x1_nhwc")
    bf = precision == "bf16"
    dt = torch.bfloat16 if bf else torch.float32
    x = x.contiguous().to(dt)
    B, H, W, Cin = x.shape
    w2 = np.ascontiguousarray(w2, dtype=np.float32)
    w3 = np.ascontiguousarray(w3, dtype=np.float32).reshape(-1, 64)
    b2 = np.ascontiguousarray(b2, dtype=np.float32)
    b3 = np.ascontiguousarray(b3, dtype=np.float32)
    N3 = w3.shape[0]
    res = residual.contiguous().to(dt) if residual is not None else None
    y = _out(out, (B, H, W, N3), dt, x.device)
    idx = x.device.index if x.device.index is not None else torch.cuda.current_device()
    _lib.check(_lib.load().pr_conv3x3_conv1x1_nhwc(
        idx, x.data_ptr(), w2.ctypes.data, b2.ctypes.data, w3.ctypes.data, b3.ctypes.data,
        res.data_ptr() if res is not None else None, y.data_ptr(), B, H, W, Cin, N3, int(relu), 1 if bf else 0,
        _stream(x.device)), "pr_conv3x3_conv1x1_nhwc")
    return y


def bottleneck_nhwc(x, w1, b1, w2, b2, w3, b3, wd=None, bd=None, repeats=0):
    """A whole layer1 Bottleneck (conv1 1x1 -> conv2 3x3 -> conv3 1x1 + identity, ReLU after each; BatchNorm folded by
    the caller) in one persistent bf16 kernel.  Without `wd`: x bf16 [B,H,W,256] CUDA, w1 [64,256], identity = x.  With
    `wd` [256,64] / `bd` [256] (the stage's first block): x bf16 [B,H,W,64], w1 [64,64], identity = the downsample branch.
    w2 [64,64,3,3], w3 [256,64] numpy.  Returns (y bf16 [B,H,W,256], ms_per_launch or None)."""
    _need_cuda(x, "bottleneck_nhwc")
    x = x.contiguous().to(torch.bfloat16)
    B, H, W, C = x.shape
    first = wd is not None
    if C != (64 if first else 256):
        raise ValueError(f"bottleneck_nhwc: {64 if first else 256} input channels expected, got {C}")
    f = lambda a, shape: np.ascontiguousarray(a, dtype=np.float32).reshape(shape)
    w1, w2, w3 = f(w1, (64, C)), f(w2, (64, 64, 3, 3)), f(w3, (256, 64))
    b1, b2, b3 = f(b1, (64,)), f(b2, (64,)), f(b3, (256,))
    if first:
        wd, bd = f(wd, (256, 64)), f(bd, (256,))
    y = torch.empty((B, H, W, 256), dtype=torch.bfloat16, device=x.device)
    ms = np.zeros(1, np.float32)
    idx = x.device.index if x.device.index is not None else torch.cuda.current_device()
    _lib.check(_lib.load().pr_bottleneck_nhwc(idx, x.data_ptr(), w1.ctypes.data, b1.ctypes.data, w2.ctypes.data,
                                              b2.ctypes.data, w3.ctypes.data, b3.ctypes.data,
                                              wd.ctypes.data if first else None, bd.ctypes.data if first else None,
                                              y.data_ptr(), B, H, W, repeats, ms.ctypes.data, _stream(x.device)),
               "pr_bottleneck_nhwc")
    return y, (float(ms[0]) if repeats > 0 else None)


def bottleneck128_nhwc(x, w1, b1, w2, b2, w3, b3, repeats=0):
    """A whole layer2 Bottleneck (plain block: conv1 1x1 512 -> 128, conv2 3x3, conv3 1x1 128 -> 512 + identity, ReLU after
    each; BatchNorm folded by the caller) in one persistent bf16 kernel.  x bf16 [B,H,W,512] CUDA (W <= 31), w1 [128,512],
    w2 [128,128,3,3], w3 [512,128] numpy.  Returns (y bf16 [B,H,W,512], ms_per_launch or None)."""
    _need_cuda(x, "bottleneck128_nhwc")
    x = x.contiguous().to(torch.bfloat16)
    B, H, W, C = x.shape
    if C != 512:
        raise ValueError(f"bottleneck128_nhwc: 512 input channels expected, got {C}")
    f = lambda a, shape: np.ascontiguousarray(a, dtype=np.float32).reshape(shape)
    w1, w2, w3 = f(w1, (128, 512)), f(w2, (128, 128, 3, 3)), f(w3, (512, 128))
    b1, b2, b3 = f(b1, (128,)), f(b2, (128,)), f(b3, (512,))
    y = torch.empty((B, H, W, 512), dtype=torch.bfloat16, device=x.device)
    ms = np.zeros(1, np.float32)
    idx = x.device.index if x.device.index is not None else torch.cuda.current_device()
    _lib.check(_lib.load().pr_bottleneck128_nhwc(idx, x.data_ptr(), w1.ctypes.data, b1.ctypes.data, w2.ctypes.data,
                                                 b2.ctypes.data, w3.ctypes.data, b3.ctypes.data, y.data_ptr(), B, H, W,
                                                 repeats, ms.ctypes.data, _stream(x.device)), "pr_bottleneck128_nhwc")
    return y, (float(ms[0]) if repeats > 0 else None)


def bottleneck256_nhwc(x, w1, b1, w2, b2, w3, b3, repeats=0):
    """A whole layer3 Bottleneck (plain block: conv1 1x1 1024 -> 256, conv2 3x3, conv3 1x1 256 -> 1024 + identity, ReLU after
    each; BatchNorm folded by the caller) in one bf16 kernel, one frame per workgroup.  x bf16 [B,H,W,1024] CUDA (H W <= 224),
    w1 [256,1024], w2 [256,256,3,3], w3 [1024,256] numpy.  Returns (y bf16 [B,H,W,1024], ms_per_launch or None)."""
    _need_cuda(x, "bottleneck256_nhwc")
    x = x.contiguous().to(torch.bfloat16)
    B, H, W, C = x.shape
    if C != 1024:
        raise ValueError(f"bottleneck256_nhwc: 1024 input channels expected, got {C}")
    f = lambda a, shape: np.ascontiguousarray(a, dtype=np.float32).reshape(shape)
    w1, w2, w3 = f(w1, (256, 1024)), f(w2, (256, 256, 3, 3)), f(w3, (1024, 256))
    b1, b2, b3 = f(b1, (256,)), f(b2, (256,)), f(b3, (1024,))
    y = torch.empty((B, H, W, 1024), dtype=torch.bfloat16, device=x.device)
    ms = np.zeros(1, np.float32)
    idx = x.device.index if x.device.index is not None else torch.cuda.current_device()
    _lib.check(_lib.load().pr_bottleneck256_nhwc(idx, x.data_ptr(), w1.ctypes.data, b1.ctypes.data, w2.ctypes.data,
                                                 b2.ctypes.data, w3.ctypes.data, b3.ctypes.data, y.data_ptr(), B, H, W,
                                                 repeats, ms.ctypes.data, _stream(x.device)), "pr_bottleneck256_nhwc")
    return y, (float(ms[0]) if repeats > 0 else None)


def stem_pool_nhwc(x, w, bias, repeats=0):
    """The bf16 encoder's stem in one kernel: 4x4 / stride-1 convolution (window rows y-2 .. y+1, i.e. padding 2 with the
    last row and column of the padded result dropped) over x bf16 [B,H,H,16] CUDA + bias + ReLU + MaxPool2d(3, 2, 1).
    w [64,16,4,4], bias [64] numpy.  Returns (y bf16 [B,H/2,H/2,64], ms_per_launch or None)."""
    _need_cuda(x, "stem_pool_nhwc")
    x = x.contiguous().to(torch.bfloat16)
    B, H, W, C = x.shape
    if C != 16 or H != W or H % 2:
        raise ValueError("stem_pool_nhwc: x must be [B,H,H,16] with even H")
    w = np.ascontiguousarray(w, dtype=np.float32).reshape(64, 16, 4, 4)
    b = np.ascontiguousarray(bias, dtype=np.float32).reshape(64)
    y = torch.empty((B, H // 2, H // 2, 64), dtype=torch.bfloat16, device=x.device)
    ms = np.zeros(1, np.float32)
    idx = x.device.index if x.device.index is not None else torch.cuda.current_device()
    _lib.check(_lib.load().pr_stem_pool_nhwc(idx, x.data_ptr(), w.ctypes.data, b.ctypes.data, y.data_ptr(), B, H, repeats,
                                             ms.ctypes.data, _stream(x.device)), "pr_stem_pool_nhwc")
    return y, (float(ms[0]) if repeats > 0 else None)


def stem_pool_f32_nhwc(x, w, bias, repeats=0):
    """The fp32 encoder's stem in one kernel: 4x4 / stride-1 convolution (window rows y-2 .. y+1) over the space-to-depth
    image x f32 [B,112,112,12] CUDA + bias + ReLU + MaxPool2d(3, 2, 1).  w [64,12,4,4], bias [64] numpy.
    Returns (y f32 [B,56,56,64], ms_per_launch or None)."""
    _need_cuda(x, "stem_pool_f32_nhwc")
    x = x.contiguous().float()
    if tuple(x.shape[1:]) != (112, 112, 12):
        raise ValueError(f"stem_pool_f32_nhwc: x must be [B,112,112,12], got {tuple(x.shape)}")
    B = x.shape[0]
    w = np.ascontiguousarray(w, dtype=np.float32).reshape(64, 12, 4, 4)
    b = np.ascontiguousarray(bias, dtype=np.float32).reshape(64)
    y = torch.empty((B, 56, 56, 64), dtype=torch.float32, device=x.device)
    ms = np.zeros(1, np.float32)
    idx = x.device.index if x.device.index is not None else torch.cuda.current_device()
    _lib.check(_lib.load().pr_stem_pool_f32_nhwc(idx, x.data_ptr(), w.ctypes.data, b.ctypes.data, y.data_ptr(), B, repeats,
                                                 ms.ctypes.data, _stream(x.device)), "pr_stem_pool_f32_nhwc")
    return y, (float(ms[0]) if repeats > 0 else None)


def crop_frames(frames, bboxes, frame_idx=None, scale=1.2, bgr=False, return_status=False):
    """GPU form of CropDataset.__getitem__ (data/demo_dataset.py:58-74) for a whole batch.
    frames u8[F,H,W,3] CUDA, bboxes f32[N,4] (cx,cy,w,h), frame_idx int32[N] or None -> f32[N,3,224,224].
    A host-side `frame_idx` is range-checked here (ValueError); one that already lives on the GPU is checked by
    the kernel, which zero-fills such crops and flags them in the int32[N] status (`return_status=True`)."""
    _need_cuda(frames, "crop_frames")
    if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[3] != 3:
        raise ValueError("frames must be uint8 [F,H,W,3]")
    frames = frames.contiguous()
    bb = torch.as_tensor(bboxes, dtype=torch.float32).to(frames.device).contiguous()
    N = bb.shape[0]
    if frame_idx is None:
        if N > frames.shape[0]:
            raise ValueError(f"{N} boxes for {frames.shape[0]} frames without a frame index")
    else:
        if not (isinstance(frame_idx, torch.Tensor) and frame_idx.device.type == "cuda"):
            host = np.asarray(frame_idx).reshape(-1)
            if host.shape[0] != N:
                raise ValueError(f"{host.shape[0]} frame indices for {N} boxes")
            if N and (int(host.min()) < 0 or int(host.max()) >= frames.shape[0]):
                raise ValueError(f"frame index out of range: {int(host.min())}..{int(host.max())} with "
                                 f"{frames.shape[0]} frames")
    if bb.dim() != 2 or bb.shape[1] != 4:
        raise ValueError(f"bboxes must be [N,4] (cx,cy,w,h), got {tuple(bb.shape)}")
    idx = None
    if frame_idx is not None:
        # the kernel reads frame_idx[n] for every n < N before it range-checks the VALUE: the array itself must hold N ints
        idx = torch.as_tensor(frame_idx).to(frames.device)
        if idx.dim() != 1 or idx.numel() != N:
            raise ValueError(f"frame_idx must be a vector of {N} frame indices, got shape {tuple(idx.shape)}")
        if idx.dtype.is_floating_point or idx.dtype == torch.bool:
            raise ValueError(f"frame_idx must hold integers, got {idx.dtype}")
        idx = idx.to(torch.int32).contiguous()
    out = torch.empty((N, 3, 224, 224), dtype=torch.float32, device=frames.device)
    status = torch.empty((N,), dtype=torch.int32, device=frames.device) if return_status else None
    F, H, W, _ = frames.shape
    _lib.check(_lib.load().pr_crop_frames(frames.data_ptr(), F, H, W, int(bool(bgr)),
                                          idx.data_ptr() if idx is not None else None, bb.data_ptr(), N,
                                          float(scale), out.data_ptr(),
                                          status.data_ptr() if status is not None else None,
                                          _stream(frames.device)), "pr_crop_frames")
    return (out, status) if return_status else out
