"""SPIN ``hmr`` encoder + iterative regressor restated with torch-CPU fp32 primitives.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED: the reference
imports this model from nkolot/SPIN (``lib/core/base.py:23``; cloned without a
commit pin by ``script/install_conda.sh:24`` and git-ignored), holds no test or
golden vector at that boundary, and the source is not in this image.  This file
restates SPIN's published architecture (``models/hmr.py``: ResNet-50 Bottleneck
[3,4,6,3] with the stride on the 3x3 conv, ``AvgPool2d(7, stride=1)``, regressor
``fc1 (2048+144+13 -> 1024) -> drop -> fc2 (1024 -> 1024) -> drop ->
decpose/decshape/deccam`` iterated 3 times with residual updates, no activation
between the FCs; ``utils/geometry.py::rot6d_to_rotmat``) and is anchored on the
reference's call sites:

  * ``lib/core/base.py:81``     hmr(cfg.SPIN.SMPL_MEAN_PARAMS).to(device)
  * ``lib/core/base.py:83-84``  load_state_dict(checkpoint['model'], strict=False)
  * ``lib/core/base.py:212,220`` eval(); rotmat, betas, cam = model(batch)

State-dict key names are SPIN's so one seeded state dict drives
both this oracle and the HIP encoder.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

NPOSE = 24 * 6
LAYERS = (3, 4, 6, 3)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = F.relu(self.bn1(self.conv1(x)))
        y = F.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return F.relu(y + idt)


def rot6d_to_rotmat(x):
    """SPIN utils/geometry.py: 6-D (view(-1,3,2)) -> rotation matrix, columns b1,b2,b3."""
    x = x.reshape(-1, 3, 2)
    a1, a2 = x[:, :, 0], x[:, :, 1]
    b1 = F.normalize(a1)
    b2 = F.normalize(a2 - (b1 * a2).sum(dim=1, keepdim=True) * b1)
    b3 = torch.cross(b1, b2, dim=1)
    return torch.stack((b1, b2, b3), dim=-1)


class HMRRef(nn.Module):
    def __init__(self, init_pose=None, init_shape=None, init_cam=None):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.layer1 = self._stage(64, LAYERS[0], 1)
        self.layer2 = self._stage(128, LAYERS[1], 2)
        self.layer3 = self._stage(256, LAYERS[2], 2)
        self.layer4 = self._stage(512, LAYERS[3], 2)
        self.fc1 = nn.Linear(2048 + NPOSE + 13, 1024)
        self.fc2 = nn.Linear(1024, 1024)
        self.decpose = nn.Linear(1024, NPOSE)
        self.decshape = nn.Linear(1024, 10)
        self.deccam = nn.Linear(1024, 3)
        self.register_buffer('init_pose', torch.zeros(1, NPOSE) if init_pose is None else init_pose)
        self.register_buffer('init_shape', torch.zeros(1, 10) if init_shape is None else init_shape)
        self.register_buffer('init_cam', torch.zeros(1, 3) if init_cam is None else init_cam)

    def _stage(self, planes, blocks, stride):
        down = None
        if stride != 1 or self.inplanes != planes * 4:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                                 nn.BatchNorm2d(planes * 4))
        mods = [Bottleneck(self.inplanes, planes, stride, down)]
        self.inplanes = planes * 4
        mods += [Bottleneck(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*mods)

    def features(self, x):
        x = F.relu(self.bn1(self.conv1(x)))
        x = F.max_pool2d(x, 3, stride=2, padding=1)
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return F.avg_pool2d(x, 7, stride=1).flatten(1)

    def regress(self, xf, n_iter=3):
        B = xf.shape[0]
        pose, shape, cam = (self.init_pose.expand(B, -1), self.init_shape.expand(B, -1),
                            self.init_cam.expand(B, -1))
        for _ in range(n_iter):
            h = self.fc2(self.fc1(torch.cat([xf, pose, shape, cam], 1)))   # Dropout = identity in eval
            pose = self.decpose(h) + pose
            shape = self.decshape(h) + shape
            cam = self.deccam(h) + cam
        return pose, shape, cam

    def forward(self, x, n_iter=3):
        pose, shape, cam = self.regress(self.features(x), n_iter)
        return rot6d_to_rotmat(pose).view(x.shape[0], 24, 3, 3), shape, cam


def _bf16(t):
    """Round an fp32 tensor to bfloat16 values (round-to-nearest-even), kept in fp32 storage."""
    return t.to(torch.bfloat16).to(torch.float32)


def _folded(conv, bn):
    """Conv weight with eval-mode BN folded in (double precision), and the matching bias."""
    s = bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)
    w = (conv.weight.double() * s.view(-1, 1, 1, 1)).float()
    b = (bn.bias.double() - bn.running_mean.double() * s).float()
    return w, b


def features_bf16(model, x):
    """Emulation of the bf16 encoder (BASELINE config 3): BN folded in double, weights and every
    stored activation rounded to bfloat16, fp32 accumulation and bias, fp32 pooled features.
    Separates kernel errors from precision loss when testing the bf16 HIP path."""
    def cba(t, conv, bn, relu=True, res=None):
        w, b = _folded(conv, bn)
        y = F.conv2d(t, _bf16(w), b, stride=conv.stride, padding=conv.padding)
        if res is not None:
            y = y + res
        return _bf16(F.relu(y) if relu else y)

    t = cba(_bf16(x), model.conv1, model.bn1)
    t = F.max_pool2d(t, 3, stride=2, padding=1)
    for stage in (model.layer1, model.layer2, model.layer3, model.layer4):
        for blk in stage:
            idt = t if blk.downsample is None else cba(t, blk.downsample[0], blk.downsample[1], relu=False)
            y = cba(t, blk.conv1, blk.bn1)
            y = cba(y, blk.conv2, blk.bn2)
            t = cba(y, blk.conv3, blk.bn3, relu=True, res=idt)
    return F.avg_pool2d(t, 7, stride=1).flatten(1)


def build(state_dict):
    """HMRRef in eval mode with a SPIN-keyed state dict loaded (strict=False like base.py:84)."""
    m = HMRRef()
    m.load_state_dict({k: torch.as_tensor(v) for k, v in state_dict.items()}, strict=False)
    return m.eval()
