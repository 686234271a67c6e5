"""CPU emulation of the encoder with its eligible 3x3 layers in Winograd F(m x m,3x3) form, fp32 arithmetic, against an
fp64 run of the same network -- on the benign He-normal synthetic weights and on a "trained-like" synthetic state dict
(heavy-tailed weights, BatchNorm statistics calibrated on data, wide gamma / beta).  Development tool for the numerics
decision in DESIGN.md 3.1b; the GPU test of the same question is tests/test_hip_parity.py::test_hmr_winograd_under_wide_dynamic_range.

    python scripts/wino_stress_cpu.py [frames]
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import hmr_ref  # noqa: E402  (development script, not product code)
from poserisk_release_amd import synth  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from stress_weights import trained_like_state_dict  # noqa: E402

BT = {4: np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0],
                   [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], np.float64),
      2: np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)}
G = {4: np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                  [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], np.float64),
     2: np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)}
AT = {4: np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], np.float64),
      2: np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)}


def wino_conv(x, w, b, m):
    """x f32[B,C,H,W], w f64 folded [Co,C,3,3], b f32[Co]: F(m x m, 3x3), pad 1, all arithmetic fp32."""
    n = m + 2
    B_, C, H, W = x.shape
    th, tw = -(-H // m), -(-W // m)
    xp = F.pad(x, (1, tw * m + 1 - W, 1, th * m + 1 - H))
    d = xp.unfold(2, n, m).unfold(3, n, m)                                      # [B,C,th,tw,n,n]
    bt = torch.from_numpy(BT[m]).float()
    v = torch.einsum("ij,bcyxjk,lk->bcyxil", bt, d, bt)                         # fp32 transforms
    u = torch.einsum("ij,ocjk,lk->ocil", torch.from_numpy(G[m]), w.double(), torch.from_numpy(G[m])).float()
    mm = torch.einsum("bcyxil,ocil->boyxil", v, u)
    at = torch.from_numpy(AT[m]).float()
    y = torch.einsum("ij,boyxjk,lk->boyxil", at, mm, at)                        # [B,Co,th,tw,m,m]
    y = y.permute(0, 1, 2, 4, 3, 5).reshape(B_, -1, th * m, tw * m)[:, :, :H, :W]
    return y + b.view(1, -1, 1, 1)


def features(model, x, m):
    """fp32 forward of hmr_ref.HMRRef with BN folded (double) and the eligible layers in Winograd form (m=0: direct)."""
    def cba(t, conv, bn, relu=True, res=None):
        s = bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)
        w = conv.weight.double() * s.view(-1, 1, 1, 1)
        b = (bn.bias.double() - bn.running_mean.double() * s).float()
        if m and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.in_channels >= 128:
            y = wino_conv(t, w, b, m)
        else:
            y = F.conv2d(t, w.float(), b, stride=conv.stride, padding=conv.padding)
        if res is not None:
            y = y + res
        return F.relu(y) if relu else y

    t = cba(x, model.conv1, model.bn1)
    t = F.max_pool2d(t, 3, stride=2, padding=1)
    for stage in (model.layer1, model.layer2, model.layer3, model.layer4):
        for blk in stage:
            idt = t if blk.downsample is None else cba(t, blk.downsample[0], blk.downsample[1], relu=False)
            y = cba(t, blk.conv1, blk.bn1)
            y = cba(y, blk.conv2, blk.bn2)
            t = cba(y, blk.conv3, blk.bn3, relu=True, res=idt)
    return F.avg_pool2d(t, 7, stride=1).flatten(1)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    x = torch.from_numpy(synth.crops(n, seed=3))
    for name, sd in (("he-normal synthetic", synth.hmr_state_dict(seed=1)), ("trained-like synthetic", trained_like_state_dict())):
        model = hmr_ref.build(sd)
        m64 = hmr_ref.build(sd).double()
        with torch.no_grad():
            xf64 = m64.features(x.double())
            p6, b64, c64 = m64.regress(xf64)
            r64 = hmr_ref.rot6d_to_rotmat(p6).view(n, 24, 3, 3)
            print(f"{name}: |xf|max {float(xf64.abs().max()):.3g}, mean {float(xf64.mean()):.3g}")
            for m in (0, 2, 4):
                xf = features(model, x, m)
                p6, b, c = model.regress(xf)
                r = hmr_ref.rot6d_to_rotmat(p6).view(n, 24, 3, 3)
                print(f"  form {m}: xf rel {float((xf.double() - xf64).abs().max() / xf64.abs().max()):.2e}  rotmat "
                      f"{float((r.double() - r64).abs().max()):.2e}  betas {float((b.double() - b64).abs().max()):.2e}  "
                      f"cam {float((c.double() - c64).abs().max()):.2e}")


if __name__ == "__main__":
    main()
