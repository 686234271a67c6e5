"""GPU parity tests: the HIP path, called through the C ABI (ctypes), against the CPU oracle and the
committed golden vectors.  Bit-exact for the integer scores; stated tolerances for floating point."""
import json

import numpy as np
import pytest
import torch

from conftest import golden, measured
from oracle import coord_ref, hmr_ref, pipeline_ref, reba_ref, rula_ref, smpl_ref
from poserisk_release_amd import _lib, ops, synth
from poserisk_release_amd.hmr import HMR
from poserisk_release_amd.pipeline import FramePipeline
from poserisk_release_amd.smpl_layer import SMPLLayer

pytestmark = pytest.mark.gpu

TOL_F32 = 1e-4  # the north star's fp32 tolerance on SMPL pose/shape and 3-D joints
TOL_MM = 0.10   # the same tolerance on joint_cam, which the reference reports in millimetres (coord_utils.py:16)


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


# ------------------------------------------------------------------------------------------------
# conv building block vs torch fp32 (CPU)
# ------------------------------------------------------------------------------------------------
CONV_CASES = [
    # B, H, Cin_real, Cin, Cout, k, stride, pad
    (2, 56, 64, 64, 64, 1, 1, 0),
    (2, 56, 64, 64, 256, 1, 1, 0),
    (3, 28, 128, 128, 128, 3, 1, 1),
    (2, 56, 128, 128, 128, 3, 2, 1),
    (2, 56, 256, 256, 512, 1, 2, 0),
    (2, 224, 3, 4, 64, 7, 2, 3),
    (5, 7, 512, 512, 512, 3, 1, 1),
    (3, 14, 1024, 1024, 256, 1, 1, 0),
    (1, 9, 64, 64, 64, 3, 1, 1),      # ragged M (81 rows)
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_matches_torch(gpu_device, case):
    B, H, Cr, Cin, Cout, k, s, p = case
    rng = np.random.default_rng(hash(case) % (2 ** 32))
    x = rng.standard_normal((B, H, H, Cin)).astype(np.float32)
    x[..., Cr:] = 0
    w = (rng.standard_normal((Cout, Cr, k, k)) / np.sqrt(Cr * k * k)).astype(np.float32)
    bias = rng.standard_normal(Cout).astype(np.float32)
    ref = torch.nn.functional.conv2d(torch.from_numpy(x[..., :Cr]).permute(0, 3, 1, 2), torch.from_numpy(w),
                                     torch.from_numpy(bias), stride=s, padding=p)
    Ho = ref.shape[2]
    res = rng.standard_normal((B, Ho, Ho, Cout)).astype(np.float32)
    ref = torch.relu(ref.permute(0, 2, 3, 1) + torch.from_numpy(res)).numpy()
    n_cfg = _lib.load().pr_conv_num_tile_cfgs()
    for cfg in [-1] + list(range(n_cfg)):
        try:
            y, _ = ops.conv2d_nhwc(_t(x, gpu_device), w, bias, _t(res, gpu_device), stride=s, pad=p, relu=True,
                                   tile_cfg=cfg)
        except _lib.PoseRiskHipError as e:
            # this tile does not fit Cout, or it is one of the reserved indices of the retired first-generation kernel
            assert cfg >= 0 and ("not a multiple of tile N" in str(e) or (cfg < 6 and "retired" in str(e)) or "bf16-only" in str(e)), str(e)
            continue
        err = np.abs(y.cpu().numpy() - ref).max()
        assert err < 2e-5 * max(1.0, np.abs(ref).max()), f"cfg {cfg}: max err {err}"


def test_conv_no_bias_no_residual_no_relu(gpu_device):
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 14, 14, 256)).astype(np.float32)
    w = (rng.standard_normal((64, 256, 1, 1)) / 16).astype(np.float32)
    ref = torch.nn.functional.conv2d(torch.from_numpy(x).permute(0, 3, 1, 2), torch.from_numpy(w)).permute(0, 2, 3, 1)
    y, _ = ops.conv2d_nhwc(_t(x, gpu_device), w)
    np.testing.assert_allclose(y.cpu().numpy(), ref.numpy(), atol=2e-5)


@pytest.mark.parametrize("case", [
    # (B, H, Cin_real, Cin, Cout, k, stride, pad): tile counts 40 (all quarters, ragged last tile), 392 (136 tail
    # tiles, per-lane tap decode), 784 (16 tail tiles, scalar tap decode), 3136 (64 tail tiles, 1x1)
    (3, 14, 1024, 1024, 256, 1, 1, 0), (2, 224, 3, 4, 64, 7, 2, 3), (64, 14, 256, 256, 256, 3, 1, 1),
    (64, 14, 256, 256, 1024, 1, 1, 0)], ids=lambda c: "x".join(map(str, c)))
def test_conv_quarter_tiles_have_the_same_bits(gpu_device, case):
    """The 64x64 kernel computes the tiles beyond the last whole round of 256 CUs as 16x16-MFMA quarter tiles;
    they must equal, bit for bit, what a whole-tile kernel (128x64: no quarter path) produces."""
    B, H, Cr, Cin, Cout, k, s, p = case
    g = torch.Generator(device=gpu_device).manual_seed(11)
    x = torch.randn((B, H, H, Cin), generator=g, device=gpu_device)
    x[..., Cr:] = 0
    rng = np.random.default_rng(5)
    w = (rng.standard_normal((Cout, Cr, k, k)) / np.sqrt(Cr * k * k)).astype(np.float32)
    bias = rng.standard_normal(Cout).astype(np.float32)
    Ho = (H + 2 * p - k) // s + 1
    res = torch.randn((B, Ho, Ho, Cout), generator=g, device=gpu_device)
    y64, _ = ops.conv2d_nhwc(x, w, bias, res, stride=s, pad=p, relu=True, tile_cfg=8)
    y128, _ = ops.conv2d_nhwc(x, w, bias, res, stride=s, pad=p, relu=True, tile_cfg=7)
    assert torch.equal(y64, y128)
    y64, _ = ops.conv2d_nhwc(x, w, None, None, stride=s, pad=p, relu=False, tile_cfg=8)
    y128, _ = ops.conv2d_nhwc(x, w, None, None, stride=s, pad=p, relu=False, tile_cfg=7)
    assert torch.equal(y64, y128)


@pytest.mark.parametrize("case", [
    # (B, H, Cin, Cout, k, stride, pad): the encoder's three split-K layers at B=64 (392 tiles), small / ragged ones
    (64, 14, 512, 512, 3, 2, 1), (64, 7, 2048, 512, 1, 1, 0), (3, 7, 2048, 512, 1, 1, 0), (2, 9, 64, 128, 3, 1, 1),
    (1, 5, 96, 64, 1, 1, 0)], ids=lambda c: "x".join(map(str, c)))
def test_conv_split_k(gpu_device, case):
    """Split-K (tile_cfg 200 + S): every tile's K-steps dealt to S workgroups, partial tiles summed in part order by
    the workgroup that draws the last ticket.  Against torch; deterministic (twenty launches, the same bits: a stale
    read or an arrival-order dependence would show); S = 1-equivalent chains per part, so S = 2, 3, 4 agree to rounding;
    a frame's bits do not depend on its batch (the split is a property of the layer)."""
    B, H, Cin, Cout, k, st, pad = case
    rng = np.random.default_rng(H * 13 + Cin)
    g = torch.Generator(device=gpu_device).manual_seed(H + Cin)
    x = torch.randn((B, H, H, Cin), generator=g, device=gpu_device)
    w = (rng.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
    bias = rng.standard_normal(Cout).astype(np.float32)
    Ho = (H + 2 * pad - k) // st + 1
    res = torch.randn((B, Ho, Ho, Cout), generator=g, device=gpu_device)
    ref = torch.relu(torch.nn.functional.conv2d(x.cpu().double().permute(0, 3, 1, 2), torch.from_numpy(w).double(),
                                                torch.from_numpy(bias).double(), stride=st, padding=pad).permute(0, 2, 3, 1)
                     + res.cpu().double())
    outs = {}
    for S in (2, 3, 4):
        y, _ = ops.conv2d_nhwc(x, w, bias, res, stride=st, pad=pad, relu=True, tile_cfg=200 + S)
        assert float((y.cpu().double() - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max())), S
        outs[S] = y
    y1, _ = ops.conv2d_nhwc(x, w, bias, res, stride=st, pad=pad, relu=True, tile_cfg=8)
    assert float((outs[4] - y1).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
    for _ in range(20):
        y, _ = ops.conv2d_nhwc(x, w, bias, res, stride=st, pad=pad, relu=True, tile_cfg=204)
        assert torch.equal(y, outs[4])
    if B > 1:
        yb, _ = ops.conv2d_nhwc(x[1:2], w, bias, res[1:2], stride=st, pad=pad, relu=True, tile_cfg=204)
        assert torch.equal(yb[0], outs[4][1])


@pytest.mark.parametrize("m", [2, 4, 5])
@pytest.mark.parametrize("case", [(3, 28, 128, 128), (5, 14, 256, 256), (6, 7, 512, 512), (2, 9, 64, 192)],
                         ids=lambda c: "x".join(map(str, c)))
def test_conv_winograd_matches_torch_and_direct(gpu_device, case, m):
    """Winograd forms (tile_cfg = -form: 2 = F(2x2,3x3), 4 = F(4x4,3x3) on Lavin & Gray's points, 5 = F(4x4,3x3) on the
    points 0, +-11/16, +-3/2) against torch fp32 and against the direct implicit-GEMM kernel; sizes that are not multiples
    of the tile leave half-empty tiles."""
    B, H, Cin, Cout = case
    rng = np.random.default_rng(H)
    x = rng.standard_normal((B, H, H, Cin)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(Cin * 9)).astype(np.float32)
    bias = rng.standard_normal(Cout).astype(np.float32)
    ref = torch.relu(torch.nn.functional.conv2d(torch.from_numpy(x).permute(0, 3, 1, 2), torch.from_numpy(w),
                                                torch.from_numpy(bias), padding=1)).permute(0, 2, 3, 1).numpy()
    yw, _ = ops.conv2d_nhwc(_t(x, gpu_device), w, bias, None, stride=1, pad=1, relu=True, tile_cfg=-m)
    yd, _ = ops.conv2d_nhwc(_t(x, gpu_device), w, bias, None, stride=1, pad=1, relu=True, tile_cfg=8)
    scale = max(1.0, np.abs(ref).max())
    assert np.abs(yw.cpu().numpy() - ref).max() < 2e-5 * scale
    assert float((yw - yd).abs().max()) < 2e-5 * scale
    # every frame is transformed and multiplied on its own: same bits alone and inside the batch
    y1, _ = ops.conv2d_nhwc(_t(x[1:2], gpu_device), w, bias, None, stride=1, pad=1, relu=True, tile_cfg=-m)
    assert torch.equal(y1[0], yw[1])
    with pytest.raises(_lib.PoseRiskHipError):
        ops.conv2d_nhwc(_t(x, gpu_device), w, bias, None, stride=2, pad=1, relu=True, tile_cfg=-m)


@pytest.mark.parametrize("case", [
    # (B, Ho, C1, H2, C2, stride2, Cout): the four first-Bottleneck tails of ResNet-50 (small batches) + a ragged one
    (2, 56, 64, 56, 64, 1, 256), (2, 28, 128, 56, 256, 2, 512), (3, 14, 256, 28, 512, 2, 1024), (5, 7, 512, 14, 1024, 2, 2048),
    (1, 5, 64, 9, 128, 2, 64)], ids=lambda c: "x".join(map(str, c)))
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_conv_dual_source_matches_torch(gpu_device, case, precision):
    """relu(conv3(t) + downsample(x) + bias) as one GEMM over K = [t's channels | x's channels] (a first Bottleneck's
    tail without the downsample tensor in HBM) against the two torch convolutions; quarter tiles included (fp32)."""
    B, Ho, C1, H2, C2, s2, Cout = case
    rng = np.random.default_rng(sum(case))
    bf = precision == "bf16"
    rnd = (lambda t: t.to(torch.bfloat16).float()) if bf else (lambda t: t)
    t = rnd(torch.from_numpy(rng.standard_normal((B, Ho, Ho, C1)).astype(np.float32)))
    x = rnd(torch.from_numpy(rng.standard_normal((B, H2, H2, C2)).astype(np.float32)))
    w1 = rnd(torch.from_numpy((rng.standard_normal((Cout, C1)) / np.sqrt(C1)).astype(np.float32)))
    w2 = rnd(torch.from_numpy((rng.standard_normal((Cout, C2)) / np.sqrt(C2)).astype(np.float32)))
    bias = rng.standard_normal(Cout).astype(np.float32)
    ref = torch.relu(torch.einsum("bhwc,oc->bhwo", t.double(), w1.double()) +
                     torch.einsum("bhwc,oc->bhwo", x[:, ::s2, ::s2].double(), w2.double()) + torch.from_numpy(bias).double())
    assert ref.shape == (B, Ho, Ho, Cout)
    cfgs = [-1, 8] if not bf else [-1, 8, 11]
    for cfg in cfgs:
        if bf and cfg == 11 and B * Ho * Ho < 256:
            continue
        y = ops.conv1x1_dual_nhwc(t.to(gpu_device), w1.numpy(), x.to(gpu_device), w2.numpy(), bias, stride2=s2, relu=True,
                                  tile_cfg=cfg, precision=precision)
        got = y.float().cpu().double()
        if bf:
            assert y.dtype == torch.bfloat16
            assert bool(((got - ref).abs() <= ref.abs() * 2.0 ** -8 + 2e-3).all()), float((got - ref).abs().max())
        else:
            err = float((got - ref).abs().max())
            assert err < 2e-5 * max(1.0, float(ref.abs().max())), (cfg, err)
    with pytest.raises(_lib.PoseRiskHipError):          # a reserved index of the retired first-generation kernel
        ops.conv1x1_dual_nhwc(t.to(gpu_device), w1.numpy(), x.to(gpu_device), w2.numpy(), bias, stride2=s2, tile_cfg=2,
                              precision=precision)


@pytest.mark.parametrize("case", [(2, 28, 128, 512), (3, 14, 256, 1024), (1, 9, 64, 256), (5, 7, 32, 64)], ids=lambda c: "x".join(map(str, c)))
def test_conv_row_panel_has_the_tile_kernels_bits(gpu_device, case):
    """The row-panel form of a short-K 1x1 convolution (tile_cfg 100: the rows' whole K resident in LDS, W streaming,
    outputs from the fragments) against the 64x64 tile kernel: the same ascending-k fmaf chains, bit for bit; with and
    without residual / bias / ReLU, ragged M, and with the second source of a first block's conv3."""
    B, H, Cin, Cout = case
    g = torch.Generator(device=gpu_device).manual_seed(H + Cin)
    x = torch.randn((B, H, H, Cin), generator=g, device=gpu_device)
    res = torch.randn((B, H, H, Cout), generator=g, device=gpu_device)
    rng = np.random.default_rng(Cout)
    w = (rng.standard_normal((Cout, Cin, 1, 1)) / np.sqrt(Cin)).astype(np.float32)
    bias = rng.standard_normal(Cout).astype(np.float32)
    for b, r, relu in ((bias, res, True), (None, None, False), (bias, None, True)):
        yp, _ = ops.conv2d_nhwc(x, w, b, r, relu=relu, tile_cfg=100)
        yt, _ = ops.conv2d_nhwc(x, w, b, r, relu=relu, tile_cfg=8)
        assert torch.equal(yp, yt)
    ref = torch.relu(torch.einsum("bhwc,oc->bhwo", x.double().cpu(), torch.from_numpy(w[:, :, 0, 0]).double()) +
                     torch.from_numpy(bias).double() + res.double().cpu())
    yp, _ = ops.conv2d_nhwc(x, w, bias, res, relu=True, tile_cfg=100)
    assert float((yp.double().cpu() - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
    if Cin <= 128:      # second source at twice the resolution, stride 2 (K1 + K2 <= 256)
        x2 = torch.randn((B, 2 * H - 1, 2 * H - 1, Cin), generator=g, device=gpu_device)
        w2 = (rng.standard_normal((Cout, Cin)) / np.sqrt(Cin)).astype(np.float32)
        yp = ops.conv1x1_dual_nhwc(x, w[:, :, 0, 0], x2, w2, bias, stride2=2, relu=True, tile_cfg=100)
        yt = ops.conv1x1_dual_nhwc(x, w[:, :, 0, 0], x2, w2, bias, stride2=2, relu=True, tile_cfg=8)
        assert torch.equal(yp, yt)


@pytest.mark.parametrize("case", [(2, 28, 128, 512), (3, 14, 256, 1024), (1, 9, 64, 256)], ids=lambda c: "x".join(map(str, c)))
def test_conv_row_panel_bf16_has_the_tile_kernels_bits(gpu_device, case):
    """bf16 twin of the row-panel kernel against the bf16 64x64 tile kernel: bit for bit (same k order, fp32 accumulate,
    one rounding to bf16), incl. the second source."""
    B, H, Cin, Cout = case
    g = torch.Generator(device=gpu_device).manual_seed(H + Cin)
    x = torch.randn((B, H, H, Cin), generator=g, device=gpu_device).to(torch.bfloat16)
    res = torch.randn((B, H, H, Cout), generator=g, device=gpu_device).to(torch.bfloat16)
    rng = np.random.default_rng(Cout)
    w = (rng.standard_normal((Cout, Cin, 1, 1)) / np.sqrt(Cin)).astype(np.float32)
    bias = rng.standard_normal(Cout).astype(np.float32)
    for b, r, relu in ((bias, res, True), (None, None, False)):
        yp, _ = ops.conv2d_nhwc(x, w, b, r, relu=relu, tile_cfg=100, precision="bf16")
        yt, _ = ops.conv2d_nhwc(x, w, b, r, relu=relu, tile_cfg=8, precision="bf16")
        assert yp.dtype == torch.bfloat16 and torch.equal(yp, yt)
    x2 = torch.randn((B, 2 * H - 1, 2 * H - 1, Cin), generator=g, device=gpu_device).to(torch.bfloat16)
    w2 = (rng.standard_normal((Cout, Cin)) / np.sqrt(Cin)).astype(np.float32)
    yp = ops.conv1x1_dual_nhwc(x, w[:, :, 0, 0], x2, w2, bias, stride2=2, relu=True, tile_cfg=100, precision="bf16")
    yt = ops.conv1x1_dual_nhwc(x, w[:, :, 0, 0], x2, w2, bias, stride2=2, relu=True, tile_cfg=8, precision="bf16")
    assert torch.equal(yp, yt)


@pytest.mark.parametrize("case", [(2, 56, 64, 256, True), (1, 9, 64, 256, True), (3, 14, 64, 128, False), (1, 5, 32, 64, True)],
                         ids=lambda c: "x".join(map(str, c)))
def test_conv3x3_conv1x1_fused_matches_torch(gpu_device, case):
    """A layer1 Bottleneck's conv2 + conv3 + residual in one kernel against the two torch convolutions, and bit for bit
    against the two separate launches of the implicit-GEMM kernel (same ascending-k fmaf chains, t2 rounded to fp32
    at the same point)."""
    B, H, Cin, N3, with_res = case
    rng = np.random.default_rng(H * 7 + N3)
    x = rng.standard_normal((B, H, H, Cin)).astype(np.float32)
    w2 = (rng.standard_normal((64, Cin, 3, 3)) / np.sqrt(Cin * 9)).astype(np.float32)
    b2 = rng.standard_normal(64).astype(np.float32)
    w3 = (rng.standard_normal((N3, 64)) / 8).astype(np.float32)
    b3 = rng.standard_normal(N3).astype(np.float32)
    res = rng.standard_normal((B, H, H, N3)).astype(np.float32) if with_res else None
    t2 = torch.relu(torch.nn.functional.conv2d(torch.from_numpy(x).permute(0, 3, 1, 2).double(), torch.from_numpy(w2).double(),
                                               torch.from_numpy(b2).double(), padding=1))
    ref = torch.einsum("bchw,oc->bhwo", t2, torch.from_numpy(w3).double()) + torch.from_numpy(b3).double()
    if with_res:
        ref = ref + torch.from_numpy(res).double()
    ref = torch.relu(ref).numpy()
    xd = _t(x, gpu_device)
    rd = _t(res, gpu_device) if with_res else None
    y = ops.conv3x3_conv1x1_nhwc(xd, w2, b2, w3, b3, rd, relu=True)
    err = np.abs(y.cpu().numpy() - ref).max()
    assert err < 2e-5 * max(1.0, np.abs(ref).max()), err
    # the same two convolutions as separate launches of the 64x64 tile
    t2d, _ = ops.conv2d_nhwc(xd, w2, b2, None, stride=1, pad=1, relu=True, tile_cfg=8)
    y2, _ = ops.conv2d_nhwc(t2d, w3.reshape(N3, 64, 1, 1), b3, rd, relu=True, tile_cfg=8)
    assert torch.equal(y, y2)
    # without the final ReLU
    y3 = ops.conv3x3_conv1x1_nhwc(xd, w2, b2, w3, b3, rd, relu=False)
    y4, _ = ops.conv2d_nhwc(t2d, w3.reshape(N3, 64, 1, 1), b3, rd, relu=False, tile_cfg=8)
    assert torch.equal(y3, y4)


@pytest.mark.parametrize("case", [(2, 56, 64, 256, True), (1, 9, 64, 256, True), (3, 14, 64, 128, False)],
                         ids=lambda c: "x".join(map(str, c)))
def test_conv3x3_conv1x1_fused_bf16(gpu_device, case):
    """The bf16 twin: against the two separate bf16 launches (t2 rounded to bf16 in both), element for element up to
    the bf16 rounding of sums accumulated in a different tile shape; and against an fp32 emulation."""
    B, H, Cin, N3, with_res = case
    rng = np.random.default_rng(H * 11 + N3)
    bf = lambda t: t.to(torch.bfloat16).float()
    x = bf(torch.from_numpy(rng.standard_normal((B, H, H, Cin)).astype(np.float32)))
    w2 = bf(torch.from_numpy((rng.standard_normal((64, Cin, 3, 3)) / np.sqrt(Cin * 9)).astype(np.float32)))
    b2 = rng.standard_normal(64).astype(np.float32)
    w3 = bf(torch.from_numpy((rng.standard_normal((N3, 64)) / 8).astype(np.float32)))
    b3 = rng.standard_normal(N3).astype(np.float32)
    res = bf(torch.from_numpy(rng.standard_normal((B, H, H, N3)).astype(np.float32))) if with_res else None
    t2 = bf(torch.relu(torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), w2, torch.from_numpy(b2), padding=1)))
    ref = torch.einsum("bchw,oc->bhwo", t2, w3) + torch.from_numpy(b3)
    if with_res:
        ref = ref + res
    ref = torch.relu(ref)
    xd = x.to(gpu_device)
    rd = res.to(gpu_device) if with_res else None
    y = ops.conv3x3_conv1x1_nhwc(xd, w2.numpy(), b2, w3.numpy(), b3, rd, relu=True, precision="bf16")
    assert y.dtype == torch.bfloat16
    got = y.float().cpu()
    # t2 values at a bf16 rounding boundary may round either way (fp32 accumulation order): allow what one such flip
    # moves the output by, on top of the output's own bf16 ulp
    tol = ref.abs() * 2.0 ** -7 + 2e-2
    assert bool(((got - ref).abs() <= tol).all()), float((got - ref).abs().max())
    t2d, _ = ops.conv2d_nhwc(xd, w2.numpy(), b2, None, stride=1, pad=1, relu=True, tile_cfg=8, precision="bf16")
    y2, _ = ops.conv2d_nhwc(t2d, w3.numpy().reshape(N3, 64, 1, 1), b3, rd, relu=True, tile_cfg=8, precision="bf16")
    assert torch.equal(y, y2)          # same 64x64 tiles, same k order: the same bits


@pytest.mark.parametrize("rows", [(1, 5), (1, 3), (3, 5), (2, 9)], ids=lambda c: "x".join(map(str, c)))
def test_fragment_epilogues_stay_inside_ragged_outputs(gpu_device, rows):
    """The fused conv2+conv3 kernel, the row-panel kernel and the dual-source form store straight from the accumulator
    fragments through range-checked buffer descriptors, rows >= M of a ragged last tile (incl. M < 28, fewer rows than
    one fragment spans) getting an out-of-range vector offset.  The outputs live inside a larger buffer with canary rows
    behind row M, which must come back untouched (and NaNs behind the residual must not be read)."""
    B, H = rows
    M = B * H * H
    rng = np.random.default_rng(M)
    CANARY = -12345.0

    def guarded(C):
        big = torch.full((M + 96, C), CANARY, dtype=torch.float32, device=gpu_device)
        return big, big[:M].view(B, H, H, C)

    def check(big, y, ref, what):
        assert bool((big[M:] == CANARY).all()), f"{what}: rows behind the output were written"
        err = (y.cpu() - ref).abs().max().item()
        assert err < 2e-5 * max(1.0, ref.abs().max().item()), (what, err)

    x = torch.from_numpy(rng.standard_normal((B, H, H, 64)).astype(np.float32))
    w2 = (rng.standard_normal((64, 64, 3, 3)) / 24).astype(np.float32)
    b2 = rng.standard_normal(64).astype(np.float32)
    w3 = (rng.standard_normal((256, 64)) / 8).astype(np.float32)
    b3 = rng.standard_normal(256).astype(np.float32)
    resbig, res = guarded(256)
    res.copy_(torch.from_numpy(rng.standard_normal((B, H, H, 256)).astype(np.float32)))
    resbig[M:] = float("nan")              # a residual row read from behind the tensor would poison the output
    t2 = torch.relu(torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), torch.from_numpy(w2).double(),
                                               torch.from_numpy(b2).double(), padding=1))
    ref = torch.relu(torch.einsum("bchw,oc->bhwo", t2, torch.from_numpy(w3).double()) + torch.from_numpy(b3).double()
                     + res.cpu().double()).float()
    big, y = guarded(256)
    ops.conv3x3_conv1x1_nhwc(x.to(gpu_device), w2, b2, w3, b3, res, relu=True, out=y)
    check(big, y, ref, "conv3x3_conv1x1_f32")
    # row panel (tile_cfg 100): 1x1, K = 128, with residual
    x1 = torch.from_numpy(rng.standard_normal((B, H, H, 128)).astype(np.float32))
    w = (rng.standard_normal((256, 128, 1, 1)) / 11).astype(np.float32)
    ref = torch.relu(torch.einsum("bhwc,oc->bhwo", x1.double(), torch.from_numpy(w[:, :, 0, 0]).double())
                     + torch.from_numpy(b3).double() + res.cpu().double()).float()
    big, y = guarded(256)
    ops.conv2d_nhwc(x1.to(gpu_device), w, b3, res, relu=True, tile_cfg=100, out=y)
    check(big, y, ref, "conv1x1_panel_f32")
    # dual-source row panel: K = 64 + 64
    xa = torch.from_numpy(rng.standard_normal((B, H, H, 64)).astype(np.float32))
    xb = torch.from_numpy(rng.standard_normal((B, H, H, 64)).astype(np.float32))
    wa = (rng.standard_normal((256, 64)) / 8).astype(np.float32)
    wb = (rng.standard_normal((256, 64)) / 8).astype(np.float32)
    ref = torch.relu(torch.einsum("bhwc,oc->bhwo", xa.double(), torch.from_numpy(wa).double())
                     + torch.einsum("bhwc,oc->bhwo", xb.double(), torch.from_numpy(wb).double())
                     + torch.from_numpy(b3).double()).float()
    big, y = guarded(256)
    ops.conv1x1_dual_nhwc(xa.to(gpu_device), wa, xb.to(gpu_device), wb, b3, relu=True, tile_cfg=100, out=y)
    check(big, y, ref, "conv1x1_panel_f32 (two sources)")


@pytest.mark.parametrize("case", [(3, 14, 256, 512, 1024), (5, 7, 512, 1024, 2048), (2, 9, 128, 256, 512), (1, 1, 64, 64, 128),
                                  (100, 14, 256, 512, 1024)], ids=lambda c: "x".join(map(str, c)))
def test_conv_bal_bf16_second_source(gpu_device, case):
    """A first block's conv3 with its downsample branch as the second, stride-2 pixel source of the evenly dealt kernel
    (tile_cfg 301 of the dual-source entry): bit for bit against the tile kernel's dual-source K loop -- layer3's and
    layer4's shapes, odd output sizes, a single pixel, several chunks per workgroup."""
    B, Ho, C1, C2, N = case
    H2 = 2 * Ho - 1 if Ho % 2 else 2 * Ho
    rng = np.random.default_rng(B * 7 + Ho)
    bf = lambda t: t.to(torch.bfloat16)
    t = bf(torch.from_numpy(rng.standard_normal((B, Ho, Ho, C1)).astype(np.float32))).to(gpu_device)
    x2 = bf(torch.from_numpy(rng.standard_normal((B, H2, H2, C2)).astype(np.float32))).to(gpu_device)
    w1 = (rng.standard_normal((N, C1)) / np.sqrt(C1)).astype(np.float32)
    w2 = (rng.standard_normal((N, C2)) / np.sqrt(C2)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    for relu in (True, False):
        y2 = ops.conv1x1_dual_nhwc(t, w1, x2, w2, bias, stride2=2, relu=relu, tile_cfg=13, precision="bf16")
        for rep in range(2):
            y = ops.conv1x1_dual_nhwc(t, w1, x2, w2, bias, stride2=2, relu=relu, tile_cfg=301, precision="bf16")
            assert y.shape == y2.shape and torch.equal(y, y2), (relu, rep, int((y != y2).sum()))
    assert float(y.float().abs().max()) > 0


@pytest.mark.parametrize("case", [(2, 28), (1, 5), (3, 9), (1, 1), (90, 28)], ids=lambda c: "x".join(map(str, c)))
def test_expand_dual_bf16_downsample_in_the_k_loop(gpu_device, case):
    """layer2's first conv3 with its downsample branch as a second, stride-2 source of the register-resident-weights kernel
    (tile_cfg 300 of the dual-source entry, K = 128 + 256, N = 512 as two column blocks): bit for bit against the tile
    kernel's dual-source K loop, odd output sizes (the source is 2 Ho - 1 wide), a single pixel, runs of several blocks per
    workgroup, repeated launches."""
    B, Ho = case
    H2 = 2 * Ho - 1 if Ho % 2 else 2 * Ho
    rng = np.random.default_rng(B * 7 + Ho)
    bf = lambda t: t.to(torch.bfloat16)
    t = bf(torch.from_numpy(rng.standard_normal((B, Ho, Ho, 128)).astype(np.float32))).to(gpu_device)
    x2 = bf(torch.from_numpy(rng.standard_normal((B, H2, H2, 256)).astype(np.float32))).to(gpu_device)
    w1 = (rng.standard_normal((512, 128)) / 11).astype(np.float32)
    w2 = (rng.standard_normal((512, 256)) / 16).astype(np.float32)
    bias = rng.standard_normal(512).astype(np.float32)
    for relu in (True, False):
        y2 = ops.conv1x1_dual_nhwc(t, w1, x2, w2, bias, stride2=2, relu=relu, tile_cfg=13, precision="bf16")
        for rep in range(3):
            y = ops.conv1x1_dual_nhwc(t, w1, x2, w2, bias, stride2=2, relu=relu, tile_cfg=300, precision="bf16")
            assert y.shape == y2.shape and torch.equal(y, y2), (relu, rep, int((y != y2).sum()))
    assert float(y.float().abs().max()) > 0


@pytest.mark.parametrize("kn", [(128, 512), (256, 1024)], ids=lambda c: "x".join(map(str, c)))
def test_expand_res_bf16_many_blocks_per_workgroup_repeatedly(gpu_device, kn):
    """The expansion kernel's steady state -- several blocks per workgroup, back-to-back launches, the stores of one block
    in flight under the next -- bit for bit against the tile kernel, six times over.  This is the shape that exposed the
    wide-store hazard (profiles/r03_t_store_hazard.txt): with a plain buffer_store_dwordx4 behind which the compiler
    reused the data registers, ~1 500 of 25.7 M elements came out as the next tile's intermediate values."""
    K, N = kn
    B, H = (64, 28) if K == 128 else (200, 14)
    rng = np.random.default_rng(K)
    x = torch.randn((B, H, H, K), device=gpu_device).bfloat16()
    res = torch.randn((B, H, H, N), device=gpu_device).bfloat16()
    w = (rng.standard_normal((N, K, 1, 1)) / 11).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    y2, _ = ops.conv2d_nhwc(x, w, bias, res, relu=True, tile_cfg=13, precision="bf16")
    for rep in range(6):
        y, _ = ops.conv2d_nhwc(x, w, bias, res, relu=True, tile_cfg=300, precision="bf16", repeats=3)
        assert torch.equal(y, y2), (rep, int((y != y2).sum()))


def test_conv_bal_bf16_at_full_size_repeatedly(gpu_device):
    """layer3's conv1 and conv2 at B = 256 on the evenly dealt kernel, back-to-back launches, against the tile kernel: the
    same bits every time (the full-size companion of test_conv_bal_bf16_equals_tile_kernel)."""
    rng = np.random.default_rng(3)
    for Cin, k in ((1024, 1), (256, 3)):
        x = torch.randn((256, 14, 14, Cin), device=gpu_device).bfloat16()
        w = (rng.standard_normal((256, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
        bias = rng.standard_normal(256).astype(np.float32)
        y2, _ = ops.conv2d_nhwc(x, w, bias, None, pad=k // 2, relu=True, tile_cfg=13, precision="bf16")
        for rep in range(4):
            y, _ = ops.conv2d_nhwc(x, w, bias, None, pad=k // 2, relu=True, tile_cfg=301, precision="bf16", repeats=3)
            assert torch.equal(y, y2), (Cin, k, rep, int((y != y2).sum()))


_BAL_CASES = [
    # B, H, Cin, Cout, k, stride, tile_cfg
    (3, 14, 1024, 256, 1, 1, 301),        # layer3 conv1
    (3, 14, 256, 256, 3, 1, 301),         # layer3 conv2
    (2, 28, 256, 256, 3, 2, 301),         # layer3 first block's conv2 (stride 2)
    (2, 28, 512, 128, 1, 1, 301),         # layer2 conv1: channel blocks of 128, chunks of <= 12 pixel tiles
    (2, 28, 128, 128, 3, 1, 301),         # layer2 conv2
    (5, 7, 512, 512, 3, 1, 301),          # layer4 conv2: two channel blocks of 256
    (5, 7, 512, 512, 3, 1, 302),          # ... as four blocks of 128
    (1, 9, 64, 128, 1, 1, 301),           # two stages, ragged tiles
    (7, 5, 128, 256, 1, 1, 302),          # partner workgroups
    (1, 1, 64, 128, 3, 1, 301),           # a single pixel: most workgroups have nothing to do
    (64, 56, 64, 256, 1, 1, 301),         # 6272 pixel tiles: runs of 24 / 25 tiles in chunks of 6 and 7
    (64, 56, 64, 128, 1, 1, 301),         # ... in chunks of 8 and 9 (groups of 3, 2, 2, 2 tiles)
    (40, 28, 64, 128, 3, 1, 301),         # 980 tiles: one chunk of 3 or 4 per workgroup, groups with no tile at all
    (7, 5, 192, 256, 3, 1, 301),          # three 64-channel slices: Cin need not be a power of two (slice-major K)
]


@pytest.mark.parametrize("case", _BAL_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_bal_bf16_equals_tile_kernel(gpu_device, case):
    """The persistent, evenly dealt bf16 convolution (tile_cfg 301 / 302, csrc/conv_bal_bf16.hip) against the tile kernel:
    the same products in the same k order and the same epilogue arithmetic, so the same bits -- for every chunk size,
    1x1 and 3x3, stride 2, ragged last tiles, one to four channel blocks."""
    B, H, Cin, Cout, k, stride, cfg = case
    rng = np.random.default_rng(B * 1000 + H * 10 + k)
    x = torch.from_numpy(rng.standard_normal((B, H, H, Cin)).astype(np.float32)).to(torch.bfloat16).to(gpu_device)
    w = (rng.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
    bias = rng.standard_normal(Cout).astype(np.float32)
    for relu in (True, False):
        y, _ = ops.conv2d_nhwc(x, w, bias, None, stride=stride, pad=k // 2, relu=relu, tile_cfg=cfg, precision="bf16")
        y2, _ = ops.conv2d_nhwc(x, w, bias, None, stride=stride, pad=k // 2, relu=relu, tile_cfg=13, precision="bf16")
        assert y.shape == y2.shape and torch.equal(y, y2), (relu, int((y != y2).sum()), y.numel())
        assert float(y.float().abs().max()) > 0


@pytest.mark.parametrize("case", [(4, 28), (1, 9), (3, 5), (1, 1), (70, 14)], ids=lambda c: "x".join(map(str, c)))
@pytest.mark.parametrize("kn", [(128, 512), (256, 1024)], ids=lambda c: "x".join(map(str, c)))
@pytest.mark.parametrize("relu", [True, False])
def test_expand_res_bf16_weights_in_registers(gpu_device, case, kn, relu):
    """conv3 of layer2's / layer3's plain blocks (1x1, 128 -> 512 or 256 -> 1024) + bias + residual + ReLU as the persistent
    kernel that keeps the weights in registers (tile_cfg 300, csrc/expand_res_bf16.hip; the 1024-channel form as two
    column blocks of 512 on partner workgroups): bit for bit against the tile kernel (same 16-wide MFMA groups, same k
    order, (acc + bias) + res), ragged last blocks and runs of several blocks per workgroup included."""
    B, H = case
    K, N = kn
    rng = np.random.default_rng(B * 31 + H + K)
    bf = lambda t: t.to(torch.bfloat16)
    x = bf(torch.from_numpy(rng.standard_normal((B, H, H, K)).astype(np.float32))).to(gpu_device)
    res = bf(torch.from_numpy(rng.standard_normal((B, H, H, N)).astype(np.float32))).to(gpu_device)
    w = (rng.standard_normal((N, K, 1, 1)) / 11).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    y, _ = ops.conv2d_nhwc(x, w, bias, res, relu=relu, tile_cfg=300, precision="bf16")
    y2, _ = ops.conv2d_nhwc(x, w, bias, res, relu=relu, tile_cfg=13, precision="bf16")
    assert y.dtype == torch.bfloat16 and torch.equal(y, y2), int((y != y2).sum())
    ref = torch.einsum("bhwc,oc->bhwo", x.float().cpu(), bf(torch.from_numpy(w[:, :, 0, 0])).float()) + torch.from_numpy(bias) + res.float().cpu()
    ref = torch.relu(ref) if relu else ref
    assert bool(((y.float().cpu() - ref).abs() <= ref.abs() * 2.0 ** -8 + 2e-3).all())


@pytest.mark.parametrize("case", [(2, 56, 56), (1, 9, 9), (3, 14, 14), (5, 7, 7), (2, 13, 6), (1, 3, 63), (7, 1, 1)],
                         ids=lambda c: "x".join(map(str, c)))
def test_bottleneck_bf16_whole_block_in_one_kernel(gpu_device, case):
    """A whole layer1 Bottleneck (conv1 -> conv2 -> conv3 + x) as ONE persistent kernel (csrc/bottleneck_bf16.hip):
    bit for bit against the separate bf16 launches it replaces (conv1 on the tile kernel, conv2 + conv3 fused: the same
    16-wide MFMA groups in the same k order, t1 and t2 rounded to bf16 where those launches store them), and against an
    fp32 emulation with bf16-rounded intermediates.  Ragged maps, a map narrower than a block, single pixels."""
    B, H, W = case
    rng = np.random.default_rng(B * 1000 + H * 10 + W)
    bf = lambda t: t.to(torch.bfloat16).float()
    x = bf(torch.from_numpy(rng.standard_normal((B, H, W, 256)).astype(np.float32)))
    w1 = bf(torch.from_numpy((rng.standard_normal((64, 256)) / 16).astype(np.float32)))
    w2 = bf(torch.from_numpy((rng.standard_normal((64, 64, 3, 3)) / 24).astype(np.float32)))
    w3 = bf(torch.from_numpy((rng.standard_normal((256, 64)) / 8).astype(np.float32)))
    b1, b2, b3 = (rng.standard_normal(n).astype(np.float32) * 0.5 for n in (64, 64, 256))
    xd = x.to(gpu_device)
    y, _ = ops.bottleneck_nhwc(xd, w1.numpy(), b1, w2.numpy(), b2, w3.numpy(), b3)
    assert y.dtype == torch.bfloat16 and y.shape == xd.shape
    # the three convolutions as the launches the encoder used before
    t1, _ = ops.conv2d_nhwc(xd, w1.numpy().reshape(64, 256, 1, 1), b1, None, relu=True, tile_cfg=8, precision="bf16")
    y2 = ops.conv3x3_conv1x1_nhwc(t1, w2.numpy(), b2, w3.numpy(), b3, xd, relu=True, precision="bf16")
    nbad = int((y != y2).sum())
    measured("bottleneck64_bf16 vs separate launches: differing elements", nbad, 0)
    # fp32 emulation, intermediates rounded to bf16
    e1 = bf(torch.relu(torch.einsum("bhwc,oc->bhwo", x, w1) + torch.from_numpy(b1)))
    e2 = bf(torch.relu(torch.nn.functional.conv2d(e1.permute(0, 3, 1, 2), w2, torch.from_numpy(b2), padding=1)))
    ref = torch.relu(torch.einsum("bchw,oc->bhwo", e2, w3) + torch.from_numpy(b3) + x)
    got = y.float().cpu()
    tol = ref.abs() * 2.0 ** -7 + 3e-2
    assert bool(((got - ref).abs() <= tol).all()), float((got - ref).abs().max())
    assert nbad == 0


@pytest.mark.parametrize("case", [(2, 28, 28), (1, 9, 9), (3, 14, 14), (5, 7, 7), (2, 13, 6), (1, 3, 31), (7, 1, 1), (40, 28, 28)],
                         ids=lambda c: "x".join(map(str, c)))
def test_bottleneck128_bf16_whole_block_in_one_kernel(gpu_device, case):
    """A whole layer2 Bottleneck (512 -> 128 -> 128 -> 512 + x) as ONE persistent kernel (csrc/bottleneck128_bf16.hip: the
    weights stream through LDS once per chunk of <= 256 pixels, the chunk's accumulators stay in registers): bit for bit
    against the three separate bf16 launches (same 16-wide MFMA groups in the same slice-major k order, t1 and t2 rounded
    to bf16 where those launches store them), and against an fp32 emulation with bf16-rounded intermediates.  Ragged maps,
    the widest map the halo allows, single pixels, several chunks per workgroup with images straddling them."""
    B, H, W = case
    rng = np.random.default_rng(B * 1000 + H * 10 + W)
    bf = lambda t: t.to(torch.bfloat16).float()
    x = bf(torch.from_numpy(rng.standard_normal((B, H, W, 512)).astype(np.float32)))
    w1 = bf(torch.from_numpy((rng.standard_normal((128, 512)) / 22).astype(np.float32)))
    w2 = bf(torch.from_numpy((rng.standard_normal((128, 128, 3, 3)) / 34).astype(np.float32)))
    w3 = bf(torch.from_numpy((rng.standard_normal((512, 128)) / 11).astype(np.float32)))
    b1, b2, b3 = (rng.standard_normal(n).astype(np.float32) * 0.5 for n in (128, 128, 512))
    xd = x.to(gpu_device)
    y, _ = ops.bottleneck128_nhwc(xd, w1.numpy(), b1, w2.numpy(), b2, w3.numpy(), b3)
    assert y.dtype == torch.bfloat16 and y.shape == xd.shape
    t1, _ = ops.conv2d_nhwc(xd, w1.numpy().reshape(128, 512, 1, 1), b1, None, relu=True, tile_cfg=13, precision="bf16")
    t2, _ = ops.conv2d_nhwc(t1, w2.numpy(), b2, None, pad=1, relu=True, tile_cfg=13, precision="bf16")
    y2, _ = ops.conv2d_nhwc(t2, w3.numpy().reshape(512, 128, 1, 1), b3, xd, relu=True, tile_cfg=13, precision="bf16")
    nbad = int((y != y2).sum())
    measured("bottleneck128_bf16 vs separate launches: differing elements", nbad, 0)
    if B * H * W <= 4000:
        e1 = bf(torch.relu(torch.einsum("bhwc,oc->bhwo", x, w1) + torch.from_numpy(b1)))
        e2 = bf(torch.relu(torch.nn.functional.conv2d(e1.permute(0, 3, 1, 2), w2, torch.from_numpy(b2), padding=1)))
        ref = torch.relu(torch.einsum("bchw,oc->bhwo", e2, w3) + torch.from_numpy(b3) + x)
        got = y.float().cpu()
        tol = ref.abs() * 2.0 ** -7 + 5e-2
        assert bool(((got - ref).abs() <= tol).all()), float((got - ref).abs().max())
    assert nbad == 0


@pytest.mark.parametrize("case", [(2, 14, 14), (1, 9, 9), (3, 7, 7), (2, 13, 6), (1, 3, 31), (5, 1, 1), (1, 14, 16), (1, 4, 8), (300, 14, 14)],
                         ids=lambda c: "x".join(map(str, c)))
def test_bottleneck256_bf16_whole_block_in_one_kernel(gpu_device, case):
    """A whole layer3 Bottleneck (1024 -> 256 -> 256 -> 1024 + x) as ONE kernel, one frame per workgroup
    (csrc/bottleneck256_bf16.hip: t1 and t2 of the frame live in LDS, W2's and W3's MFMA fragments come straight from L2 into
    registers): bit for bit against the three separate bf16 launches (same 16-wide MFMA groups in the same slice-major k
    order, t1 and t2 rounded to bf16 where those launches store them), and against an fp32 emulation with bf16-rounded
    intermediates.  Ragged maps, the largest map a frame may have (224 pixels), maps of one to seven pixel tiles (so every
    split of the tiles between the two wave groups), single pixels, more frames than CUs."""
    B, H, W = case
    rng = np.random.default_rng(B * 1000 + H * 10 + W)
    bf = lambda t: t.to(torch.bfloat16).float()
    x = bf(torch.from_numpy(rng.standard_normal((B, H, W, 1024)).astype(np.float32)))
    w1 = bf(torch.from_numpy((rng.standard_normal((256, 1024)) / 32).astype(np.float32)))
    w2 = bf(torch.from_numpy((rng.standard_normal((256, 256, 3, 3)) / 48).astype(np.float32)))
    w3 = bf(torch.from_numpy((rng.standard_normal((1024, 256)) / 16).astype(np.float32)))
    b1, b2, b3 = (rng.standard_normal(n).astype(np.float32) * 0.5 for n in (256, 256, 1024))
    xd = x.to(gpu_device)
    y, _ = ops.bottleneck256_nhwc(xd, w1.numpy(), b1, w2.numpy(), b2, w3.numpy(), b3)
    assert y.dtype == torch.bfloat16 and y.shape == xd.shape
    t1, _ = ops.conv2d_nhwc(xd, w1.numpy().reshape(256, 1024, 1, 1), b1, None, relu=True, tile_cfg=13, precision="bf16")
    t2, _ = ops.conv2d_nhwc(t1, w2.numpy(), b2, None, pad=1, relu=True, tile_cfg=13, precision="bf16")
    y2, _ = ops.conv2d_nhwc(t2, w3.numpy().reshape(1024, 256, 1, 1), b3, xd, relu=True, tile_cfg=13, precision="bf16")
    nbad = int((y != y2).sum())
    measured("bottleneck256_bf16 vs separate launches: differing elements", nbad, 0)
    if B * H * W <= 2000:
        e1 = bf(torch.relu(torch.einsum("bhwc,oc->bhwo", x, w1) + torch.from_numpy(b1)))
        e2 = bf(torch.relu(torch.nn.functional.conv2d(e1.permute(0, 3, 1, 2), w2, torch.from_numpy(b2), padding=1)))
        ref = torch.relu(torch.einsum("bchw,oc->bhwo", e2, w3) + torch.from_numpy(b3) + x)
        got = y.float().cpu()
        tol = ref.abs() * 2.0 ** -7 + 5e-2
        assert bool(((got - ref).abs() <= tol).all()), float((got - ref).abs().max())
    assert nbad == 0


def test_hmr_fused_downsample_equals_separate_launches(gpu_device):
    """The encoder sums each first Bottleneck's downsample branch into its conv3's K loop (49 launches for the 53
    convolutions).  Against the same network with the branch as its own launch + residual add (environment switch of
    the A/B timing, own process): same results up to the summation order of one 1x1 convolution."""
    import os, subprocess, sys
    from conftest import REPO
    code = ("import sys, numpy as np, torch; sys.path.insert(0, %r)\n"
            "from poserisk_release_amd import synth\nfrom poserisk_release_amd.hmr import HMR\n"
            "m = HMR(max_batch=4, conv_form='direct').to('cuda:0'); m.load_state_dict(synth.hmr_state_dict(seed=1))\n"
            "r, b, c, xf, _ = m(torch.from_numpy(synth.crops(4, seed=8)).cuda(), return_features=True)\n"
            "np.savez(sys.argv[1], r=r.cpu().numpy(), b=b.cpu().numpy(), xf=xf.cpu().numpy())\n") % REPO
    outs = []
    for flag in ("1", "0"):
        path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"pr_fuse_{os.getpid()}_{flag}.npz")
        r = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, POSERISK_FUSE_DOWNSAMPLE=flag),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(dict(np.load(path)))
        os.remove(path)
    fused, sep = outs
    exf = np.abs(fused["xf"] - sep["xf"]).max() / np.abs(sep["xf"]).max()
    er = np.abs(fused["r"] - sep["r"]).max()
    measured("hmr fused vs separate downsample: xf (relative to max)", exf, 5e-6)
    measured("hmr fused vs separate downsample: rotmat", er, 2e-5)
    assert 0 < exf < 5e-6 and er < 2e-5


@pytest.mark.parametrize("case", [(2, 56, 56), (1, 9, 9), (3, 14, 14), (2, 13, 6), (7, 1, 1)], ids=lambda c: "x".join(map(str, c)))
def test_bottleneck_bf16_first_block_in_one_kernel(gpu_device, case):
    """The stage's first block (64-channel input, downsample branch summed into conv3's K loop, no residual) as one
    launch of the same kernel: bit for bit against the three launches it replaces (conv1, conv2, dual-source conv3)."""
    B, H, W = case
    rng = np.random.default_rng(B * 999 + H * 10 + W)
    bf = lambda t: t.to(torch.bfloat16).float()
    x = bf(torch.from_numpy(rng.standard_normal((B, H, W, 64)).astype(np.float32)))
    w1 = bf(torch.from_numpy((rng.standard_normal((64, 64)) / 8).astype(np.float32)))
    w2 = bf(torch.from_numpy((rng.standard_normal((64, 64, 3, 3)) / 24).astype(np.float32)))
    w3 = bf(torch.from_numpy((rng.standard_normal((256, 64)) / 8).astype(np.float32)))
    wd = bf(torch.from_numpy((rng.standard_normal((256, 64)) / 8).astype(np.float32)))
    b1, b2, b3, bd = (rng.standard_normal(n).astype(np.float32) * 0.5 for n in (64, 64, 256, 256))
    xd = x.to(gpu_device)
    y, _ = ops.bottleneck_nhwc(xd, w1.numpy(), b1, w2.numpy(), b2, w3.numpy(), b3, wd=wd.numpy(), bd=bd)
    assert y.dtype == torch.bfloat16 and y.shape == (B, H, W, 256)
    t1, _ = ops.conv2d_nhwc(xd, w1.numpy().reshape(64, 64, 1, 1), b1, None, relu=True, tile_cfg=8, precision="bf16")
    t2, _ = ops.conv2d_nhwc(t1, w2.numpy(), b2, None, stride=1, pad=1, relu=True, tile_cfg=8, precision="bf16")
    b3d = (b3.astype(np.float64) + bd.astype(np.float64)).astype(np.float32)
    y2 = ops.conv1x1_dual_nhwc(t2, w3.numpy(), xd, wd.numpy(), b3d, relu=True, tile_cfg=8, precision="bf16")
    nbad = int((y != y2).sum())
    measured("bottleneck64_bf16 (first block) vs separate launches: differing elements", nbad, 0)
    e1 = bf(torch.relu(torch.einsum("bhwc,oc->bhwo", x, w1) + torch.from_numpy(b1)))
    e2 = bf(torch.relu(torch.nn.functional.conv2d(e1.permute(0, 3, 1, 2), w2, torch.from_numpy(b2), padding=1)))
    ref = torch.relu(torch.einsum("bchw,oc->bhwo", e2, w3) + torch.einsum("bhwc,oc->bhwo", x, wd) + torch.from_numpy(b3d))
    got = y.float().cpu()
    tol = ref.abs() * 2.0 ** -7 + 3e-2
    assert bool(((got - ref).abs() <= tol).all()), float((got - ref).abs().max())
    assert nbad == 0


@pytest.mark.parametrize("case", [(2, 112), (3, 28), (1, 30), (5, 2), (2, 58)], ids=lambda c: "x".join(map(str, c)))
def test_stem_pool_bf16_in_one_kernel(gpu_device, case):
    """The bf16 stem as one kernel (4x4 / stride-1 conv on the space-to-depth image + bias + ReLU + 3x3 / stride-2 max-pool,
    the conv map never leaving the chip) against torch: conv in fp32 on the bf16-rounded operands, the map rounded to bf16
    before the pool as the two launches store it.  A conv value at a bf16 rounding boundary may round either way (fp32
    summation order), and the pool passes such a flip on: one bf16 ulp of slack.  Whole and ragged bands, a 1x1 pooled map."""
    B, H = case
    rng = np.random.default_rng(B * 100 + H)
    bf = lambda t: t.to(torch.bfloat16).float()
    x = bf(torch.from_numpy(rng.random((B, H, H, 16)).astype(np.float32)))
    x[..., 12:] = 0                                  # the space-to-depth image has 12 real channels
    w = bf(torch.from_numpy((rng.standard_normal((64, 16, 4, 4)) / 12).astype(np.float32)))
    bias = rng.standard_normal(64).astype(np.float32) * 0.3
    y, _ = ops.stem_pool_nhwc(x.to(gpu_device), w.numpy(), bias)
    conv = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), w, torch.from_numpy(bias), padding=2)[:, :, :H, :H]
    ref = torch.nn.functional.max_pool2d(bf(torch.relu(conv)), 3, stride=2, padding=1).permute(0, 2, 3, 1)
    assert y.shape == ref.shape and y.dtype == torch.bfloat16
    got = y.float().cpu()
    tol = ref.abs() * 2.0 ** -7 + 1e-6
    bad = (got - ref).abs() > tol
    assert not bool(bad.any()), (int(bad.sum()), float((got - ref).abs().max()))
    exact = float((got == ref).float().mean())
    measured("stem_pool_bf16 vs torch: elements equal bit for bit", exact, None)
    assert exact > 0.98


@pytest.mark.parametrize("switch,batch", [("POSERISK_FUSE_STEM", 3), ("POSERISK_EXPAND_REGS", 64), ("POSERISK_BALANCED", 64), ("POSERISK_FUSE_BOTTLENECK2", 20), ("POSERISK_B128_LEAD", 64), ("POSERISK_FUSE_BOTTLENECK3", 230)])
def test_hmr_bf16_fused_stem_equals_separate_launches(gpu_device, switch, batch):
    """The bf16 encoder with its stem as one kernel against the same network with conv1 and the max-pool as two launches;
    with layer2's and layer3's expansions on the register-resident-weights kernel against the tile kernel (batch 64:
    several blocks per workgroup); and, at a batch where the evenly dealt persistent kernel takes layers (64: layer2's
    first 1x1 reduction and layer3's first), against the tile kernel everywhere; and with layer2's plain blocks as one
    kernel each against the three launches per block; and with every second workgroup of that kernel opening with a short
    chunk against equal chunks; and, at a batch that fills the CUs (230 frames), with layer3's plain blocks as one launch each
    (a frame per workgroup) against three launches per block (environment switches of the A/B timing, own process): the same bits."""
    import os, subprocess, sys
    from conftest import REPO
    code = ("import sys, numpy as np, torch; sys.path.insert(0, %r)\n"
            "from poserisk_release_amd import synth\nfrom poserisk_release_amd.hmr import HMR\n"
            "B = %d\n"
            "m = HMR(max_batch=B, precision='bf16').to('cuda:0'); m.load_state_dict(synth.hmr_state_dict(seed=1))\n"
            "r, b, c, xf, _ = m(torch.from_numpy(synth.crops(B, seed=9)).cuda(), return_features=True)\n"
            "np.savez(sys.argv[1], r=r.cpu().numpy(), xf=xf.cpu().numpy())\n") % (REPO, batch)
    outs = []
    for flag in ("1", "0"):
        path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"pr_stem_{os.getpid()}_{flag}.npz")
        r = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, **{switch: flag}),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(dict(np.load(path)))
        os.remove(path)
    fused, sep = outs
    assert np.array_equal(fused["xf"], sep["xf"]) and np.array_equal(fused["r"], sep["r"])
    assert np.abs(fused["xf"]).max() > 0


def test_hmr_bf16_whole_bottleneck_kernel_equals_separate_launches(gpu_device):
    """The bf16 encoder runs layer1's blocks 1 and 2 as ONE kernel each (conv1 -> conv2 -> conv3 + x).  Against the same
    network with those blocks as separate launches (environment switch of the A/B timing, own process): the same bits,
    pooled features and regressor outputs alike."""
    import os, subprocess, sys
    from conftest import REPO
    code = ("import sys, numpy as np, torch; sys.path.insert(0, %r)\n"
            "from poserisk_release_amd import synth\nfrom poserisk_release_amd.hmr import HMR\n"
            "m = HMR(max_batch=5, precision='bf16').to('cuda:0'); m.load_state_dict(synth.hmr_state_dict(seed=1))\n"
            "r, b, c, xf, _ = m(torch.from_numpy(synth.crops(5, seed=8)).cuda(), return_features=True)\n"
            "np.savez(sys.argv[1], r=r.cpu().numpy(), b=b.cpu().numpy(), xf=xf.cpu().numpy())\n") % REPO
    outs = []
    for flag in ("1", "0"):
        path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"pr_bneck_{os.getpid()}_{flag}.npz")
        r = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, POSERISK_FUSE_BOTTLENECK=flag),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(dict(np.load(path)))
        os.remove(path)
    fused, sep = outs
    assert np.array_equal(fused["xf"], sep["xf"]) and np.array_equal(fused["r"], sep["r"]) and np.array_equal(fused["b"], sep["b"])
    assert np.abs(fused["xf"]).max() > 0


# ------------------------------------------------------------------------------------------------
# HMR encoder + regressor vs the torch-CPU restatement (parity unpinned upstream: SURVEY 8c)
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def hmr_pair(gpu_device):
    sd = synth.hmr_state_dict(seed=1)
    ref = hmr_ref.build(sd)
    m = HMR(max_batch=8).to(gpu_device)
    m.load_state_dict(sd)
    return m.eval(), ref


def test_hmr_forward_matches_oracle(gpu_device, hmr_pair):
    m, ref = hmr_pair
    x = synth.crops(4, seed=0)
    with torch.no_grad():
        xf_ref = ref.features(torch.from_numpy(x))
        p6_ref, b_ref, c_ref = ref.regress(xf_ref)
        r_ref = hmr_ref.rot6d_to_rotmat(p6_ref).view(4, 24, 3, 3)
    rot, betas, cam, xf, p6 = m(_t(x, gpu_device), return_features=True)
    scale = float(xf_ref.abs().max())
    assert float((xf.cpu() - xf_ref).abs().max()) < 2e-5 * scale
    np.testing.assert_allclose(p6.cpu().numpy(), p6_ref.numpy(), atol=TOL_F32)
    np.testing.assert_allclose(rot.cpu().numpy(), r_ref.numpy(), atol=TOL_F32)
    np.testing.assert_allclose(betas.cpu().numpy(), b_ref.numpy(), atol=TOL_F32)
    np.testing.assert_allclose(cam.cpu().numpy(), c_ref.numpy(), atol=TOL_F32)


def test_hmr_frames_are_independent(gpu_device, hmr_pair):
    """Frames shard across GPUs only if a frame's result does not depend on its batch: bit-identical."""
    m, _ = hmr_pair
    x = _t(synth.crops(6, seed=5), gpu_device)
    full = [t.cpu().numpy() for t in m(x)]
    a = [t.cpu().numpy() for t in m(x[:2])]
    b = [t.cpu().numpy() for t in m(x[2:])]
    for f, pa, pb in zip(full, a, b):
        np.testing.assert_array_equal(f, np.concatenate([pa, pb]))
    # every frame on its own (all-quarter launches, one-tile Winograd GEMMs) and in an odd-sized batch
    x9 = _t(synth.crops(9, seed=6), gpu_device)
    full9 = [t.cpu().numpy() for t in m(x9[:7])]
    for i in (0, 3, 6):
        one = [t.cpu().numpy() for t in m(x9[i:i + 1])]
        for f, o in zip(full9, one):
            np.testing.assert_array_equal(f[i:i + 1], o)


def test_hmr_bf16_frames_do_not_depend_on_which_kernels_their_batch_takes(gpu_device):
    """The bf16 encoder picks kernels by the batch (the evenly dealt convolution where it pays; layer3's plain blocks as one
    launch each with a frame per workgroup from 218 frames on a 256-CU device, their three ordinary launches below).  The
    same 217 frames in a batch of 217 and as the head of a batch of 230, and 40 of them on their own: bit-identical."""
    m = HMR(max_batch=230, precision="bf16").to(gpu_device)
    m.load_state_dict(synth.hmr_state_dict(seed=1))
    x = _t(synth.crops(230, seed=12), gpu_device)
    big = [t.cpu().numpy() for t in m(x)]
    mid = [t.cpu().numpy() for t in m(x[:217])]
    small = [t.cpu().numpy() for t in m(x[100:140])]
    for f, a, b in zip(big, mid, small):
        np.testing.assert_array_equal(f[:217], a)
        np.testing.assert_array_equal(f[100:140], b)
    assert np.abs(big[0]).max() > 0


def test_hmr_bf16_sub_batch_streams_equal_one_stream(gpu_device):
    """460 frames as one sub-batch (two rounds of frame-per-workgroup kernels in layer3), as two sub-batches of 230 on their own
    streams (each taking those kernels), and 300 of them as 2 x 150 (ordinary launches): bit-identical."""
    m = HMR(max_batch=460, precision="bf16").to(gpu_device)
    m.load_state_dict(synth.hmr_state_dict(seed=1))
    x = _t(synth.crops(460, seed=4), gpu_device)
    one = [t.cpu().numpy() for t in m(x)]
    m.set_streams(2)
    two = [t.cpu().numpy() for t in m(x)]
    head = [t.cpu().numpy() for t in m(x[:300])]
    for a, b, c in zip(one, two, head):
        np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(a[:300], c)


def test_hmr_capacity_and_empty(gpu_device, hmr_pair):
    m, _ = hmr_pair
    r, b, c = m(_t(synth.crops(1, seed=9), gpu_device))
    assert r.shape == (1, 24, 3, 3) and b.shape == (1, 10) and c.shape == (1, 3)
    R = r.cpu().numpy().reshape(-1, 3, 3)
    np.testing.assert_allclose(R @ R.transpose(0, 2, 1), np.broadcast_to(np.eye(3), R.shape), atol=1e-5)
    np.testing.assert_allclose(np.linalg.det(R), 1.0, atol=1e-5)


# ------------------------------------------------------------------------------------------------
# bf16 encoder (BASELINE config 3): bf16 MFMA, fp32 accumulate; compared with a bf16-rounding emulation
# ------------------------------------------------------------------------------------------------
BF16_CONV_CASES = [(2, 56, 64, 64, 64, 1, 1, 0), (2, 28, 128, 128, 128, 3, 1, 1), (2, 56, 128, 128, 128, 3, 2, 1),
                   (2, 56, 256, 256, 512, 1, 2, 0), (1, 224, 3, 8, 64, 7, 2, 3), (3, 7, 512, 512, 512, 3, 1, 1),
                   (1, 9, 64, 64, 64, 3, 1, 1)]


@pytest.mark.parametrize("case", BF16_CONV_CASES)
def test_conv_bf16_matches_emulation(gpu_device, case):
    B, H, Cr, Cin, Cout, k, s, p = case
    rng = np.random.default_rng(hash(case) % (2 ** 32))
    bf = lambda t: t.to(torch.bfloat16).float()
    x = bf(torch.from_numpy(rng.standard_normal((B, H, H, Cin)).astype(np.float32)))
    x[..., Cr:] = 0
    w = bf(torch.from_numpy((rng.standard_normal((Cout, Cr, k, k)) / np.sqrt(Cr * k * k)).astype(np.float32)))
    bias = rng.standard_normal(Cout).astype(np.float32)
    ref = torch.nn.functional.conv2d(x[..., :Cr].permute(0, 3, 1, 2), w, torch.from_numpy(bias), stride=s, padding=p)
    Ho = ref.shape[2]
    res = bf(torch.from_numpy(rng.standard_normal((B, Ho, Ho, Cout)).astype(np.float32)))
    ref = torch.relu(ref.permute(0, 2, 3, 1) + res)               # fp32, before the final bf16 rounding
    n_cfg = _lib.load().pr_conv_num_tile_cfgs()
    ran = 0
    for cfg in range(6, n_cfg):
        try:
            y, _ = ops.conv2d_nhwc(x.to(gpu_device), w.numpy(), bias, res.to(gpu_device), stride=s, pad=p, relu=True,
                                   tile_cfg=cfg, precision="bf16")
        except _lib.PoseRiskHipError as e:
            assert "not a multiple of tile N" in str(e) or "bad channels" in str(e)
            continue
        ran += 1
        got = y.float().cpu()
        assert y.dtype == torch.bfloat16
        # one bf16 ulp (2^-8 relative) around the fp32 value, plus fp32 accumulation-order slack
        tol = ref.abs() * 2.0 ** -8 + 1e-3
        assert bool(((got - ref).abs() <= tol).all()), f"cfg {cfg}: max err {(got - ref).abs().max()}"
    assert ran >= 1


def test_hmr_bf16_encoder(gpu_device):
    sd = synth.hmr_state_dict(seed=1)
    ref = hmr_ref.build(sd)
    m = HMR(max_batch=4, precision="bf16").to(gpu_device)
    m.load_state_dict(sd)
    x = synth.crops(3, seed=4)
    with torch.no_grad():
        xf_emu = hmr_ref.features_bf16(ref, torch.from_numpy(x))
        xf_f32 = ref.features(torch.from_numpy(x))
        p6, b_ref, c_ref = ref.regress(xf_f32)
        r_f32 = hmr_ref.rot6d_to_rotmat(p6).view(3, 24, 3, 3)
    rot, betas, cam, xf, _ = m(_t(x, gpu_device), return_features=True)
    scale = float(xf_f32.abs().max())
    # against the bf16-rounding emulation: only accumulation order (and the rounding flips it causes) differ
    err_emu = float((xf.cpu() - xf_emu).abs().max()) / scale
    # against the fp32 reference: the precision cost of the bf16 configuration (stated, not 1e-4)
    err_f32 = float((xf.cpu() - xf_f32).abs().max()) / scale
    assert err_emu < 2e-2, err_emu
    assert err_f32 < 5e-2, err_f32
    np.testing.assert_allclose(rot.cpu().numpy(), r_f32.numpy(), atol=5e-2)
    np.testing.assert_allclose(betas.cpu().numpy(), b_ref.numpy(), atol=5e-2)
    R = rot.cpu().numpy().reshape(-1, 3, 3)
    np.testing.assert_allclose(R @ R.transpose(0, 2, 1), np.broadcast_to(np.eye(3), R.shape), atol=1e-5)
    print(f"bf16 encoder: xf rel err vs emulation {err_emu:.2e}, vs fp32 {err_f32:.2e}, "
          f"rotmat max abs err vs fp32 {np.abs(rot.cpu().numpy() - r_f32.numpy()).max():.2e}")


def test_hmr_bf16_full_batch_properties(gpu_device):
    """BASELINE config 3 at its full size (B=256, bf16 encoder): size-independent properties -- rotations are
    orthonormal with det +1, the run is deterministic, and a frame does not depend on its batch."""
    m = HMR(max_batch=256, precision="bf16").to(gpu_device)
    m.load_state_dict(synth.hmr_state_dict(seed=1))
    x = torch.rand((256, 3, 224, 224), device=gpu_device, generator=torch.Generator(device=gpu_device).manual_seed(3))
    rot, betas, cam = [t.clone() for t in m(x)]
    rot2, betas2, cam2 = m(x)
    assert torch.equal(rot, rot2) and torch.equal(betas, betas2) and torch.equal(cam, cam2)
    R = rot.reshape(-1, 3, 3).double()
    eye = torch.eye(3, dtype=torch.float64, device=gpu_device)
    assert float((R @ R.transpose(1, 2) - eye).abs().max()) < 1e-5
    assert float((torch.linalg.det(R) - 1).abs().max()) < 1e-5
    sub = m(x[100:116])
    assert torch.equal(sub[0], rot[100:116]) and torch.equal(sub[1], betas[100:116]) and torch.equal(sub[2], cam[100:116])
    assert bool(torch.isfinite(rot).all() and torch.isfinite(betas).all() and torch.isfinite(cam).all())
    # every one of the 256 frames against the bf16-rounding emulation of the same network (oracle/hmr_ref.py)
    ref = hmr_ref.build(synth.hmr_state_dict(seed=1))
    xf = m(x, return_features=True)[3].cpu()
    with torch.no_grad():
        parts = [hmr_ref.features_bf16(ref, x[i:i + 16].cpu()) for i in range(0, 256, 16)]
    xf_emu = torch.cat(parts)
    err = float((xf - xf_emu).abs().max() / xf_emu.abs().max())
    measured("B=256 bf16: pooled features vs bf16 emulation (relative to max)", err, 2e-2)
    assert err < 2e-2, err


def test_rot6d(gpu_device):
    rng = np.random.default_rng(1)
    p = rng.standard_normal((7, 144)).astype(np.float32)
    ref = hmr_ref.rot6d_to_rotmat(torch.from_numpy(p)).view(7, 24, 3, 3).numpy()
    out = ops.rot6d_to_rotmat(_t(p, gpu_device)).cpu().numpy()
    np.testing.assert_allclose(out, ref, atol=2e-5)   # Gram-Schmidt on N(0,1) 6-D input cancels digits


# ------------------------------------------------------------------------------------------------
# rotmat -> axis-angle -> Euler
# ------------------------------------------------------------------------------------------------
def test_pose_to_euler_matches_golden(gpu_device):
    g = golden("euler.npz")
    aa, eul, st = ops.pose_to_euler(_t(g["rotmat"], gpu_device))
    np.testing.assert_allclose(aa.cpu().numpy(), g["axis_angle"], atol=1e-6)
    # Euler degrees are taken from the float32 axis-angle; where ours differs by one float32 ulp the
    # angle moves by ~1e-5 deg, so compare through our own axis-angle with the oracle's Euler stage.
    ours_aa = aa.cpu().numpy()
    ref = np.stack([coord_ref.axis_angle_to_euler_angle(f) for f in ours_aa])
    d = np.abs(eul.cpu().numpy() - ref)
    d = np.minimum(d, 360 - d)
    # device libm vs glibc differ in the last double ulp of sin/cos; where that flips the float32
    # rounding of a matrix entry (coord_utils.py:86 returns float32) the angle moves by ~1e-6 degrees
    assert d.max() < 1e-5 and np.mean(d < 1e-9) > 0.999, (d.max(), np.mean(d < 1e-9))
    # every frame: our float32 axis-angle is the reference's or its float32 neighbour (device libm vs glibc in the
    # last double ulp of acos / sqrt), and the Euler angles then move by at most that much
    # (ulp of the vector's largest component: a component near zero carries the absolute error of the others)
    big = np.abs(g["axis_angle"]).max(axis=2, keepdims=True).astype(np.float32)
    ulp = np.broadcast_to(np.spacing(np.maximum(big, np.float32(1e-30))).astype(np.float64), g["axis_angle"].shape)
    off = np.abs(ours_aa.astype(np.float64) - g["axis_angle"].astype(np.float64))
    assert (off <= ulp).all(), float((off / ulp).max())
    measured("pose_to_euler: axis-angle vs reference golden (float32 ulps)", (off / ulp).max(), 1.0, "ulp")
    dg = np.abs(eul.cpu().numpy() - g["euler_deg"])
    dg = np.minimum(dg, 360 - dg)
    measured("pose_to_euler: Euler degrees vs reference golden, all frames", dg.max(), 2e-5, "deg")
    assert dg.max() < 2e-5
    same = (ours_aa == g["axis_angle"]).all(axis=(1, 2))
    np.testing.assert_allclose(eul.cpu().numpy()[same], g["euler_deg"][same], atol=1e-9)
    assert int(st.abs().sum()) == 0


def test_pose_to_euler_flags_non_rotation(gpu_device):
    rot = synth.rotmats(3, seed=8)
    rot[1, 5] *= 1.5                      # not orthonormal -> OpenCV's SVD repairs it; flags stay clear
    rot[2, 3] = np.nan                    # NaN: Rodrigues returns zeros (range check)
    aa, eul, st = ops.pose_to_euler(_t(rot, gpu_device))
    ref = coord_ref.rot_to_angle(rot[1])
    np.testing.assert_allclose(aa.cpu().numpy()[1], ref, atol=1e-6)
    assert np.all(aa.cpu().numpy()[2, 3] == 0)


# ------------------------------------------------------------------------------------------------
# SMPL
# ------------------------------------------------------------------------------------------------
def _model(tag):
    if tag == "small":
        return synth.smpl_model(V=97, seed=2)
    if tag == "dense":
        return synth.smpl_model(V=64, seed=7, dense_weights=True, model_betas=np.linspace(-0.5, 0.5, 10))
    return synth.smpl_model(V=6890, seed=2)


def _smpl_layer(model, gpu_device, max_batch, tile, monkeypatch):
    """A handle with the register-tiled skinning kernel on or off (POSERISK_SMPL_TILE is read by pr_smpl_create)."""
    monkeypatch.setenv("POSERISK_SMPL_TILE", "1" if tile else "0")
    layer = SMPLLayer(model, device=gpu_device, max_batch=max_batch)
    layer._ensure()
    return layer


@pytest.mark.parametrize("variant", ["skin_tile", "skin_split_waves", "skin_rows"])
@pytest.mark.parametrize("tag", ["small", "dense"])
def test_smpl_matches_reference_golden(gpu_device, tag, variant, monkeypatch):
    """Against the reference's own SMPL_Layer.forward (tests/golden/smpl.npz).  max_batch <= 128 selects the
    register-tiled kernel for sparse weights (smpl_skin_tile) or, switched off or with dense weights, the wave-split
    kernel (smpl_skin<4>, smpl_skin<24>); larger handles the rows variant (smpl_skin_rows<...>): all are compared."""
    max_batch = 256 if variant == "skin_rows" else 32
    g = golden("smpl.npz")
    layer = _smpl_layer(_model(tag), gpu_device, max_batch, variant == "skin_tile", monkeypatch)
    worst = 0.0
    for B in (1, 4):
        for bt in ("zero", "rand"):
            v, j = layer(_t(g[f"{tag}_B{B}_{bt}_pose"], gpu_device), _t(g[f"{tag}_B{B}_{bt}_betas"], gpu_device))
            worst = max(worst, np.abs(v.cpu().numpy() - g[f"{tag}_B{B}_{bt}_verts"]).max(),
                        np.abs(j.cpu().numpy() - g[f"{tag}_B{B}_{bt}_joints"]).max())
            np.testing.assert_allclose(v.cpu().numpy(), g[f"{tag}_B{B}_{bt}_verts"], atol=1e-5)
            np.testing.assert_allclose(j.cpu().numpy(), g[f"{tag}_B{B}_{bt}_joints"], atol=1e-5)
    measured(f"smpl {tag} {variant}: verts/joints vs reference golden", worst, 1e-5, "m")


def test_smpl_tiled_skinning_has_the_wave_split_kernels_bits(gpu_device, monkeypatch):
    """smpl_skin_tile (252 rows x 16 frames per workgroup, LDS-staged coefficients) performs smpl_skin's arithmetic
    operation for operation: full-size model, batches that fill, straddle and underfill the 16-frame groups and the
    handle's chunk (64), with and without shape coefficients."""
    m = _model("full")
    tile = _smpl_layer(m, gpu_device, 64, True, monkeypatch)
    split = _smpl_layer(m, gpu_device, 64, False, monkeypatch)
    for B in (1, 16, 23, 64, 150):
        pose = _t(synth.poses(B, seed=40 + B), gpu_device)
        for betas in (_t(synth.betas(B, seed=41 + B), gpu_device), torch.zeros((B, 10), device=gpu_device)):
            vt, jt = tile(pose, betas)
            vs, js = split(pose, betas)
            assert torch.equal(vt, vs) and torch.equal(jt, js), B
    jc_t = tile.joint_cam(_t(synth.poses(40, seed=5), gpu_device))
    jc_s = split.joint_cam(_t(synth.poses(40, seed=5), gpu_device))
    assert torch.equal(jc_t, jc_s)


def test_smpl_rodrigues_edge_vectors_on_the_gpu(gpu_device):
    """tests/golden/rodrigues.npz holds the reference's batch_rodrigues on zero / 1e-9 / 1e-6 ... vectors (the
    `norm(v + 1e-8)` quirk); the CPU oracle is pinned on them.  Here they go through pr_smpl_forward:
    every joint of every frame carries one of the edge vectors, and mesh + joints must equal the oracle's."""
    g = golden("rodrigues.npz")
    vec = g["axisang"]                                   # [46,3]
    assert (np.linalg.norm(vec, axis=1) < 1e-5).sum() >= 4 and (np.linalg.norm(vec, axis=1) == 0).any()
    B = 8
    idx = (np.arange(B * 24).reshape(B, 24) * 7 + np.arange(B)[:, None]) % len(vec)
    pose = vec[idx].reshape(B, 72).astype(np.float32)
    pose[3] = np.tile(vec[np.argmin(np.linalg.norm(vec, axis=1))], 24)        # a frame of all-zero vectors
    tiny = vec[np.linalg.norm(vec, axis=1) < 1e-5]
    pose[5] = np.tile(tiny, (24 // len(tiny) + 1, 1))[:24].reshape(72)        # a frame of near-zero vectors only
    betas = synth.betas(B, seed=21)
    m = _model("full")
    om = smpl_ref.SMPLModel(**{k: m[k] for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "weights")})
    vr, jr = smpl_ref.smpl_forward(om, pose, betas)
    for mb in (32, 256):
        layer = SMPLLayer(m, device=gpu_device, max_batch=mb)
        v, j = layer(_t(pose, gpu_device), _t(betas, gpu_device))
        ev, ej = np.abs(v.cpu().numpy() - vr).max(), np.abs(j.cpu().numpy() - jr).max()
        measured(f"smpl edge vectors max_batch={mb}: verts / joints vs oracle", max(ev, ej), 1e-5, "m")
        assert ev < 1e-5 and ej < 1e-5 and np.isfinite(v.cpu().numpy()).all()


def test_smpl_translation_and_cpu_tensors(gpu_device):
    g = golden("smpl.npz")
    layer = SMPLLayer(_model("small"), device=gpu_device)
    v, j = layer(torch.from_numpy(g["trans_pose"]), torch.from_numpy(g["trans_betas"]), torch.from_numpy(g["trans_trans"]))
    assert v.device.type == "cpu"
    np.testing.assert_allclose(v.numpy(), g["trans_verts"], atol=1e-5)
    np.testing.assert_allclose(j.numpy(), g["trans_joints"], atol=1e-5)
    # the reference's default placeholders: betas = zeros(1), trans = zeros(1)
    v2, j2 = layer(torch.from_numpy(g["small_B4_zero_pose"]), torch.zeros(1), torch.zeros(1))
    np.testing.assert_allclose(v2.numpy(), g["small_B4_zero_verts"], atol=1e-5)


def test_smpl_full_size_and_batching(gpu_device):
    g = golden("smpl.npz")
    layer = SMPLLayer(_model("full"), device=gpu_device, max_batch=32)
    v, j = layer(_t(g["full_pose"], gpu_device), _t(g["full_betas"], gpu_device))
    np.testing.assert_allclose(v.cpu().numpy()[:, ::53], g["full_verts_stride53"], atol=1e-5)
    np.testing.assert_allclose(j.cpu().numpy(), g["full_joints"], atol=1e-5)
    # B = 70 crosses the handle's chunk size (32) and the 16-frame wave groups; frames independent
    pose = _t(synth.poses(70, seed=12), gpu_device)
    betas = _t(synth.betas(70, seed=13), gpu_device)
    vb, jb = layer(pose, betas)
    v1, j1 = layer(pose[33:34], betas[33:34])
    np.testing.assert_array_equal(vb[33:34].cpu().numpy(), v1.cpu().numpy())
    np.testing.assert_array_equal(jb[33:34].cpu().numpy(), j1.cpu().numpy())
    om = smpl_ref.SMPLModel(**{k: _model("full")[k] for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "weights")})
    vr, jr = smpl_ref.smpl_forward(om, pose.cpu().numpy()[60:], betas.cpu().numpy()[60:])
    np.testing.assert_allclose(vb[60:].cpu().numpy(), vr, atol=1e-5)
    np.testing.assert_allclose(jb[60:].cpu().numpy(), jr, atol=1e-5)


def test_smpl_rest_pose_is_shape_blend(gpu_device):
    """Size-independent property: zero pose -> verts = v_template + shapedirs.beta (rotations = I)."""
    m = _model("full")
    layer = SMPLLayer(m, device=gpu_device)
    B = 64
    betas = synth.betas(B, seed=3)
    v, j = layer(torch.zeros((B, 72), device=gpu_device), _t(betas, gpu_device))
    expect = m["v_template"][None] + np.einsum("vcl,bl->bvc", m["shapedirs"], betas)
    np.testing.assert_allclose(v.cpu().numpy(), expect, atol=2e-6)
    np.testing.assert_allclose(j.cpu().numpy(), np.einsum("jv,bvc->bjc", m["J_regressor"], expect), atol=2e-6)


@pytest.mark.parametrize("tag", ["small", "full"])
def test_joint_cam_matches_reference_golden(gpu_device, tag):
    g = golden("joint_cam.npz")
    layer = SMPLLayer(_model(tag), device=gpu_device)
    aa = _t(g[f"{tag}_axis_angle_in"], gpu_device)
    jc = layer.joint_cam(aa)
    np.testing.assert_allclose(jc.cpu().numpy(), g[f"{tag}_joint_cam"], atol=1e-2)  # millimetres
    np.testing.assert_array_equal(aa.cpu().numpy(), g[f"{tag}_axis_angle_after"])   # Q5 in-place root overwrite


# ------------------------------------------------------------------------------------------------
# REBA / RULA: exact
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["example", "default", "loaded"])
def test_scores_match_reference_exactly(gpu_device, name):
    g = golden("scores.npz")
    infos = json.loads(str(g["infos_json"]))
    pose = _t(g["pose"], gpu_device)
    np.testing.assert_array_equal(ops.reba(pose, infos[name]["REBA"]).cpu().numpy(), g[f"reba_{name}"])
    np.testing.assert_array_equal(ops.rula(pose, infos[name]["RULA"]).cpu().numpy(), g[f"rula_{name}"])


def test_scores_large_batch_against_oracle(gpu_device):
    rng = np.random.default_rng(11)
    pose = rng.uniform(-180, 180, (20000, 24, 3))
    info = synth.EXAMPLE_INFO
    np.testing.assert_array_equal(ops.reba(_t(pose, gpu_device), info["REBA"]).cpu().numpy(),
                                  reba_ref.reba_packed(pose, info["REBA"]))
    np.testing.assert_array_equal(ops.rula(_t(pose, gpu_device), info["RULA"]).cpu().numpy(),
                                  rula_ref.rula_packed(pose, info["RULA"]))


# ------------------------------------------------------------------------------------------------
# whole hot path: crops -> scores (BASELINE config 1 plumbing, batched)
# ------------------------------------------------------------------------------------------------
_THRESHOLDS = np.array([0, 1, 5, 10, 15, 20, 30, 45, 60, 70, 90, 100, 110], np.float64)


def _knife_edge(euler, eps):
    d = np.abs(np.abs(euler)[..., None] - _THRESHOLDS).min(axis=-1)
    return (d < eps).any(axis=(1, 2))


def test_pipeline_matches_oracle(gpu_device, hmr_pair):
    m, ref = hmr_pair
    sm = synth.smpl_model(V=6890, seed=2)
    layer = SMPLLayer(sm, device=gpu_device)
    info = synth.EXAMPLE_INFO
    x = synth.crops(6, seed=21)
    want = pipeline_ref.run(ref, smpl_ref.SMPLModel(**{k: sm[k] for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "weights")}),
                            x, info, batch_size=8)
    pipe = FramePipeline(m, layer, info, with_verts=True)
    got = {k: v.cpu().numpy() for k, v in pipe(_t(x, gpu_device)).items()}
    np.testing.assert_allclose(got["rotmat"], want["rotmat"], atol=TOL_F32)
    np.testing.assert_allclose(got["betas"], want["betas"], atol=TOL_F32)
    np.testing.assert_allclose(got["cam"], want["cam"], atol=TOL_F32)
    np.testing.assert_allclose(got["axis_angle"], want["axis_angle"], atol=TOL_F32)
    d = np.abs(got["euler"] - want["euler"])
    assert np.minimum(d, 360 - d).max() < 2e-2                      # degrees, from 1e-4 rotmat agreement
    np.testing.assert_allclose(got["joint_cam"], want["joint_cam"], atol=TOL_MM)   # millimetres: 1e-4 m
    assert int(np.abs(got["status"]).sum()) == 0
    safe = ~_knife_edge(want["euler"], 5e-2)
    np.testing.assert_array_equal(got["reba"][safe], want["reba"][safe])
    np.testing.assert_array_equal(got["rula"][safe], want["rula"][safe])
    # scores from OUR Euler angles through the oracle scorer: exact for every frame
    np.testing.assert_array_equal(got["reba"], reba_ref.reba_packed(got["euler"], info["REBA"]))
    np.testing.assert_array_equal(got["rula"], rula_ref.rula_packed(got["euler"], info["RULA"]))
    assert np.all(got["axis_angle"][:, 0] == np.array([3.14, 0, 0], np.float32))


def _compare_with_oracle(tag, got, want, info, rot_tol=TOL_F32):
    """All frames of a batch against the CPU oracle, north-star tolerances, measured maxima reported."""
    e = {k: np.abs(got[k].astype(np.float64) - want[k].astype(np.float64)).max() for k in ("rotmat", "betas", "cam", "axis_angle", "joint_cam")}
    d = np.abs(got["euler"] - want["euler"])
    e["euler"] = np.minimum(d, 360 - d).max()
    for k, tol, unit in (("rotmat", rot_tol, ""), ("betas", rot_tol, ""), ("cam", rot_tol, ""), ("axis_angle", rot_tol, "rad"),
                         ("joint_cam", TOL_MM * rot_tol / TOL_F32, "mm"), ("euler", 2e-2 * rot_tol / TOL_F32, "deg")):
        measured(f"{tag}: {k} vs oracle", e[k], tol, unit)
        assert e[k] < tol, (tag, k, e[k])
    assert int(np.abs(got["status"]).sum()) == 0
    # a frame can only score differently if one of its angles lies within the (measured, asserted above) Euler error
    # of a threshold
    safe = ~_knife_edge(want["euler"], 2 * e["euler"] + 1e-9)
    assert safe.mean() > 0.8, safe.mean()
    np.testing.assert_array_equal(got["reba"][safe], want["reba"][safe])
    np.testing.assert_array_equal(got["rula"][safe], want["rula"][safe])
    np.testing.assert_array_equal(got["reba"], reba_ref.reba_packed(got["euler"], info["REBA"]))
    np.testing.assert_array_equal(got["rula"], rula_ref.rula_packed(got["euler"], info["RULA"]))


def _oracle_smpl(sm):
    return smpl_ref.SMPLModel(**{k: sm[k] for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "weights")})


def test_pipeline_full_batch_against_oracle(gpu_device, hmr_pair):
    """BASELINE configs[1] at its full size (B=64 fp32): every frame against the CPU oracle pipeline at the north
    star's tolerance, then the size-independent properties (orthonormality, permutation, determinism)."""
    _, ref = hmr_pair
    sd = synth.hmr_state_dict(seed=1)
    m = HMR(max_batch=64).to(gpu_device)
    m.load_state_dict(sd)
    sm = synth.smpl_model(V=6890, seed=2)
    layer = SMPLLayer(sm, device=gpu_device)
    info = synth.EXAMPLE_INFO
    pipe = FramePipeline(m, layer, info, with_verts=True)
    xh = synth.crops(64, seed=33)
    x = _t(xh, gpu_device)
    out = {k: v.clone() for k, v in pipe(x).items()}
    want = pipeline_ref.run(ref, _oracle_smpl(sm), xh, info, batch_size=8)
    _compare_with_oracle("B=64 fp32", {k: v.cpu().numpy() for k, v in out.items()}, want, info)
    R = out["rotmat"].cpu().numpy().reshape(-1, 3, 3).astype(np.float64)
    np.testing.assert_allclose(R @ R.transpose(0, 2, 1), np.broadcast_to(np.eye(3), R.shape), atol=1e-5)
    np.testing.assert_allclose(np.linalg.det(R), 1.0, atol=1e-5)
    assert np.isfinite(out["verts"].cpu().numpy()).all()
    assert (out["reba"][:, 0] >= 1).all() and (out["reba"][:, 0] <= 15).all()
    assert (out["rula"][:, 0] >= 1).all() and (out["rula"][:, 0] <= 7).all()
    # permuting the frames permutes the results bit-for-bit (frames are independent -> shardable)
    perm = torch.randperm(64, generator=torch.Generator().manual_seed(0)).to(gpu_device)
    out2 = pipe(x[perm])
    for k in ("rotmat", "betas", "cam", "euler", "joint_cam", "reba", "rula"):
        assert torch.equal(out2[k], out[k][perm]), k
    # determinism
    out3 = pipe(x[perm])
    for k in ("rotmat", "verts", "euler"):
        assert torch.equal(out3[k], out2[k]), k


def test_frames_forward_and_predictor_against_the_references_driver_loop(gpu_device):
    """a15 / a14 against fixtures the REFERENCE's own functions produced (tests/golden/driver_loop.npz: Predictor.
    get_pose_estimation_results and .post_processing of lib/core/base.py called unbound in the build container, with the
    oracle's encoder standing in for SPIN): pr_frames_forward through FramePipeline, and the drop-in Predictor's loop over
    the same batches of 8 + 6, give the loop's four outputs -- dtypes, frame order, the root rows overwritten (Q5), images --
    within the north star's tolerance, and the drop-in's post_processing gives the reference's five numbers."""
    import types
    from poserisk_release_amd import dropin
    dropin.install()
    from core import base
    from smpl import SMPL
    g = golden("driver_loop.npz")
    n, bs = int(g["n_frames"]), int(g["batch_size"])
    crop_seed, weight_seed, smpl_seed = (int(v) for v in g["seeds"])
    sd = synth.hmr_state_dict(seed=weight_seed)
    sm = synth.smpl_model(V=6890, seed=smpl_seed)
    crops = synth.crops(n, seed=crop_seed)
    m = HMR(max_batch=bs).to(gpu_device)
    m.load_state_dict(sd)
    info = synth.EXAMPLE_INFO

    def check(tag, euler, joint_cam, axis_angle):
        assert euler.dtype == np.float64 and joint_cam.dtype == np.float32 and axis_angle.dtype == np.float32
        d = np.abs(euler - g["result"])
        e_eul = np.minimum(d, 360 - d).max()
        e_jc = np.abs(joint_cam - g["joint_cam"]).max()
        e_aa = np.abs(axis_angle - g["debug_result"]).max()
        measured(f"{tag}: euler vs the reference's loop", e_eul, 2e-2, "deg")
        measured(f"{tag}: joint_cam vs the reference's loop", e_jc, TOL_MM, "mm")
        measured(f"{tag}: axis-angle vs the reference's loop", e_aa, TOL_F32, "rad")
        assert e_eul < 2e-2 and e_jc < TOL_MM and e_aa < TOL_F32
        assert np.all(axis_angle[:, 0] == np.array([3.14, 0, 0], np.float32))
        return e_eul

    # (1) the C ABI's per-batch driver, the batches the reference's loader yields
    pipe = FramePipeline(m, SMPLLayer(sm, device=gpu_device), info, with_verts=False)
    parts = []
    for i in range(0, n, bs):
        parts.append({k: v.cpu().numpy() for k, v in pipe(_t(crops[i:i + bs], gpu_device)).items()})
    got = {k: np.concatenate([p[k] for p in parts]) for k in ("euler", "joint_cam", "axis_angle", "reba", "rula", "status")}
    e_eul = check("pr_frames_forward", got["euler"], got["joint_cam"], got["axis_angle"])
    assert int(np.abs(got["status"]).sum()) == 0
    safe = ~_knife_edge(g["result"], 2 * e_eul + 1e-9)
    assert safe.mean() > 0.8
    np.testing.assert_array_equal(got["reba"][safe, 0], g["reba_scores"][safe])
    np.testing.assert_array_equal(got["rula"][safe, 0], g["rula_scores"][safe])
    # (2) the plugin surface: Predictor.get_pose_estimation_results(loader) as base.py:126 calls it, then the scorers and
    # post_processing as base.py:151-154 do
    from models import hmr as dropin_hmr
    model = dropin_hmr()
    model.load_state_dict(sd, strict=False)
    args = types.SimpleNamespace(gpu="0", type="REBA,RULA", debug=False, debug_joints="", debug_frame=-1)
    pred = base.Predictor(args, spin_model=model, smpl_model=SMPL(models={"neutral": sm}, device=gpu_device), batch_size=bs)
    loader = [torch.tensor(crops[i:i + bs]) for i in range(0, n, bs)]
    result, joint_cam, images, debug_result = pred.get_pose_estimation_results(loader)
    check("dropin.Predictor", result, joint_cam, debug_result)
    assert images.dtype == np.float32 and np.array_equal(images, crops) and bool(g["images_equal_crops"])
    np.testing.assert_array_equal(result, got["euler"])            # the same kernels behind both surfaces
    for title, scorer in (("reba", pred.reba), ("rula", pred.rula)):
        final, scores_log, logs = pred.post_processing(scorer(result, joint_cam, info), scorer.eval_items, None, None, title)
        np.testing.assert_array_equal(scores_log[safe], g[f"{title}_scores"][safe])
        if safe.all():
            np.testing.assert_array_equal(np.array(final, np.float64), g[f"{title}_final"])
            assert np.asarray(logs).tolist() == json.loads(str(g[f"{title}_logs_json"]))
    # the aggregation alone, on the reference's own score vectors (Q20: NaN top-10 % below ten frames)
    for n_agg in (5, 10, 101):
        res = [dict(score=v, log_score=[0]) for v in g[f"agg{n_agg}_scores"]]
        final, scores_log, _ = pred.post_processing(res)
        np.testing.assert_array_equal(np.array(final, np.float64), g[f"agg{n_agg}_final"])
        np.testing.assert_array_equal(scores_log, g[f"agg{n_agg}_scores"])
        assert [type(v).__name__ for v in final] == json.loads(str(g[f"agg{n_agg}_final_types_json"]))


def test_pipeline_config4_slice_against_oracle(gpu_device, hmr_pair):
    """configs[3]'s per-GPU slice (256 frames per GPU, fp32, two batches in flight): every frame against the oracle."""
    _, ref = hmr_pair
    sd = synth.hmr_state_dict(seed=1)
    m = HMR(max_batch=256).to(gpu_device)
    m.load_state_dict(sd)
    sm = synth.smpl_model(V=6890, seed=2)
    layer = SMPLLayer(sm, device=gpu_device, max_batch=256)
    info = synth.EXAMPLE_INFO
    pipe = FramePipeline(m, layer, info, with_verts=False, lanes=2)
    xh = synth.crops(256, seed=34)
    x = _t(xh, gpu_device)
    first = pipe(x)
    second = pipe(x.flip(0))                    # the other lane, the frames in reverse order
    FramePipeline.wait(first); FramePipeline.wait(second)
    got = {k: v.cpu().numpy() for k, v in first.items()}
    rev = {k: v.cpu().numpy() for k, v in second.items()}
    for k in ("rotmat", "euler", "joint_cam", "reba", "rula"):
        np.testing.assert_array_equal(rev[k][::-1], got[k])
    want = pipeline_ref.run(ref, _oracle_smpl(sm), xh, info, batch_size=8)
    _compare_with_oracle("B=256 fp32 lanes=2", got, want, info)


def test_pipeline_config3_bf16_against_oracle(gpu_device, hmr_pair):
    """BASELINE configs[2] as a pipeline (B=256, bf16 encoder -> fp32 regressor -> fp32 SMPL -> REBA/RULA, two batches
    in flight): every frame against the CPU oracle pipeline fed the bf16-rounding emulation of the encoder
    (oracle/hmr_ref.py::features_bf16).  Stated tolerance of this configuration: 5e-2 on rotation matrices,
    hence 5e-2 rad on axis-angle, 3 degrees on Euler angles away from gimbal lock and 5e-2 x 2 m = 100 mm on joint_cam.
    Scores are the oracle scorer's on OUR Euler angles for every frame, and the REBA / RULA agreement of this
    configuration with the fp32 one (what a user of the bf16 encoder gives up) is measured and reported."""
    _, ref = hmr_pair
    sd = synth.hmr_state_dict(seed=1)
    sm = synth.smpl_model(V=6890, seed=2)
    info = synth.EXAMPLE_INFO
    layer = SMPLLayer(sm, device=gpu_device, max_batch=256)
    m = HMR(max_batch=256, precision="bf16").to(gpu_device)
    m.load_state_dict(sd)
    pipe = FramePipeline(m, layer, info, with_verts=False, lanes=2)
    xh = synth.crops(256, seed=35)
    x = _t(xh, gpu_device)
    first = pipe(x)
    second = pipe(x.flip(0))                    # the other lane, the frames in reverse order
    FramePipeline.wait(first); FramePipeline.wait(second)
    got = {k: v.cpu().numpy() for k, v in first.items()}
    rev = {k: v.cpu().numpy() for k, v in second.items()}
    for k in ("rotmat", "euler", "joint_cam", "reba", "rula"):
        np.testing.assert_array_equal(rev[k][::-1], got[k])

    class Emulated:                              # the oracle's HMR with the bf16 emulation as its encoder
        def __call__(self, crops):
            p6, b, c = ref.regress(hmr_ref.features_bf16(ref, crops))
            return hmr_ref.rot6d_to_rotmat(p6).view(-1, 24, 3, 3), b, c
    want = pipeline_ref.run(Emulated(), _oracle_smpl(sm), xh, info, batch_size=16)
    e = {k: np.abs(got[k].astype(np.float64) - want[k].astype(np.float64)).max() for k in ("rotmat", "betas", "cam", "axis_angle", "joint_cam")}
    d = np.abs(got["euler"] - want["euler"])
    d = np.minimum(d, 360 - d)
    for k, tol, unit in (("rotmat", 5e-2, ""), ("betas", 5e-2, ""), ("cam", 5e-2, ""), ("axis_angle", 5e-2, "rad"), ("joint_cam", 100.0, "mm")):
        measured(f"B=256 bf16 pipeline: {k} vs oracle with the bf16-emulated encoder", e[k], tol, unit)
        assert e[k] < tol, (k, e[k])
    measured("B=256 bf16 pipeline: euler vs oracle (99th percentile)", float(np.percentile(d, 99)), 3.0, "deg")
    measured("B=256 bf16 pipeline: euler vs oracle (max, incl. near gimbal lock)", float(d.max()), None, "deg")
    assert np.percentile(d, 99) < 3.0
    assert int(np.abs(got["status"]).sum()) == 0
    np.testing.assert_array_equal(got["reba"], reba_ref.reba_packed(got["euler"], info["REBA"]))
    np.testing.assert_array_equal(got["rula"], rula_ref.rula_packed(got["euler"], info["RULA"]))
    # what the bf16 encoder costs a user: the same crops through the fp32 configuration
    m32 = HMR(max_batch=256).to(gpu_device)
    m32.load_state_dict(sd)
    f32 = {k: v.cpu().numpy() for k, v in FramePipeline(m32, layer, info, with_verts=False)(x).items()}
    dr = np.abs(got["rotmat"] - f32["rotmat"]).max()
    de = np.abs(got["euler"] - f32["euler"]); de = np.minimum(de, 360 - de)
    agree_reba = float((got["reba"][:, 0] == f32["reba"][:, 0]).mean())
    agree_rula = float((got["rula"][:, 0] == f32["rula"][:, 0]).mean())
    within1 = float(((np.abs(got["reba"][:, 0] - f32["reba"][:, 0]) <= 1) & (np.abs(got["rula"][:, 0] - f32["rula"][:, 0]) <= 1)).mean())
    measured("bf16 vs fp32 configuration: rotmat", float(dr), 5e-2)
    measured("bf16 vs fp32 configuration: euler (median)", float(np.median(de)), None, "deg")
    measured("bf16 vs fp32 configuration: euler (99th percentile)", float(np.percentile(de, 99)), None, "deg")
    measured("bf16 vs fp32 configuration: frames with the same REBA score", agree_reba, None)
    measured("bf16 vs fp32 configuration: frames with the same RULA score", agree_rula, None)
    measured("bf16 vs fp32 configuration: frames with both scores within one point", within1, None)
    assert dr < 5e-2 and agree_reba > 0.5 and agree_rula > 0.5


def test_hmr_conv_forms_side_by_side(gpu_device, hmr_pair):
    """The forms of the ten 3x3 layers as handles in ONE process (pr_hmr_create's conv_form): each within the fp32
    tolerance of the oracle."""
    _, ref = hmr_pair
    sd = synth.hmr_state_dict(seed=1)
    x = synth.crops(8, seed=0)
    with torch.no_grad():
        xf = ref.features(torch.from_numpy(x))
        p6, b, c = ref.regress(xf)
        r = hmr_ref.rot6d_to_rotmat(p6).view(8, 24, 3, 3)
    outs = {}
    for form in ("direct", "winograd2", "winograd4", "winograd244", "winograd5"):
        m = HMR(max_batch=8, conv_form=form).to(gpu_device)
        m.load_state_dict(sd)
        rot, betas, cam, xfg, _ = m(_t(x, gpu_device), return_features=True)
        outs[form] = rot.clone()
        err = dict(xf=float((xfg.cpu() - xf).abs().max() / xf.abs().max()), rotmat=float((rot.cpu() - r).abs().max()),
                   betas=float((betas.cpu() - b).abs().max()), cam=float((cam.cpu() - c).abs().max()))
        for k, v in err.items():
            measured(f"hmr {form}: {k} vs oracle", v, TOL_F32)
        assert err["rotmat"] < TOL_F32 and err["betas"] < TOL_F32 and err["cam"] < TOL_F32 and err["xf"] < 2e-5, (form, err)
    assert not torch.equal(outs["direct"], outs["winograd4"])     # different rounding patterns: really different forms
    dflt = HMR(max_batch=8).to(gpu_device)
    dflt.load_state_dict(sd)
    assert torch.equal(dflt(_t(x, gpu_device))[0], outs["winograd5"])         # the default form: F(4x4) on the improved points
    assert not torch.equal(outs["winograd5"], outs["winograd4"])              # other points, other rounding pattern
    assert not torch.equal(outs["winograd244"], outs["winograd4"])            # the per-stage digits really select


def test_hmr_winograd_under_wide_dynamic_range(gpu_device):
    """F(4x4,3x3) in fp32 loses accuracy as the dynamic range of weights and activations grows, and the He-normal
    synthetic weights are benign.  Stress (tests/stress_weights.py): heavy-tailed filters, BatchNorm statistics
    calibrated on data with variances over ~8 decades, gamma 0.1..10, offset sparse activations, a 30x more sensitive
    decoder.  Every conv form against an fp64 run of the same network and against the fp32 oracle (torch-CPU: what the
    reference computes), 64 frames x 24 joints.

    The decision is made on DISTRIBUTIONS, not on a maximum (scripts/exp_wino_stats.py -> profiles/r04_wino_stats.txt: 256
    frames x 3 weight seeds): with this random, high-gain decoder ~13 % of the joints have a nearly degenerate 6-D vector,
    rot6d_to_rotmat amplifies ANY fp32 difference 10-100x there, and the maximum of |rotmat error| over 166 k samples is a
    lottery (direct 4.8e-4, the fp32 oracle itself 1.1e-3, round 3's default 244 9.3e-4): it says nothing about a form.
    What does: rms and 99th percentile of the 6-D pose (the regressor's output, before the amplification), of the
    rotation matrices (all joints) and the pooled features' relative rms.  Asserted here:
      * every form: pose / shape / camera and the rotation matrices of well-conditioned joints within 1e-4 of fp64 and of
        the fp32 oracle (absolute, the north star's tolerance);
      * the built-in default (5 = F(4x4,3x3) on the points 0, +-11/16, +-3/2): pose6d rms and p99, the rotation matrices' p99
        on the well-conditioned joints and the features' rms within 1.10x the direct form's (measured over 768 frames, all
        joints: 1.02 / 1.02 / 1.05 / 1.03), the all-joints p99 within 1.25x (64 frames are few for that tail);
      * Lavin & Gray's points (form 4) are measurably worse than that (pose6d rms 1.19x over 768 frames) -- recorded, and
        the reason form 5 exists."""
    from stress_weights import trained_like_state_dict
    sd = trained_like_state_dict()
    var = np.concatenate([v.reshape(-1) for k, v in sd.items() if k.endswith("running_var")])
    assert var.max() / var.min() > 1e6                      # the premise: a really wide per-channel range
    ref64 = hmr_ref.build(sd).double()
    ref32 = hmr_ref.build(sd)
    n = 64
    x = synth.crops(n, seed=3)
    with torch.no_grad():
        xf = ref64.features(torch.from_numpy(x).double())
        p6, b, c = ref64.regress(xf)
        r = hmr_ref.rot6d_to_rotmat(p6).view(n, 24, 3, 3)
        p6f, _, _ = ref32.regress(ref32.features(torch.from_numpy(x)))
        r32 = hmr_ref.rot6d_to_rotmat(p6f).view(n, 24, 3, 3).double()
    assert torch.isfinite(xf).all() and 0.05 < float(xf.mean()) < 50
    v = p6.view(n * 24, 3, 2)
    a1, a2 = v[:, :, 0], v[:, :, 1]
    b1 = a1 / a1.norm(dim=1, keepdim=True)
    u2 = a2 - (b1 * a2).sum(1, keepdim=True) * b1
    cond = torch.minimum(a1.norm(dim=1), u2.norm(dim=1)).view(n, 24)       # small = ill-conditioned normalisations
    well = cond > 0.5
    assert 0.3 < float(well.float().mean()) < 1.0
    measured("hmr trained-like weights: fraction of joints dropped by the conditioning mask", float(1 - well.float().mean()), None)
    measured("hmr trained-like weights: fp32 oracle vs fp64, pose6d rms", float((p6f.double() - p6).pow(2).mean().sqrt()), None)

    def dist(e):
        e = e.abs().flatten()
        return dict(rms=float(e.pow(2).mean().sqrt()), p99=float(torch.quantile(e, 0.99)), max=float(e.max()))

    err = {}
    for form in ("direct", "winograd2", "winograd244", "winograd4", "winograd5", "default"):
        m = HMR(max_batch=n, conv_form=form).to(gpu_device)
        m.load_state_dict(sd)
        rot, betas, cam, xfg, p6g = m(_t(x, gpu_device), return_features=True)
        dp = p6g.cpu().double() - p6
        dr = rot.cpu().double() - r
        err[form] = dict(xf=float((xfg.cpu().double() - xf).abs().max() / xf.abs().max()),
                         xf_rms=float((xfg.cpu().double() - xf).pow(2).mean().sqrt() / xf.pow(2).mean().sqrt()),
                         pose6d=float(dp.abs().max()), pose6d_rms=dist(dp)["rms"], pose6d_p99=dist(dp)["p99"],
                         betas=float((betas.cpu().double() - b).abs().max()), cam=float((cam.cpu().double() - c).abs().max()),
                         rotmat_well=float(dr[well].abs().max()), rotmat_well_p99=dist(dr[well])["p99"],
                         rotmat_all_rms=dist(dr)["rms"], rotmat_all_p99=dist(dr)["p99"], rotmat_all_max=dist(dr)["max"],
                         rotmat_all_p99_vs_fp32=dist(rot.cpu().double() - r32)["p99"],
                         pose6d_vs_fp32=float((p6g.cpu().double() - p6f.double()).abs().max()))
        for k, val in err[form].items():
            measured(f"hmr trained-like weights, {form}: {k}" + ("" if k.endswith("fp32") else " vs fp64"), val,
                     TOL_F32 if k in ("pose6d", "betas", "cam", "rotmat_well", "pose6d_vs_fp32") else None)
    for form, e in err.items():
        assert max(e[k] for k in ("pose6d", "betas", "cam", "rotmat_well", "pose6d_vs_fp32")) < TOL_F32, (form, e)
        # no form is materially worse than the direct one on the regressor's outputs
        assert e["pose6d_rms"] < 1.3 * err["direct"]["pose6d_rms"] and e["pose6d"] < 2 * err["direct"]["pose6d"], (form, e)
    # the built-in default IS form 5, and its error DISTRIBUTION is the direct form's (the bound: 1.10x on 64 frames)
    assert err["default"] == err["winograd5"]
    for k in ("pose6d_rms", "pose6d_p99", "rotmat_well_p99", "rotmat_all_p99_vs_fp32", "xf_rms"):
        assert err["default"][k] <= 1.10 * err["direct"][k], (k, err["default"][k], err["direct"][k])
    # all joints against fp64: the 99th percentile sits in the tail the ~13 % ill-conditioned joints make, and 64 frames are
    # few for it (768 frames: 1.05x; this set: 1.13x) -- a looser bound here, the distribution itself in profiles/r04_wino_stats.txt
    assert err["default"]["rotmat_all_p99"] <= 1.25 * err["direct"]["rotmat_all_p99"]
    # Lavin & Gray's points are worse than the improved ones on the same layers (why form 5 exists)
    assert err["winograd4"]["xf_rms"] > 1.05 * err["winograd5"]["xf_rms"]


# ------------------------------------------------------------------------------------------------
# crop front-end: bit-exact against the restated OpenCV fixed-point warp
# ------------------------------------------------------------------------------------------------
def test_crop_frames_bit_exact(gpu_device):
    from oracle import crop_ref
    rng = np.random.default_rng(5)
    frames = rng.integers(0, 256, (3, 450, 800, 3), dtype=np.uint8)
    bboxes = np.array([[400.3, 220.7, 150.2, 310.9], [5.0, 5.0, 100.0, 100.0], [790.0, 440.0, 60.5, 200.25],
                       [300.0, 200.0, 224 / 1.2, 224 / 1.2], [400.0, 225.0, 1200.0, 900.0], [123.456, 78.9, 33.3, 44.4]],
                      np.float32)
    idx = np.array([0, 1, 2, 0, 1, 2], np.int32)
    got = ops.crop_frames(_t(frames, gpu_device), bboxes, idx, scale=1.2).cpu().numpy()
    for n in range(len(bboxes)):
        want = crop_ref.crop_to_tensor(frames[idx[n]], bboxes[n], 1.2)
        np.testing.assert_array_equal(got[n], want, err_msg=f"crop {n}")
    # BGR input (cv2.imread order) gives the same crop as the RGB frame
    got_bgr = ops.crop_frames(_t(frames[..., ::-1].copy(), gpu_device), bboxes, idx, bgr=True).cpu().numpy()
    np.testing.assert_array_equal(got_bgr, got)
    # no index: crop n from frame n
    got2 = ops.crop_frames(_t(frames, gpu_device), bboxes[:3]).cpu().numpy()
    np.testing.assert_array_equal(got2[1], got[1])
    assert got.min() >= 0.0 and got.max() <= 1.0


def test_crop_frames_random_boxes_bit_exact(gpu_device):
    """120 seeded boxes over odd-sized frames: centred anywhere from well outside the frame to inside it, from 2 pixels
    (36x magnification) to several frame sizes (heavy minification), extreme aspect ratios -- every crop bit for bit."""
    from oracle import crop_ref
    rng = np.random.default_rng(77)
    F, H, W = 4, 241, 317
    frames = rng.integers(0, 256, (F, H, W, 3), dtype=np.uint8)
    n = 120
    cx = rng.uniform(-0.4 * W, 1.4 * W, n); cy = rng.uniform(-0.4 * H, 1.4 * H, n)
    bw = np.exp(rng.uniform(np.log(2.0), np.log(3.0 * W), n)); bh = np.exp(rng.uniform(np.log(2.0), np.log(3.0 * H), n))
    bboxes = np.stack([cx, cy, bw, bh], 1).astype(np.float32)
    idx = rng.integers(0, F, n).astype(np.int32)
    got = ops.crop_frames(_t(frames, gpu_device), bboxes, idx, scale=1.2).cpu().numpy()
    bad = [i for i in range(n) if not np.array_equal(got[i], crop_ref.crop_to_tensor(frames[idx[i]], bboxes[i], 1.2))]
    assert not bad, (bad[:5], bboxes[bad[:5]])
    assert sum(float(got[i].max()) == 0 for i in range(n)) >= 1         # some boxes lie wholly outside: all-zero crops


def test_crop_frames_rejects_frame_indices_out_of_range(gpu_device):
    """A tracker result that does not belong to the decoded frames must not become an out-of-range device read:
    host indices raise before the launch; device indices are checked by the kernel (zero crop + status 1)."""
    rng = np.random.default_rng(6)
    frames = _t(rng.integers(1, 256, (3, 120, 160, 3), dtype=np.uint8), gpu_device)
    bboxes = np.tile(np.array([[80.0, 60.0, 50.0, 90.0]], np.float32), (4, 1))
    for bad in ([0, 1, 3, 2], [0, -1, 1, 2]):
        with pytest.raises(ValueError, match="frame index out of range"):
            ops.crop_frames(frames, bboxes, np.array(bad, np.int32))
    with pytest.raises(ValueError, match="without a frame index"):
        ops.crop_frames(frames, bboxes)                                  # 4 boxes, 3 frames, no index
    idx = torch.tensor([0, 7, -2, 2], dtype=torch.int32, device=gpu_device)
    crops, status = ops.crop_frames(frames, bboxes, idx, return_status=True)
    assert status.cpu().tolist() == [0, 1, 1, 0]
    assert float(crops[1].abs().max()) == 0 and float(crops[2].abs().max()) == 0 and float(crops[0].max()) > 0
    good = ops.crop_frames(frames, bboxes[:1], np.array([2], np.int32))
    assert torch.equal(good[0], crops[3])
    # through the C ABI without a status pointer: still no fault, zeros
    import ctypes as C
    out = torch.full((4, 3, 224, 224), 5.0, device=gpu_device)
    bb = _t(bboxes, gpu_device)
    st = _lib.load().pr_crop_frames(frames.data_ptr(), 3, 120, 160, 0, idx.data_ptr(), bb.data_ptr(), 4, C.c_float(1.2),
                                    out.data_ptr(), None, None)
    torch.cuda.synchronize()
    assert st == 0 and float(out[1].abs().max()) == 0 and torch.equal(out[0], crops[0])


def test_pipeline_lanes_and_ragged_batches(gpu_device, hmr_pair):
    """Batches in flight on separate streams give the same bits as one stream; empty / single / odd batches work."""
    m, _ = hmr_pair
    layer = SMPLLayer(synth.smpl_model(V=6890, seed=2), device=gpu_device, max_batch=16)
    info = synth.EXAMPLE_INFO
    one = FramePipeline(m, layer, info, with_verts=True, lanes=1)
    three = FramePipeline(m, layer, info, with_verts=True, lanes=3)
    xs = [_t(synth.crops(b, seed=40 + b), gpu_device) for b in (5, 1, 3, 5, 8)]
    want = [{k: v.clone() for k, v in one(x).items()} for x in xs]
    outs = []
    for x in xs:                                   # five batches over three lanes: lanes are reused
        o = three(x)
        FramePipeline.wait(o)                      # current stream waits for that lane
        outs.append({k: v.clone() for k, v in o.items()})
    three.synchronize()
    for w, g in zip(want, outs):
        for k in ("rotmat", "betas", "cam", "euler", "joint_cam", "verts", "reba", "rula", "status"):
            assert torch.equal(w[k], g[k]), k
    empty = one(torch.zeros((0, 3, 224, 224), device=gpu_device))
    assert empty["rotmat"].shape == (0, 24, 3, 3) and empty["reba"].shape == (0, 10)


def test_c_abi_error_paths(gpu_device):
    """Status codes and messages instead of exceptions or crashes across the boundary."""
    import ctypes as C
    lib = _lib.load()
    h = C.c_void_p()
    blob = np.zeros(10, np.float32)
    st = lib.pr_hmr_create(0, blob.ctypes.data, blob.size, 4, 0, -1, C.byref(h))
    assert st == -1 and b"floats" in lib.pr_last_error()
    st = lib.pr_hmr_create(0, blob.ctypes.data, lib.pr_hmr_weight_floats(), 4, 7, -1, C.byref(h))
    assert st == -1 and b"precision" in lib.pr_last_error()
    assert lib.pr_pose_to_euler(None, 1, None, None, None, None) == -1
    m = synth.smpl_model(V=30, seed=1)
    bad_parents = np.array(synth.SMPL_PARENTS, np.int32)
    bad_parents[5] = 9                                                  # not topological
    st = lib.pr_smpl_create(0, m["v_template"].ctypes.data, m["shapedirs"].ctypes.data, m["posedirs"].ctypes.data,
                            m["J_regressor"].ctypes.data, m["weights"].ctypes.data, bad_parents.ctypes.data, None,
                            30, 24, 10, 8, C.byref(h))
    assert st == -1 and b"topological" in lib.pr_last_error()
    with pytest.raises(_lib.PoseRiskHipError):
        ops.conv2d_nhwc(torch.zeros((1, 8, 8, 6), device=gpu_device), np.zeros((64, 6, 1, 1), np.float32))   # Cin % 4


def test_c_abi_refuses_allocation_under_a_graph_capture(gpu_device):
    """pr_hmr_create / pr_smpl_create / pr_hmr_set_streams / pr_*_destroy allocate, copy and synchronise: reached while the
    caller's stream is being captured into a hipGraph they return PR_ERR_INVALID with a message and touch nothing (round 5:
    a released SMPL handle re-created inside a capture aborted the process at the next synchronisation).  The capture
    survives the refused calls, replays with the eager bits, and the same calls succeed once it has ended."""
    import ctypes as C
    lib = _lib.load()
    sd = synth.hmr_state_dict(seed=1)
    sm = synth.smpl_model(V=6890, seed=2)
    m = HMR(max_batch=4).to(gpu_device)
    m.load_state_dict(sd)
    layer = SMPLLayer(sm, device=gpu_device, max_batch=16)
    pipe = FramePipeline(m, layer, synth.EXAMPLE_INFO, with_verts=False)
    static_x = _t(synth.crops(4, seed=11), gpu_device)
    eager = {k: v.clone() for k, v in pipe(static_x).items() if k in ("rotmat", "joint_cam", "reba")}
    fresh, layer2 = HMR(max_batch=4).to(gpu_device), SMPLLayer(sm, device=gpu_device, max_batch=16)
    fresh.load_state_dict(sd)
    torch.cuda.synchronize()
    seen = []
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for what, call in (("pr_hmr_create", lambda: fresh._ensure(4)), ("pr_smpl_create", layer2._ensure),
                           ("pr_hmr_set_streams", lambda: m.set_streams(2)), ("pr_hmr_destroy", m._release),
                           ("pr_smpl_destroy", layer._release)):
            with pytest.raises(_lib.PoseRiskHipError) as e:
                call()
            seen.append((what, str(e.value)))
        # the raw ABI, with the capturing stream declared by hand: status -1, no handle written
        h = C.c_void_p()
        blob = np.zeros(int(lib.pr_hmr_weight_floats()), np.float32)
        lib.pr_declare_stream(torch.cuda.current_stream(gpu_device).cuda_stream, 1)
        assert lib.pr_hmr_create(0, blob.ctypes.data, blob.size, 4, 0, -1, C.byref(h)) == -1 and not h.value
        assert b"captured" in lib.pr_last_error()
        out = pipe(static_x)            # the handles the refused destroys kept are still the captured ones
    for what, msg in seen:
        assert what in msg and "captured into a hipGraph" in msg, (what, msg)
    g.replay()
    torch.cuda.synchronize()
    for k, v in eager.items():
        assert torch.equal(out[k], v), k
    # an undeclared stream is not guarded (C callers that never capture need not declare anything) ...
    lib.pr_declare_stream(None, 0)
    # ... and outside the capture everything that was refused works
    fresh._ensure(4)
    layer2._ensure()
    m.set_streams(2)
    m._release()
    layer._release()
    assert m.handle is None and fresh.handle is not None


def test_hmr_large_batch_is_chunked(gpu_device):
    """A batch beyond one conv launch's 2 GiB tensor limit (B > 512) runs as serial sub-batches, same bits."""
    sd = synth.hmr_state_dict(seed=1)
    m = HMR(max_batch=520).to(gpu_device)
    m.load_state_dict(sd)
    x = torch.rand((520, 3, 224, 224), device=gpu_device)
    r, b, c = m(x)
    r2, b2, c2 = m(x[510:520])
    assert torch.equal(r[510:520], r2) and torch.equal(b[510:520], b2)


def test_frames_forward_is_graph_capturable(gpu_device):
    """The header promises that compute calls do no host synchronisation, allocation or blocking copy, so they can
    be captured into a hipGraph: capture one whole batch (encoder, regressor, Euler, SMPL, scores), replay it on new
    crops, and get the eager call's bits."""
    sd = synth.hmr_state_dict(seed=1)
    m = HMR(max_batch=8).to(gpu_device)
    m.load_state_dict(sd)
    layer = SMPLLayer(synth.smpl_model(V=6890, seed=2), device=gpu_device, max_batch=16)
    pipe = FramePipeline(m, layer, synth.EXAMPLE_INFO, with_verts=True)
    x = _t(synth.crops(8, seed=11), gpu_device)
    static_x = torch.empty_like(x)
    static_x.copy_(x)
    keys = ("rotmat", "betas", "cam", "euler", "joint_cam", "verts", "reba", "rula", "status")
    eager = {k: v.clone() for k, v in pipe(static_x).items() if k in keys}   # also warms every kernel up
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = pipe(static_x)
    static_x.copy_(_t(synth.crops(8, seed=12), gpu_device))
    g.replay()                                             # different crops: results must change ...
    torch.cuda.synchronize()
    assert not torch.equal(out["rotmat"], eager["rotmat"])
    static_x.copy_(x)
    g.replay()                                             # ... and come back bit for bit
    torch.cuda.synchronize()
    for k in keys:
        assert torch.equal(out[k], eager[k]), k


def test_pipeline_graph_mode_has_the_eager_bits(gpu_device):
    """FramePipeline(graph=True): every lane captures its batch into a hipGraph on first use and replays it (bench.py --graph,
    the per-rank host budget of an 8-rank node).  Two lanes, different crops per call, a consumer on another stream between
    calls (release_after): the eager pipeline's bits on every call."""
    sd = synth.hmr_state_dict(seed=1)
    sm = synth.smpl_model(V=6890, seed=2)

    def make(graph):
        m = HMR(max_batch=6).to(gpu_device)
        m.load_state_dict(sd)
        return FramePipeline(m, SMPLLayer(sm, device=gpu_device, max_batch=16), synth.EXAMPLE_INFO, with_verts=True, lanes=2,
                             graph=graph)

    eager, graph = make(False), make(True)
    keys = ("rotmat", "betas", "cam", "euler", "joint_cam", "verts", "reba", "rula", "status")
    side = torch.cuda.Stream(gpu_device)
    for call in range(5):
        x = _t(synth.crops(6, seed=40 + call), gpu_device)
        a, b = eager(x), graph(x)
        FramePipeline.wait(a)
        FramePipeline.wait(b, side)
        with torch.cuda.stream(side):
            got = {k: b[k].clone() for k in keys}
        FramePipeline.release_after(b, side)
        torch.cuda.synchronize()
        for k in keys:
            assert torch.equal(got[k], a[k]), (call, k)


def test_pipeline_graph_is_recaptured_when_the_handles_change(gpu_device):
    """A captured hipGraph bakes in the handles' device pointers.  Whatever reallocates them at the SAME batch size -- new
    weights (load_state_dict releases the handle), a second pipeline growing the shared model's capacity, set_streams --
    must force a recapture: after each, the graph pipeline still has the eager pipeline's bits (it replayed a graph over
    freed memory before the capture was keyed by the models' generations)."""
    sd1, sd2 = synth.hmr_state_dict(seed=1), synth.hmr_state_dict(seed=5)
    sm = synth.smpl_model(V=6890, seed=2)
    m = HMR(max_batch=4).to(gpu_device)
    m.load_state_dict(sd1)
    layer = SMPLLayer(sm, device=gpu_device, max_batch=16)
    graph = FramePipeline(m, layer, synth.EXAMPLE_INFO, with_verts=True, graph=True)
    ref_m = HMR(max_batch=4).to(gpu_device)
    eager = FramePipeline(ref_m, SMPLLayer(sm, device=gpu_device, max_batch=16), synth.EXAMPLE_INFO, with_verts=True)
    keys = ("rotmat", "betas", "cam", "euler", "joint_cam", "verts", "reba", "rula", "status")
    x = _t(synth.crops(4, seed=21), gpu_device)

    def same(tag, sd):
        ref_m.load_state_dict(sd)
        for rep in range(2):                                  # capture, then replay
            a, b = eager(x), graph(x)
            graph.synchronize()
            torch.cuda.synchronize()
            for k in keys:
                assert torch.equal(a[k], b[k]), (tag, rep, k)

    same("first capture", sd1)
    gen = m.generation
    m.load_state_dict(sd2)                                    # releases the handle: every captured pointer is stale
    same("new weights", sd2)
    assert m.generation > gen
    other = FramePipeline(m, layer, synth.EXAMPLE_INFO, with_verts=True)
    other(_t(synth.crops(12, seed=22), gpu_device))           # regrows the shared model beyond max_batch=4
    torch.cuda.synchronize()
    same("capacity regrown through another pipeline", sd2)
    m.set_streams(2)                                          # reallocates the workspaces in place
    same("set_streams", sd2)
    layer._release()                                          # the SMPL handle recreated
    same("SMPL handle recreated", sd2)


def test_regressor_tile_shapes_have_the_same_bits(gpu_device, monkeypatch):
    """fc_rows16_f32 can compute MT x NT output tiles of 16x16 per workgroup (round 5; the default stays 1 x 1: the larger shapes
    halve the operand traffic and measured no gain).  Every output is its wave's ascending-k chain over a quarter of K and the
    same four-term sum whatever the shape, so the regressor's outputs are the same bits for every shape (POSERISK_FC_SHAPE =
    10 MT + NT is read per handle), on a ragged batch (70 frames: partial M tiles in every shape)."""
    sd = synth.hmr_state_dict(seed=1)
    x = _t(synth.crops(70, seed=5), gpu_device)
    outs = {}
    for shape in ("", "11", "21", "41", "12", "22", "42"):
        if shape:
            monkeypatch.setenv("POSERISK_FC_SHAPE", shape)
        m = HMR(max_batch=70).to(gpu_device)
        m.load_state_dict(sd)
        outs[shape] = [t.clone() for t in m(x, return_features=True)]
        del m
    monkeypatch.delenv("POSERISK_FC_SHAPE")
    for shape, got in outs.items():
        for a, b in zip(outs["11"], got):
            assert torch.equal(a, b), shape
    small = HMR(max_batch=8).to(gpu_device)           # and a small batch (1 x 1 tiles) against the same frames of the large one
    small.load_state_dict(sd)
    for a, b in zip(outs[""], small(x[:8], return_features=True)):
        assert torch.equal(a[:8], b)


def test_concurrency_hint_changes_no_bits(gpu_device):
    """pr_hmr_set_concurrency (HMR.set_concurrency, set by FramePipeline to its number of lanes): the persistent kernels'
    grids shrink when other batches are in flight, the assignment of work units to workgroups changes, no result does --
    features and outputs are the same bits at 1, 2 and 3, before and after switching back."""
    sd = synth.hmr_state_dict(seed=1)
    m = HMR(max_batch=9).to(gpu_device)
    m.load_state_dict(sd)
    x = _t(synth.crops(9, seed=77), gpu_device)
    want = [t.clone() for t in m(x, return_features=True)]
    for n in (2, 3, 1):
        m.set_concurrency(n)
        got = m(x, return_features=True)
        for a, b in zip(want, got):
            assert torch.equal(a, b), n
    with pytest.raises(_lib.PoseRiskHipError):
        m.set_concurrency(1)
        _lib.check(_lib.load().pr_hmr_set_concurrency(m.handle, 0), "pr_hmr_set_concurrency")
    pipe = FramePipeline(m, SMPLLayer(synth.smpl_model(V=6890, seed=2), device=gpu_device, max_batch=16), synth.EXAMPLE_INFO, lanes=3)
    pipe(x)
    pipe.synchronize()
    assert m._concurrency == 3 and all(l.hmr._concurrency == 3 for l in pipe._lanes)


@pytest.mark.parametrize("lanes", [1, 2])
def test_frame_feed_equals_the_resident_path(gpu_device, lanes):
    """feed.FrameFeed: host frames through a pinned ring (upload, crop, pose / SMPL / scores, read-back, each on its own
    stream) against the same frames resident on the GPU through ops.crop_frames + FramePipeline: the same bits, in order,
    for whole batches and a ragged last one, with the ring reused several times."""
    from poserisk_release_amd.feed import FrameFeed
    sd = synth.hmr_state_dict(seed=1)
    sm = synth.smpl_model(V=6890, seed=2)
    B, H, W, F = 6, 120, 160, 6 * 7 + 4
    rng = np.random.default_rng(21)
    frames = rng.integers(0, 256, (F, H, W, 3), dtype=np.uint8)
    bboxes = np.stack([rng.uniform(40, 120, F), rng.uniform(30, 90, F), rng.uniform(30, 80, F), rng.uniform(40, 100, F)], 1).astype(np.float32)

    def make():
        m = HMR(max_batch=B).to(gpu_device)
        m.load_state_dict(sd)
        return FramePipeline(m, SMPLLayer(sm, device=gpu_device, max_batch=16), synth.EXAMPLE_INFO, lanes=lanes)

    ref_pipe = make()
    want = {k: [] for k in ("euler", "joint_cam", "axis_angle", "reba", "rula", "status")}
    dframes = _t(frames, gpu_device)
    for i in range(0, F, B):
        out = ref_pipe(ops.crop_frames(dframes[i:i + B], bboxes[i:i + B]))
        FramePipeline.wait(out)
        for k in want:
            want[k].append(out[k].cpu().numpy().copy())
    feed = FrameFeed(make(), B, (H, W), gpu_device)
    assert len(feed.slots) == lanes + 2
    got = {k: [] for k in want}
    crop_status = []
    for res in feed.run((frames[i:i + B], bboxes[i:i + B]) for i in range(0, F, B)):
        for k in want:
            got[k].append(res[k])
        crop_status.append(res["crop_status"])      # pr_crop_frames' flag rides in the batch's one read-back
    for k in want:
        a, b = np.concatenate(want[k]), np.concatenate(got[k])
        assert a.shape == b.shape and a.shape[0] == F and np.array_equal(a, b, equal_nan=True), k
    cs = np.concatenate(crop_status)
    assert cs.shape == (F,) and not cs.any()
    # the ragged last batch ran at the slot's full B: every lane still holds ONE shape, every slot pinned its blob once
    assert all(list(lane.bufs) == [(B, str(gpu_device))] for lane in feed.pipe._lanes)
    assert all(s.h_blob is None or s.layout_B == B for s in feed.slots)
    # ... over BLANK tail crops, not over an earlier batch's frames (round 5's advisor): the two tail rows of the last
    # slot's blob are the results of all-zero crops -- equal to each other, different from the batch's last real frame
    last = feed.slots[(feed.submitted - 1) % len(feed.slots)]
    p0, nb, shape, _ = last.layout["joint_cam"]
    jc = last.h_blob.numpy()[p0:p0 + nb].view(np.float32).reshape((B,) + tuple(shape))
    n_last = F % B
    assert last.n == n_last and np.array_equal(jc[n_last], jc[n_last + 1]) and not np.array_equal(jc[n_last], jc[n_last - 1])


@pytest.mark.parametrize("case", [
    # (B, H, Cin, Cout, residual): layer3's conv3, layer2's conv3, layer2.0's conv1, ragged pixel counts
    (4, 14, 256, 1024, True), (3, 28, 128, 512, True), (2, 56, 256, 128, False), (3, 7, 256, 192, True), (1, 5, 128, 256, False)],
    ids=lambda c: "x".join(map(str, c)))
def test_conv_register_weights_unit_shapes_have_the_same_bits(gpu_device, case, monkeypatch):
    """Round 5: a unit of conv1x1_regw_f32 is (NB blocks of 64 channels, T tiles of 16 pixels); with NB > 1 a wave keeps its
    16 channels of NB blocks in registers and the pixel rows stream through LDS once per NB blocks instead of once per block
    (half / a quarter of the layer's L2 -> LDS traffic).  An output's fmaf chain is bias, then k ascending, whatever (T, NB):
    every variant gives the bits of T = 2, NB = 1, round 4's kernel (the grouped form -- the 36 GEMMs of a Winograd layer -- runs
    with the defaults in the Winograd tests and the pipeline tests).  (POSERISK_REGW_T / POSERISK_REGW_NB are read per call by the stand-alone entry.)"""
    B, H, Cin, Cout, with_res = case
    rng = np.random.default_rng(B * 10 + H)
    g = torch.Generator(device=gpu_device).manual_seed(H * 3 + Cin)
    x = torch.randn((B, H, H, Cin), generator=g, device=gpu_device)
    w = (rng.standard_normal((Cout, Cin, 1, 1)) / np.sqrt(Cin)).astype(np.float32)
    bias = rng.standard_normal(Cout).astype(np.float32)
    res = torch.randn((B, H, H, Cout), generator=g, device=gpu_device) if with_res else None
    monkeypatch.setenv("POSERISK_REGW_T", "2")
    monkeypatch.setenv("POSERISK_REGW_NB", "1")
    want, _ = ops.conv2d_nhwc(x, w, bias, res, relu=True, tile_cfg=400)
    want = want.clone()
    ref = torch.nn.functional.conv2d(x.cpu().double().permute(0, 3, 1, 2), torch.from_numpy(w).double(),
                                     torch.from_numpy(bias).double()).permute(0, 2, 3, 1)
    ref = torch.relu(ref + res.cpu().double()) if with_res else torch.relu(ref)
    assert float((want.cpu().double() - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
    for T in (1, 2):
        for NB in (1, 2, 4):
            monkeypatch.setenv("POSERISK_REGW_T", str(T))
            monkeypatch.setenv("POSERISK_REGW_NB", str(NB))       # (falls back to the largest block count the shape allows)
            got, _ = ops.conv2d_nhwc(x, w, bias, res, relu=True, tile_cfg=400)
            assert torch.equal(got, want), (T, NB)
    monkeypatch.delenv("POSERISK_REGW_T")
    monkeypatch.delenv("POSERISK_REGW_NB")
    got, _ = ops.conv2d_nhwc(x, w, bias, res, relu=True, tile_cfg=400)     # the defaults
    assert torch.equal(got, want)


@pytest.mark.parametrize("case", [
    # (B, H, Cin, Cout, residual): layer3's conv3, layer2's conv3, layer1's conv1, ragged pixel counts (M % 32 != 0)
    (4, 14, 256, 1024, True), (3, 28, 128, 512, True), (2, 56, 256, 64, False), (3, 7, 256, 64, True), (1, 5, 128, 192, False)],
    ids=lambda c: "x".join(map(str, c)))
def test_conv_register_weights_matches_torch(gpu_device, case):
    """tile_cfg 400 = conv1x1_regw_f32: a wave's 16 x K weight slice stays in registers, only the activations stream through
    LDS.  Against torch fp64 (its k order is its own, not the tile kernel's); deterministic; a frame's bits do not depend on
    its batch; shapes the kernel does not take are refused by name."""
    B, H, Cin, Cout, with_res = case
    rng = np.random.default_rng(B * 100 + H)
    g = torch.Generator(device=gpu_device).manual_seed(H + Cin)
    x = torch.randn((B, H, H, Cin), generator=g, device=gpu_device)
    w = (rng.standard_normal((Cout, Cin, 1, 1)) / np.sqrt(Cin)).astype(np.float32)
    bias = rng.standard_normal(Cout).astype(np.float32)
    res = torch.randn((B, H, H, Cout), generator=g, device=gpu_device) if with_res else None
    ref = torch.nn.functional.conv2d(x.cpu().double().permute(0, 3, 1, 2), torch.from_numpy(w).double(),
                                     torch.from_numpy(bias).double()).permute(0, 2, 3, 1)
    if with_res:
        ref = ref + res.cpu().double()
    ref = torch.relu(ref)
    y, _ = ops.conv2d_nhwc(x, w, bias, res, relu=True, tile_cfg=400)
    err = float((y.cpu().double() - ref).abs().max())
    assert err < 2e-5 * max(1.0, float(ref.abs().max())), err
    y2, _ = ops.conv2d_nhwc(x, w, bias, res, relu=True, tile_cfg=400)
    assert torch.equal(y, y2)
    y1, _ = ops.conv2d_nhwc(x[B - 1:], w, bias, res[B - 1:] if with_res else None, relu=True, tile_cfg=400)
    assert torch.equal(y1[0], y[B - 1])
    yt, _ = ops.conv2d_nhwc(x, w, bias, res, relu=True, tile_cfg=8)
    assert float((y - yt).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
    yn, _ = ops.conv2d_nhwc(x, w, None, None, relu=False, tile_cfg=400)
    refn = torch.nn.functional.conv2d(x.cpu().double().permute(0, 3, 1, 2), torch.from_numpy(w).double()).permute(0, 2, 3, 1)
    assert float((yn.cpu().double() - refn).abs().max()) < 2e-5 * max(1.0, float(refn.abs().max()))
    with pytest.raises(_lib.PoseRiskHipError):
        ops.conv2d_nhwc(x[..., :64].contiguous(), w[:, :64], bias, None, relu=True, tile_cfg=400)


def test_stem_pool_f32_matches_torch(gpu_device):
    """stem_pool_f32 (conv1 as 4x4 taps over the 12-channel space-to-depth image + bias + ReLU + MaxPool2d(3,2,1), weights in
    registers, pooling in registers) against torch fp32 on CPU; every band of every image incl. the image borders; a frame's
    bits do not depend on its batch or position."""
    rng = np.random.default_rng(33)
    B = 3
    x = rng.standard_normal((B, 112, 112, 12)).astype(np.float32)
    # the 7x7 kernel in the 4x4 taps' 8x8 window (a zero row and a zero column in front), as pr_hmr_create lays it out
    w7 = (rng.standard_normal((64, 3, 7, 7)) / np.sqrt(147)).astype(np.float32)
    w = np.zeros((64, 12, 4, 4), np.float32)
    for kh in range(7):
        for kw in range(7):
            th, di, tw, dj = (kh + 1) >> 1, (kh + 1) & 1, (kw + 1) >> 1, (kw + 1) & 1
            w[:, (2 * di + dj) * 3:(2 * di + dj) * 3 + 3, th, tw] = w7[:, :, kh, kw]
    bias = rng.standard_normal(64).astype(np.float32)
    xt = torch.from_numpy(x).permute(0, 3, 1, 2)
    # window rows y-2 .. y+1: pad 2 up/left, 1 down/right
    conv = torch.nn.functional.conv2d(torch.nn.functional.pad(xt.double(), (2, 1, 2, 1)), torch.from_numpy(w).double(),
                                      torch.from_numpy(bias).double())
    ref = torch.nn.functional.max_pool2d(torch.relu(conv), 3, 2, 1).permute(0, 2, 3, 1).numpy()
    y, _ = ops.stem_pool_f32_nhwc(_t(x, gpu_device), w, bias)
    got = y.cpu().numpy()
    assert got.shape == (B, 56, 56, 64)
    err = np.abs(got - ref).max()
    measured("stem_pool_f32 vs torch fp64 (max abs)", float(err), 2e-5)
    assert err < 2e-5 * max(1.0, np.abs(ref).max())
    y1, _ = ops.stem_pool_f32_nhwc(_t(x[2:3], gpu_device), w, bias)
    assert torch.equal(y1[0], y[2])
    with pytest.raises(ValueError):
        ops.stem_pool_f32_nhwc(_t(x[:, :56], gpu_device), w, bias)
    # the kernel packs the half-empty taps of the zero row / column together: dense 4x4x12 weights are refused by name
    dense = (rng.standard_normal((64, 12, 4, 4)) / np.sqrt(192)).astype(np.float32)
    with pytest.raises(_lib.PoseRiskHipError, match="7x7 kernel"):
        ops.stem_pool_f32_nhwc(_t(x, gpu_device), dense, bias)


def test_hmr_capacity_error_is_a_status_not_a_crash(gpu_device):
    """A batch beyond the handle's max_batch returns PR_ERR_CAPACITY (-4) through the C ABI; the Python mirror
    re-creates the handle for the larger batch instead."""
    import ctypes as C
    from poserisk_release_amd import weights
    lib = _lib.load()
    blob = weights.flatten_state_dict(synth.hmr_state_dict(seed=1))
    h = C.c_void_p()
    _lib.check(lib.pr_hmr_create(gpu_device.index or 0, blob.ctypes.data, blob.size, 2, 0, -1, C.byref(h)), "create")
    x = torch.rand((3, 3, 224, 224), device=gpu_device)
    rot = torch.empty((3, 24, 3, 3), device=gpu_device)
    st = lib.pr_hmr_forward(h, x.data_ptr(), 3, rot.data_ptr(), None, None, None, None, None)
    assert st == -4 and "max_batch" in lib.pr_last_error().decode()
    assert lib.pr_hmr_forward(h, x.data_ptr(), 2, rot.data_ptr(), None, None, None, None, None) == 0
    torch.cuda.synchronize()
    assert lib.pr_hmr_destroy(h) == 0
    m = HMR(max_batch=2).to(gpu_device)
    m.load_state_dict(synth.hmr_state_dict(seed=1))
    r3 = m(x)[0]                                            # grows the handle
    torch.testing.assert_close(r3[:2], rot[:2], rtol=0, atol=0)
