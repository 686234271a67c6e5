"""The device-free half of pr_hmr_create under AddressSanitizer + UndefinedBehaviorSanitizer, on the CPU.

`poserisk_release_amd/csrc/host_plan.cc` (blob layout, BatchNorm folding, every kernel family's weight packing, the Winograd
G-transform, the 53-convolution launch plan, workspace sizes) only ever runs behind `pr_hmr_create`, which needs a GPU -- so
its index arithmetic had never run under a sanitizer (SURVEY.md section 5 asks for a sanitizer build of the host C++; round
3's advisor found host UB there by reading).  `tests/native/host_plan_check.cc` builds the full ResNet-50 plan for
B in {1, 7, 64, 230, 256, 460}, both precisions, every conv form and every plan-shaping switch, against exact-size heap
blocks, and checks the plans structurally (symbolic dataflow over the buffer rotation, routing, 4 087 136 256 MAC per frame,
launch counts, workspace sizes).  Here: it is compiled with g++ -fsanitize=address,undefined, run, and the packed weights it
produced are compared with a numpy restatement from the same state dict (weights.py's blob order = lib/core/base.py:83-84's
checkpoint['model'])."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import REPO
from poserisk_release_amd import synth, weights

CSRC = os.path.join(REPO, "poserisk_release_amd", "csrc")


@pytest.fixture(scope="module")
def native_run(tmp_path_factory):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    d = tmp_path_factory.mktemp("host_plan")
    exe = str(d / "host_plan_check")
    cmd = [gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
           "-Wall", "-Werror", "-o", exe, os.path.join(REPO, "tests", "native", "host_plan_check.cc"), os.path.join(CSRC, "host_plan.cc")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
    sd = synth.hmr_state_dict(seed=1)
    blob = weights.flatten_state_dict(sd)
    blob.tofile(str(d / "blob.f32"))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    for k in list(env):
        if k.startswith("POSERISK_"):
            del env[k]                      # the binary sets the plan's switches itself, one at a time
    r = subprocess.run([exe, str(d / "blob.f32"), str(d / "manifest.json"), str(d / "dump.bin")], capture_output=True, text=True,
                       timeout=900, env=env)
    return r, sd, json.load(open(d / "manifest.json")) if os.path.exists(d / "manifest.json") else None, str(d / "dump.bin")


def test_host_plan_is_clean_under_asan_and_ubsan(native_run):
    r, _, manifest, _ = native_run
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-6000:])
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr and "CHECK FAILED" not in r.stderr, r.stderr[-6000:]
    last = r.stdout.strip().splitlines()[-1]
    assert last.startswith("host_plan_check: ") and last.endswith(" 0 failures"), last
    assert int(last.split()[1]) >= 35                      # plans built: 8 forms + 12 batch sizes x precisions + 15 switches
    assert [m["precision"] for m in manifest] == [0, 1]


def _fold(sd, conv, bn):
    """BatchNorm (eval, eps 1e-5) folded in double: scale[o], bias[o] (hmr.py / SPIN models/hmr.py: conv -> bn)."""
    g, b, mu, var = (np.asarray(sd[f"{bn}.{s}"], np.float64) for s in ("weight", "bias", "running_mean", "running_var"))
    s = g / np.sqrt(var + 1e-5)
    return np.asarray(sd[f"{conv}.weight"], np.float32), s, b - mu * s


def _pack_f32(w, s, cin_pad=None):
    """[Cout][Kpad], k = (kh KW + kw) cin_pad + ci, zero padded to 32, each value (float)((double)w * s)."""
    co, ci, kh, kw = w.shape
    cp = cin_pad or ci
    v = (w.astype(np.float64) * s[:, None, None, None]).astype(np.float32)
    out = np.zeros((co, kh * kw, cp), np.float32)
    out[:, :, :ci] = v.transpose(0, 2, 3, 1).reshape(co, kh * kw, ci)
    out = out.reshape(co, -1)
    kpad = -(-out.shape[1] // 32) * 32
    return np.pad(out, ((0, 0), (0, kpad - out.shape[1])))


def _bf16(a):
    """float32 -> bf16 bits, round to nearest even (finite values)."""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def _s2d(w):
    """conv1.weight [64,3,7,7] -> the 4x4-tap kernel over the 12-channel space-to-depth image (zero row / column in front):
    W2[o][(2 di + dj) 3 + c][th][tw] = W[o][c][2 th + di - 1][2 tw + dj - 1]."""
    out = np.zeros((64, 12, 4, 4), np.float32)
    for kh in range(7):
        for kw in range(7):
            th, di, tw, dj = (kh + 1) >> 1, (kh + 1) & 1, (kw + 1) >> 1, (kw + 1) & 1
            out[:, (2 * di + dj) * 3:(2 * di + dj) * 3 + 3, th, tw] = w[:, :, kh, kw]
    return out


def _dumped(manifest_entry, dump, i, dtype):
    u = manifest_entry["uploads"][i]
    assert u["dump_at"] >= 0, f"upload {i} ({u['bytes']} B) is not the first of its size"
    return np.fromfile(dump, dtype=dtype, count=u["bytes"] // np.dtype(dtype).itemsize, offset=u["dump_at"])


def test_packed_weights_equal_a_numpy_restatement(native_run):
    r, sd, manifest, dump = native_run
    assert r.returncode == 0 and manifest is not None
    f32, b16 = manifest
    ups = f32["uploads"]
    assert [u["bytes"] for u in ups[-5:]] == [64 * 2048 * 4, 64 * 1024 * 4, 64 * 1024 * 4, 64 * 1024 * 4, 64 * 192 * 4]   # regressor workspaces at B=64
    assert all(u["zeros"] for u in ups[-5:]) and not any(u["zeros"] for u in ups[:-5])
    # fp32: stem on the space-to-depth image, BN folded in double
    w, s, b = _fold(sd, "conv1", "bn1")
    np.testing.assert_array_equal(_dumped(f32, dump, 0, np.float32).reshape(64, 192), _pack_f32(_s2d(w), s))
    np.testing.assert_array_equal(_dumped(f32, dump, 1, np.float32), b.astype(np.float32))
    # layer1.0.conv1 (1x1, 64 -> 64) and conv2 (3x3): [Cout][K] with ci fastest
    w, s, b = _fold(sd, "layer1.0.conv1", "layer1.0.bn1")
    np.testing.assert_array_equal(_dumped(f32, dump, 2, np.float32).reshape(64, 64), _pack_f32(w, s))
    w, s, b = _fold(sd, "layer1.0.conv2", "layer1.0.bn2")
    np.testing.assert_array_equal(_dumped(f32, dump, 4, np.float32).reshape(64, 576), _pack_f32(w, s))
    # layer1.0's conv3 with its downsample branch side by side: [256][64 + 64], biases summed in double
    w3, s3, b3 = _fold(sd, "layer1.0.conv3", "layer1.0.bn3")
    wd, sdn, bd = _fold(sd, "layer1.0.downsample.0", "layer1.0.downsample.1")
    np.testing.assert_array_equal(_dumped(f32, dump, 6, np.float32).reshape(256, 128), np.concatenate([_pack_f32(w3, s3), _pack_f32(wd, sdn)], 1))
    np.testing.assert_array_equal(_dumped(f32, dump, 7, np.float32), (b3 + bd).astype(np.float32))
    # the first Winograd layer (layer2.1.conv2, form 5): U = G g G^T in double on the points 0, +-11/16, +-3/2
    iu = next(i for i, u in enumerate(ups) if u["bytes"] == 36 * 128 * 128 * 4)
    w, s, _ = _fold(sd, "layer2.1.conv2", "layer2.1.bn2")
    a, bb = 11.0 / 16.0, 1.5
    a2, b2 = a * a, bb * bb
    Na, Nb = 2 * a2 * (a2 - b2), 2 * b2 * (b2 - a2)
    G = np.array([[1 / (a2 * b2), 0, 0], [1 / Na, a / Na, a2 / Na], [1 / Na, -a / Na, a2 / Na], [1 / Nb, bb / Nb, b2 / Nb],
                  [1 / Nb, -bb / Nb, b2 / Nb], [0, 0, 1]])
    g = w.astype(np.float64) * s[:, None, None, None]                      # [Cout][Cin][3][3]
    U = np.einsum("ia,ocab,jb->ijoc", G, g, G).reshape(36, 128, 128)
    got = _dumped(f32, dump, iu, np.float32).reshape(36, 128, 128)
    np.testing.assert_allclose(got, U, rtol=3e-7, atol=1e-12)           # same formula, numpy's summation order: float rounding only
    # regressor: fc1 split into its feature part and its state part (157 inputs padded to 192), decoders stacked, init state
    n = len(ups) - 5
    fc1 = np.asarray(sd["fc1.weight"], np.float32)
    np.testing.assert_array_equal(_dumped(f32, dump, n - 9, np.float32).reshape(1024, 2048), fc1[:, :2048])
    fc1s = np.zeros((1024, 192), np.float32)
    fc1s[:, :157] = fc1[:, 2048:]
    np.testing.assert_array_equal(_dumped(f32, dump, n - 7, np.float32).reshape(1024, 192), fc1s)
    dec = np.zeros((192, 1024), np.float32)
    dec[:144], dec[144:154], dec[154:157] = sd["decpose.weight"], sd["decshape.weight"], sd["deccam.weight"]
    np.testing.assert_array_equal(_dumped(f32, dump, n - 3, np.float32).reshape(192, 1024), dec)
    decb = np.zeros(192, np.float32)
    decb[:144], decb[144:154], decb[154:157] = sd["decpose.bias"], sd["decshape.bias"], sd["deccam.bias"]
    np.testing.assert_array_equal(_dumped(f32, dump, n - 2, np.float32), decb)
    init = np.zeros(160, np.float32)
    init[:144], init[144:154], init[154:157] = np.ravel(sd["init_pose"]), np.ravel(sd["init_shape"]), np.ravel(sd["init_cam"])
    np.testing.assert_array_equal(_dumped(f32, dump, n - 1, np.float32), init)
    # bf16: the stem over the 16-channel space-to-depth image, k = tap * 16 + c, values rounded to nearest even
    w, s, b = _fold(sd, "conv1", "bn1")
    v = (_s2d(w).astype(np.float64) * s[:, None, None, None]).astype(np.float32)       # [64][12][4][4]
    want = np.zeros((64, 16, 16), np.float32)                                            # [o][tap][c]
    want[:, :, :12] = v.transpose(0, 2, 3, 1).reshape(64, 16, 12)
    np.testing.assert_array_equal(_dumped(b16, dump, 0, np.uint16).reshape(64, 256), _bf16(want.reshape(64, 256)))
