"""Interpolation points for Winograd F(4x4,3x3) in fp32: Cook-Toom matrices for any point set (exact rationals) and
their single-layer error against an fp64 direct convolution, relative to the direct fp32 convolution's own error.
Development tool for DESIGN.md 3.1b (which points the `5` conv form uses); CPU only.

    python scripts/wino_points.py            # the candidate table
"""
import itertools
import sys
from fractions import Fraction as Fr

import numpy as np
import torch
import torch.nn.functional as F


def cook_toom(points, m=4, r=3):
    """A^T [m x n], G [n x r], B^T [n x n] for F(m, r) on the finite `points` (n - 1 of them) plus infinity, so that
    y = A^T [(G g) * (B^T d)].  Exact rationals."""
    n = m + r - 1
    assert len(points) == n - 1
    p = [Fr(x) for x in points]
    AT = [[(p[j] ** i if j < n - 1 else Fr(int(i == m - 1))) for j in range(n)] for i in range(m)]
    G = []
    for j in range(n - 1):
        N = Fr(1)
        for l in range(n - 1):
            if l != j:
                N *= p[j] - p[l]
        G.append([p[j] ** k / N for k in range(r)])
    G.append([Fr(0)] * (r - 1) + [Fr(1)])

    def polymul(a, b):
        out = [Fr(0)] * (len(a) + len(b) - 1)
        for i, x in enumerate(a):
            for j, y in enumerate(b):
                out[i + j] += x * y
        return out
    BT = []
    for j in range(n - 1):
        poly = [Fr(1)]
        for l in range(n - 1):
            if l != j:
                poly = polymul(poly, [-p[l], Fr(1)])
        BT.append(poly + [Fr(0)] * (n - len(poly)))          # degree n-2 -> n-1 coefficients, pad to n
    poly = [Fr(1)]
    for l in range(n - 1):
        poly = polymul(poly, [-p[l], Fr(1)])
    BT.append(poly)                                          # M(x) = prod (x - p_l), degree n-1
    return AT, G, BT


def rescale(AT, G, BT, row_scale):
    """Move a factor s_j from G's row j to B^T's row j (the products are unchanged): G_j /= s_j, BT_j *= s_j."""
    G = [[x / row_scale[j] for x in row] for j, row in enumerate(G)]
    BT = [[x * row_scale[j] for x in row] for j, row in enumerate(BT)]
    return AT, G, BT


def as_np(M):
    return np.array([[float(x) for x in row] for row in M], np.float64)


def check_exact(AT, G, BT):
    rng = np.random.default_rng(0)
    d, g = rng.normal(size=6), rng.normal(size=3)
    y = as_np(AT) @ ((as_np(G) @ g) * (as_np(BT) @ d))
    ref = np.array([d[i] * g[0] + d[i + 1] * g[1] + d[i + 2] * g[2] for i in range(4)])
    assert np.allclose(y, ref, atol=1e-9), (y, ref)


def layer_error(AT, G, BT, x, w, m=4):
    """fp32 Winograd of one 3x3 layer against fp64 direct; returns (rms, max) of the error and of the fp32 direct conv's."""
    n = m + 2
    B_, C, H, W = x.shape
    th, tw = -(-H // m), -(-W // m)
    ref = F.conv2d(x.double(), w.double(), padding=1)
    direct = F.conv2d(x, w, padding=1).double() - ref
    xp = F.pad(x, (1, tw * m + 1 - W, 1, th * m + 1 - H))
    d = xp.unfold(2, n, m).unfold(3, n, m)
    bt = torch.from_numpy(as_np(BT)).float()
    v = torch.einsum("ij,bcyxjk->bcyxik", bt, d)
    v = torch.einsum("bcyxik,lk->bcyxil", v, bt)
    g = torch.from_numpy(as_np(G))
    u = torch.einsum("ij,ocjk,lk->ocil", g, w.double(), g).float()
    mm = torch.einsum("bcyxil,ocil->boyxil", v, u)
    at = torch.from_numpy(as_np(AT)).float()
    y = torch.einsum("ij,boyxjk->boyxik", at, mm)
    y = torch.einsum("boyxik,lk->boyxil", y, at)
    y = y.permute(0, 1, 2, 4, 3, 5).reshape(B_, -1, th * m, tw * m)[:, :, :H, :W]
    e = y.double() - ref
    s = float(ref.pow(2).mean().sqrt())
    return float(e.pow(2).mean().sqrt()) / s, float(e.abs().max()) / s, float(direct.pow(2).mean().sqrt()) / s, float(direct.abs().max()) / s


def test_data(seed, C=128, Co=128, H=28, heavy=True):
    rng = np.random.default_rng(seed)
    x = np.maximum(rng.normal(0.3, 1.0, (4, C, H, H)), 0)            # post-ReLU, offset
    w = rng.normal(0, np.sqrt(2.0 / (9 * C)), (Co, C, 3, 3))
    if heavy:                                                        # the stress set's spreads (tests/stress_weights.py)
        x = x * np.exp(rng.normal(0, 1.0, (1, C, 1, 1)))
        w = w * np.exp(rng.normal(0, 0.8, w.shape)) * np.exp(rng.normal(0, 1.0, (1, C, 1, 1)))
    return torch.from_numpy(x.astype(np.float32)), torch.from_numpy(w.astype(np.float32))


CANDIDATES = {
    "lavin 0,+-1,+-2": [0, 1, -1, 2, -2],
    "0,+-1,+-1/2": [0, 1, -1, Fr(1, 2), Fr(-1, 2)],
    "0,+-1/2,+-2": [0, Fr(1, 2), Fr(-1, 2), 2, -2],
    "0,+-1,1/2,-2": [0, 1, -1, Fr(1, 2), -2],
    "0,+-1,2,-1/2": [0, 1, -1, 2, Fr(-1, 2)],
    "0,+-1,+-3/2": [0, 1, -1, Fr(3, 2), Fr(-3, 2)],
    "0,+-1,+-2/3": [0, 1, -1, Fr(2, 3), Fr(-2, 3)],
    "0,+-3/4,+-4/3": [0, Fr(3, 4), Fr(-3, 4), Fr(4, 3), Fr(-4, 3)],
    "0,+-1/2,+-3/2": [0, Fr(1, 2), Fr(-1, 2), Fr(3, 2), Fr(-3, 2)],
    "0,+-2/3,+-3/2": [0, Fr(2, 3), Fr(-2, 3), Fr(3, 2), Fr(-3, 2)],
    "0,+-1/sqrt2~0.7,+-1.4": [0, Fr(7, 10), Fr(-7, 10), Fr(7, 5), Fr(-7, 5)],
    "0,+-5/8,+-8/5": [0, Fr(5, 8), Fr(-5, 8), Fr(8, 5), Fr(-8, 5)],
}

if __name__ == "__main__":
    torch.set_num_threads(8)
    shapes = [(128, 128, 28), (256, 256, 14), (512, 512, 7)]
    print(f"{'points':28s}  " + "  ".join(f"C={c:3d}@{h:2d}: rms/direct max/direct" for c, _, h in shapes))
    for name, pts in CANDIDATES.items():
        AT, G, BT = cook_toom(pts)
        check_exact(AT, G, BT)
        row = f"{name:28s}"
        for C, Co, H in shapes:
            acc = np.zeros(4)
            for seed in range(3):
                x, w = test_data(seed, C, Co, H)
                acc += np.array(layer_error(AT, G, BT, x, w))
            acc /= 3
            row += f"  {acc[0] / acc[2]:10.2f} {acc[1] / acc[3]:10.2f}      "
        print(row, flush=True)
