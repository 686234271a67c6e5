"""GPU-crop front-end oracle: bbox -> affine matrix -> cv2.warpAffine(INTER_LINEAR, BORDER_CONSTANT) -> ToTensor.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED: the reference does this with OpenCV
(``lib/utils/_img_utils.py:53-101`` gen_trans_from_patch_cv / generate_patch_image_cv, ``:219-252``
get_single_image_crop_demo, ``:259-266`` convert_cvimg_to_tensor; called from ``data/demo_dataset.py:58-74``),
opencv-python is unpinned (``requirements.txt:10``) and absent from this image, and the reference holds no
fixture for a crop.  This file restates OpenCV's published fixed-point algorithm for 8-bit bilinear
warpAffine (AB_BITS = 10, INTER_BITS = 5, INTER_REMAP_COEF_BITS = 15, 32x32 weight table whose four
coefficients are corrected to sum to 32768, round-half-even `saturate_cast<int>`), which the HIP kernel
must then match bit for bit.
"""
import numpy as np

AB_BITS, INTER_BITS, COEF_BITS = 10, 5, 15
AB_SCALE, TAB = 1 << AB_BITS, 1 << INTER_BITS
COEF_SCALE = 1 << COEF_BITS


def affine_from_bbox(bbox, scale=1.2, crop=224):
    """gen_trans_from_patch_cv with rot = 0 (_img_utils.py:53-86): the 2x3 forward matrix (double), built
    from float32-rounded control points as the reference builds them."""
    cx, cy, w, h = [float(v) for v in bbox]
    f32 = np.float32
    down = f32((h * scale) * 0.5)          # np.array([0, src_h*0.5], dtype=float32)
    right = f32((w * scale) * 0.5)
    sx0, sy0 = float(f32(cx)), float(f32(cy))
    sy1 = float(f32(cy + float(down)))      # src[1] = center + downdir, stored as float32
    sx2 = float(f32(cx + float(right)))
    half = crop * 0.5
    a = half / (sx2 - sx0)                  # getAffineTransform: x' = half + a (x - sx0)
    d = half / (sy1 - sy0)
    return np.array([[a, 0.0, half - a * sx0], [0.0, d, half - d * sy0]], dtype=np.float64)


def invert_affine(M):
    """cv::warpAffine's in-place inversion of the forward matrix."""
    M = M.astype(np.float64).copy().reshape(-1)
    D = M[0] * M[4] - M[1] * M[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[4] * D, M[0] * D
    M[0] = A11; M[1] *= -D; M[3] *= -D; M[4] = A22
    b1 = -M[0] * M[2] - M[1] * M[5]
    b2 = -M[3] * M[2] - M[4] * M[5]
    M[2], M[5] = b1, b2
    return M.reshape(2, 3)


def bilinear_table():
    """initInterTab2D(INTER_LINEAR, fixpt): int16[32*32, 4] (w00, w01, w10, w11), each row sums to 32768."""
    t = np.arange(TAB, dtype=np.float32) / np.float32(TAB)
    tab1 = np.stack([np.float32(1) - t, t], 1)                      # [32,2] float32
    out = np.zeros((TAB * TAB, 4), np.int32)
    for fy in range(TAB):
        for fx in range(TAB):
            w = np.array([tab1[fy, 0] * tab1[fx, 0], tab1[fy, 0] * tab1[fx, 1],
                          tab1[fy, 1] * tab1[fx, 0], tab1[fy, 1] * tab1[fx, 1]], np.float32)
            iw = np.clip(np.rint(w.astype(np.float64) * COEF_SCALE), -32768, 32767).astype(np.int32)  # saturate_cast<short>
            diff = int(iw.sum()) - COEF_SCALE
            if diff != 0:
                # Only the (0,0) entry is affected: 1.0 * 32768 saturates to 32767.  OpenCV's correction
                # loop starts at the centre tap [ksize/2][ksize/2] = the LAST of the four bilinear taps and,
                # for ksize = 2, finds nothing larger, so the missing unit lands on w11: (32767, 0, 0, 1).
                iw[3] -= diff
            out[fy * TAB + fx] = iw
    return out.astype(np.int16)


_TABLE = None


def warp_affine_u8(img, M_fwd, out_hw=(224, 224)):
    """cv2.warpAffine(img u8[H,W,C], M, (W',H'), INTER_LINEAR, BORDER_CONSTANT=0) restated."""
    global _TABLE
    if _TABLE is None:
        _TABLE = bilinear_table().astype(np.int64)
    H, W, C = img.shape
    Mi = invert_affine(M_fwd).reshape(-1)
    oh, ow = out_hw
    xs = np.arange(ow, dtype=np.float64)
    ys = np.arange(oh, dtype=np.float64)
    adelta = np.rint(Mi[0] * xs * AB_SCALE).astype(np.int64)
    bdelta = np.rint(Mi[3] * xs * AB_SCALE).astype(np.int64)
    rd = AB_SCALE // TAB // 2
    X0 = np.rint((Mi[1] * ys + Mi[2]) * AB_SCALE).astype(np.int64) + rd
    Y0 = np.rint((Mi[4] * ys + Mi[5]) * AB_SCALE).astype(np.int64) + rd
    X = (X0[:, None] + adelta[None, :]) >> (AB_BITS - INTER_BITS)
    Y = (Y0[:, None] + bdelta[None, :]) >> (AB_BITS - INTER_BITS)
    sx = np.clip(X >> INTER_BITS, -32768, 32767)
    sy = np.clip(Y >> INTER_BITS, -32768, 32767)
    alpha = (Y & (TAB - 1)) * TAB + (X & (TAB - 1))
    w = _TABLE[alpha]                                              # [oh,ow,4]

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        v = img[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)].astype(np.int64)
        return np.where(ok[..., None], v, 0)

    acc = (tap(sy, sx) * w[..., 0:1] + tap(sy, sx + 1) * w[..., 1:2] +
           tap(sy + 1, sx) * w[..., 2:3] + tap(sy + 1, sx + 1) * w[..., 3:4])
    out = (acc + (1 << (COEF_BITS - 1))) >> COEF_BITS
    return np.clip(out, 0, 255).astype(np.uint8)


def crop_to_tensor(img_rgb_u8, bbox, scale=1.2, crop=224):
    """get_single_image_crop_demo + ToTensor: u8[H,W,3] RGB, bbox (cx,cy,w,h) -> f32[3,crop,crop] in [0,1]."""
    patch = warp_affine_u8(img_rgb_u8, affine_from_bbox(bbox, scale, crop), (crop, crop))
    return (patch.astype(np.float32) / np.float32(255)).transpose(2, 0, 1).copy()
