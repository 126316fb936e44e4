"""TEST INFRASTRUCTURE ONLY — CPU restatement of the image side of preprocess.  Nothing outside tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

What is restated, and how it is pinned:
  * LetterBox geometry + label update, Format (ultralytics/data/augment.py:1544-1633, 1926-2107) and the box container
    (ultralytics/utils/instance.py): PINNED — tests/golden/image.npz holds outputs of the reference's own classes run
    through oracle/gen_golden_image.py.
  * The pixels of cv2.resize(INTER_LINEAR) / cv2.copyMakeBorder: PARITY UNPINNED against a real cv2.  opencv-python is
    a reference dependency (requirements: opencv-python>=4.6.0) that is neither vendored in /root/reference nor
    installed in this image, so no cv2 output exists to compare with.  `cv2_resize_linear_u8` restates the published
    8-bit algorithm of OpenCV 4.x imgproc/src/resize.cpp:
        scale = 1 / (dsize / ssize)  (double);   f = (float)((d + 0.5) * scale - 0.5);  s = floor(f);  f -= s
        columns outside the image pin f = 0 on the border column; rows keep f and clip the row indices
        coefficients = round-half-even(float(1 - f) * 2048), round-half-even(f * 2048)          (INTER_RESIZE_COEF_BITS = 11)
        horizontal pass  h = S[x0] * a0 + S[x1] * a1                                            (int32, no shift)
        vertical pass    d = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2     (VResizeLinear, 8u)
        an exact 2x shrink in both directions is rerouted to INTER_AREA: (a + b + c + d + 2) >> 2
    The generator feeds this function to the reference's LetterBox in place of the missing cv2 call, so the golden
    pixels are "reference control flow + restated cv2", and say so.
  * F.interpolate(mode="bilinear", align_corners=False) of the multi_scale branch (models/yolo/detect/train.py:60-73):
    the oracle IS torch's CPU kernel (floating point, tolerance 1e-6 stated in the test).
"""
from __future__ import annotations

import numpy as np


def _coef(dsize, ssize, pin):
    inv = np.float64(dsize) / np.float64(ssize)
    scale = np.float64(1.0) / inv
    d = np.arange(dsize, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if pin:
        lo, hi = s < 0, s >= ssize - 1
        f = np.where(lo | hi, np.float32(0), f)
        s = np.where(lo, 0, np.where(hi, ssize - 1, s))
    s0 = np.clip(s, 0, ssize - 1)
    s1 = np.clip(s + 1, 0, ssize - 1)
    a0 = np.rint((np.float32(1) - f).astype(np.float32) * np.float32(2048)).astype(np.int64)
    a1 = np.rint(f * np.float32(2048)).astype(np.int64)
    return s0, s1, a0, a1


def cv2_resize_linear_u8(img, dsize):
    """cv2.resize(img, dsize=(w, h), interpolation=cv2.INTER_LINEAR) for an (H, W, C) uint8 image (see module header)."""
    assert img.dtype == np.uint8 and img.ndim == 3
    dw, dh = int(dsize[0]), int(dsize[1])
    sh, sw = img.shape[:2]
    if (dw, dh) == (sw, sh):
        return img.copy()
    if sw == 2 * dw and sh == 2 * dh:
        a = img.astype(np.int64)
        return ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    x0, x1, ax0, ax1 = _coef(dw, sw, True)
    y0, y1, by0, by1 = _coef(dh, sh, False)
    src = img.astype(np.int64)
    hor = src[:, x0, :] * ax0[None, :, None] + src[:, x1, :] * ax1[None, :, None]          # (sh, dw, C)
    h0, h1 = hor[y0], hor[y1]                                                              # (dh, dw, C)
    out = (((by0[:, None, None] * (h0 >> 4)) >> 16) + ((by1[:, None, None] * (h1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def cv2_copy_make_border(img, top, bottom, left, right, value=114):
    """cv2.copyMakeBorder(..., cv2.BORDER_CONSTANT, value=(v, v, v))."""
    h, w, c = img.shape
    out = np.full((h + top + bottom, w + left + right, c), value, dtype=img.dtype)
    out[top:top + h, left:left + w] = img
    return out


def letterbox(img, new_shape=(640, 640), auto=False, scaleFill=False, scaleup=True, center=True, stride=32):
    """LetterBox.__call__(image=img) (data/augment.py:1544-1591): returns (image, ratio, (left, top))."""
    shape = img.shape[:2]
    if isinstance(new_shape, int):
        new_shape = (new_shape, new_shape)
    r = min(new_shape[0] / shape[0], new_shape[1] / shape[1])
    if not scaleup:
        r = min(r, 1.0)
    ratio = r, r
    new_unpad = int(round(shape[1] * r)), int(round(shape[0] * r))
    dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]
    if auto:
        dw, dh = np.mod(dw, stride), np.mod(dh, stride)
    elif scaleFill:
        dw, dh = 0.0, 0.0
        new_unpad = (new_shape[1], new_shape[0])
        ratio = new_shape[1] / shape[1], new_shape[0] / shape[0]
    if center:
        dw /= 2
        dh /= 2
    if shape[::-1] != new_unpad:
        img = cv2_resize_linear_u8(img, new_unpad)
    top, bottom = int(round(dh - 0.1)) if center else 0, int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)) if center else 0, int(round(dw + 0.1))
    return cv2_copy_make_border(img, top, bottom, left, right, 114), ratio, (left, top)


def predictor_preprocess(images, new_shape=(640, 640), stride=32, half=False):
    """BasePredictor.pre_transform + preprocess (engine/predictor.py:118-163) for a list of HWC BGR uint8 images on the
    `pt` path: LetterBox(auto = all shapes equal), stack, BGR->RGB, BHWC->BCHW, float, /255."""
    import torch
    same = len({x.shape for x in images}) == 1
    lb = [letterbox(x, new_shape, auto=same, stride=stride)[0] for x in images]
    im = np.ascontiguousarray(np.stack(lb)[..., ::-1].transpose((0, 3, 1, 2)))
    t = torch.from_numpy(im)
    t = t.half() if half else t.float()
    t /= 255
    return t


def preprocess_batch_multi_scale(img_u8, sz, stride=32):
    """The multi_scale branch of DetectionTrainer.preprocess_batch (models/yolo/detect/train.py:59-73) for a drawn
    size `sz` (the draw itself — random.randrange(int(imgsz*0.5), int(imgsz*1.5 + stride)) // stride * stride — is the
    caller's; this is the arithmetic after it)."""
    import math
    import torch
    imgs = torch.from_numpy(img_u8).float() / 255
    sf = sz / max(imgs.shape[2:])
    if sf != 1:
        ns = [math.ceil(x * sf / stride) * stride for x in imgs.shape[2:]]
        imgs = torch.nn.functional.interpolate(imgs, size=ns, mode="bilinear", align_corners=False)
    return imgs


# ------------------------------------------------------------------------------------------------ augmentation pixels
# cv2.warpAffine / cv2.cvtColor(BGR2HSV, HSV2BGR) / cv2.LUT / cv2.getRotationMatrix2D as called by RandomPerspective
# (data/augment.py:1000-1078) and RandomHSV (:1303-1390).  Same status as the resize: restated from OpenCV 4.x
# (imgproc/src/imgwarp.cpp classic fixed-point path, color_hsv.simd.hpp scalar path); PARITY UNPINNED against a real cv2.
def cv2_get_rotation_matrix_2d(center, angle, scale):
    """imgwarp.cpp getRotationMatrix2D: angle in degrees, positive = counter-clockwise; 2x3 float64."""
    import math
    a = angle * math.pi / 180
    alpha, beta = math.cos(a) * scale, math.sin(a) * scale
    cx, cy = center
    return np.array([[alpha, beta, (1 - alpha) * cx - beta * cy], [-beta, alpha, beta * cx + (1 - alpha) * cy]], np.float64)


def invert_affine(M):
    """The in-place inversion at the top of cv::warpAffine (float64, this exact operation order)."""
    m = [float(v) for v in np.asarray(M, np.float64).reshape(6)]
    D = m[0] * m[4] - m[1] * m[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = m[4] * D, m[0] * D
    m[0] = A11
    m[1] *= -D
    m[3] *= -D
    m[4] = A22
    b1 = -m[0] * m[2] - m[1] * m[5]
    b2 = -m[3] * m[2] - m[4] * m[5]
    m[2], m[5] = b1, b2
    return m


def warp_coords(minv, W, H):
    """WarpAffineInvoker: 10-bit fixed-point source coordinates with 5 fractional bits kept for the interpolation table.
    Returns (sx, sy, fx, fy) int arrays of shape (H, W)."""
    x = np.arange(W, dtype=np.float64)
    y = np.arange(H, dtype=np.float64)
    adelta = np.rint(minv[0] * x * 1024).astype(np.int64)
    bdelta = np.rint(minv[3] * x * 1024).astype(np.int64)
    X0 = np.rint((minv[1] * y + minv[2]) * 1024).astype(np.int64) + 16
    Y0 = np.rint((minv[4] * y + minv[5]) * 1024).astype(np.int64) + 16
    X = (X0[:, None] + adelta[None, :]) >> 5
    Y = (Y0[:, None] + bdelta[None, :]) >> 5
    sx = np.clip(X >> 5, -32768, 32767)
    sy = np.clip(Y >> 5, -32768, 32767)
    return sx, sy, X & 31, Y & 31


def cv2_warp_affine_u8(img, M, dsize, border_value=114):
    """cv2.warpAffine(img, M (2x3), dsize=(w, h), borderValue=(v, v, v)) — INTER_LINEAR, BORDER_CONSTANT.
    Every tap outside the source counts as border_value; weights are (32-fy)(32-fx)*32 ... fy*fx*32 of 32768."""
    W, H = int(dsize[0]), int(dsize[1])
    sh, sw = img.shape[:2]
    sx, sy, fx, fy = warp_coords(invert_affine(M), W, H)
    src = img.astype(np.int64)

    def tap(xx, yy):
        inside = (xx >= 0) & (xx < sw) & (yy >= 0) & (yy < sh)
        v = src[np.clip(yy, 0, sh - 1), np.clip(xx, 0, sw - 1)]
        return np.where(inside[..., None], v, border_value)

    w00 = ((32 - fy) * (32 - fx) * 32)[..., None]
    w01 = ((32 - fy) * fx * 32)[..., None]
    w10 = (fy * (32 - fx) * 32)[..., None]
    w11 = (fy * fx * 32)[..., None]
    out = (tap(sx, sy) * w00 + tap(sx + 1, sy) * w01 + tap(sx, sy + 1) * w10 + tap(sx + 1, sy + 1) * w11 + 16384) >> 15
    return np.clip(out, 0, 255).astype(np.uint8)


def _hsv_tables():
    i = np.arange(1, 256, dtype=np.float64)
    hdiv = np.zeros(256, np.int64)
    sdiv = np.zeros(256, np.int64)
    hdiv[1:] = np.rint((180 << 12) / (6.0 * i)).astype(np.int64)
    sdiv[1:] = np.rint((255 << 12) / (1.0 * i)).astype(np.int64)
    return hdiv, sdiv


def cv2_bgr2hsv_u8(img):
    """cv2.cvtColor(img, cv2.COLOR_BGR2HSV) for uint8 (H in [0, 180)): RGB2HSV_b's integer table arithmetic."""
    hdiv, sdiv = _hsv_tables()
    b, g, r = (img[..., k].astype(np.int64) for k in range(3))
    v = np.maximum(np.maximum(b, g), r)
    vmin = np.minimum(np.minimum(b, g), r)
    diff = v - vmin
    s = (diff * sdiv[v] + (1 << 11)) >> 12
    h = np.where(v == r, g - b, np.where(v == g, b - r + 2 * diff, r - g + 4 * diff))
    h = (h * hdiv[diff] + (1 << 11)) >> 12
    h = h + np.where(h < 0, 180, 0)
    return np.stack((np.clip(h, 0, 255), s, v), -1).astype(np.uint8)


def cv2_hsv2bgr_u8(hsv):
    """cv2.cvtColor(hsv, cv2.COLOR_HSV2BGR) for uint8: HSV2RGB_b = float32 sector arithmetic, then round-half-even."""
    f32 = np.float32
    h = hsv[..., 0].astype(f32)
    s = hsv[..., 1].astype(f32) * f32(1.0 / 255.0)
    v = hsv[..., 2].astype(f32) * f32(1.0 / 255.0)
    hh = h * f32(6.0 / 180.0)
    hh = np.fmod(hh, f32(6.0)).astype(f32)
    sector = np.floor(hh).astype(np.int64)
    frac = (hh - sector.astype(f32)).astype(f32)
    bad = (sector < 0) | (sector >= 6)
    sector = np.where(bad, 0, sector)
    frac = np.where(bad, f32(0), frac)
    one = f32(1)
    tab = np.stack((v, v * (one - s), v * (one - s * frac), v * (one - s * (one - frac))), -1).astype(f32)
    sector_data = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])
    idx = sector_data[sector]                                                    # (..., 3) -> tab index of b, g, r
    bgr = np.take_along_axis(tab, idx, -1)
    bgr = np.where((hsv[..., 1] == 0)[..., None], v[..., None], bgr).astype(f32)
    return np.clip(np.rint(bgr * f32(255.0)), 0, 255).astype(np.uint8)


def hsv_luts(r):
    """RandomHSV.__call__ (data/augment.py:1380-1386): the three 256-entry tables for gains r = (h, s, v)."""
    x = np.arange(0, 256, dtype=np.asarray(r).dtype)
    return (((x * r[0]) % 180).astype(np.uint8), np.clip(x * r[1], 0, 255).astype(np.uint8), np.clip(x * r[2], 0, 255).astype(np.uint8))


def random_hsv(img, r):
    """The pixel half of RandomHSV for drawn gains r: BGR -> HSV -> three LUTs -> BGR."""
    lh, ls, lv = hsv_luts(r)
    hsv = cv2_bgr2hsv_u8(img)
    return cv2_hsv2bgr_u8(np.stack((lh[hsv[..., 0]], ls[hsv[..., 1]], lv[hsv[..., 2]]), -1))
