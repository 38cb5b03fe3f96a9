"""The two OpenCV steps of the reference's image loading, restated (cv2 is not importable here).

Every prepared ZJU-MoCap directory carries per-frame ``distortions`` (tools/prepare_zju_mocap/prepare_dataset.py:172-176)
and every 387 / wild yaml sets ``resize_img_scale: 0.5`` (configs/human_nerf/zju_mocap/387/adventure.yaml:37), so
``Dataset.load_image`` (core/data/human_nerf/train.py:351-417, freeview.py:137-166) runs, per frame:

    orig, mask = cv2.undistort(uint8 image / mask, K, D)                  train.py:366-371
    img  = mask/255 * orig + (1 - mask/255) * bg                          float64
    img  = cv2.resize(img,  None, fx=s, fy=s, interpolation=INTER_LANCZOS4)
    mask = cv2.resize(mask, None, fx=s, fy=s, interpolation=INTER_LINEAR)  train.py:408-417

This module is the host (numpy) statement of those two functions as OpenCV 4.x defines them; the device statement
is csrc/hnrf_image.hip (ops.undistort_image / ops.composite_windows / ops.resize_mask), written to the same
operation order so that the two routes agree (tests/test_gpu_image.py: undistortion bit for bit, resize to the last
float32 bit).  PARITY UNPINNED against OpenCV's binaries -- what is restated, and what can differ:

undistort (imgproc/src/undistort.dispatch.cpp, undistort.simd.hpp, remap):
  * stripes of ``min(max(1, 4096 // W), H)`` rows; per stripe the inverse of K with its principal point shifted to
    the stripe, the Brown-Conrady forward model on the normalised coordinates (k1, k2, p1, p2, k3; the rational /
    thin-prism / tilt terms are zero for a 5-vector), ``u = fx*xd + u0``;
  * the CV_16SC2 fixed-point map: ``iu = cvRound(u * 32)``, integer part ``iu >> 5``, 5-bit fractions;
  * bilinear taps with the 15-bit integer weight table of remap (for 1/32 fractions every weight is exact), result
    ``(sum + 2^14) >> 15``, BORDER_CONSTANT 0 for taps outside the image.
  OpenCV's AVX2 line routine forms the per-pixel normalised coordinate as ``(_x + 4k ir0) + ir0 * {0..3}`` with FMA,
  its scalar tail as a running sum; here it is ``j * ir0 + _x`` -- last-ulp differences of ``u`` that move a pixel
  only when ``32 u`` sits within 1e-11 of a half-integer.  With D = 0 the map is the integer grid and the output is
  the input, bit for bit.

resize (imgproc/src/resize.cpp, the generic CV_64F path: float weights, double accumulation, horizontal pass then
vertical pass, each a left-to-right sum; taps clamped to the border):
  * destination size ``cvRound(src * scale)`` (half to even), source coordinate ``(dst + 0.5) / scale - 0.5``
    evaluated in double and rounded to float before the floor / fraction split;
  * INTER_LANCZOS4: 8 taps from ``interpolateLanczos4`` (the sine-product form of sinc(x) sinc(x/4), float
    normalisation);  INTER_LINEAR: ``(1 - f, f)`` with the edge clamps of the coefficient loop -- and, exactly as
    hal::resize does, the 2 x 2 box mean (INTER_AREA's fast path) when the scale is exactly 1/2.
  OpenCV builds may contract a*b+c into FMA; this statement does not.
"""
import math

import numpy as np

INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS


# ------------------------------------------------------------------------------------------------ undistort
def distortion_vector(D):
    """(k1, k2, p1, p2, k3) as float64; longer OpenCV vectors are accepted only if their extra terms are zero."""
    d = np.zeros(5, dtype=np.float64) if D is None else np.asarray(D, dtype=np.float64).reshape(-1)
    if d.size > 5 and np.any(d[5:] != 0):
        raise NotImplementedError('distortion model with %d coefficients (rational / thin prism / tilt terms)' % d.size)
    out = np.zeros(5, dtype=np.float64)
    out[:min(5, d.size)] = d[:5]
    return out


def undistort_stripes(K, H, W):
    """cv2.undistort's stripe height and, per stripe, the row-major inverse of K with cy moved to the stripe's first
    row (undistort.dispatch.cpp: ``Ar(1,2) = v0 - y``; initUndistortRectifyMap: ``ir = (Ar R)^-1``, R = I)."""
    A = np.asarray(K, dtype=np.float64)[:3, :3]
    rows = min(max(1, (1 << 12) // max(int(W), 1)), int(H))
    n = -(-int(H) // rows)
    Ar = np.repeat(A[None], n, axis=0)
    Ar[:, 1, 2] = A[1, 2] - np.arange(n, dtype=np.float64) * rows
    return rows, np.linalg.inv(Ar).reshape(n, 9)


def undistort_maps(K, D, H, W):
    """The fixed-point map cv2.undistort builds: ix, iy (int32, top-left tap) and fx, fy (0..31) per output pixel."""
    A = np.asarray(K, dtype=np.float64)[:3, :3]
    k1, k2, p1, p2, k3 = distortion_vector(D)
    fx_, fy_, u0, v0 = A[0, 0], A[1, 1], A[0, 2], A[1, 2]
    rows, ir = undistort_stripes(A, H, W)
    y = np.arange(H)
    s, i = y // rows, (y % rows).astype(np.float64)
    irs = ir[s]                                                          # [H, 9]
    _x0 = (i * irs[:, 1] + irs[:, 2])[:, None]
    _y0 = (i * irs[:, 4] + irs[:, 5])[:, None]
    _w0 = (i * irs[:, 7] + irs[:, 8])[:, None]
    j = np.arange(W, dtype=np.float64)[None, :]
    w = 1.0 / (j * irs[:, 6:7] + _w0)
    x = (j * irs[:, 0:1] + _x0) * w
    yy = (j * irs[:, 3:4] + _y0) * w
    x2, y2 = x * x, yy * yy
    r2, _2xy = x2 + y2, 2 * x * yy
    kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / 1.0
    xd = x * kr + p1 * _2xy + p2 * (r2 + 2 * x2)
    yd = yy * kr + p1 * (r2 + 2 * y2) + p2 * _2xy
    u, v = fx_ * xd + u0, fy_ * yd + v0
    lim = float(2 ** 31 - 1)
    iu = np.rint(np.clip(u * INTER_TAB_SIZE, -lim - 1, lim)).astype(np.int64)   # saturate_cast<int>(double) = cvRound
    iv = np.rint(np.clip(v * INTER_TAB_SIZE, -lim - 1, lim)).astype(np.int64)
    return ((iu >> INTER_BITS).astype(np.int32), (iv >> INTER_BITS).astype(np.int32),
            (iu & (INTER_TAB_SIZE - 1)).astype(np.int32), (iv & (INTER_TAB_SIZE - 1)).astype(np.int32))


def remap_bilinear_u8(src, ix, iy, fx, fy):
    """remap(..., INTER_LINEAR, BORDER_CONSTANT 0) of a uint8 image through a fixed-point map.  The 15-bit table
    entries for 1/32 fractions are ``32 (32-fx)(32-fy)`` etc. exactly, so ``(sum w v + 2^14) >> 15`` is
    ``(sum_1024 + 512) >> 10``."""
    src = np.asarray(src)
    assert src.dtype == np.uint8
    squeeze = src.ndim == 2
    if squeeze:
        src = src[:, :, None]
    H, W = src.shape[:2]
    acc = np.zeros(ix.shape + (src.shape[2],), dtype=np.int32)
    for dy, dx, wgt in ((0, 0, (INTER_TAB_SIZE - fx) * (INTER_TAB_SIZE - fy)), (0, 1, fx * (INTER_TAB_SIZE - fy)),
                        (1, 0, (INTER_TAB_SIZE - fx) * fy), (1, 1, fx * fy)):
        xx, yy = ix + dx, iy + dy
        ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
        v = src[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)].astype(np.int32)
        acc += np.where(ok[..., None], v, 0) * wgt[..., None]
    out = ((acc + 512) >> 10).astype(np.uint8)
    return out[:, :, 0] if squeeze else out


_MAP_CACHE = {}


def undistort_u8(img, K, D):
    """cv2.undistort(img, K, D) for a uint8 image (H, W[, C]).  Maps are kept per camera (a monocular subject has one)."""
    img = np.asarray(img)
    H, W = img.shape[:2]
    key = (np.asarray(K, np.float64)[:3, :3].tobytes(), distortion_vector(D).tobytes(), H, W)
    maps = _MAP_CACHE.get(key)
    if maps is None:
        if len(_MAP_CACHE) >= 8:
            _MAP_CACHE.pop(next(iter(_MAP_CACHE)))
        maps = _MAP_CACHE[key] = undistort_maps(K, D, H, W)
    return remap_bilinear_u8(img, *maps)


# ------------------------------------------------------------------------------------------------ resize
def cv_round(x):
    """cvRound / saturate_cast<int>(double): nearest, ties to even."""
    return int(np.rint(x))


def resized_size(h, w, scale):
    """dsize of cv2.resize(src, None, fx=scale, fy=scale): (rows, cols)."""
    return cv_round(h * float(scale)), cv_round(w * float(scale))


_F32 = np.float32
_S45 = 0.70710678118654752440084436210485
_CS = ((1, 0), (-_S45, -_S45), (0, 1), (_S45, -_S45), (-1, 0), (_S45, _S45), (0, -1), (-_S45, _S45))


def lanczos4_coeffs(x):
    """interpolateLanczos4(float x, float* coeffs) of OpenCV 4.x: 8 float weights for the fraction ``x`` in [0, 1)."""
    x = _F32(x)
    co = np.zeros(8, dtype=np.float32)
    y0 = -float(_F32(x + _F32(3))) * math.pi * 0.25
    s0, c0 = math.sin(y0), math.cos(y0)
    total = _F32(0)
    for i in range(8):
        y0_ = _F32(_F32(x + _F32(3)) - _F32(i))
        if abs(float(y0_)) >= 1e-6:
            y = -float(y0_) * math.pi * 0.25
            co[i] = _F32((_CS[i][0] * s0 + _CS[i][1] * c0) / (y * y))
        else:
            co[i] = _F32(1e30)
        total = _F32(total + co[i])
    inv = _F32(_F32(1) / total)
    return (co * inv).astype(np.float32)


def resize_tables(src_len, dst_len, scale_inv, kind):
    """The coefficient loop of hal::resize for one axis: ``ofs`` (int32, index of the tap with weight index
    ksize/2 - 1) and ``w`` (float32, [dst_len, ksize]); kind 'lanczos4' | 'linear'.  ``scale_inv`` = 1 / fx."""
    ksize = 8 if kind == 'lanczos4' else 2
    ofs = np.zeros(dst_len, dtype=np.int32)
    w = np.zeros((dst_len, ksize), dtype=np.float32)
    cache = {}
    for d in range(dst_len):
        f = _F32((d + 0.5) * scale_inv - 0.5)
        s = int(math.floor(float(f)))
        f = _F32(f - _F32(s))
        if kind == 'linear':
            if s < 0:
                f, s = _F32(0), 0
            if s >= src_len - 1:
                f, s = _F32(0), src_len - 1
            w[d] = (_F32(1) - f, f)
        else:
            key = float(f)
            if key not in cache:
                cache[key] = lanczos4_coeffs(f)
            w[d] = cache[key]
        ofs[d] = s
    return ofs, w


def _axis_pass(a, ofs, w, axis):
    """Left-to-right sum over the taps of one axis, taps clamped to the border (HResize* / the row clipping of
    resizeGeneric_Invoker), float64 accumulation with float32 weights."""
    ksize = w.shape[1]
    n = a.shape[axis]
    shape = [1] * a.ndim
    shape[axis] = -1
    acc = None
    for k in range(ksize):
        idx = np.clip(ofs + (k - ksize // 2 + 1), 0, n - 1)
        term = np.take(a, idx, axis=axis) * w[:, k].astype(np.float64).reshape(shape)
        acc = term if acc is None else acc + term
    return acc


def is_half_scale(scale):
    inv = 1.0 / float(scale)
    return abs(inv - 2.0) < np.finfo(np.float64).eps


def resize_f64(img, scale, kind):
    """cv2.resize(img, None, fx=scale, fy=scale, interpolation=INTER_LANCZOS4 | INTER_LINEAR) of a float64 image
    (H, W[, C]).  Returns float64 (rows, cols[, C])."""
    img = np.asarray(img, dtype=np.float64)
    H, W = img.shape[:2]
    Hd, Wd = resized_size(H, W, scale)
    inv = 1.0 / float(scale)
    if kind == 'linear' and is_half_scale(scale) and 2 * Hd <= H and 2 * Wd <= W:   # hal::resize: INTER_LINEAR at 1/2 IS the box mean
        a = img[:2 * Hd, :2 * Wd]
        return (((a[0::2, 0::2] + a[0::2, 1::2]) + a[1::2, 0::2]) + a[1::2, 1::2]) * 0.25
    xo, xw = resize_tables(W, Wd, inv, kind)
    yo, yw = resize_tables(H, Hd, inv, kind)
    return _axis_pass(_axis_pass(img, xo, xw, axis=1), yo, yw, axis=0)


# ------------------------------------------------------------------------------------------------ the loading step
def composite_over(orig_u8, alpha_u8, bg_color):
    """train.py:359-406: ``alpha/255 * orig + (1 - alpha/255) * bg`` in float64 (bg is the dataset's float32 colour)."""
    a = alpha_u8 / 255.
    return a * orig_u8 + (1.0 - a) * np.asarray(bg_color)[None, None, :], a


def load_step(orig_u8, alpha_u8, bg_color, K=None, D=None, scale=1.0):
    """Everything Dataset.load_image does after decoding the two PNGs: -> img (float64, 0..255), alpha (float64)."""
    if D is not None:                                                    # the reference undistorts whenever the key exists
        orig_u8, alpha_u8 = undistort_u8(orig_u8, K, D), undistort_u8(alpha_u8, K, D)
    img, a = composite_over(orig_u8, alpha_u8, bg_color)
    if float(scale) != 1.0:
        img, a = resize_f64(img, scale, 'lanczos4'), resize_f64(a, scale, 'linear')
    return img, a
