"""TEST INFRASTRUCTURE -- CPU oracle for the cube-extraction stage.

Restates, in numpy, what the reference computes at ``vad_datasets.py:70-93`` (``get_foreground``): for every box
``(x1, y1, x2, y2)`` take ``img[..., ceil(y1):ceil(y2), ceil(x1):ceil(x2)]`` and ``cv2.resize`` it to
``patch_size x patch_size`` with the default ``INTER_LINEAR`` interpolation; and at ``calc_optical_flow.py:46-59,82``
(whole-frame ``cv2.resize`` of uint8 frames to 512x384 and of the float32 flow field back to the frame size).

The arithmetic lives in a third-party dependency that is ABSENT from /root/reference and from this image: OpenCV
(``import cv2``; the reference pins no version -- README.md:10-16 lists only torch/mmdet/mmcv -- the 3.4/4.x ``resize``
C++ path is what is restated here, ``modules/imgproc/src/resize.cpp``):

  * coordinates:  ``scale = 1.0 / (dst / src)`` (double);  ``f = (float)((d + 0.5) * scale - 0.5)``;  ``s = floor(f)``;
    ``f -= s``.  Horizontally ``s < 0 -> (s, f) = (0, 0)`` and ``s >= src-1 -> (src-1, 0)``; vertically the two source
    rows are clamped to ``[0, src-1]`` and ``f`` is left alone.
  * uint8:  weights are ``short(rint(w * 2048))``;  the horizontal pass keeps ``S[s]*a0 + S[s+1]*a1`` as int32, the
    vertical pass is ``(((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2``.
  * float32:  ``r = S[s]*a0 + S[s+1]*a1`` then ``r0*b0 + r1*b1``, every product and sum rounded to float32 (no FMA).
  * an exact 2x2 decimation (src == 2*dst on both axes) is rerouted by cv2 to INTER_AREA:  uint8
    ``(s00 + s01 + s10 + s11 + 2) >> 2``, float32 ``(((s00 + s01) + s10) + s11) * 0.25f``.
  * ``dst size == src size`` is a plain copy.

PARITY UNPINNED: cv2 cannot be imported here and the reference holds no test vectors for this stage, so the restatement
is anchored only on the published algorithm above and on hand-computed known answers in tests/test_extract.py.  cv2
builds that dispatch to IPP / FMA-contracted SIMD can differ from this by 1 LSB (uint8) / 1 ulp (float32).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math

import numpy as np

_COEF_BITS = 11
_ONE = 1 << _COEF_BITS


def _axis(dst, src, horizontal):
    """-> (s0, s1, w0, w1 as float32) for one axis; see module docstring."""
    scale = 1.0 / (float(dst) / float(src))
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if horizontal:
        lo = s < 0
        f[lo], s[lo] = 0, 0
        hi = s >= src - 1
        f[hi], s[hi] = 0, src - 1
        s0, s1 = s, np.minimum(s + 1, src - 1)
    else:
        s0, s1 = np.clip(s, 0, src - 1), np.clip(s + 1, 0, src - 1)
    w0 = (np.float32(1.0) - f).astype(np.float32)
    return s0, s1, w0, f


def _fixed(w):
    return np.clip(np.rint(w * np.float32(_ONE)), -32768, 32767).astype(np.int32)


def resize_linear(src, dsize):
    """``cv2.resize(src, dsize)`` (default INTER_LINEAR) for HxW or HxWxC uint8 / float32 arrays; ``dsize=(w, h)``."""
    src = np.asarray(src)
    squeeze = src.ndim == 2
    if squeeze:
        src = src[:, :, None]
    H, W, _ = src.shape
    dw, dh = int(dsize[0]), int(dsize[1])
    if H <= 0 or W <= 0 or dw <= 0 or dh <= 0:
        raise ValueError('resize_linear: empty source or destination (cv2 asserts !ssize.empty())')
    if src.dtype == np.uint8:
        integer = True
    elif src.dtype == np.float32:
        integer = False
    else:
        raise TypeError('resize_linear restates the uint8 and float32 paths only, got %s' % src.dtype)
    if (dh, dw) == (H, W):
        out = src.copy()
    elif W == 2 * dw and H == 2 * dh:
        a, b, c, d = src[0::2, 0::2], src[0::2, 1::2], src[1::2, 0::2], src[1::2, 1::2]
        if integer:
            out = ((a.astype(np.int32) + b + c + d + 2) >> 2).astype(np.uint8)
        else:
            out = (((a + b) + c) + d) * np.float32(0.25)
    else:
        x0, x1, a0, a1 = _axis(dw, W, True)
        y0, y1, b0, b1 = _axis(dh, H, False)
        if integer:
            a0, a1, b0, b1 = _fixed(a0), _fixed(a1), _fixed(b0), _fixed(b1)
            s = src.astype(np.int32)
            rows = s[:, x0] * a0[None, :, None] + s[:, x1] * a1[None, :, None]          # [H, dw, C] int32
            r0, r1 = rows[y0], rows[y1]
            out = ((((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2)
            out = np.clip(out, 0, 255).astype(np.uint8)
        else:
            rows = (src[:, x0] * a0[None, :, None]).astype(np.float32) + (src[:, x1] * a1[None, :, None]).astype(np.float32)
            rows = rows.astype(np.float32)
            out = ((rows[y0] * b0[:, None, None]).astype(np.float32) + (rows[y1] * b1[:, None, None]).astype(np.float32))
            out = out.astype(np.float32)
    return out[:, :, 0] if squeeze else out


def box_to_crop(box, H, W):
    """``(x_min, y_min, x_max, y_max)`` of the slice ``img[:, ceil(y1):ceil(y2), ceil(x1):ceil(x2)]``
    (vad_datasets.py:74-76) after numpy's clipping of the slice ends to the array (negative starts are clamped to 0:
    the detector never emits them and numpy's wrap-around there is not behaviour anyone relies on)."""
    x_min, x_max = int(math.ceil(box[0])), int(math.ceil(box[2]))
    y_min, y_max = int(math.ceil(box[1])), int(math.ceil(box[3]))
    x_min, y_min = max(x_min, 0), max(y_min, 0)
    return x_min, y_min, min(x_max, W), min(y_max, H)


def get_foreground(img, bboxes, patch_size):
    """reference vad_datasets.py:70-93: ``img`` is ``[C,H,W]`` or ``[T,C,H,W]``; returns ``[n,C,P,P]`` / ``[n,T,C,P,P]``."""
    img = np.asarray(img)
    single = img.ndim == 3
    frames = img[None] if single else img
    H, W = frames.shape[2], frames.shape[3]
    out = []
    for b in bboxes:
        x0, y0, x1, y1 = box_to_crop(b, H, W)
        cube = []
        for t in range(frames.shape[0]):
            patch = np.ascontiguousarray(np.transpose(frames[t, :, y0:y1, x0:x1], [1, 2, 0]))
            r = resize_linear(patch, (patch_size, patch_size))
            cube.append(np.transpose(r, [2, 0, 1]))
        out.append(cube[0] if single else np.array(cube))
    return np.array(out)
