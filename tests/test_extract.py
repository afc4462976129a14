"""Cube extraction and score aggregation (8 f-2).

CPU part: the numpy oracle of cv2.resize(INTER_LINEAR) against hand-computed known answers and against an independent
bilinear implementation (torch interpolate, align_corners=False -- same half-pixel mapping, float arithmetic).
GPU part: vv_crop_resize / vv_frame_scores / vv_roc_auc_counts through the C ABI against the oracle, bit-exact.
"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import resize_oracle as R  # noqa: E402


def _torch_bilinear(img, dsize):
    t = torch.from_numpy(img.astype(np.float64)).permute(2, 0, 1)[None]
    o = torch.nn.functional.interpolate(t, size=(dsize[1], dsize[0]), mode='bilinear', align_corners=False)
    return o[0].permute(1, 2, 0).numpy()


def test_oracle_known_answers():
    # [[0,100]] -> width 4: weights (1,0) (.75,.25) (.25,.75) (0,1) in 11-bit fixed point -> exactly 0,25,75,100
    src = np.array([[0, 100], [0, 100]], np.uint8)
    out = R.resize_linear(src, (4, 2))
    assert out.tolist() == [[0, 25, 75, 100], [0, 25, 75, 100]]
    # vertical: rows 0 / 200, 2 -> 4 rows
    out = R.resize_linear(np.array([[0, 0], [200, 200]], np.uint8), (2, 4))
    assert out[:, 0].tolist() == [0, 50, 150, 200]
    # float path, same weights
    out = R.resize_linear(np.array([[0, 100], [0, 100]], np.float32), (4, 2))
    assert out.dtype == np.float32 and out.tolist() == [[0, 25, 75, 100], [0, 25, 75, 100]]
    # same size = copy; exact 2x decimation = rounded 2x2 mean
    a = np.arange(48, dtype=np.uint8).reshape(4, 4, 3)
    assert np.array_equal(R.resize_linear(a, (4, 4)), a)
    d = R.resize_linear(a, (2, 2))
    exp = (a.astype(int).reshape(2, 2, 2, 2, 3).sum(axis=(1, 3)) + 2) >> 2
    assert np.array_equal(d, exp)
    # 3:1 downscale samples the centre pixel (f = 0 exactly): picks elements 1, 4, 7
    b = np.arange(9, dtype=np.uint8).reshape(1, 9) * 10
    assert R.resize_linear(np.repeat(b, 3, 0), (3, 1)).tolist() == [[10, 40, 70]]


def test_oracle_against_independent_bilinear():
    rng = np.random.default_rng(0)
    for (h, w, c, dw, dh) in [(37, 53, 3, 32, 32), (11, 7, 2, 32, 32), (240, 360, 3, 512, 384), (96, 128, 2, 90, 60),
                              (1, 1, 3, 32, 32), (5, 64, 1, 32, 32)]:
        img8 = rng.integers(0, 256, (h, w, c), dtype=np.uint8)
        imgf = (rng.standard_normal((h, w, c)) * 3).astype(np.float32)
        ref8, reff = _torch_bilinear(img8, (dw, dh)), _torch_bilinear(imgf, (dw, dh))
        o8, of = R.resize_linear(img8, (dw, dh)), R.resize_linear(imgf, (dw, dh))
        assert o8.shape == (dh, dw, c) and o8.dtype == np.uint8
        assert np.abs(o8.astype(np.float64) - ref8).max() <= 1.0, (h, w, dw, dh)      # fixed point: within 1 LSB
        assert np.abs(o8.astype(np.float64) - ref8).mean() < 0.3
        # cv2 rounds the source coordinate to float32 before taking its fraction (ulp 8e-6 at x ~ 100) and the values
        # span ~ +-12, so a float64 evaluation differs by up to ~1e-4
        assert np.abs(of.astype(np.float64) - reff).max() < 2e-4, (h, w, dw, dh)


def test_oracle_get_foreground_shapes():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (5, 3, 60, 80), dtype=np.uint8)
    boxes = np.array([[3.2, 4.1, 40.7, 50.0, 0.9], [10, 10, 74, 58.5, 0.8], [70.5, 2.0, 95.0, 70.0, 0.5]])
    out = R.get_foreground(img, boxes, 32)
    assert out.shape == (3, 5, 3, 32, 32) and out.dtype == np.uint8
    one = R.get_foreground(img[2], boxes, 32)
    assert one.shape == (3, 3, 32, 32) and np.array_equal(one, out[:, 2])
    # box 1 is exactly 64 x 48 -> general path; crop [10:59,10:74]; box 2 is clipped at the frame border
    assert R.box_to_crop(boxes[2], 60, 80) == (71, 2, 80, 60)


def test_frame_roc_auc_matches_sklearn():
    from sklearn.metrics import roc_auc_score, roc_curve, auc
    from utils import frame_roc_auc
    rng = np.random.default_rng(2)
    for n in (17, 400):
        s = np.round(rng.standard_normal(n), 1)          # many ties
        y = rng.random(n) < 0.3
        fpr, tpr, _ = roc_curve(y.astype(float), s)
        assert abs(frame_roc_auc(s, y) - auc(fpr, tpr)) < 1e-12
        assert abs(frame_roc_auc(s, y) - roc_auc_score(y, s)) < 1e-12


# ------------------------------------------------------------------------------------------------------------- GPU
def _rand_boxes(rng, n, H, W):
    x0 = rng.uniform(0, W - 2, n)
    y0 = rng.uniform(0, H - 2, n)
    bw = rng.uniform(1.0, W * 0.6, n)
    bh = rng.uniform(1.0, H * 0.6, n)
    b = np.stack([x0, y0, x0 + bw, y0 + bh, rng.random(n)], 1)
    b[0, :4] = (4, 6, 4 + 64, 6 + 64)            # exact 2x -> area path
    b[1, :4] = (10, 12, 10 + 32, 12 + 32)        # same size -> copy
    b[2, :4] = (W - 7.5, H - 3.2, W + 30, H + 30)  # clipped by the frame border
    b[3, :4] = (5.0, 5.0, 5.5, 5.5)              # ceil -> 1x1 crop
    b[4, :4] = (0, 0, W, H)
    return b


@pytest.mark.gpu
@pytest.mark.parametrize('dtype,C', [(np.uint8, 3), (np.float32, 2), (np.uint8, 1), (np.float32, 3)])
def test_crop_resize_bit_exact(dtype, C):
    from vec_vad_amd import extract
    rng = np.random.default_rng(3)
    T, H, W = 5, 120, 180
    if dtype == np.uint8:
        fr = rng.integers(0, 256, (T, H, W, C), dtype=np.uint8)
    else:
        fr = (rng.standard_normal((T, H, W, C)) * 4).astype(np.float32)
    boxes = _rand_boxes(rng, 40, H, W)
    crops = extract.boxes_to_crops(boxes, H, W)
    for i, b in enumerate(boxes):
        assert tuple(crops[i]) == R.box_to_crop(b, H, W)
    out = extract.crop_resize(torch.from_numpy(fr).cuda(), crops, 32, 32).cpu().numpy()
    assert out.shape == (40, T, 32, 32, C)
    for i, (x0, y0, x1, y1) in enumerate(crops):
        for t in range(T):
            ref = R.resize_linear(np.ascontiguousarray(fr[t, y0:y1, x0:x1]), (32, 32))
            assert np.array_equal(out[i, t], ref), (i, t, crops[i])
    # drop-in get_foreground ([T,C,H,W] in, [n,T,C,P,P] out), 3-d and 4-d forms
    chw = np.ascontiguousarray(np.transpose(fr, [0, 3, 1, 2]))
    got = extract.get_foreground(chw, boxes, 32)
    assert np.array_equal(got, R.get_foreground(chw, boxes, 32))
    assert np.array_equal(extract.get_foreground(chw[1], boxes, 32), got[:, 1])
    with pytest.raises(ValueError):
        extract.get_foreground(chw, np.array([[W + 5.0, 3, W + 9.0, 9, 1.0]]), 32)


@pytest.mark.gpu
def test_whole_frame_resize_like_calc_optical_flow():
    from vec_vad_amd import extract
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (240, 360, 3), dtype=np.uint8)
    assert np.array_equal(extract.resize(img, (512, 384)), R.resize_linear(img, (512, 384)))
    gray = img[:, :, 0]
    assert np.array_equal(extract.resize(gray, (512, 384)), R.resize_linear(gray, (512, 384)))
    flow = (rng.standard_normal((384, 512, 2)) * 5).astype(np.float32)
    assert np.array_equal(extract.resize(flow, (360, 240)), R.resize_linear(flow, (360, 240)))
    big = rng.integers(0, 256, (768, 1024, 3), dtype=np.uint8)       # exact 2x -> area
    assert np.array_equal(extract.resize(big, (512, 384)), R.resize_linear(big, (512, 384)))


@pytest.mark.gpu
def test_frame_scores_and_auc_on_device():
    from sklearn.metrics import roc_auc_score
    from vec_vad_amd import scoring
    import test as T                         # repo-root test.py (paint_frame = the reference's mask arithmetic)
    rng = np.random.default_rng(5)
    h, w, F = 60, 90, 37
    counts = rng.integers(0, 6, F)
    counts[3] = 0
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    n = int(off[-1])
    raw = rng.random(n).astype(np.float32) * 50
    of = rng.random(n).astype(np.float32) * 9
    stats = np.array([[20.0, 7.5, 4.0, 1.25], [25.0, 3.0, 5.0, 2.0]])
    cs = rng.integers(-1, 2, n).astype(np.int32)
    cs[:5] = [0, 1, 0, 1, 0]
    boxes = np.stack([rng.uniform(0, w - 5, n), rng.uniform(0, h - 5, n), np.zeros(n), np.zeros(n)], 1)
    boxes[:, 2] = boxes[:, 0] + rng.uniform(0.5, 20, n)
    boxes[:, 3] = boxes[:, 1] + rng.uniform(0.5, 20, n)
    boxes[1] = (3.2, 3.2, 3.9, 9.0)          # ceil(x1) == ceil(x2): paints nothing
    paints = scoring.box_paints(boxes, h, w)
    assert paints[1] == 0 and paints.sum() >= n - 3
    for use_flow in (True, False):
        got = scoring.frame_scores(torch.from_numpy(raw).cuda(), torch.from_numpy(of).cuda() if use_flow else None,
                                   off, cs, stats, paints, 0.3, 1.0).cpu().numpy()
        for f in range(F):
            sl = slice(off[f], off[f + 1])
            sc = np.empty(counts[f])
            for k, m in enumerate(range(off[f], off[f + 1])):
                if cs[m] < 0:
                    sc[k] = T.BIG
                else:
                    st = stats[cs[m]]
                    sc[k] = 0.3 * ((raw[m] - st[0]) / st[1])
                    if use_flow:
                        sc[k] = sc[k] + 1.0 * ((of[m] - st[2]) / st[3])
            ref = T.paint_frame(sc, boxes[sl], h, w).max()
            assert got[f] == ref, (f, got[f], ref)
    assert got[3] == -T.BIG
    # AUC: exact pair counts vs sklearn, with ties
    s = np.round(rng.standard_normal(1500), 1)
    y = rng.random(1500) < 0.25
    a = scoring.roc_auc(torch.from_numpy(s).cuda(), y)
    assert abs(a - roc_auc_score(y, s)) < 1e-12
    assert np.isnan(scoring.roc_auc(torch.from_numpy(s).cuda(), np.zeros(1500, bool)))
