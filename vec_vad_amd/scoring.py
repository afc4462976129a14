"""Score aggregation on the GPU.

``frame_scores``: reference test.py:330-358 turns each cube's (raw, flow) reconstruction error into
``w_raw * (raw - mu_r) / sd_r + w_of * (of - mu_o) / sd_o``, paints it into a per-cube h x w float64 mask, max-combines the
masks, saves the frame mask with torch.save, re-loads it and takes ``.max()`` (test.py:387-392).  The maximum of a max-combined
mask is the maximum over the cubes whose painted rectangle is non-empty (``-1e5`` if there is none), which is what
``vv_frame_scores`` computes directly from the device-resident per-cube errors.

``roc_auc``: frame-level ROC-AUC (utils.py:29-41, sklearn roc_curve + auc) as the exact Mann-Whitney pair count.
"""
import math

import numpy as np
import torch

from . import _lib

BIG = 100000          # test.py:188 big_number


def box_paints(bboxes, h, w):
    """1 where ``mask[ceil(y1):ceil(y2), ceil(x1):ceil(x2)] = score`` (test.py:352-355) touches at least one pixel."""
    out = np.zeros(len(bboxes), np.uint8)
    for m, b in enumerate(bboxes):
        x0, x1 = int(math.ceil(b[0])), int(math.ceil(b[2]))
        y0, y1 = int(math.ceil(b[1])), int(math.ceil(b[3]))
        out[m] = len(range(*slice(y0, y1).indices(h))) > 0 and len(range(*slice(x0, x1).indices(w))) > 0
    return out


def frame_scores(raw, of, frame_off, cube_stat, stats, paints, w_raw, w_of, out=None):
    """raw / of: CUDA float32 ``[n]`` per-cube errors (``of=None`` when useFlow is off); frame_off: int32 ``[F+1]`` CSR
    offsets of each frame's cubes; cube_stat: int32 ``[n]`` row of ``stats`` (``-1`` = block without a trained model ->
    score ``BIG``); stats: float64 ``[S,4]`` (mu_r, sd_r, mu_o, sd_o); paints: uint8 ``[n]``.
    Returns (and max-accumulates into ``out`` if given) the CUDA float64 ``[F]`` frame scores."""
    dev = raw.device
    if not raw.is_cuda:
        raise _lib.VecVadHipError('frame_scores needs device-resident scores; vec_vad_amd has no CPU path')

    def dv(a, dt):
        t = a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))
        return t.to(device=dev, dtype=dt).contiguous()

    frame_off, cube_stat, paints = dv(frame_off, torch.int32), dv(cube_stat, torch.int32), dv(paints, torch.uint8)
    stats = dv(np.asarray(stats, np.float64).reshape(-1, 4) if not torch.is_tensor(stats) else stats, torch.float64)
    if stats.numel() == 0:
        stats = torch.zeros((1, 4), dtype=torch.float64, device=dev)
    raw = raw.to(torch.float32).contiguous()
    of = of.to(torch.float32).contiguous() if of is not None else None
    F = frame_off.numel() - 1
    if out is None:
        out = torch.full((F,), -float(BIG), dtype=torch.float64, device=dev)
    _lib.check(_lib.lib().vv_frame_scores(raw.data_ptr(), of.data_ptr() if of is not None else None,
                                         frame_off.data_ptr(), cube_stat.data_ptr(), stats.data_ptr(), paints.data_ptr(),
                                         float(w_raw), float(w_of), float(BIG), F, out.data_ptr(),
                                         torch.cuda.current_stream(dev).cuda_stream), 'vv_frame_scores')
    return out


def roc_auc(scores, labels):
    """Frame-level ROC-AUC of CUDA float64 ``scores`` against boolean ``labels`` (tie-aware; nan when a class is empty)."""
    if not scores.is_cuda:
        raise _lib.VecVadHipError('roc_auc needs device-resident scores; vec_vad_amd has no CPU path')
    dev = scores.device
    scores = scores.to(torch.float64).contiguous().view(-1)
    labels = (labels if torch.is_tensor(labels) else torch.from_numpy(np.ascontiguousarray(labels)))
    labels = (labels.to(dev) != 0).to(torch.uint8).contiguous().view(-1)
    out = torch.zeros(3, dtype=torch.int64, device=dev)
    _lib.check(_lib.lib().vv_roc_auc_counts(scores.data_ptr(), labels.data_ptr(), scores.numel(), out.data_ptr(),
                                           torch.cuda.current_stream(dev).cuda_stream), 'vv_roc_auc_counts')
    c2, p, n = (int(v) for v in out.cpu())
    return float('nan') if p == 0 or n == 0 else c2 / (2.0 * p * n)
