"""ORACLE (test infrastructure, NOT product code) -- numpy restatements of FlowNet2's three CUDA ops (forward only).

PARITY UNPINNED: the reference ships no tests or golden vectors for these ops, its CUDA sources cannot be compiled in
this image (no nvcc; THC / torch.utils.ffi were removed from PyTorch) and the prebuilt .so files are cpython-3.6/CUDA
binaries, so these restatements are checked only against closed-form known answers (tests/test_flow_ops.py), not
against outputs of the reference itself.  Each function follows the cited kernel line by line.

  correlation_fwd  <- FlowNet2_src/models/components/ops/correlation/src/correlation_cuda_kernel.cu:10-106
                      (+ output-size rule correlation_cuda.c:25-34)
  resample2d_fwd   <- ops/resample2d/src/Resample2d_kernel.cu:20-66
  channelnorm_fwd  <- ops/channelnorm/src/ChannelNorm_kernel.cu:19-51
"""
import math

import numpy as np


def correlation_out_shape(C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2):
    kr = (kernel_size - 1) // 2
    br = kr + max_displacement
    d = (max_displacement // stride2) * 2 + 1
    oH = int(math.ceil(float(H + 2 * pad_size - 2 * br) / float(stride1)))
    oW = int(math.ceil(float(W + 2 * pad_size - 2 * br) / float(stride1)))
    return d * d, oH, oW


def correlation_fwd(in1, in2, pad_size, kernel_size, max_displacement, stride1, stride2, corr_multiply=1):
    """in1, in2: [B,C,H,W] float32.  out[n, tc, y, x] = sum_{j,i,ch} p1[n,y1+j,x1+i,ch] * p2[n,y2+j,x2+i,ch] / (k*k*C)
    with p* the zero-padded NHWC copies (channels_first kernel), y1 = y*stride1 + max_displacement + kernel_rad,
    y2 = y1 + tj*stride2, tc = (tj + dr)*D + (ti + dr)  (tj = y displacement)."""
    B, C, H, W = in1.shape
    oC, oH, oW = correlation_out_shape(C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2)
    kr = (kernel_size - 1) // 2
    dr = max_displacement // stride2
    D = 2 * dr + 1
    p1 = np.zeros((B, H + 2 * pad_size, W + 2 * pad_size, C), np.float32)
    p2 = np.zeros_like(p1)
    p1[:, pad_size:pad_size + H, pad_size:pad_size + W] = np.transpose(in1, (0, 2, 3, 1))
    p2[:, pad_size:pad_size + H, pad_size:pad_size + W] = np.transpose(in2, (0, 2, 3, 1))
    out = np.zeros((B, oC, oH, oW), np.float32)
    nelems = float(kernel_size * kernel_size * C)
    ys = np.arange(oH) * stride1 + max_displacement + kr
    xs = np.arange(oW) * stride1 + max_displacement + kr
    for tj in range(-dr, dr + 1):
        for ti in range(-dr, dr + 1):
            acc = np.zeros((B, oH, oW), np.float64)
            for j in range(-kr, kr + 1):
                for i in range(-kr, kr + 1):
                    a = p1[:, (ys + j)[:, None], (xs + i)[None, :], :]
                    b = p2[:, (ys + tj * stride2 + j)[:, None], (xs + ti * stride2 + i)[None, :], :]
                    acc += np.sum(a.astype(np.float64) * b.astype(np.float64), axis=-1)
            out[:, (tj + dr) * D + (ti + dr)] = (acc / nelems).astype(np.float32)
    return out


def resample2d_fwd(img, flow, kernel_size=1):
    """img [B,C,H,W], flow [B,2,fH,fW] (dx, dy).  Bilinear weights from the un-clamped coordinate, corner indices
    clamped to [0, dim-1] of the OUTPUT extents; double-precision products accumulated into a float (the kernel mixes
    double literals with float data)."""
    assert kernel_size == 1
    B, C, H, W = img.shape
    _, _, fH, fW = flow.shape
    x = np.arange(fW, dtype=np.float32)[None, None, :]
    y = np.arange(fH, dtype=np.float32)[None, :, None]
    xf = (x + flow[:, 0]).astype(np.float32)
    yf = (y + flow[:, 1]).astype(np.float32)
    fx, fy = np.floor(xf), np.floor(yf)
    alpha = (xf - fx).astype(np.float32)
    beta = (yf - fy).astype(np.float32)
    xL = np.clip(fx.astype(np.int64), 0, fW - 1)
    xR = np.clip((fx + 1).astype(np.int64), 0, fW - 1)
    yT = np.clip(fy.astype(np.int64), 0, fH - 1)
    yB = np.clip((fy + 1).astype(np.int64), 0, fH - 1)
    a, b = alpha.astype(np.float64), beta.astype(np.float64)
    out = np.zeros((B, C, fH, fW), np.float32)
    bi = np.arange(B)[:, None, None]
    for c in range(C):
        p = img[:, c]
        val = np.zeros((B, fH, fW), np.float32)
        for wgt, yy, xx in (((1. - a) * (1. - b), yT, xL), (a * (1. - b), yT, xR), ((1. - a) * b, yB, xL), (a * b, yB, xR)):
            val = (val.astype(np.float64) + wgt * p[bi, yy, xx].astype(np.float64)).astype(np.float32)
        out[:, c] = val
    return out


def channelnorm_fwd(x, norm_deg=2):
    """x [B,C,H,W] -> [B,1,H,W] = sqrt(sum_c x^2), float accumulation in channel order with fused multiply-add."""
    acc = np.zeros((x.shape[0],) + x.shape[2:], np.float32)
    for c in range(x.shape[1]):
        v = x[:, c].astype(np.float64)
        acc = (v * v + acc.astype(np.float64)).astype(np.float32)      # fmaf: one rounding
    return np.sqrt(acc)[:, None].astype(np.float32)
