"""FlowNet2's native ops as nn.Modules on the hand-written gfx950 kernels (forward only -- FlowNet2 is inference-only in
VEC_VAD, calc_optical_flow.py:56-57,74-75).

Same class names, constructor arguments and tensor contract (NCHW fp32 contiguous CUDA tensors, output allocated by the
wrapper, launch on the current stream) as the reference's cffi wrappers:
  Correlation   FlowNet2_src/models/components/ops/correlation/modules/correlation.py:6-27, functions/correlation.py:5-36
  Resample2d    ops/resample2d/modules/resample2d.py:6-14, functions/resample2d.py:5-21
  ChannelNorm   ops/channelnorm/modules/channelnorm.py:6-13, functions/channelnorm.py:5-16
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib as L


def _check_in(*ts):
    for t in ts:
        if not t.is_cuda:
            raise L.VecVadHipError('FlowNet2 ops run on the GPU only (no CPU fallback)')
        if t.dtype != torch.float32:
            raise TypeError('fp32 expected, got %s' % t.dtype)
        assert t.is_contiguous()     # functions/correlation.py:17-18


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def correlation(input1, input2, pad_size=3, kernel_size=3, max_displacement=20, stride1=1, stride2=2, corr_multiply=1):
    _check_in(input1, input2)
    lib = L.lib()
    B, Cc, H, W = input1.shape
    oc, oh, ow = C.c_int32(), C.c_int32(), C.c_int32()
    L.check(lib.vv_correlation_out_shape(Cc, H, W, pad_size, kernel_size, max_displacement, stride1, stride2,
                                         C.byref(oc), C.byref(oh), C.byref(ow)), 'correlation_out_shape')
    out = torch.empty(B, oc.value, oh.value, ow.value, device=input1.device, dtype=torch.float32)
    L.check(lib.vv_correlation_fwd(input1.data_ptr(), input2.data_ptr(), out.data_ptr(), B, Cc, H, W, pad_size, kernel_size,
                                   max_displacement, stride1, stride2, corr_multiply, _stream(input1)), 'correlation_fwd')
    return out


def resample2d(input1, input2, kernel_size=1):
    _check_in(input1, input2)
    B, Cc, H, W = input1.shape
    _, two, fH, fW = input2.shape
    assert two == 2
    out = torch.empty(B, Cc, fH, fW, device=input1.device, dtype=torch.float32)
    L.check(L.lib().vv_resample2d_fwd(input1.data_ptr(), input2.data_ptr(), out.data_ptr(), B, Cc, H, W, fH, fW, kernel_size,
                                      _stream(input1)), 'resample2d_fwd')
    return out


def channelnorm(input1, norm_deg=2):
    _check_in(input1)
    B, Cc, H, W = input1.shape
    out = torch.empty(B, 1, H, W, device=input1.device, dtype=torch.float32)
    L.check(L.lib().vv_channelnorm_fwd(input1.data_ptr(), out.data_ptr(), B, Cc, H, W, norm_deg, _stream(input1)),
            'channelnorm_fwd')
    return out


class Correlation(nn.Module):
    def __init__(self, pad_size=0, kernel_size=0, max_displacement=0, stride1=1, stride2=2, corr_multiply=1):
        super().__init__()
        self.pad_size, self.kernel_size, self.max_displacement = pad_size, kernel_size, max_displacement
        self.stride1, self.stride2, self.corr_multiply = stride1, stride2, corr_multiply

    def forward(self, input1, input2):
        return correlation(input1, input2, self.pad_size, self.kernel_size, self.max_displacement, self.stride1,
                           self.stride2, self.corr_multiply)


class Resample2d(nn.Module):
    def __init__(self, kernel_size=1):
        super().__init__()
        self.kernel_size = kernel_size

    def forward(self, input1, input2):
        return resample2d(input1.contiguous(), input2, self.kernel_size)      # resample2d.py:12 makes input1 contiguous


class ChannelNorm(nn.Module):
    def __init__(self, norm_deg=2):
        super().__init__()
        self.norm_deg = norm_deg

    def forward(self, input1):
        return channelnorm(input1, self.norm_deg)
