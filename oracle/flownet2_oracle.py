"""ORACLE (test infrastructure, NOT product code) -- functional restatement of the FlowNet2 forward graph.

Follows FlowNet2_src/models/flownet2.py:65-149 and components/FlowNet{C,S,SD,Fusion}.py forward methods on a
reference-named ``state_dict`` with torch CPU functional ops; the three native ops come from
``oracle/flow_ops_oracle.py``.  Pinned by tests/golden/flownet2_*.npz, produced by importing the real FlowNet2 python
graph (tests/golden/make_flownet2_golden.py) with the native ops stubbed by the same numpy restatements -- i.e. the
conv/deconv/upsample/concat graph is pinned to the reference, the three CUDA ops are not ("parity unpinned", see
flow_ops_oracle.py).  No pretrained weights exist offline: weights are formula-seeded.
"""
import zlib
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from . import flow_ops_oracle as ops


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def seeded_state_dict(shapes, seed=0):
    """shapes: iterable of (name, shape).  Xavier-uniform-like weights, small biases; PCG64 stream keyed by name."""
    sd = OrderedDict()
    for name, shape in shapes:
        rng = np.random.default_rng(zlib.crc32(name.encode()) + 104729 * seed)
        if len(shape) == 4:
            rf = shape[2] * shape[3]
            bound = np.sqrt(6.0 / ((shape[0] + shape[1]) * rf))
            v = rng.uniform(-bound, bound, shape)
        else:
            v = rng.uniform(-0.1, 0.1, shape)
        sd[name] = torch.from_numpy(v.astype(np.float32))
    return sd


def _conv(sd, name, x, stride=1, relu=True):
    w = sd[name + '.0.weight']
    y = F.conv2d(x, w, sd[name + '.0.bias'], stride=stride, padding=(w.shape[2] - 1) // 2)
    return F.leaky_relu(y, 0.1) if relu else y


def _deconv(sd, name, x):
    return F.leaky_relu(F.conv_transpose2d(x, sd[name + '.0.weight'], sd[name + '.0.bias'], stride=2, padding=1), 0.1)


def _pf(sd, name, x):
    return F.conv2d(x, sd[name + '.weight'], sd[name + '.bias'], padding=1)


def _upflow(sd, name, x):
    return F.conv_transpose2d(x, sd[name + '.weight'], sd.get(name + '.bias'), stride=2, padding=1)


def _decoder(sd, p, c6, c5, c4, c3, c2, inter=False):
    flow6 = _pf(sd, p + 'predict_flow6', c6)
    cat5 = torch.cat((c5, _deconv(sd, p + 'deconv5', c6), _upflow(sd, p + 'upsampled_flow6_to_5', flow6)), 1)
    cats = {5: cat5}
    skips = {4: c4, 3: c3, 2: c2}
    for lvl in (5, 4, 3):
        cat = cats[lvl]
        src = _conv(sd, p + 'inter_conv%d' % lvl, cat, relu=False) if inter else cat
        flow = _pf(sd, p + 'predict_flow%d' % lvl, src)
        cats[lvl - 1] = torch.cat((skips[lvl - 1], _deconv(sd, p + 'deconv%d' % (lvl - 1), cat),
                                   _upflow(sd, p + 'upsampled_flow%d_to_%d' % (lvl, lvl - 1), flow)), 1)
    src = _conv(sd, p + 'inter_conv2', cats[2], relu=False) if inter else cats[2]
    return _pf(sd, p + 'predict_flow2', src)


def flownetc(sd, p, x):
    x1, x2 = x[:, :3], x[:, 3:]
    a = _conv(sd, p + 'conv3', _conv(sd, p + 'conv2', _conv(sd, p + 'conv1', x1, 2), 2), 2)
    c2a = _conv(sd, p + 'conv2', _conv(sd, p + 'conv1', x1, 2), 2)
    b = _conv(sd, p + 'conv3', _conv(sd, p + 'conv2', _conv(sd, p + 'conv1', x2, 2), 2), 2)
    corr = _t(ops.correlation_fwd(a.numpy(), b.numpy(), 20, 1, 20, 1, 2, 1))
    corr = F.leaky_relu(corr, 0.1)
    in31 = torch.cat((_conv(sd, p + 'conv_redir', a), corr), 1)
    c3 = _conv(sd, p + 'conv3_1', in31)
    c4 = _conv(sd, p + 'conv4_1', _conv(sd, p + 'conv4', c3, 2))
    c5 = _conv(sd, p + 'conv5_1', _conv(sd, p + 'conv5', c4, 2))
    c6 = _conv(sd, p + 'conv6_1', _conv(sd, p + 'conv6', c5, 2))
    return _decoder(sd, p, c6, c5, c4, c3, c2a)


def flownets(sd, p, x):
    c2 = _conv(sd, p + 'conv2', _conv(sd, p + 'conv1', x, 2), 2)
    c3 = _conv(sd, p + 'conv3_1', _conv(sd, p + 'conv3', c2, 2))
    c4 = _conv(sd, p + 'conv4_1', _conv(sd, p + 'conv4', c3, 2))
    c5 = _conv(sd, p + 'conv5_1', _conv(sd, p + 'conv5', c4, 2))
    c6 = _conv(sd, p + 'conv6_1', _conv(sd, p + 'conv6', c5, 2))
    return _decoder(sd, p, c6, c5, c4, c3, c2)


def flownetsd(sd, p, x):
    c0 = _conv(sd, p + 'conv0', x)
    c1 = _conv(sd, p + 'conv1_1', _conv(sd, p + 'conv1', c0, 2))
    c2 = _conv(sd, p + 'conv2_1', _conv(sd, p + 'conv2', c1, 2))
    c3 = _conv(sd, p + 'conv3_1', _conv(sd, p + 'conv3', c2, 2))
    c4 = _conv(sd, p + 'conv4_1', _conv(sd, p + 'conv4', c3, 2))
    c5 = _conv(sd, p + 'conv5_1', _conv(sd, p + 'conv5', c4, 2))
    c6 = _conv(sd, p + 'conv6_1', _conv(sd, p + 'conv6', c5, 2))
    return _decoder(sd, p, c6, c5, c4, c3, c2, inter=True)


def flownetfusion(sd, p, x):
    c0 = _conv(sd, p + 'conv0', x)
    c1 = _conv(sd, p + 'conv1_1', _conv(sd, p + 'conv1', c0, 2))
    c2 = _conv(sd, p + 'conv2_1', _conv(sd, p + 'conv2', c1, 2))
    flow2 = _pf(sd, p + 'predict_flow2', c2)
    cat1 = torch.cat((c1, _deconv(sd, p + 'deconv1', c2), _upflow(sd, p + 'upsampled_flow2_to_1', flow2)), 1)
    flow1 = _pf(sd, p + 'predict_flow1', _conv(sd, p + 'inter_conv1', cat1, relu=False))
    cat0 = torch.cat((c0, _deconv(sd, p + 'deconv0', cat1), _upflow(sd, p + 'upsampled_flow1_to_0', flow1)), 1)
    return _pf(sd, p + 'predict_flow0', _conv(sd, p + 'inter_conv0', cat0, relu=False))


def _resample(img, flow):
    return _t(ops.resample2d_fwd(img.contiguous().numpy(), flow.contiguous().numpy()))


def _cnorm(x):
    return _t(ops.channelnorm_fwd(x.contiguous().numpy()))


@torch.no_grad()
def flownet2_forward(sd, inputs, rgb_max=255.0, div_flow=20.0, return_parts=False, align_corners=False):
    """inputs [B,3,2,H,W] float32 0..255 -> flow [B,2,H,W]  (flownet2.py:65-149).  ``nn.Upsample(scale_factor=4,
    mode='bilinear')`` (flownet2.py:28,34) is align_corners=False under the torch that imports the reference today (the pinned
    golden) and was align_corners=True under the authors' PyTorch 0.3 (README.md:10,64): ``align_corners`` selects which."""
    rgb_mean = inputs.contiguous().view(inputs.size()[:2] + (-1,)).mean(dim=-1).view(inputs.size()[:2] + (1, 1, 1))
    x = (inputs - rgb_mean) / rgb_max
    x1, x2 = x[:, :, 0], x[:, :, 1]
    x = torch.cat((x1, x2), 1)
    up_b = lambda t: F.interpolate(t, scale_factor=4, mode='bilinear', align_corners=bool(align_corners))
    up_n = lambda t: F.interpolate(t, scale_factor=4, mode='nearest')
    c2 = flownetc(sd, 'flownetc.', x)
    c_flow = up_b(c2 * div_flow)

    def pack(flow):
        warped = _resample(x[:, 3:], flow)
        return torch.cat([x, warped, flow / div_flow, _cnorm(x[:, :3] - warped)], 1)

    s1_2 = flownets(sd, 'flownets_1.', pack(c_flow))
    s1_flow = up_b(s1_2 * div_flow)
    s2_2 = flownets(sd, 'flownets_2.', pack(s1_flow))
    s2_flow = up_n(s2_2 * div_flow)
    norm_s2 = _cnorm(s2_flow)
    diff_s2 = _cnorm(x[:, :3] - _resample(x[:, 3:], s2_flow))
    sd_2 = flownetsd(sd, 'flownets_d.', x)
    sd_flow = up_n(sd_2 / div_flow)
    norm_sd = _cnorm(sd_flow)
    diff_sd = _cnorm(x[:, :3] - _resample(x[:, 3:], sd_flow))
    cat3 = torch.cat((x[:, :3], sd_flow, s2_flow, norm_sd, norm_s2, diff_sd, diff_s2), 1)
    out = flownetfusion(sd, 'flownetfusion.', cat3)
    if return_parts:
        return out, dict(c_flow2=c2, s1_flow2=s1_2, s2_flow2=s2_2, sd_flow2=sd_2)
    return out
