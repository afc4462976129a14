"""FlowNet2 forward pass (inference only, as VEC_VAD uses it: calc_optical_flow.py:15-22,56-57) on gfx950 kernels.

Module tree / constructor signatures / ``state_dict`` keys follow the reference (FlowNet2_src/models/flownet2.py:10-149,
components/FlowNetC.py:10-132, FlowNetS.py:11-96, FlowNetSD.py:9-103, FlowNetFusion.py:9-64, misc.py:8-44) so that
``FlowNet2_checkpoint.pth.tar['state_dict']`` loads unchanged.  What runs: every conv / deconv / predict_flow layer is
the hand-written MFMA kernel ``vv_conv2d_mfma`` on NHWC buffers (producers write straight into the channel slices of
the consumer's concat buffer), the three native ops are ``vv_correlation_fwd / vv_resample2d_fwd / vv_channelnorm_fwd``
and the plumbing between the sub-networks (input normalisation, x4 flow up-sampling, warp, brightness error, concat) is three
NHWC kernels (``vv_flownet_prep / vv_warp_pack12 / vv_fusion_pack11``).  No torch.nn op, no ATen kernel between input and
output, no fallback.

``with_bn`` must be False and ``fp16`` False (what VEC_VAD instantiates, flownet2.py:12-17).
"""
import os
import ctypes as C

import torch
import torch.nn as nn

from . import _lib as L
from .flow_ops import Correlation, Resample2d, ChannelNorm, correlation, resample2d, channelnorm


# split-K: a tiny-map layer is split until its workgroups fill the chip ONCE (256 CUs); 512 measured 0.17 ms slower per forward
# (twice the partial-sum traffic through vv_conv2d_splitk_finish), 128 / 192 leave half the chip idle
_KS_TARGET = int(os.environ.get('VV_FN2_KS_TARGET', '256'))
_WINO = os.environ.get('VV_FN2_WINO', '1') != '0'
_WINO_MIN_WGS = int(os.environ.get('VV_FN2_WINO_MIN_WGS', '100'))


def _c4(c):
    return (c + 3) // 4 * 4


def _c16(c):
    return (c + 15) // 16 * 16


def _c32(c):
    return (c + 31) // 32 * 32


def conv(in_channels, out_channels, kernel_size=3, stride=1, bias=True, with_bn=False, with_relu=True):
    """reference components/misc.py:8-28 (with_bn=False only)."""
    if with_bn:
        raise NotImplementedError('FlowNet2 is built with_bn=False in VEC_VAD (flownet2.py:13)')
    layers = [nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=(kernel_size - 1) // 2, bias=True)]
    if with_relu:
        layers.append(nn.LeakyReLU(0.1, inplace=True))
    return nn.Sequential(*layers)


def deconv(in_channels, out_channels):
    """reference components/misc.py:31-39."""
    return nn.Sequential(nn.ConvTranspose2d(in_channels, out_channels, kernel_size=4, stride=2, padding=1, bias=True),
                         nn.LeakyReLU(0.1, inplace=True))


def predict_flow(in_channels):
    """reference components/misc.py:42-44."""
    return nn.Conv2d(in_channels, 2, kernel_size=3, stride=1, padding=1, bias=True)


def _xavier_init(module):
    # flownet2.py:50-59 / FlowNetC.py:64-73: uniform(bias), xavier_uniform(weight)
    for m in module.modules():
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            if m.bias is not None:
                nn.init.uniform_(m.bias)
            nn.init.xavier_uniform_(m.weight)


class _Pool:
    """Activation buffers of one forward, handed out in call order and kept per input shape: the zero fill (pad channels
    must be finite, producers only ever write real channels) happens once, not ~90 times per forward."""

    def __init__(self):
        self.by_key, self.cur, self.i = {}, None, 0

    def begin(self, key):
        self.cur, self.i = self.by_key.setdefault(key, []), 0

    def end(self):
        self.cur = None

    def take(self, shape, device):
        if self.cur is None:
            return torch.zeros(shape, device=device, dtype=torch.float32)
        if self.i < len(self.cur) and tuple(self.cur[self.i].shape) == tuple(shape) and self.cur[self.i].device == device:
            t = self.cur[self.i]
        else:
            t = torch.zeros(shape, device=device, dtype=torch.float32)
            del self.cur[self.i:]
            self.cur.append(t)
        self.i += 1
        return t


_ACTIVE_POOL = [None]


class _Buf:
    """NHWC activation buffer [B,H,W,ceil4(C)] (zero initialised so pad channels are finite)."""

    def __init__(self, B, H, W, C, device):
        self.B, self.H, self.W, self.C, self.cs = B, H, W, C, _c4(C)
        pool = _ACTIVE_POOL[0]
        shape = (B, H, W, self.cs)
        self.t = pool.take(shape, torch.device(device)) if pool is not None else \
            torch.zeros(shape, device=device, dtype=torch.float32)

    def view(self, coff=0):
        return L.View(self.t.data_ptr(), 0, self.cs, coff)

    def nchw(self, c0=0, c1=None):
        c1 = self.C if c1 is None else c1
        if self.cs == 4 and c0 == 0 and self.t.is_cuda:
            # (the two-channel flow maps: a first-party launch, so a captured forward holds no framework kernel)
            out = torch.empty(self.B, c1, self.H, self.W, device=self.t.device, dtype=torch.float32)
            L.check(L.lib().vv_out4_to_nchw(self.B, self.H * self.W, c1, self.t.data_ptr(), out.data_ptr(), c1, 0,
                                            torch.cuda.current_stream(self.t.device).cuda_stream), 'out4_to_nchw')
            return out
        return self.t[..., c0:c1].permute(0, 3, 1, 2).contiguous()


class _Runner:
    """Launches one conv / deconv layer; packed weight panels are cached per module and refreshed when the parameter
    tensor changes (load_state_dict bumps ``_version``)."""

    def __init__(self):
        self.lib = L.lib()
        self.cache = {}
        self.hook = None        # hook(label, flop, e0, e1): HIP events around every conv launch (bench diagnostics, eager mode only)
        self.last_wino = False

    def _packed(self, m):
        key = id(m)
        ver = (m.weight.data_ptr(), m.weight._version)
        ent = self.cache.get(key)
        if ent is not None and ent[0] == ver:
            return ent[1]
        w = m.weight.detach().contiguous().float()
        transposed = isinstance(m, nn.ConvTranspose2d)
        K, N = (w.shape[0], w.shape[1]) if transposed else (w.shape[1], w.shape[0])
        taps = w.shape[2] * w.shape[3]
        KP, NP = _c16(K), _c32(N)
        packed = torch.empty(taps * KP * NP, device=w.device, dtype=torch.float32)
        L.check(self.lib.vv_pack_conv2d(w.data_ptr(), packed.data_ptr(), taps, K, KP, N, NP, 1 if transposed else 0,
                                        torch.cuda.current_stream(w.device).cuda_stream), 'pack_conv2d')
        self.cache[key] = (ver, packed, K, KP, N, NP)
        return packed

    def _flow_head(self, m, de, src, dst, dst_coff, slope, stream):
        """Two-channel layers (predict_flow 3x3 conv, upsampled_flow 4x4 deconv): dedicated bandwidth kernels."""
        bias = m.bias.data_ptr() if m.bias is not None else None
        assert m.in_channels == src.C, (m.in_channels, src.C)
        if de:
            if m.in_channels != 2 or m.kernel_size != (4, 4) or m.stride != (2, 2) or m.padding != (1, 1):
                return False
            assert (dst.H, dst.W) == (2 * src.H, 2 * src.W) and dst_coff + 2 <= dst.cs
            w = m.weight.detach()
            assert w.is_contiguous() and w.dtype == torch.float32
            L.check(self.lib.vv_deconv4x4_c2(src.t.data_ptr(), src.cs, src.B, src.H, src.W, w.data_ptr(), bias, slope,
                                             dst.t.data_ptr(), dst.cs, dst_coff, stream), 'deconv4x4_c2')
            return True
        if m.kernel_size != (3, 3) or m.stride != (1, 1) or m.padding != (1, 1):
            return False
        assert (dst.H, dst.W) == (src.H, src.W) and dst_coff + 2 <= dst.cs
        key = ('n2', id(m))
        ver = (m.weight.data_ptr(), m.weight._version)
        ent = self.cache.get(key)
        if ent is None or ent[0] != ver:
            w = m.weight.detach().float()                      # [2][Cin][3][3]
            cin = w.shape[1]
            cp = (cin + 31) // 32 * 32
            wq = torch.zeros(9, cp // 4, 2, 4, device=w.device, dtype=torch.float32)
            wp = torch.zeros(2, cp, 3, 3, device=w.device, dtype=torch.float32)
            wp[:, :cin] = w
            wq.copy_(wp.permute(2, 3, 1, 0).reshape(9, cp // 4, 4, 2).permute(0, 1, 3, 2))
            ent = (ver, wq.contiguous(), cp // 4)
            self.cache[key] = ent
        L.check(self.lib.vv_conv3x3_n2(src.t.data_ptr(), src.cs, src.B, src.H, src.W, m.in_channels, ent[1].data_ptr(), ent[2],
                                       bias, slope, dst.t.data_ptr(), dst.cs, dst_coff, stream), 'conv3x3_n2')
        return True

    def _wino(self, m, src, dst, dst_coff, slope, stream):
        """Large stride-1 3x3 layers in Winograd F(2x2,3x3) form (vv_conv2d_wino): the layers with at least
        VV_FN2_WINO_MIN_WGS (default 100) workgroups of 4 x 32 pixels x 32 channels -- where 2.25x fewer MFMAs is time (the H/32 and H/64 levels stay on the direct
        kernel with its split-K).  VV_FN2_WINO=0 switches it off."""
        if not _WINO or m.kernel_size != (3, 3) or m.stride != (1, 1) or m.padding != (1, 1) or m.out_channels % 32:
            return False
        if src.H % 2 or src.W % 32 or src.B * ((src.H + 3) // 4) * (src.W // 32) * (m.out_channels // 32) < _WINO_MIN_WGS:
            return False
        if src.t.numel() * 4 >= 2 ** 31 or dst.t.numel() * 4 >= 2 ** 31:
            return False
        assert m.in_channels == src.C and (dst.H, dst.W) == (src.H, src.W) and dst_coff + m.out_channels <= dst.cs
        key = ('wino', id(m))
        ver = (m.weight.data_ptr(), m.weight._version)
        ent = self.cache.get(key)
        if ent is None or ent[0] != ver:
            w = m.weight.detach().contiguous().float()          # [Cout][Cin][3][3]
            N, K = w.shape[0], w.shape[1]
            KP = (K + 7) // 8 * 8
            panel = torch.empty(16 * KP * N, device=w.device, dtype=torch.float32)
            tab = torch.frombuffer(bytearray(bytes((L.PackEntry * 1)(L.PackEntry(0, 0, 0, K, KP, N)))), dtype=torch.uint8).to(w.device)
            L.check(self.lib.vv_pack_wino(tab.data_ptr(), 1, 1, w.data_ptr(), w.numel(), panel.data_ptr(), panel.numel(), KP * N, stream),
                    'pack_wino')
            ent = (ver, panel, KP, tab)
            self.cache[key] = ent
        bias = m.bias.data_ptr() if m.bias is not None else None
        L.check(self.lib.vv_conv2d_wino(src.t.data_ptr(), src.cs, 0, src.t.numel(), ent[1].data_ptr(), bias, slope, dst.t.data_ptr(),
                                        dst.cs, dst_coff, src.B, src.H, src.W, ent[2], m.out_channels, stream), 'conv2d_wino')
        self.last_wino = True           # (bench diagnostics: this launch executed 16/36 of the direct form's multiply-adds)
        return True

    def _rowk(self, m, src, dst, dst_coff, slope, stream):
        """Few-channel first layers (FlowNetC conv1: 7x7 s2 on 3 channels; FlowNetSD conv0: 3x3 s1 on 6): vv_conv2d_mfma kind 2,
        K = the flattened (kx, c) run under one filter row instead of taps x 16 zero-padded channels."""
        R, stride, cs = m.kernel_size[0], m.stride[0], src.cs
        if os.environ.get('VV_FN2_ROWK', '1') == '0' or (R, stride, cs) not in ((7, 2, 4), (3, 1, 8)) or m.kernel_size[1] != R:
            return False
        pad = (R - 1) // 2
        if m.padding != (pad, pad) or m.in_channels != src.C or src.C > cs:
            return False
        KF = (R * cs + 7) // 8 * 8
        N = m.out_channels
        NP = _c32(N)
        key = ('rowk', id(m))
        ver = (m.weight.data_ptr(), m.weight._version)
        ent = self.cache.get(key)
        if ent is None or ent[0] != ver:
            w = m.weight.detach().float()                                  # [N][Cin][ky][kx]
            wr = torch.zeros(N, KF, R, device=w.device, dtype=torch.float32)
            wr[:, :R * cs].view(N, R, cs, R)[:, :, :m.in_channels] = w.permute(0, 3, 1, 2)      # [N][kx][c][ky]
            packed = torch.empty(R * KF * NP, device=w.device, dtype=torch.float32)
            L.check(self.lib.vv_pack_conv2d(wr.data_ptr(), packed.data_ptr(), R, KF, KF, N, NP, 0,
                                            torch.cuda.current_stream(w.device).cuda_stream), 'pack_conv2d (row-K)')
            ent = (ver, packed)
            self.cache[key] = ent
        OH, OW = (src.H + 2 * pad - R) // stride + 1, (src.W + 2 * pad - R) // stride + 1
        assert (dst.H, dst.W) == (OH, OW) and dst_coff + N <= dst.cs
        bias = m.bias.data_ptr() if m.bias is not None else None
        p = L.Conv2dParams(2, R, stride, src.B, src.H, src.W, KF, KF, N, NP, src.view(0), ent[1].data_ptr(), bias, slope, 0,
                           dst.view(dst_coff))
        L.check(self.lib.vv_conv2d_mfma(C.byref(p), stream), 'conv2d row-K %dx%d s%d %d->%d' % (R, R, stride, m.in_channels, N))
        return True

    def __call__(self, layer, src, dst, dst_coff=0):
        if self.hook is None:
            return self._launch(layer, src, dst, dst_coff)
        m = layer[0] if isinstance(layer, nn.Sequential) else layer
        de = isinstance(m, nn.ConvTranspose2d)
        label = '%s%dx%d_s%d' % ('deconv' if de else 'conv', m.kernel_size[0], m.kernel_size[1], m.stride[0])
        if m.out_channels == 2:
            label += '_n2'
        oh, ow = dst.H, dst.W
        flop = 2.0 * src.B * (src.H * src.W if de else oh * ow) * m.in_channels * m.out_channels * m.kernel_size[0] * m.kernel_size[1]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.last_wino = False
        e0.record()
        r = self._launch(layer, src, dst, dst_coff)
        e1.record()
        self.hook(label + ('_wino' if self.last_wino else ''), flop, e0, e1)
        return r

    def _launch(self, layer, src, dst, dst_coff=0):
        """layer: nn.Sequential(Conv2d|ConvTranspose2d[, LeakyReLU]) or a bare Conv2d / ConvTranspose2d."""
        if isinstance(layer, nn.Sequential):
            m = layer[0]
            slope = 0.1 if len(layer) > 1 else 1.0
        else:
            m, slope = layer, 1.0
        de = isinstance(m, nn.ConvTranspose2d)
        stream = torch.cuda.current_stream(src.t.device).cuda_stream
        if m.out_channels == 2 and self._flow_head(m, de, src, dst, dst_coff, slope, stream):
            return dst
        if not de and self._rowk(m, src, dst, dst_coff, slope, stream):
            return dst
        if not de and self._wino(m, src, dst, dst_coff, slope, stream):
            return dst
        packed = self._packed(m)
        _, _, K, KP, N, NP = self.cache[id(m)]
        assert K == src.C, (K, src.C)
        if de:
            assert m.kernel_size == (4, 4) and m.stride == (2, 2) and m.padding == (1, 1)
            OH, OW = 2 * src.H, 2 * src.W
            R, stride = 4, 2
        else:
            R, stride = m.kernel_size[0], m.stride[0]
            pad = (R - 1) // 2
            assert m.padding == (pad, pad)
            OH, OW = (src.H + 2 * pad - R) // stride + 1, (src.W + 2 * pad - R) // stride + 1
        assert (dst.H, dst.W) == (OH, OW), ((dst.H, dst.W), (OH, OW))
        assert dst_coff + N <= dst.cs
        bias = m.bias.data_ptr() if m.bias is not None else None
        # tiny-M / huge-K layers (the H/32 and H/64 levels): split the input-channel loop over workgroups
        lh, lw = (src.H, src.W) if de else (OH, OW)
        wgs = src.B * (4 if de else 1) * (NP // (64 if (NP % 64 == 0 and N > 32) else 32)) * ((lh + 7) // 8) * ((lw + 31) // 32)
        nchunk = KP // (16 if (de or stride == 1) else 8)
        ks = 1
        if wgs < 256 and nchunk >= 4:
            ks = max(1, min(nchunk // 2, _KS_TARGET // wgs, 16))
        if ks > 1:
            M = src.B * OH * OW
            ws = torch.empty(ks * M * NP, device=src.t.device, dtype=torch.float32)
            p = L.Conv2dParams(1 if de else 0, R, stride, src.B, src.H, src.W, K, KP, N, NP, src.view(0), packed.data_ptr(),
                               None, 1.0, ks, L.View(ws.data_ptr(), 0, NP, 0))
            L.check(self.lib.vv_conv2d_mfma(C.byref(p), stream), 'conv2d split-k')
            L.check(self.lib.vv_conv2d_splitk_finish(ws.data_ptr(), ks, M, N, NP, bias, slope, dst.t.data_ptr(), dst.cs, dst_coff,
                                                     stream), 'conv2d split-k finish')
            return dst
        p = L.Conv2dParams(1 if de else 0, R, stride, src.B, src.H, src.W, K, KP, N, NP, src.view(0), packed.data_ptr(),
                           bias, slope, 0, dst.view(dst_coff))
        L.check(self.lib.vv_conv2d_mfma(C.byref(p), stream), 'conv2d %dx%d s%d %d->%d' % (R, R, stride, K, N))
        return dst


def _upsample4(x_nchw, bilinear, scale, align_corners=False):
    """nn.Upsample(scale_factor=4) on NCHW planes times ``scale``: nearest, bilinear, or bilinear with align_corners=True."""
    x = x_nchw.contiguous()
    B, Cc, H, W = x.shape
    out = torch.empty(B, Cc, 4 * H, 4 * W, device=x.device, dtype=torch.float32)
    L.check(L.lib().vv_upsample4(x.data_ptr(), out.data_ptr(), B * Cc, H, W, (2 if align_corners else 1) if bilinear else 0, float(scale),
                                 torch.cuda.current_stream(x.device).cuda_stream), 'upsample4')
    return out


def _to_buf(x_nchw, device=None):
    B, Cc, H, W = x_nchw.shape
    b = _Buf(B, H, W, Cc, x_nchw.device)
    b.t[..., :Cc] = x_nchw.permute(0, 2, 3, 1)
    return b


class _Decoder:
    """The refinement ladder shared by FlowNetC / FlowNetS (predict_flow on the concat buffers, FlowNetC.py:104-127)."""

    @staticmethod
    def run(net, run, out_conv6, cat5, cat4, cat3, cat2, inter=False):
        dev = out_conv6.t.device
        B = out_conv6.B

        def flow_of(pred, src):
            f = _Buf(B, src.H, src.W, 2, dev)
            run(pred, src, f)
            return f

        flow6 = flow_of(net.predict_flow6, out_conv6)
        run(net.upsampled_flow6_to_5, flow6, cat5, cat5.C - 2)
        run(net.deconv5, out_conv6, cat5, cat5.C - 2 - net.deconv5[0].out_channels)
        cats = [(cat5, 5, cat4), (cat4, 4, cat3), (cat3, 3, cat2)]
        for cat, lvl, nxt in cats:
            src = cat
            if inter:
                ic = getattr(net, 'inter_conv%d' % lvl)
                src = _Buf(B, cat.H, cat.W, ic[0].out_channels, dev)
                run(ic, cat, src)
            flow = flow_of(getattr(net, 'predict_flow%d' % lvl), src)
            run(getattr(net, 'upsampled_flow%d_to_%d' % (lvl, lvl - 1)), flow, nxt, nxt.C - 2)
            d = getattr(net, 'deconv%d' % (lvl - 1))
            run(d, cat, nxt, nxt.C - 2 - d[0].out_channels)
        src = cat2
        if inter:
            src = _Buf(B, cat2.H, cat2.W, net.inter_conv2[0].out_channels, dev)
            run(net.inter_conv2, cat2, src)
        return flow_of(net.predict_flow2, src)


_TOWER_STREAMS = {}


class FlowNetC(nn.Module):
    def __init__(self, with_bn=False, fp16=False):
        super().__init__()
        assert not with_bn and not fp16
        self.with_bn, self.fp16 = with_bn, fp16
        self.conv1 = conv(3, 64, kernel_size=7, stride=2)
        self.conv2 = conv(64, 128, kernel_size=5, stride=2)
        self.conv3 = conv(128, 256, kernel_size=5, stride=2)
        self.conv_redir = conv(256, 32, kernel_size=1, stride=1)
        self.corr = Correlation(pad_size=20, kernel_size=1, max_displacement=20, stride1=1, stride2=2, corr_multiply=1)
        self.corr_activation = nn.LeakyReLU(0.1, inplace=True)
        self.conv3_1 = conv(473, 256)
        self.conv4 = conv(256, 512, stride=2)
        self.conv4_1 = conv(512, 512)
        self.conv5 = conv(512, 512, stride=2)
        self.conv5_1 = conv(512, 512)
        self.conv6 = conv(512, 1024, stride=2)
        self.conv6_1 = conv(1024, 1024)
        self.deconv5 = deconv(1024, 512)
        self.deconv4 = deconv(1026, 256)
        self.deconv3 = deconv(770, 128)
        self.deconv2 = deconv(386, 64)
        self.predict_flow6 = predict_flow(1024)
        self.predict_flow5 = predict_flow(1026)
        self.predict_flow4 = predict_flow(770)
        self.predict_flow3 = predict_flow(386)
        self.predict_flow2 = predict_flow(194)
        for a, b in ((6, 5), (5, 4), (4, 3), (3, 2)):
            setattr(self, 'upsampled_flow%d_to_%d' % (a, b), nn.ConvTranspose2d(2, 2, 4, 2, 1, bias=True))
        self.upsample1 = nn.Upsample(scale_factor=4, mode='bilinear')
        _xavier_init(self)

    def run(self, run, img0, img1):
        """img0 / img1: _Buf [B,H,W,3].  Returns flow2 _Buf [B,H/4,W/4,2] (FlowNetC.py:75-132)."""
        dev, B, H, W = img0.t.device, img0.B, img0.H, img0.W
        nb = lambda h, w, c: _Buf(B, h, w, c, dev)
        cat2 = nb(H // 4, W // 4, 194)
        c1a, c1b = nb(H // 2, W // 2, 64), nb(H // 2, W // 2, 64)
        c2b = nb(H // 4, W // 4, 128)
        c2a_view = _SliceView(cat2, 0, 128)
        c3a, c3b = nb(H // 8, W // 8, 256), nb(H // 8, W // 8, 256)
        # the two towers of the siamese front end are independent up to the correlation: the second image's tower runs on its own
        # stream (a sibling of the FlowNetSD branch, forked from the same stream -- not nested).  conv2 / conv3 are 224-workgroup
        # launches at one image pair: two of them together fill the 256 CUs twice instead of 7/8 once
        main = torch.cuda.current_stream(dev)
        tower = None
        if os.environ.get('VV_FN2_OVERLAP', '1') not in ('0', 'sd'):
            tower = _TOWER_STREAMS.get(str(dev))
            if tower is None:
                tower = _TOWER_STREAMS[str(dev)] = torch.cuda.Stream(device=dev)
            tower.wait_stream(main)
        with torch.cuda.stream(tower if tower is not None else main):
            run(self.conv1, img1, c1b); run(self.conv2, c1b, c2b); run(self.conv3, c2b, c3b)
        run(self.conv1, img0, c1a); run(self.conv2, c1a, cat2, 0); run(self.conv3, c2a_view, c3a)
        if tower is not None:
            main.wait_stream(tower)
        in31 = nb(H // 8, W // 8, 473)
        # corr + corr_activation + the cat with conv_redir (FlowNetC.py:88-96,120): one launch on the NHWC maps, written
        # into channels [32, 473) of conv3_1's input; widths the specialised kernel does not cover take the generic op
        rc = L.lib().vv_correlation_nhwc(c3a.t.data_ptr(), c3b.t.data_ptr(), c3a.cs, B, 256, H // 8, W // 8, in31.t.data_ptr(),
                                         in31.cs, 32, 0.1, torch.cuda.current_stream(dev).cuda_stream)
        if rc == 3:       # VV_ERR_UNSUPPORTED
            corr = correlation(c3a.nchw(), c3b.nchw(), 20, 1, 20, 1, 2, 1)
            in31.t[..., 32:473] = torch.where(corr > 0, corr, corr * 0.1).permute(0, 2, 3, 1)
        else:
            L.check(rc, 'correlation_nhwc')
        run(self.conv_redir, c3a, in31, 0)
        cat3 = nb(H // 8, W // 8, 386)
        run(self.conv3_1, in31, cat3, 0)
        cat4 = nb(H // 16, W // 16, 770)
        t4 = nb(H // 16, W // 16, 512)
        run(self.conv4, _SliceView(cat3, 0, 256), t4); run(self.conv4_1, t4, cat4, 0)
        cat5 = nb(H // 32, W // 32, 1026)
        t5 = nb(H // 32, W // 32, 512)
        run(self.conv5, _SliceView(cat4, 0, 512), t5); run(self.conv5_1, t5, cat5, 0)
        t6, c6 = nb(H // 64, W // 64, 1024), nb(H // 64, W // 64, 1024)
        run(self.conv6, _SliceView(cat5, 0, 512), t6); run(self.conv6_1, t6, c6)
        return _Decoder.run(self, run, c6, cat5, cat4, cat3, cat2)


class _SliceView:
    """The first C channels of a concat buffer seen as a conv input (coff 0)."""

    def __init__(self, buf, c0, c):
        assert c0 == 0
        self.B, self.H, self.W, self.C, self.cs, self.t = buf.B, buf.H, buf.W, c, buf.cs, buf.t

    def view(self, coff=0):
        return L.View(self.t.data_ptr(), 0, self.cs, coff)


class FlowNetS(nn.Module):
    def __init__(self, input_channels=12, with_bn=False):
        super().__init__()
        assert not with_bn
        self.with_bn = with_bn
        self.conv1 = conv(input_channels, 64, kernel_size=7, stride=2)
        self.conv2 = conv(64, 128, kernel_size=5, stride=2)
        self.conv3 = conv(128, 256, kernel_size=5, stride=2)
        self.conv3_1 = conv(256, 256)
        self.conv4 = conv(256, 512, stride=2)
        self.conv4_1 = conv(512, 512)
        self.conv5 = conv(512, 512, stride=2)
        self.conv5_1 = conv(512, 512)
        self.conv6 = conv(512, 1024, stride=2)
        self.conv6_1 = conv(1024, 1024)
        self.deconv5 = deconv(1024, 512)
        self.deconv4 = deconv(1026, 256)
        self.deconv3 = deconv(770, 128)
        self.deconv2 = deconv(386, 64)
        self.predict_flow6 = predict_flow(1024)
        self.predict_flow5 = predict_flow(1026)
        self.predict_flow4 = predict_flow(770)
        self.predict_flow3 = predict_flow(386)
        self.predict_flow2 = predict_flow(194)
        for a, b in ((6, 5), (5, 4), (4, 3), (3, 2)):
            setattr(self, 'upsampled_flow%d_to_%d' % (a, b), nn.ConvTranspose2d(2, 2, 4, 2, 1, bias=False))
        self.upsample1 = nn.Upsample(scale_factor=4, mode='bilinear')
        _xavier_init(self)

    def run(self, run, x):
        """x: _Buf [B,H,W,12] (FlowNetS.py:63-96)."""
        dev, B, H, W = x.t.device, x.B, x.H, x.W
        nb = lambda h, w, c: _Buf(B, h, w, c, dev)
        c1 = nb(H // 2, W // 2, 64)
        run(self.conv1, x, c1)
        cat2 = nb(H // 4, W // 4, 194)
        run(self.conv2, c1, cat2, 0)
        t3, cat3 = nb(H // 8, W // 8, 256), nb(H // 8, W // 8, 386)
        run(self.conv3, _SliceView(cat2, 0, 128), t3); run(self.conv3_1, t3, cat3, 0)
        t4, cat4 = nb(H // 16, W // 16, 512), nb(H // 16, W // 16, 770)
        run(self.conv4, _SliceView(cat3, 0, 256), t4); run(self.conv4_1, t4, cat4, 0)
        t5, cat5 = nb(H // 32, W // 32, 512), nb(H // 32, W // 32, 1026)
        run(self.conv5, _SliceView(cat4, 0, 512), t5); run(self.conv5_1, t5, cat5, 0)
        t6, c6 = nb(H // 64, W // 64, 1024), nb(H // 64, W // 64, 1024)
        run(self.conv6, _SliceView(cat5, 0, 512), t6); run(self.conv6_1, t6, c6)
        return _Decoder.run(self, run, c6, cat5, cat4, cat3, cat2)


class FlowNetSD(nn.Module):
    def __init__(self, with_bn=False):
        super().__init__()
        assert not with_bn
        self.with_bn = with_bn
        self.conv0 = conv(6, 64)
        self.conv1 = conv(64, 64, stride=2)
        self.conv1_1 = conv(64, 128)
        self.conv2 = conv(128, 128, stride=2)
        self.conv2_1 = conv(128, 128)
        self.conv3 = conv(128, 256, stride=2)
        self.conv3_1 = conv(256, 256)
        self.conv4 = conv(256, 512, stride=2)
        self.conv4_1 = conv(512, 512)
        self.conv5 = conv(512, 512, stride=2)
        self.conv5_1 = conv(512, 512)
        self.conv6 = conv(512, 1024, stride=2)
        self.conv6_1 = conv(1024, 1024)
        self.deconv5 = deconv(1024, 512)
        self.deconv4 = deconv(1026, 256)
        self.deconv3 = deconv(770, 128)
        self.deconv2 = deconv(386, 64)
        self.inter_conv5 = conv(1026, 512, with_relu=False)
        self.inter_conv4 = conv(770, 256, with_relu=False)
        self.inter_conv3 = conv(386, 128, with_relu=False)
        self.inter_conv2 = conv(194, 64, with_relu=False)
        self.predict_flow6 = predict_flow(1024)
        self.predict_flow5 = predict_flow(512)
        self.predict_flow4 = predict_flow(256)
        self.predict_flow3 = predict_flow(128)
        self.predict_flow2 = predict_flow(64)
        for a, b in ((6, 5), (5, 4), (4, 3), (3, 2)):
            setattr(self, 'upsampled_flow%d_to_%d' % (a, b), nn.ConvTranspose2d(2, 2, 4, 2, 1))
        self.upsample1 = nn.Upsample(scale_factor=4, mode='bilinear')
        _xavier_init(self)

    def run(self, run, x):
        """x: _Buf [B,H,W,6] (FlowNetSD.py:60-103)."""
        dev, B, H, W = x.t.device, x.B, x.H, x.W
        nb = lambda h, w, c: _Buf(B, h, w, c, dev)
        c0 = nb(H, W, 64)
        run(self.conv0, x, c0)
        t1, c1 = nb(H // 2, W // 2, 64), nb(H // 2, W // 2, 128)
        run(self.conv1, c0, t1); run(self.conv1_1, t1, c1)
        t2, cat2 = nb(H // 4, W // 4, 128), nb(H // 4, W // 4, 194)
        run(self.conv2, c1, t2); run(self.conv2_1, t2, cat2, 0)
        t3, cat3 = nb(H // 8, W // 8, 256), nb(H // 8, W // 8, 386)
        run(self.conv3, _SliceView(cat2, 0, 128), t3); run(self.conv3_1, t3, cat3, 0)
        t4, cat4 = nb(H // 16, W // 16, 512), nb(H // 16, W // 16, 770)
        run(self.conv4, _SliceView(cat3, 0, 256), t4); run(self.conv4_1, t4, cat4, 0)
        t5, cat5 = nb(H // 32, W // 32, 512), nb(H // 32, W // 32, 1026)
        run(self.conv5, _SliceView(cat4, 0, 512), t5); run(self.conv5_1, t5, cat5, 0)
        t6, c6 = nb(H // 64, W // 64, 1024), nb(H // 64, W // 64, 1024)
        run(self.conv6, _SliceView(cat5, 0, 512), t6); run(self.conv6_1, t6, c6)
        return _Decoder.run(self, run, c6, cat5, cat4, cat3, cat2, inter=True)


class FlowNetFusion(nn.Module):
    def __init__(self, with_bn=False):
        super().__init__()
        assert not with_bn
        self.with_bn = with_bn
        self.conv0 = conv(11, 64)
        self.conv1 = conv(64, 64, stride=2)
        self.conv1_1 = conv(64, 128)
        self.conv2 = conv(128, 128, stride=2)
        self.conv2_1 = conv(128, 128)
        self.deconv1 = deconv(128, 32)
        self.deconv0 = deconv(162, 16)
        self.inter_conv1 = conv(162, 32, with_relu=False)
        self.inter_conv0 = conv(82, 16, with_relu=False)
        self.predict_flow2 = predict_flow(128)
        self.predict_flow1 = predict_flow(32)
        self.predict_flow0 = predict_flow(16)
        self.upsampled_flow2_to_1 = nn.ConvTranspose2d(2, 2, 4, 2, 1)
        self.upsampled_flow1_to_0 = nn.ConvTranspose2d(2, 2, 4, 2, 1)
        _xavier_init(self)

    def run(self, run, x):
        """x: _Buf [B,H,W,11] -> full-resolution flow _Buf [B,H,W,2] (FlowNetFusion.py:43-64)."""
        dev, B, H, W = x.t.device, x.B, x.H, x.W
        nb = lambda h, w, c: _Buf(B, h, w, c, dev)
        cat0 = nb(H, W, 82)
        run(self.conv0, x, cat0, 0)
        t1, cat1 = nb(H // 2, W // 2, 64), nb(H // 2, W // 2, 162)
        run(self.conv1, _SliceView(cat0, 0, 64), t1); run(self.conv1_1, t1, cat1, 0)
        t2, c2 = nb(H // 4, W // 4, 128), nb(H // 4, W // 4, 128)
        run(self.conv2, _SliceView(cat1, 0, 128), t2); run(self.conv2_1, t2, c2)
        f2 = nb(H // 4, W // 4, 2)
        run(self.predict_flow2, c2, f2)
        run(self.upsampled_flow2_to_1, f2, cat1, 160)
        run(self.deconv1, c2, cat1, 128)
        i1, f1 = nb(H // 2, W // 2, 32), nb(H // 2, W // 2, 2)
        run(self.inter_conv1, cat1, i1); run(self.predict_flow1, i1, f1)
        run(self.upsampled_flow1_to_0, f1, cat0, 80)
        run(self.deconv0, cat1, cat0, 64)
        i0, f0 = nb(H, W, 16), nb(H, W, 2)
        run(self.inter_conv0, cat0, i0); run(self.predict_flow0, i0, f0)
        return f0


class FlowNet2(nn.Module):
    """FlowNetC -> warp -> FlowNetS -> warp -> FlowNetS, || FlowNetSD, -> FlowNetFusion (flownet2.py:65-149)."""

    def __init__(self, with_bn=False, fp16=False, rgb_max=255., div_flow=20., grads=None, upsample_align_corners=False):
        """Signature of flownet2.py:12-17 plus ``upsample_align_corners``: the two bilinear x4 up-samplings
        (``nn.Upsample(scale_factor=4, mode='bilinear')``, flownet2.py:28,34) meant align_corners=True under the PyTorch 0.3 the
        authors ran (README.md:10,64) and mean align_corners=False under every torch >= 0.4 -- which is what importing the
        reference today computes and therefore the default; pass True to reproduce the published checkpoint's behaviour."""
        super().__init__()
        if with_bn or fp16:
            raise NotImplementedError('VEC_VAD instantiates FlowNet2() with_bn=False, fp16=False (calc_optical_flow.py:15)')
        self.upsample_align_corners = bool(upsample_align_corners)
        self.with_bn, self.div_flow, self.rgb_max = with_bn, div_flow, rgb_max
        self.grads = {} if grads is None else grads
        self.channelnorm = ChannelNorm()
        self.flownetc = FlowNetC(with_bn=with_bn, fp16=fp16)
        self.upsample1 = nn.Upsample(scale_factor=4, mode='bilinear')
        self.resample1 = Resample2d()
        self.flownets_1 = FlowNetS(with_bn=with_bn)
        self.upsample2 = nn.Upsample(scale_factor=4, mode='bilinear')
        self.resample2 = Resample2d()
        self.flownets_2 = FlowNetS(with_bn=with_bn)
        self.flownets_d = FlowNetSD(with_bn=with_bn)
        self.upsample3 = nn.Upsample(scale_factor=4, mode='nearest')
        self.upsample4 = nn.Upsample(scale_factor=4, mode='nearest')
        self.resample3 = Resample2d()
        self.resample4 = Resample2d()
        self.flownetfusion = FlowNetFusion(with_bn=with_bn)
        _xavier_init(self)
        self._runner = None
        self._graphs = {}
        self._pool = _Pool()
        self._side = {}          # device -> the second stream FlowNetSD runs on

    @torch.no_grad()
    def forward(self, inputs):
        """inputs [B,3,2,H,W] fp32 in 0..rgb_max (H, W multiples of 64) -> flow [B,2,H,W]."""
        if not inputs.is_cuda:
            raise L.VecVadHipError('FlowNet2 runs on the GPU only (no CPU fallback)')
        if inputs.shape[3] % 64 or inputs.shape[4] % 64:
            raise ValueError('H and W must be multiples of 64 (the reference fails with a cat size mismatch otherwise)')
        if self._runner is None:
            self._runner = _Runner()
        run = self._runner
        inputs = inputs.float()
        self._pool.begin((tuple(inputs.shape), str(inputs.device)))
        _ACTIVE_POOL[0] = self._pool
        try:
            return self._forward(run, inputs)
        finally:
            _ACTIVE_POOL[0] = None
            self._pool.end()

    def _forward(self, run, inputs):
        """flownet2.py:65-149 as launches: vv_flownet_prep (mean / normalise / split / cat), the five conv stacks, and one
        packing launch in front of each refinement network (x4 up-sampling + Resample2d + ChannelNorm + torch.cat fused,
        written straight into the consumer's NHWC buffer).  No ATen kernel runs between the input and the returned flow."""
        lib = L.lib()
        dev = inputs.device
        st = torch.cuda.current_stream(dev).cuda_stream
        inputs = inputs.contiguous()
        B, _, _, H, W = inputs.shape
        x6, img0, img1 = _Buf(B, H, W, 6, dev), _Buf(B, H, W, 3, dev), _Buf(B, H, W, 3, dev)
        ws = self._pool.take((int(lib.vv_flownet_prep_workspace_bytes(B)) // 4,), dev)
        L.check(lib.vv_flownet_prep(inputs.data_ptr(), B, H, W, float(self.rgb_max), ws.data_ptr(), ws.numel() * 4,
                                    x6.t.data_ptr(), img0.t.data_ptr(), img1.t.data_ptr(), st), 'flownet_prep')
        bil = 2 if self.upsample_align_corners else 1

        def warp_pack(flow2):
            """[x, resample(img1, flow), flow / div_flow, |img0 - warped|], flow = upsample x4(flow2 * div_flow) (flownet2.py:76-86)."""
            out = _Buf(B, H, W, 12, dev)
            L.check(lib.vv_warp_pack12(x6.t.data_ptr(), img1.t.data_ptr(), flow2.t.data_ptr(), flow2.cs, B, H, W, bil,
                                       float(self.div_flow), float(self.div_flow), out.t.data_ptr(), st), 'warp_pack12')
            return out

        # FlowNetSD reads only x (flownet2.py:96-98): it runs on a second HIP stream beside the FlowNetC -> S1 -> S2 chain and
        # joins in front of the fusion network.  At one image pair most layers below H/8 are a single wave of workgroups (or a
        # split-K launch sized to one chip fill): two independent sub-networks in flight fill the CUs such launches leave idle.
        # Same kernels, same per-kernel summation order: the result is bit-identical to the serial schedule (VV_FN2_OVERLAP=0).
        # Forked when FlowNetC is done (VV_FN2_SD_AT=1): beside FlowNetC's full-chip front end it only contends, beside S1 -> S2 it
        # fills gaps.  Measured per forward: serial 6.00 ms; forked at the start 5.83, after FlowNetC 5.71, after S1 5.74.
        # (Finer forks -- each level's flow head beside its deconv -- measured +-0; NESTED forks, a fork on the already forked
        # FlowNetSD stream, crash hipStreamEndCapture in this ROCm, so there is one level: this one and FlowNetC's second tower.)
        main = torch.cuda.current_stream(dev)
        side = None
        if os.environ.get('VV_FN2_OVERLAP', '1') != '0':
            side = self._side.get(dev)
            if side is None:
                side = self._side[dev] = torch.cuda.Stream(device=dev)
            at = int(os.environ.get('VV_FN2_SD_AT', '1'))

        def fork_sd(point):
            if side is not None and at == point:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    return self.flownets_d.run(run, x6)
            return None
        sd_flow2 = fork_sd(0)
        c_flow2 = self.flownetc.run(run, img0, img1)
        sd_flow2 = fork_sd(1) or sd_flow2
        s1_flow2 = self.flownets_1.run(run, warp_pack(c_flow2))
        sd_flow2 = fork_sd(2) or sd_flow2
        s2_flow2 = self.flownets_2.run(run, warp_pack(s1_flow2))
        if side is None:
            sd_flow2 = self.flownets_d.run(run, x6)
        else:
            main.wait_stream(side)
        cat3 = _Buf(B, H, W, 11, dev)
        L.check(lib.vv_fusion_pack11(x6.t.data_ptr(), img1.t.data_ptr(), s2_flow2.t.data_ptr(), s2_flow2.cs, sd_flow2.t.data_ptr(),
                                     sd_flow2.cs, B, H, W, float(self.div_flow), cat3.t.data_ptr(), st), 'fusion_pack11')
        return self.flownetfusion.run(run, cat3).nchw(0, 2)

    @torch.no_grad()
    def forward_graphed(self, inputs):
        """Same result as forward(), replayed from a hipGraph captured once per input shape: the ~250 launches of one
        forward (conv stack, native ops, plumbing) are launch-bound when issued from python one by one."""
        key = tuple(inputs.shape)
        ent = self._graphs.get(key)
        if ent is None:
            static_in = inputs.clone()
            side = torch.cuda.Stream(device=inputs.device)
            side.wait_stream(torch.cuda.current_stream(inputs.device))
            with torch.cuda.stream(side):
                for _ in range(2):               # warm-up: packs the weights, fills the allocator
                    self.forward(static_in)
            torch.cuda.current_stream(inputs.device).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_out = self.forward(static_in)
            ent = (graph, static_in, static_out)
            self._graphs[key] = ent
        graph, static_in, static_out = ent
        static_in.copy_(inputs)
        graph.replay()
        return static_out.clone()
